#!/usr/bin/env python3
"""Random 24-word-code geometries (768 x 2 bit / 1536 x 1 bit: the headline instantiations of k_search_fast) through the queued kernel
variants — two-row gather (VS_F_MINW=5) with and without the written-bucket bitmap
(VS_F_VIRGIN=1) — against the oracle: SBQ stream (ids + Hamming distances) and work counters exactly.  scripts/fuzz_emu.py draws
its dimensions at random and almost never lands on these instantiations.

  make -C tests/emu && python scripts/fuzz_variants.py --cases 60 [--seed 1] [--gpu]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

VARIANTS = [{"VS_F_MINW": "5"}, {"VS_F_MINW": "5", "VS_F_VIRGIN": "1"}, {"VS_F_VIRGIN": "1"},
            # the bucket bitmap on sparser tables, cleared tables (round 3's default)
            {"VS_F_VIRGIN": "1", "VS_F_GCAP": "16384"}, {"VS_F_VIRGIN": "0"},
            # occupancy bit per slot of the dedup table (VS_F_VIRGIN=2), on a fitted and on a tight table
            {"VS_F_VIRGIN": "2"}, {"VS_F_VIRGIN": "2", "VS_F_GCAP": "1536"},
            # 16-bit entries in buckets of eight + overflow table (VS_F_VIRGIN=3): fitted, tight (full buckets, overflow inserts and
            # lookups), tight near the load limit (second attempts)
            {"VS_F_VIRGIN": "3", "VS_F_MINW": "7"}, {"VS_F_VIRGIN": "3", "VS_F_MINW": "7", "VS_F_GCAP": "2048"},  # ... the lean 7-wave layout
            {"VS_F_VIRGIN": "3"}, {"VS_F_VIRGIN": "3", "VS_F_GCAP": "2048"}, {"VS_F_VIRGIN": "3", "VS_F_GCAP": "1024", "VS_F_GLOAD_PCT": "90"},
            # ... as long lists run them since round 6 (taken although they cost resident scans, heap top 255)
            {"VS_F_VIRGIN": "3", "VS_F_SLOTMAP_FORCE": "1", "VS_F_HL": "255"}]
KNOBS = sorted({k for v in VARIANTS for k in v})
COUNTERS = ("visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "node_reads", "next_calls")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--gpu", action="store_true", help="libvsgpu.so on a real device instead of the interpreter")
    args = ap.parse_args()
    from pgvectorscale_amd import _lib
    if not args.gpu:
        _lib.LIB_PATH = os.path.join(ROOT, "tests", "emu", "libvsgpu_emu.so")
    import pgvectorscale_amd as P
    from helpers import TestIndex
    ctx = P.Context(0)
    rng = np.random.default_rng(args.seed)
    fails = 0
    for case in range(args.cases):
        n = int(rng.choice([60, 300, 900, 2000]))
        R = int(rng.choice([8, 20, 32, 50, 64, 80]))  # (80: two chunks per list)
        bits = int(rng.choice([1, 2]))
        labels = int(rng.choice([0, 0, 4]))
        L = int(rng.choice([1, 3, 20, 60]))
        m = int(rng.choice([10, 80, 300]))
        kw = dict(n=n, dim_full=768 if bits == 2 else 1536, bits=bits, R=R, distance=int(rng.choice([1, 2])),
                  seed=int(rng.integers(1, 1 << 30)), kind="gauss", L_build=max(R, 20))
        if labels:
            kw.update(n_labels=labels, deleted_frac=0.1)
        ti = TestIndex(**kw)
        ix = ti.upload(ctx)
        q = ti.queries(12, seed=3, kind="gauss")
        ql = None
        if labels:
            ql = [sorted(set(int(v) for v in rng.integers(1, labels + 1, int(rng.integers(1, 3))))) for _ in range(len(q))]
        oi, oh, ost = ti.oracle.stream_batch(q, L=L, m=m, qlabels=ql)
        for v in VARIANTS:
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update({"VS_F_LDS_MAX_INS": "0", "VS_F_VR": "0"})
            os.environ.update(v)
            gi, gh, gst = ix.stream_batch(q, search_list_size=L, m=m, qlabels=ql)
            if not ((gi == oi).all() and (gh == oh).all() and all(gst[c] == ost[c] for c in COUNTERS)):
                fails += 1
                print(f"FAIL case {case} (seed {args.seed}): {kw} L={L} m={m} keys={'yes' if ql else 'no'} variant={v}", flush=True)
        ix.close()
    print(f"{args.cases} cases x {len(VARIANTS)} variants, {fails} failures")
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
