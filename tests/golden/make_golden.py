#!/usr/bin/env python3
"""Generates the committed golden fixtures tests/golden/*.npz.

The reference (Rust + PGRX + PostgreSQL) cannot be built or imported in this environment, so its own outputs cannot be
recorded; the fixtures are produced by the CPU oracle (oracle/vs_oracle.cpp, pinned against the reference's KATs by
tests/test_oracle_kat.py) on small seeded indexes.  They freeze the oracle's behaviour (tests/test_golden.py checks both
the oracle and the HIP path against them), so a change to either side that alters tie order, dedup semantics, the rescore
window or the f32 accumulation order shows up as a diff against a file in git.

  python tests/golden/make_golden.py        # rewrites the .npz files (do this only for a deliberate semantic change)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from helpers import TestIndex  # noqa: E402

CASES = {
    # name: (index kwargs, query kind, L, rescore, k, with label keys)
    "l2_labels_deleted": (dict(n=600, dim_full=64, bits=2, R=24, distance=1, seed=41, kind="gauss", L_build=50, n_labels=5,
                               deleted_frac=0.08), "gauss", 40, 20, 10, True),
    "cosine_matryoshka": (dict(n=500, dim_full=96, dim_index=64, bits=2, R=20, distance=0, seed=42, kind="gauss", L_build=50),
                          "gauss", 50, 25, 10, False),
    "ip_one_bit": (dict(n=400, dim_full=200, bits=1, R=16, distance=2, seed=43, kind="gauss", L_build=40), "gauss", 30, 0, 8,
                   False),
}


def main():
    for name, (kw, qkind, L, rescore, k, use_labels) in CASES.items():
        ti = TestIndex(**kw)
        q = ti.queries(24, seed=7, kind=qkind)
        qlabels = None
        if use_labels:
            rng = np.random.default_rng(3)
            qlabels = [sorted(set(int(v) for v in rng.integers(1, 6, int(rng.integers(1, 3))))) for _ in range(len(q))]
        m = rescore + k + 3
        s_ids, s_ham, s_st = ti.oracle.stream_batch(q, L=L, m=m, qlabels=qlabels)
        r_ids, r_dist, r_st = ti.oracle.search_batch(q, L=L, rescore=rescore, k=k, qlabels=qlabels)
        out = dict(
            codes=ti.codes, nbrs=ti.nbrs, tids=ti.tids, vecs=ti.vecs, mean=ti.mean,
            m2=ti.m2 if ti.m2 is not None else np.zeros(0, np.float32), count=np.uint64(ti.count),
            bits=np.uint32(ti.bits), dim_index=np.uint32(ti.dim_index), R=np.uint32(ti.R), distance=np.uint32(ti.distance),
            start=np.uint32(ti.start), queries=q, L=np.uint32(L), rescore=np.uint32(rescore), k=np.uint32(k), m=np.uint32(m),
            stream_ids=s_ids, stream_ham=s_ham, rows_ids=r_ids, rows_dist=r_dist,
            visited_nodes=np.uint64(s_st["visited_nodes"]),
            quantized_distance_comparisons=np.uint64(s_st["quantized_distance_comparisons"]),
        )
        if ti.label_off is not None:
            out["label_off"] = ti.label_off
            out["label_val"] = ti.label_val
            keys = sorted(ti.label_starts)
            out["label_start_labels"] = np.array(keys, np.int16)
            out["label_start_nodes"] = np.array([ti.label_starts[x] for x in keys], np.uint32)
        if qlabels is not None:
            off = np.zeros(len(q) + 1, np.uint32)
            vals = []
            for i, l in enumerate(qlabels):
                vals.extend(l)
                off[i + 1] = len(vals)
            out["qlabel_off"] = off
            out["qlabel_val"] = np.array(vals, np.int16)
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
