#!/usr/bin/env python3
"""The library's own A/B of the search kernel's launch variants (vs_index_autotune) on a device-manufactured index:
every variant timed on one batch at one operating point, rows / distance bits / counters held to the default's.  No torch, no oracle.

  VS_NO_TORCH=1 python scripts/ab_autotune.py --n 10000000 --nq 262144 --L 3 --rescore 196 --reps 3 > ab.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=262144)
    ap.add_argument("--L", type=int, default=3)
    ap.add_argument("--rescore", type=int, default=196)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    from pgvectorscale_amd import _lib
    if args.lib:
        _lib.LIB_PATH = args.lib
    import pgvectorscale_amd as P
    from pgvectorscale_amd.datagen import DatagenParams, fill_device
    t0 = time.time()
    ctx = P.Context(0)
    ix = P.DiskAnnIndex.alloc(ctx, n=args.n, dim_full=args.dim, num_neighbors=50, distance_type=P.VS_L2)
    gp = DatagenParams(seed=args.seed, dim=args.dim)
    vp, _ = ix.array(_lib.ARR_VECS)
    fill_device(ctx, gp, 0, args.n, vp)
    ix.refresh_norms()
    ix.sbq_train()
    ix.sbq_quantize_corpus()
    tb = time.time()
    ix.build_graph(search_list_size=100, max_alpha=1.2)
    build_s = time.time() - tb
    dq = ctx.alloc(args.nq * args.dim * 4)
    fill_device(ctx, gp, 1 << 40, args.nq, dq)
    ta = time.time()
    rep = ix.autotune(dq, args.nq, args.L, args.rescore, 10, reps=args.reps)
    tune_s = time.time() - ta
    # the work of one step (for GB/s): the same batch once more through the chosen variant
    out = ctx.alloc(args.nq * 10 * 4)
    ix.search_batch_dev(dq, args.nq, args.L, args.rescore, 10, out)
    st = ix.search_batch_dev_finish()
    W, R = ix.desc.words, ix.desc.num_neighbors
    alg = st["visited_nodes"] * 4 * R + st["quantized_distance_comparisons"] * 8 * W
    base = rep[0]
    for e in rep:
        if e["applicable"] and e["search_ms"] > 0:
            e["search_vs_default_pct"] = round(100.0 * (e["search_ms"] / base["search_ms"] - 1.0), 2)
            e["step_vs_default_pct"] = round(100.0 * (e["step_ms"] / base["step_ms"] - 1.0), 2)
            e["search_alg_GBps"] = round(alg / (e["search_ms"] * 1e-3) / 1e9, 1)
            e["frac_of_8TBps"] = round(alg / (e["search_ms"] * 1e-3) / 8e12, 4)
    print(json.dumps({"device": ctx.device_name(), "n": args.n, "dim": args.dim, "scans_per_launch": args.nq, "search_list_size": args.L,
                      "rescore": args.rescore, "reps": args.reps, "chosen": ix.variant(), "alg_bytes_per_launch": int(alg),
                      "visits_per_query": round(st["visited_nodes"] / args.nq, 1),
                      "d_quantized_per_query": round(st["quantized_distance_comparisons"] / args.nq, 1),
                      "build_s": round(build_s, 2), "autotune_s": round(tune_s, 2), "total_s": round(time.time() - t0, 2),
                      "candidates": rep}, indent=1), flush=True)
    ix.close()
    ctx.close()


if __name__ == "__main__":
    main()
