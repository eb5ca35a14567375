#!/usr/bin/env python3
"""Latency of ONE scan through the amrescan / amgettuple cursor (vs_rescan + vs_gettuple), the way a single backend sees it:
milliseconds to the first row and to the first 10 / 100 / 1000 rows, and the device work behind them (launches, visits), next to
the batched path's throughput for scale.  python scripts/cursor_latency.py --n 10000000"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--L", type=int, default=100)
    ap.add_argument("--rescore", type=int, default=50)
    ap.add_argument("--queries", type=int, default=32)
    ap.add_argument("--graph-cache", default=None)
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    from pgvectorscale_amd.datagen import DatagenParams, fill_device, rows_numpy

    ctx = P.Context(0)
    ix = P.DiskAnnIndex.alloc(ctx, n=args.n, dim_full=768, num_neighbors=50, distance_type=P.VS_L2)
    seed = {1_000_000: 3, 10_000_000: 5, 50_000_000: 6}.get(args.n, 3)
    gp = DatagenParams(seed=seed, dim=768)
    vp, _ = ix.array(_lib.ARR_VECS)
    fill_device(ctx, gp, 0, args.n, vp)
    ix.refresh_norms()
    ix.sbq_train()
    ix.sbq_quantize_corpus()
    cache = args.graph_cache and f"{args.graph_cache}.{args.n}x768.l2.b{ix.desc.bits}.R50.L100.s{seed}"
    if cache and os.path.exists(cache):
        ix.load_graph(cache)
    else:
        ix.build_graph(search_list_size=100, max_alpha=1.2)
        if cache:
            ix.save_graph(cache)
    q = rows_numpy(gp, 1 << 40, args.queries)
    scan = ix.beginscan()
    marks = (1, 10, 100, 1000)
    t_at = {m: [] for m in marks}
    work = []
    for i in range(args.queries):
        ctx.sync()
        t0 = time.perf_counter()
        scan.rescan(q[i], search_list_size=args.L, rescore=args.rescore)
        got = 0
        while got < marks[-1]:
            r = scan.gettuple()
            if r is None:
                break
            got += 1
            if got in t_at:
                t_at[got].append((time.perf_counter() - t0) * 1e3)
        work.append(scan.work())
    scan.endscan()
    skip = 2  # (the first scans pay allocations)
    print(f"cursor latency, {args.n} x 768, L={args.L} rescore={args.rescore}, {args.queries - skip} scans (median / max ms):")
    for m in marks:
        v = np.array(t_at[m][skip:])
        print(f"  first {m:5d} rows: {np.median(v):8.3f} / {v.max():8.3f} ms")
    w = work[-1]
    print(f"  one scan of {marks[-1]} rows: {w['launches']} launches, {w['visited_nodes']} visits, {w['quantized_distance_comparisons']} Hamming evaluations, "
          f"{w['retries']} restarts")
    ix.close()
    ctx.close()


if __name__ == "__main__":
    main()
