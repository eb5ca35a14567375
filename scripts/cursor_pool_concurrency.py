#!/usr/bin/env python3
"""Many backends streaming at once through the shared-memory server (vs_shm_*): T client threads, each with its own mapping of the
segment and its own scan, each pulling `--rows` rows in chunks of `--chunk` (the first chunk out of a shared OP_SEARCH launch, every
later one an OP_FETCH continuation).  Wall time of 1 / 8 / 64 concurrent scans with the continuations served (a) by a cursor per scan on
the dispatcher thread, (b) on 8 cursor lanes, (c) out of SCAN POOLS — the continuations that arrive in one dispatcher round share one
resumed search launch and one rerank launch (vs_scanpool.cpp).  VERDICT r04: 64 concurrent cursors within 2x the wall time of one.

  python scripts/cursor_pool_concurrency.py --n 1000000 [--rows 1000] [--chunk 16] [--threads 1,8,64]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--L", type=int, default=100)
    ap.add_argument("--rescore", type=int, default=50)
    ap.add_argument("--rows", type=int, default=1000)
    ap.add_argument("--chunk", type=int, default=16)
    ap.add_argument("--threads", default="1,8,64")
    ap.add_argument("--modes", default="cursor,lanes8,pool")
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    if os.environ.get("VS_EMU"):  # (dry run of the control flow on the interpreter)
        _lib.LIB_PATH = os.path.join(ROOT, "tests", "emu", "libvsgpu_emu.so")
    from pgvectorscale_amd.datagen import DatagenParams, fill_device, rows_numpy
    from pgvectorscale_amd.shm_clients import stream_many

    ctx = P.Context(0)
    ix = P.DiskAnnIndex.alloc(ctx, n=args.n, dim_full=768, num_neighbors=50, distance_type=P.VS_L2)
    gp = DatagenParams(seed=3, dim=768)
    vp, _ = ix.array(_lib.ARR_VECS)
    fill_device(ctx, gp, 0, args.n, vp)
    ix.refresh_norms()
    ix.sbq_train()
    ix.sbq_quantize_corpus()
    ix.build_graph(search_list_size=100, max_alpha=1.2)
    tmax = max(int(t) for t in args.threads.split(","))
    q = rows_numpy(gp, 1 << 40, tmax)
    print(f"{args.n} x 768, L={args.L} rescore={args.rescore}, {args.rows} rows per scan in chunks of {args.chunk}")
    ref_rows = None
    modes = {"cursor": dict(cursor_lanes=0, cursor_pool=0), "lanes8": dict(cursor_lanes=8, cursor_pool=0), "pool": dict(cursor_lanes=0, cursor_pool=tmax)}
    for mode in args.modes.split(","):
        name = f"/vs_shm_conc_{os.getpid()}_{mode}"
        srv = P.ShmServer(ix, name, nslots=max(tmax, 4), kmax=args.chunk, max_batch=256, max_wait_us=100, **modes[mode])
        for nt in [int(x) for x in args.threads.split(",")]:
            # the backends are PROCESSES (as under PostgreSQL): threads of this interpreter would take turns on its lock
            st0 = srv.stats()
            wall, out = stream_many(name, _lib.LIB_PATH, [q[t] for t in range(nt)], args.L, args.rescore, args.rows, args.chunk)
            st1 = srv.stats()
            if ref_rows is None:
                ref_rows = out[0]
            same = out[0] == ref_rows  # scan 0 returns the same rows whoever runs next to it, however it is served
            print(f"  {mode:7s} scans {nt:3d}: {wall:9.1f} ms wall  ({wall / nt:8.2f} ms per scan; {st1['tasks'] - st0['tasks']} cursor requests, "
                  f"{st1['batches'] - st0['batches']} shared first-row launches)  rows of scan 0 unchanged: {same}", flush=True)
        srv.close()
    ix.close()
    ctx.close()


if __name__ == "__main__":
    main()
