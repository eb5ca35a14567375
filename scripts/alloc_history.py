#!/usr/bin/env python3
"""Does the allocation history of the process change the search kernel's speed?  The 50M bench ran 174.4 ms per 262144 scans right
after the on-device index build and 166.4 ms in a second process that loaded the same graph from the cache (profiles/r03/pre/).
One process: (A) build in process, search; (B) tear everything down, allocate afresh, load the graph, search; (C) the same as B with
the search workspace reserved BEFORE anything else is allocated."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50_000_000)
    ap.add_argument("--nq", type=int, default=262144)
    ap.add_argument("--L", type=int, default=3)
    ap.add_argument("--rescore", type=int, default=196)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    from pgvectorscale_amd.datagen import DatagenParams, fill_device

    seed = {1_000_000: 3, 10_000_000: 5, 50_000_000: 6}.get(args.n, 3)
    gp = DatagenParams(seed=seed, dim=768)
    path = "/tmp/alloc_history.graph"

    def make(load):
        ctx = P.Context(0)
        ix = P.DiskAnnIndex.alloc(ctx, n=args.n, dim_full=768, num_neighbors=50, distance_type=P.VS_L2)
        vp, _ = ix.array(_lib.ARR_VECS)
        fill_device(ctx, gp, 0, args.n, vp)
        ix.refresh_norms()
        ix.sbq_train()
        ix.sbq_quantize_corpus()
        t0 = time.time()
        if load:
            ix.load_graph(path)
        else:
            ix.build_graph(search_list_size=100, max_alpha=1.2)
            ix.save_graph(path)
        print(f"graph {'loaded' if load else 'built'} in {time.time() - t0:.1f}s", flush=True)
        return ctx, ix

    def run(tag, ctx, ix):
        q = ctx.alloc(args.nq * 768 * 4)
        fill_device(ctx, gp, 1 << 40, args.nq, q)
        out = ctx.alloc(args.nq * 10 * 4)
        for _ in range(2):  # (the second warm-up launch runs with the fitted table)
            ix.search_batch_dev(q, args.nq, args.L, args.rescore, 10, out)
            ix.search_batch_dev_finish()
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        for _ in range(args.reps):
            ix.search_batch_dev(q, args.nq, args.L, args.rescore, 10, out)
            ix.search_batch_dev_finish()
        prof = ctx.profile_read(reset=True)
        ids = ctx.download(out, np.empty((args.nq, 10), np.uint32))
        print(f"{tag}: search {prof['search'][0] / prof['search'][1]:.3f} ms  rerank {prof['rerank'][0] / prof['rerank'][1]:.3f} ms  "
              f"checksum {int(ids.astype(np.uint64).sum())}", flush=True)
        ctx.free(q)
        ctx.free(out)

    ctx, ix = make(False)
    run("A  built in process          ", ctx, ix)
    run("A' again                      ", ctx, ix)
    ix.close()
    ctx.close()
    ctx, ix = make(True)
    run("B  fresh handles, graph loaded", ctx, ix)
    ix.close()
    ctx.close()
    os.remove(path)


if __name__ == "__main__":
    main()
