#!/usr/bin/env python3
"""K5 flat SBQ scan throughput: device-resident corpus (generated, trained and quantised in HBM, no graph), timed with
the library's HIP-event profile.  Algorithmic bytes per launch = tiles * n * 8 * code_stride.

  python scripts/perf_scan.py --n 10000000 --nq 8,64,512
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", default="8,64")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    from pgvectorscale_amd.datagen import DatagenParams, fill_device

    ctx = P.Context(0)
    ix = P.DiskAnnIndex.alloc(ctx, n=args.n, dim_full=args.dim, num_neighbors=50, distance_type=P.VS_L2)
    gp = DatagenParams(seed=3, dim=args.dim)
    vp, _ = ix.array(_lib.ARR_VECS)
    t0 = time.time()
    fill_device(ctx, gp, 0, args.n, vp)
    ix.sbq_train()
    ix.sbq_quantize_corpus()
    print(f"corpus ready in {time.time() - t0:.2f}s", flush=True)
    W = ix.desc.words
    cs = W + (W & 1)
    rng = np.random.default_rng(1)
    for nq in [int(x) for x in args.nq.split(",")]:
        q = rng.standard_normal((nq, args.dim)).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        qcodes = ix.quantize(q)
        ctx.profile_enable(True)
        ix.scan_topk(qcodes, args.k)
        ctx.profile_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            ids, ham = ix.scan_topk(qcodes, args.k)
        wall = (time.perf_counter() - t0) / args.reps
        prof = ctx.profile_read(reset=True)
        ms = prof["scan"][0] / prof["scan"][1]
        qt = int(os.environ.get('VS_SCAN_Q', 4 if nq <= 4 else 8))
        tiles = (nq + qt - 1) // qt
        byts = tiles * args.n * 8 * cs
        print(f"nq={nq:5d} tiles={tiles:4d}: k_scan_topk {ms:8.3f} ms/launch  {byts / ms / 1e6:8.1f} GB/s algorithmic "
              f"({byts / ms / 1e6 / 8000:.3f} of 8 TB/s)  wall {wall * 1e3:8.3f} ms  ham[0]={ham[0][:3]}", flush=True)
    ix.close()
    ctx.close()


if __name__ == "__main__":
    main()
