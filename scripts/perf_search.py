#!/usr/bin/env python3
"""Kernel-level A/B harness for the search path: builds one device-resident index, then times the batched scan pipeline
under different on-chip state splits (VS_HL = heap entries in LDS, VS_LH = LDS dedup slots) in ONE process.

  python scripts/perf_search.py --n 1000000 --nq 16384 --configs VS_FAST=0,VS_FAST=1:VS_F_LH=2048
"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--nq", type=int, default=16384)
    ap.add_argument("--L", type=int, default=100)
    ap.add_argument("--rescore", type=int, default=50)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--configs", default="VS_FAST=1")
    ap.add_argument("--kind", default="clustered", choices=["clustered", "hard"])
    ap.add_argument("--graph-cache", default=None, help="neighbor-array file (see bench.py --graph-cache)")
    ap.add_argument("--scan", type=int, default=0, help="also run the flat SBQ scan (K5) with this many queries (PMC calibration)")
    ap.add_argument("--order", default="asis", choices=["asis", "cluster", "cluster_xcd"],
                    help="experiment: order of the queries inside the batch — as generated, sorted by the cluster they were drawn "
                         "from (scans that run at the same time touch the same part of the graph), or sorted and dealt to the 8 XCDs "
                         "by cluster (block b runs on XCD b %% 8)")
    ap.add_argument("--host", default=None, help="also time the host-buffer entry point (vs_search_batch: PCIe inclusive) over all --nq queries "
                                                 "under these ','-separated configs, e.g. VS_HOST_CHUNKS=1,VS_HOST_CHUNKS=4")
    ap.add_argument("--lib", default=None, help="time this libvsgpu build instead of pgvectorscale_amd/libvsgpu.so (scripts/ab_branch.sh)")
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    from pgvectorscale_amd.datagen import DatagenParams, fill_device
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)

    ctx = P.Context(0)
    ix = P.DiskAnnIndex.alloc(ctx, n=args.n, dim_full=args.dim, num_neighbors=50, distance_type=P.VS_L2)
    seed = {1_000_000: 3, 10_000_000: 5, 50_000_000: 6}.get(args.n, 3)  # same corpora as bench.py
    gp = DatagenParams(seed=seed, dim=args.dim) if args.kind == "clustered" else \
        DatagenParams(seed=seed, dim=args.dim, latent_dim=64, n_clusters=16, intra_pct=100, noise_pct=40)
    vp, _ = ix.array(_lib.ARR_VECS)
    t0 = time.time()
    fill_device(ctx, gp, 0, args.n, vp)
    ix.refresh_norms()
    ix.sbq_train()
    ix.sbq_quantize_corpus()
    # same file name as bench.py's graph_cache_path() for an explicit prefix
    cache = args.graph_cache and f"{args.graph_cache}.{args.n}x{args.dim}.l2.b{ix.desc.bits}.R50.L100.s{seed}"
    if cache and os.path.exists(cache) and args.kind == "clustered":
        ix.load_graph(cache)
    else:
        ix.build_graph(search_list_size=100, max_alpha=1.2)
        if cache and args.kind == "clustered":
            ix.save_graph(cache)
    print(f"index ready in {time.time() - t0:.2f}s", flush=True)
    nq, k = args.nq, args.k
    q = ctx.alloc(nq * args.dim * 4)
    fill_device(ctx, gp, 1 << 40, nq, q)
    if args.order != "asis":
        from pgvectorscale_amd.datagen import _hash
        qh = ctx.download(q, np.empty((nq, args.dim), np.float32))
        cl = (_hash(gp.seed, np.arange((1 << 40), (1 << 40) + nq, dtype=np.uint64), np.uint64(0)) % np.uint64(gp.n_clusters)).astype(np.int64)
        order = np.argsort(cl, kind="stable")
        if args.order == "cluster_xcd":
            lists = [order[(cl[order] % 8) == x] for x in range(8)]
            m = max(len(l) for l in lists)
            dealt = np.full((m, 8), -1, np.int64)
            for x, l in enumerate(lists):
                dealt[:len(l), x] = l
            order = dealt.reshape(-1)
            order = order[order >= 0]
        ctx.upload(q, np.ascontiguousarray(qh[order]))
        print(f"queries ordered: {args.order}", flush=True)
    out = ctx.alloc(nq * k * 4)
    W, R = ix.desc.words, ix.desc.num_neighbors
    ref_ids = None
    for cfg in args.configs.split(","):
        # a config is a ':'-separated list of NAME=VALUE environment overrides read by libvsgpu (VS_FAST, VS_F_LH, VS_F_HL,
        # VS_F_VCAP for the fast kernel; VS_HL, VS_LH, VS_G0 for the general one)
        # (VARIANT=<name>: not an environment variable — the launch variant of vs_index_autotune by name)
        for kv in cfg.split(":"):
            if kv:
                k_, v_ = kv.split("=")
                if k_ == "VARIANT":
                    ix.set_variant(v_)
                elif k_ == "NQ":  # scans per launch of this and the following configs (<= --nq)
                    nq = min(int(v_), args.nq)
                else:
                    os.environ[k_] = v_
        ctx.profile_enable(True)
        ix.search_batch_dev(q, nq, args.L, args.rescore, k, out)
        ix.search_batch_dev_finish()
        ctx.profile_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            ix.search_batch_dev(q, nq, args.L, args.rescore, k, out)
            st = ix.search_batch_dev_finish()
        wall = (time.perf_counter() - t0) / args.reps
        prof = ctx.profile_read(reset=True)
        ids = ctx.download(out, np.empty((args.nq, k), np.uint32))[:nq]
        same = "ref" if ref_ids is None else str(bool((ids == ref_ids[:nq]).all()))
        if ref_ids is None:
            ref_ids = ids
        ms = prof["search"][0] / max(prof["search"][1], 1)
        fb = prof["search_fallback"][0] / max(prof["search_fallback"][1], 1)
        bytes_ = st["visited_nodes"] * 4 * R + st["quantized_distance_comparisons"] * 8 * W
        print(f"{cfg:40s}: search {ms:8.3f} ms (+fallback {fb:6.3f} ms, {st['fallback_scans']} scans)  rerank "
              f"{prof['rerank'][0] / prof['rerank'][1]:.3f} ms  wall {wall * 1e3:8.3f} ms "
              f"-> {nq / wall:10.0f} QPS  {bytes_ / (ms + fb) / 1e6:7.1f} GB/s alg  visits/q {st['visited_nodes'] / nq:.1f} "
              f"dq/q {st['quantized_distance_comparisons'] / nq:.1f}  same_ids={same}", flush=True)
    if args.host:
        qh = ctx.download(q, np.empty((args.nq, args.dim), np.float32))
        ref = None
        for cfg in args.host.split(","):
            for kv in cfg.split(":"):
                if kv:
                    k_, v_ = kv.split("=")
                    os.environ[k_] = v_
            ix.search_batch(qh, search_list_size=args.L, rescore=args.rescore, k=k)
            t0 = time.perf_counter()
            for _ in range(args.reps):
                hi, _, hd, _ = ix.search_batch(qh, search_list_size=args.L, rescore=args.rescore, k=k)
            wall = (time.perf_counter() - t0) / args.reps
            same = "ref" if ref is None else str(bool((hi == ref).all()))
            if ref is None:
                ref = hi
            print(f"host {cfg:35s}: wall {wall * 1e3:8.3f} ms -> {args.nq / wall:10.0f} QPS (PCIe inclusive)  same_ids={same}", flush=True)
    if args.scan:
        os.environ.setdefault("VS_SCAN_Q", "4")
        rng = np.random.default_rng(1)
        qs = rng.standard_normal((args.scan, args.dim)).astype(np.float32)
        qs /= np.linalg.norm(qs, axis=1, keepdims=True)
        qcodes = ix.quantize(qs)
        for _ in range(3):
            ix.scan_topk(qcodes, k)
        qt = int(os.environ["VS_SCAN_Q"])
        print(f"scan: nq={args.scan} tiles={(args.scan + qt - 1) // qt} bytes_per_launch={(args.scan + qt - 1) // qt * args.n * 8 * (W + (W & 1))}", flush=True)
    ix.close()
    ctx.close()


if __name__ == "__main__":
    main()
