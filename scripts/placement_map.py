#!/usr/bin/env python3
"""Where in device memory does the search kernel's private state run fast?  (DESIGN.md 7 "State": the same launch takes 156 or 170 ms
depending on where 0.7 GB of dedup tables + heap spill arrays were allocated.)

One process: the 50M index (built or loaded), then as many 1-GB chunks as the device has left are allocated and EACH is (1) probed with
vs_ws_probe (the kernel's private-state request shapes on that chunk, a few ms) and — for a sample of them, fastest / slowest / spread —
(2) used as the workspace slab of a fresh view (vs_index_set_slab) that runs the real 262 144-scan batch.  Prints the probe map in
allocation order and the (probe, search) pairs: does the cheap probe predict the search?"""
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
    ap.add_argument("--rescore", type=int, default=195)
    ap.add_argument("--graph", default="/tmp/diag_graph")
    ap.add_argument("--chunk-mb", type=int, default=1024)
    ap.add_argument("--max-chunks", type=int, default=96)
    ap.add_argument("--sample", type=int, default=10, help="chunks that also run the real batch")
    ap.add_argument("--early-chunks", type=int, default=4, help="chunks allocated BEFORE the index arrays")
    args = ap.parse_args()
    import torch  # noqa: F401
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    if os.environ.get("VS_EMU"):
        _lib.LIB_PATH = os.path.join(ROOT, "tests", "emu", "libvsgpu_emu.so")
    from pgvectorscale_amd.datagen import DatagenParams, fill_device

    ctx = P.Context(0)
    cb = args.chunk_mb << 20
    chunks = [(f"early{i}", ctx.alloc(cb)) for i in range(args.early_chunks)]
    ix = P.DiskAnnIndex.alloc(ctx, n=args.n, dim_full=768, num_neighbors=50, distance_type=P.VS_L2)
    seed = {1_000_000: 3, 10_000_000: 5, 50_000_000: 6}.get(args.n, 3)
    gp = DatagenParams(seed=seed, dim=768)
    vp, _ = ix.array(_lib.ARR_VECS)
    fill_device(ctx, gp, 0, args.n, vp)
    ix.refresh_norms()
    ix.sbq_train()
    ix.sbq_quantize_corpus()
    chunks += [(f"prebuild{i}", ctx.alloc(cb)) for i in range(4)]
    t0 = time.time()
    if os.path.exists(args.graph):
        ix.load_graph(args.graph)
        print(f"graph loaded in {time.time() - t0:.1f} s", flush=True)
    else:
        ix.build_graph(search_list_size=100, max_alpha=1.2)
        print(f"graph built in {time.time() - t0:.1f} s", flush=True)
    nq, k = args.nq, 10
    q = ctx.alloc(nq * 768 * 4)
    out = ctx.alloc(nq * k * 4)
    fill_device(ctx, gp, 1 << 40, nq, q)
    free_b, _ = ctx.mem_info()
    nmore = int(min(args.max_chunks, max(0, (free_b - (24 << 30)) // cb)))  # (24 GB stay free for the views' other buffers)
    for i in range(nmore):
        try:
            chunks.append((f"late{i}", ctx.alloc(cb)))
        except P.VsError:
            break
    print(f"{len(chunks)} chunks of {args.chunk_mb} MB; device free before them {free_b / 1e9:.1f} GB", flush=True)
    probe = {}
    for rep in range(2):
        for name, p in chunks:
            ms = ix.ws_probe_mix(p, cb, 400)  # (the whole request mix; the private-state requests alone — ctx.ws_probe — are flat: s4)
            probe[name] = min(probe.get(name, 1e9), ms)
    print("probe map (allocation order), ms:")
    for name, p in chunks:
        print(f"  {name:10s} {p.value:#014x} {probe[name]:7.3f}")
    order = sorted(chunks, key=lambda c: probe[c[0]])
    pick = []
    ns = max(2, args.sample)
    for j in range(ns):
        pick.append(order[round(j * (len(order) - 1) / (ns - 1))])
    pick += [c for c in chunks if c[0] in ("early0", "prebuild0")]

    def timed_on(label, slab):
        ctx2 = P.Context(0)
        vw = ix.view(ctx2)
        if slab is not None:
            vw.set_slab(slab, cb)
        ctx2.profile_enable(True)
        for _ in range(2):
            vw.search_batch_dev(q, nq, args.L, args.rescore, k, out)
            vw.search_batch_dev_finish()
        ctx2.profile_read(reset=True)
        ms = []
        for _ in range(2):
            vw.search_batch_dev(q, nq, args.L, args.rescore, k, out)
            vw.search_batch_dev_finish()
            pr = ctx2.profile_read(reset=True)
            ms.append(pr["search"][0] / max(pr["search"][1], 1))
        vw.close()
        ctx2.close()
        print(f"  {label:44s} search " + " ".join(f"{x:7.2f}" for x in ms) + " ms", flush=True)
        return min(ms)

    print("real batch with the workspace slab on a chunk:")
    pairs = []
    for name, p in pick:
        s_ms = timed_on(f"{name} (probe {probe[name]:.3f} ms)", p)
        pairs.append((probe[name], s_ms))
    os.environ["VS_WS_SLAB_MB"] = "0"
    timed_on("own allocations (VS_WS_SLAB_MB=0)", None)
    os.environ.pop("VS_WS_SLAB_MB")
    os.environ["VS_WS_SLAB_PRIVATE"] = "1"
    timed_on("library slab of the view's own, 4096 MB", None)
    os.environ.pop("VS_WS_SLAB_PRIVATE")
    if len(pairs) > 2:
        import numpy as np
        a = np.array(pairs)
        print(f"correlation(probe, search) over {len(pairs)} chunks: {np.corrcoef(a[:, 0], a[:, 1])[0, 1]:.3f}; "
              f"probe {a[:, 0].min():.3f}..{a[:, 0].max():.3f} ms, search {a[:, 1].min():.2f}..{a[:, 1].max():.2f} ms")
    if not os.path.exists(args.graph):
        ix.save_graph(args.graph)
    ix.close()
    ctx.close()


if __name__ == "__main__":
    main()
