#!/usr/bin/env python3
"""Why is the search kernel ~8 % slower right after an on-device graph build than on the same graph loaded from a file
(round 4's final session: 167.2 ms in the run that built the 50M graph, 153.6 / 154.1 ms in the two runs that loaded it)?
Same work (identical counters), so it is a state of the device or of the process.  One process builds and then times the same
batch (a) at once, (b) after an idle minute (clocks / temperature), (c) through a fresh view of the index, i.e. with a search
workspace allocated NOW (placement of the dedup tables / heap spill arrays), (d) after a 'torch.cuda.empty_cache()'-free
re-allocation of nothing — the control; a second process loads the saved graph and times the batch again (e).

  python scripts/diag_state.py --n 50000000 --phase build   # (a)-(d), writes the graph file
  python scripts/diag_state.py --n 50000000 --phase load    # (e)
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showtemp", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("sclk", "mclk", "Temperature (Sensor junction)", "Temperature (Sensor memory)", "Power"))]
        return " | ".join(keep[:8])
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi: {e!r}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50_000_000)
    ap.add_argument("--nq", type=int, default=262144)
    ap.add_argument("--L", type=int, default=3)
    ap.add_argument("--rescore", type=int, default=195)
    ap.add_argument("--phase", default="build", choices=["build", "load"])
    ap.add_argument("--graph", default="/tmp/diag_graph")
    ap.add_argument("--idle", type=float, default=0.0, help="(b) seconds to idle before timing again (0: skip)")
    ap.add_argument("--pad-mb", type=int, default=0, help="allocate and hold this much device memory BEFORE the index arrays (shifts everything)")
    ap.add_argument("--ws-pad-mb", type=int, default=0, help="allocate and hold this much right before the first search (shifts the workspace only)")
    ap.add_argument("--lib", default=None, help="another build of libvsgpu (docs/experiments/ws_spread.patch: VS_WS_SPREAD_MB)")
    ap.add_argument("--spread", default=None, help="after the first timing: ','-separated WHAT:MB pairs, each timed in this process with the regions of "
                                                   "the persistent grid spread over MB of device memory (WHAT: 1 heap spill, 2 dedup tables, 3 both; needs --lib)")
    ap.add_argument("--placement", type=int, default=0, help="placement study: this many fresh views (see the code)")
    ap.add_argument("--private-slabs", default="", help="','-separated MB: after (c0), fresh views with a slab of their own of these sizes")
    ap.add_argument("--early", action="store_true", help="allocate the query buffers and the whole search workspace right after the index "
                                                          "arrays (one dummy batch on the still empty graph), before anything else")
    args = ap.parse_args()
    import numpy as np  # noqa: F401
    import torch  # noqa: F401
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    if os.environ.get("VS_EMU"):  # (dry run of the control flow on the interpreter)
        _lib.LIB_PATH = os.path.join(ROOT, "tests", "emu", "libvsgpu_emu.so")
    from pgvectorscale_amd.datagen import DatagenParams, fill_device

    ctx = P.Context(0)
    pads = []
    if args.pad_mb:
        pads.append(ctx.alloc(args.pad_mb << 20))
    ix = P.DiskAnnIndex.alloc(ctx, n=args.n, dim_full=768, num_neighbors=50, distance_type=P.VS_L2)
    seed = {1_000_000: 3, 10_000_000: 5, 50_000_000: 6}.get(args.n, 3)
    gp = DatagenParams(seed=seed, dim=768)
    nq, k = args.nq, 10
    q = out = None
    if args.early:
        q = ctx.alloc(nq * 768 * 4)
        out = ctx.alloc(nq * k * 4)
        ix.search_batch_dev(q, nq, args.L, args.rescore, k, out)  # (empty graph: every scan ends at once; the workspace is allocated)
        ix.search_batch_dev_finish()
        print("workspace allocated right after the index arrays (NOTE: this dummy batch also leaves scan statistics for this (L, M) that "
              "under-size the tables of the timed batches below: run the timed batches at another rescore, or ignore their times)", flush=True)
    vp, _ = ix.array(_lib.ARR_VECS)
    fill_device(ctx, gp, 0, args.n, vp)
    ix.refresh_norms()
    ix.sbq_train()
    ix.sbq_quantize_corpus()
    t0 = time.time()
    if args.phase == "build":
        ix.build_graph(search_list_size=100, max_alpha=1.2)
        print(f"graph built in {time.time() - t0:.1f} s; {smi()}", flush=True)
    else:
        ix.load_graph(args.graph)
        print(f"graph loaded in {time.time() - t0:.1f} s; {smi()}", flush=True)
    if q is None:
        q = ctx.alloc(nq * 768 * 4)
        out = ctx.alloc(nq * k * 4)
    fill_device(ctx, gp, 1 << 40, nq, q)
    if args.ws_pad_mb:
        pads.append(ctx.alloc(args.ws_pad_mb << 20))
    print(f"pad {args.pad_mb} MB before the index, {args.ws_pad_mb} MB before the workspace; q at {q.value:#x}, out at {out.value:#x}", flush=True)

    def timed(handle, c, label, reps=4):
        c.profile_enable(True)
        handle.search_batch_dev(q, nq, args.L, args.rescore, k, out)
        handle.search_batch_dev_finish()
        handle.search_batch_dev(q, nq, args.L, args.rescore, k, out)  # (second warm-up: the fitted tables)
        handle.search_batch_dev_finish()
        c.profile_read(reset=True)
        ms = []
        for _ in range(reps):
            handle.search_batch_dev(q, nq, args.L, args.rescore, k, out)
            handle.search_batch_dev_finish()
            p = c.profile_read(reset=True)
            ms.append(p["search"][0] / max(p["search"][1], 1))
        print(f"{label:58s}: search " + " ".join(f"{x:7.2f}" for x in ms) + f" ms   {smi()}", flush=True)

    timed(ix, ctx, "(a) right after the " + ("build" if args.phase == "build" else "load"))
    if args.spread:
        for item in args.spread.split(","):
            what, mb = item.split(":")
            os.environ["VS_WS_SPREAD_WHAT"], os.environ["VS_WS_SPREAD_MB"] = what, mb
            timed(ix, ctx, f"(s) regions spread: what={what} over {mb} MB each")
        os.environ["VS_WS_SPREAD_MB"] = "0"
    if True:
        if args.idle > 0:
            time.sleep(args.idle)
            timed(ix, ctx, f"(b) after {args.idle:.0f} idle seconds")
        q2, out2 = q, out

        def timed_view(label, slab_mb=None, private=False, sleep=0.0):
            # a fresh view = a new context and a new workspace; slab_mb = 0: its hot regions in allocations of their own size (the
            # round-4 library), None: out of the index's slab (the library default)
            if slab_mb is not None:
                os.environ["VS_WS_SLAB_MB"] = str(slab_mb)
            if private:
                os.environ["VS_WS_SLAB_PRIVATE"] = "1"
            if sleep:
                time.sleep(sleep)
            ctx2 = P.Context(0)
            vw = ix.view(ctx2)
            nonlocal q2, out2
            ctx2.profile_enable(True)
            for _ in range(2):
                vw.search_batch_dev(q2, nq, args.L, args.rescore, k, out2)
                vw.search_batch_dev_finish()
            ctx2.profile_read(reset=True)
            ms = []
            for _ in range(2 if args.placement else 4):
                vw.search_batch_dev(q2, nq, args.L, args.rescore, k, out2)
                vw.search_batch_dev_finish()
                p = ctx2.profile_read(reset=True)
                ms.append(p["search"][0] / max(p["search"][1], 1))
            print(f"{label:58s}: search " + " ".join(f"{x:7.2f}" for x in ms) + f" ms   {smi()}", flush=True)
            vw.close()
            ctx2.close()
            os.environ.pop("VS_WS_SLAB_MB", None)
            os.environ.pop("VS_WS_SLAB_PRIVATE", None)

        timed_view("(c) through a fresh view (regions out of the index's slab)")
        timed_view("(c0) through a fresh view, VS_WS_SLAB_MB=0 (own allocations)", slab_mb=0)
        if args.placement:
            # placement study: many fresh views, each with its own slab (or own allocations), pads of odd sizes held in between so that
            # every trial lands somewhere else; VS_WS_DEBUG prints the addresses of the two hot arrays next to each timing
            import random
            rnd = random.Random(12345)
            os.environ["VS_WS_DEBUG"] = "1"
            held = []
            for t in range(args.placement):
                kind = t % 4
                pad_mb = rnd.choice([0, 3, 64, 200, 1000, 2500])
                if pad_mb:
                    held.append(ctx.alloc(pad_mb << 20))
                if kind == 0:
                    timed_view(f"(t{t}) pad {pad_mb} MB, own allocations", slab_mb=0)
                elif kind == 1:
                    timed_view(f"(t{t}) pad {pad_mb} MB, private slab 1024 MB", slab_mb=1024, private=True)
                elif kind == 2:
                    os.environ["VS_WS_SLAB_WHAT"] = "1"
                    timed_view(f"(t{t}) pad {pad_mb} MB, private slab 1024 MB, tables only", slab_mb=1024, private=True)
                    os.environ.pop("VS_WS_SLAB_WHAT")
                else:
                    os.environ["VS_WS_SLAB_WHAT"] = "2"
                    timed_view(f"(t{t}) pad {pad_mb} MB, private slab 1024 MB, heap only", slab_mb=1024, private=True)
                    os.environ.pop("VS_WS_SLAB_WHAT")
            os.environ.pop("VS_WS_DEBUG")
        for mb in [int(x) for x in args.private_slabs.split(",") if x]:
            timed_view(f"(p) fresh view, private slab of {mb} MB allocated now", slab_mb=mb, private=True)
        if args.private_slabs:
            timed_view("(p) fresh view, private slab of 4096 MB after 15 idle s", slab_mb=4096, private=True, sleep=15.0)
        timed(ix, ctx, "(d) the first handle again")
        if args.phase == "build":
            ix.save_graph(args.graph)
    ix.close()
    ctx.close()


if __name__ == "__main__":
    main()
