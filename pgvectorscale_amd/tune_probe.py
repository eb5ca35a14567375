"""Child-process probe of the search kernel's launch variants (vs_index_autotune, include/vsgpu.h).

A variant of k_search_fast is only ever chosen after it has reproduced the default's rows on the caller's own batch — but that
comparison runs in the caller's process, and a kernel that never returns would take the process (a database backend's GPU broker,
a benchmark) with it.  `run()` therefore launches every variant once in a CHILD process first: a small index of the same code
width manufactured on the device (unlabeled and labeled, so both instantiations of every variant run: with and without label
keys), plus three corner legs (a corpus of near-duplicates whose row order the heap's array mechanics decide, scans that exhaust
their graph, dedup tables at their load limit), forced into the table-less regime the large indexes live in, every variant
through vs_index_autotune (each held to the default's rows on every leg), under a hard timeout.  A variant that comes back anything but clean — or a child that has to be killed — is returned as a name to pass to
`DiskAnnIndex.autotune(skip=...)`.

    python -m pgvectorscale_amd.tune_probe [--dim 768] [--n 100000] [--nq 8192] [--device 0] [--lib PATH]

prints one JSON object: {"ok": true, "variants": {name: {"applicable", "rows_identical", "error", "legs", "failed_legs"}},
"legs": {...}, "seconds": s}.

Product code: libvsgpu.so only (no oracle, no CPU path); without a HIP device the child fails and every variant is skipped.
"""
import json
import os
import subprocess
import sys
import time

# the table-less regime with the LDS-ring visited list: what indexes of tens of millions of nodes run in (csrc/vs_api.hip,
# initial_caps), forced here because a small index would otherwise keep its dedup table in LDS, where no variant applies
FORCED_REGIME = {"VS_F_LDS_MAX_INS": "0", "VS_F_VR": "0"}


def _labels(np, n, rng):
    """four labels, one or two per node (sorted sets, AM/labels/mod.rs:15-37) -> CSR"""
    two = rng.random(n) < 0.5
    first = rng.integers(1, 5, n).astype(np.int16)
    second = (first % 4 + 1).astype(np.int16)
    cnt = 1 + two.astype(np.int64)
    off = np.zeros(n + 1, np.uint32)
    np.cumsum(cnt, out=off[1:])
    val = np.empty(int(off[-1]), np.int16)
    val[off[:-1]] = np.minimum(first, np.where(two, second, first))
    val[off[:-1][two] + 1] = np.maximum(first, second)[two]
    return off, val, cnt


def _child(args):
    import numpy as np

    from . import _lib
    if args.lib:
        _lib.LIB_PATH = args.lib
    import pgvectorscale_amd as P
    from .datagen import DatagenParams, fill_device

    t0 = time.time()
    os.environ.update(FORCED_REGIME)
    ctx = P.Context(args.device)
    small = max(300, min(args.n, 900))
    # the legs: what the caller's index looks like (with and without label keys), and the corners the parity tests of the default
    # kernel care about — rows decided by the heap's array mechanics alone (a corpus of near-duplicates: few distinct Hamming
    # distances, thousands of ties), scans that exhaust their graph, dedup tables at their load limit (second attempts)
    legs = [
        dict(name="plain", n=args.n, gen={}, labeled=False, L=args.search_list_size, S=args.rescore, nq=args.nq, env={}),
        dict(name="label_keys", n=args.n, gen={}, labeled=True, L=args.search_list_size, S=args.rescore, nq=args.nq, env={}),
        dict(name="ties", n=max(small, args.n // 8), gen=dict(n_clusters=4, intra_pct=3, noise_pct=2), labeled=False, L=40, S=60,
             nq=max(16, args.nq // 8), env={}),
        dict(name="exhausted", n=small, gen={}, labeled=False, L=1000, S=0, nq=max(8, min(64, args.nq // 8)), env={}),
        dict(name="tight_tables", n=max(small, args.n // 8), gen={}, labeled=False, L=args.search_list_size, S=args.rescore,
             nq=max(16, args.nq // 8), env={"VS_F_GCAP": "1024" if args.rescore < 100 else "4096"}),
    ]
    out, leg_report = {}, {}
    for leg in legs:
        tl = time.time()
        held = []
        ix = None
        saved = {k: os.environ.get(k) for k in leg["env"]}
        try:
            n, nq = leg["n"], leg["nq"]
            ix = P.DiskAnnIndex.alloc(ctx, n=n, dim_full=args.dim, num_neighbors=50, distance_type=P.VS_L2)
            gp = DatagenParams(seed=77, dim=args.dim, **leg["gen"])
            vecs_ptr, _ = ix.array(_lib.ARR_VECS)
            fill_device(ctx, gp, 0, n, vecs_ptr)
            ix.refresh_norms()
            ix.sbq_train()
            ix.sbq_quantize_corpus()
            d_val = d_off = None
            rng = np.random.default_rng(5)
            if leg["labeled"]:
                off, val, cnt = _labels(np, n, rng)
                ix.set_labels(off, val)
            ix.build_graph(search_list_size=args.build_l, max_alpha=1.2)
            if leg["labeled"]:
                owner = np.repeat(np.arange(n, dtype=np.uint32), cnt)
                labs, firsts = np.unique(val, return_index=True)
                ix.set_start_nodes(0, {int(l): int(owner[i]) for l, i in zip(labs, firsts)})
                d_val, d_off = ctx.alloc(nq * 2), ctx.alloc((nq + 1) * 4)
                held += [d_val, d_off]
                ctx.upload(d_val, rng.integers(1, 5, nq).astype(np.int16))  # keys of one label
                ctx.upload(d_off, np.arange(nq + 1, dtype=np.uint32))
            dq = ctx.alloc(nq * args.dim * 4)
            held.append(dq)
            fill_device(ctx, gp, 1 << 40, nq, dq)
            os.environ.update(leg["env"])
            rep = ix.autotune(dq, nq, leg["L"], leg["S"], 10, d_qlabels=d_val, d_qlabel_off=d_off, reps=1)
            for e in rep:
                o = out.setdefault(e["name"], {"applicable": False, "rows_identical": True, "error": 0, "legs": 0, "failed_legs": []})
                if e["applicable"] or e["error"]:
                    o["applicable"] = True
                    o["legs"] += 1
                    if e["error"] or not e["rows_identical"]:
                        o["rows_identical"] &= bool(e["rows_identical"]) and not e["error"]
                        o["error"] = o["error"] or e["error"]
                        o["failed_legs"].append(leg["name"])
            leg_report[leg["name"]] = {"ok": True, "n": n, "scans": nq, "seconds": round(time.time() - tl, 2)}
        except Exception as e:  # noqa: BLE001 — a leg that cannot be set up judges nobody (the first one has to run, though)
            leg_report[leg["name"]] = {"ok": False, "error": repr(e)[:300]}
            if leg["name"] == "plain":
                raise
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            if ix is not None:
                try:
                    ix.set_variant("default")
                    for p_ in held:
                        ctx.free(p_)
                    ix.close()
                except Exception:  # noqa: BLE001
                    pass
    ctx.close()
    print(json.dumps({"ok": True, "variants": out, "legs": leg_report, "seconds": round(time.time() - t0, 2),
                      "index": f"{args.n}x{args.dim} + corner legs, table-less regime forced"}), flush=True)


def run(dim=768, n=100_000, nq=8192, device=0, lib=None, timeout=240.0, search_list_size=3, rescore=196, build_l=64):
    """-> (names to skip, report dict); the first is None when the child did not finish cleanly (do not autotune at all then).
    Never raises: whatever goes wrong in the child only costs the variants."""
    all_names = None
    cmd = [sys.executable, "-m", "pgvectorscale_amd.tune_probe", "--dim", str(dim), "--n", str(n), "--nq", str(nq), "--device", str(device),
           "--search-list-size", str(search_list_size), "--rescore", str(rescore), "--build-l", str(build_l)]
    if lib:
        cmd += ["--lib", lib]
    env = dict(os.environ)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    env["VS_NO_TORCH"] = "1"  # (the child never touches torch: its import would be most of the probe's run time)
    t0 = time.time()
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=root, start_new_session=True)
        try:
            so, se = proc.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            proc.kill()  # (its own session: nothing else carries this pid)
            try:
                proc.communicate(timeout=30)
            except Exception:
                pass
            return all_names, {"ok": False, "error": f"probe child killed after {timeout:.0f} s", "seconds": round(time.time() - t0, 2)}
        if proc.returncode != 0:
            return all_names, {"ok": False, "error": f"probe child exited with {proc.returncode}: {se.decode(errors='replace')[-400:]}",
                               "seconds": round(time.time() - t0, 2)}
        rep = json.loads(so.decode().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001 — the probe is a safety net, never the reason a caller fails
        return all_names, {"ok": False, "error": repr(e), "seconds": round(time.time() - t0, 2)}
    skip = [name for name, v in rep["variants"].items()
            if name != "default" and v["applicable"] and (not v["rows_identical"] or v["error"])]
    if "bucket_bitmap" in skip:  # (the same kernel under the name it carries for indexes whose default is the LDS-table regime)
        skip.append("table_less_bitmap")
    rep["skip"] = skip
    return skip, rep


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--nq", type=int, default=8192)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--search-list-size", type=int, default=3)
    ap.add_argument("--rescore", type=int, default=196)
    ap.add_argument("--build-l", type=int, default=64)
    ap.add_argument("--lib", default=None, help="path of the shared library to load instead of libvsgpu.so")
    _child(ap.parse_args())
