"""Backend processes that stream scans through a vs_shm server at the same time — the measurement harness of the scan pools
(scripts/cursor_pool_concurrency.py, bench.py `cursor_pool` extra).  A backend maps the segment (no GPU, no device context), takes the
first chunk of its scan out of a shared OP_SEARCH launch and every later chunk as an OP_FETCH continuation, as a PostgreSQL backend's
amgettuple calls would (AM/scan.rs:369-436)."""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _backend(name, lib_path, t, query, L, rescore, rows_wanted, chunk, bar, outq):
    try:
        sys.path.insert(0, ROOT)
        os.environ["VS_NO_TORCH"] = "1"
        from pgvectorscale_amd import _lib
        _lib.LIB_PATH = lib_path
        import pgvectorscale_amd as P
        cl = P.ShmClient(name)
        rows = []
        for rep in range(3):  # (two untimed passes pay the serving process's allocations and lazy initialisations; the third is timed)
            sid = 1000 * (rep + 1) + t
            if rep == 2:
                bar.wait()  # ready (warm-up pass done)
                bar.wait()  # go
            ids, _, _ = cl.search(query, None, L, rescore, chunk)
            rows = ids.tolist()
            while len(rows) < rows_wanted:
                ids, _, _ = cl.fetch(sid, query, len(rows), chunk, None, L, rescore)
                rows.extend(ids.tolist())
                if len(ids) < chunk:
                    break
            cl.end_scan(sid)
        cl.close()
        outq.put(("ok", t, rows))
    except Exception as e:  # noqa: BLE001
        outq.put(("error", t, repr(e)))


def stream_many_c(name, queries, L, rescore, rows, chunk, timeout=600):
    """the backends as plain C processes (pgvectorscale_amd/vs_shm_lat, stream mode: what a PostgreSQL backend is to the server — the Python
    backends below spend about as long in the interpreter per fetch as the fetch takes) -> (wall ms of the timed pass, {backend: node ids})"""
    import json
    import subprocess
    import tempfile

    import numpy as np
    q = np.ascontiguousarray(np.stack([np.asarray(x, np.float32) for x in queries]))
    shm_dir = "/dev/shm" if os.path.isdir("/dev/shm") else None
    qf = tempfile.NamedTemporaryFile(prefix="vs_stream_q_", suffix=".f32", dir=shm_dir, delete=False)
    qf.write(q.tobytes())
    qf.close()
    out_path = qf.name + ".ids"
    try:
        cp = subprocess.run([os.path.join(ROOT, "pgvectorscale_amd", "vs_shm_lat"), name, qf.name, str(q.shape[1]), str(len(q)), str(len(q)), "1",
                             str(L), str(rescore), str(chunk), str(rows), out_path], capture_output=True, text=True, timeout=timeout)
        line = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
        res = json.loads(line[-1]) if line else {"error": (cp.stderr or "no output")[-300:]}
        if "error" in res:
            raise RuntimeError(f"vs_shm_lat (stream mode) failed: {res['error']}")
        ids = np.fromfile(out_path, np.uint32).reshape(len(q), rows)
        return float(res["wall_ms"]), {b: [int(v) for v in ids[b] if v != 0xFFFFFFFF] for b in range(len(q))}
    finally:
        for f in (qf.name, out_path):
            try:
                os.unlink(f)
            except OSError:
                pass


def stream_many(name, lib_path, queries, L, rescore, rows, chunk, timeout=600):
    """len(queries) backend processes, one scan each, `rows` rows in chunks of `chunk` -> (wall ms of the timed pass, {backend: node ids}).
    C backends where the harness binary is built (VS_PY_BACKENDS=1 keeps the Python ones)"""
    if os.path.exists(os.path.join(ROOT, "pgvectorscale_amd", "vs_shm_lat")) and not os.environ.get("VS_PY_BACKENDS"):
        return stream_many_c(name, queries, L, rescore, rows, chunk, timeout)
    nt = len(queries)
    mpc = mp.get_context("spawn")
    bar = mpc.Barrier(nt + 1)
    outq = mpc.Queue()
    procs = [mpc.Process(target=_backend, args=(name, lib_path, t, queries[t], L, rescore, rows, chunk, bar, outq)) for t in range(nt)]
    for pr in procs:
        pr.start()
    res = []
    try:
        bar.wait(timeout)  # every backend has mapped the segment and run its warm-up passes
        bar.wait(timeout)  # go
        t0 = time.perf_counter()
        res = [outq.get(timeout=timeout) for _ in procs]
        wall = (time.perf_counter() - t0) * 1e3
    finally:
        try:
            bar.abort()  # (whoever still waits at the barrier leaves it with an error instead of hanging)
        except Exception:  # noqa: BLE001
            pass
        deadline = time.time() + 20
        for pr in procs:
            pr.join(max(0.1, deadline - time.time()))
        for pr in procs:
            if pr.is_alive():
                pr.terminate()
    errors = [r for r in res if r[0] != "ok"]
    if errors:
        raise RuntimeError(f"backend processes failed: {errors[:3]}")
    return wall, {r[1]: r[2] for r in res}
