"""vs_shm_*: the broker's request queue in POSIX shared memory.  Scans posted by client PROCESSES (PostgreSQL backends are
processes, AM/mod.rs:63) are coalesced by the one dispatcher that owns the device context, and every client gets exactly the rows
a scan of its own would have returned (= the oracle's rows): mixed GUCs, label keys (unsorted, with a duplicate), NULL queries,
more clients than slots, a client that dies holding a slot, and a dispatcher that goes away."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from helpers import TestIndex

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
KW = dict(n=2500, dim_full=64, bits=2, R=32, distance=1, seed=21, kind="gauss", n_labels=5, deleted_frac=0.1, L_build=64)


def _client(name, lib_path, jobs, out_q, die_after_claim=False):
    """runs in a child process: no device context, only the segment"""
    try:
        from pgvectorscale_amd import _lib
        if lib_path:
            _lib.LIB_PATH = lib_path
        import pgvectorscale_amd as P
        c = P.ShmClient(name)
        res = {}
        for (i, q, labels, L, S) in jobs:
            res[i] = c.search(q, labels, L, S, 10)
        c.close()
        out_q.put(("ok", res))
    except Exception as e:  # noqa: BLE001
        out_q.put(("err", repr(e)))


def test_client_processes_share_launches_and_get_the_oracles_rows(gpu_ctx, oracle):
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    ti = TestIndex(**KW)
    ix = ti.upload(gpu_ctx)
    nproc, per = 6, 8
    q = ti.queries(nproc * per, seed=77, kind="gauss")
    rng = np.random.default_rng(1)
    kinds = rng.integers(0, 4, len(q))
    keys = [sorted(set(int(v) for v in rng.integers(1, 6, int(rng.integers(1, 3))))) for _ in range(len(q))]
    want, jobs = {}, [[] for _ in range(nproc)]
    for i in range(len(q)):
        kind = int(kinds[i])
        if kind == 0:
            oi, od, _ = ti.oracle.search_batch(q[i:i + 1], L=40, rescore=20, k=10)
            job = (i, q[i], None, 40, 20)
        elif kind == 1:
            oi, od, _ = ti.oracle.search_batch(q[i:i + 1], L=40, rescore=20, k=10, qlabels=[keys[i]])
            job = (i, q[i], list(reversed(keys[i])) + keys[i][:1], 40, 20)  # unsorted, with a duplicate
        elif kind == 2:
            oi, od, _ = ti.oracle.search_batch(q[i:i + 1], L=25, rescore=5, k=10)
            job = (i, q[i], None, 25, 5)
        else:
            oi, od, _ = ti.oracle.search_batch(np.zeros((1, 64), np.float32), L=40, rescore=20, k=10)
            job = (i, None, [3], 40, 20)  # NULL query: its key is ignored
        want[i] = (oi[0], od[0])
        jobs[i % nproc].append(job)
    name = f"/vs_shm_test_{os.getpid()}"
    srv = P.ShmServer(ix, name, nslots=4, kmax=16, max_batch=64, max_wait_us=20000)  # fewer slots than clients: they queue
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    procs = [ctx.Process(target=_client, args=(name, _lib.LIB_PATH, jobs[p], out_q)) for p in range(nproc)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        status, payload = out_q.get(timeout=300)
        assert status == "ok", payload
        got.update(payload)
    for p in procs:
        p.join(60)
    assert len(got) == len(q)
    for i, (ids, tids, dist) in got.items():
        assert (ids == want[i][0]).all(), i
        live = ids != 0xFFFFFFFF
        assert (dist[live].view(np.uint32) == want[i][1][live].view(np.uint32)).all()
        assert (tids[live] == ti.tids[ids[live]]).all()
    st = srv.stats()
    assert st["scans"] == len(q) and st["batches"] < len(q)  # launches were shared
    # a client that asks for more rows than the segment holds per scan is told so; one that arrives after the dispatcher left too
    c = P.ShmClient(name)
    with pytest.raises(P.VsError):
        c.search(q[0], None, 40, 20, 17)
    ids, _, _ = c.search(q[0], None, 40, 20, 10)
    oi, _, _ = ti.oracle.search_batch(q[0:1], L=40, rescore=20, k=10)
    assert (ids == oi[0]).all()
    srv.close()
    with pytest.raises(P.VsError):
        c.search(q[0], None, 40, 20, 10)
    c.close()
    with pytest.raises(P.VsError):
        P.ShmClient(name)  # the segment is gone
    ix.close()
