"""vs_broker_config.cursor_lanes: the continuations of the scans' cursors (amgettuple past the rows of the shared first launch) run
on lanes — a thread, a HIP stream and a view of the index each — concurrently with each other and with the dispatcher's shared
launches.  Every backend still sees the oracle's rows and the oracle's GreedySearchStats; masks are replaced while scans stream;
a scan stays on its lane; everything the lanes allocated goes away with the broker."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from helpers import TestIndex

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
EMU = bool(os.environ.get("VS_EMU"))


def _live():
    if not EMU:
        return 0
    from conftest import EMU_LIB
    lib = C.CDLL(EMU_LIB)
    lib.vs_emu_live_allocations.restype = C.c_size_t
    return int(lib.vs_emu_live_allocations())


@pytest.mark.parametrize("lanes", [1, 3])
def test_deep_scans_stream_on_lanes(gpu_ctx, oracle, lanes):
    import pgvectorscale_amd as P
    O = oracle
    ti = TestIndex(n=2200, dim_full=48, bits=2, R=24, distance=O.L2, seed=41, kind="clustered", n_labels=4, deleted_frac=0.05, L_build=48)
    before = _live()
    ix = ti.upload(gpu_ctx)
    rng = np.random.default_rng(6)
    masks = {1: (rng.random(ti.n) > 0.3).astype(np.uint8), 2: (rng.random(ti.n) > 0.6).astype(np.uint8)}
    nthreads = 9
    q = ti.queries(nthreads, seed=12, kind="clustered")
    broker = P.Broker(ix, max_batch=16, max_wait_us=500, cursor_lanes=lanes)
    for sid, m in masks.items():
        broker.snapshot_put(sid, m)
    # what every backend must see: (snapshot, label key, GUCs, rows pulled)
    plans = [(t % 3, [1 + t % 4] if t % 2 else None, 8 + t, 4 + t % 5, 60 + 7 * t) for t in range(nthreads)]
    want = {}
    for t, (snap, key, L, S, pull) in enumerate(plans):
        ti.oracle.set_visibility(masks.get(snap))
        os_ = ti.oracle.scan(q[t], labels=key, L=L, rescore=S)
        rows = []
        for _ in range(pull):
            r = os_.gettuple()
            if r is None:
                break
            rows.append((r[0], r[1], np.float32(r[2]).view(np.uint32)))
        want[t] = (rows, os_.stats())
    ti.oracle.set_visibility(None)
    got, errors = {}, []
    start = threading.Barrier(nthreads + 1)

    def backend(t):
        try:
            snap, key, L, S, pull = plans[t]
            scan = broker.beginscan()
            scan.set_snapshot(snap)
            scan.rescan(q[t], labels=key, search_list_size=L, rescore=S)
            start.wait()
            rows = []
            for _ in range(pull):
                r = scan.gettuple()
                if r is None:
                    break
                rows.append((r[1], r[0], np.float32(r[2]).view(np.uint32)))
            got[t] = (rows, scan.stats(), scan.work())
            scan.endscan()
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=backend, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    start.wait()
    # masks nobody scans under are replaced while the scans stream: a put waits for the lanes to be between two tasks
    for i in range(6):
        broker.snapshot_put(5 + i % 2, (rng.random(ti.n) > 0.5).astype(np.uint8))
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(nthreads):
        rows, st, work = got[t]
        wrows, wst = want[t]
        assert [(a, b) for a, b, _ in rows] == [(a, b) for a, b, _ in wrows], t
        assert [c for _, _, c in rows] == [c for _, _, c in wrows], t
        for key in ("visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "full_distance_comparisons", "node_reads",
                    "node_heap_reads", "next_calls"):
            assert st[key] == wst[key], (t, key, st[key], wst[key])
        assert work["launches"] >= 1  # every plan pulls past the 16 rows of the shared launch: it was continued on a lane
    bst = broker.stats()
    assert bst["tasks"] >= nthreads
    broker.close()
    # a direct batch through the index afterwards: the lanes left its own mask and workspace alone
    gi, _, _, _ = ix.search_batch(q[:3], search_list_size=20, rescore=10, k=10)
    oi, _, _ = ti.oracle.search_batch(q[:3], L=20, rescore=10, k=10)
    assert (gi == oi).all()
    ix.close()
    assert _live() == before, "device / pinned memory of the lanes (contexts, views, cursors) outlived the broker and the index"


def test_backend_processes_stream_on_lanes(gpu_ctx, oracle, monkeypatch):
    """the same across processes (vs_shm_server with cursor lanes): the streaming and the badly-behaved-client tests of the shared-memory
    transport, with the serving process's cursors spread over two lanes"""
    monkeypatch.setenv("VS_BROKER_LANES", "2")  # servers created without a lane count take it from here
    import test_gpu_zu_shm as Z
    Z.test_backend_processes_stream_past_the_first_rows(gpu_ctx, oracle, 0)
    Z.test_fetch_protocol_random_walk(gpu_ctx, oracle)


def test_shm_server_with_lanes_gives_its_memory_back(gpu_ctx, oracle):
    import pgvectorscale_amd as P
    ti = TestIndex(n=600, dim_full=32, bits=2, R=16, distance=oracle.L2, seed=5, kind="gauss", L_build=30)
    before = _live()
    ix = ti.upload(gpu_ctx)
    name = f"/vs_shm_lanes_{os.getpid()}"
    srv = P.ShmServer(ix, name, nslots=4, kmax=8, cursor_lanes=3)
    cl = P.ShmClient(name)
    q = ti.queries(3, seed=2, kind="gauss")
    for i in range(3):  # three streamed scans (they hash to lanes by (pid, scan id)); two are left open for the server to drop
        os_ = ti.oracle.scan(q[i], L=10, rescore=4)
        scan = cl.beginscan(chunk=8)
        scan.rescan(q[i], search_list_size=10, rescore=4)
        for j in range(30):
            r, o = scan.gettuple(), os_.gettuple()
            assert r is not None and o is not None and r[1] == o[0] and r[0] == o[1], (i, j)
        if i == 0:
            scan.endscan()
    assert srv.stats()["cursors"] == 2
    cl.close()
    srv.close()
    ix.close()
    assert _live() == before
