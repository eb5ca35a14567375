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


def _doomed_client(name, lib_path, q):
    from pgvectorscale_amd import _lib
    if lib_path:
        _lib.LIB_PATH = lib_path
    import pgvectorscale_amd as P
    c = P.ShmClient(name)
    c.search(q, None, 40, 20, 10)  # the parent kills this process while the request waits for its batch


def test_slot_of_a_dead_client_is_reclaimed(gpu_ctx, oracle):
    """a backend that dies while its scan is queued must not leak its slot: with ONE slot in the segment the next client can
    only be served if the dispatcher took the slot back"""
    import signal
    import time
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    ti = TestIndex(**KW)
    ix = ti.upload(gpu_ctx)
    q = ti.queries(2, seed=5, kind="gauss")
    name = f"/vs_shm_dead_{os.getpid()}"
    srv = P.ShmServer(ix, name, nslots=1, kmax=16, max_batch=64, max_wait_us=1_500_000)  # requests wait 1.5 s for company
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_doomed_client, args=(name, _lib.LIB_PATH, q[0]))
    p.start()
    deadline = time.time() + 120
    while srv.stats()["scans"] == 0 and p.is_alive() and time.time() < deadline:  # wait until the request is in the segment ...
        time.sleep(0.05)
        if p.pid and os.path.exists(f"/proc/{p.pid}"):
            try:  # ... (posted = the child sleeps in its futex) and kill it before the batch window closes
                if "futex" in open(f"/proc/{p.pid}/wchan").read():
                    break
            except OSError:
                pass
    time.sleep(0.2)
    os.kill(p.pid, signal.SIGKILL)
    p.join(30)
    c = P.ShmClient(name)
    ids, _, _ = c.search(q[1], None, 40, 20, 10)  # blocks in the slot claim until the dead client's slot is free again
    oi, _, _ = ti.oracle.search_batch(q[1:2], L=40, rescore=20, k=10)
    assert (ids == oi[0]).all()
    c.close()
    srv.close()
    ix.close()


def test_snapshot_masks_across_processes(gpu_ctx, oracle):
    """a client names the snapshot of the serving process its scan runs under; scans of different snapshots are not grouped"""
    import pgvectorscale_amd as P
    ti = TestIndex(**KW)
    ix = ti.upload(gpu_ctx)
    q = ti.queries(3, seed=15, kind="gauss")
    mask = (np.random.default_rng(8).random(ti.n) > 0.4).astype(np.uint8)
    name = f"/vs_shm_snap_{os.getpid()}"
    srv = P.ShmServer(ix, name, nslots=4, kmax=16, max_batch=64, max_wait_us=1000)
    c = P.ShmClient(name)
    with pytest.raises(P.VsError, match="no visibility mask"):
        c.search(q[0], None, 30, 12, 10, snapshot=3)
    srv.snapshot_put(3, mask)
    for snap, m in ((3, mask), (0, None)):
        ti.oracle.set_visibility(m)
        oi, od, _ = ti.oracle.search_batch(q, L=30, rescore=12, k=10)
        for i in range(len(q)):
            ids, _, dist = c.search(q[i], None, 30, 12, 10, snapshot=snap)
            assert (ids == oi[i]).all() and (dist.view(np.uint32) == od[i].view(np.uint32)).all()
    ti.oracle.set_visibility(None)
    with pytest.raises(P.VsError):
        c.search(q[0], None, 30, 12, 10, snapshot=99)
    c.close()
    srv.close()
    ix.close()


def _doomed_server(name, lib_path, emu, ready, kw):
    """a serving process (device context + index + dispatcher) that the parent kills without ceremony"""
    import time
    if emu:
        os.environ["VS_EMU"] = emu
    from pgvectorscale_amd import _lib
    if lib_path:
        _lib.LIB_PATH = lib_path
    import pgvectorscale_amd as P
    ti = TestIndex(**kw)
    ctx = P.Context(0)
    ix = ti.upload(ctx)
    srv = P.ShmServer(ix, name, nslots=2, kmax=16, max_batch=64, max_wait_us=3_000_000)  # requests wait 3 s for company
    ready.set()
    time.sleep(600)
    srv.close()


def test_client_notices_a_dispatcher_that_died(oracle):
    """SIGKILL of the serving process leaves `serving` set in the segment: a client that is already waiting for its rows, and one
    that arrives afterwards, must both come back with an error instead of sleeping forever (the pid in the header is checked)"""
    import signal
    import threading
    import time
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    kw = dict(n=400, dim_full=32, bits=2, R=16, distance=1, seed=3, kind="gauss", L_build=32)
    name = f"/vs_shm_crash_{os.getpid()}"
    ctx = mp.get_context("spawn")
    ready = ctx.Event()
    p = ctx.Process(target=_doomed_server, args=(name, _lib.LIB_PATH, os.environ.get("VS_EMU", ""), ready, kw))
    p.start()
    assert ready.wait(300), "serving process did not come up"
    q = TestIndex(**kw).queries(2, seed=1, kind="gauss")
    c = P.ShmClient(name)
    result = {}

    def waiting_client():
        try:
            c.search(q[0], None, 20, 8, 10)  # posted, then held back by the 3 s gather window
            result["first"] = "returned rows"
        except P.VsError as e:
            result["first"] = f"error: {e}"

    th = threading.Thread(target=waiting_client)
    th.start()
    time.sleep(0.5)
    os.kill(p.pid, signal.SIGKILL)
    p.join(30)
    th.join(20)
    assert not th.is_alive(), "the waiting client never came back"
    assert result["first"].startswith("error"), result
    c2 = P.ShmClient(name)  # the segment is still there (nobody unlinked it), its dispatcher is not
    t0 = time.time()
    with pytest.raises(P.VsError):
        c2.search(q[1], None, 20, 8, 10)
    assert time.time() - t0 < 10
    c2.close()
    c.close()
    try:  # what vs_shm_server_destroy would have done
        os.unlink("/dev/shm" + name)
    except OSError:
        pass


def _streaming_client(name, lib_path, jobs, out_q):
    """a backend process pulling rows one at a time (amgettuple): chunks of 8 through ShmScan"""
    try:
        from pgvectorscale_amd import _lib
        if lib_path:
            _lib.LIB_PATH = lib_path
        import pgvectorscale_amd as P
        c = P.ShmClient(name)
        res = {}
        scan = c.beginscan(chunk=8)
        for (i, q, labels, L, S, nrows) in jobs:
            scan.rescan(q, labels, L, S)
            rows = []
            for _ in range(nrows):
                r = scan.gettuple()
                if r is None:
                    break
                rows.append(r)
            res[i] = rows
        scan.endscan()
        # a backend that lost track: rows 20..27 of its last scan under a new name — the server opens a cursor and fast-forwards
        (i, q, labels, L, S, nrows) = jobs[-1]
        ids, tids, dist = c.fetch(10_000 + i, q, 20, 8, labels, L, S)
        res[("again", i)] = list(zip(tids.tolist(), ids.tolist(), dist.tolist()))
        # ... and asks for the same rows once more (the cursor is past them: it starts over)
        ids, tids, dist = c.fetch(10_000 + i, q, 20, 8, labels, L, S)
        res[("again2", i)] = list(zip(tids.tolist(), ids.tolist(), dist.tolist()))
        c.end_scan(10_000 + i)
        c.end_scan(424242)  # (a name the server has never seen: nothing to drop, no error)
        c.close()
        out_q.put(("ok", res))
    except Exception as e:  # noqa: BLE001
        out_q.put(("err", repr(e)))


@pytest.mark.parametrize("cursor_pool", [0, 6], ids=["cursor_per_scan", "scan_pools"])
def test_backend_processes_stream_past_the_first_rows(gpu_ctx, oracle, cursor_pool):
    """(scan_pools: the streamed scans live in scan pools of six slots — more scans than slots, three different GUC pairs — and the
    continuations of one dispatcher round share their launches, vs_scanpool.cpp.)  amgettuple across processes: the first chunk of a scan comes out of a shared launch, every later chunk continues the cursor
    the serving process keeps for that scan (AM/scan.rs:162-174,370-405) — rows equal the oracle's streaming scan, one row at a
    time, for plain, label-keyed, NULL and exhausted scans, from four processes at once"""
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    ti = TestIndex(**KW)
    ix = ti.upload(gpu_ctx)
    nproc = 4
    q = ti.queries(nproc * 3, seed=81, kind="gauss")
    jobs = [[] for _ in range(nproc)]
    want = {}
    for i in range(len(q)):
        kind = i % 3
        if kind == 0:
            job = (i, q[i], None, 30, 12, 70)
            os_ = ti.oracle.scan(q[i], L=30, rescore=12)
        elif kind == 1:
            job = (i, q[i], [4, 2, 4], 20, 6, 50)
            os_ = ti.oracle.scan(q[i], labels=[2, 4], L=20, rescore=6)
        else:
            job = (i, None, [3], 3, 0, 10_000)  # NULL query, pulled to the end (the scan visits every live node)
            os_ = ti.oracle.scan(None, L=3, rescore=0)
        rows = []
        for _ in range(job[5]):
            o = os_.gettuple()
            if o is None:
                break
            rows.append(o)
        want[i] = rows
        jobs[i % nproc].append(job)
    name = f"/vs_shm_stream_{os.getpid()}"
    srv = P.ShmServer(ix, name, nslots=3, kmax=8, max_batch=64, max_wait_us=5000, cursor_pool=cursor_pool)
    ctx = mp.get_context("spawn")
    out_q = ctx.Queue()
    procs = [ctx.Process(target=_streaming_client, args=(name, _lib.LIB_PATH, jobs[p], out_q)) for p in range(nproc)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        status, payload = out_q.get(timeout=600)
        assert status == "ok", payload
        got.update(payload)
    for p in procs:
        p.join(60)

    def same(rows, ref, what):
        assert len(rows) == len(ref), (what, len(rows), len(ref))
        for j, ((tid, node, d), o) in enumerate(zip(rows, ref)):
            assert node == o[0] and tid == o[1], (what, j)
            if not (np.isnan(d) and np.isnan(o[2])):
                assert np.float32(d).view(np.uint32) == np.float32(o[2]).view(np.uint32), (what, j)

    for i in range(len(q)):
        same(got[i], want[i], i)
    assert any(len(want[i]) > 2000 for i in range(len(q)))  # the NULL scans really ran to the end
    for key, rows in got.items():
        if isinstance(key, tuple):
            same(rows, want[key[1]][20:28], key)
    st = srv.stats()
    assert st["tasks"] > len(q)  # cursor requests were served ...
    assert st["scans"] == len(q)  # ... and only the first chunk of every scan went through a shared launch
    srv.close()
    ix.close()


def test_fetch_protocol_random_walk(gpu_ctx, oracle):
    """the cursor protocol under a client that behaves badly: positions out of order, rewinds, scans dropped and resumed, an id
    reused for another scan — every answer is rows [skip, skip + k) of the oracle's scan, fewer only at its end"""
    import pgvectorscale_amd as P
    ti = TestIndex(**KW)
    ix = ti.upload(gpu_ctx)
    name = f"/vs_shm_walk_{os.getpid()}"
    srv = P.ShmServer(ix, name, nslots=2, kmax=8, max_batch=8, max_wait_us=0)
    c = P.ShmClient(name)  # (a client may live in the serving process too: the dispatcher is a thread of its own)
    q = ti.queries(4, seed=91, kind="gauss")
    scans = [dict(query=q[0], labels=None, L=25, S=9), dict(query=q[1], labels=[5, 1], L=12, S=4), dict(query=None, labels=None, L=2, S=0),
             dict(query=q[2], labels=None, L=40, S=0)]
    want = []
    for s in scans:
        os_ = ti.oracle.scan(s["query"], labels=None if s["query"] is None else s["labels"], L=s["L"], rescore=s["S"])
        rows = []
        while len(rows) < 400:
            o = os_.gettuple()
            if o is None:
                break
            rows.append(o)
        want.append(rows)
    rng = np.random.default_rng(17)
    pos = [0] * len(scans)
    ids_of = list(range(100, 100 + len(scans)))
    for step in range(150):
        j = int(rng.integers(0, len(scans)))
        u = rng.random()
        if u < 0.08:
            c.end_scan(ids_of[j])
            continue
        if u < 0.12:  # two scans swap their names: the server must notice that the scan behind an id changed
            a, b = int(rng.integers(0, len(scans))), int(rng.integers(0, len(scans)))
            ids_of[a], ids_of[b] = ids_of[b], ids_of[a]
            continue
        skip = pos[j] if u < 0.75 else (0 if u < 0.82 else int(rng.integers(0, min(len(want[j]), 390) + 1)))
        k = int(rng.integers(1, 9))
        s = scans[j]
        ids, tids, dist = c.fetch(ids_of[j], s["query"], skip, k, s["labels"], s["L"], s["S"])
        ref = want[j][skip:skip + k]
        if skip + k <= 400 or len(want[j]) < 400:
            assert len(ids) == len(ref), (step, j, skip, k, len(ids), len(ref))
        for (node, tid, d), gi, gt, gd in zip(ref, ids.tolist(), tids.tolist(), dist.tolist()):
            assert gi == node and gt == tid, (step, j, skip)
            if not (np.isnan(gd) and np.isnan(d)):
                assert np.float32(gd).view(np.uint32) == np.float32(d).view(np.uint32), (step, j, skip)
        pos[j] = skip + len(ids)
    assert srv.stats()["tasks"] >= 100
    c.close()
    srv.close()
    ix.close()


def _cursor_holder(name, lib_path, q, ready):
    from pgvectorscale_amd import _lib
    if lib_path:
        _lib.LIB_PATH = lib_path
    import time
    import pgvectorscale_amd as P
    c = P.ShmClient(name)
    c.fetch(7, q, 0, 8, None, 30, 10)  # opens a cursor in the serving process ...
    ready.put("open")
    time.sleep(600)  # ... and never ends the scan: the parent kills this process


def test_cursors_of_a_dead_client_are_dropped(gpu_ctx, oracle):
    """a backend that dies in the middle of a streamed scan must not leave its cursor (device memory) behind in the serving process"""
    import signal
    import time
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    ti = TestIndex(**KW)
    ix = ti.upload(gpu_ctx)
    q = ti.queries(2, seed=6, kind="gauss")
    name = f"/vs_shm_cur_{os.getpid()}"
    srv = P.ShmServer(ix, name, nslots=2, kmax=8, max_batch=8, max_wait_us=0)
    ctx = mp.get_context("spawn")
    ready = ctx.Queue()
    p = ctx.Process(target=_cursor_holder, args=(name, _lib.LIB_PATH, q[0], ready))
    p.start()
    assert ready.get(timeout=300) == "open"
    assert srv.stats()["cursors"] == 1
    c = P.ShmClient(name)
    ids, _, _ = c.fetch(1, q[1], 0, 8, None, 30, 10)  # a live client's cursor next to it
    assert len(ids) == 8 and srv.stats()["cursors"] == 2
    os.kill(p.pid, signal.SIGKILL)
    p.join(30)
    deadline = time.time() + 30
    while srv.stats()["cursors"] != 1 and time.time() < deadline:
        time.sleep(0.05)
    assert srv.stats()["cursors"] == 1  # the dead client's is gone, the live one's stays ...
    os_ = ti.oracle.scan(q[1], L=30, rescore=10)
    want = [os_.gettuple() for _ in range(16)]
    ids, _, _ = c.fetch(1, q[1], 8, 8, None, 30, 10)  # ... and continues
    assert ids.tolist() == [o[0] for o in want[8:]]
    c.end_scan(1)
    assert srv.stats()["cursors"] == 0
    c.close()
    srv.close()
    ix.close()


def test_a_client_that_rewrites_its_request_after_posting_cannot_move_the_dispatcher(gpu_ctx, oracle):
    """The segment is writable by every client, so the dispatcher copies a posted request into its own memory and validates the
    copy (vs_shm.cpp, header comment).  A hostile client — this test, through a raw mapping of the segment — posts a valid request
    and rewrites k, the label count, the GUCs and the header's geometry the moment the slot is taken; and posts requests that are
    out of range to begin with.  Whatever happens, the dispatcher writes at most the rows the request it TOOK asked for, inside that
    slot: the neighbouring slot's bytes do not change, the server stays alive and keeps answering honest clients with the oracle's
    rows."""
    import mmap
    import struct
    import time
    import pgvectorscale_amd as P
    ti = TestIndex(**KW)
    ix = ti.upload(gpu_ctx)
    name = f"/vs_shm_hostile_{os.getpid()}"
    nslots, kmax, dim = 4, 16, 64
    srv = P.ShmServer(ix, name, nslots=nslots, kmax=kmax, max_batch=8, max_wait_us=100)
    S_FREE, S_CLAIMED, S_READY, S_RUNNING, S_DONE = 0, 1, 2, 3, 4
    HDR = 64                     # ShmHeader: 9 x u32 + pad, 16-byte aligned
    OFF = dict(state=0, pid=4, L=8, rescore=12, k=16, n_labels=20, has_key=24, null_q=28, rc=32, snapshot=36, op=40, skip=44, n_rows=48,
               scan_id=56, err=64, labels=232, query=360)
    fd = os.open("/dev/shm" + name, os.O_RDWR)
    try:
        size = os.fstat(fd).st_size
        m = mmap.mmap(fd, size)
        magic, version, h_nslots, h_dim, h_kmax, slot_bytes = struct.unpack_from("<6I", m, 0)
        assert (h_nslots, h_dim, h_kmax) == (nslots, dim, kmax)
        u32 = lambda off: struct.unpack_from("<I", m, off)[0]  # noqa: E731

        def put32(off, v):
            struct.pack_into("<I", m, off, v & 0xFFFFFFFF)

        def post(slot, k, L=20, rescore=8, n_labels=0, op=0, rewrite=None):
            base = HDR + slot * slot_bytes
            assert u32(base + OFF["state"]) == S_FREE
            put32(base + OFF["state"], S_CLAIMED)
            put32(base + OFF["pid"], os.getpid())
            for key, v in (("L", L), ("rescore", rescore), ("k", k), ("n_labels", n_labels), ("has_key", 1 if n_labels else 0), ("null_q", 0),
                           ("snapshot", 0), ("op", op), ("skip", 0)):
                put32(base + OFF[key], v)
            q = ti.queries(1, seed=5, kind="gauss")[0].astype(np.float32)
            m[base + OFF["query"]: base + OFF["query"] + dim * 4] = q.tobytes()
            put32(base + OFF["state"], S_READY)
            put32(28, u32(28) + 1)  # work_seq: the dispatcher also polls every 50 ms, so no futex wake is needed
            t0 = time.time()
            while u32(base + OFF["state"]) not in (S_RUNNING, S_DONE) and time.time() - t0 < 20:
                pass
            if rewrite:  # the slot has been taken: now lie about what was asked
                for key, v in rewrite.items():
                    put32(base + OFF[key], v)
            while u32(base + OFF["state"]) != S_DONE and time.time() - t0 < 30:
                time.sleep(0.001)
            assert u32(base + OFF["state"]) == S_DONE, "the dispatcher never finished the slot"
            rc = struct.unpack_from("<i", m, base + OFF["rc"])[0]
            ids = np.frombuffer(m, np.uint32, kmax, base + OFF["query"] + dim * 4 + kmax * 8).copy()
            put32(base + OFF["pid"], 0)
            put32(base + OFF["state"], S_FREE)
            return rc, ids, q

        oracle_rows = lambda q, k: ti.oracle.search_batch(q[None, :], L=20, rescore=8, k=k)[0][0]  # noqa: E731
        victim = HDR + 1 * slot_bytes
        for trial in range(6):
            before = bytes(m[victim: victim + slot_bytes])
            rc, ids, q = post(0, k=5, rewrite={"k": 1 << 20, "n_labels": 60000, "L": 0, "rescore": 1 << 30, "op": 77, "skip": 1 << 31})
            assert rc in (0, -1)  # served as posted, or (the lie landed before the copy) rejected as out of range
            if rc == 0:
                assert (ids[:5] == oracle_rows(q, 5)).all()
            assert bytes(m[victim: victim + slot_bytes]) == before, "the dispatcher wrote outside the slot it was serving"
        # requests that are out of range from the start are failed, not run
        for bad in (dict(k=kmax + 1), dict(k=0), dict(k=4, L=0), dict(k=4, rescore=5000), dict(k=4, n_labels=65), dict(k=4, op=9)):
            rc, _, _ = post(2, **bad)
            assert rc == -1, bad
        # the header's geometry is not read again either: a client that rewrites it changes nothing for the dispatcher
        struct.pack_into("<4I", m, 8, 1, 7, 1 << 20, 64)
        rc, ids, q = post(3, k=6)
        assert rc == 0 and (ids[:6] == oracle_rows(q, 6)).all()
        struct.pack_into("<4I", m, 8, nslots, dim, kmax, slot_bytes)
        m.close()
    finally:
        os.close(fd)
        srv.close()
    # ... and with cursor lanes: a streamed request (OP_FETCH) is read by a LANE thread while the dispatcher keeps polling.  A client
    # that writes READY over the state word of its RUNNING slot with k = 0xFFFFFFFF must neither get that request copied over the one
    # the lane is serving nor get the slot queued twice (round-4 advisor finding: take() published before it validated)
    name2 = f"/vs_shm_hostile_lanes_{os.getpid()}"
    srv2 = P.ShmServer(ix, name2, nslots=nslots, kmax=kmax, max_batch=8, max_wait_us=100, cursor_lanes=2)
    fd = os.open("/dev/shm" + name2, os.O_RDWR)
    try:
        size = os.fstat(fd).st_size
        m = mmap.mmap(fd, size)
        slot_bytes = struct.unpack_from("<6I", m, 0)[5]
        victim = HDR + 1 * slot_bytes
        q = ti.queries(1, seed=5, kind="gauss")[0].astype(np.float32)
        want = ti.oracle.search_batch(q[None, :], L=20, rescore=8, k=5)[0][0]
        for trial in range(8):
            before = bytes(m[victim: victim + slot_bytes])
            base = HDR
            assert u32(base + OFF["state"]) == S_FREE
            put32(base + OFF["state"], S_CLAIMED)
            put32(base + OFF["pid"], os.getpid())
            for key, v in (("L", 20), ("rescore", 8), ("k", 5), ("n_labels", 0), ("has_key", 0), ("null_q", 0), ("snapshot", 0), ("op", 1),
                           ("skip", 0)):
                put32(base + OFF[key], v)
            struct.pack_into("<Q", m, base + OFF["scan_id"], 1000 + trial)
            m[base + OFF["query"]: base + OFF["query"] + dim * 4] = q.tobytes()
            put32(base + OFF["state"], S_READY)
            put32(28, u32(28) + 1)
            t0 = time.time()
            while u32(base + OFF["state"]) not in (S_RUNNING, S_DONE) and time.time() - t0 < 20:
                pass
            # the slot is with a lane: post again over it, out of range
            put32(base + OFF["k"], 0xFFFFFFFF)
            put32(base + OFF["skip"], 0)
            put32(base + OFF["state"], S_READY)
            put32(28, u32(28) + 1)
            deadline = time.time() + 30
            seen_done = 0
            while time.time() < deadline:  # the honest request's DONE, then (if the re-post survived it) the rejection's
                if u32(base + OFF["state"]) == S_DONE:
                    seen_done += 1
                    time.sleep(0.05)
                    if u32(base + OFF["state"]) == S_DONE:
                        break
                time.sleep(0.001)
            assert u32(base + OFF["state"]) == S_DONE, "the dispatcher never finished the slot"
            assert bytes(m[victim: victim + slot_bytes]) == before, "a lane wrote outside the slot it was serving"
            put32(base + OFF["pid"], 0)
            put32(base + OFF["state"], S_FREE)
        m.close()
        # the server is alive and exact for an honest client
        cl = P.ShmClient(name2)
        try:
            gi = cl.search(q, search_list_size=20, rescore=8, k=5)[0]
            assert (np.asarray(gi)[:5] == want).all()
        finally:
            cl.close()
    finally:
        os.close(fd)
        srv2.close()
        ix.close()


def _pool_pressure_client(name, lib_path, t, q, chunk, nrows, bar, out_q):
    """a backend that streams ONE scan in lockstep with the others (a barrier before every fetch), so that every dispatcher round
    sees more streamed scans of one (L, rescore, snapshot) than the pool has slots"""
    try:
        from pgvectorscale_amd import _lib
        if lib_path:
            _lib.LIB_PATH = lib_path
        import pgvectorscale_amd as P
        c = P.ShmClient(name)
        sid = 7000 + t
        ids, tids, dist = c.search(q, None, 30, 12, chunk)
        rows = list(zip(tids.tolist(), ids.tolist(), dist.tolist()))
        while len(rows) < nrows:
            try:
                bar.wait(30)  # (a backend that is done leaves the others waiting: they go on alone after the timeout)
            except Exception:  # noqa: BLE001
                pass
            ids, tids, dist = c.fetch(sid, q, len(rows), chunk, None, 30, 12)
            rows.extend(zip(tids.tolist(), ids.tolist(), dist.tolist()))
            if len(ids) < chunk:
                break
        c.end_scan(sid)
        c.close()
        out_q.put(("ok", t, rows))
    except Exception as e:  # noqa: BLE001
        out_q.put(("err", t, repr(e)))


@pytest.mark.parametrize("chunks", [(8, 8, 8, 8, 8), (8, 5, 8, 3, 5)], ids=["equal_k", "mixed_k"])
def test_more_streamed_scans_than_pool_slots_never_share_a_slot(gpu_ctx, oracle, chunks):
    """Advisor finding of round 5 (vs_shm.cpp run_fetch_pooled): with a pool SMALLER than the streamed scans of one dispatcher round,
    the LRU eviction could hand the slot of a request already placed in this round to the next one — equal k: `slot listed twice`
    for the whole group; mixed k: the earlier client silently got the later client's rows.  Five backends in lockstep over a pool of
    two slots: every backend gets exactly the oracle's rows, whatever is evicted in between."""
    import pgvectorscale_amd as P
    from pgvectorscale_amd import _lib
    ti = TestIndex(**KW)
    ix = ti.upload(gpu_ctx)
    nproc = len(chunks)
    q = ti.queries(nproc, seed=83, kind="gauss")
    name = f"/vs_shm_pp_{os.getpid()}"
    srv = P.ShmServer(ix, name, nslots=8, kmax=8, max_batch=64, max_wait_us=20000, cursor_pool=2)
    ctx = mp.get_context("spawn")
    bar = ctx.Barrier(nproc)
    out_q = ctx.Queue()
    nrows = 40
    procs = [ctx.Process(target=_pool_pressure_client, args=(name, _lib.LIB_PATH, t, q[t], chunks[t], nrows, bar, out_q)) for t in range(nproc)]
    for p in procs:
        p.start()
    got = {}
    for _ in procs:
        status, t, payload = out_q.get(timeout=600)
        assert status == "ok", payload
        got[t] = payload
    for p in procs:
        p.join(60)
    for t in range(nproc):
        os_ = ti.oracle.scan(q[t], L=30, rescore=12)
        ref = []
        while len(ref) < len(got[t]):
            o = os_.gettuple()
            if o is None:
                break
            ref.append(o)
        assert len(got[t]) >= nrows and len(ref) == len(got[t]), (t, len(got[t]), len(ref))
        for j, ((tid, node, d), o) in enumerate(zip(got[t], ref)):
            assert node == o[0] and tid == o[1], (t, j, chunks)
            assert np.float32(d).view(np.uint32) == np.float32(o[2]).view(np.uint32), (t, j)
    ps = srv.pool_stats()
    assert ps["pools"] == 1 and ps["rounds"] > 0, ps  # the pool really served rounds
    srv.close()
    ix.close()


def test_scan_pools_are_retired_when_the_combinations_move_on(gpu_ctx, oracle):
    """Advisor finding of round 5: pools are keyed by (search_list_size, rescore, snapshot) and capped at four; they were never
    retired, so the fifth combination fell back to single cursors for good.  Seven combinations one after another: every scan
    streams out of a pool (the pool of a finished combination is re-keyed at the cap) and returns the oracle's rows."""
    import pgvectorscale_amd as P
    ti = TestIndex(**KW)
    ix = ti.upload(gpu_ctx)
    q = ti.queries(7, seed=85, kind="gauss")
    name = f"/vs_shm_ret_{os.getpid()}"
    srv = P.ShmServer(ix, name, nslots=2, kmax=8, max_batch=8, max_wait_us=0, cursor_pool=3)
    c = P.ShmClient(name)
    combos = [(20, 5), (24, 6), (28, 7), (32, 8), (36, 9), (40, 10), (20, 5)]
    rounds_before = 0
    for i, (L, S) in enumerate(combos):
        os_ = ti.oracle.scan(q[i], L=L, rescore=S)
        want = [os_.gettuple() for _ in range(24)]
        ids, _, _ = c.search(q[i], None, L, S, 8)
        rows = ids.tolist()
        while len(rows) < 24:
            ids, _, _ = c.fetch(500 + i, q[i], len(rows), 8, None, L, S)
            rows.extend(ids.tolist())
        assert rows == [o[0] for o in want], (i, L, S)
        c.end_scan(500 + i)
        ps = srv.pool_stats()
        assert ps["rounds"] > rounds_before, (i, ps)  # this combination was served out of a pool, not by a cursor of its own
        rounds_before = ps["rounds"]
        assert ps["pools"] <= 4 and ps["scans"] == 0, ps
    ps = srv.pool_stats()
    assert ps["retired"] >= 2, ps
    assert srv.stats()["cursors"] == 0
    c.close()
    srv.close()
    ix.close()
