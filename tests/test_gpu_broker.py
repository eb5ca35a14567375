"""vs_broker_*: scans arriving from many client threads are coalesced into batched launches and every client gets exactly
the rows a scan of its own would have returned (= the oracle's rows)."""
import threading

import numpy as np
import pytest

from helpers import TestIndex

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def test_concurrent_scans_are_batched_and_exact(gpu_ctx, oracle):
    import pgvectorscale_amd as P
    O = oracle
    ti = TestIndex(n=2500, dim_full=64, bits=2, R=32, distance=O.L2, seed=21, kind="gauss", n_labels=5, deleted_frac=0.1, L_build=64)
    ix = ti.upload(gpu_ctx)
    nthreads, per_thread = 24, 6
    q = ti.queries(nthreads * per_thread, seed=77, kind="gauss")
    rng = np.random.default_rng(1)
    # a mix the dispatcher has to keep apart: plain scans, scans with a label key, another (L, rescore), NULL queries
    kinds = rng.integers(0, 4, len(q))
    keys = [sorted(set(int(v) for v in rng.integers(1, 7, int(rng.integers(1, 3))))) for _ in range(len(q))]
    want = {}
    for i in range(len(q)):
        kind = int(kinds[i])
        if kind == 0:
            oi, od, _ = ti.oracle.search_batch(q[i:i + 1], L=40, rescore=20, k=10)
        elif kind == 1:
            oi, od, _ = ti.oracle.search_batch(q[i:i + 1], L=40, rescore=20, k=10, qlabels=[keys[i]])
        elif kind == 2:
            oi, od, _ = ti.oracle.search_batch(q[i:i + 1], L=25, rescore=5, k=10)
        else:
            oi, od, _ = ti.oracle.search_batch(np.zeros((1, 64), np.float32), L=40, rescore=20, k=10)
        want[i] = (oi[0], od[0])
    broker = P.Broker(ix, max_batch=64, max_wait_us=20000)
    got, errors = {}, []
    start = threading.Barrier(nthreads)

    def client(t):
        try:
            start.wait()
            for j in range(per_thread):
                i = t * per_thread + j
                kind = int(kinds[i])
                if kind == 0:
                    got[i] = broker.search(q[i], None, 40, 20, 10)
                elif kind == 1:
                    got[i] = broker.search(q[i], list(reversed(keys[i])) + keys[i][:1], 40, 20, 10)  # unsorted, with a duplicate
                elif kind == 2:
                    got[i] = broker.search(q[i], None, 25, 5, 10)
                else:
                    got[i] = broker.search(None, [3], 40, 20, 10)  # NULL query: its key is ignored
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=client, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    for i in range(len(q)):
        ids, tids, dist = got[i]
        oi, od = want[i]
        assert (ids == oi).all(), i
        both_nan = np.isnan(dist) & np.isnan(od)
        assert (both_nan | (dist.view(np.uint32) == od.view(np.uint32))).all(), i
        live = ids != 0xFFFFFFFF
        assert (tids[live] == ti.tids[ids[live]]).all()
    st = broker.stats()
    assert st["scans"] == len(q)
    assert st["batches"] < len(q) / 2 and st["max_batch"] >= 4, st  # the scans really shared launches
    # an invalid GUC is reported to the caller that sent it, and the broker keeps serving
    with pytest.raises(P.VsError, match="query_rescore"):
        broker.search(q[0], None, 40, 5000, 10)
    ids, _, _ = broker.search(q[0], None, 40, 20, 10) if int(kinds[0]) != 0 else got[0]
    broker.close()
    ix.close()


def test_amgettuple_mirror_on_a_broker(gpu_ctx, oracle):
    """many "backends" (threads), each running amrescan + amgettuple loops on its own scan descriptor through one broker:
    every row sequence equals the oracle's streaming scan, and windows of different scans shared launches"""
    import pgvectorscale_amd as P
    O = oracle
    ti = TestIndex(n=1500, dim_full=48, bits=2, R=24, distance=O.COSINE, seed=33, kind="gauss", deleted_frac=0.1, L_build=50)
    ix = ti.upload(gpu_ctx)
    broker = P.Broker(ix, max_batch=32, max_wait_us=20000)
    nthreads = 12
    q = ti.queries(nthreads, seed=5, kind="gauss")
    rows, errors, stats, work = {}, [], {}, {}
    start = threading.Barrier(nthreads)

    def backend(t):
        try:
            scan = broker.beginscan()
            start.wait()
            scan.rescan(q[t], search_list_size=30, rescore=10)
            out = []
            npull = 45 if t % 3 else 7  # past the first 16 rows (a shared launch) the scan continues on a cursor of its own
            for _ in range(npull):
                r = scan.gettuple()
                if r is None:
                    break
                out.append(r)
            rows[t] = out
            stats[t] = scan.stats()
            work[t] = scan.work()
            scan.endscan()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=backend, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    for t in range(nthreads):
        os_ = ti.oracle.scan(q[t], L=30, rescore=10)
        for (tid, node, d) in rows[t]:
            o = os_.gettuple()
            assert o is not None and node == o[0] and tid == o[1]
            assert np.float32(d).view(np.uint32) == np.float32(o[2]).view(np.uint32)
        assert len(rows[t]) == (45 if t % 3 else 7)
        # GreedySearchStats are the reference's after that many amgettuple calls, whether the rows came from the cursor or (7 rows)
        # from the shared launch alone, in which case the scan is replayed on a cursor when the statistics are asked for
        ref = os_.stats()
        for key in ("visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "full_distance_comparisons", "node_reads",
                    "node_heap_reads", "next_calls"):
            assert stats[t][key] == ref[key], (t, key, stats[t][key], ref[key])
        # continued, not repeated: one replay of the first 16 rows, then at most the prefetch allowance on top of the scan itself
        assert work[t]["visited_nodes"] <= 1.1 * ref["visited_nodes"] + 16, (work[t], ref)
        assert work[t]["retries"] == 0
    st = broker.stats()
    assert st["max_batch"] >= 3 and st["batches"] < st["scans"], st
    assert st["scans"] == nthreads  # one shared-launch request per scan: nothing was fetched again with a larger window
    assert st["tasks"] >= nthreads
    broker.close()
    ix.close()


def test_backends_with_different_snapshots_never_share_a_mask(gpu_ctx, oracle):
    """Backends see different snapshots, so a shared launch cannot take "the index's" visibility mask: every request names its
    snapshot, the dispatcher only groups scans of one snapshot and runs the group under that mask (AM/scan.rs:268-272: a candidate
    the snapshot cannot see is fetched, counted and dropped before the rescore window).  Three snapshots in flight at once."""
    import pgvectorscale_amd as P
    O = oracle
    ti = TestIndex(n=2000, dim_full=48, bits=2, R=24, distance=O.L2, seed=31, kind="clustered", L_build=48)
    ix = ti.upload(gpu_ctx)
    rng = np.random.default_rng(2)
    masks = {0: None, 1: (rng.random(ti.n) > 0.3).astype(np.uint8), 2: (rng.random(ti.n) > 0.6).astype(np.uint8)}
    # a direct caller's index-level mask must survive the broker's launches untouched, and must not leak into them
    own = (rng.random(ti.n) > 0.5).astype(np.uint8)
    ix.set_visibility(own)
    nthreads, per_thread = 12, 5
    q = ti.queries(nthreads * per_thread, seed=5, kind="clustered")
    snap_of = rng.integers(0, 3, len(q))
    want = {}
    for i in range(len(q)):
        ti.oracle.set_visibility(masks[int(snap_of[i])])
        oi, od, _ = ti.oracle.search_batch(q[i:i + 1], L=20, rescore=15, k=10)
        want[i] = (oi[0], od[0])
    ti.oracle.set_visibility(None)
    broker = P.Broker(ix, max_batch=64, max_wait_us=20000)
    with pytest.raises(P.VsError, match="no visibility mask"):
        broker.search(q[0], None, 20, 15, 10, snapshot=1)  # named before it was put
    broker.snapshot_put(1, masks[1])
    broker.snapshot_put(2, masks[2])
    got, errors = {}, []
    start = threading.Barrier(nthreads)

    def client(t):
        try:
            start.wait()
            for j in range(per_thread):
                i = t * per_thread + j
                got[i] = broker.search(q[i], None, 20, 15, 10, snapshot=int(snap_of[i]))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=client, args=(t,)) for t in range(nthreads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]
    for i in range(len(q)):
        assert (got[i][0] == want[i][0]).all(), (i, int(snap_of[i]))
        assert (got[i][2].view(np.uint32) == want[i][1].view(np.uint32)).all(), i
    assert broker.stats()["batches"] < len(q)  # scans of one snapshot did share launches
    # the index-level mask is what it was: a direct scan still runs under it
    ti.oracle.set_visibility(own)
    gi, _, _, _ = ix.search_batch(q[:4], search_list_size=20, rescore=15, k=10)
    oi, _, _ = ti.oracle.search_batch(q[:4], L=20, rescore=15, k=10)
    ti.oracle.set_visibility(None)
    assert (gi == oi).all()
    broker.snapshot_put(2, None)
    with pytest.raises(P.VsError, match="no visibility mask"):
        broker.search(q[0], None, 20, 15, 10, snapshot=2)
    broker.close()
    ix.close()


def test_broker_scan_under_a_snapshot_streams_like_the_oracle(gpu_ctx, oracle):
    """a backend's scan on a broker runs under ITS snapshot's visibility mask in the shared launch and on its cursor alike, and the
    prefetch hint makes the rows of a known LIMIT available without further continuations"""
    import pgvectorscale_amd as P
    O = oracle
    ti = TestIndex(n=1800, dim_full=48, bits=2, R=24, distance=O.L2, seed=35, kind="clustered", L_build=48)
    ix = ti.upload(gpu_ctx)
    rng = np.random.default_rng(4)
    mask = (rng.random(ti.n) > 0.4).astype(np.uint8)
    own = (rng.random(ti.n) > 0.5).astype(np.uint8)
    ix.set_visibility(own)  # a direct caller's mask: must neither leak into the broker's scans nor be lost
    broker = P.Broker(ix, max_batch=8, max_wait_us=1000)
    broker.snapshot_put(3, mask)
    q = ti.queries(2, seed=9, kind="clustered")
    scan = broker.beginscan()
    for snap, m in ((3, mask), (0, None)):
        scan.set_snapshot(snap)
        scan.rescan(q[0], search_list_size=12, rescore=9)
        ti.oracle.set_visibility(m)
        os_ = ti.oracle.scan(q[0], L=12, rescore=9)
        for j in range(80):
            r, o = scan.gettuple(), os_.gettuple()
            assert r is not None and o is not None and r[1] == o[0] and r[0] == o[1], (snap, j)
            assert np.float32(r[2]).view(np.uint32) == np.float32(o[2]).view(np.uint32)
        g, ref = scan.stats(), os_.stats()
        for key in ("visited_nodes", "quantized_distance_comparisons", "full_distance_comparisons", "node_heap_reads", "next_calls"):
            assert g[key] == ref[key], (snap, key, g[key], ref[key])
    # LIMIT 40 announced: one shared launch hands out all 40 rows, no cursor is ever opened
    tasks_before = broker.stats()["tasks"]
    scan.rescan(q[1], search_list_size=12, rescore=9)
    scan.prefetch(40)
    ti.oracle.set_visibility(None)
    os_ = ti.oracle.scan(q[1], L=12, rescore=9)
    for j in range(40):
        r, o = scan.gettuple(), os_.gettuple()
        assert r[1] == o[0] and r[0] == o[1], j
    assert broker.stats()["tasks"] == tasks_before
    assert scan.work()["launches"] == 0
    scan.endscan()
    ti.oracle.set_visibility(own)
    gi, _, _, _ = ix.search_batch(q[:2], search_list_size=20, rescore=15, k=10)
    oi, _, _ = ti.oracle.search_batch(q[:2], L=20, rescore=15, k=10)
    ti.oracle.set_visibility(None)
    assert (gi == oi).all()
    broker.close()
    ix.close()


def test_broker_scan_statistics_inside_the_shared_window_and_scans_that_end_there(gpu_ctx, oracle):
    """found by the differential fuzzer: statistics asked for after 1, 2, 9 rows — all still rows of the shared launch — need the
    replay cursor extended each time, and a scan that ends inside the shared window must end there, with or without a replay"""
    import pgvectorscale_amd as P
    O = oracle
    for kw, L, rescore, pull in ((dict(n=900, dim_full=24, bits=2, R=12, distance=O.L2, seed=3, kind="gauss", L_build=30), 20, 6, 40),
                                 (dict(n=3, dim_full=16, bits=2, R=4, distance=O.L2, seed=4, kind="gauss", L_build=8), 10, 5, 10),
                                 (dict(n=11, dim_full=16, bits=1, R=4, distance=O.COSINE, seed=5, kind="gauss", L_build=8), 4, 0, 30)):
        ti = TestIndex(**kw)
        ix = ti.upload(gpu_ctx)
        broker = P.Broker(ix, max_batch=4, max_wait_us=0)
        q = ti.queries(2, seed=8, kind="gauss")
        scan = broker.beginscan()
        for qi in range(2):
            scan.rescan(q[qi], search_list_size=L, rescore=rescore)
            os_ = ti.oracle.scan(q[qi], L=L, rescore=rescore)
            for j in range(pull):
                r, o = scan.gettuple(), os_.gettuple()
                assert (r is None) == (o is None), (kw["n"], qi, j)
                if r is not None:
                    assert r[1] == o[0] and r[0] == o[1], (kw["n"], qi, j)
                if j in (0, 1, 8, 20) or r is None:
                    g, ref = scan.stats(), os_.stats()
                    for key in ("visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "full_distance_comparisons",
                                "node_reads", "node_heap_reads", "next_calls"):
                        assert g[key] == ref[key], (kw["n"], qi, j, key, g[key], ref[key])
        scan.endscan()
        broker.close()
        ix.close()
