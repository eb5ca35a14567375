"""Batched cursor continuations (vs_scanpool_*, csrc/vs_scanpool.cpp): many streamed scans — each one a backend's amrescan / amgettuple
cursor (AM/scan.rs:162-174,369-436) — continued by launches they SHARE.  Every slot must hand out the oracle's rows in the oracle's
order, chunk after chunk, with the oracle's GreedySearchStats after every chunk (the counters are recorded per emitted stream row, so a
slot that was carried further ahead by another slot's larger request shows nothing of it), whatever mix of slots a fetch names: all of
them, a few, keyed and unkeyed scans together, scans that end early, a slot that is rescanned while the others go on."""
import numpy as np
import pytest

import pgvectorscale_amd as P
from helpers import TestIndex

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["resumable_fast_kernel", "general_kernel"])
def pool_kernel(request, monkeypatch):
    """round 6: a pool continues its scans with the resumable instantiation of k_search_fast where the index allows it; VS_POOL_FAST=0 keeps
    the general kernel's path (plain storage, wide codes) — every test holds both to the oracle"""
    monkeypatch.setenv("VS_POOL_FAST", "1" if request.param == "resumable_fast_kernel" else "0")
    return request.param

STAT_KEYS = ("visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "full_distance_comparisons", "node_reads",
             "node_heap_reads", "next_calls")


def _check_chunk(got_rows, ids, tids, dist, oscan, k, where):
    """the next k amgettuple calls of the oracle's scan against one slot's chunk"""
    n = 0
    for j in range(k):
        o = oscan.gettuple()
        if o is None:
            break
        assert j < got_rows, (where, "the pool ended the scan early", j)
        assert ids[j] == o[0] and tids[j] == o[1], (where, j)
        if not (np.isnan(dist[j]) and np.isnan(o[2])):
            assert np.float32(dist[j]).view(np.uint32) == np.float32(o[2]).view(np.uint32), (where, j)
        n += 1
    assert got_rows == n, (where, got_rows, n)
    return n


@pytest.mark.parametrize("name,kw,L,rescore", [
    ("l2_window", dict(n=3000, dim_full=64, bits=2, R=24, seed=5, kind="clustered"), 20, 50),
    ("no_window", dict(n=3000, dim_full=64, bits=2, R=24, seed=5, kind="clustered"), 10, 0),
    ("labels_deleted", dict(n=3000, dim_full=48, bits=2, R=20, seed=41, kind="uniform", n_labels=4, deleted_frac=0.15), 30, 10),
])
def test_pooled_scans_hand_out_the_oracles_rows_and_stats(gpu_ctx, oracle, name, kw, L, rescore):
    ti = TestIndex(distance=oracle.L2, **kw)
    ix = ti.upload(gpu_ctx)
    G, k = 12, 16
    q = ti.queries(G + 2, seed=11, kind=kw["kind"])
    labeled = "n_labels" in kw
    keys = [None] * G
    if labeled:  # keyed and unkeyed scans in one pool: two launches per round
        rng = np.random.default_rng(3)
        keys = [None if i % 3 == 0 else sorted(set(int(v) for v in rng.integers(1, 5, int(rng.integers(1, 3))))) for i in range(G)]
    pool = P.ScanPool(ix, G, search_list_size=L, rescore=rescore, kmax=k, rows_cap=2048)
    try:
        oscans = []
        for i in range(G):
            pool.rescan(i, q[i], labels=keys[i])
            oscans.append(ti.oracle.scan(q[i], labels=keys[i], L=L, rescore=rescore))
        handed = [0] * G
        ended = [False] * G
        rng = np.random.default_rng(9)
        for rnd in range(40):
            # who asks in this round: everybody, a random few, or one slot alone; chunk sizes vary
            if rnd % 3 == 0:
                who = [i for i in range(G) if not ended[i]]
            elif rnd % 3 == 1:
                who = [i for i in range(G) if not ended[i] and rng.random() < 0.4]
            else:
                who = [i for i in range(G) if not ended[i]][:1]
            if not who:
                break
            kk = int(rng.choice([1, 5, k]))
            rows, ids, tids, dist = pool.fetch(who, kk)
            for a, i in enumerate(who):
                assert rows[a] >= 0, (rnd, i, rows[a])
                n = _check_chunk(int(rows[a]), ids[a], tids[a], dist[a], oscans[i], kk, (name, rnd, i))
                handed[i] += n
                ended[i] = n < kk
                g, o = pool.stats(i), oscans[i].stats()
                for key in STAT_KEYS:
                    assert g[key] == o[key], (name, rnd, i, key, g[key], o[key])
            if rnd == 7:  # one slot is rescanned with another query while the others go on
                pool.rescan(1, q[G], labels=keys[1])
                oscans[1] = ti.oracle.scan(q[G], labels=keys[1], L=L, rescore=rescore)
                handed[1], ended[1] = 0, False
        assert sum(handed) > 40 * 4
        w = pool.work()
        assert w["rounds"] <= w["launches"] <= 2 * w["rounds"]  # one launch per round (two with keyed AND unkeyed scans in it)
    finally:
        pool.close()
        ix.close()


def test_a_pooled_scan_equals_the_single_cursor_and_ends_like_it(gpu_ctx, oracle):
    """every live row of a small index exactly once through a pool slot, as through vs_gettuple; calls past the end keep asking next() in
    vain; a scan that outgrows the pool's row budget fails alone (VS_ERR_CAPACITY) while its neighbour is served"""
    ti = TestIndex(n=700, dim_full=32, bits=2, R=12, distance=oracle.COSINE, seed=9, kind="gauss", deleted_frac=0.1)
    ix = ti.upload(gpu_ctx)
    q = ti.queries(2, seed=3, kind="gauss")
    pool = P.ScanPool(ix, 4, search_list_size=3, rescore=25, kmax=64, rows_cap=1024)
    scan = ix.beginscan()
    try:
        pool.rescan(2, q[0])
        scan.rescan(q[0], search_list_size=3, rescore=25)
        os_ = ti.oracle.scan(q[0], L=3, rescore=25)
        total = 0
        while True:
            rows, ids, tids, dist = pool.fetch([2], 64)
            n = int(rows[0])
            for j in range(n):
                r = scan.gettuple()
                assert r is not None and r[1] == ids[0][j] and r[0] == tids[0][j]
                assert np.float32(r[2]).view(np.uint32) == dist[0][j].view(np.uint32) or (np.isnan(r[2]) and np.isnan(dist[0][j]))
            total += _check_chunk(n, ids[0], tids[0], dist[0], os_, 64, ("exhaustive", total))
            if n < 64:
                break
        live = int(((ti.tids & np.uint64(0xFFFF)) != 0).sum())
        assert total == live
        for _ in range(2):  # past the end
            rows, _, _, _ = pool.fetch([2], 8)
            assert rows[0] == 0 and os_.gettuple() is None
        g, o = pool.stats(2), os_.stats()
        for key in STAT_KEYS:
            assert g[key] == o[key], (key, g[key], o[key])
    finally:
        scan.endscan()
        pool.close()
    # a row budget smaller than the scan: the slot fails alone
    pool = P.ScanPool(ix, 2, search_list_size=3, rescore=25, kmax=32, rows_cap=96)
    try:
        pool.rescan(0, q[0])
        pool.rescan(1, q[1])
        r0, _, _, _ = pool.fetch([0], 32)
        assert r0[0] == 32
        r0, _, _, _ = pool.fetch([0], 32)  # 25 + 64 - 1 = 88 stream rows: still inside
        assert r0[0] == 32
        rows, ids, _, _ = pool.fetch([0, 1], 32)  # slot 0 would need 120 stream rows
        assert rows[0] == -4 and rows[1] == 32  # VS_ERR_CAPACITY for slot 0 alone
        want = ti.oracle.search_batch(q[1][None, :], L=3, rescore=25, k=32)[0][0]
        assert (ids[1] == want).all()
    finally:
        pool.close()
        ix.close()


def test_a_round_carries_the_listed_scans_that_are_streamed_ahead(gpu_ctx, oracle):
    """round 6: backends that stream side by side but out of phase (one scan started earlier, chunks of different sizes before they meet) —
    a round continues every listed scan that is being streamed, not only the ones that ran out of rows, so the group needs the rounds of
    its neediest scan instead of one round per fetch; the rows and the per-row counters are the same with and without (VS_POOL_TOPUP=0)"""
    ti = TestIndex(n=4000, dim_full=64, bits=2, R=24, distance=oracle.L2, seed=17, kind="clustered")
    ix = ti.upload(gpu_ctx)
    G, k = 8, 8
    q = ti.queries(G, seed=23, kind="clustered")
    out, rounds = {}, {}
    try:
        for mode in ("1", "0", "prefetch"):
            # "1" / "0": rounds the caller waits for only, with / without carrying the other listed scans; "prefetch": the default — the next
            # round of the streamed scans is launched at the end of a fetch and booked by a later call (VS_EMU_EVENT_LAG walks both ways)
            P.set_option("VS_POOL_TOPUP", "0" if mode == "0" else "1")
            P.set_option("VS_POOL_PREFETCH", "1" if mode == "prefetch" else "0")
            pool = P.ScanPool(ix, G, search_list_size=20, rescore=10, kmax=k, rows_cap=1024)
            try:
                for i in range(G):
                    pool.rescan(i, q[i])
                got = [[] for _ in range(G)]
                for i in range(G):  # out of phase: scan i has pulled i chunks of i + 1 rows before the others join
                    for _ in range(i):
                        rows, ids, tids, dist = pool.fetch([i], i + 1)
                        got[i] += [(int(ids[0][j]), int(tids[0][j]), float(dist[0][j])) for j in range(int(rows[0]))]
                w0 = pool.work()["rounds"]
                for _ in range(24):  # ... then side by side
                    rows, ids, tids, dist = pool.fetch(list(range(G)), k)
                    for i in range(G):
                        got[i] += [(int(ids[i][j]), int(tids[i][j]), float(dist[i][j])) for j in range(int(rows[i]))]
                rounds[mode] = pool.work()["rounds"] - w0
                out[mode] = (got, [pool.stats(i) for i in range(G)])
            finally:
                pool.close()
        for other in ("0", "prefetch"):
            assert out["1"][0] == out[other][0], other
            for a, b in zip(out["1"][1], out[other][1]):
                for key in STAT_KEYS:
                    assert a[key] == b[key], (other, key)
        os0 = ti.oracle.scan(q[3], L=20, rescore=10)
        for j, row in enumerate(out["1"][0][3]):
            o = os0.gettuple()
            assert o is not None and row[0] == o[0] and row[1] == o[1], j
        assert rounds["1"] < rounds["0"], rounds  # (24 fetches of 8 scans out of phase on the interpreter: 12 rounds against 19)
    finally:
        P.set_option("VS_POOL_TOPUP", None)
        P.set_option("VS_POOL_PREFETCH", None)
        ix.close()
