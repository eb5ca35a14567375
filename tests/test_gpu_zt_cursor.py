"""The amgettuple cursor (AM/scan.rs:162-174,370-405): a scan keeps its ListSearchResult and resort_buffer on the device between
calls and CONTINUES when the executor asks for more rows.  Pulled one row at a time, the rows and — after 1, 17, 65, 1000 ... rows —
the GreedySearchStats must be the oracle's (the reference's streaming iterator restated), the device must not have done more
than 1.1 x the work of one scan of the final length, and nothing may be run again from the start."""
import numpy as np
import pytest

from helpers import TestIndex

pytestmark = pytest.mark.gpu

STAT_KEYS = ("visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "full_distance_comparisons", "node_reads",
             "node_heap_reads", "next_calls")


def _same_stats(g, o, where):
    for key in STAT_KEYS:
        assert g[key] == o[key], (where, key, g[key], o[key])


def _pull_and_compare(scan, os_, checkpoints, limit):
    pulled = 0
    ended = False
    while pulled < limit:
        r, o = scan.gettuple(), os_.gettuple()
        assert (r is None) == (o is None), pulled
        if r is None:
            ended = True
            break
        assert r[1] == o[0] and r[0] == o[1], pulled
        if not (np.isnan(r[2]) and np.isnan(o[2])):
            assert np.float32(r[2]).view(np.uint32) == np.float32(o[2]).view(np.uint32), pulled
        pulled += 1
        if pulled in checkpoints:
            _same_stats(scan.stats(), os_.stats(), f"after {pulled} rows")
    return pulled, ended


@pytest.mark.parametrize("name,kw,L,rescore,labels", [
    ("l2_window", dict(n=3000, dim_full=64, bits=2, R=24, seed=5, kind="clustered"), 20, 50, None),
    ("no_window", dict(n=3000, dim_full=64, bits=2, R=24, seed=5, kind="clustered"), 10, 0, None),
    ("tiny_list", dict(n=2500, dim_full=48, bits=1, R=16, seed=6, kind="gauss"), 1, 7, None),
    ("labels_deleted", dict(n=3000, dim_full=48, bits=2, R=20, seed=41, kind="uniform", n_labels=4, deleted_frac=0.15), 30, 10, [2, 3]),
])
def test_rows_and_stats_one_row_at_a_time(gpu_ctx, oracle, name, kw, L, rescore, labels):
    ti = TestIndex(distance=oracle.L2, **kw)
    ix = ti.upload(gpu_ctx)
    q = ti.queries(3, seed=11, kind=kw["kind"])
    scan = ix.beginscan()
    for qi in range(2):
        scan.rescan(q[qi], labels=labels, search_list_size=L, rescore=rescore)
        os_ = ti.oracle.scan(q[qi], labels=labels, L=L, rescore=rescore)
        pulled, ended = _pull_and_compare(scan, os_, {1, 2, 17, 65, 300, 1000}, 1000)
        _same_stats(scan.stats(), os_.stats(), "at the end")
        work = scan.work()
        ref = os_.stats()
        assert work["retries"] == 0
        # the scan was continued, never repeated: what the device did is at most 1.1 x one scan of the final length (+ the
        # eight rows of prefetch a short scan is allowed)
        assert work["visited_nodes"] <= 1.1 * ref["visited_nodes"] + 8, (work, ref)
        assert work["quantized_distance_comparisons"] <= 1.1 * ref["quantized_distance_comparisons"] + 8 * ti.R
        assert work["launches"] <= 8 + pulled // 16 + 24
        if ended:  # calls past the end keep asking next() in vain, like the reference's iterator
            for _ in range(3):
                assert scan.gettuple() is None and os_.gettuple() is None
            _same_stats(scan.stats(), os_.stats(), "past the end")
    scan.endscan()
    ix.close()


def test_exhaustive_scan_and_rescan_on_one_descriptor(gpu_ctx, oracle):
    """every live row of a small index exactly once, in the oracle's order; then the descriptor is reused (amrescan)"""
    ti = TestIndex(n=700, dim_full=32, bits=2, R=12, distance=oracle.COSINE, seed=9, kind="gauss", deleted_frac=0.1)
    ix = ti.upload(gpu_ctx)
    q = ti.queries(2, seed=3, kind="gauss")
    scan = ix.beginscan()
    scan.rescan(q[0], search_list_size=3, rescore=25)
    os_ = ti.oracle.scan(q[0], L=3, rescore=25)
    pulled, ended = _pull_and_compare(scan, os_, {1, 100, 400}, 10_000)
    live = int(((ti.tids & np.uint64(0xFFFF)) != 0).sum())
    assert ended and pulled == live
    _same_stats(scan.stats(), os_.stats(), "exhausted")
    assert scan.work()["retries"] == 0
    scan.rescan(None, search_list_size=5, rescore=0)  # the SQL-NULL query (AM/labels/mod.rs:214-216)
    os_ = ti.oracle.scan(None, L=5, rescore=0)
    pulled, ended = _pull_and_compare(scan, os_, {1, 50}, 10_000)
    assert ended and pulled == live
    scan.endscan()
    ix.close()


def test_invisible_heap_tuples_under_the_cursor(gpu_ctx, oracle):
    """candidates the snapshot cannot see are fetched, counted and dropped before the window (AM/scan.rs:268-272)"""
    ti = TestIndex(n=2000, dim_full=48, bits=2, R=16, distance=oracle.L2, seed=21, kind="clustered")
    ix = ti.upload(gpu_ctx)
    vis = (np.random.default_rng(4).random(ti.n) > 0.3).astype(np.uint8)
    ix.set_visibility(vis)
    ti.oracle.set_visibility(vis)
    q = ti.queries(1, seed=8, kind="clustered")
    scan = ix.beginscan()
    scan.rescan(q[0], search_list_size=15, rescore=20)
    os_ = ti.oracle.scan(q[0], L=15, rescore=20)
    _pull_and_compare(scan, os_, {1, 9, 64, 333}, 400)
    _same_stats(scan.stats(), os_.stats(), "at the end")
    ti.oracle.set_visibility(None)
    scan.endscan()
    ix.close()


def test_scan_that_outgrows_its_capacities_is_restarted_and_fast_forwarded(gpu_ctx, oracle, monkeypatch):
    """capacities sized for a few dozen rows (VS_CURSOR_HORIZON) on a scan that goes on for thousands: every overflow restarts the
    scan with four times the room, the rows already handed out are skipped, the stream continues seamlessly"""
    monkeypatch.setenv("VS_CURSOR_HORIZON", "4")
    monkeypatch.setenv("VS_HL", "63")
    ti = TestIndex(n=6000, dim_full=32, bits=2, R=24, distance=oracle.L2, seed=13, kind="uniform")
    ix = ti.upload(gpu_ctx)
    q = ti.queries(1, seed=5)
    scan = ix.beginscan()
    scan.rescan(q[0], search_list_size=4, rescore=30)
    os_ = ti.oracle.scan(q[0], L=4, rescore=30)
    pulled, ended = _pull_and_compare(scan, os_, {1, 500, 3000}, 10_000)
    assert ended and pulled == ti.n
    _same_stats(scan.stats(), os_.stats(), "exhausted")
    w = scan.work()
    assert w["retries"] >= 1  # 6000 rows are far beyond the initial horizon
    scan.endscan()
    ix.close()
