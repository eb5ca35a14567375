"""vs_index_view: two handles on one device-resident index, each with its own context (stream) and workspace; batches submitted
alternately through them, the second one enqueued before the first is finished, give the rows of the oracle."""
import ctypes as C

import numpy as np
import pytest

import pgvectorscale_amd as P
from helpers import cached_index

pytestmark = pytest.mark.gpu


def test_batches_through_two_views_overlap_and_are_exact(gpu_ctx):
    ti = cached_index(n=3000, dim_full=96, bits=2, R=32, distance=1, seed=41, kind="gauss", L_build=60)
    ix = ti.upload(gpu_ctx)
    ctx2 = P.Context(0)
    vw = ix.view(ctx2)
    try:
        L, S, k, nq = 40, 30, 10, 48
        batches = [ti.queries(nq, seed=100 + b, kind="gauss") for b in range(4)]
        want = [ti.oracle.search_batch(q, L=L, rescore=S, k=k) for q in batches]
        handles = [(ix, gpu_ctx), (vw, ctx2)]
        bufs = []
        for h, c in handles:
            bufs.append((c.alloc(nq * ti.dim_full * 4), c.alloc(nq * k * 4), c.alloc(nq * k * 4)))
        pending = None
        got = [None] * len(batches)

        def collect(b):
            h, c = handles[b % 2]
            h.search_batch_dev_finish()
            _, oi, od = bufs[b % 2]
            got[b] = (c.download(oi, np.empty((nq, k), np.uint32)), c.download(od, np.empty((nq, k), np.float32)))

        for b, q in enumerate(batches):
            h, c = handles[b % 2]
            qb, oi, od = bufs[b % 2]
            c.upload(qb, np.ascontiguousarray(q, np.float32))
            h.search_batch_dev(qb, nq, L, S, k, oi, None, od)  # enqueued while the other handle's batch is still in flight
            if pending is not None:
                collect(pending)
            pending = b
        collect(pending)
        for b in range(len(batches)):
            wi, wd, _ = want[b]
            assert (got[b][0] == wi).all()
            assert (got[b][1].view(np.uint32) == wd.view(np.uint32)).all()
    finally:
        vw.close()
        ix.close()
        ctx2.close()


def test_entry_points_that_move_the_arrays_refuse_while_a_view_is_alive(gpu_ctx):
    """a view holds its owner's device pointers: replacing the label sets or the start map frees what a lane / a second stream / a
    vs_multi shard could still launch on — refused with VS_ERR_STATE until the last view is gone (freed in either order)"""
    ti = cached_index(n=600, dim_full=32, bits=2, R=16, distance=1, seed=43, kind="gauss", L_build=30, n_labels=4)
    ix = ti.upload(gpu_ctx)
    ctx2 = P.Context(0)
    vw = ix.view(ctx2)
    vw2 = vw.view(ctx2)  # a view of a view is a view of the owner
    try:
        for _ in range(2):
            with pytest.raises(P.VsError) as ei:
                ix.set_labels(ti.label_off, ti.label_val)
            assert ei.value.code == -5 and "view" in str(ei.value)
            with pytest.raises(P.VsError):
                ix.set_start_nodes(ti.start, ti.label_starts)
            vw2.close()
            vw2 = vw.view(ctx2)
        # ... and a VIEW handle is never allowed to move them (its live-view count is not the owner's)
        with pytest.raises(P.VsError) as ei:
            vw.set_labels(ti.label_off, ti.label_val)
        assert ei.value.code == -5 and "view" in str(ei.value)
        with pytest.raises(P.VsError):
            vw2.set_start_nodes(ti.start, ti.label_starts)
        vw2.close()
        vw.close()
        ix.set_labels(ti.label_off, ti.label_val)  # no view left: allowed again
        ix.set_start_nodes(ti.start, ti.label_starts)
        q = ti.queries(8, seed=3, kind="gauss")
        gi, _, gd, _ = ix.search_batch(q, search_list_size=20, rescore=10, k=5)
        oi, od, _ = ti.oracle.search_batch(q, L=20, rescore=10, k=5)
        assert (gi == oi).all()
    finally:
        ix.close()
        ctx2.close()


def test_a_stale_view_does_not_touch_the_count_of_a_new_index_at_the_same_address(gpu_ctx):
    """the registry of live views is keyed by an owner id, not by the owner's address: freeing a view whose owner is gone must not
    decrement the count of an index that was allocated where the old owner lived (round-4 advisor finding)"""
    ti = cached_index(n=600, dim_full=32, bits=2, R=16, distance=1, seed=43, kind="gauss", L_build=30, n_labels=4)
    ctx2 = P.Context(0)
    ix = ti.upload(gpu_ctx)
    stale = ix.view(ctx2)
    ix.close()  # (the library warns: the view must not be used any more — it is only freed below)
    fresh = [ti.upload(gpu_ctx) for _ in range(4)]  # one of them is likely to reuse the freed handle's address
    views = [f.view(ctx2) for f in fresh]
    try:
        stale.close()
        for f in fresh:  # every new index still counts its own view
            with pytest.raises(P.VsError) as ei:
                f.set_labels(ti.label_off, ti.label_val)
            assert ei.value.code == -5
    finally:
        for v in views:
            v.close()
        for f in fresh:
            f.close()
        ctx2.close()
