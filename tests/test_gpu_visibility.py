"""Heap visibility inside the rescore window: next_with_resort fetches the heap tuple of every candidate the stream hands it and
drops the ones index_fetch_tuple cannot see under the scan's snapshot BEFORE they enter the window (AM/scan.rs:258-272,
AM/sbq/storage.rs:313-317).  vs_index_set_visibility carries that snapshot result across the boundary as one byte per node; the
rows, distances and counters must equal the oracle's with the same mask — in every kernel regime, through the batched API and
through the amrescan / amgettuple mirror — and the mask must be ignored where the reference does not look at the heap
(query_rescore = 0, the SBQ-ordered stream)."""
import os

import numpy as np
import pytest

from helpers import cached_index

pytestmark = pytest.mark.gpu

REGIMES = {"default": {}, "tableless": {"VS_F_LDS_MAX_INS": "0"}, "general_kernel": {"VS_FAST": "0"}}
KW = dict(n=3000, dim_full=64, bits=2, R=32, distance=1, seed=11, kind="gauss", L_build=64, n_labels=6, deleted_frac=0.1)


def _mask(n, seed, frac):
    return (np.random.default_rng(seed).random(n) >= frac).astype(np.uint8)


@pytest.fixture
def vis_index(gpu_ctx):
    ti = cached_index(**KW)
    ix = ti.upload(gpu_ctx)
    yield ti, ix
    ti.oracle.set_visibility(None)
    ix.close()


@pytest.mark.parametrize("regime", list(REGIMES))
def test_invisible_rows_never_enter_the_window(vis_index, regime):
    ti, ix = vis_index
    q = ti.queries(64, seed=5, kind="gauss")
    rng = np.random.default_rng(6)
    keys = [sorted(set(int(v) for v in rng.integers(1, 7, int(rng.integers(1, 3))))) for _ in range(len(q))]
    saved = {k: os.environ.get(k) for k in REGIMES[regime]}
    try:
        os.environ.update(REGIMES[regime])
        base_i, _, _, base_st = ix.search_batch(q, search_list_size=60, rescore=30, k=10)
        for frac in (0.1, 0.5, 1.0):
            vis = _mask(ti.n, 7, frac)
            ti.oracle.set_visibility(vis)
            ix.set_visibility(vis)
            for kk in (None, keys):
                oi, od, ost = ti.oracle.search_batch(q, L=60, rescore=30, k=10, qlabels=kk)
                gi, _, gd, gst = ix.search_batch(q, search_list_size=60, rescore=30, k=10, qlabels=kk)
                assert (gi == oi).all()
                assert (gd.view(np.uint32) == od.view(np.uint32)).all()
                for c in ("visited_nodes", "quantized_distance_comparisons", "full_distance_comparisons", "node_heap_reads", "next_calls"):
                    assert gst[c] == ost[c], (frac, c, gst[c], ost[c])
                live = gi[gi != 0xFFFFFFFF]
                assert vis[live].all()  # no invisible row is ever returned
            if frac == 1.0:
                assert (gi == 0xFFFFFFFF).all()
        # query_rescore = 0: the access method does not fetch the heap, the mask has no effect (AM/scan.rs:249-251) ...
        vis = _mask(ti.n, 7, 0.5)
        ti.oracle.set_visibility(vis)
        ix.set_visibility(vis)
        oi, _, _ = ti.oracle.search_batch(q, L=60, rescore=0, k=10)
        gi, _, _, _ = ix.search_batch(q, search_list_size=60, rescore=0, k=10)
        assert (gi == oi).all() and not vis[gi[gi != 0xFFFFFFFF]].all()
        # ... and neither does the SBQ-ordered stream
        si, sh, _ = ix.stream_batch(q, search_list_size=60, m=40)
        ti.oracle.set_visibility(None)
        ti_i, ti_h, _ = ti.oracle.stream_batch(q, L=60, m=40)
        assert (si == ti_i).all() and (sh == ti_h).all()
        # clearing the mask restores the first answer
        ix.set_visibility(None)
        gi, _, _, gst = ix.search_batch(q, search_list_size=60, rescore=30, k=10)
        assert (gi == base_i).all() and gst["full_distance_comparisons"] == base_st["full_distance_comparisons"]
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_gettuple_mirror_with_a_snapshot(vis_index):
    """amrescan + amgettuple one row at a time, 10 % of the heap invisible, past the first prefetched window"""
    import pgvectorscale_amd as P
    ti, ix = vis_index
    vis = _mask(ti.n, 9, 0.1)
    ti.oracle.set_visibility(vis)
    ix.set_visibility(vis)
    q = ti.queries(3, seed=8, kind="gauss")
    sc = P.IndexScan(ix)
    for qi, labels in ((0, None), (1, [2]), (2, [1, 5])):
        sc.rescan(q[qi], labels=labels, search_list_size=40, rescore=20)
        os_ = ti.oracle.scan(q[qi], labels, 40, 20)
        for _ in range(150):
            a, b = sc.gettuple(), os_.gettuple()
            if b is None:
                assert a is None
                break
            # (the mirror returns (heap tid, node, distance), the oracle (node, heap tid, distance))
            assert a is not None and a[0] == b[1] and a[1] == b[0] and np.float32(a[2]).view(np.uint32) == np.float32(b[2]).view(np.uint32)
            assert vis[a[1]]
    sc.endscan()


def test_mask_shape_is_checked(vis_index):
    ti, ix = vis_index
    with pytest.raises(ValueError):
        ix.set_visibility(np.ones(ti.n - 1, np.uint8))
