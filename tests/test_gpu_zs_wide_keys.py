"""Scan keys of any width (LabelSet is an unbounded sorted Vec<i16>, AM/labels/mod.rs:19-37): keys of up to 64 distinct labels ride in
LDS, wider ones are handed from the LDS-resident kernel to the general kernel, which reads them from global memory.  Rows, stream
and counters must be the oracle's either way, through every entry point (host batch, device batch, the amgettuple cursor)."""
import ctypes as C

import numpy as np
import pytest

from helpers import TestIndex

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wide(gpu_ctx, oracle):
    ti = TestIndex(n=2500, dim_full=48, bits=2, R=20, distance=oracle.L2, seed=77, kind="uniform", n_labels=300)
    ix = ti.upload(gpu_ctx)
    yield ti, ix
    ix.close()


def _keys(rng, nq):
    keys = []
    for i in range(nq):
        width = [1, 3, 64, 65, 100, 300, 0, 200][i % 8]
        keys.append(sorted(int(v) for v in rng.choice(np.arange(1, 401), size=width, replace=False)))
    return keys


@pytest.mark.parametrize("fast", ["1", "0"])
def test_wide_keys_host_batch(wide, monkeypatch, fast):
    monkeypatch.setenv("VS_FAST", fast)
    ti, ix = wide
    rng = np.random.default_rng(5)
    q = ti.queries(24, seed=4)
    keys = _keys(rng, 24)
    keys[2] = keys[2][::-1] + keys[2][:5]  # unsorted with duplicates: LabelSet::from sorts and de-duplicates
    gi, gt, gd, gst = ix.search_batch(q, search_list_size=20, rescore=12, k=10, qlabels=keys)
    oi, od, ost = ti.oracle.search_batch(q, L=20, rescore=12, k=10, qlabels=keys)
    assert (gi == oi).all()
    assert (gd.view(np.uint32) == od.view(np.uint32))[gi != 0xFFFFFFFF].all()
    for key in ("visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "node_reads", "full_distance_comparisons"):
        assert gst[key] == ost[key], key
    if fast == "1":
        assert gst["fallback_scans"] >= sum(len(set(k)) > 64 for k in keys)  # the wide keys ran on the general kernel
    # every row satisfies its key
    for i, key in enumerate(keys):
        for node in gi[i][gi[i] != 0xFFFFFFFF]:
            ls = set(ti.label_val[ti.label_off[node]:ti.label_off[node + 1]].tolist())
            assert ls & set(key)


def test_wide_keys_device_batch_and_cursor(wide, gpu_ctx):
    ti, ix = wide
    ctx = gpu_ctx
    rng = np.random.default_rng(6)
    nq, k = 16, 10
    q = ti.queries(nq, seed=9)
    keys = _keys(rng, nq)
    off = np.zeros(nq + 1, np.uint32)
    off[1:] = np.cumsum([len(x) for x in keys])
    val = np.array([v for x in keys for v in x], np.int16)
    d_q = ctx.alloc(q.nbytes)
    d_val = ctx.alloc(max(val.nbytes, 2))
    d_off = ctx.alloc(off.nbytes)
    d_out = ctx.alloc(nq * k * 4)
    ctx.upload(d_q, q)
    ctx.upload(d_val, val)
    ctx.upload(d_off, off)
    ix.search_batch_dev(d_q, nq, 20, 12, k, d_out, d_qlabels=d_val, d_qlabel_off=d_off)
    ix.search_batch_dev_finish()
    gi = ctx.download(d_out, np.empty((nq, k), np.uint32))
    oi, _, _ = ti.oracle.search_batch(q, L=20, rescore=12, k=k, qlabels=keys)
    assert (gi == oi).all()  # (the device-resident entry point used to read only the first 64 labels of a key)
    for p in (d_q, d_val, d_off, d_out):
        ctx.free(p)
    scan = ix.beginscan()
    for i in (4, 5):  # 100 and 300 labels
        scan.rescan(q[i], labels=keys[i], search_list_size=15, rescore=8)
        os_ = ti.oracle.scan(q[i], labels=keys[i], L=15, rescore=8)
        for _ in range(120):
            r, o = scan.gettuple(), os_.gettuple()
            assert (r is None) == (o is None)
            if r is None:
                break
            assert r[1] == o[0]
        g, o = scan.stats(), os_.stats()
        assert g["visited_nodes"] == o["visited_nodes"] and g["quantized_distance_comparisons"] == o["quantized_distance_comparisons"]
    scan.endscan()


def test_label_masks_for_any_label_values(gpu_ctx, oracle, monkeypatch):
    """<= 64 DISTINCT labels of any smallint value: node label sets become 64-bit masks through the per-index label -> bit table
    (one load + AND per fresh neighbor in the LDS-resident kernel); keys may name labels the index does not know."""
    O = oracle
    ti = TestIndex(n=2200, dim_full=48, bits=2, R=20, distance=O.L2, seed=78, kind="uniform", n_labels=6)
    lut = np.zeros(7, np.int16)
    lut[1:] = [-300, -1, 0, 63, 64, 32767]  # monotone: sets stay sorted
    ti.label_val = lut[ti.label_val]
    ti.label_starts = {int(lut[l]): n for l, n in ti.label_starts.items()}
    ti.oracle = O.OracleIndex(codes=ti.codes, nbrs=ti.nbrs, heap_tids=ti.tids, vecs=ti.vecs, mean=ti.mean, m2=ti.m2, count=ti.count,
                              bits=ti.bits, dim_index=ti.dim_index, num_neighbors=ti.R, distance_type=ti.distance,
                              default_start=ti.start, label_off=ti.label_off, label_val=ti.label_val, label_starts=ti.label_starts)
    ix = ti.upload(gpu_ctx)
    q = ti.queries(18, seed=2)
    keys = [[-300], [64, 32767], [0], [5], [-1, 63], [-32768, 64], [32767], [-300, -1, 0, 63, 64, 32767], [1, 2, 3]] * 2
    for regime in ({}, {"VS_F_LDS_MAX_INS": "0"}, {"VS_F_NBRMASK": "1"}, {"VS_F_NBRMASK": "1", "VS_F_LDS_MAX_INS": "0"}, {"VS_FAST": "0"}):
        for k_, v_ in regime.items():
            monkeypatch.setenv(k_, v_)
        gi, _, gd, gst = ix.search_batch(q, search_list_size=25, rescore=10, k=10, qlabels=keys)
        oi, od, ost = ti.oracle.search_batch(q, L=25, rescore=10, k=10, qlabels=keys)
        assert (gi == oi).all(), regime
        assert (gd.view(np.uint32) == od.view(np.uint32))[gi != 0xFFFFFFFF].all()
        for key in ("visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "node_reads"):
            assert gst[key] == ost[key], (regime, key)
        for k_ in regime:
            monkeypatch.delenv(k_)
    assert (gi[3] == 0xFFFFFFFF).all() and (gi[8] == 0xFFFFFFFF).all()  # labels nobody carries
    assert ix._L.vs_index_has_neighbor_masks(ix.h) == 1  # the neighbors' masks were cached next to the neighbor rows and used
    monkeypatch.setenv("VS_F_NBRMASK", "1")
    # a change of the neighbor lists through the raw array makes the cache stale: it is rebuilt before the next labeled scan
    from pgvectorscale_amd import _lib
    ptr, stride = ix.array(_lib.ARR_NBRS)
    assert ix._L.vs_index_has_neighbor_masks(ix.h) == 0
    row = np.full((1, stride), 0xFFFFFFFF, np.uint32)  # node 0 loses its neighbors
    gpu = ix.ctx
    gpu.upload(ptr, row)
    ti.nbrs[0, :] = 0xFFFFFFFF
    ti.oracle = O.OracleIndex(codes=ti.codes, nbrs=ti.nbrs, heap_tids=ti.tids, vecs=ti.vecs, mean=ti.mean, m2=ti.m2, count=ti.count,
                              bits=ti.bits, dim_index=ti.dim_index, num_neighbors=ti.R, distance_type=ti.distance,
                              default_start=ti.start, label_off=ti.label_off, label_val=ti.label_val, label_starts=ti.label_starts)
    gi, _, _, _ = ix.search_batch(q, search_list_size=25, rescore=10, k=10, qlabels=keys)
    oi, _, _ = ti.oracle.search_batch(q, L=25, rescore=10, k=10, qlabels=keys)
    assert (gi == oi).all() and ix._L.vs_index_has_neighbor_masks(ix.h) == 1
    ix.close()
