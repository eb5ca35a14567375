"""The multi-GPU entry points of the C ABI (vs_index_replicate, vs_multi_*, vs_comm_*) on ONE device: two contexts
of GPU 0 stand in for two GPUs — the replica is made with the same device-to-device copy a second GPU would receive over xGMI,
the shards run on their own contexts / streams / host threads, and the rows of the whole batch must be the oracle's.  The
process-per-device half (RCCL) runs with a world of one here (the real librccl on hardware) and with two processes over a
stand-in RCCL in tests/test_multi_comm.py (CPU tier)."""
import ctypes as C
import os

import numpy as np
import pytest

import pgvectorscale_amd as P
from helpers import cached_index
from oracle import oracle_py as O
from pgvectorscale_amd import multi as M

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _index():
    return cached_index(n=1500, dim_full=64, bits=2, R=24, distance=O.L2, seed=4, kind="uniform", n_labels=6, deleted_frac=0.05, L_build=40)


def _keys(nq, seed):
    rng = np.random.default_rng(seed)
    return [sorted(set(int(x) for x in rng.integers(1, 7, int(rng.integers(1, 3))))) for _ in range(nq)]


def test_replica_on_a_second_context_outlives_its_source(gpu_ctx, oracle):
    ti = _index()
    src = ti.upload(gpu_ctx)
    vis = (np.random.default_rng(8).random(ti.n) > 0.1).astype(np.uint8)
    src.set_visibility(vis)
    ctx2 = P.Context(0)
    rep = M.replicate(src, ctx2)
    src.close()  # the replica owns its arrays: labels, start map, quantizer, visibility mask and all
    try:
        q = ti.queries(40, seed=21)
        keys = _keys(40, 3)
        ti.oracle.set_visibility(vis)
        for qlabels in (None, keys):
            gi, gt, gd, st = rep.search_batch(q, search_list_size=20, rescore=15, k=8, qlabels=qlabels)
            oi, od, ost = ti.oracle.search_batch(q, L=20, rescore=15, k=8, qlabels=qlabels)
            assert (gi == oi).all()
            assert np.allclose(gd, od, rtol=1e-5, atol=0, equal_nan=True)
            assert st["visited_nodes"] == ost["visited_nodes"] and st["quantized_distance_comparisons"] == ost["quantized_distance_comparisons"]
    finally:
        ti.oracle.set_visibility(None)
        rep.close()
        ctx2.close()


@pytest.mark.parametrize("copy_always", [False, True], ids=["view_plus_replica", "two_replicas"])
@pytest.mark.parametrize("nq", [33, 2, 1])  # uneven shards; fewer queries than devices leaves a shard empty
def test_multi_search_batch_returns_the_single_device_rows(gpu_ctx, oracle, copy_always, nq):
    ti = _index()
    src = ti.upload(gpu_ctx)
    m = M.MultiIndex(src, [0, 0, 0] if nq == 33 else [0, 0], copy_always=copy_always)
    try:
        assert len(m) == (3 if nq == 33 else 2)
        q = ti.queries(nq, seed=5)
        keys = _keys(nq, 9)
        for qlabels in (None, keys):
            gi, gt, gd, st = m.search_batch(q, search_list_size=25, rescore=12, k=9, qlabels=qlabels)
            si, stid, sd, sst = src.search_batch(q, search_list_size=25, rescore=12, k=9, qlabels=qlabels)
            oi, od, ost = ti.oracle.search_batch(q, L=25, rescore=12, k=9, qlabels=qlabels)
            assert (gi == oi).all() and (gi == si).all() and (gt == stid).all()
            assert gd.tobytes() == sd.tobytes()
            for key in ("queries", "visited_nodes", "candidate_nodes", "quantized_distance_comparisons", "full_distance_comparisons", "next_calls"):
                assert st[key] == sst[key], key
            assert st["visited_nodes"] == ost["visited_nodes"]
        ids, ham, st = m.stream_batch(q, search_list_size=10, m=17)
        sids, sham, _ = src.stream_batch(q, search_list_size=10, m=17)
        assert (ids == sids).all() and (ham == sham).all()
        # a shard's failure is the call's failure, with the device named
        with pytest.raises(P.VsError) as ei:
            m.search_batch(q, search_list_size=20000, rescore=5, k=3)
        assert "query_search_list_size" in str(ei.value)
    finally:
        m.close()
        src.close()


def test_shard_arithmetic_matches_the_python_mirror():
    from pgvectorscale_amd.sharding import shard_range
    for nq, world in ((10, 4), (11, 2), (0, 3), (5, 8), (262144, 8), (262145, 8)):
        got = [M.shard_range(nq, world, r) for r in range(world)]
        assert got == [shard_range(nq, world, r) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == nq and all(a[1] == b[0] for a, b in zip(got, got[1:]))
    with pytest.raises(P.VsError):
        M.shard_range(4, 2, 2)


def test_comm_world_of_one_gathers_and_replicates(gpu_ctx, oracle, monkeypatch):
    """RCCL itself on the device (hardware: the real librccl, found by dlopen; interpreter: the stand-in): communicator, the top-k
    gather behind a device-resident search, broadcast, index replication — with one rank every collective is a copy in place."""
    if os.environ.get("VS_EMU"):
        monkeypatch.setenv("VS_RCCL_LIB", os.path.join(ROOT, "tests", "emu", "libfakerccl.so"))
    ti = _index()
    ix = ti.upload(gpu_ctx)
    comm = M.Comm(gpu_ctx, M.comm_unique_id(), 0, 1)
    try:
        nq, k = 24, 6
        q = ti.queries(nq, seed=77)
        d_q = gpu_ctx.alloc(q.nbytes)
        gpu_ctx.upload(d_q, q)
        bufs = [gpu_ctx.alloc(nq * k * 4) for _ in range(4)]
        ix.search_batch_dev(d_q, nq, 20, 10, k, bufs[0], None, bufs[1])
        comm.gather_topk(bufs[0], bufs[1], nq, nq, k, bufs[2], bufs[3])
        ix.search_batch_dev_finish()
        gi = gpu_ctx.download(bufs[2], np.empty((nq, k), np.uint32))
        gd = gpu_ctx.download(bufs[3], np.empty((nq, k), np.float32))
        oi, od, _ = ti.oracle.search_batch(q, L=20, rescore=10, k=k)
        assert (gi == oi).all() and np.allclose(gd, od, rtol=1e-5, atol=0)
        with pytest.raises(P.VsError):  # a block that is not this rank's shard of the batch
            comm.gather_topk(bufs[0], bufs[1], nq - 1, nq, k, bufs[2], bufs[3])
        comm.bcast(bufs[0], nq * k * 4, 0)
        comm.replicate_index(ix, 0)
        gi2, _, gd2, _ = ix.search_batch(q, search_list_size=20, rescore=10, k=k)
        assert (gi2 == oi).all()
        for b in bufs + [d_q]:
            gpu_ctx.free(b)
    finally:
        comm.close()
        ix.close()


def test_multi_across_two_devices(gpu_ctx, oracle, monkeypatch):
    """devices [0, 1]: the replica on device 1 is made with hipMemcpyPeerAsync (peer access enabled where the devices can reach each
    other) and shard 1 really runs on the other device.  Needs two devices: the interpreter provides them (VS_EMU_DEVICES), a
    single-GPU box skips, a multi-GPU box runs it for real."""
    if os.environ.get("VS_EMU"):
        monkeypatch.setenv("VS_EMU_DEVICES", "2")
    try:
        ctx1 = P.Context(1)
    except P.VsError:
        pytest.skip("one device only")
    ti = _index()
    src = ti.upload(gpu_ctx)
    rep = M.replicate(src, ctx1)  # device 0 -> device 1
    m = M.MultiIndex(src, [0, 1])
    try:
        q = ti.queries(21, seed=15)
        keys = _keys(21, 4)
        oi, od, ost = ti.oracle.search_batch(q, L=25, rescore=12, k=9, qlabels=keys)
        ri, _, rd, _ = rep.search_batch(q, search_list_size=25, rescore=12, k=9, qlabels=keys)
        gi, gt, gd, st = m.search_batch(q, search_list_size=25, rescore=12, k=9, qlabels=keys)
        assert (ri == oi).all() and (gi == oi).all()
        assert np.allclose(gd, od, rtol=1e-5, atol=0, equal_nan=True) and gd.tobytes() == rd.tobytes()
        assert st["visited_nodes"] == ost["visited_nodes"]
    finally:
        m.close()
        rep.close()
        src.close()
        ctx1.close()
