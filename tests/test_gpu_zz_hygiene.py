"""Device-memory hygiene of the handles (a GPU broker process lives for weeks): whatever an index, its scans, cursors, brokers,
views and the autotuner allocated on the device or as pinned memory is returned when they are closed.  Counted by the wave64
interpreter's allocator (tests/emu: every hipMalloc / hipHostMalloc is one tracked mapping) — on hardware there is no such counter,
so there the test only runs the sequence (it must not fail or leave the context unusable)."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import TestIndex

pytestmark = pytest.mark.gpu
EMU = bool(os.environ.get("VS_EMU"))


def _live():
    if not EMU:
        return 0
    from conftest import EMU_LIB
    lib = C.CDLL(EMU_LIB)
    lib.vs_emu_live_allocations.restype = C.c_size_t
    return int(lib.vs_emu_live_allocations())


def _workout(ctx, ti, q, labeled):
    import pgvectorscale_amd as P
    ix = ti.upload(ctx)
    keys = [[1], [2, 3]] * (len(q) // 2) if labeled else None
    ix.search_batch(q, search_list_size=30, rescore=20, k=10, qlabels=keys)
    ix.stream_batch(q, search_list_size=5, m=150, qlabels=keys)  # long streams: heap spill + global dedup regions
    scan = ix.beginscan()
    scan.rescan(q[0], labels=keys and keys[0], search_list_size=10, rescore=5)
    for _ in range(70):
        scan.gettuple()
    scan.rescan(q[1], labels=keys and keys[1], search_list_size=4, rescore=0)
    for _ in range(20):
        scan.gettuple()
    scan.endscan()
    dq = ctx.alloc(q.nbytes)
    ctx.upload(dq, q)
    saved = {k: os.environ.get(k) for k in ("VS_F_LDS_MAX_INS", "VS_F_VR")}
    try:
        os.environ.update({"VS_F_LDS_MAX_INS": "0", "VS_F_VR": "0"})
        if not labeled:
            ix.autotune(dq, len(q), 10, 8, 10, reps=1, skip=())
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ctx.free(dq)
    view_ctx = P.Context(0)
    v = ix.view(view_ctx)
    v.search_batch(q[:4], search_list_size=20, rescore=10, k=5)
    v.close()
    view_ctx.close()
    broker = P.Broker(ix, max_batch=8, max_wait_us=100)
    broker.search(q[0], search_list_size=20, rescore=10, k=10)
    bscan = broker.beginscan()
    bscan.rescan(q[1], search_list_size=8, rescore=4)
    for _ in range(40):
        bscan.gettuple()
    bscan.endscan()
    broker.close()
    ix.close()


@pytest.mark.parametrize("labeled", [False, True])
def test_handles_return_their_device_memory(gpu_ctx, labeled):
    kw = dict(n=1200, dim_full=768, bits=2, R=32, distance=1, seed=13, kind="gauss", L_build=40)
    if labeled:
        kw.update(n_labels=4, deleted_frac=0.05)
    ti = TestIndex(**kw)
    q = ti.queries(8, seed=5, kind="gauss")
    _workout(gpu_ctx, ti, q, labeled)  # first pass: the context's own lazily created resources (event pool, staging ring)
    before = _live()
    for _ in range(2):
        _workout(gpu_ctx, ti, q, labeled)
    after = _live()
    assert after == before, f"{after - before} device / pinned allocations outlived their handles"


def test_staging_ring_round_trip(gpu_ctx):
    """host -> pinned ring -> HBM -> pinned ring -> host: chunks of 8 MiB and more are copied into / out of the pinned buffers by several
    threads (stage_copy, csrc/vs_api.hip); sizes around the chunk (32 MiB) and the split boundaries must come back bit for bit"""
    rng = np.random.default_rng(3)
    # (sizes whose per-thread share is page aligned while the total is not: the split must still cover the last bytes)
    for nbytes in (5, (8 << 20) - 1, (8 << 20) + 4097, (32 << 20) + 12345, (70 << 20) + 1, 2 * 4096 * 1100 + 1, 3 * 4096 * 700 + 1,
                   4 * 4096 * 600 + 3, 5 * 4096 * 500 + 4, 16 * 4096 * 140 + 15):
        a = rng.integers(0, 256, nbytes, dtype=np.uint8)
        d = gpu_ctx.alloc(nbytes)
        gpu_ctx.upload(d, a)
        b = np.zeros(nbytes, np.uint8)
        gpu_ctx.download(d, b)
        gpu_ctx.free(d)
        assert (a == b).all(), nbytes


def test_row_wise_staging_pads_and_copies_every_row(gpu_ctx):
    """index upload through the public entry point: neighbor lists of R = 50 go into device rows of 64 through the pinned ring row by
    row, zero padded, and 9-dimensional vectors into rows of 12 floats; a chunk of 8 MiB and more is filled by several threads —
    every row must arrive (vs_index_download undoes the padding)"""
    import pgvectorscale_amd as P
    rng = np.random.default_rng(11)
    n, R, dim, W = 200_001, 50, 9, 1
    nbrs = ((np.arange(n, dtype=np.uint64)[:, None] + 1 + np.arange(R, dtype=np.uint64)[None, :] * 7919) % n).astype(np.uint32)
    nbrs[::17, 40:] = 0xFFFFFFFF  # some short lists (the list ends at the first invalid id)
    codes = rng.integers(0, 1 << 18, (n, W), dtype=np.uint64)
    tids = ((np.arange(n, dtype=np.uint64) + 3) << np.uint64(16)) | np.uint64(1)
    vecs = rng.standard_normal((n, dim)).astype(np.float32)
    ix = P.DiskAnnIndex.upload(gpu_ctx, codes=codes, nbrs=nbrs, heap_tids=tids, vecs=vecs, mean=np.zeros(dim, np.float32),
                               m2=np.ones(dim, np.float32), count=n, bits=2, dim_index=dim, num_neighbors=R, distance_type=P.VS_L2,
                               default_start=0)
    try:
        host = ix.download(vecs=True)
        assert (host["nbrs"] == nbrs).all()
        assert (host["codes"] == codes).all() and (host["heap_tids"] == tids).all()
        assert (host["vecs"].view(np.uint32) == vecs.view(np.uint32)).all()
    finally:
        ix.close()
