"""The process-per-device half of the multi-GPU path (vs_comm_*, include/vsgpu.h) on CPU: two PROCESSES, each with its own
context on the wave64 interpreter build of the kernel sources (tests/emu/libvsgpu_emu.so), joined by a stand-in for librccl
(tests/emu/libfakerccl.so, named through VS_RCCL_LIB) — the library's own dlopen / communicator / collective code runs unmodified.
Rank 0 holds the index; rank 1 allocates the geometry and receives it (vs_comm_replicate_index); both search their shard of one
batch on their "device"; vs_comm_gather_topk must leave the whole batch's oracle rows on BOTH ranks.  Even and uneven shards."""
import multiprocessing as mp
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")


def _worker(rank, world, conn, nq_total, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["VS_RCCL_LIB"] = os.path.join(EMU_DIR, "libfakerccl.so")
    os.environ["VS_NO_TORCH"] = "1"
    from pgvectorscale_amd import _lib
    _lib.LIB_PATH = os.path.join(EMU_DIR, "libvsgpu_emu.so")
    import pgvectorscale_amd as P
    from helpers import cached_index
    from oracle import oracle_py as O
    from pgvectorscale_amd import multi as M
    ti = cached_index(n=700, dim_full=32, bits=2, R=16, distance=O.COSINE, seed=3, kind="uniform", n_labels=4, L_build=40)
    ctx = P.Context(0)
    if rank == 0:
        uid = M.comm_unique_id()
        conn.send(uid)  # (the host's own channel: a pipe here, shared memory under PostgreSQL)
    else:
        uid = conn.recv()
    comm = M.Comm(ctx, uid, rank, world)
    if rank == 0:
        ix = ti.upload(ctx)
    else:  # the geometry only; arrays, quantizer, labels, start map arrive over the communicator
        ix = P.DiskAnnIndex.alloc(ctx, n=ti.n, dim_full=ti.dim_full, bits=ti.bits, num_neighbors=ti.R, distance_type=ti.distance)
    comm.replicate_index(ix, 0)
    q = ti.queries(nq_total, seed=31)
    rng = np.random.default_rng(5)
    keys = [sorted(set(int(x) for x in rng.integers(1, 5, 2))) for _ in range(nq_total)]
    b, e = M.shard_range(nq_total, world, rank)
    k = 5
    res = {}
    for name, qlabels in (("plain", None), ("keys", keys)):
        gi, gt, gd, st = ix.search_batch(q[b:e], search_list_size=15, rescore=8, k=k, qlabels=None if qlabels is None else qlabels[b:e])
        # the device-resident blocks of this rank -> the whole batch on every rank
        d_i, d_d = ctx.alloc(max(gi.nbytes, 16)), ctx.alloc(max(gd.nbytes, 16))
        o_i, o_d = ctx.alloc(nq_total * k * 4), ctx.alloc(nq_total * k * 4)
        if e > b:
            ctx.upload(d_i, gi)
            ctx.upload(d_d, gd)
        comm.gather_topk(d_i, d_d, e - b, nq_total, k, o_i, o_d)
        ctx.sync()
        res[name + "_ids"] = ctx.download(o_i, np.empty((nq_total, k), np.uint32))
        res[name + "_dist"] = ctx.download(o_d, np.empty((nq_total, k), np.float32))
        for p in (d_i, d_d, o_i, o_d):
            ctx.free(p)
    np.savez(out_path % rank, **res)
    comm.close()
    ix.close()
    ctx.close()


@pytest.fixture(scope="module")
def emu_libs():
    r = subprocess.run(["make", "-C", EMU_DIR, "-j8", "-s"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("nq_total", [12, 7, 1])  # even shards (ncclAllGather), uneven (grouped broadcasts), an empty shard
def test_two_ranks_replicate_search_and_gather(tmp_path, oracle, emu_libs, nq_total):
    from helpers import cached_index
    from oracle import oracle_py as O
    ti = cached_index(n=700, dim_full=32, bits=2, R=16, distance=O.COSINE, seed=3, kind="uniform", n_labels=4, L_build=40)
    q = ti.queries(nq_total, seed=31)
    rng = np.random.default_rng(5)
    keys = [sorted(set(int(x) for x in rng.integers(1, 5, 2))) for _ in range(nq_total)]
    out = str(tmp_path / "rank%d.npz")
    mpc = mp.get_context("spawn")
    a, b = mpc.Pipe()
    procs = [mpc.Process(target=_worker, args=(r, 2, (a, b)[r], nq_total, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for name, qlabels in (("plain", None), ("keys", keys)):
        want_i, want_d, _ = ti.oracle.search_batch(q, L=15, rescore=8, k=5, qlabels=qlabels)
        for r in range(2):
            got = np.load(out % r)
            assert (got[name + "_ids"] == want_i).all(), (name, r)
            assert np.allclose(got[name + "_dist"], want_d, rtol=1e-5, atol=0, equal_nan=True), (name, r)


def _worker_mismatch(rank, world, conn, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["VS_RCCL_LIB"] = os.path.join(EMU_DIR, "libfakerccl.so")
    os.environ["VS_NO_TORCH"] = "1"
    from pgvectorscale_amd import _lib
    _lib.LIB_PATH = os.path.join(EMU_DIR, "libvsgpu_emu.so")
    import pgvectorscale_amd as P
    from helpers import cached_index
    from oracle import oracle_py as O
    from pgvectorscale_amd import multi as M
    ti = cached_index(n=700, dim_full=32, bits=2, R=16, distance=O.COSINE, seed=3, kind="uniform", n_labels=4, L_build=40)
    ctx = P.Context(0)
    if rank == 0:
        uid = M.comm_unique_id()
        conn.send(uid)
    else:
        uid = conn.recv()
    comm = M.Comm(ctx, uid, rank, world)
    if rank == 0:
        ix = ti.upload(ctx)
    else:  # ANOTHER geometry than the root's
        ix = P.DiskAnnIndex.alloc(ctx, n=ti.n + 1, dim_full=ti.dim_full, bits=ti.bits, num_neighbors=ti.R, distance_type=ti.distance)
    msg = "no error"
    try:
        comm.replicate_index(ix, 0)
    except P.VsError as e:
        msg = str(e)
    open(out_path % rank, "w").write(msg)
    comm.close()
    ix.close()
    ctx.close()


def test_replicate_fails_on_every_rank_together(tmp_path, oracle, emu_libs):
    """a rank that cannot take the root's index (another geometry here) fails the call on ALL ranks before the array broadcasts start:
    nobody is left waiting inside a collective (round-4 advisor finding)"""
    out = str(tmp_path / "rank%d.txt")
    mpc = mp.get_context("spawn")
    a, b = mpc.Pipe()
    procs = [mpc.Process(target=_worker_mismatch, args=(r, 2, (a, b)[r], out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, "a rank hung or died"
    m0, m1 = open(out % 0).read(), open(out % 1).read()
    assert "rank 1 could not take the root's index" in m0, m0
    assert "another geometry" in m1, m1
