"""N>1 path on CPU: world_size-2 gloo processes shard a query batch, each computes its block (the oracle stands in for
the GPU here — this test is about the rank/shard arithmetic and the gather), and the gathered result must equal the
single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq_total, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import cached_index
    from pgvectorscale_amd.sharding import gather_topk, shard_range
    ti = cached_index(n=800, dim_full=32, bits=2, R=16, distance=1, seed=2, kind="uniform", L_build=50)
    q = ti.queries(nq_total, seed=55)
    b, e = shard_range(nq_total, world, rank)
    ids, d, _ = ti.oracle.search_batch(q[b:e], L=30, rescore=10, k=7)
    counts = [shard_range(nq_total, world, r)[1] - shard_range(nq_total, world, r)[0] for r in range(world)]
    gi, gd = gather_topk(torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(d), counts=counts, verify=True)
    gi2, gd2 = gather_topk(torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(d), nq_total=nq_total)  # sizes by arithmetic
    assert torch.equal(gi, gi2) and torch.equal(gd.view(torch.int32), gd2.view(torch.int32))
    if nq_total % world:  # uneven shards passed off as equal ones: caught on every rank before the data collective
        try:
            gather_topk(torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(d), verify=True)
            raise AssertionError("uneven shards went unnoticed")
        except ValueError as e:
            assert "ranks hold" in str(e)
    try:
        gather_topk(torch.from_numpy(ids.astype(np.int64)), torch.from_numpy(d.astype(np.float64)), counts=counts)
        raise AssertionError("float64 distances accepted")
    except TypeError:
        pass
    if rank == 0:
        np.savez(out_path, ids=gi.numpy(), dist=gd.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nq_total", [10, 11])  # even and uneven shards
def test_query_sharding_and_topk_gather_gloo(tmp_path, oracle, nq_total):
    from helpers import cached_index
    from pgvectorscale_amd.sharding import shard_range
    assert [shard_range(11, 2, r) for r in range(2)] == [(0, 6), (6, 11)]
    assert [shard_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    ti = cached_index(n=800, dim_full=32, bits=2, R=16, distance=1, seed=2, kind="uniform", L_build=50)
    q = ti.queries(nq_total, seed=55)
    want_ids, want_d, _ = ti.oracle.search_batch(q, L=30, rescore=10, k=7)
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, _free_port(), nq_total, out), nprocs=2, join=True)
    got = np.load(out)
    assert (got["ids"] == want_ids.astype(np.int64)).all()
    assert got["dist"].tobytes() == want_d.tobytes()
