"""Query sharding across the GPUs of one node + the final top-k gather.

The path shards by query: scans are independent and read-only, every rank holds the whole index, rank g takes a
contiguous block of the query batch, and ONE all_gather of the [nq_local, k] id / distance blocks closes the step
(RCCL over xGMI when the tensors live on GPUs — backend "nccl" — and gloo on CPU tensors in the tests).

This module is the torch.distributed convenience for callers that already live in torch.  The product path, and `bench.py --gpus N`
since round 4, is behind the C ABI instead: `vs_comm_gather_topk` / `vs_multi_search_batch` (include/vsgpu.h, pgvectorscale_amd/multi.py),
which need no torch; `shard_range` here and `vs_shard_range` there are the same arithmetic (tests/test_gpu_multi.py holds them together)."""
import torch
import torch.distributed as dist


def shard_range(nq_total: int, world: int, rank: int):
    """Contiguous block [begin, end) of rank `rank`; blocks differ by at most one query."""
    base, rem = divmod(nq_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_topk(ids: torch.Tensor, dists: torch.Tensor, group=None, counts=None, nq_total=None, verify=False):
    """all_gather of per-rank [nq_local, k] blocks -> ([nq_total, k] ids, dists) in rank order, as ONE collective and without
    a host synchronisation: ids and the bit patterns of the f32 distances travel in one int32 block of 2k columns
    (`all_gather_into_tensor`), and the shard sizes are arithmetic, not exchanged.

    Shard sizes: `counts` lists the rows of every rank; or `nq_total` = the size of the whole batch, sharded by `shard_range`;
    with neither, EVERY rank must hold the same number of rows (the benchmark: every rank runs the same number of scans) —
    that cannot be checked locally, so a caller whose shards may be uneven must pass one of the two, or `verify=True`, which
    exchanges the row counts first (one extra small collective + a host sync) and raises on every rank if they contradict
    what was passed.  Uneven shards are padded to the largest block for the collective and trimmed afterwards.

    ids: int32 (the u32 node ids bit for bit; 0xFFFFFFFF = "no row" travels as -1) or int64 holding u32 VALUES (0 ...
    0xFFFFFFFF, returned as such; negative sentinels are not representable — use 0xFFFFFFFF).  dists: float32."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ids, dists
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n, k = ids.shape
    if dists.dtype != torch.float32 or tuple(dists.shape) != (n, k):
        raise TypeError(f"gather_topk: dists must be float32 [{n}, {k}], got {dists.dtype} {tuple(dists.shape)}")
    if ids.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"gather_topk: ids must be int32 or int64, got {ids.dtype}")
    if counts is None and nq_total is not None:
        counts = [shard_range(int(nq_total), world, r)[1] - shard_range(int(nq_total), world, r)[0] for r in range(world)]
    counts = [n] * world if counts is None else [int(c) for c in counts]
    if len(counts) != world or counts[rank] != n:
        raise ValueError(f"gather_topk: rank {rank} holds {n} rows, counts say {counts}")
    if verify:
        mine = torch.tensor([n], dtype=torch.int64, device=ids.device)
        seen = torch.empty(world, dtype=torch.int64, device=ids.device)
        dist.all_gather_into_tensor(seen, mine, group=group)
        if seen.tolist() != counts:
            raise ValueError(f"gather_topk: ranks hold {seen.tolist()} rows, the caller assumed {counts}")
    m = max(counts)
    packed = torch.zeros((m, 2 * k), dtype=torch.int32, device=ids.device)
    packed[:n, :k] = ids if ids.dtype == torch.int32 else ids.to(torch.int32)  # (u32 node ids: 0xFFFFFFFF travels as -1)
    packed[:n, k:] = dists.contiguous().view(torch.int32)
    out = torch.empty((world * m, 2 * k), dtype=torch.int32, device=ids.device)
    dist.all_gather_into_tensor(out, packed, group=group)
    out = out.view(world, m, 2 * k)
    if min(counts) != m:
        out = torch.cat([out[r, :c] for r, c in enumerate(counts)], 0)
    else:
        out = out.reshape(world * m, 2 * k)
    gi = out[:, :k].contiguous()
    if ids.dtype == torch.int64:
        gi = gi.to(torch.int64) & 0xFFFFFFFF
    return gi, out[:, k:].contiguous().view(torch.float32)
