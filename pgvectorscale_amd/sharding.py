"""Query sharding across the GPUs of one node + the final top-k gather.

The path shards by query: scans are independent and read-only, every rank holds the whole index, rank g takes a
contiguous block of the query batch, and ONE all_gather of the [nq_local, k] id / distance blocks closes the step
(RCCL over xGMI when the tensors live on GPUs — backend "nccl" — and gloo on CPU tensors in the tests)."""
import torch
import torch.distributed as dist


def shard_range(nq_total: int, world: int, rank: int):
    """Contiguous block [begin, end) of rank `rank`; blocks differ by at most one query."""
    base, rem = divmod(nq_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_topk(ids: torch.Tensor, dists: torch.Tensor, group=None, counts=None):
    """all_gather of per-rank [nq_local, k] blocks -> ([nq_total, k] ids, dists) in rank order, as ONE collective and without
    a host synchronisation: ids and the bit patterns of the f32 distances travel in one int32 block of 2k columns
    (`all_gather_into_tensor`), and the shard sizes are arithmetic, not exchanged — `counts` lists the rows of every rank
    (`shard_range`); None means equal shards (the benchmark: every rank runs the same number of scans).  Uneven shards are
    padded to the largest block for the collective and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ids, dists
    world = dist.get_world_size(group)
    n, k = ids.shape
    counts = [n] * world if counts is None else [int(c) for c in counts]
    assert len(counts) == world and counts[dist.get_rank(group)] == n
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
    if ids.dtype != torch.int32:
        gi = gi.to(ids.dtype)
        if ids.dtype == torch.int64:
            gi = gi & 0xFFFFFFFF
    return gi, out[:, k:].contiguous().view(torch.float32)
