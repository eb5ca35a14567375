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


def gather_topk(ids: torch.Tensor, dists: torch.Tensor, group=None):
    """all_gather of per-rank [nq_local, k] blocks -> ([nq_total, k] ids, dists) in rank order.
    Uneven shards are padded to the largest block for the collective and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ids, dists
    world = dist.get_world_size(group)
    n_local = torch.tensor([ids.shape[0]], dtype=torch.int64, device=ids.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    k = ids.shape[1]

    def pad(t, fill):
        if t.shape[0] == m:
            return t.contiguous()
        p = torch.full((m, k), fill, dtype=t.dtype, device=t.device)
        p[: t.shape[0]] = t
        return p

    gi = [torch.empty((m, k), dtype=ids.dtype, device=ids.device) for _ in range(world)]
    gd = [torch.empty((m, k), dtype=dists.dtype, device=dists.device) for _ in range(world)]
    dist.all_gather(gi, pad(ids, -1), group=group)
    dist.all_gather(gd, pad(dists, float("nan")), group=group)
    return (torch.cat([g[:c] for g, c in zip(gi, counts)], 0), torch.cat([g[:c] for g, c in zip(gd, counts)], 0))
