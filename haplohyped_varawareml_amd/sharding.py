"""Per-chromosome sharding across the GPUs of one node: shards are independent
(the reference already treats (donor, chromosome) pairs as independent jobs,
/root/reference/src/haplohyped/vcf_to_h5.py:142-152,191-192), so there is NO data-path collective;
torch.distributed is used only for the start/stop barrier and for reducing the timing."""
import os


def lpt_assign(sizes, n_ranks):
    """longest-processing-time-first bin packing: -> list (per rank) of sorted shard indices"""
    loads = [0] * n_ranks
    out = [[] for _ in range(n_ranks)]
    for i in sorted(range(len(sizes)), key=lambda i: (-sizes[i], i)):
        r = loads.index(min(loads))
        out[r].append(i)
        loads[r] += sizes[i]
    return [sorted(o) for o in out]


def plan(sizes, rank, world, scaling="weak"):
    """-> (shard indices for this rank, seed offset).  weak: every rank owns a full cohort of its own
    (per-GPU work fixed); strong: the shards of ONE cohort are split over the ranks."""
    if scaling == "strong":
        return lpt_assign(sizes, world)[rank], 0
    return list(range(len(sizes))), 100_000 * rank


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def reduce_job(dist, seconds, units, device=None):
    """max over ranks of the elapsed time, sum over ranks of the units processed"""
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())
