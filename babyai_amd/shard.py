"""Multi-GPU decomposition: one process per GPU, contiguous env shards, NO collective on the
step path (envs never interact: SURVEY.md section 8e).  Env i's trajectory is a pure function of
(seed_i, its actions), with seed_i = base + global index, so results do not depend on the
number of shards (tests/test_gpu_parity.py::test_shard_independence checks this on one GPU).

The only collectives are (a) bench timing: barrier + max-over-ranks, (b) the OPTIONAL gather of
encoded observations to rank 0 (torch.distributed.gather: RCCL over xGMI on GPUs, gloo on CPU).
"""


def shard_range(total_envs, world_size, rank):
    """Contiguous shard [first, first+count) of `total_envs` for `rank`; sizes differ by at most 1."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, rem = divmod(total_envs, world_size)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def shard_seeds(seed_base, total_envs, world_size, rank):
    """Seeds of this rank's envs: seed_base + global env index (mirrors babyai/evaluate.py:105-106)."""
    import numpy as np
    first, count = shard_range(total_envs, world_size, rank)
    return np.arange(first, first + count, dtype=np.uint64) + np.uint64(seed_base)


def max_over_ranks(value, dist=None, device=None):
    """max-reduce a python float over ranks (bench timing contract)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(value)
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t)
    return int(t.item())


def gather_to_rank0(tensor, dist):
    """Optional obs gather: equal-sized shards -> rank 0 gets the concatenation in rank order."""
    import torch
    world = dist.get_world_size()
    if dist.get_rank() == 0:
        parts = [torch.empty_like(tensor) for _ in range(world)]
        dist.gather(tensor, gather_list=parts, dst=0)
        return torch.cat(parts, dim=0)
    dist.gather(tensor, gather_list=None, dst=0)
    return None
