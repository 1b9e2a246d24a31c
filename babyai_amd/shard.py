"""Multi-GPU decomposition and the measured rollout loop: one process per GPU, contiguous env shards, NO collective on
the step path (envs never interact: SURVEY.md section 8e; the reference's only parallelism is one env per worker
process, babyai/rl/utils/penv.py:28-52).  Env i's trajectory is a pure function of (seed_i, its actions) with seed_i =
base + global index and actions keyed on the global index (action_stream.py), so results do not depend on the number of
shards.  `bench.py` drives exactly the functions below; tests/test_distributed_shard.py runs the same functions under
world_size-2 gloo on CPU over oracle envs, tests/test_gpu_multirank.py runs bench.py itself with 4 ranks on one GPU.

The only collectives are (a) timing: barrier + max-over-ranks, (b) sums of counters for the report, (c) the OPTIONAL
gather of encoded observations to rank 0 (RCCL over xGMI on GPUs, gloo on CPU).
"""
import os


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


def scattered_ids(count, want, salt=0):
    """`want` distinct local env indices of a shard of `count` envs, ascending: both ends, wave (64) and step-block (256)
    boundaries at the start, the middle and the end, the rest spread pseudo-randomly (a fixed LCG, so every run and every
    rank count picks the same envs for the same shard).  What the in-run parity tap of bench.py and the full-size GPU
    test check against the oracle."""
    want = min(int(want), int(count))
    if want <= 0:
        return []
    n = int(count)
    edges = [0, 1, 63, 64, 65, 255, 256, 257, n // 2 - 1, n // 2, n // 2 + 1, n - 257, n - 256, n - 255, n - 65, n - 64, n - 2, n - 1]
    picked = []
    seen = set()
    for i in edges:
        if 0 <= i < n and i not in seen and len(picked) < want:
            seen.add(i)
            picked.append(i)
    x = (0x9E3779B97F4A7C15 ^ (n * 0x100000001B3) ^ (int(salt) * 0xD1B54A32D192ED03)) & 0xFFFFFFFFFFFFFFFF
    while len(picked) < want:
        x = (x * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        i = (x >> 20) % n
        if i not in seen:
            seen.add(i)
            picked.append(i)
    return sorted(picked)


class Ranks(object):
    """The process group of a bench / rollout run (or a single process when WORLD_SIZE is 1 / unset)."""

    def __init__(self, dist=None, reduce_device=None, sync=None):
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1
        self.reduce_device = reduce_device
        self._sync = sync or (lambda: None)

    @classmethod
    def from_env(cls, backend="nccl", share_device=False):
        """torch.distributed.run contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  backend
        "nccl" (= RCCL, one rank per GPU) or "gloo" (test rigs: ranks may share cuda:0 with share_device)."""
        import torch
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = 0 if share_device else int(os.environ.get("LOCAL_RANK", "0"))
        device = torch.device("cuda", local_rank) if torch.cuda.is_available() else torch.device("cpu")
        if device.type == "cuda":
            torch.cuda.set_device(device)
        dist = None
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=device)
            else:
                dist.init_process_group(backend=backend)
        sync = (lambda: torch.cuda.synchronize(device)) if device.type == "cuda" else None
        r = cls(dist, device if (backend == "nccl" and device.type == "cuda") else torch.device("cpu"), sync)
        r.device = device
        return r

    def barrier(self):
        """Device idle on every rank: sync, barrier, sync (the bench contract's bracket)."""
        self._sync()
        if self.dist:
            self.dist.barrier()
        self._sync()

    def max(self, value):
        return max_over_ranks(value, self.dist, self.reduce_device)

    def sum(self, value):
        return sum_over_ranks(value, self.dist, self.reduce_device)

    def gather_objects(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank (one all_gather_object; [obj] in a single process)."""
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def describe(self):
        """What the LIVE process group looks like -- the answer to "did N ranks on N devices really take part?":
        world size and backend as the group reports them, every rank's device (index, name, PCI bus id where the
        runtime gives one, host pid), and the result of one all-reduce of ones on the reduce device (== world)."""
        import torch
        dev = getattr(self, "device", None)
        mine = {"rank": self.rank, "pid": os.getpid(), "device": str(dev) if dev is not None else None,
                "local_rank": int(os.environ.get("LOCAL_RANK", "0"))}
        if dev is not None and dev.type == "cuda":
            props = torch.cuda.get_device_properties(dev)
            mine["device_name"] = props.name
            mine["pci_bus_id"] = getattr(props, "pci_bus_id", None)
            mine["device_uuid"] = str(getattr(props, "uuid", "")) or None
        ranks = self.gather_objects(mine)
        ones = self.sum(1)
        return {"world": self.world, "backend": self.dist.get_backend() if self.dist else None,
                "device_ids": [r.get("device") for r in ranks],
                "distinct_devices": len(set((r.get("device"), r.get("pci_bus_id"), r.get("device_uuid")) for r in ranks)),
                "allreduce_of_ones": ones, "ranks": ranks}

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


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


def gather_to_rank0(tensor, dist, via_all_gather=False):
    """Optional obs gather: equal-sized shards -> rank 0 gets the concatenation in rank order.  `via_all_gather`: use the
    all-gather collective (every rank receives; rank 0's copy is the gather) -- the one RCCL path every installation
    exercises, for callers that must not risk a point-to-point based `gather` (bench.py)."""
    import torch
    world = dist.get_world_size()
    if via_all_gather:
        parts = [torch.empty_like(tensor) for _ in range(world)]
        dist.all_gather(parts, tensor.contiguous())
        return torch.cat(parts, dim=0) if dist.get_rank() == 0 else None
    if dist.get_rank() == 0:
        parts = [torch.empty_like(tensor) for _ in range(world)]
        dist.gather(tensor, gather_list=parts, dst=0)
        return torch.cat(parts, dim=0)
    dist.gather(tensor, gather_list=None, dst=0)
    return None


def timed_blocks(env, actions, warmup, steps, blocks, ranks, after_step=None, after_block=None, before_block=None, local_out=None,
                 barrier_out=None, run_steps=None, step_fn=None):
    """The measured loop.  `actions`: uint8[warmup + blocks*steps, E] resident on the env's device.  `warmup` untimed
    steps, then `blocks` timed blocks of EXACTLY `steps` steps.  Every block is bracketed by ranks.barrier() (device idle,
    barrier, device idle) on both sides; a rank's clock runs from the end of the opening barrier until ITS OWN device is
    idle again, and the block's time is the MAX of those over the ranks.  The closing barrier and the max-reduce happen
    after every clock has stopped: with several ranks they cost O(100 us) of collective + host synchronisation, which is
    not part of any rank's steps (a block of 20 steps of a 131 072-env shard lasts 4.5 ms).  What the closing barrier
    costs -- waiting for the slowest rank included -- goes to `barrier_out` (max over ranks, seconds per block).
    `after_step(t)` (t = index into `actions`) runs inside the timed region (parity tap, digests), `before_block(i)` /
    `after_block(i)` between blocks (untimed).  `run_steps(t0, k)`, when given, REPLACES the per-step loop and `after_step`: it
    enqueues steps t0 .. t0 + k - 1 (tap included) in one call -- the engine's open-loop rollout entry (include/bbai.h
    bbai_rollout; bench.py --rollout-entry).  `step_fn(t)`, when given, is called instead of env.step(actions[t]) (the step that logs its
    own tap rows: include/bbai.h bbai_step_tapped); `after_step` still follows it.  Returns the list of per-block seconds (max over ranks); `local_out`
    (a list) receives this rank's own per-block seconds.  The cyclic garbage collector is off for the length of the loop (as
    `timeit` does): a generation-2 pass between two launches is a multi-millisecond host stall that no kernel caused."""
    import gc
    import time
    gc_was_on = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        return _timed_blocks(env, actions, warmup, steps, blocks, ranks, after_step, after_block, before_block, local_out, barrier_out, run_steps, time, step_fn)
    finally:
        if gc_was_on:
            gc.enable()


def _timed_blocks(env, actions, warmup, steps, blocks, ranks, after_step, after_block, before_block, local_out, barrier_out, run_steps, time, step_fn=None):
    do_step = step_fn if step_fn is not None else (lambda t: env.step(actions[t]))
    t = 0
    if run_steps is not None:
        if warmup:
            run_steps(0, warmup)
        t = warmup
    else:
        for _ in range(warmup):
            do_step(t)
            if after_step:
                after_step(t)
            t += 1
    out = []
    for b in range(blocks):
        if before_block:
            before_block(b)
        ranks.barrier()
        t0 = time.perf_counter()
        if run_steps is not None:
            run_steps(t, steps)
            t += steps
        else:
            for _ in range(steps):
                do_step(t)
                if after_step:
                    after_step(t)
                t += 1
        ranks._sync()
        mine = time.perf_counter() - t0                      # this rank's own device went idle: its clock stops here
        ranks.barrier()
        closing = time.perf_counter() - t0 - mine
        if local_out is not None:
            local_out.append(mine)
        out.append(ranks.max(mine))
        if barrier_out is not None:
            barrier_out.append(ranks.max(closing))
        if after_block:
            after_block(len(out) - 1)
    return out


class EnvDigest(object):
    """Per-env running 64-bit digest of everything a step hands out (image, direction, float64 reward bits, done),
    kept on the env's device with a handful of tensor ops per step.  Per-env, so the digests of the shards of a run
    concatenate to the digests of the unsharded run -- the multi-rank equality check."""

    def __init__(self, num_envs, device, obs_bytes):
        import torch
        g = torch.Generator(device="cpu")
        g.manual_seed(20260924)
        self.w = (torch.randint(-2 ** 62, 2 ** 62, (obs_bytes,), generator=g, dtype=torch.int64) | 1).to(device)
        self.h = torch.zeros(num_envs, dtype=torch.int64, device=device)
        self.torch = torch

    def update(self, image, direction, reward64, done):
        torch = self.torch
        n = self.h.shape[0]
        x = (image.reshape(n, -1).to(torch.int64) * self.w).sum(dim=1)
        x = x + direction.to(torch.int64) * 0x5851F42D4C957F2D + done.to(torch.int64) * 0x14057B7EF767814F
        x = x ^ reward64.contiguous().view(torch.int64)
        self.h = (self.h ^ x) * (-0x61C8864680B583EB) + 0x2545F4914F6CDD1D       # odd multiplier: wraps mod 2^64

    def numpy(self):
        return self.h.cpu().numpy().view("uint64")
