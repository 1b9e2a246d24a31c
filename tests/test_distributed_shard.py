"""N>1 host logic on CPU: world_size-2 gloo processes run THE functions bench.py runs (babyai_amd/shard.py: shard
ranges and seeds, the counter-based action stream keyed on the global env index, the barrier-bracketed timed blocks with
max-over-ranks, the per-env output digests, the optional observation gather) over oracle envs, and the concatenated
per-rank results must equal the unsharded run.  (The step path itself has no collective.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from babyai_amd import shard
from babyai_amd.action_stream import actions_numpy, actions_torch, action_scalar


def test_shard_range_partitions():
    for total in (1, 7, 64, 1048576, 1048577):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert spans[-1][0] + spans[-1][1] == total
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard.shard_range(8, 2, 2)


def test_action_stream_is_one_function_of_seed_step_and_global_index():
    a = actions_numpy(1234, 5, 1000, 4096)
    assert np.array_equal(a, actions_torch(1234, 3, 8, 1000, 4096, "cpu")[2].numpy())
    assert a.tolist() == [action_scalar(1234, 5, 1000 + i) for i in range(4096)]
    # a shard sees the same actions whatever the number of ranks
    assert np.array_equal(a[1024:2048], actions_numpy(1234, 5, 2024, 1024))
    counts = np.bincount(actions_numpy(7, 0, 0, 700000), minlength=7)
    assert counts.min() > 98500 and counts.max() < 101500 and len(counts) == 7


def test_scattered_ids_cover_both_ends_and_the_boundaries():
    for n in (1, 5, 64, 300, 1024, 65536, 1048576):
        for want in (1, 16, 1024):
            ids = shard.scattered_ids(n, want)
            assert ids == sorted(set(ids)) and len(ids) == min(n, want) and all(0 <= i < n for i in ids)
            assert ids == shard.scattered_ids(n, want)                      # deterministic
    ids = shard.scattered_ids(1048576, 1024)
    assert {0, 1, 63, 64, 255, 256, 257, 524288, 1048576 - 256, 1048576 - 64, 1048575} <= set(ids)
    quart = np.bincount(np.asarray(ids) * 4 // 1048576, minlength=4)
    assert quart.min() > 200                                              # a spread, not a prefix
    assert shard.scattered_ids(131072, 128, salt=131072) != shard.scattered_ids(131072, 128, salt=0)


LEVEL, TOTAL, W, K, BLOCKS, SEED, ASEED = "GoToObjS4", 24, 3, 8, 2, 700, 99


def _rollout(ranks, first, count):
    """What bench.py does with a shard, on the oracle-backed tensor env."""
    from rollout_util import OracleTensorEnv
    seeds = np.arange(first, first + count, dtype=np.uint64) + np.uint64(SEED)
    env = OracleTensorEnv(LEVEL, seeds)
    actions = actions_torch(ASEED, 0, W + K * BLOCKS, first, count, "cpu")
    env.reset()
    dig = shard.EnvDigest(count, "cpu", 147)
    blocks = shard.timed_blocks(env, actions, W, K, BLOCKS, ranks,
                                lambda t: dig.update(env.image, env.direction, env.reward64, env.done))
    return env, dig, blocks


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    ranks = shard.Ranks.from_env("gloo")
    assert (ranks.rank, ranks.world) == (rank, world)
    first, count = shard.shard_range(TOTAL, world, rank)
    assert np.array_equal(shard.shard_seeds(SEED, TOTAL, world, rank), np.arange(first, first + count, dtype=np.uint64) + np.uint64(SEED))
    env, dig, blocks = _rollout(ranks, first, count)
    group = ranks.describe()                  # what bench.py records as `rccl`
    full_img = shard.gather_to_rank0(env.image, ranks.dist)
    full_dig = shard.gather_to_rank0(dig.h, ranks.dist, via_all_gather=True)
    local = []
    shard.timed_blocks(env, actions_torch(ASEED, 100, 104, first, count, "cpu"), 0, 2, 2, ranks, local_out=local,
                       before_block=lambda i: local.append(("before", i)))
    # the clock of a block stops when the rank's own device is idle: a rank that dawdles INSIDE the closing barrier must not
    # lengthen the block (its delay shows up in barrier_out); a rank whose STEPS are slow must (max over ranks)
    import time

    class SlowBarrier(object):
        def __init__(self, inner):
            self.inner = inner
            self.calls = 0

        def __getattr__(self, k):
            return getattr(self.inner, k)

        def barrier(self):
            self.calls += 1
            if self.inner.rank == 1 and self.calls % 2 == 0:        # the CLOSING barrier of every block
                time.sleep(0.4)
            self.inner.barrier()

    class SlowSteps(object):
        def step(self, a):
            if rank == 1:
                time.sleep(0.05)

    bar, own = [], []
    quick = shard.timed_blocks(SlowSteps(), actions_torch(ASEED, 0, 4, first, count, "cpu"), 0, 2, 2, SlowBarrier(ranks), barrier_out=bar, local_out=own)
    t = ranks.max(0.5 + rank)
    n = ranks.sum(count)
    if rank == 0:
        q.put((full_img.numpy(), full_dig.numpy(), blocks, t, n, group, local, (quick, bar, own)))
    ranks.barrier()
    ranks.close()


def test_two_rank_gloo_shards_equal_the_unsharded_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    full_img, full_dig, blocks, t, n, group, local, (quick, bar, own) = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert group["world"] == 2 and group["backend"] == "gloo" and group["allreduce_of_ones"] == 2
    assert [r["rank"] for r in group["ranks"]] == [0, 1] and len(set(r["pid"] for r in group["ranks"])) == 2
    assert local[0] == ("before", 0) and local[2] == ("before", 1) and local[1] > 0 and local[3] > 0      # hook order, own times
    env, dig, _ = _rollout(shard.Ranks(), 0, TOTAL)
    assert np.array_equal(full_img, env.image.numpy())          # rank-ordered concatenation == unsharded result
    assert np.array_equal(full_dig, dig.h.numpy())              # ... at every step, every output (running digests)
    assert len(blocks) == BLOCKS and all(b > 0 for b in blocks)
    assert t == 1.5                                             # max over ranks
    # rank 1 steps 2 x 50 ms per block and then sleeps 400 ms in the closing barrier; rank 0 does nothing
    assert all(0.09 < b < 0.3 for b in quick), quick            # the block = the slowest rank's STEPS, not its barrier nap
    assert all(b > 0.35 for b in bar), bar                      # ... which is reported on its own
    assert all(o < 0.05 for o in own), own                      # rank 0's own clock
    assert n == TOTAL
