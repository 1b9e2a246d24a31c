"""N>1 host logic on CPU: world_size-2 gloo processes exercise the shard decomposition, the timing
max-reduce and the optional observation gather (the step path itself has no collective)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from babyai_amd import shard


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


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = shard.shard_seeds(1000, total, world, rank)
    # a stand-in "observation" that is a pure function of the seed: shard-count independent
    obs = torch.from_numpy((seeds[:, None] * np.arange(1, 5, dtype=np.uint64)[None, :] % 251).astype(np.uint8))
    full = shard.gather_to_rank0(obs, dist)
    t = shard.max_over_ranks(0.5 + rank, dist)
    n = shard.sum_over_ranks(len(seeds), dist)
    if rank == 0:
        q.put((full.numpy(), t, n))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shards():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    total, world = 64, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, t, n = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seeds = np.arange(total, dtype=np.uint64) + np.uint64(1000)
    expect = (seeds[:, None] * np.arange(1, 5, dtype=np.uint64)[None, :] % 251).astype(np.uint8)
    assert np.array_equal(full, expect)       # rank-ordered concatenation == unsharded result
    assert t == 1.5                           # max over ranks
    assert n == total
