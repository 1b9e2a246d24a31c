"""bench.py with WORLD_SIZE > 1 on ONE GPU (the box has one): `torch.distributed.run --nproc-per-node 4 bench.py --gpus 4
--share-device --dist-backend gloo` -- the launch line, rank plumbing, sharding, action stream, barrier-bracketed blocks,
max-reduce and JSON contract of the multi-GPU run -- and the per-env output digests of the four shards must concatenate
to those of the unsharded 16384-env run.  A second test tries the RCCL backend with two ranks on the one visible
device; RCCL may refuse that ("duplicate GPU"), in which case it is skipped with RCCL's message."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--level", "GoToLocal", "--no-pixel", "--steps", "16", "--warmup", "4", "--min-seconds", "30", "--max-blocks", "3",
          "--no-cpu-baseline", "--prewarm-seconds", "0", "--parity-envs", "64"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONDONTWRITEBYTECODE="1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def _launch(world, backend, envs, prefix, extra=()):
    return _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                 "--master-port", str(_free_port()), "bench.py", "--gpus", str(world), "--share-device", "--dist-backend", backend,
                 "--envs", str(envs), "--dump-digest", prefix] + COMMON + list(extra))


@pytest.mark.gpu
def test_bench_four_ranks_on_one_gpu_equal_the_unsharded_run(gpu, tmp_path):
    p1, one = _run([sys.executable, "bench.py", "--envs", "16384", "--dump-digest", str(tmp_path / "one")] + COMMON)
    assert p1.returncode == 0, p1.stderr[-2000:]
    p4, four = _launch(4, "gloo", 4096, str(tmp_path / "four"))
    assert p4.returncode == 0, p4.stderr[-2000:]
    assert one["n_gpus"] == 1 and four["n_gpus"] == 4 and four["scaling"] == "weak"
    assert four["config"]["total_envs"] == 16384 and four["config"]["envs_per_gpu"] == 4096
    assert four["steps"] == 16 and four["warmup"] == 4 and four["value"] > 0 and four["timing"]["blocks"] >= 1
    assert one["timing"]["blocks"] == four["timing"]["blocks"]          # same number of steps in both runs
    assert one["parity"]["mismatches"] == 0 and four["parity"]["mismatches_all_ranks"] == 0
    assert four["parity"]["envs_all_ranks"] == 4 * four["parity"]["envs"]
    whole = np.load(str(tmp_path / "one") + ".rank0.npy")
    parts = np.concatenate([np.load(str(tmp_path / "four") + ".rank%d.npy" % r) for r in range(4)])
    assert whole.shape == (16384,) and np.array_equal(whole, parts)
    assert one["config"]["resets_in_timed_region"] == four["config"]["resets_in_timed_region"] > 0
    # the optional obs gather to rank 0 ran (over gloo here) and is reported outside `value`
    assert four["obs_gather"]["ms"] > 0 and four["obs_gather"]["bytes_per_peer"] == 4096 * 147 and "obs_gather" not in one


@pytest.mark.gpu
def test_bench_two_ranks_rccl_on_one_gpu(gpu, tmp_path):
    p, two = _launch(2, "nccl", 4096, str(tmp_path / "two"))
    if p.returncode != 0:
        err = p.stderr or ""
        hits = [l for l in err.splitlines() if any(w in l for w in ("uplicate GPU", "ncclInvalidUsage", "invalid usage", "NCCL error", "RCCL",
                                                                    "ncclUnhandled", "ProcessGroupNCCL"))]
        if hits:
            pytest.skip("RCCL refuses two ranks on one device: " + hits[0].strip()[:240])
        assert False, err[-3000:]
    assert two["n_gpus"] == 2 and two["parity"]["mismatches_all_ranks"] == 0
