#!/usr/bin/env python3
"""Demonstration throughput: babyai_amd.demos.generate_demos (device rollout, bbai_bot_rollout) against the step-by-step
host loop it replaced (rounds 1-2: bbai_bot_act + bbai_step per step, six device->host reads per step), same streams, same
demos (compared).  python tools/demo_bench.py [Level] [n_demos] [batch]"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from babyai_amd.demos import _generate_batch_stepwise, generate_demos  # noqa: E402
from babyai_amd.engine import BatchedBabyAIEnv  # noqa: E402


def stepwise_batch(env_name, seed, n, device):
    out = [None] * n
    _generate_batch_stepwise(env_name, seed, n, device, 0, None, None, out, 0)
    return out


def digest(demos):
    h = hashlib.sha256()
    for m, img, d, a in demos:
        h.update(m.encode()); h.update(img.tobytes()); h.update(bytes(d)); h.update(bytes(a))
    return h.hexdigest()[:16]


def main():
    level = sys.argv[1] if len(sys.argv) > 1 else "BossLevel"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    name = "BabyAI-%s-v0" % level
    generate_demos(name, 256, 1, batch=256, rollout=True)        # warm-up: library, allocator, first launches
    res = {"level": level, "demos": n, "batch": batch}
    t0 = time.perf_counter()
    new = generate_demos(name, n, 1000, batch=batch, rollout=True)
    res["rollout_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    old = []
    for start in range(0, n, batch):
        old += stepwise_batch(name, 1000 + start, min(batch, n - start), "cuda:0")
    res["stepwise_s"] = time.perf_counter() - t0
    res["rollout_demos_per_s"] = n / res["rollout_s"]
    res["stepwise_demos_per_s"] = n / res["stepwise_s"]
    res["speedup"] = res["stepwise_s"] / res["rollout_s"]
    res["mean_length"] = float(np.mean([len(d[3]) for d in new]))
    res["same_demos"] = digest(new) == digest(old)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
