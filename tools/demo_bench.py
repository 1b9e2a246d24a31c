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

from babyai_amd.demos import generate_demos  # noqa: E402
from babyai_amd.engine import BatchedBabyAIEnv  # noqa: E402


def stepwise_batch(env_name, seed, n, device):
    """Round 2's _generate_batch (one host round trip per step), kept as the baseline of this measurement."""
    env = BatchedBabyAIEnv(env_name, n, device=device, seeds=[seed + k for k in range(n)], auto_reset=True)
    obs = env.reset()
    missions = list(obs["mission"])
    hist_img, hist_dir, hist_act = [], [], []
    ep_start = np.zeros(n, dtype=np.int64)
    span = np.full((n, 2), -1, dtype=np.int64)
    open_ = np.ones(n, dtype=bool)
    reset_cmd = torch.full((n,), env.RESET_ENV, dtype=torch.uint8, device=env.device)
    for t in range(64 * env.max_steps_bound):
        if not open_.any():
            break
        hist_img.append(obs["image"].cpu().numpy())
        hist_dir.append(obs["direction"].cpu().numpy())
        act = env.bot_actions(None)
        crashed = act == env.BOT_GAVE_UP
        act = torch.where(crashed, reset_cmd, act)
        obs, reward, done, _ = env.step(act)
        hist_act.append(act.cpu().numpy())
        crashed_h = crashed.cpu().numpy()
        reward_h, done_h = reward.cpu().numpy(), done.cpu().numpy().astype(bool)
        solved = open_ & done_h & ~crashed_h & (reward_h > 0)
        span[solved, 0], span[solved, 1] = ep_start[solved], t
        open_ &= ~solved
        again = open_ & done_h
        if again.any():
            fresh = obs["mission"]
            for i in np.nonzero(again)[0]:
                missions[i] = fresh[i]
        ep_start[done_h] = t + 1
    env.close()
    img, dirs, acts = np.stack(hist_img), np.stack(hist_dir), np.stack(hist_act)
    out = []
    for i in range(n):
        lo, hi = span[i, 0], span[i, 1] + 1
        out.append((missions[i], np.ascontiguousarray(img[lo:hi, i]), [int(v) for v in dirs[lo:hi, i]], [int(v) for v in acts[lo:hi, i]]))
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
    generate_demos(name, 256, 1, batch=256)                      # warm-up: library, allocator, first launches
    res = {"level": level, "demos": n, "batch": batch}
    t0 = time.perf_counter()
    new = generate_demos(name, n, 1000, batch=batch)
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
