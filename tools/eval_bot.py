#!/usr/bin/env python3
"""The reference's expert check (scripts/eval_bot.py:83-190, non-advise mode) on the batched engine: for every level,
`--num_runs` missions `Level(seed = --seed + run)` solved by the device expert (bbai_bot_act), one row per level in the
reference's format  `level: success %, mean reward, mean steps`  plus the number of bots that gave up where the
reference bot would have raised.  python tools/eval_bot.py [--num_runs 1024] [--seed 0] [--levels A,B]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from babyai_amd.engine import BatchedBabyAIEnv  # noqa: E402
from babyai_amd.levels import LEVELS  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--num_runs", type=int, default=1024)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--levels", default=None)
args = ap.parse_args()
names = args.levels.split(",") if args.levels else sorted(LEVELS)
n = args.num_runs
t_start = time.time()
total_steps = 0
not_all = []
for name in names:
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % name, n, device="cuda:0", seeds=args.seed, auto_reset=False)
    env.reset()
    finished = torch.zeros(n, dtype=torch.bool, device="cuda:0")
    reward = torch.zeros(n, device="cuda:0")
    steps = torch.zeros(n, dtype=torch.int64, device="cuda:0")
    crashed = torch.zeros(n, dtype=torch.bool, device="cuda:0")
    reset_cmd = torch.full((n,), env.RESET_ENV, dtype=torch.uint8, device="cuda:0")
    for t in range(env.max_steps_bound + 1):
        a = env.bot_actions(None)
        gone = (a == env.BOT_GAVE_UP) & ~finished
        crashed |= gone
        _, r, d, _ = env.step(torch.where(a == env.BOT_GAVE_UP, reset_cmd, a))
        d = d.bool()
        newly = d & ~finished
        reward = torch.where(newly, r, reward)
        steps += (~finished).long()
        finished |= d
        if bool(finished.all()):
            break
    ok = (reward > 0) & ~crashed
    num_success = int(ok.sum())
    total_steps += int(steps.sum())
    print("%28s: %.1f%%, r=%.3f, s=%.2f   (bot gave up in %d runs, %d of them on the stack limit)" % (
        name, 100.0 * num_success / n, float(reward[~crashed].sum()) / n, float(steps[ok].sum()) / n, int(crashed.sum()),
        env.bot_stats()["capacity"]), flush=True)
    if num_success != n:
        not_all.append(name)
    env.close()
print("total time: %.1fs, total episode_steps: %d, %d levels x %d runs" % (time.time() - t_start, total_steps, len(names), n))
print("levels with failures:", ", ".join(not_all) if not_all else "none")
