#!/usr/bin/env python3
"""Throughput of the batched expert: env-steps/s with bbai_bot_act choosing every action (demo-generation loop of
scripts/make_agent_demos.py:93-107 without the host).  python tools/bot_bench.py [Level] [envs] [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from babyai_amd.engine import BatchedBabyAIEnv  # noqa: E402

level = sys.argv[1] if len(sys.argv) > 1 else "BossLevel"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device="cuda:0", seeds=0)
env.reset()
rnd = torch.randint(0, 7, (steps + 20, n), dtype=torch.uint8, device="cuda:0")


def run(k0, k1):
    succ = torch.zeros((), dtype=torch.int64, device="cuda:0")
    eps = torch.zeros((), dtype=torch.int64, device="cuda:0")
    for t in range(k0, k1):
        a = env.bot_actions(None)
        a = torch.where(a == 255, rnd[t], a)          # a bot that gave up: random until the episode ends
        _, r, d, _ = env.step(a)
        succ += (r > 0).sum()
        eps += d.sum()
    torch.cuda.synchronize()
    return int(succ), int(eps)


run(0, 20)
t0 = time.perf_counter()
succ, eps = run(20, 20 + steps)
dt = time.perf_counter() - t0
print(json.dumps({"level": level, "envs": n, "steps": steps, "bot_env_steps_per_s": n * steps / dt, "ms_per_step": dt / steps * 1e3,
                  "episodes": eps, "success_rate": succ / max(eps, 1), "bot": env.bot_stats(), "bot_group": env.get_option("bot_group")}))
