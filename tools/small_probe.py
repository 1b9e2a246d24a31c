#!/usr/bin/env python3
"""Where does a step go at the small per-GPU shard sizes (BASELINE C2 / C4)?  (experiment tool)

For one (level, envs) and the knobs given in the environment (BBAI_LOOKAHEAD, BBAI_PREGEN_BLOCKS, BBAI_PREGEN_PRIORITY)
prints one JSON line: wall us/step over K auto-reset steps of random actions, and the HIP-event time of the step
group alone.  `--no-reset` steps without auto-reset (frozen envs, no k_consume / k_pregen): the k_step floor.
    python tools/small_probe.py GoToLocal 65536 [--steps 512] [--no-reset]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from babyai_amd.engine import BatchedBabyAIEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("level")
ap.add_argument("envs", type=int)
ap.add_argument("--steps", type=int, default=512)
ap.add_argument("--no-reset", action="store_true")
ap.add_argument("--tag", default="")
ap.add_argument("--pixel", action="store_true")
args = ap.parse_args()

env = BatchedBabyAIEnv("BabyAI-%s-v0" % args.level, args.envs, seeds=0, auto_reset=not args.no_reset, pixel=args.pixel)
env.reset()
K = args.steps
acts = torch.randint(0, 3 if args.no_reset else 7, (K + 32, args.envs), dtype=torch.uint8, device="cuda")
for t in range(32):
    env.step(acts[t])
torch.cuda.synchronize()
r0 = env.reset_count()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
a.record()
for t in range(32, 32 + K):
    env.step(acts[t])
b.record()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({
    "tag": args.tag, "level": args.level, "envs": args.envs, "auto_reset": not args.no_reset, "steps": K,
    "wall_us_per_step": dt / K * 1e6, "host_issue_us_per_step": t_issue / K * 1e6, "gpu_us_per_step": a.elapsed_time(b) / K * 1e3,
    "env_steps_per_s": K * args.envs / dt, "resets_per_step": (env.reset_count() - r0) / K,
    "knobs": {k: os.environ.get(k) for k in ("BBAI_LOOKAHEAD", "BBAI_PREGEN_BLOCKS", "BBAI_PREGEN_PRIORITY", "BBAI_RING_GIB")}, "pixel": args.pixel}))
env.close()
