#!/usr/bin/env python3
"""Level-generator throughput through the engine's own entry point: bbai_seed fills every env's look-ahead ring with the
first D levels of its stream (k_seed + ONE k_pregen launch over all envs), so seed() time = n * D levels of generation.
    BBAI_PREGEN_GROUP=16|32|64 python tools/gen_rate.py [level envs lookahead]...
prints one JSON line per (level, envs): levels/s, ns per level chip-wide, seconds."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from babyai_amd.engine import BatchedBabyAIEnv

jobs = [("BossLevel", 262144, 4), ("GoTo", 131072, 8), ("PickupLoc", 262144, 8), ("GoToLocal", 65536, 16)]
if len(sys.argv) > 3:
    a = sys.argv[1:]
    jobs = [(a[i], int(a[i + 1]), int(a[i + 2])) for i in range(0, len(a), 3)]
for level, n, b in jobs:
    os.environ["BBAI_LOOKAHEAD"] = str(b)
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device="cuda:0")
    seeds = np.arange(n, dtype=np.uint64)
    env.seed(seeds)                      # warm: code objects loaded, clocks up
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.seed(seeds + np.uint64(1000003 * (rep + 1)))
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    levels = n * 2 * b
    print(json.dumps({"level": level, "envs": n, "levels": levels, "pregen_group": int(os.environ.get("BBAI_PREGEN_GROUP", "32")), "pregen_lane": env.get_option("pregen_lane"),
                      "seconds": best, "levels_per_s": levels / best, "ns_per_level": best / levels * 1e9,
                      "generator_failures": env.generator_failures()}), flush=True)
    env.close()
