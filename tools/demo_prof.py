#!/usr/bin/env python3
"""Where generate_demos spends its time (host profile; device time shows up in the calls that synchronise).
python tools/demo_prof.py [Level] [n_demos] [batch]"""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from babyai_amd.demos import generate_demos  # noqa: E402

level = sys.argv[1] if len(sys.argv) > 1 else "GoToLocal"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
name = "BabyAI-%s-v0" % level
generate_demos(name, 256, 1, batch=256, rollout=True)
pr = cProfile.Profile()
pr.enable()
generate_demos(name, n, 1000, batch=batch, rollout=True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(24)
# the stepwise host loop it is measured against (tools/demo_bench.py), same process, same box
sys.path.insert(0, os.path.join(ROOT, "tools"))
import demo_bench  # noqa: E402
pr = cProfile.Profile()
pr.enable()
for start in range(0, n, batch):
    demo_bench.stepwise_batch(name, 1000 + start, min(batch, n - start), "cuda:0")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(16)
