#!/usr/bin/env python3
"""Phase breakdown of the expert kernel (experiment): needs an engine built with -DBBAI_BOT_PROF
(tools/bot_prof.sh builds it next to the product library and points BBAI_ENGINE_LIB at it).
    python tools/bot_prof.py BossLevel 262144 40"""
import ctypes
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from babyai_amd.engine import BatchedBabyAIEnv, load_library  # noqa: E402
level, n, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = load_library()
env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, seeds=0)
env.reset()
buf = (ctypes.c_ulonglong * 32)()
for t in range(10):
    env.step(env.bot_actions(None))
lib.bbai_bot_prof_read(buf, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for t in range(steps):
    env.step(env.bot_actions(None))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
lib.bbai_bot_prof_read(buf, 0)
names = ["decide", "process_obs", "after_action", "find_obj_pos", "shortest_path", "search", "build_rows", "find_drop_pos", "before_action", "init", "go_keys"]
tot = buf[0] or 1
out = {"level": level, "envs": n, "ms_per_step": dt / steps * 1e3, "note": "wave wall-clock ticks (100 MHz) summed over waves; nested scopes overlap",
       "phases": {names[k]: {"share_of_decide": round(buf[k] / tot, 3), "scopes": int(buf[16 + k]),
                             "us_per_scope": round(buf[k] / max(buf[16 + k], 1) * 0.01, 2)} for k in range(len(names))}}
print(json.dumps(out))
