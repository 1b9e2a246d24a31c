#!/usr/bin/env python3
"""Where a decision of the expert goes: wave wall-clock per phase of bbai_bot.hpp (BOT_PROF scopes), on an engine built with
-DBBAI_BOT_PROF (an experiment build, never the product's):

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DBBAI_BOT_PROF -o tools/_prof/libbbai_botprof.so babyai_amd/csrc/bbai_engine.hip
    BBAI_ENGINE_LIB=tools/_prof/libbbai_botprof.so python tools/bot_prof.py BossLevel 262144 20

Scopes nest (decide > before > path > search ...): every figure is the time between a scope's entry and exit as seen by the first active
lane of a wave, summed over the waves; shares are of `decide`."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from babyai_amd import engine  # noqa: E402
from babyai_amd.engine import BatchedBabyAIEnv  # noqa: E402

NAMES = ["decide", "obs", "after_action", "find_obj_pos", "shortest_path", "search", "mask_rows", "find_drop_pos", "before_action", "eager_search_1", "keys"]
level = sys.argv[1] if len(sys.argv) > 1 else "BossLevel"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device="cuda:0", seeds=0)
env.reset()
L = engine.load_library()
L.bbai_bot_prof_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 32)()
for t in range(steps + 10):
    if t == 10:
        torch.cuda.synchronize()
        L.bbai_bot_prof_read(out, 1)
    a = env.bot_actions(None)
    env.step(torch.where(a == 255, torch.zeros_like(a), a))
torch.cuda.synchronize()
L.bbai_bot_prof_read(out, 0)
total = max(out[0], 1)
print(json.dumps({"level": level, "envs": n, "steps": steps, "bot_group": env.get_option("bot_group"),
                  "ticks": {k: int(out[i]) for i, k in enumerate(NAMES)}, "entries": {k: int(out[16 + i]) for i, k in enumerate(NAMES)},
                  "share_of_decide": {k: round(out[i] / total, 3) for i, k in enumerate(NAMES)}}))
