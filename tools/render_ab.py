#!/usr/bin/env python3
"""In-process A/B of k_render variants (same box, same clocks, interleaved): BBAI_RENDER_VARIANT is read per call."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from babyai_amd.engine import BatchedBabyAIEnv
n = 1048576
env = BatchedBabyAIEnv("BabyAI-BossLevel-v0", n, pixel=True, seeds=0)
env.reset()
acts = torch.randint(0, 7, (8, n), dtype=torch.uint8, device="cuda")
for t in range(8):
    env.step(acts[t])
def time_render(variant, iters=10):
    os.environ["BBAI_RENDER_VARIANT"] = variant
    env._obs(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        env.lib.bbai_render(env.handle, env.image.data_ptr(), env.pixels.data_ptr(), env._stream())
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
ref = None
for rep in range(4):
    for v in sys.argv[1:] or ["0", "1"]:
        ms = time_render(v)
        print("rep %d variant %s: %.4f ms  (%.0f GB/s)" % (rep, v, ms, n * 9555 / ms / 1e6), flush=True)
# correctness of every variant against variant 0
os.environ["BBAI_RENDER_VARIANT"] = "0"; env._obs(); torch.cuda.synchronize(); base = env.pixels.clone()
for v in sys.argv[1:] or ["0", "1"]:
    os.environ["BBAI_RENDER_VARIANT"] = v; env.pixels.zero_(); env._obs(); torch.cuda.synchronize()
    print("variant", v, "equal to variant 0:", bool(torch.equal(base, env.pixels)))
