#!/usr/bin/env python3
"""In-process ablation of k_step phases (experiment): BBAI_STEP_ABLATE 0 full, 1 no transition/verifier,
2 no observation, 3 no obs copy-out.  Times the whole bbai_step call (no auto-reset so only k_step runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from babyai_amd.engine import BatchedBabyAIEnv
level, n = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("BossLevel", 1048576)
env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, seeds=0, auto_reset=False)
env.reset()
acts = torch.randint(0, 2, (16, n), dtype=torch.uint8, device="cuda")   # turns/forward only: nobody finishes early
def run(v, iters=12):
    os.environ["BBAI_STEP_ABLATE"] = v
    env.step(acts[0]); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for t in range(iters):
        env.step(acts[t % 16])
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for rep in range(3):
    print("rep", rep, {v: round(run(v), 4) for v in ["0", "4", "2", "3"]}, flush=True)

# correctness of variant 4 against variant 0 on a fresh pair of engines
a = BatchedBabyAIEnv("BabyAI-%s-v0" % level, 65536, seeds=5); b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, 65536, seeds=5)
a.reset(); b.reset()
acts2 = torch.randint(0, 7, (64, 65536), dtype=torch.uint8, device="cuda")
ok = True
for t in range(64):
    os.environ["BBAI_STEP_ABLATE"] = "0"; a.step(acts2[t])
    os.environ["BBAI_STEP_ABLATE"] = "4"; b.step(acts2[t])
    torch.cuda.synchronize()
    ok = ok and bool(torch.equal(a.image, b.image))
print("variant 4 == variant 0 over 64 random steps:", ok)
