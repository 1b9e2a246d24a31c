#!/usr/bin/env python3
"""Why does a render launch shape rank differently inside the step loop than alone?  Times bbai_render of the engine
(a) alone on real observations, (b) alone on uniformly random cells, (c) alternating with k_step + k_consume, per shape.
BBAI_RENDER_GROUP / BBAI_RENDER_TPB pick the shape (read at create).  python tools/render_probe.py [envs]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from babyai_amd.action_stream import actions_torch  # noqa: E402
from babyai_amd.engine import BatchedBabyAIEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
dev = torch.device("cuda:0")
env = BatchedBabyAIEnv("BabyAI-BossLevel-v0", n, device=dev, pixel=True, seeds=0)
env.reset()
acts = actions_torch(1234, 0, 40, 0, n, dev)
for t in range(8):
    env.step(acts[t])
torch.cuda.synchronize()


def timed_renders(k):
    evs = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        env._obs()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs[4:])
    return ts[len(ts) // 2]


out = {"envs": n, "group": os.environ.get("BBAI_RENDER_GROUP"), "tpb": os.environ.get("BBAI_RENDER_TPB")}
out["alone_real_obs_ms"] = timed_renders(24)
# inside the loop: events around the render only
evs = []
for t in range(8, 40):
    env.lib.bbai_step(env.handle, acts[t].data_ptr(), env.image.data_ptr(), env.direction.data_ptr(), env.reward.data_ptr(),
                      env.reward64.data_ptr(), env.done.data_ptr(), 1, env._stream())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    env._obs()
    b.record()
    evs.append((a, b))
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in evs[4:])
out["in_loop_ms"] = ts[len(ts) // 2]
out["alone_again_ms"] = timed_renders(24)
saved = env.image.clone()
r = torch.randint(0, 256, saved.shape, dtype=torch.uint8, device=dev)
env.image.copy_(torch.stack([r[..., 0] % 8, r[..., 1] % 6, r[..., 2] % 3], dim=-1))
out["alone_random_cells_ms"] = timed_renders(24)
env.image.copy_(saved)
out["alone_real_after_ms"] = timed_renders(24)
print(json.dumps(out))
