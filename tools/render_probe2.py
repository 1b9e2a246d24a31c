#!/usr/bin/env python3
"""Which part of 'inside the step loop' slows the many-block render shapes?  The engine's render alternated with
(a) nothing, (b) a tiny unrelated kernel, (c) a rewrite of its own input, (d) 1 GiB of unrelated store traffic,
(e) 2 GB of unrelated scattered reads, (f) the real k_step + k_consume.  python tools/render_probe2.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from babyai_amd.action_stream import actions_torch  # noqa: E402
from babyai_amd.engine import BatchedBabyAIEnv  # noqa: E402

n = 1048576
dev = torch.device("cuda:0")
env = BatchedBabyAIEnv("BabyAI-BossLevel-v0", n, device=dev, pixel=True, seeds=0)
env.reset()
acts = actions_torch(1234, 0, 64, 0, n, dev)
for t in range(8):
    env.step(acts[t])
saved = env.image.clone()
small = torch.zeros(1024, device=dev)
scratch = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
table = torch.empty(1 << 28, dtype=torch.int64, device=dev)          # 2 GiB
idx = torch.randint(0, 1 << 28, (1 << 22,), device=dev)
torch.cuda.synchronize()
tcount = [8]


def real_step():
    t = tcount[0]
    tcount[0] = 8 + (t - 7) % 50
    env.lib.bbai_step(env.handle, acts[t].data_ptr(), env.image.data_ptr(), env.direction.data_ptr(), env.reward.data_ptr(),
                      env.reward64.data_ptr(), env.done.data_ptr(), 1, env._stream())


between = {
    "nothing": lambda: None,
    "tiny kernel": lambda: small.add_(1),
    "rewrite own input": lambda: env.image.copy_(saved),
    "1 GiB store traffic": lambda: scratch.fill_(3),
    "scattered reads": lambda: table[idx].sum(),
    "k_step + k_consume": real_step,
}
out = {"group": os.environ.get("BBAI_RENDER_GROUP"), "tpb": os.environ.get("BBAI_RENDER_TPB")}
for name, fn in between.items():
    evs = []
    for _ in range(28):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        env._obs()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs[4:])
    out[name] = round(ts[len(ts) // 2], 4)
print(json.dumps(out))
