#!/usr/bin/env python3
"""Follow-up to render_probe2.py: after 1 GiB of unrelated stores, which WARM-UP gives the (512, 2) render its 'alone' speed
back?  (a) none, (b) one read per 4 KiB of the pixel buffer (address translation only), (c) a read of the render's inputs,
(d) both, (e) one read per 64 B line of the pixel buffer's first 256 MiB, (f) a REWRITE of the render's inputs.  python tools/render_probe3.py"""
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
acts = actions_torch(1234, 0, 16, 0, n, dev)
for t in range(8):
    env.step(acts[t])
scratch = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
saved = env.image.clone()
pix_flat = env.pixels.view(-1)
pages = pix_flat[:: 4096]
sink = torch.zeros(1, dtype=torch.int64, device=dev)
torch.cuda.synchronize()


def touch_pages():
    sink.add_(pages.sum(dtype=torch.int64))


def read_inputs():
    sink.add_(env.image.view(-1).sum(dtype=torch.int64))


def read_head():
    sink.add_(pix_flat[: 1 << 28: 64].sum(dtype=torch.int64))


warm = {
    "none": lambda: None,
    "touch every 4 KiB of pixels": touch_pages,
    "read the inputs": read_inputs,
    "both": lambda: (touch_pages(), read_inputs()),
    "read head of pixels": read_head,
    "rewrite the inputs": lambda: env.image.copy_(saved),
    "rewrite the inputs twice": lambda: (env.image.copy_(saved), env.image.copy_(saved)),
}
out = {"group": os.environ.get("BBAI_RENDER_GROUP"), "tpb": os.environ.get("BBAI_RENDER_TPB")}
for name, fn in warm.items():
    evs = []
    for _ in range(24):
        scratch.fill_(3)
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        env._obs()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs[4:])
    out[name] = round(ts[len(ts) // 2], 4)
evs = []
for _ in range(24):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); env._obs(); b.record(); evs.append((a, b))
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in evs[4:])
out["alone"] = round(ts[len(ts) // 2], 4)
print(json.dumps(out))
