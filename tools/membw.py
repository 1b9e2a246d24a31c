#!/usr/bin/env python3
"""Achievable HBM bandwidth on this box for the access shapes the engine uses (reported next to
the 8 TB/s spec peak): pure write stream (fill), read+write stream (copy)."""
import json
import torch

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3

n = 1048576 * 9408
x = torch.empty(n, dtype=torch.uint8, device="cuda")
y = torch.empty(n, dtype=torch.uint8, device="cuda")
xi = x.view(torch.int32)
yi = y.view(torch.int32)
out = {
    "bytes": n,
    "fill_u8_GBs": n / timeit(lambda: x.fill_(7)) / 1e9,
    "fill_i32_GBs": n / timeit(lambda: xi.fill_(7)) / 1e9,
    "memset_GBs": n / timeit(lambda: x.zero_()) / 1e9,
    "copy_total_GBs": 2 * n / timeit(lambda: yi.copy_(xi)) / 1e9,
}
print(json.dumps(out))
