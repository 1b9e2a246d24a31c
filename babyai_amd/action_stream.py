"""Counter-based synthetic action stream: the action of global env `i` at step `t` of a bench run
is a pure function of (bench seed, t, i), i.i.d. uniform over the 7 MiniGrid actions -- the reference's own smoke test
samples `randint(0, n - 1)` over all of them, `done` included (babyai/levels/levelgen.py:522-527).  The GPU leg
(torch, any device), the CPU baseline leg and the parity checker (numpy) evaluate the same function, so they consume
identical actions without any transfer, and a shard sees the same actions whatever the number of ranks.

Mixing = the splitmix64 finaliser over an odd-multiplier combination of the three counters; the top 24 bits are
scaled to 0..6 (bias < 2^-21 per action).
"""
import numpy as np

_A = 0x9E3779B97F4A7C15
_B = 0xD1B54A32D192ED03
_C = 0x94D049BB133111EB
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
_MASK = (1 << 64) - 1


def actions_numpy(seed, t, first, count):
    """uint8[count]: actions of global envs first .. first+count-1 at step t."""
    with np.errstate(over="ignore"):
        i = np.arange(first, first + count, dtype=np.uint64)
        x = i * np.uint64(_A) + np.uint64((t * _B + seed * _C) & _MASK)
        x ^= x >> np.uint64(30)
        x *= np.uint64(_M1)
        x ^= x >> np.uint64(27)
        x *= np.uint64(_M2)
        x ^= x >> np.uint64(31)
        return (((x >> np.uint64(40)) * np.uint64(7)) >> np.uint64(24)).astype(np.uint8)


def action_scalar(seed, t, i):
    """The same function for one (t, i) in plain Python integers."""
    x = (i * _A + t * _B + seed * _C) & _MASK
    x ^= x >> 30
    x = (x * _M1) & _MASK
    x ^= x >> 27
    x = (x * _M2) & _MASK
    x ^= x >> 31
    return ((x >> 40) * 7) >> 24


def _s64(v):
    """Python int (mod 2^64) -> the int64 with the same bit pattern."""
    v &= _MASK
    return v - (1 << 64) if v >= (1 << 63) else v


def actions_torch(seed, t0, t1, first, count, device):
    """uint8[t1 - t0, count] on `device`: the block of steps t0 .. t1-1 (int64 arithmetic wraps mod 2^64; logical
    right shifts are emulated with a mask).  Evaluated for several steps at a time (<= 2^24 elements per pass: a long
    run of a small shard is thousands of steps)."""
    import torch
    i = torch.arange(first, first + count, dtype=torch.int64, device=device)
    base = (i * _s64(_A)).unsqueeze(0)
    out = torch.empty((t1 - t0, count), dtype=torch.uint8, device=device)

    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)

    chunk = max(1, (1 << 24) // max(1, count))
    for c0 in range(t0, t1, chunk):
        c1 = min(t1, c0 + chunk)
        offs = torch.tensor([_s64(t * _B + seed * _C) for t in range(c0, c1)], dtype=torch.int64, device=device).unsqueeze(1)
        x = base + offs
        x = x ^ lsr(x, 30)
        x = x * _s64(_M1)
        x = x ^ lsr(x, 27)
        x = x * _s64(_M2)
        x = x ^ lsr(x, 31)
        out[c0 - t0:c1 - t0] = ((lsr(x, 40) * 7) >> 24).to(torch.uint8)
    return out
