#!/usr/bin/env python3
"""One-off GPU soak (not part of the test suite): large batches, many steps, scattered envs vs the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from babyai_amd.engine import BatchedBabyAIEnv
from oracle import levels as olevels

def soak(level, n, T, nspots, seed):
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, seeds=seed)
    env.reset()
    rs = np.random.RandomState(seed)
    spots = sorted(set([0, n - 1] + [int(x) for x in rs.randint(0, n, size=nspots)]))
    refs = []
    for i in spots:
        e = olevels.make_env(level); e.seed(seed + i); refs.append([e, e.reset()])
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    probs = torch.tensor([0.15, 0.15, 0.32, 0.1, 0.08, 0.15, 0.05], device="cuda")
    bad = 0
    t0 = time.time()
    for t in range(T):
        a = torch.multinomial(probs, n, replacement=True, generator=gen).to(torch.uint8)
        ah = a[spots].cpu().numpy()
        img = env.image[spots].cpu().numpy()
        for k, (e, o) in enumerate(refs):
            if not np.array_equal(img[k], o["image"]):
                bad += 1; print("MISMATCH", level, spots[k], t); break
        env.step(a)
        rew = env.reward[spots].cpu().numpy(); dn = env.done[spots].cpu().numpy()
        for k, (e, _) in enumerate(refs):
            o, r, d, _ = e.step(int(ah[k]))
            if np.float32(r) != rew[k] or bool(d) != bool(dn[k]):
                bad += 1; print("MISMATCH reward/done", level, spots[k], t)
            refs[k][1] = e.reset() if d else o
        if bad: break
    print("%-22s n=%8d T=%4d spots=%3d resets=%9d bad=%d gen_failures=%d  %.1fs" % (
        level, n, T, len(spots), env.reset_count() - n, bad, env.generator_failures(), time.time() - t0), flush=True)
    env.close()
    return bad

if __name__ == "__main__":
    total = 0
    total += soak("BossLevel", 1048576, 500, 40, 1)
    total += soak("SynthSeq", 262144, 400, 40, 2)
    total += soak("MiniBossLevel", 131072, 600, 40, 3)
    total += soak("Unlock", 131072, 400, 30, 4)
    total += soak("GoTo", 131072, 400, 30, 5)
    total += soak("KeyCorridorS6R3", 65536, 400, 30, 6)
    total += soak("PutNextS7N4Carrying", 65536, 300, 30, 7)
    total += soak("KeyInBox", 65536, 300, 30, 8)
    total += soak("GoToObjS4", 65536, 300, 40, 9)
    print("TOTAL BAD", total)
