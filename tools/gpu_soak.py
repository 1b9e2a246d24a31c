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

def soak_bot(level, n, T, nspots, seed):
    """Expert-driven (bbai_bot_act chooses every action, 5 % random): scattered envs against the host build of the
    same headers (tests/hostsim: env + bot in lockstep) -- much higher reset rates than random actions reach."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from babyai_amd.levels import make_cfg
    from hostsim_util import HostBot, HostEnv
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, seeds=seed)
    env.reset()
    rs = np.random.RandomState(seed)
    spots = sorted(set([0, n - 1] + [int(x) for x in rs.randint(0, n, size=nspots)]))
    hosts = []
    for i in spots:
        h = HostEnv(make_cfg(level), seed + i)
        img = h.reset()
        hosts.append([h, HostBot(h), img, True, None])
    gen = torch.Generator(device="cuda"); gen.manual_seed(seed)
    bad = 0
    prev = None
    t0 = time.time()
    for t in range(T):
        sug = env.bot_actions(prev)
        rnd = torch.randint(0, 7, (n,), dtype=torch.uint8, device="cuda", generator=gen)
        noise = torch.rand((n,), device="cuda", generator=gen) < 0.05
        act = torch.where((sug == 255) | noise, rnd, sug)
        sug_h, act_h = sug[spots].cpu().numpy(), act[spots].cpu().numpy()
        img = env.image[spots].cpu().numpy()
        for k, hb in enumerate(hosts):
            h, bot, himg, first, last = hb
            a = bot.decide(first, last)
            if not np.array_equal(img[k], himg) or sug_h[k] != (255 if a is None else a):
                bad += 1; print("MISMATCH", level, spots[k], t, sug_h[k], a); break
        if bad: break
        env.step(act)
        prev = act
        dn = env.done[spots].cpu().numpy()
        for k, hb in enumerate(hosts):
            himg, r, d = hb[0].step(int(act_h[k]))
            if bool(d) != bool(dn[k]):
                bad += 1; print("MISMATCH done", level, spots[k], t)
            hb[2], hb[3], hb[4] = (hb[0].reset(), True, None) if d else (himg, False, int(act_h[k]))
    print("%-22s n=%8d T=%4d spots=%3d resets=%9d bad=%d gen_failures=%d bot=%s  %.1fs (expert-driven)" % (
        level, n, T, len(spots), env.reset_count() - n, bad, env.generator_failures(), env.bot_stats(), time.time() - t0), flush=True)
    env.close()
    return bad


if __name__ == "__main__":
    total = 0
    if "bot" in sys.argv[1:]:
        total += soak_bot("BossLevel", 524288, 300, 40, 11)
        total += soak_bot("GoToObjS4", 131072, 400, 40, 12)
        total += soak_bot("PutNextLocal", 131072, 300, 40, 13)
        total += soak_bot("UnlockToUnlock", 65536, 300, 30, 14)
        total += soak_bot("PickupDist", 65536, 200, 30, 15)
        print("TOTAL BAD", total)
        sys.exit(1 if total else 0)
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
