#!/usr/bin/env python3
"""Speed of the stand-alone oracle port (oracle/levels.py) against the REFERENCE itself (/root/reference/babyai on the
oracle shim) on the BASELINE workloads -- build container only (the reference tree never reaches the GPU box).

For every workload: the same envs (seeds base + i), the same counter-based action stream, auto-reset, the pixel wrapper
where the config has it; reference and port are stepped alternately in rounds (so that box noise hits both), every
(image, direction, float64 reward, done) goes into a SHA-256 per implementation, and the digests must be equal.
Writes profiles/r03/cpu_port_vs_reference.json; bench.py quotes `reference_over_port` from it next to `cpu_baseline`
(kind "port"), so that a reader can convert the port's figure into the reference's.

    python tools/cpu_port_vs_reference.py [--steps 3000] [--envs 4] [--rounds 3]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKLOADS = [("GoToRedBall", False), ("GoToLocal", False), ("PickupLoc", False), ("GoTo", False), ("BossLevel", False), ("BossLevel", True)]


def rollout(make, level, pixel, seed, action_seed, index, steps, h):
    import numpy as np
    from babyai_amd.action_stream import action_scalar
    from gym_minigrid.wrappers import RGBImgPartialObsWrapper
    env = make(level)
    env.seed(seed + index)
    w = RGBImgPartialObsWrapper(env) if pixel else env
    t0 = time.perf_counter()
    o = w.reset()
    for t in range(steps):
        o, r, d, _ = w.step(action_scalar(action_seed, t, index))
        h.update(np.ascontiguousarray(o["image"]).tobytes())
        h.update(np.float64(r).tobytes())
        h.update(bytes([int(d)]) if pixel else bytes([int(o["direction"]), int(d)]))
        if d:
            o = w.reset()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--envs", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03", "cpu_port_vs_reference.json"))
    args = ap.parse_args()
    from oracle import refenv
    refenv.import_reference()
    import gym
    from oracle import levels as olevels

    def make_ref(level):
        return gym.make("BabyAI-%s-v0" % level)

    def make_port(level):
        return olevels.make_env(level)

    table = {}
    for level, pixel in WORKLOADS:
        t_ref = t_port = 0.0
        ok = True
        for rnd in range(args.rounds):
            for i in range(args.envs):
                idx = rnd * args.envs + i
                ha, hb = hashlib.sha256(), hashlib.sha256()
                if (rnd + i) % 2:        # alternate who goes first
                    t_port += rollout(make_port, level, pixel, 0, 1234, idx, args.steps, hb)
                    t_ref += rollout(make_ref, level, pixel, 0, 1234, idx, args.steps, ha)
                else:
                    t_ref += rollout(make_ref, level, pixel, 0, 1234, idx, args.steps, ha)
                    t_port += rollout(make_port, level, pixel, 0, 1234, idx, args.steps, hb)
                ok = ok and ha.hexdigest() == hb.hexdigest()
        n = args.rounds * args.envs * args.steps
        row = {"reference_steps_per_s": n / t_ref, "port_steps_per_s": n / t_port, "reference_over_port": t_port / t_ref,
               "digest_equal": ok, "steps": n, "envs": args.rounds * args.envs}
        table["%s/%s" % (level, "pixel" if pixel else "encoded")] = row
        print(level, "pixel" if pixel else "encoded", json.dumps(row), flush=True)
        assert ok, "port and reference disagree on %s" % level
    try:
        commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()
    except Exception:
        commit = None
    with open(args.out, "w") as f:
        json.dump({"what": "one core, reference (babyai on the oracle shim) vs port (oracle/levels.py), same seeds and actions, alternated",
                   "where": "build container (%d usable cores), python %s" % (len(os.sched_getaffinity(0)), sys.version.split()[0]),
                   "commit": commit, "workloads": table}, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
