"""CPU baseline leg of bench.py: the stand-alone oracle (oracle/levels.py, one Python env object per
environment -- the reference's own execution model, babyai/rl/utils/penv.py:4-16) stepping the same
workload (same level, random actions over all 7 actions, auto-reset, optional pixel wrapper) on the
host cores of the box the bench runs on.  TEST/REPORTING INFRASTRUCTURE: a reported baseline
(kind = "port"), never a product path."""
import multiprocessing as mp
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)


def _worker(args):
    level, pixel, seconds, seed = args
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    import numpy as np
    from oracle import levels as olevels
    from gym_minigrid.wrappers import RGBImgPartialObsWrapper
    env = olevels.make_env(level)
    env.seed(seed)
    wrapped = RGBImgPartialObsWrapper(env) if pixel else env
    wrapped.reset()
    rng = np.random.RandomState(seed)
    acts = rng.randint(0, 7, size=4096)
    steps = 0
    t0 = time.perf_counter()
    while True:
        for a in acts:
            _, _, done, _ = wrapped.step(int(a))
            if done:
                wrapped.reset()
        steps += len(acts)
        if time.perf_counter() - t0 >= seconds:
            break
    return steps, time.perf_counter() - t0


def run(level="BossLevel", pixel=True, seconds=12.0):
    cores = os.cpu_count() or 1
    # one env-per-process on every host core (ParallelEnv's model), plus the single-core figure
    one_steps, one_dt = _worker((level, pixel, min(4.0, seconds / 3), 0))
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        res = pool.map(_worker, [(level, pixel, seconds, 1000 + i) for i in range(cores)])
    total = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return {
        "value": total / wall, "unit": "env-steps/s", "cores": cores, "kind": "port",
        "sample": "oracle/levels.py BabyAI-%s-v0%s, %d procs x %.0f s random-action rollouts with auto-reset (%d steps)"
                  % (level, " + RGBImgPartialObsWrapper" if pixel else "", cores, seconds, total),
        "single_core_value": one_steps / one_dt,
    }


def c1(steps=10000, seed=0, use_reference=False):
    """BASELINE.json configs[0]: BabyAI-GoToRedBall-v0, ONE env, seed 0, `steps` random actions over
    all 7 actions with auto-reset, encoded obs -- the plumbing / single-core CPU figure.  `use_reference` runs the
    reference's own level class on the shim instead of the stand-alone oracle (build container only).
    Also returns a digest of every (obs, reward, done) so the two can be compared."""
    import hashlib
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    import numpy as np
    if use_reference:
        from oracle import refenv
        refenv.import_reference()
        import gym
        env = gym.make("BabyAI-GoToRedBall-v0")
    else:
        from oracle import levels as olevels
        env = olevels.make_env("GoToRedBall")
    env.seed(seed)
    obs = env.reset()
    acts = np.random.RandomState(seed).randint(0, 7, size=steps)
    h = hashlib.sha256()
    episodes = 0
    t0 = time.perf_counter()
    for a in acts:
        obs, reward, done, _ = env.step(int(a))
        h.update(obs["image"].tobytes())
        h.update(np.float32(reward).tobytes())
        h.update(bytes([int(obs["direction"]), int(done)]))
        if done:
            episodes += 1
            obs = env.reset()
    dt = time.perf_counter() - t0
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "reference" if use_reference else "port",
            "sample": "BabyAI-GoToRedBall-v0, 1 env, seed %d, %d random-action steps, %d episodes" % (seed, steps, episodes),
            "digest": h.hexdigest()}


if __name__ == "__main__":
    import json
    if sys.argv[1:2] == ["c1"]:
        print(json.dumps(c1(use_reference="--reference" in sys.argv)))
        sys.exit(0)
    print(json.dumps(run(*(sys.argv[1:2] or ["BossLevel"]), pixel="--no-pixel" not in sys.argv, seconds=4.0)))
