"""The stand-alone oracle (oracle/levels.py on the restated gym_minigrid shim) must reproduce the
golden traces recorded from the reference itself (tools/gen_golden.py) byte for byte."""
import glob
import os

import numpy as np
import pytest

from oracle import levels as olevels

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def replay(g, n_envs=None, n_steps=None):
    from gym_minigrid.wrappers import RGBImgPartialObsWrapper
    name = str(g["level"])
    seeds = g["seeds"][:n_envs]
    T = g["actions"].shape[0] if n_steps is None else n_steps
    n_pix = min(g["pixels"].shape[1], len(seeds))
    for i, s in enumerate(seeds):
        env = olevels.make_env(name)
        env.seed(int(s))
        pix = RGBImgPartialObsWrapper(env) if i < n_pix else None
        for r in range(g["pre_image"].shape[0]):
            o = env.reset()
            assert np.array_equal(o["image"], g["pre_image"][r, i])
            assert o["mission"] == str(g["pre_mission"][r, i])
        o = env.reset()
        ev = {int(t): str(m) for t, e, m in zip(g["event_t"], g["event_env"], g["event_mission"]) if e == i}
        for t in range(T + 1):
            assert np.array_equal(o["image"], g["image"][t, i]), (name, i, t)
            assert o["direction"] == g["direction"][t, i]
            assert env.max_steps == g["max_steps"][t, i]
            if t in ev:
                assert o["mission"] == ev[t]
            if pix is not None:
                assert np.array_equal(pix.observation(o)["image"], g["pixels"][t, i])
            if t == T:
                break
            o, r, d, _ = env.step(int(g["actions"][t, i]))
            assert np.float32(r).view(np.uint32) == g["reward"][t, i].view(np.uint32)
            assert np.float64(r).view(np.uint64) == g["reward64"][t, i].view(np.uint64)
            assert bool(d) == bool(g["done"][t, i])
            if d:
                o = env.reset()


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_reference_trace(path):
    with np.load(path, allow_pickle=False) as f:
        g = {k: f[k] for k in f.files}      # decompress once (NpzFile re-reads on every access)
    replay(g)


C1_DIGEST = "d86a76540bee1af95d9efc7e4f6217a1f849f603dd33e6ba5eebccd24a079582"


def test_c1_plumbing_config_known_answer():
    """BASELINE.json configs[0]: GoToRedBall, 1 env, seed 0, 10 000 random-action steps.  The digest of every
    (image, reward, direction, done) was produced by the reference's own Level_GoToRedBall on the shim
    (`python -m oracle.cpu_baseline c1 --reference`) and is reproduced by the stand-alone oracle."""
    from oracle import cpu_baseline
    out = cpu_baseline.c1()
    assert out["digest"] == C1_DIGEST and "182 episodes" in out["sample"]
