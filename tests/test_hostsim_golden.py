"""The engine's C++ core (host build, one lane) replays EVERY golden trace recorded from the reference
(random-action and expert-driven) in full: images, direction, reward bits, done, missions, max_steps.
Cheap (native code), so it covers all envs and all steps of every fixture on the CPU."""
import glob
import os

import numpy as np
import pytest

from babyai_amd.levels import make_cfg
from hostsim_util import HostEnv

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("order", ["reference", "k_step"])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_core_replays_reference_trace(path, order):
    """order = "k_step": the step in the kernel's order of operations (pose, front-cell id fetched before the object
    actions and corrected by what they wrote, verifier on that id: bbai_step.hpp step_env_prefetch)."""
    with np.load(path, allow_pickle=False) as f:
        g = {k: f[k] for k in f.files}      # decompress once (NpzFile re-reads on every access)
    name = str(g["level"])
    cfg = make_cfg(name)
    T = g["actions"].shape[0]
    for i, s in enumerate(g["seeds"]):
        sim = HostEnv(cfg, int(s))
        sim.prefetch_order = order == "k_step"
        for r in range(g["pre_image"].shape[0]):
            assert np.array_equal(sim.reset(), g["pre_image"][r, i])
            assert sim.mission == str(g["pre_mission"][r, i])
        img = sim.reset()
        ev = {int(t): str(m) for t, e, m in zip(g["event_t"], g["event_env"], g["event_mission"]) if e == i}
        for t in range(T + 1):
            assert np.array_equal(img, g["image"][t, i]), (name, i, t)
            assert sim.agent[2] == g["direction"][t, i]
            assert sim.max_steps == g["max_steps"][t, i]
            if t in ev:
                assert sim.mission == ev[t]
            if t == T:
                break
            img, rew, done = sim.step(int(g["actions"][t, i]))
            assert rew.view(np.uint32) == g["reward"][t, i].view(np.uint32), (name, i, t)
            assert np.float64(sim.last_reward64).view(np.uint64) == g["reward64"][t, i].view(np.uint64), (name, i, t)
            assert done == bool(g["done"][t, i]), (name, i, t)
            if done:
                img = sim.reset()
