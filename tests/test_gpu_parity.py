"""GPU parity (through the C ABI): the HIP engine replays the golden traces recorded from the
reference itself (tests/golden/*.npz, made by tools/gen_golden.py) and must reproduce every
byte: observation image, direction, reward bit pattern, done flag, mission string, max_steps."""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_trace(gpu, path):
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    g = np.load(path, allow_pickle=False)
    name = str(g["level"])
    seeds = g["seeds"]
    n = len(seeds)
    n_pix = g["pixels"].shape[1]
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % name, n, device=gpu, pixel=n_pix > 0)
    env.seed(seeds)
    for r in range(g["pre_image"].shape[0]):
        env.reset()
        torch.cuda.synchronize()
        assert np.array_equal(env.image.cpu().numpy(), g["pre_image"][r]), "pre-reset %d image" % r
        assert env.missions() == list(g["pre_mission"][r]), "pre-reset %d missions" % r
    obs = env.reset()
    ev = {}
    for t, e, m in zip(g["event_t"], g["event_env"], g["event_mission"]):
        ev.setdefault(int(t), []).append((int(e), str(m)))

    def check_obs(t, obs):
        torch.cuda.synchronize()
        assert np.array_equal(env.image.cpu().numpy(), g["image"][t]), "image at t=%d" % t
        assert np.array_equal(env.direction.cpu().numpy(), g["direction"][t]), "direction at t=%d" % t
        if n_pix:
            assert obs["image"].shape == (n, 56, 56, 3)
            assert np.array_equal(obs["image"][:n_pix].cpu().numpy(), g["pixels"][t]), "pixels at t=%d" % t
        if t in ev:
            ms = env.max_steps()
            for e, m in ev[t]:
                assert obs["mission"][e] == m, "mission env %d at t=%d" % (e, t)
                assert ms[e] == g["max_steps"][t, e]

    check_obs(0, obs)
    actions = torch.as_tensor(g["actions"], device=gpu)
    for t in range(actions.shape[0]):
        obs, reward, done, _ = env.step(actions[t])
        torch.cuda.synchronize()
        assert np.array_equal(reward.cpu().numpy().view(np.uint32), g["reward"][t].view(np.uint32)), "reward bits t=%d" % t
        assert np.array_equal(done.cpu().numpy(), g["done"][t]), "done t=%d" % t
        check_obs(t + 1, obs)
    env.close()


@pytest.mark.gpu
def test_manyenvs_freeze(gpu):
    """auto_reset=False: a finished env re-emits its last result (babyai/evaluate.py:73-81)."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    n = 64
    env = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", n, device=gpu, auto_reset=False)
    env.seed(7)
    env.reset()
    rng = np.random.RandomState(0)
    last = None
    was_done = np.zeros(n, bool)
    for t in range(80):     # max_steps = 64: everything finishes
        a = torch.as_tensor(rng.randint(0, 7, size=n).astype(np.uint8), device=gpu)
        obs, reward, done, _ = env.step(a)
        torch.cuda.synchronize()
        cur = (env.image.cpu().numpy().copy(), reward.cpu().numpy().copy(), done.cpu().numpy().copy())
        if last is not None and was_done.any():
            for k in range(3):
                assert np.array_equal(cur[k][was_done], last[k][was_done])
        was_done |= cur[2].astype(bool)
        last = cur
    assert was_done.all()
    assert env.reset_count() == n
    env.close()
