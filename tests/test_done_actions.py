"""The reference's BABYAI_DONE_ACTIONS verifier mode (babyai/levels/verifier.py:17,216-230,543-545).

Fixtures: tests/golden/done_actions/*.npz, recorded by tools/gen_golden_done.py from the reference ITSELF imported with the
variable set (expert-driven episodes with `done` actions mixed in: successes by a well-timed `done`, failures by any
other).  Checked against them: the oracle's restatement (CPU), the engine's per-env core compiled for the host (CPU), and
the HIP engine through the C ABI (-m gpu)."""
import ctypes
import glob
import os

import numpy as np
import pytest

from oracle import levels as olevels

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "done_actions", "*.npz")))
IDS = [os.path.basename(p)[:-4] for p in GOLDEN]
# the same protocol with every `done` passed as the ENUM MEMBER env.actions.done (tools/gen_golden_done.py --enum): what the
# reference's own expert returns (babyai/bot.py:593), and the only kind of `done` AndInstr's failure rule reacts to (verifier.py:543-545)
GOLDEN_ENUM = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "done_actions_enum", "*.npz")))
IDS_ENUM = [os.path.basename(p)[:-4] for p in GOLDEN_ENUM]


def load(path):
    with np.load(path, allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def test_fixtures_cover_successes_and_failures():
    assert len(GOLDEN) >= 8
    for path in GOLDEN:
        g = load(path)
        by_done = (g["actions"] == 6) & (g["done"] == 1)
        assert ((g["reward64"] > 0) & by_done).sum() >= 2, path        # a `done` right after the completing step
        assert ((g["reward64"] == 0) & by_done).sum() >= 2, path       # a `done` anywhere else fails the instruction


def _oracle_replay(g, as_enum, stop_at_mismatch=False):
    """Replays a fixture through the oracle; `as_enum`: a `done` is passed as env.actions.done instead of the int 6."""
    name = str(g["level"])
    T = g["actions"].shape[0]
    for i, s in enumerate(g["seeds"]):
        env = olevels.make_env(name)
        env.seed(int(s))
        o = env.reset()
        for t in range(T):
            a = int(g["actions"][t, i])
            o, r, d, _ = env.step(env.actions.done if (as_enum and a == 6) else a)
            ok = np.float64(r).view(np.uint64) == g["reward64"][t, i].view(np.uint64) and bool(d) == bool(g["done"][t, i])
            if d:
                o = env.reset()
            ok = ok and np.array_equal(o["image"], g["image"][t + 1, i]) and o["direction"] == g["direction"][t + 1, i]
            if not ok:
                if stop_at_mismatch:
                    return (name, i, t)
                raise AssertionError((name, i, t))
    return None


@pytest.mark.parametrize("path", GOLDEN_ENUM, ids=IDS_ENUM)
def test_oracle_matches_reference_when_done_is_the_enum_member(path, monkeypatch):
    monkeypatch.setattr(olevels, "DONE_ACTIONS", True)
    assert _oracle_replay(load(path), True) is None


def test_the_enum_fixtures_depend_on_the_identity_of_done(monkeypatch):
    """Replayed with int 6 instead of the enum member the oracle must diverge somewhere: AndInstr's failure rule fired while
    the fixtures were recorded."""
    monkeypatch.setattr(olevels, "DONE_ACTIONS", True)
    assert len(GOLDEN_ENUM) >= 3
    diverged = [p for p in GOLDEN_ENUM if _oracle_replay(load(p), False, stop_at_mismatch=True) is not None]
    assert len(diverged) >= 2, diverged


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("path", GOLDEN_ENUM, ids=IDS_ENUM)
def test_host_build_matches_reference_when_done_is_the_enum_member(path, order):
    """bbai_step.hpp verify_side's `enum_done` rule, compiled for the host, in the reference's order of operations (0) and in
    k_step's (1)."""
    from babyai_amd.levels import make_cfg
    from hostsim_util import HostEnv, lib
    L = lib()
    L.hs_step64_done_enum.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    g = load(path)
    name = str(g["level"])
    T = g["actions"].shape[0]
    for i, s in enumerate(g["seeds"]):
        h = HostEnv(make_cfg(name), int(s))
        img = h.reset()
        lsm = ctypes.c_uint32(0)
        for t in range(T + 1):
            assert np.array_equal(img, g["image"][t, i]), (name, i, t)
            if t == T:
                break
            rew = ctypes.c_double(0)
            d = L.hs_step64_done_enum(ctypes.byref(h.cfg), h.rec.ctypes.data, h.hot.ctypes.data, ctypes.byref(h.stale),
                                      int(g["actions"][t, i]), ctypes.byref(rew), ctypes.byref(lsm), order)
            assert np.float64(rew.value).view(np.uint64) == g["reward64"][t, i].view(np.uint64), (name, i, t)
            assert bool(d) == bool(g["done"][t, i]), (name, i, t)
            if d:
                img = h.reset()
                lsm = ctypes.c_uint32(0)
            else:
                img = h.observe()


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_oracle_matches_reference_in_done_action_mode(path, monkeypatch):
    monkeypatch.setattr(olevels, "DONE_ACTIONS", True)
    g = load(path)
    name = str(g["level"])
    T = g["actions"].shape[0]
    for i, s in enumerate(g["seeds"]):
        env = olevels.make_env(name)
        env.seed(int(s))
        o = env.reset()
        ev = {int(t): str(m) for t, e, m in zip(g["event_t"], g["event_env"], g["event_mission"]) if e == i}
        for t in range(T + 1):
            assert np.array_equal(o["image"], g["image"][t, i]), (name, i, t)
            assert o["direction"] == g["direction"][t, i]
            if t in ev:
                assert o["mission"] == ev[t]
            if t == T:
                break
            o, r, d, _ = env.step(int(g["actions"][t, i]))
            assert np.float64(r).view(np.uint64) == g["reward64"][t, i].view(np.uint64), (name, i, t)
            assert bool(d) == bool(g["done"][t, i]), (name, i, t)
            if d:
                o = env.reset()


def test_oracle_normal_mode_differs_on_these_traces(monkeypatch):
    """The fixtures must actually depend on the mode: replayed in the normal mode the oracle diverges."""
    monkeypatch.setattr(olevels, "DONE_ACTIONS", False)
    g = load([p for p in GOLDEN if p.endswith("GoToLocal.npz")][0])
    env = olevels.make_env("GoToLocal")
    env.seed(int(g["seeds"][0]))
    env.reset()
    same = True
    for t in range(g["actions"].shape[0]):
        _, r, d, _ = env.step(int(g["actions"][t, 0]))
        if bool(d) != bool(g["done"][t, 0]) or np.float64(r) != g["reward64"][t, 0]:
            same = False
            break
        if d:
            env.reset()
    assert not same


@pytest.mark.parametrize("order", ["reference", "k_step"])
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_host_build_matches_reference_in_done_action_mode(path, order):
    """bbai_step.hpp's verifier with the lastStepMatch bits (`lsm`), compiled for the host, against the same traces -- in the
    reference's order of operations and in k_step's (step_env_prefetch)."""
    from babyai_amd.levels import make_cfg
    from hostsim_util import HostEnv, lib
    L = lib()
    L.hs_step64_done.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    step = L.hs_step64_done if order == "reference" else L.hs_step64_prefetch
    g = load(path)
    name = str(g["level"])
    T = g["actions"].shape[0]
    for i, s in enumerate(g["seeds"]):
        h = HostEnv(make_cfg(name), int(s))
        img = h.reset()
        lsm = ctypes.c_uint32(0)
        for t in range(T + 1):
            assert np.array_equal(img, g["image"][t, i]), (name, i, t)
            assert h.agent[2] == g["direction"][t, i]
            if t == T:
                break
            rew = ctypes.c_double(0)
            d = step(ctypes.byref(h.cfg), h.rec.ctypes.data, h.hot.ctypes.data, ctypes.byref(h.stale),
                     int(g["actions"][t, i]), ctypes.byref(rew), ctypes.byref(lsm))
            assert np.float64(rew.value).view(np.uint64) == g["reward64"][t, i].view(np.uint64), (name, i, t)
            assert bool(d) == bool(g["done"][t, i]), (name, i, t)
            if d:
                img = h.reset()
                lsm = ctypes.c_uint32(0)
            else:
                img = h.observe()


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_engine_matches_reference_in_done_action_mode(gpu, path):
    """The HIP engine through the C ABI, done-action mode switched on for the batch (bbai_set_done_actions)."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    g = load(path)
    name = str(g["level"])
    seeds = g["seeds"]
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % name, len(seeds), device=gpu, seeds=seeds, done_actions=True)
    assert env.done_actions
    obs = env.reset()
    ev = {(int(t), int(e)): str(m) for t, e, m in zip(g["event_t"], g["event_env"], g["event_mission"])}
    T = g["actions"].shape[0]
    for t in range(T + 1):
        assert np.array_equal(env.image.cpu().numpy(), g["image"][t]), (name, t)
        assert np.array_equal(env.direction.cpu().numpy(), g["direction"][t])
        for (tt, e), m in ev.items():
            if tt == t:
                assert obs["mission"][e] == m
        if t == T:
            break
        obs, r, d, _ = env.step(torch.as_tensor(g["actions"][t], device=gpu))
        assert np.array_equal(env.reward64.cpu().numpy().view(np.uint64), g["reward64"][t].view(np.uint64)), (name, t)
        assert np.array_equal(d.cpu().numpy(), g["done"][t]), (name, t)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN_ENUM, ids=IDS_ENUM)
def test_engine_matches_reference_when_done_is_the_enum_member(gpu, path):
    """The HIP engine with bbai_set_option("done_action_enum", 1): bbai_step's `done` actions count as the enum member."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    g = load(path)
    name = str(g["level"])
    seeds = g["seeds"]
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % name, len(seeds), device=gpu, seeds=seeds, done_actions=True)
    env.set_option("done_action_enum", 1)
    env.reset()
    T = g["actions"].shape[0]
    for t in range(T + 1):
        assert np.array_equal(env.image.cpu().numpy(), g["image"][t]), (name, t)
        if t == T:
            break
        _, r, d, _ = env.step(torch.as_tensor(g["actions"][t], device=gpu))
        assert np.array_equal(env.reward64.cpu().numpy().view(np.uint64), g["reward64"][t].view(np.uint64)), (name, t)
        assert np.array_equal(d.cpu().numpy(), g["done"][t]), (name, t)
    env.close()


@pytest.mark.gpu
def test_bot_rollout_treats_the_experts_done_as_the_enum_member(gpu):
    """bbai_bot_rollout's actions are the expert's own (babyai/bot.py:593 returns the member): in done-action mode it steps
    like the host loop bot_actions() + step() with "done_action_enum" on."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    n, T = 256, 96
    a = BatchedBabyAIEnv("BabyAI-SynthSeq-v0", n, device=gpu, seeds=900, done_actions=True)
    b = BatchedBabyAIEnv("BabyAI-SynthSeq-v0", n, device=gpu, seeds=900, done_actions=True)
    b.set_option("done_action_enum", 1)
    a.reset()
    b.reset()
    r = a.bot_rollout(T)
    reset_cmd = torch.full((n,), b.RESET_ENV, dtype=torch.uint8, device=gpu)
    for t in range(T):
        assert torch.equal(r["image"][t], b.image), t
        act = b.bot_actions(None)
        act = torch.where(act == b.BOT_GAVE_UP, reset_cmd, act)
        assert torch.equal(r["action"][t], act), t
        _, rew, done, _ = b.step(act)
        assert torch.equal(r["reward"][t], rew) and torch.equal(r["done"][t], done), t
    assert torch.equal(a.image, b.image)
    assert int(r["done"].sum()) > n // 4
    a.close()
    b.close()


@pytest.mark.gpu
def test_done_action_mode_follows_the_environment_variable_and_survives_checkpoints(gpu, monkeypatch):
    """verifier.py:17: the mode is on iff BABYAI_DONE_ACTIONS is non-empty when the reference is imported -- here, when the
    batch is created; the per-env lastStepMatch bits are part of a checkpoint."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    assert not BatchedBabyAIEnv("BabyAI-GoToLocal-v0", 4, device=gpu).done_actions
    monkeypatch.setenv("BABYAI_DONE_ACTIONS", "0")            # any non-empty string, as in the reference
    a = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", 64, device=gpu, seeds=5)
    b = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", 64, device=gpu, seeds=77)
    assert a.done_actions and b.done_actions
    a.reset()
    b.reset()
    rng = np.random.RandomState(0)
    for t in range(40):
        a.step(torch.as_tensor(rng.choice([0, 1, 2, 2, 2, 6], 64).astype(np.uint8), device=gpu))
    b.load_checkpoint(a.save_checkpoint())
    for t in range(80):
        act = torch.as_tensor(rng.choice([0, 1, 2, 2, 2, 6], 64).astype(np.uint8), device=gpu)
        _, ra, da, _ = a.step(act)
        _, rb, db, _ = b.step(act)
        assert torch.equal(a.image, b.image) and torch.equal(ra, rb) and torch.equal(da, db), t
    a.close()
    b.close()
