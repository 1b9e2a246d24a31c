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
    with np.load(path, allow_pickle=False) as f:
        g = {k: f[k] for k in f.files}      # decompress once (NpzFile re-reads on every access)
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
        # the reference's own return value (a Python float, levelgen.py:59-61), bit for bit
        assert np.array_equal(env.reward64.cpu().numpy().view(np.uint64), g["reward64"][t].view(np.uint64)), "f64 reward bits t=%d" % t
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


@pytest.mark.gpu
@pytest.mark.parametrize("level,pixel", [("GoToLocal", False), ("PickupLoc", True), ("BossLevel", False)])
def test_parallel_env_adapter_vs_oracle(gpu, level, pixel):
    """The ParallelEnv-protocol adapter against a list of oracle envs driven the way
    babyai/rl/utils/penv.py drives them (seeds 100*seed+i as in scripts/train_rl.py:59)."""
    from babyai_amd.vec_env import BatchedParallelEnv
    from babyai_amd import integrate
    from oracle import levels as olevels
    from gym_minigrid.wrappers import RGBImgPartialObsWrapper
    n, T = 12, 150
    seeds = [100 * 3 + i for i in range(n)]
    # built the way a training script would with the no-edit integration (babyai_amd/integrate.py): seeds 100*seed+i, and
    # the object answers the list operations BaseAlgo / train_rl.py apply to `envs`
    venv = integrate.make_envs("BabyAI-%s-v0" % level, n, 3, pixel=pixel, device=gpu)
    assert isinstance(venv, BatchedParallelEnv) and len(venv) == n and venv[0].action_space.n == 7
    assert [int(s) for s in venv.engine.seeds] == seeds
    refs = []
    for s in seeds:
        e = olevels.make_env(level)
        if pixel:
            e = RGBImgPartialObsWrapper(e)
        e.seed(s)
        refs.append(e)
    obs = venv.reset()
    robs = [e.reset() for e in refs]
    rng = np.random.RandomState(5)
    assert venv.action_space.n == 7
    assert venv.observation_space.spaces["image"].shape == ((56, 56, 3) if pixel else (7, 7, 3))
    for t in range(T):
        for i in range(n):
            assert np.array_equal(obs[i]["image"], robs[i]["image"]), (t, i)
            assert obs[i]["mission"] == robs[i]["mission"], (t, i)
            if not pixel:
                assert obs[i]["direction"] == robs[i]["direction"]
        a = rng.randint(0, 7, size=n)
        obs, reward, done, info = venv.step(a)
        rr = []
        for i, e in enumerate(refs):
            o, r, d, _ = e.step(int(a[i]))
            if d:
                o = e.reset()
            robs[i] = o
            rr.append((r, d))
        assert [float(x[0]) for x in rr] == list(reward) and all(type(x) is float for x in reward)     # f64, as the reference
        assert [bool(x[1]) for x in rr] == list(done)
    venv.close()


@pytest.mark.gpu
def test_shard_independence(gpu):
    """Env i's trajectory depends only on its seed and actions: running the batch as two shards gives the
    same bytes as one batch (the multi-GPU decomposition, exercised on one device)."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    n, T = 512, 96
    acts = torch.randint(0, 7, (T, n), dtype=torch.uint8, device=gpu)
    whole = BatchedBabyAIEnv("BabyAI-BossLevel-v0", n, device=gpu, seeds=50)
    a = BatchedBabyAIEnv("BabyAI-BossLevel-v0", n // 2, device=gpu, seeds=50)
    b = BatchedBabyAIEnv("BabyAI-BossLevel-v0", n // 2, device=gpu, seeds=50 + n // 2)
    for e in (whole, a, b):
        e.reset()
    for t in range(T):
        whole.step(acts[t])
        a.step(acts[t, : n // 2])
        b.step(acts[t, n // 2:])
    torch.cuda.synchronize()
    assert torch.equal(whole.image, torch.cat([a.image, b.image]))
    assert torch.equal(whole.reward, torch.cat([a.reward, b.reward]))
    assert torch.equal(whole.done, torch.cat([a.done, b.done]))
    assert whole.missions() == a.missions() + b.missions()


@pytest.mark.gpu
@pytest.mark.parametrize("level", ["BossLevel", "PutNextLocal", "SynthSeq"])
def test_device_mission_tokens(gpu, level):
    """k_tokens == tokenising the mission string the oracle/reference produces (format.py:64 regex split),
    initially and after auto-resets; the tensor preprocessor pads like InstructionsPreprocessor."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.missions import tokenize
    from babyai_amd.preprocess import TensorObssPreprocessor
    n = 256
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=11)
    env.reset()
    pre = TensorObssPreprocessor(env)          # registered after reset: rows are filled immediately
    rng = np.random.RandomState(1)
    for t in range(130 if level == "PutNextLocal" else 30):
        obs, _, _, _ = env.step(torch.as_tensor(rng.randint(0, 7, size=n).astype(np.uint8), device=gpu))
    batch = pre(obs)
    torch.cuda.synchronize()
    tok = env.instr.cpu().numpy()
    missions = env.missions()
    longest = 0
    for i in range(n):
        ids = tokenize(missions[i])
        assert list(tok[i, :len(ids)]) == ids and not tok[i, len(ids):].any(), (i, missions[i])
        longest = max(longest, len(ids))
    assert batch.instr.shape == (n, longest) and batch.instr.dtype == torch.int64
    assert batch.image.dtype == torch.float32 and batch.image.shape == (n, 7, 7, 3)
    assert torch.equal(batch.image.to(torch.uint8), env.image)
    sub = batch[torch.arange(0, n, 2, device=gpu)]
    assert len(sub) == n // 2 and sub.instr.shape[0] == n // 2
    # a reference vocabulary (ids in first-seen order, format.py:15-41) through the remap table on the device
    words = list(reversed(pre.words()))[:20]
    ref_vocab = {w: i + 1 for i, w in enumerate(words)}
    pre2 = TensorObssPreprocessor(env, vocab=ref_vocab)
    b2 = pre2(obs)
    full = pre2.vocab_dict()
    assert all(full[w] == ref_vocab[w] for w in ref_vocab) and sorted(full.values()) == list(range(1, 33))
    got = b2.instr.cpu().numpy()
    import re
    for i in range(0, n, 7):
        ids = [full[w] for w in re.findall("([a-z]+)", missions[i])]
        assert list(got[i, :len(ids)]) == ids and not got[i, len(ids):].any()
    fixed = pre2(obs, trim=False)                  # the level's fixed width: no host synchronisation
    assert fixed.instr.shape == (n, env.max_mission_tokens) and torch.equal(fixed.instr[:, :longest], b2.instr)
    if level == "PutNextLocal":
        assert env.reset_count() > n
    env.close()


def _oracle_envs(level, seeds, pixel=False):
    from oracle import levels as olevels
    out = []
    for s in seeds:
        e = olevels.make_env(level)
        e.seed(int(s))
        out.append(e)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 257, 1000])
def test_odd_batch_sizes(gpu, n):
    """Batch sizes that are not multiples of the wave / block / render-group sizes."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    env = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", n, device=gpu, pixel=True, seeds=3)
    obs = env.reset()
    spots = sorted(set([0, n // 2, n - 1]))
    refs = _oracle_envs("GoToLocal", [3 + i for i in spots])
    ro = [e.reset() for e in refs]
    rng = np.random.RandomState(n)
    from gym_minigrid.wrappers import RGBImgPartialObsWrapper
    for t in range(90):
        img = env.image.cpu().numpy()
        for k, i in enumerate(spots):
            assert np.array_equal(img[i], ro[k]["image"]), (n, i, t)
        if t % 30 == 0:
            pix = obs["image"].cpu().numpy()
            for k, i in enumerate(spots):
                assert np.array_equal(pix[i], refs[k].get_obs_render(ro[k]["image"], tile_size=8)), (n, i, t)
        a = rng.randint(0, 7, size=n).astype(np.uint8)
        obs, reward, done, _ = env.step(torch.as_tensor(a, device=gpu))
        for k, i in enumerate(spots):
            o, r, d, _ = refs[k].step(int(a[i]))
            if d:
                o = refs[k].reset()
            ro[k] = o
    env.close()


@pytest.mark.gpu
def test_explicit_resets_reseed_and_manyenvs_windows(gpu):
    """reset() at arbitrary points (the look-ahead windows must hand out the stream's levels in order), ManyEnvs
    mode (no auto-reset), then re-seeding the same engine."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    n = 96
    env = BatchedBabyAIEnv("BabyAI-MiniBossLevel-v0", n, device=gpu, auto_reset=False)
    rng = np.random.RandomState(4)
    for base in (21, 500):                       # second pass = re-seed of a used engine
        env.seed(base)
        refs = _oracle_envs("MiniBossLevel", [base + i for i in range(n)])
        frozen = np.zeros(n, bool)
        last = [None] * n
        for phase, steps in enumerate([0, 3, 1, 7, 0, 2, 25, 4, 1, 1, 1, 9]):
            env.reset()
            torch.cuda.synchronize()
            ro = [e.reset() for e in refs]
            frozen[:] = False
            img = env.image.cpu().numpy()
            ms = env.missions()
            for i in range(n):
                assert np.array_equal(img[i], ro[i]["image"]), (base, phase, i)
                assert ms[i] == ro[i]["mission"], (base, phase, i)
            for t in range(steps):
                a = rng.randint(0, 7, size=n).astype(np.uint8)
                obs, reward, done, _ = env.step(torch.as_tensor(a, device=gpu))
                img = env.image.cpu().numpy()
                rew = reward.cpu().numpy()
                dn = done.cpu().numpy()
                for i in range(n):
                    if not frozen[i]:
                        o, r, d, _ = refs[i].step(int(a[i]))
                        last[i] = (o["image"], np.float32(r), bool(d))
                        frozen[i] = bool(d)
                    assert np.array_equal(img[i], last[i][0]) and rew[i] == last[i][1] and bool(dn[i]) == last[i][2]
    env.close()


@pytest.mark.gpu
def test_every_registered_level_vs_oracle(gpu):
    """All 105 level ids on the device (generator wave paths, verifier, observation) against the oracle:
    two resets + 30 mixed steps each, 12 envs per level."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.levels import LEVELS
    n = 12
    rng = np.random.RandomState(77)
    for name in sorted(LEVELS):
        env = BatchedBabyAIEnv("BabyAI-%s-v0" % name, n, device=gpu, seeds=40)
        refs = _oracle_envs(name, [40 + i for i in range(n)])
        for rep in range(2):
            env.reset()
            ro = [e.reset() for e in refs]
        ms = env.missions()
        mx = env.max_steps()
        for i in range(n):
            assert ms[i] == ro[i]["mission"], (name, i)
            assert mx[i] == refs[i].max_steps, (name, i)
        for t in range(30):
            img = env.image.cpu().numpy()
            for i in range(n):
                assert np.array_equal(img[i], ro[i]["image"]), (name, i, t)
            a = rng.choice(7, size=n, p=[0.15, 0.15, 0.3, 0.12, 0.1, 0.15, 0.03]).astype(np.uint8)
            obs, reward, done, _ = env.step(torch.as_tensor(a, device=gpu))
            rew = reward.cpu().numpy()
            dn = done.cpu().numpy()
            for i in range(n):
                o, r, d, _ = refs[i].step(int(a[i]))
                assert np.float32(r) == rew[i] and bool(d) == bool(dn[i]), (name, i, t)
                if d:
                    o = refs[i].reset()
                ro[i] = o
        env.close()


@pytest.mark.gpu
def test_render_every_tile_vs_oracle(gpu):
    """k_render on synthetic encodings that contain every (type, colour, state) cell the engine can emit, in
    ordinary cells and in the agent's own cell, against the oracle's RGBImgPartialObsWrapper rasteriser."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from oracle import levels as olevels
    cells = [(0, 0, 0), (1, 0, 0), (2, 5, 0)]
    cells += [(t, c, 0) for t in (5, 6, 7) for c in range(6)]
    cells += [(4, c, s) for c in range(6) for s in range(3)]
    carried = [(1, 0, 0)] + [(t, c, 0) for t in (5, 6, 7) for c in range(6)]
    n = len(carried)
    rng = np.random.RandomState(3)
    img = np.zeros((n, 7, 7, 3), np.uint8)
    for e in range(n):
        for vi in range(7):
            for vj in range(7):
                img[e, vi, vj] = cells[(e * 49 + vi * 7 + vj + int(rng.randint(0, 3))) % len(cells)]
        img[e, 3, 6] = carried[e]
    env = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", n, device=gpu, pixel=True, seeds=0)
    env.reset()
    torch.cuda.synchronize()
    env.image.copy_(torch.as_tensor(img, device=gpu))
    pix = env.render_encoding().cpu().numpy()          # a hand-made encoding: the general entry, not the current-obs fast path
    ref_env = olevels.make_env("GoToLocal")
    seen = set()
    for e in range(n):
        assert np.array_equal(pix[e], ref_env.get_obs_render(img[e], tile_size=8)), e
        seen |= {tuple(x) for x in img[e].reshape(-1, 3)}
    assert seen >= set(cells)
    env.close()


class _ScriptedAgent(object):
    """Deterministic stand-in for ModelAgent: the action depends on the step count and the obs bytes only."""

    def __init__(self):
        self.t = 0

    def act_batch(self, many_obs):
        self.t += 1
        acts = [(int(o["image"].astype(np.int64).sum()) * 7 + self.t * 3 + k) % 7 for k, o in enumerate(many_obs)]
        return {"action": np.array(acts)}

    def analyze_feedback(self, reward, done):
        pass


@pytest.mark.gpu
def test_batch_evaluate_matches_reference_driver(gpu):
    """babyai_amd.evaluate.evaluate_policy (device-resident bookkeeping, a reference-style agent plugged in) vs the
    reference's evaluation loop (babyai/evaluate.py:85-140) restated over oracle envs with the ManyEnvs freeze-after-done
    rule (evaluate.py:73-81): identical logs, float64 returns included."""
    from babyai_amd.evaluate import evaluate_policy
    name, seed, episodes = "BabyAI-GoToLocal-v0", 31, 20
    logs = evaluate_policy(None, name, seed, episodes, device=gpu, agent=_ScriptedAgent())
    # reference driver restated over oracle envs
    envs = _oracle_envs("GoToLocal", [0] * episodes)
    agent = _ScriptedAgent()
    for e, s in zip(envs, range(seed, seed + episodes)):
        e.seed(s)
    many_obs = [e.reset() for e in envs]
    last = [None] * episodes
    done_flags = [False] * episodes
    num_frames = np.zeros(episodes, dtype="int64")
    returns = np.zeros(episodes)
    already = np.zeros(episodes, dtype=bool)
    cur = 0
    while (num_frames == 0).any():
        action = agent.act_batch(many_obs)["action"]
        results = [e.step(int(a)) if not d else last[k] for k, (e, a, d) in enumerate(zip(envs, action, done_flags))]
        done_flags = [r[2] for r in results]
        last = results
        many_obs = [r[0] for r in results]
        reward = np.array([r[1] for r in results])
        done = np.array(done_flags)
        just = done & ~already
        returns += reward * just
        cur += 1
        num_frames[just] = cur
        already[done] = True
    assert list(logs["num_frames_per_episode"]) == list(num_frames)
    assert list(logs["return_per_episode"]) == list(returns)          # float64, bit for bit
    assert list(logs["seed_per_episode"]) == list(range(seed, seed + episodes))
    # the wrapper with the reference's own signature and log shape (babyai/evaluate.py:85): same numbers, plus the
    # observations / actions of every episode up to its end
    from babyai_amd.evaluate import batch_evaluate
    ref_like = batch_evaluate(_ScriptedAgent(), name, seed, episodes, return_obss_actions=True, device=str(gpu))
    assert set(ref_like) == {"num_frames_per_episode", "return_per_episode", "observations_per_episode", "actions_per_episode", "seed_per_episode"}
    assert list(ref_like["num_frames_per_episode"]) == list(num_frames) and list(ref_like["return_per_episode"]) == list(returns)
    assert [len(a) for a in ref_like["actions_per_episode"]] == list(num_frames)
    assert [len(o) for o in ref_like["observations_per_episode"]] == list(num_frames)
    o0 = ref_like["observations_per_episode"][3][0]
    assert set(o0) == {"image", "direction", "mission"} and o0["image"].shape == (7, 7, 3) and isinstance(o0["mission"], str)
    # the reference rounds the episode count up to whole rounds of min(256, episodes) envs (evaluate.py:86,104)
    big = batch_evaluate(_ScriptedAgent(), name, seed, 300, device=str(gpu))
    assert len(big["return_per_episode"]) == 512 and big["seed_per_episode"] == list(range(seed, seed + 512))
    assert list(big["num_frames_per_episode"][:episodes]) == list(num_frames) and big["observations_per_episode"] == []


@pytest.mark.gpu
def test_evaluate_policy_tensor_path(gpu):
    """The tensor path of evaluate_policy (no host hop per frame, polling every 16 frames) gives the same logs as the
    per-frame agent path for the same decisions, on a batch that spans two chunks."""
    import torch
    from babyai_amd.evaluate import evaluate_policy

    def policy(obs, t):
        return ((obs["image"].to(torch.int64).sum(dim=(1, 2, 3)) * 13 // 5 + obs["direction"].to(torch.int64)) % 7).to(torch.uint8)

    class Agent(object):
        def act_batch(self, many_obs):
            return {"action": np.array([(int(o["image"].astype(np.int64).sum()) * 13 // 5 + o["direction"]) % 7 for o in many_obs])}

        def analyze_feedback(self, reward, done):
            pass

    a = evaluate_policy(policy, "BabyAI-PickupLoc-v0", 7, 300, device=gpu, chunk=256)
    b = evaluate_policy(None, "BabyAI-PickupLoc-v0", 7, 300, device=gpu, chunk=256, agent=Agent())
    assert a == b and len(a["return_per_episode"]) == 300 and min(a["num_frames_per_episode"]) >= 1


def _consume_mode(monkeypatch, mode):
    """How finished envs get their next level: "0" = k_consume launch, "1" = inside k_step by the stepping wave, "inplace" = the in-place
    state layout (the look-ahead slot IS the live record; bbai_engine.hip live_slot)."""
    if mode == "inplace":
        monkeypatch.setenv("BBAI_INPLACE", "1")
    else:
        monkeypatch.setenv("BBAI_INPLACE", "0")
        monkeypatch.setenv("BBAI_CONSUME_FUSED", mode)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["0", "1", "inplace"])
@pytest.mark.parametrize("period", ["1", "2", "8"])
def test_autoreset_with_interleaved_resets(gpu, period, fused, monkeypatch):
    """Auto-reset stepping with explicit reset() calls thrown in, for several refill periods (BBAI_LOOKAHEAD):
    the look-ahead ring must hand every env its stream's levels in order whatever the window phase -- with the finished
    envs consumed by a k_consume launch (0) and inside k_step by the wave that stepped them (1)."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    monkeypatch.setenv("BBAI_LOOKAHEAD", period)
    _consume_mode(monkeypatch, fused)
    n = 64
    env = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=9)     # tiny level: episodes of a few steps
    refs = _oracle_envs("GoToObjS4", [9 + i for i in range(n)])
    rng = np.random.RandomState(int(period))
    env.reset()
    ro = [e.reset() for e in refs]
    for t in range(160):
        if t in (5, 6, 40, 41, 42, 43, 97):           # explicit resets at assorted window phases
            env.reset()
            ro = [e.reset() for e in refs]
        img = env.image.cpu().numpy()
        for i in range(n):
            assert np.array_equal(img[i], ro[i]["image"]), (period, t, i)
        a = rng.choice([0, 1, 2], size=n).astype(np.uint8)
        env.step(torch.as_tensor(a, device=gpu))
        for i in range(n):
            o, r, d, _ = refs[i].step(int(a[i]))
            ro[i] = refs[i].reset() if d else o
    assert env.reset_count() > 7 * n
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level", ["BossLevel", "GoTo", "MiniBossLevel", "KeyCorridorS3R3", "PutNextS5N2Carrying", "Unlock"])
def test_device_generator_equals_host_build_exhaustively(gpu, level):
    """Every env of a batch: the records, hot state and programs produced by the wave-per-env device generator
    (LDS working set, lane-split MT twist / grid fill / flood fill) equal, byte for byte, what the same headers
    produce when compiled for the host with one lane -- three consecutive levels of 8192 seeds."""
    import ctypes
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.levels import make_cfg
    from hostsim_util import HostEnv
    n = 8192
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=70000)
    cfg = make_cfg(level)
    sims = [HostEnv(cfg, 70000 + i) for i in range(n)]
    for ep in range(3):
        env.reset()
        torch.cuda.synchronize()
        rec, hot, stale = env.export_state()
        img = env.image.cpu().numpy()
        for i, sim in enumerate(sims):
            first = sim.reset()
            assert np.array_equal(rec[i], sim.rec), (level, ep, i, "record")
            h = sim.hot.copy()
            g = hot[i].copy()
            g[15] = h[15] = 0           # ring slot index: engine bookkeeping only
            assert np.array_equal(g, h), (level, ep, i, "hot", g, h)
            assert stale[i] == sim.stale.value
            assert np.array_equal(img[i], first), (level, ep, i, "first obs")
    assert env.generator_failures() == 0
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level", ["BossLevel", "MiniBossLevel", "PutNextS5N2Carrying", "OpenDoorsOrderN4Debug"])
def test_device_step_equals_host_build_exhaustively(gpu, level):
    """Every env of a batch, every step: k_step (LDS-staged window, SoA verifier view, fused copy-out) + auto-reset
    against the host build of the same headers -- image, reward bits, done, hot state and stale set."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.levels import make_cfg
    from hostsim_util import HostEnv
    n, T = 3072, 70
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=5000)
    cfg = make_cfg(level)
    sims = [HostEnv(cfg, 5000 + i) for i in range(n)]
    env.reset()
    for s in sims:
        s.reset()
    rng = np.random.RandomState(12)
    for t in range(T):
        a = rng.choice(7, size=n, p=[0.14, 0.14, 0.3, 0.12, 0.1, 0.15, 0.05]).astype(np.uint8)
        env.step(torch.as_tensor(a, device=gpu))
        torch.cuda.synchronize()
        img = env.image.cpu().numpy()
        rew = env.reward.cpu().numpy()
        dn = env.done.cpu().numpy()
        _, hot, stale = env.export_state()
        for i, s in enumerate(sims):
            o, r, d = s.step(int(a[i]))
            assert r.view(np.uint32) == rew[i].view(np.uint32) and d == bool(dn[i]), (level, t, i)
            if d:
                o = s.reset()
            assert np.array_equal(img[i], o), (level, t, i)
            g, h = hot[i].copy(), s.hot.copy()
            g[15] = h[15] = 0
            assert np.array_equal(g, h) and stale[i] == s.stale.value, (level, t, i)
    env.close()


@pytest.mark.gpu
def test_single_env_protocol(gpu):
    """seed / reset / step on a batch of one, with the attributes the reference's callers read
    (babyai/evaluate.py:20-33: mission, step loop; scripts/manual_control.py: actions enum)."""
    from babyai_amd.vec_env import SingleEnv
    from oracle import levels as olevels
    env = SingleEnv("BabyAI-PickupLoc-v0", device=gpu, seed=5)
    ref = olevels.make_env("PickupLoc")
    ref.seed(5)
    rng = np.random.RandomState(2)
    for ep in range(4):
        o, ro = env.reset(), ref.reset()
        assert o["mission"] == ro["mission"] == env.mission and env.max_steps == ref.max_steps
        assert np.array_equal(o["image"], ro["image"]) and o["direction"] == ro["direction"]
        while True:
            a = int(rng.randint(0, 7))
            (o, r, d, info), (ro, rr, rd, _) = env.step(a), ref.step(a)
            assert np.array_equal(o["image"], ro["image"]) and np.float32(rr) == np.float32(r) and d == bool(rd) and info == {}
            assert env.step_count == ref.step_count
            # unwrapped.grid / agent pose: the full-state view levelgen.py:531-537 compares between same-seed envs
            assert np.array_equal(env.unwrapped.grid.encode(), ref.grid.encode())
            assert env.agent_pos == tuple(ref.agent_pos) and env.agent_dir == ref.agent_dir
            if d:
                break
    # the expert through the single-env protocol: solves the episode, like `Bot(env).replan()` in scripts/enjoy.py
    env.seed(5)
    env.reset()
    reward, done = 0.0, False
    while not done:
        _, reward, done, _ = env.step(env.bot_action())
    assert reward > 0
    twin = SingleEnv("BabyAI-PickupLoc-v0", device=gpu, seed=5)
    env.seed(5)
    env.reset(), twin.reset()
    assert env.unwrapped.grid == twin.unwrapped.grid and env.surface == twin.surface
    twin.close()
    assert env.actions.toggle == 5 and env.action_space.n == 7
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level", ["GoToObjS4", "PickupLoc"])
def test_device_rollout_engine_equals_oracle_rollout(gpu, level):
    """babyai_amd.rollout.DeviceRollout on the engine vs the same collector over oracle envs on CPU tensors
    (which tests/test_rollout.py pins to the reference's BaseAlgo.collect_experiences, base.py:131-260):
    every experience field and the episode logs must agree exactly across rollouts with auto-resets."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.rollout import DeviceRollout
    from rollout_util import OracleTensorEnv, ToyACModel
    n, T = 48, 32
    seeds = [700 + i for i in range(n)]
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=seeds)
    dev = DeviceRollout(env, ToyACModel(), T, 0.99, 0.95, reward_scale=20.0)
    cpu = DeviceRollout(OracleTensorEnv(level, seeds), ToyACModel(), T, 0.99, 0.95, reward_scale=20.0)
    done_total = 0
    for it in range(3):
        e1, l1 = dev.collect_experiences()
        e2, l2 = cpu.collect_experiences()
        for f in ("memory", "mask", "action", "value", "reward", "advantage", "returnn", "log_prob"):
            a, b = e1[f].cpu(), e2[f]
            assert a.shape == b.shape and a.dtype == b.dtype, f
            assert torch.equal(a, b), (level, it, f)
        assert torch.equal(e1.obs.image.cpu(), e2.obs.image) and torch.equal(e1.obs.instr.cpu(), e2.obs.instr)
        assert l1 == l2
        done_total += l1["episodes_done"]
    assert done_total >= n // 2
    env.close()


@pytest.mark.gpu
def test_gae_kernel_equals_the_reference_loop(gpu):
    """bbai_gae (one reverse scan per env, env-major buffers) vs the reference's loop (base.py:196-202) in torch float32
    ops on [T, P] tensors: every advantage / return bit for bit, including masked episode boundaries."""
    import torch
    from babyai_amd.rollout import gae_env_major
    g = torch.Generator(device="cpu")
    g.manual_seed(3)
    P, T, d, lam = 3001, 40, 0.99, 0.95
    rewards = (torch.rand(T, P, generator=g) * 20 * (torch.rand(T, P, generator=g) < 0.1)).to(torch.float32)
    values = torch.randn(T, P, generator=g)
    masks = (torch.rand(T, P, generator=g) > 0.08).to(torch.float32)
    last_mask = (torch.rand(P, generator=g) > 0.08).to(torch.float32)
    last_value = torch.randn(P, generator=g)
    adv = torch.zeros(T, P)
    for i in reversed(range(T)):                   # base.py:196-202, verbatim semantics
        next_mask = masks[i + 1] if i < T - 1 else last_mask
        next_value = values[i + 1] if i < T - 1 else last_value
        next_advantage = adv[i + 1] if i < T - 1 else 0
        delta = rewards[i] + d * next_value * next_mask - values[i]
        adv[i] = delta + d * lam * next_advantage * next_mask
    dev = lambda x: x.t().contiguous().to(gpu)
    a, r = torch.zeros(P, T, device=gpu), torch.zeros(P, T, device=gpu)
    gae_env_major(dev(rewards), dev(values), dev(masks), last_mask.to(gpu), last_value.to(gpu), d, lam, a, r)
    torch.cuda.synchronize()
    assert torch.equal(a.cpu(), adv.t()) and torch.equal(r.cpu(), (values + adv).t())


@pytest.mark.gpu
def test_state_errors_are_reported_not_undefined(gpu):
    """Protocol misuse comes back as an error code + message through the C ABI, never as undefined device work."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv, EngineError
    env = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", 8, device=gpu)          # not seeded
    with pytest.raises(EngineError, match="reset before seed"):
        env.reset()
    with pytest.raises(EngineError, match="step before reset"):
        env.step(torch.zeros(8, dtype=torch.uint8, device=gpu))
    donor = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", 8, device=gpu, seeds=1)
    donor.reset()
    env.import_state(*donor.export_state())                               # live, but without a level stream
    with pytest.raises(EngineError, match="auto-reset step before seed"):
        env.step(torch.zeros(8, dtype=torch.uint8, device=gpu))
    env.auto_reset = False                                                # ManyEnvs protocol needs no stream
    donor.auto_reset = False
    a = torch.full((8,), 2, dtype=torch.uint8, device=gpu)
    o1, o2 = env.step(a), donor.step(a)
    assert torch.equal(o1[0]["image"], o2[0]["image"]) and torch.equal(o1[1], o2[1]) and torch.equal(o1[2], o2[2])
    with pytest.raises(ValueError):
        env.step(torch.zeros(7, dtype=torch.uint8, device=gpu))
    # checkpoints: a blob only loads into a handle of the same level, batch size and look-ahead depth
    blob = donor.save_checkpoint()
    other = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", 16, device=gpu)
    with pytest.raises(EngineError, match="different level / batch size"):
        other.load_checkpoint(blob)
    other.close()
    other = BatchedBabyAIEnv("BabyAI-PickupLoc-v0", 8, device=gpu)
    with pytest.raises(EngineError, match="different level / batch size"):
        other.load_checkpoint(blob)
    with pytest.raises(EngineError, match="not a bbai checkpoint"):
        other.load_checkpoint(np.zeros(blob.size, np.uint8))
    with pytest.raises(EngineError):
        other.load_checkpoint(blob[:100])
    other.close()
    # the parity tap refuses misaligned pixel rows instead of issuing misaligned 16-byte accesses
    pix = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", 8, device=gpu, seeds=2, pixel=True)
    pix.reset()
    img, dr = torch.zeros((4, 7, 7, 3), dtype=torch.uint8, device=gpu), torch.zeros(4, dtype=torch.uint8, device=gpu)
    rw, dn = torch.zeros(4, dtype=torch.float64, device=gpu), torch.zeros(4, dtype=torch.uint8, device=gpu)
    raw = torch.zeros(2 * 9408 + 1, dtype=torch.uint8, device=gpu)
    with pytest.raises(EngineError, match="misaligned"):
        pix.tap(img, dr, rw, dn, raw[1:].view(2, 56, 56, 3))
    pix.tap(img, dr, rw, dn, raw[:-1].view(2, 56, 56, 3))
    torch.cuda.synchronize()
    assert torch.equal(img, pix.image[:4]) and torch.equal(raw[:-1].view(2, 56, 56, 3), pix.pixels[:2])
    pix.close()
    env.close(); donor.close()


@pytest.mark.gpu
def test_c1_plumbing_config_on_the_engine(gpu):
    """BASELINE.json configs[0] through the single-env protocol: same digest as the reference (tests/test_oracle_golden.py)."""
    import hashlib
    from babyai_amd.vec_env import SingleEnv
    from test_oracle_golden import C1_DIGEST
    env = SingleEnv("BabyAI-GoToRedBall-v0", device=gpu, seed=0)
    env.reset()
    h = hashlib.sha256()
    for a in np.random.RandomState(0).randint(0, 7, size=10000):
        obs, reward, done, _ = env.step(int(a))
        h.update(obs["image"].tobytes())
        h.update(np.float32(reward).tobytes())
        h.update(bytes([int(obs["direction"]), int(done)]))
        if done:
            env.reset()
    env.close()
    assert h.hexdigest() == C1_DIGEST


BOT_LEVELS = ["GoToLocal", "PickupLoc", "PutNextLocal", "GoTo", "Open", "Unlock", "UnblockPickup", "GoToImpUnlock", "PickupDist",
              "KeyCorridorS4R3", "BlockedUnlockPickup", "UnlockToUnlock", "SynthSeq", "MiniBossLevel", "BossLevel",
              "TestLotsOfBlockers", "MoveTwoAcrossS8N9", "KeyInBox"]


@pytest.mark.gpu
@pytest.mark.parametrize("level", BOT_LEVELS)
@pytest.mark.parametrize("mode", ["pure", "advised"])
def test_bot_decisions_on_the_engine_match_reference(gpu, level, mode):
    """bbai_bot_act (k_bot) against the decisions recorded from the reference's babyai/bot.py (tests/golden/bot/,
    tools/gen_golden_bot.py): every suggestion of every env, through auto-resets, 255 where the reference bot raised."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "bot", level + ".npz")) as f:
        g = {k: f[k] for k in f.files}
    suggest, action, done = g[mode + "_suggest"], g[mode + "_action"], g[mode + "_done"]
    n_steps, n = suggest.shape
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=int(g["seed_base"]))
    env.reset()
    prev = None
    for t in range(n_steps):
        got = env.bot_actions(prev).cpu().numpy().astype(np.int16)
        got[got == 255] = -1
        assert np.array_equal(got, suggest[t].astype(np.int16)), (level, mode, t, got, suggest[t])
        prev = torch.as_tensor(action[t], device=gpu)
        _, _, d, _ = env.step(prev)
        assert np.array_equal(d.cpu().numpy(), done[t]), (level, mode, t)
    stats = env.bot_stats()
    assert stats["capacity"] == 0 or level == "UnlockToUnlock"
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level,n,steps", [("BossLevel", 384, 260), ("MiniBossLevel", 512, 200), ("PutNextS7N4", 256, 120)])
def test_bot_device_equals_host_build(gpu, level, n, steps):
    """Many more seeds than the golden fixtures hold: the device expert against the host build of the same header
    (which tests/test_hostsim_bot.py pins to the reference), decision for decision, with the bot driving."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.levels import make_cfg
    from hostsim_util import HostBot, HostEnv
    base = 31000
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=base)
    env.reset()
    hosts = [HostEnv(make_cfg(level), base + i) for i in range(n)]
    bots = []
    for h in hosts:
        h.reset()
        bots.append(HostBot(h))
    first = [True] * n
    rng = np.random.RandomState(5)
    prev = None
    last = [None] * n
    episodes = successes = 0
    for t in range(steps):
        got = env.bot_actions(prev).cpu().numpy()
        act = np.zeros(n, np.uint8)
        for i in range(n):
            a = bots[i].decide(first[i], last[i])
            first[i] = False
            assert got[i] == (255 if a is None else a), (level, i, t, got[i], a)
            if a is None or rng.rand() < 0.05:
                a = int(rng.randint(0, 7))
            act[i] = a
        prev = torch.as_tensor(act, device=gpu)
        _, r, d, _ = env.step(prev)
        d = d.cpu().numpy()
        r = r.cpu().numpy()
        for i in range(n):
            _, hr, hd = hosts[i].step(int(act[i]))
            assert bool(hd) == bool(d[i]) and np.float32(hr) == r[i]
            last[i] = int(act[i])
            if hd:
                episodes += 1
                successes += hr > 0
                hosts[i].reset()
                first[i], last[i] = True, None
    assert episodes > n // 4 and successes > 0.8 * episodes, (episodes, successes)
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["0", "1", "inplace"])
@pytest.mark.parametrize("lookahead", [None, "2"])
def test_very_short_episodes_every_window_tick(gpu, lookahead, fused, monkeypatch):
    """Expert-driven GoToObjS4: episodes of 1-4 steps, so roughly a third of the batch finishes on EVERY step and an env
    finishes several times within one look-ahead window (regression: the window's refill list must hold one entry
    per (tick, finished env), not one per env).  Every env against the host build, bot and env in lockstep."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.levels import make_cfg
    from hostsim_util import HostBot, HostEnv
    if lookahead:
        monkeypatch.setenv("BBAI_LOOKAHEAD", lookahead)
    _consume_mode(monkeypatch, fused)
    level, n, base = "GoToObjS4", 768, 52000
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=base)
    env.reset()
    hosts = [HostEnv(make_cfg(level), base + i) for i in range(n)]
    bots = []
    for h in hosts:
        h.reset()
        bots.append(HostBot(h))
    first = [True] * n
    resets = 0
    for t in range(72):
        got = env.bot_actions(None).cpu().numpy()
        for i in range(n):
            assert got[i] == bots[i].decide(first[i], None), (i, t)
            first[i] = False
        obs, r, d, _ = env.step(torch.as_tensor(got, device=gpu))
        img, r, d = obs["image"].cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        for i in range(n):
            himg, hr, hd = hosts[i].step(int(got[i]))
            assert bool(hd) == bool(d[i]) and np.float32(hr) == r[i], (i, t)
            if hd:
                himg = hosts[i].reset()
                first[i] = True
                resets += 1
            assert np.array_equal(img[i], himg), (i, t)
    assert resets > 72 * n // 5
    assert env.generator_failures() == 0
    env.close()


@pytest.mark.gpu
def test_per_env_reset_command(gpu):
    """Action 7 = env.reset() for that env only (a ParallelEnv worker's `reset` command): done, reward 0, and the
    next observation is the first one of the env's next level -- against oracle envs reset individually."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    n = 96
    env = BatchedBabyAIEnv("BabyAI-PickupLoc-v0", n, device=gpu, seeds=77)
    refs = _oracle_envs("PickupLoc", [77 + i for i in range(n)])
    env.reset()
    ro = [e.reset() for e in refs]
    rng = np.random.RandomState(3)
    for t in range(120):
        a = rng.randint(0, 7, size=n).astype(np.uint8)
        a[rng.rand(n) < 0.07] = env.RESET_ENV
        obs, r, d, _ = env.step(torch.as_tensor(a, device=gpu))
        img, r, d = obs["image"].cpu().numpy(), r.cpu().numpy(), d.cpu().numpy()
        for i in range(n):
            if a[i] == env.RESET_ENV:
                o, rr, dd = refs[i].reset(), 0.0, True
            else:
                o, rr, dd, _ = refs[i].step(int(a[i]))
                if dd:
                    o = refs[i].reset()
            assert np.array_equal(img[i], o["image"]) and np.float32(rr) == r[i] and bool(dd) == bool(d[i]), (i, t)
            assert obs["mission"][i] == o["mission"]
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("rollout", [True, False, None])
@pytest.mark.parametrize("level", ["GoToLocal", "PutNextLocal", "PickupDist", "BossLevel", "UnlockToUnlock"])
def test_generate_demos_matches_reference_script(gpu, level, rollout):
    """babyai_amd.demos.generate_demos vs demonstrations made by the reference's own loop (scripts/make_agent_demos.py:
    71-137 with BotAgent; tools/gen_golden_bot.py demos): same missions, actions, directions and images, including the
    streams where the reference bot crashed or failed first and the script moved on to the stream's next level."""
    import hashlib
    from babyai_amd.demos import generate_demos
    with np.load(os.path.join(os.path.dirname(__file__), "golden", "demos", "bot_demos.npz")) as f:
        g = {k[len(level) + 1:]: f[k] for k in f.files if k.startswith(level + "_")}
    n = len(g["length"])
    demos = generate_demos("BabyAI-%s-v0" % level, n, int(g["seed"]), device=gpu, batch=32, rollout=rollout)
    ends = np.cumsum(g["length"])
    for k, (mission, images, directions, actions) in enumerate(demos):
        lo, hi = ends[k] - g["length"][k], ends[k]
        assert mission == str(g["mission"][k]), (level, k)
        assert list(actions) == list(g["actions"][lo:hi]), (level, k)
        assert list(directions) == list(g["directions"][lo:hi]), (level, k)
        assert images.dtype == np.uint8 and images.shape == (hi - lo, 7, 7, 3)
        assert hashlib.sha256(images.tobytes()).hexdigest() == str(g["image_sha"][k]), (level, k)


@pytest.mark.gpu
@pytest.mark.parametrize("level,n,steps", [("BossLevel", 2048, 200), ("GoToLocal", 4096, 120), ("KeyCorridorS6R3", 1024, 200),
                                           ("PutNextS7N4", 1024, 120), ("UnlockToUnlock", 512, 200), ("SynthSeq", 1024, 160)])
def test_lane_group_expert_decides_like_the_lane_per_env_expert(gpu, level, n, steps):
    """k_botg (option bot_group = 16: one 16-lane group per env, search 1 in LDS) against k_bot (lane = env) on twin handles, the
    lane-per-env expert driving both with 5 % random actions: every suggestion, every give-up, the same statistics.  (The group form
    of the header is pinned to the reference's bot by tests/test_hostsim_bot.py's fiber emulation on all 105 levels.)"""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    a = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=77000)
    b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=77000)
    b.set_option("bot_group", 16)
    assert b.get_option("bot_group") == 16 and a.get_option("bot_group") == 0
    a.reset()
    b.reset()
    try:
        b.bot_actions()
    except Exception as exc:                 # the shipped library carries no k_botg (an experiment build: -DBBAI_BOT_GROUP_BUILD=1)
        if "built without the lane-group expert" in str(exc):
            a.close()
            b.close()
            pytest.skip("k_botg is an experiment build, not part of the shipped library")
        raise
    b.close()
    b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=77000)
    b.set_option("bot_group", 16)
    b.reset()
    gen = torch.Generator(device=gpu).manual_seed(3)
    prev = None
    for t in range(steps):
        ga, gb = a.bot_actions(prev), b.bot_actions(prev)
        assert torch.equal(ga, gb), (level, t, (ga != gb).nonzero()[:4].tolist())
        rnd = torch.randint(0, 7, (n,), device=gpu, dtype=torch.uint8, generator=gen)
        noisy = torch.rand((n,), device=gpu, generator=gen) < 0.05
        prev = torch.where(noisy | (ga == 255), rnd, ga)
        _, _, da, _ = a.step(prev)
        _, _, db, _ = b.step(prev)
        assert torch.equal(da, db)
    assert a.bot_stats() == b.bot_stats()
    a.close()
    b.close()


@pytest.mark.gpu
def test_bot_device_equals_host_build_on_every_level(gpu):
    """All 105 levels: k_bot against the host build of bbai_bot.hpp (which tests/test_hostsim_bot.py pins to the
    reference on every level), bot-driven with 6 % random actions so the undo logic runs too."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.levels import LEVELS, make_cfg
    from hostsim_util import HostBot, HostEnv
    n, steps = 24, 48
    rng = np.random.RandomState(11)
    for level in sorted(LEVELS):
        env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=61000)
        env.reset()
        hosts = [HostEnv(make_cfg(level), 61000 + i) for i in range(n)]
        bots = []
        for h in hosts:
            h.reset()
            bots.append(HostBot(h))
        first, last, prev = [True] * n, [None] * n, None
        for t in range(steps):
            got = env.bot_actions(prev).cpu().numpy()
            act = np.zeros(n, np.uint8)
            for i in range(n):
                a = bots[i].decide(first[i], last[i])
                first[i] = False
                assert got[i] == (255 if a is None else a), (level, i, t, got[i], a)
                act[i] = a if (a is not None and rng.rand() > 0.06) else rng.randint(0, 7)
            prev = torch.as_tensor(act, device=gpu)
            _, _, d, _ = env.step(prev)
            d = d.cpu().numpy()
            for i in range(n):
                _, _, hd = hosts[i].step(int(act[i]))
                assert bool(hd) == bool(d[i]), (level, i, t)
                last[i] = int(act[i])
                if hd:
                    hosts[i].reset()
                    first[i], last[i] = True, None
        # the port's one representable divergence (a 48-entry subgoal stack) is visible and agrees between the two builds
        host_cap = sum(1 for b in bots if b.dead_reason == 2)
        stats = env.bot_stats()
        assert stats["capacity"] >= host_cap and (stats["capacity"] == 0 or level in ("UnlockToUnlock",)), (level, stats, host_cap)
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["0", "1"])
@pytest.mark.parametrize("level,lookahead,use_bot", [("GoToObjS4", "2", False), ("PickupLoc", None, False), ("GoToObjS4", None, True),
                                                     ("MiniBossLevel", "4", True)])
def test_checkpoint_resume_is_bit_identical(gpu, level, lookahead, use_bot, layout, monkeypatch):
    """bbai_checkpoint_save mid-rollout (at a tick that is not a window boundary), bbai_checkpoint_load into a FRESH
    handle, and both continue through many auto-resets: every output of every later step is identical, the expert's
    decisions included -- i.e. the blob carries the RNG streams, the look-ahead ring, the window bookkeeping and the
    expert's plans, not just the live grids.  Both state layouts (BBAI_INPLACE)."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    monkeypatch.setenv("BBAI_INPLACE", layout)
    if lookahead:
        monkeypatch.setenv("BBAI_LOOKAHEAD", lookahead)
    n = 512
    a = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=4242)
    a.reset()
    gen = torch.Generator(device=gpu)
    gen.manual_seed(5)
    acts = torch.randint(0, 7, (200, n), dtype=torch.uint8, device=gpu, generator=gen)

    def act(env, t):
        if not use_bot:
            return acts[t]
        noise = acts[t] == 6                  # the expert, with 1/7 of the envs acting randomly
        return torch.where(noise, acts[(t + 1) % 200], env.bot_actions(None))

    prev = None
    for t in range(37):                       # 37 is not a multiple of any refill period
        prev = act(a, t).clone()
        a.step(prev)
    blob = a.save_checkpoint()
    b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu)      # never seeded: everything comes from the blob
    b.load_checkpoint(blob)
    resets0 = a.reset_count()
    assert b.reset_count() == resets0
    for t in range(37, 200):
        xa, xb = act(a, t).clone(), act(b, t).clone()
        assert torch.equal(xa, xb), t
        oa, ra, da, _ = a.step(xa)
        ob, rb, db, _ = b.step(xb)
        torch.cuda.synchronize()
        assert torch.equal(a.image, b.image) and torch.equal(a.direction, b.direction), t
        assert torch.equal(a.reward64, b.reward64) and torch.equal(da, db), t
        if t % 40 == 0:
            assert a.missions() == b.missions()
    assert a.reset_count() - resets0 > 2 * n // 3 and a.reset_count() == b.reset_count()
    assert a.generator_failures() == 0
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["0", "1"])
@pytest.mark.parametrize("save_period,load_period", [("8", "2"), ("2", None), (None, "4")])
def test_checkpoint_loads_into_a_handle_with_another_lookahead_period(gpu, save_period, load_period, layout, monkeypatch):
    """The look-ahead period is chosen from the memory that is free at bbai_create (or pinned by BBAI_LOOKAHEAD): two
    identically configured handles may differ in it.  The ring travels in the blob, so the loading handle takes the
    blob's shape and continues bit-identically.  Both state layouts; a blob of the other layout is refused."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv, EngineError
    n = 300
    monkeypatch.setenv("BBAI_INPLACE", layout)
    if save_period:
        monkeypatch.setenv("BBAI_LOOKAHEAD", save_period)
    a = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=77)
    monkeypatch.delenv("BBAI_LOOKAHEAD", raising=False)
    a.reset()
    gen = torch.Generator(device=gpu)
    gen.manual_seed(6)
    acts = torch.randint(0, 3, (120, n), dtype=torch.uint8, device=gpu, generator=gen)
    for t in range(21):
        a.step(acts[t])
    blob = a.save_checkpoint()
    if load_period:
        monkeypatch.setenv("BBAI_LOOKAHEAD", load_period)
    b = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=1)        # seeded with something else, other ring shape
    b.reset()
    b.load_checkpoint(blob)
    for t in range(21, 120):
        a.step(acts[t])
        b.step(acts[t])
        assert torch.equal(a.image, b.image) and torch.equal(a.reward64, b.reward64) and torch.equal(a.done, b.done), t
    assert a.reset_count() == b.reset_count() and a.reset_count() > 5 * n
    assert len(b.save_checkpoint()) == len(blob)
    monkeypatch.setenv("BBAI_INPLACE", "1" if layout == "0" else "0")
    c = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=1)
    with pytest.raises(EngineError, match="state layout"):
        c.load_checkpoint(blob)
    c.close()
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lookahead,first_leg", [("2", 3), ("2", 5), ("4", 6), ("4", 9), ("8", 13), ("8", 21)])
def test_reseed_in_the_middle_of_a_window(gpu, lookahead, first_leg, monkeypatch):
    """seed() on an engine that stopped `first_leg` auto-reset steps into a run -- not a multiple of the refill period, so
    the window in progress was using buffer 1 or 2 of the window bookkeeping -- must start a clean run: three and more
    windows of the new run against the oracle (round-1 advisor finding: only buffer 0 used to be cleared)."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    monkeypatch.setenv("BBAI_LOOKAHEAD", lookahead)
    n = 96
    env = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=1)
    env.reset()
    rng = np.random.RandomState(first_leg)
    for t in range(first_leg):
        env.step(torch.as_tensor(rng.choice([0, 1, 2], size=n).astype(np.uint8), device=gpu))
    assert env.reset_count() > n               # envs did finish in the first leg
    env.seed(500)
    refs = _oracle_envs("GoToObjS4", [500 + i for i in range(n)])
    env.reset()
    ro = [e.reset() for e in refs]
    for t in range(5 * int(lookahead) + 7):
        img = env.image.cpu().numpy()
        for i in range(n):
            assert np.array_equal(img[i], ro[i]["image"]), (lookahead, first_leg, t, i)
        a = rng.choice([0, 1, 2], size=n).astype(np.uint8)
        env.step(torch.as_tensor(a, device=gpu))
        for i in range(n):
            o, r, d, _ = refs[i].step(int(a[i]))
            ro[i] = refs[i].reset() if d else o
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("call_events", [False, True])
def test_caller_may_change_streams_between_calls(gpu, call_events):
    """A handle follows one caller stream at a time; when the caller comes back on another stream the engine orders it
    behind the work it enqueued before (include/bbai.h).  Alternating streams per step, with no synchronisation by the
    caller, must give the single-stream result.  With bbai_set_call_events the previous stream is never touched again:
    there the caller creates a FRESH stream for every step and drops the old one."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    n = 4096
    a = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=9, pixel=True)
    b = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=9, pixel=True)
    if call_events:
        b.set_call_events(True)
    a.reset()
    b.reset()
    acts = torch.randint(0, 7, (96, n), dtype=torch.uint8, device=gpu)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)]
    for t in range(96):
        a.step(acts[t])
        if call_events:
            s = torch.cuda.Stream(device=gpu)        # a new stream every step; the previous one is released
            s.wait_stream(torch.cuda.current_stream(gpu))      # (acts was produced on the default stream)
            with torch.cuda.stream(s):
                b.step(acts[t])
            streams[t & 1] = s
        else:
            with torch.cuda.stream(streams[t & 1]):
                b.step(acts[t])
    for s in streams:
        s.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(a.image, b.image) and torch.equal(a.pixels, b.pixels) and torch.equal(a.reward64, b.reward64)
    assert a.reset_count() == b.reset_count() > n
    a.close()
    b.close()


@pytest.mark.gpu
def test_stored_observations_keep_their_missions(gpu):
    """The reference's collectors keep the obs of every frame and read their missions at the end of the rollout
    (base.py:146-147,207-232): an obs list handed out by the adapter must keep ITS episode's mission after the env has
    auto-reset into another one (round-1 advisor finding), and a raw tensor-level view must refuse instead of lying."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv, EngineError
    from babyai_amd.vec_env import BatchedParallelEnv
    n = 48
    venv = BatchedParallelEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=21)
    refs = _oracle_envs("GoToObjS4", [21 + i for i in range(n)])
    kept = [(venv.reset(), [e.reset()["mission"] for e in refs])]
    rng = np.random.RandomState(2)
    for t in range(40):
        a = rng.choice([0, 1, 2], size=n)
        obs, _, _, _ = venv.step(a)
        want = []
        for i, e in enumerate(refs):
            o, r, d, _ = e.step(int(a[i]))
            if d:
                o = e.reset()
            want.append(o["mission"])
        kept.append((obs, want))
    assert venv.engine.reset_count() > 3 * n
    changed = 0
    for obs, want in kept:                     # read only now, many resets later
        got = [obs[i]["mission"] for i in range(n)]
        assert got == want
        changed += sum(g != w for g, w in zip(got, kept[-1][1]))
    assert changed > n                         # the missions really did change over the rollout
    venv.close()
    env = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", 8, device=gpu, seeds=1)
    old = env.reset()
    env.step(torch.zeros(8, dtype=torch.uint8, device=gpu))
    with pytest.raises(EngineError):
        old["mission"][0]
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("group,tpb", [("2", "512"), ("8", "1024"), ("4", "256"), ("2", "1024"), ("8", "512")])
def test_render_launch_shapes_are_byte_identical(gpu, group, tpb, monkeypatch):
    """bbai_render picks (envs per one-shot block, threads per block) by batch size -- (2, 512) or (8, 1024); these and the
    other instantiated shapes (forced here with BBAI_RENDER_GROUP / BBAI_RENDER_TPB on a batch that is not a multiple of
    any group size) must produce the reference wrapper's pixels."""
    monkeypatch.setenv("BBAI_RENDER_TPB", tpb)
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from gym_minigrid.wrappers import RGBImgPartialObsWrapper
    monkeypatch.setenv("BBAI_RENDER_GROUP", group)
    n = 1003
    env = BatchedBabyAIEnv("BabyAI-PickupLoc-v0", n, device=gpu, pixel=True, seeds=40)
    spots = [0, 1, 2, 3, 7, 8, 500, 501, n - 9, n - 8, n - 3, n - 2, n - 1]
    refs = _oracle_envs("PickupLoc", [40 + i for i in spots])
    wr = [RGBImgPartialObsWrapper(e) for e in refs]
    obs = env.reset()
    ro = [w.reset() for w in wr]
    rng = np.random.RandomState(int(group))
    for t in range(40):
        pix = obs["image"][spots].cpu().numpy()
        for k in range(len(spots)):
            assert np.array_equal(pix[k], ro[k]["image"]), (group, spots[k], t)
        a = rng.randint(0, 7, size=n).astype(np.uint8)
        obs, _, _, _ = env.step(torch.as_tensor(a, device=gpu))
        for k, i in enumerate(spots):
            o, r, d, _ = wr[k].step(int(a[i]))
            ro[k] = wr[k].reset() if d else o
    env.close()


@pytest.mark.gpu
def test_unknown_actions_are_defined_and_can_be_rejected(gpu):
    """The reference asserts on an action outside MiniGridEnv.Actions (gym_minigrid step: `assert False, "unknown action"`,
    reached from levelgen.py:50).  The kernel cannot raise: include/bbai.h DEFINES bytes 8..255 as the `done` action; the
    binding can check first (validate_actions / step(validate=True): AssertionError like the reference), and the
    reference-protocol adapters always do."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.vec_env import BatchedParallelEnv, SingleEnv
    n = 512
    a = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", n, device=gpu, seeds=31)
    b = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", n, device=gpu, seeds=31)
    a.reset()
    b.reset()
    gen = torch.Generator(device=gpu)
    gen.manual_seed(1)
    for t in range(150):
        acts = torch.randint(0, 7, (n,), dtype=torch.uint8, device=gpu, generator=gen)
        junk = torch.randint(8, 256, (n,), dtype=torch.int64, device=gpu, generator=gen).to(torch.uint8)
        swap = acts == 6
        a.step(acts)
        b.step(torch.where(swap, junk, acts))                 # every `done` replaced by an arbitrary byte 8..255
        assert torch.equal(a.image, b.image) and torch.equal(a.reward64, b.reward64) and torch.equal(a.done, b.done)
    assert a.reset_count() == b.reset_count() > n
    # a WIDER integer tensor must not wrap into a valid action on its way to a byte: 263 (= 7 mod 256: the per-env reset),
    # 256 (= 0: turn left), -1, 2**40 + 2 all behave like `done`
    wide = torch.tensor([263, 256, -1, 2 ** 40 + 2], dtype=torch.int64, device=gpu).repeat(n // 4)
    a.step(torch.full((n,), 6, dtype=torch.uint8, device=gpu))
    b.step(wide)
    assert torch.equal(a.image, b.image) and torch.equal(a.reward64, b.reward64) and torch.equal(a.done, b.done)
    with pytest.raises(TypeError):
        b.step(torch.zeros(n, dtype=torch.float32, device=gpu))
    bad = torch.zeros(n, dtype=torch.uint8, device=gpu)
    bad[n // 2] = 9
    with pytest.raises(AssertionError, match="unknown action"):
        a.step(bad, validate=True)
    ok = torch.full((n,), a.RESET_ENV, dtype=torch.uint8, device=gpu)
    a.step(ok, validate=True)                                  # the per-env reset command is known
    with pytest.raises(AssertionError, match="unknown action"):
        a.step(bad.cpu().numpy())                              # host data is checked for free, always
    c = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", 64, device=gpu, seeds=1, validate_actions=True)
    c.reset()
    with pytest.raises(AssertionError, match="unknown action"):
        c.step(torch.full((64,), 200, dtype=torch.uint8, device=gpu))
    for e in (a, b, c):
        e.close()
    p = BatchedParallelEnv("BabyAI-GoToLocal-v0", 4, device=gpu, seeds=[1, 2, 3, 4])
    p.reset()
    with pytest.raises(AssertionError, match="unknown action"):
        p.step(np.array([0, 1, 7, 2]))                        # 7 is the engine's own command, not a MiniGrid action
    p.close()
    s = SingleEnv("BabyAI-GoToLocal-v0", device=gpu, seed=3)
    s.reset()
    with pytest.raises(AssertionError, match="unknown action"):
        s.step(7)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [5000, 64, 37])
def test_frozen_envs_re_emit_their_observation_through_the_cell_stream(gpu, n):
    """ManyEnvs mode (babyai/evaluate.py:73-81): a finished env is frozen and keeps re-emitting its last observation.  k_step
    stages every block's observations as 49-byte cell rows and expands them on the way out (bbai_step.hpp); a frozen lane
    re-derives its cells from the caller-kept encoding.  Every frozen env's image, direction and pixels must stay byte for
    byte what they were at its terminal step while its neighbours keep moving -- ragged last block included."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    env = BatchedBabyAIEnv("BabyAI-PickupLoc-v0", n, device=gpu, pixel=True, seeds=77, auto_reset=False)
    env.reset()
    gen = torch.Generator(device=gpu)
    gen.manual_seed(3)
    frozen = torch.zeros(n, dtype=torch.bool, device=gpu)
    keep_img, keep_pix, keep_dir = env.image.clone(), env.pixels.clone(), env.direction.clone()
    for t in range(140):
        env.step(torch.randint(0, 7, (n,), dtype=torch.uint8, device=gpu, generator=gen))
        assert torch.equal(env.image[frozen], keep_img[frozen]) and torch.equal(env.pixels[frozen], keep_pix[frozen]), t
        assert torch.equal(env.direction[frozen], keep_dir[frozen]), t
        newly = env.done.bool() & ~frozen
        keep_img[newly], keep_pix[newly], keep_dir[newly] = env.image[newly], env.pixels[newly], env.direction[newly]
        frozen |= newly
        assert torch.equal(env.done.bool(), frozen), t             # a frozen env keeps reporting done
    assert int(frozen.sum()) > n // 2                             # most envs sat frozen for a while
    env.reset()
    env.step(torch.randint(0, 7, (n,), dtype=torch.uint8, device=gpu, generator=gen))
    assert int(env.done.sum()) < n // 4                           # a fresh episode everywhere
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level,n,steps", [("GoToLocal", 300, 96), ("BossLevel", 129, 160), ("UnlockToUnlock", 64, 200)])
def test_bot_rollout_equals_the_stepwise_loop(gpu, level, n, steps):
    """bbai_bot_rollout (T expert decisions + auto-reset steps per call, history written by the engine) against the same
    loop driven from the host with bbai_bot_act + bbai_step: every history row, the final observation and the mission
    tokens, across chunk boundaries and bot crashes (UnlockToUnlock: the stack-capacity give-ups)."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.missions import detokenize
    a = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=4400)
    b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=4400)
    oa = a.reset()
    b.reset()
    reset_cmd = torch.full((n,), a.RESET_ENV, dtype=torch.uint8, device=a.device)
    ref = {k: [] for k in ("image", "direction", "action", "reward", "done", "gave_up", "mission")}
    for t in range(steps):
        ref["image"].append(a.image.cpu().numpy().copy())
        ref["direction"].append(a.direction.cpu().numpy().copy())
        ref["mission"].append(list(oa["mission"]))
        act = a.bot_actions(None)
        crashed = act == a.BOT_GAVE_UP
        act = torch.where(crashed, reset_cmd, act)
        oa, r, d, _ = a.step(act)
        ref["action"].append(act.cpu().numpy().copy())
        ref["gave_up"].append(crashed.cpu().numpy().astype(np.uint8))
        ref["reward"].append(r.cpu().numpy().copy())
        ref["done"].append(d.cpu().numpy().copy())
    got = {k: [] for k in ("image", "direction", "action", "reward", "done", "gave_up", "tokens")}
    t = 0
    for chunk in (1, 7, 32, steps):          # uneven chunks: the state carried between calls is the engine's own
        chunk = min(chunk, steps - t)
        if chunk <= 0:
            break
        r = b.bot_rollout(chunk, tokens=True)
        for k in got:
            got[k].append(r[k].cpu().numpy())
        t += chunk
    assert t == steps
    for k in ("image", "direction", "action", "reward", "done", "gave_up"):
        assert np.array_equal(np.concatenate(got[k]), np.stack(ref[k])), k
    toks = np.concatenate(got["tokens"])
    for t in range(0, steps, 13):
        for i in range(0, n, 17):
            assert detokenize(toks[t, i]) == ref["mission"][t][i], (t, i)
    assert np.array_equal(a.image.cpu().numpy(), b.image.cpu().numpy())
    assert np.array_equal(a.direction.cpu().numpy(), b.direction.cpu().numpy())
    assert np.stack(ref["done"]).any()
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("offset", [0, 4, 8, 1, 3])
def test_step_writes_observations_into_unaligned_caller_buffers(gpu, offset):
    """k_step's copy-out streams 16 bytes per lane when the caller's image buffer allows it and falls back to dwords /
    bytes when it does not (a row of a [T][n][147] history with odd n): same bytes either way, nothing outside the rows."""
    import ctypes
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv, _check
    n = 777                                         # three full blocks + a partial one
    a = BatchedBabyAIEnv("BabyAI-GoTo-v0", n, device=gpu, seeds=90)
    b = BatchedBabyAIEnv("BabyAI-GoTo-v0", n, device=gpu, seeds=90)
    a.reset()
    b.reset()
    big = torch.full((n * 147 + 64,), 0xA5, dtype=torch.uint8, device=a.device)
    rng = np.random.RandomState(3)
    for t in range(12):
        act = torch.as_tensor(rng.randint(0, 7, n).astype(np.uint8), device=a.device)
        a.step(act)
        big.fill_(0xA5)
        _check(b.lib, b.lib.bbai_step(b.handle, act.data_ptr(), big.data_ptr() + 16 + offset, b.direction.data_ptr(), b.reward.data_ptr(),
                                      b.reward64.data_ptr(), b.done.data_ptr(), 1, b._stream()), "bbai_step")
        torch.cuda.synchronize()
        h = big.cpu().numpy()
        assert np.array_equal(h[16 + offset:16 + offset + n * 147], a.image.cpu().numpy().reshape(-1)), t
        assert (h[:16 + offset] == 0xA5).all() and (h[16 + offset + n * 147:] == 0xA5).all(), t
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["0", "1", "inplace"])
@pytest.mark.parametrize("level", ["BossLevel", "PutNextS6N3Carrying", "KeyInBox"])
def test_record_path_equals_window_plane_path(gpu, level, fused, monkeypatch):
    """BBAI_VPLANE=0 (the step's window and front cell come out of the record's appearance plane: round 2's path, kept for
    A/B measurements) against the default window-plane path, with the finished envs consumed by k_consume and inside
    k_step -- all four k_step instantiations: same observations, rewards, dones and pixels at every step, object actions
    included."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    n, steps = 1500, 120
    _consume_mode(monkeypatch, fused)
    a = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=31, pixel=True)
    monkeypatch.setenv("BBAI_VPLANE", "0")
    b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=31, pixel=True)
    monkeypatch.delenv("BBAI_VPLANE")
    oa, ob = a.reset(), b.reset()
    gen = torch.Generator(device=gpu)
    gen.manual_seed(8)
    weights = torch.tensor([1.0, 1.0, 2.0, 2.0, 2.0, 2.0, 0.2], device=gpu)      # plenty of pickup / drop / toggle
    for t in range(steps):
        assert torch.equal(a.image, b.image) and torch.equal(oa["image"], ob["image"]) and torch.equal(a.direction, b.direction), t
        act = torch.multinomial(weights, n, replacement=True, generator=gen).to(torch.uint8)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(a.reward64, b.reward64) and torch.equal(da, db), t
    assert a.reset_count() == b.reset_count() and a.reset_count() > n
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", "inplace"])
@pytest.mark.parametrize("vplane", ["1", "0"])
@pytest.mark.parametrize("level", ["GoToLocal", "PutNextS6N3Carrying", "BossLevel", "KeyInBox"])
def test_fused_consume_equals_k_consume_under_reset_storms(gpu, level, vplane, mode, monkeypatch):
    """The finished envs consumed inside k_step (consume_fused = 1), and moved on in place by their own lanes (the in-place layout),
    against the k_consume launch when waves carry anything from none to sixty-four finished envs: a reset command on a tenth of the
    envs, for a stretch on most of them, explicit reset() calls in between.  Every output byte and pixel at every step; afterwards the
    missions, the reset counts and the exported records / hot state / stale sets."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    n, steps = 2500, 140
    monkeypatch.setenv("BBAI_VPLANE", vplane)
    _consume_mode(monkeypatch, "0")
    a = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=77, pixel=True)
    _consume_mode(monkeypatch, mode)
    b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=77, pixel=True)
    assert b.get_option("inplace") == (1 if mode == "inplace" else 0) and a.get_option("inplace") == 0
    oa, ob = a.reset(), b.reset()
    gen = torch.Generator(device=gpu)
    gen.manual_seed(5)
    weights = torch.tensor([1.0, 1.0, 2.0, 1.5, 1.5, 1.5, 0.3, 0.9], device=gpu)      # 7 = "reset this env now"
    heavy = torch.tensor([1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 21.0], device=gpu)
    for t in range(steps):
        assert torch.equal(a.image, b.image) and torch.equal(oa["image"], ob["image"]) and torch.equal(a.direction, b.direction), t
        if t in (33, 34, 101):
            oa, ob = a.reset(), b.reset()
            continue
        w = heavy if 60 <= t < 70 else weights
        act = torch.multinomial(w, n, replacement=True, generator=gen).to(torch.uint8)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(a.reward64, b.reward64) and torch.equal(da, db), t
    assert a.reset_count() == b.reset_count() and a.reset_count() > 8 * n
    assert a.missions() == b.missions()
    (ra_, ha_, sa_), (rb_, hb_, sb_) = a.export_state(), b.export_state()
    ha_[:, 15] = hb_[:, 15] = 0          # the ring slot: the in-place ring is one slot deeper
    assert np.array_equal(ra_, rb_) and np.array_equal(ha_, hb_) and np.array_equal(sa_, sb_)
    a.close()
    b.close()


N_QUEUE_SHAPES = 11


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 9, 5003, 70001])
def test_render_queue_equals_one_shot_render(gpu, n):
    """k_render_q (persistent blocks fed by ticket counters; the default from 786 432 envs up) against the one-shot k_render:
    every pixel byte, every queue shape of render_launch, ragged last groups / tickets, fewer tickets than blocks, odd block
    counts, across resets -- and launch after launch, because the kernel leaves its own ticket counters at zero for the
    next one."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    a = BatchedBabyAIEnv("BabyAI-GoToObjS6-v0", n, device=gpu, pixel=True, seeds=5)
    b = BatchedBabyAIEnv("BabyAI-GoToObjS6-v0", n, device=gpu, pixel=True, seeds=5)
    a.set_option("render_queue", 0)
    b.set_option("render_queue", 1)
    oa, ob = a.reset(), b.reset()
    gen = torch.Generator(device=gpu)
    gen.manual_seed(1)
    for t in range(4 * N_QUEUE_SHAPES):
        assert torch.equal(oa["image"], ob["image"]), t
        b.set_option("render_queue", 1 + t % N_QUEUE_SHAPES)
        b.set_option("render_queue_bpc", (0, 1, 3, 0)[t // N_QUEUE_SHAPES])
        b.set_option("render_queue_blocks", (0, 0, 0, 77)[t // N_QUEUE_SHAPES])
        b.set_option("render_pace", (0, 170, 0, 183)[t // N_QUEUE_SHAPES])            # (time-gated tickets: same bytes)
        act = torch.randint(0, 7, (n,), dtype=torch.uint8, device=gpu, generator=gen)
        oa, _, _, _ = a.step(act)
        ob, _, _, _ = b.step(act)
    assert torch.equal(oa["image"], ob["image"])
    # and through bbai_render of a stored encoding, every shape
    pa = torch.zeros_like(a.pixels)
    a.render_encoding(a.image, pa)
    for mode in range(1, N_QUEUE_SHAPES + 2):
        b.set_option("render_queue", mode)
        pb = torch.zeros_like(b.pixels)
        b.render_encoding(a.image, pb)
        assert torch.equal(pa, pb), mode
    a.close()
    b.close()


@pytest.mark.gpu
def test_options_do_not_change_results(gpu):
    """include/bbai.h bbai_set_option: knobs choose launch shapes, never bytes.  One batch stepped with the defaults, one
    with a different setting of every knob every few steps; unknown names are rejected."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv, EngineError
    n = 3001
    a = BatchedBabyAIEnv("BabyAI-PickupLoc-v0", n, device=gpu, pixel=True, seeds=11)
    b = BatchedBabyAIEnv("BabyAI-PickupLoc-v0", n, device=gpu, pixel=True, seeds=11)
    with pytest.raises(EngineError):
        b.set_option("no_such_knob", 1)
    oa, ob = a.reset(), b.reset()
    gen = torch.Generator(device=gpu)
    gen.manual_seed(3)
    settings = [("step_prio", 0), ("pregen_group", 64), ("pregen_blocks", 64), ("render_group", 4), ("render_tpb", 256),
                ("consume_fused", 1), ("pregen_group", 16), ("render_queue", 1), ("consume_fused", 0),
                ("render_queue", 6), ("pregen_group", 32), ("step_prio", 1), ("consume_fused", 1), ("consume_fused", -1), ("pregen_min", 0),
                ("pregen_min", 7), ("pregen_group", 64), ("pregen_min", 2048), ("gate_strict", 1), ("pregen_blocks", 64), ("gate_strict", 0),
                ("lookahead_streams", 3), ("lookahead_streams", 8), ("gate_strict", 1), ("lookahead_streams", 2), ("gate_strict", 0), ("lookahead_streams", 1)]
    for t in range(20 * len(settings)):
        assert torch.equal(oa["image"], ob["image"]) and torch.equal(a.image, b.image) and torch.equal(a.direction, b.direction), t
        if t % 20 == 0:
            b.set_option(*settings[t // 20])
        act = torch.randint(0, 7, (n,), dtype=torch.uint8, device=gpu, generator=gen)
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(a.reward64, b.reward64) and torch.equal(da, db), t
    assert a.reset_count() == b.reset_count() and a.reset_count() > 2 * n
    assert a.gate_timeouts() == 0 and b.gate_timeouts() == 0
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["0", "1"])
@pytest.mark.parametrize("level,n,period", [("GoToObjS4", 20000, "4"), ("GoTo", 6000, "2"), ("PickupLoc", 30000, None)])
def test_steps_run_ahead_of_a_reset_storms_refill(gpu, level, n, period, layout, monkeypatch):
    """The step stream no longer waits for the refill of window w at the start of window w + 2: it runs on while every env is sure to
    keep a window's worth of ready levels (bbai_engine.hip k_gate).  A reset command to EVERY env on one tick (a storm: one long
    refill) followed at once by many windows of ordinary steps, every few windows another storm, a tick on which a quarter of the envs
    reset again inside the same window (M = 2), and stretches of resets on every tick (the worst case: the gate degenerates to the old
    rule) -- against a second batch that runs the old rule (gate_strict) and, for scattered envs, against the oracle."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from oracle import levels as olevels
    monkeypatch.setenv("BBAI_INPLACE", layout)
    if period:
        monkeypatch.setenv("BBAI_LOOKAHEAD", period)
    a = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=77)
    b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=77)
    b.set_option("gate_strict", 1)
    B = a.get_option("lookahead_period")
    ids = [0, 1, 63, 64, n // 2, n - 65, n - 1]
    refs = []
    for i in ids:
        e = olevels.make_env(level)
        e.seed(77 + i)
        refs.append([e, e.reset()])
    a.reset()
    b.reset()
    gen = torch.Generator(device=gpu)
    gen.manual_seed(5)
    T = 14 * B + 9
    for t in range(T):
        img = a.image[torch.as_tensor(ids, device=gpu)].cpu().numpy()
        for k, i in enumerate(ids):
            assert np.array_equal(img[k], refs[k][1]["image"]), (level, layout, t, i)
        act = torch.randint(0, 7, (n,), dtype=torch.uint8, device=gpu, generator=gen)
        if t % (5 * B) == 1:
            act[:] = a.RESET_ENV                                  # the storm
        elif t % (5 * B) == 2:
            act[::4] = a.RESET_ENV                                # ... and again inside the same window for a quarter of the envs
        elif 8 * B <= t < 10 * B + 3:
            act[5::7] = a.RESET_ENV                               # resets on every tick for more than two windows
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert torch.equal(a.image, b.image) and torch.equal(a.direction, b.direction) and torch.equal(a.reward64, b.reward64) and torch.equal(da, db), (level, layout, t)
        acts = act[torch.as_tensor(ids, device=gpu)].cpu().numpy()
        dn = da[torch.as_tensor(ids, device=gpu)].cpu().numpy()
        for k in range(len(ids)):
            if acts[k] == a.RESET_ENV:
                refs[k][1] = refs[k][0].reset()
                assert dn[k] == 1
            else:
                o, r, d, _ = refs[k][0].step(int(acts[k]))
                assert bool(d) == bool(dn[k]), (level, layout, t, ids[k])
                refs[k][1] = refs[k][0].reset() if d else o
    assert a.reset_count() == b.reset_count() and a.reset_count() > 3 * n
    assert a.gate_timeouts() == 0 and b.gate_timeouts() == 0
    a.close()
    b.close()


@pytest.mark.gpu
def test_gate_probe_and_sticky_fault(gpu, monkeypatch):
    """the relaxed window gate spins on the device for a value the look-ahead stream stores, which needs the two streams to run
    concurrently -- HIP does not promise that.  (1) the first window a caller's stream opens probes it: on this box the verdict is
    "concurrent"; a handle whose probe fails (gate_probe = 2) falls back to the strict rule and yields the same bytes.  (2) a gate that
    timed out leaves a sticky flag in pinned host memory: the next step() raises, at the call, and a re-seed makes the handle whole."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv, EngineError
    monkeypatch.setenv("BBAI_LOOKAHEAD", "2")
    n = 4096
    a = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=5)
    b = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=5)
    b.set_option("gate_probe", 2)
    a.reset()
    b.reset()
    assert a.get_option("gate_forced_strict") == 0 and b.get_option("gate_forced_strict") == 1
    gen = torch.Generator(device=gpu)
    gen.manual_seed(11)
    side = torch.cuda.Stream(device=gpu)
    for t in range(40):
        act = torch.randint(0, 7, (n,), dtype=torch.uint8, device=gpu, generator=gen)
        if t == 20:                                     # another caller stream: probed at the next window it opens
            torch.cuda.synchronize()
            torch.cuda.set_stream(side)
        _, ra, da, _ = a.step(act)
        _, rb, db, _ = b.step(act)
        assert torch.equal(a.image, b.image) and torch.equal(a.reward64, b.reward64) and torch.equal(da, db), t
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream(gpu))
    assert a.get_option("gate_forced_strict") == 0 and a.gate_timeouts() == 0 and not a.gate_fault()
    # the sticky fault: refused at the call, cleared by a re-seed
    a.set_option("gate_fault_inject", 1)
    assert a.gate_fault()
    with pytest.raises(EngineError, match="window gate"):
        a.step(act)
    with pytest.raises(EngineError, match="window gate"):
        a.reset()
    a.seed(5)
    assert not a.gate_fault()
    a.reset()
    b.seed(5)
    b.reset()
    a.step(act)
    b.step(act)
    assert torch.equal(a.image, b.image)
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level,auto_reset", [("PickupLoc", True), ("GoTo", True), ("GoToLocal", False)])
def test_step_render_split_equals_the_plain_step_and_render(gpu, level, auto_reset):
    """include/bbai.h bbai_step_render with option "step_render_split": the batch stepped in two halves, the second half's step kernel on
    the handle's split stream under the first half's render -- every byte of every output equals the unsplit call's, tokens and
    auto-resets (in-place and in-wave consume) included, and bbai_rollout takes the same path."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.action_stream import actions_torch
    n = 262144 + 64 + 5
    a = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, pixel=True, seeds=31, auto_reset=auto_reset)
    b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, pixel=True, seeds=31, auto_reset=auto_reset)
    b.set_option("step_render_split", 1)
    a.set_option("step_render_split", 0)
    ta, tb = a.enable_instr_tokens(), b.enable_instr_tokens()
    a.reset()
    b.reset()
    T = 48
    acts = actions_torch(9, 0, T, 0, n, gpu)
    for t in range(T - 8):
        oa, ra, da, _ = a.step(acts[t])
        ob, rb, db, _ = b.step(acts[t])
        assert torch.equal(oa["image"], ob["image"]), (level, t)
        assert torch.equal(a.image, b.image) and torch.equal(a.direction, b.direction) and torch.equal(a.reward64, b.reward64) and torch.equal(da, db), (level, t)
        assert torch.equal(ta, tb), (level, t)
    a.rollout(acts[T - 8:])
    b.rollout(acts[T - 8:])
    assert torch.equal(a.pixels, b.pixels) and torch.equal(a.image, b.image) and torch.equal(a.done, b.done) and torch.equal(ta, tb)
    assert a.reset_count() == b.reset_count() and (a.reset_count() > n or not auto_reset)
    a.close()
    b.close()


@pytest.mark.gpu
def test_a_refused_checkpoint_leaves_the_handle_as_it_was(gpu, monkeypatch):
    """bbai_checkpoint_load validates EVERYTHING before it touches the handle: a blob of another look-ahead period that is
    refused for another reason -- wrong done-action mode, truncated, another format version -- must not re-shape the ring on its way
    out.  The handle keeps stepping in lock-step with a twin afterwards."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv, EngineError
    n = 600
    monkeypatch.setenv("BBAI_LOOKAHEAD", "4")
    donor = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=5, done_actions=True)
    donor.reset()
    blob = donor.save_checkpoint()
    donor.close()
    monkeypatch.setenv("BBAI_LOOKAHEAD", "2")
    a = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=9)
    b = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=9)
    a.reset()
    b.reset()
    gen = torch.Generator(device=gpu)
    gen.manual_seed(1)

    def lockstep(steps):
        for t in range(steps):
            act = torch.randint(0, 7, (n,), dtype=torch.uint8, device=gpu, generator=gen)
            a.step(act)
            b.step(act)
            assert torch.equal(a.image, b.image) and torch.equal(a.reward64, b.reward64) and torch.equal(a.done, b.done), t

    lockstep(7)
    with pytest.raises(EngineError, match="done-action mode"):
        a.load_checkpoint(blob)                       # period 4 into a period-2 handle, refused for its done-action bits
    assert a.get_option("lookahead_period") == 2
    lockstep(9)
    donor = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=5)      # (created under BBAI_LOOKAHEAD=2 ...)
    monkeypatch.setenv("BBAI_LOOKAHEAD", "4")
    donor4 = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=5)     # (... and 4: a blob of another ring shape, same mode)
    donor4.reset()
    blob = donor4.save_checkpoint()
    donor.close()
    donor4.close()
    with pytest.raises(EngineError, match="size does not match"):
        a.load_checkpoint(blob[:-64])
    bad = blob.copy()
    bad[8:12] = np.frombuffer(np.int32(2).tobytes(), dtype=np.uint8)      # CkptHeader.version
    with pytest.raises(EngineError, match="unsupported checkpoint version 2"):
        a.load_checkpoint(bad)
    assert a.get_option("lookahead_period") == 2
    lockstep(40)
    assert a.reset_count() == b.reset_count() > n
    a.close()
    b.close()


@pytest.mark.gpu
def test_import_state_keeps_the_importing_handles_ring_position(gpu, monkeypatch):
    """hot[15] (the ring slot) of an exported state belongs to the EXPORTER's ring: an in-place handle with a 65-slot ring exports slots a
    classic handle with a 4-slot ring does not have.  The import keeps the importer's own slot in both layouts, and the
    importer goes on auto-resetting through ITS level stream."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    n = 300
    monkeypatch.setenv("BBAI_INPLACE", "1")
    monkeypatch.delenv("BBAI_LOOKAHEAD", raising=False)
    src = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=3)
    src.reset()
    gen = torch.Generator(device=gpu)
    gen.manual_seed(2)
    for t in range(150):                                  # far enough that ring slots beyond 4 are in use
        src.step(torch.randint(0, 7, (n,), dtype=torch.uint8, device=gpu, generator=gen))
    rec, hot, stale = src.export_state()
    assert hot[:, 15].max() >= 4
    monkeypatch.setenv("BBAI_INPLACE", "0")
    monkeypatch.setenv("BBAI_LOOKAHEAD", "2")
    dst = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=1000)
    twin = BatchedBabyAIEnv("BabyAI-GoToObjS4-v0", n, device=gpu, seeds=1000)
    dst.reset()
    twin.reset()
    dst.import_state(rec, hot, stale)
    _, hot2, _ = dst.export_state()
    assert hot2[:, 15].max() < 4                          # the importer's own slots
    assert np.array_equal(hot2[:, :15], hot[:, :15])
    # the imported episodes play on exactly like the exporter's ...
    first_done = np.zeros(n, bool)
    for t in range(60):
        act = torch.randint(0, 7, (n,), dtype=torch.uint8, device=gpu, generator=gen)
        src.step(act)
        dst.step(act)
        twin.step(act)
        live = ~first_done
        assert np.array_equal(src.reward64.cpu().numpy()[live], dst.reward64.cpu().numpy()[live]), t
        assert np.array_equal(src.done.cpu().numpy()[live], dst.done.cpu().numpy()[live]), t
        fin_now = dst.done.cpu().numpy().astype(bool) & live
        first_done |= fin_now
    # ... and once finished, every env continues in the IMPORTER's own level stream: the levels its twin (same seeds, one reset() at the
    # same point of the stream) gets
    assert first_done.sum() > n // 2
    assert dst.gate_timeouts() == 0
    src.close(); dst.close(); twin.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level,n,pixel", [("GoToLocal", 3000, False), ("BossLevel", 1500, True), ("PickupLoc", 700, True)])
def test_rollout_entry_equals_per_step_calls(gpu, level, n, pixel):
    """bbai_rollout (T steps [+ render] [+ tap] enqueued by ONE call: what bench.py times) against the same T steps as per-step
    calls: every logged output of the tapped envs at every step, and the final outputs of every env."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.action_stream import actions_torch
    from babyai_amd.shard import scattered_ids
    T, P, PP = 70, 96, (8 if pixel else 0)
    a = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, pixel=pixel, seeds=21)
    b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, pixel=pixel, seeds=21)
    a.reset()
    b.reset()
    acts = actions_torch(5, 0, T, 0, n, gpu)
    ids = torch.as_tensor(scattered_ids(n, P), dtype=torch.int64, device=gpu)

    def mklog():
        lg = {"image": torch.zeros((T + 1, P, 7, 7, 3), dtype=torch.uint8, device=gpu), "direction": torch.zeros((T + 1, P), dtype=torch.uint8, device=gpu),
              "reward64": torch.zeros((T, P), dtype=torch.float64, device=gpu), "done": torch.zeros((T, P), dtype=torch.uint8, device=gpu), "ids": ids}
        if PP:
            lg["pixels"] = torch.zeros((T + 1, PP, 56, 56, 3), dtype=torch.uint8, device=gpu)
        return lg

    la, lb = mklog(), mklog()
    for t in range(T):
        a.step(acts[t])
        a.tap(la["image"][t + 1], la["direction"][t + 1], la["reward64"][t], la["done"][t], la["pixels"][t + 1] if PP else None, ids=ids)
    obs = b.rollout(acts[:40], tap=lb, obs_row0=1, row0=0)             # two calls: the second continues where the first stopped
    obs = b.rollout(acts[40:], tap=lb, obs_row0=41, row0=40)
    for k in la:
        assert torch.equal(la[k], lb[k]), k
    assert torch.equal(a.image, b.image) and torch.equal(a.direction, b.direction) and torch.equal(a.reward64, b.reward64)
    assert torch.equal(a.done, b.done) and torch.equal(a.reward, b.reward)
    if pixel:
        assert torch.equal(a.pixels, b.pixels) and obs["image"] is b.pixels
    assert a.reset_count() == b.reset_count() and obs["mission"][0] == a.missions()[0]
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level,n,opts", [("GoToLocal", 3000, {}), ("PickupLoc", 700, {}), ("BossLevel", 1500, {}), ("GoTo", 1300, {}), ("GoTo", 1100, {"consume_fused": 0}),
                                          ("GoToLocal", 1000, {"auto_reset": False}), ("PutNextS5N2Carrying", 900, {}), ("KeyInBox", 800, {}),
                                          ("GoToLocal", 2500, {"inplace_off": True}), ("SynthS5R2", 1000, {}), ("GoToRedBallGrey", 700, {"done_actions": True}),
                                          ("GoToLocal", 9000, {"streams": 3}), ("GoTo", 5000, {"streams": 5}), ("BossLevel", 4000, {"streams": 2}), ("PickupLoc", 2000, {"streams": 8})])
def test_rollout_steps_many_ticks_per_launch(gpu, level, n, opts):
    """bbai_rollout's fast path -- ONE k_step launch per look-ahead window's remaining ticks, log rows written by the stepping lanes -- against
    one bbai_step + bbai_tap_ids per step: every logged byte of every step (through resets, window boundaries inside a call, calls that start
    in mid-window), the final outputs of every env, the reset count, and the state the run leaves behind (both continue per step and must stay
    equal).  Shapes where the fast path does not apply (unfused consume) take the same entry and must give the same bytes.  `streams`: the
    refills of the two rollout batches are split over that many look-ahead streams (option "lookahead_streams"), the per-step batch keeps one."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.action_stream import actions_torch
    from babyai_amd.shard import scattered_ids
    T, T2, P = 117, 40, 96
    auto = opts.get("auto_reset", True)
    env_kw = {}
    if opts.get("inplace_off"):
        os.environ["BBAI_INPLACE"] = "0"
    if opts.get("done_actions"):
        env_kw["done_actions"] = True
    try:
        a = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=33, auto_reset=auto, **env_kw)
        b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=33, auto_reset=auto, **env_kw)
        c = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=33, auto_reset=auto, **env_kw)
    finally:
        os.environ.pop("BBAI_INPLACE", None)
    for env in (a, b, c):
        for k, v in opts.items():
            if k not in ("auto_reset", "inplace_off", "done_actions", "streams"):
                env.set_option(k, v)
        if "streams" in opts and env is not a:          # the window refills split over several look-ahead streams (the reference batch: one)
            env.set_option("lookahead_streams", opts["streams"])
            assert env.get_option("lookahead_streams") == opts["streams"]
        env.reset()
    assert b.get_option("rollout_multi") == 1
    acts = actions_torch(9, 0, T + T2, 0, n, gpu)
    idl = list(scattered_ids(n, P))
    idl = idl[1::2] + idl[0::2]
    ids = torch.as_tensor(idl, dtype=torch.int64, device=gpu)

    def mklog():
        return {"image": torch.zeros((T, P, 7, 7, 3), dtype=torch.uint8, device=gpu), "direction": torch.zeros((T, P), dtype=torch.uint8, device=gpu),
                "reward64": torch.zeros((T, P), dtype=torch.float64, device=gpu), "done": torch.zeros((T, P), dtype=torch.uint8, device=gpu)}

    la, lb = mklog(), mklog()
    for t in range(T):
        a.step(acts[t])
        a.tap(la["image"][t], la["direction"][t], la["reward64"][t], la["done"][t], None, ids=ids)
    b.set_step_tap(idl)
    cuts = [0, 5, 6, 45, 110, T]                # calls that end and start in mid-window, one that spans three windows, a one-step call
    for t0, t1 in zip(cuts, cuts[1:]):
        b.rollout(acts[t0:t1], tap=lb, obs_row0=t0, row0=t0, step_tap=True)
    c.rollout(acts[:T])                         # no log at all
    for k in la:
        assert torch.equal(la[k], lb[k]), k
    for other in (b, c):
        assert torch.equal(a.image, other.image) and torch.equal(a.direction, other.direction) and torch.equal(a.reward64, other.reward64)
        assert torch.equal(a.done, other.done) and torch.equal(a.reward, other.reward)
        assert a.reset_count() == other.reset_count()
    if auto and level in ("GoToLocal", "PickupLoc"):
        assert a.reset_count() > n + 100
    # the states are equal too: the three go on, step by step / in one more rollout, and stay together
    for t in range(T, T + T2):
        a.step(acts[t])
        b.step(acts[t])
    c.rollout(acts[T:])
    for other in (b, c):
        assert torch.equal(a.image, other.image) and torch.equal(a.direction, other.direction) and torch.equal(a.reward64, other.reward64) and torch.equal(a.done, other.done)
        assert a.reset_count() == other.reset_count()
    assert a.missions()[:50] == b.missions()[:50] == c.missions()[:50]
    with pytest.raises(Exception):              # the log must match the listed envs
        b.rollout(acts[:2], tap={k: v[:, :P - 1].contiguous() for k, v in lb.items()}, step_tap=True)
    for env in (a, b, c):
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level,n", [("GoToLocal", 2000), ("BossLevel", 1200)])
def test_rollout_with_a_token_buffer_keeps_one_step_per_launch(gpu, level, n):
    """A registered mission-token buffer has to follow every step's resets (k_tokens reads each step's `done` bytes), so bbai_rollout keeps one
    step per launch then -- through the same entry, with the same bytes, the token rows included; the stepping lanes' tap rows still work there;
    option rollout_multi 0 does the same without a token buffer."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.action_stream import actions_torch
    from babyai_amd.shard import scattered_ids
    T, P = 100, 64
    envs = [BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=77) for _ in range(3)]
    a, b, c = envs
    for env in (a, b):
        env.enable_instr_tokens()
    c.set_option("rollout_multi", 0)
    for env in envs:
        env.reset()
    acts = actions_torch(13, 0, T, 0, n, gpu)
    idl = list(scattered_ids(n, P))

    def mklog():
        return {"image": torch.zeros((T, P, 7, 7, 3), dtype=torch.uint8, device=gpu), "direction": torch.zeros((T, P), dtype=torch.uint8, device=gpu),
                "reward64": torch.zeros((T, P), dtype=torch.float64, device=gpu), "done": torch.zeros((T, P), dtype=torch.uint8, device=gpu)}

    la, lb, lc = mklog(), mklog(), mklog()
    a.set_step_tap(idl)
    for t in range(T):
        a.step_tapped(acts[t], la["image"][t], la["direction"][t], la["reward64"][t], la["done"][t])
    b.set_step_tap(idl)
    b.rollout(acts[:37], tap=lb, obs_row0=0, row0=0, step_tap=True)
    b.rollout(acts[37:], tap=lb, obs_row0=37, row0=37, step_tap=True)
    c.set_step_tap(idl)
    c.rollout(acts, tap=lc, obs_row0=0, row0=0, step_tap=True)
    for k in la:
        assert torch.equal(la[k], lb[k]) and torch.equal(la[k], lc[k]), k
    assert torch.equal(a.instr, b.instr)
    for other in (b, c):
        assert torch.equal(a.image, other.image) and torch.equal(a.reward64, other.reward64) and torch.equal(a.done, other.done)
        assert a.reset_count() == other.reset_count()
    if level == "GoToLocal":
        assert a.reset_count() > n + 100
    with pytest.raises(Exception):              # a tap log of the stepping lanes has no pixel rows; rollouts of a pixel batch keep the launch (ids)
        p = BatchedBabyAIEnv("BabyAI-%s-v0" % level, 256, device=gpu, seeds=1, pixel=True)
        try:
            p.reset()
            p.set_step_tap([0, 1])
            p.rollout(acts[:2, :256].contiguous(), tap={k: v[:, :2].contiguous() for k, v in mklog().items()}, step_tap=True)
        finally:
            p.close()
    for env in envs:
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level,n,opts", [("GoToLocal", 3000, {}), ("PickupLoc", 700, {}), ("BossLevel", 1500, {}), ("GoTo", 1300, {"consume_fused": 0}),
                                          ("GoToLocal", 1000, {"auto_reset": False}), ("PutNextS5N2Carrying", 900, {})])
def test_step_tapped_equals_step_plus_tap(gpu, level, n, opts):
    """bbai_step_tapped (the stepping lanes write the listed envs' log rows; bench.py's timed loop) against bbai_step + bbai_tap_ids: every
    logged byte of every step -- through resets (the new episode's first observation is what the row must hold), in both state layouts,
    with the unfused consume (there the tap stays a launch behind k_consume) and without auto-reset (frozen envs re-emit)."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.action_stream import actions_torch
    from babyai_amd.shard import scattered_ids
    T, P = 90, 96
    auto = opts.get("auto_reset", True)
    a = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=21, auto_reset=auto)
    b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=21, auto_reset=auto)
    for env in (a, b):
        for k, v in opts.items():
            if k != "auto_reset":
                env.set_option(k, v)
        env.reset()
    acts = actions_torch(5, 0, T, 0, n, gpu)
    idl = list(scattered_ids(n, P))
    idl = idl[1::2] + idl[0::2]                # (not ascending: log row k = env ids[k] whatever the order)
    ids = torch.as_tensor(idl, dtype=torch.int64, device=gpu)

    def mklog():
        return {"image": torch.zeros((T, P, 7, 7, 3), dtype=torch.uint8, device=gpu), "direction": torch.zeros((T, P), dtype=torch.uint8, device=gpu),
                "reward64": torch.zeros((T, P), dtype=torch.float64, device=gpu), "done": torch.zeros((T, P), dtype=torch.uint8, device=gpu)}

    la, lb = mklog(), mklog()
    b.set_step_tap(idl)
    for t in range(T):
        a.step(acts[t])
        a.tap(la["image"][t], la["direction"][t], la["reward64"][t], la["done"][t], None, ids=ids)
        b.step_tapped(acts[t], lb["image"][t], lb["direction"][t], lb["reward64"][t], lb["done"][t])
    for k in la:
        assert torch.equal(la[k], lb[k]), k
    assert torch.equal(a.image, b.image) and torch.equal(a.direction, b.direction) and torch.equal(a.reward64, b.reward64) and torch.equal(a.done, b.done)
    assert a.reset_count() == b.reset_count()
    if auto and level in ("GoToLocal", "PickupLoc"):
        assert a.reset_count() > n + 100       # (episodes ended under the tap)
    with pytest.raises(Exception):
        b.set_step_tap([0, 0])
    b.set_step_tap(None)
    with pytest.raises(Exception):
        b.step_tapped(acts[0], lb["image"][0], lb["direction"][0], lb["reward64"][0], lb["done"][0])
    a.close()
    b.close()
