"""Full-size (BASELINE.json configs) checks on the GPU through size-independent properties, plus
oracle spot checks on scattered envs of the 1M-env batch."""
import numpy as np
import pytest

N_FULL = 1048576


def _expected_pixels(torch, image, atlas_np, lut_np):
    """RGBImgPartialObsWrapper as a torch gather (independent restatement of k_render's indexing)."""
    dev = image.device
    atlas = torch.as_tensor(atlas_np, device=dev)                 # [T, 8, 8, 3]
    lut = torch.as_tensor(lut_np.astype(np.int64), device=dev)    # [2, 256]
    img = image.to(torch.int64)
    key = img[..., 0] | (img[..., 1] << 3) | (img[..., 2] << 6)   # [n, 7(vx), 7(vy)]
    agent = torch.zeros((7, 7), dtype=torch.int64, device=dev)
    agent[3, 6] = 1
    tile = lut[agent.expand_as(key), key]                          # [n, 7, 7]
    t = atlas[tile]                                                # [n, vx, vy, ty, tx, 3]
    return t.permute(0, 2, 3, 1, 4, 5).reshape(image.shape[0], 56, 56, 3)


@pytest.mark.gpu
def test_render_full_size_matches_gather(gpu):
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv, ATLAS_PATH
    env = BatchedBabyAIEnv("BabyAI-BossLevel-v0", N_FULL, device=gpu, pixel=True, seeds=0)
    obs = env.reset()
    acts = torch.randint(0, 7, (24, N_FULL), dtype=torch.uint8, device=gpu)
    for t in range(24):
        obs, _, _, _ = env.step(acts[t])
    at = np.load(ATLAS_PATH)
    chunk = 131072
    for lo in range(0, N_FULL, chunk):
        exp = _expected_pixels(torch, env.image[lo:lo + chunk], at["tiles"], at["lut"])
        assert torch.equal(obs["image"][lo:lo + chunk], exp), "pixel mismatch in envs [%d, %d)" % (lo, lo + chunk)
    env.close()


@pytest.mark.gpu
def test_full_size_determinism_invariants_and_oracle_spots(gpu):
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from oracle import levels as olevels
    T = 40
    a = BatchedBabyAIEnv("BabyAI-BossLevel-v0", N_FULL, device=gpu, seeds=0)
    b = BatchedBabyAIEnv("BabyAI-BossLevel-v0", N_FULL, device=gpu, seeds=0)
    a.reset()
    b.reset()
    gen = torch.Generator(device=gpu)
    gen.manual_seed(99)
    acts = torch.randint(0, 7, (T, N_FULL), dtype=torch.uint8, device=gpu, generator=gen)
    spots = [0, 1, 255, 256, 65535, 65536, 524287, 524288, N_FULL - 257, N_FULL - 1]
    refs = []
    for i in spots:
        e = olevels.make_env("BossLevel")
        e.seed(i)
        refs.append((e, e.reset()))
    img0 = a.image[spots].cpu().numpy()
    for k, (e, o) in enumerate(refs):
        assert np.array_equal(img0[k], o["image"]), "env %d first obs" % spots[k]
        assert a._obs()["mission"][spots[k]] == o["mission"]
    acts_host = acts[:, spots].cpu().numpy()
    for t in range(T):
        a.step(acts[t])
        b.step(acts[t])
        img = a.image[spots].cpu().numpy()
        rew = a.reward[spots].cpu().numpy()
        dn = a.done[spots].cpu().numpy()
        for k, (e, _) in enumerate(refs):
            o, r, d, _ = e.step(int(acts_host[t, k]))
            if d:
                o = e.reset()
            assert np.array_equal(img[k], o["image"]), "env %d step %d" % (spots[k], t)
            assert np.float32(r) == rew[k] and bool(d) == bool(dn[k])
    torch.cuda.synchronize()
    # determinism: two engines, same seeds and actions -> identical bytes (atomics / scheduling leak nothing)
    assert torch.equal(a.image, b.image) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done)
    assert torch.equal(a.direction, b.direction)
    # invariants of the encoding
    img = a.image
    assert int(a.direction.max()) <= 3
    types = torch.unique(img[..., 0]).tolist()
    assert set(types) <= {0, 1, 2, 4, 5, 6, 7}
    assert bool((img[:, 3, 6, 0] != 0).all()), "the agent's own cell is always visible"
    unseen = img[..., 0] == 0
    assert bool((img[..., 1][unseen] == 0).all()) and bool((img[..., 2][unseen] == 0).all())
    r = a.reward
    assert bool(((r == 0) | ((r > 0.0999) & (r <= 1.0))).all())
    assert bool((a.done[r > 0] == 1).all())
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level,n", [("GoToLocal", 65536), ("PickupLoc", 262144), ("GoTo", 131072)])
def test_config_sizes_oracle_spots(gpu, level, n):
    """BASELINE.json configs[1..3] at their full sizes: scattered envs against the oracle, crossing resets."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from oracle import levels as olevels
    T = 150 if level != "GoTo" else 60
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=7)
    env.reset()
    spots = [0, 63, 64, 4097, n // 2, n - 65, n - 1]
    refs = []
    for i in spots:
        e = olevels.make_env(level)
        e.seed(7 + i)
        refs.append([e, e.reset()])
    gen = torch.Generator(device=gpu)
    gen.manual_seed(5)
    acts = torch.randint(0, 7, (T, n), dtype=torch.uint8, device=gpu, generator=gen)
    acts_host = acts[:, spots].cpu().numpy()
    for t in range(T):
        img = env.image[spots].cpu().numpy()
        for k, (e, o) in enumerate(refs):
            assert np.array_equal(img[k], o["image"]), (level, spots[k], t)
        env.step(acts[t])
        rew = env.reward[spots].cpu().numpy()
        for k, (e, _) in enumerate(refs):
            o, r, d, _ = e.step(int(acts_host[t, k]))
            if d:
                o = e.reset()
            refs[k][1] = o
            assert np.float32(r) == rew[k]
    if level != "GoTo":
        assert env.reset_count() > n       # every env crossed at least one auto-reset on average
    env.close()


@pytest.mark.gpu
def test_expert_solves_every_episode_at_scale(gpu):
    """Size-independent property of the expert (bbai_bot_act) on 262 144 BossLevel envs: with the bot choosing every
    action every finished episode is a success, no bot gives up, no capacity limit is hit, no generator failure."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    n = 262144
    env = BatchedBabyAIEnv("BabyAI-BossLevel-v0", n, device=gpu, seeds=9_000_000)
    env.reset()
    episodes = torch.zeros((), dtype=torch.int64, device=gpu)
    solved = torch.zeros((), dtype=torch.int64, device=gpu)
    for t in range(48):
        a = env.bot_actions(None)
        assert int((a == env.BOT_GAVE_UP).sum()) == 0
        _, r, d, _ = env.step(a)
        episodes += d.sum()
        solved += (r > 0).sum()
    assert int(episodes) > n // 8 and int(solved) == int(episodes)
    assert env.bot_stats() == {"gave_up": 0, "capacity": 0} and env.generator_failures() == 0
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level,n,pixel,steps", [("GoToLocal", 65536, False, 160), ("PickupLoc", 262144, False, 160),
                                                 ("GoTo", 131072, False, 640), ("BossLevel", N_FULL, True, 600)])
def test_baseline_configs_1024_scattered_envs_vs_oracle(gpu, level, n, pixel, steps):
    """Every BASELINE.json GPU config at its full per-GPU size: 1024 envs scattered over the whole batch (block and wave
    boundaries, both ends, a pseudo-random spread), EVERY output of EVERY step -- image, direction, float64 reward bits,
    done, and the pixels of 32 of them -- re-derived by the CPU oracle from the seeds and the counter-based action
    stream in a process pool: >= 2 x max_steps steps for the single-room levels (every env crosses auto-resets), >= 600
    steps for the mazes."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.action_stream import actions_torch
    from oracle import cpu_baseline
    from babyai_amd.shard import scattered_ids
    ids = scattered_ids(n, 1024)                   # (the list bench.py taps in its timed region)
    assert len(ids) == 1024 and ids[0] == 0 and ids[-1] == n - 1 and {63, 64, 255, 256, n // 2, n - 256, n - 64} <= set(ids)
    # pixels are checked on the first PP log rows: put a spread (both ends) there
    head = sorted(set(ids[1023 * k // 31] for k in range(32))) if pixel else []
    ids = head + [i for i in ids if i not in set(head)]
    PP = len(head)
    sel = torch.as_tensor(ids, device=gpu)
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, pixel=pixel, seeds=5000)
    env.reset()
    log = {"image": torch.zeros((steps + 1, 1024, 7, 7, 3), dtype=torch.uint8, device=gpu),
           "direction": torch.zeros((steps + 1, 1024), dtype=torch.uint8, device=gpu),
           "reward64": torch.zeros((steps, 1024), dtype=torch.float64, device=gpu),
           "done": torch.zeros((steps, 1024), dtype=torch.uint8, device=gpu)}
    if PP:
        log["pixels"] = torch.zeros((steps + 1, PP, 56, 56, 3), dtype=torch.uint8, device=gpu)
    log["image"][0] = env.image[sel]
    log["direction"][0] = env.direction[sel]
    if PP:
        log["pixels"][0] = env.pixels[sel[:PP]]
    chunk = 64
    for t0 in range(0, steps, chunk):
        acts = actions_torch(777, t0, min(steps, t0 + chunk), 0, n, gpu)
        for k in range(acts.shape[0]):
            t = t0 + k
            env.step(acts[k])
            if t % 2:           # alternately through the one-launch tap (bbai_tap_ids) and through torch indexing
                env.tap(log["image"][t + 1], log["direction"][t + 1], log["reward64"][t], log["done"][t],
                        log["pixels"][t + 1] if PP else None, ids=sel)
                continue
            log["image"][t + 1] = env.image[sel]
            log["direction"][t + 1] = env.direction[sel]
            log["reward64"][t] = env.reward64[sel]
            log["done"][t] = env.done[sel]
            if PP:
                log["pixels"][t + 1] = env.pixels[sel[:PP]]
    torch.cuda.synchronize()
    finished = int(log["done"].sum())
    host = {k: v.cpu().numpy() for k, v in log.items()}
    resets = env.reset_count() - n
    env.close()
    res = cpu_baseline.parity_replay(level, host, 5000, 777, 0, PP, env_ids=ids)
    assert res["mismatches"] == 0, res
    assert res["envs"] == 1024 and res["steps"] == steps
    if level in ("GoToLocal", "PickupLoc"):
        assert finished >= 2 * 1024 and resets >= 2 * n          # everybody crossed auto-resets, twice on average
    else:
        assert finished >= 1024 // 2 if level == "GoTo" else finished > 100


@pytest.mark.gpu
@pytest.mark.parametrize("level,n,T", [("BossLevel", N_FULL, 72), ("GoTo", 131072, 230), ("PickupLoc", 262144, 150), ("GoToLocal", 65536, 150)])
def test_rollout_full_size_equals_per_step_calls(gpu, level, n, T):
    """bbai_rollout's window-per-launch path at the BASELINE.json batch sizes (the sizes bench.py runs it at) against per-step calls: every output byte of
    EVERY env after T steps -- several look-ahead windows, thousands to millions of resets -- the reset totals, and the states the runs leave behind
    (both go on per step and must stay equal)."""
    import torch
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.action_stream import actions_torch
    a = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=5)
    b = BatchedBabyAIEnv("BabyAI-%s-v0" % level, n, device=gpu, seeds=5)
    a.reset()
    b.reset()
    chunk = 24                                    # (the action stream of a million envs, a few steps at a time)
    t = 0
    while t < T:
        k = min(chunk, T - t)
        acts = actions_torch(3, t, t + k, 0, n, gpu)
        for j in range(k):
            a.step(acts[j])
        b.rollout(acts)
        t += k
        assert torch.equal(a.image, b.image) and torch.equal(a.direction, b.direction), t
        assert torch.equal(a.reward64, b.reward64) and torch.equal(a.done, b.done), t
    assert a.reset_count() == b.reset_count() and a.reset_count() > n
    acts = actions_torch(3, T, T + 8, 0, n, gpu)
    for j in range(8):
        a.step(acts[j])
        b.step(acts[j])
    assert torch.equal(a.image, b.image) and torch.equal(a.reward64, b.reward64) and torch.equal(a.done, b.done)
    assert a.gate_timeouts() == 0 and b.gate_timeouts() == 0
    a.close()
    b.close()
