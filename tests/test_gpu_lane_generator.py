"""k_pregen_lane (one lane = one level, babyai_amd/csrc/bbai_genl.hpp + bbai_genlane.hip) against the lane-group kernel k_pregen, on the
device, through the C ABI, on EVERY level kind the lane generator covers: two handles with the same seeds, one per generator, must produce
the same levels (records, poses, missions), the same observations, rewards and episode ends, step for step, through resets, refills of the
look-ahead ring (every env is sent into a new episode at random: action 7) and a switch of generators in mid-run.

The lane-group kernel is pinned to the oracle and the reference by the rest of the suite; this file is what keeps the SECOND form equal
to it on the hardware -- including against the compiler: ROCm 7.2 miscompiles k_pregen_lane with machine-CSE on (bbai_genlane.hip), which
12 envs x 2 levels per kind (test_every_registered_level_vs_oracle) caught on one kind only."""
import os

import numpy as np
import pytest

from babyai_amd.levels import LEVELS, make_cfg


def covered():
    out = []
    for name in sorted(LEVELS):
        c = make_cfg(name)
        if c.kind == 1 or (c.kind == 0 and not c.lock):
            out.append(name)
    return out


def _make(name, n, lane, gpu):
    from babyai_amd.engine import BatchedBabyAIEnv
    old = os.environ.get("BBAI_PREGEN_LANE")
    os.environ["BBAI_PREGEN_LANE"] = "1" if lane else "0"
    try:
        env = BatchedBabyAIEnv("BabyAI-%s-v0" % name, n, device=gpu, seeds=4000)
    finally:
        if old is None:
            del os.environ["BBAI_PREGEN_LANE"]
        else:
            os.environ["BBAI_PREGEN_LANE"] = old
    return env


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    if not np.array_equal(a, b):
        bad = np.nonzero((a != b).reshape(len(a), -1).any(axis=1))[0]
        raise AssertionError("%s: %d envs differ, first %s" % (what, len(bad), bad[:8].tolist()))


def _run(name, n, steps, gpu, switch=False):
    import torch
    a = _make(name, n, True, gpu)
    b = _make(name, n, False, gpu)
    if not a.get_option("inplace") == b.get_option("inplace"):
        pytest.skip("layouts differ")
    if a.get_option("pregen_lane") != 1:
        uncovered = a.get_option("inplace") == 1 and a.get_option("cplane") == 0
        a.close(); b.close()
        if uncovered:       # (BBAI_INPLACE=1 forced on a kind whose default is the classic layout: in place, the lane generator needs the C plane row of the
            pytest.skip("%s in the in-place layout without a C plane: not covered by the lane generator (bbai_create)" % name)      # small single rooms)
        raise AssertionError("%s: BBAI_PREGEN_LANE=1 did not select the lane generator" % name)
    assert b.get_option("pregen_lane") == 0
    rng = np.random.RandomState(11)
    for rep in range(3):
        a.reset(); b.reset()
        ra, ha, _ = a.export_state()
        rb, hb, _ = b.export_state()
        _same(ra, rb, "%s reset %d records" % (name, rep))
        _same(ha, hb, "%s reset %d hot" % (name, rep))
        _same(a.image.cpu().numpy(), b.image.cpu().numpy(), "%s reset %d first observation" % (name, rep))
        assert a.missions() == b.missions(), (name, rep)
    for t in range(steps):
        act = rng.choice(8, size=n, p=[0.13, 0.13, 0.3, 0.1, 0.08, 0.12, 0.02, 0.12]).astype(np.uint8)     # 7 = "reset this env now"
        ta = torch.as_tensor(act, device=gpu)
        a.step(ta); b.step(ta)
        _same(a.image.cpu().numpy(), b.image.cpu().numpy(), "%s step %d image" % (name, t))
        _same(a.reward.cpu().numpy().view(np.uint32), b.reward.cpu().numpy().view(np.uint32), "%s step %d reward" % (name, t))
        _same(a.done.cpu().numpy(), b.done.cpu().numpy(), "%s step %d done" % (name, t))
        if switch and t in (steps // 3, 2 * steps // 3):      # the lane handle changes generators (canonical RNG form and back)
            a.set_option("pregen_lane", 0 if a.get_option("pregen_lane") else 1)
    ra, ha, _ = a.export_state()
    rb, hb, _ = b.export_state()
    _same(ra, rb, "%s final records" % name)
    assert a.generator_failures() == 0 and b.generator_failures() == 0
    assert a.reset_count() == b.reset_count() and a.reset_count() > n
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", covered())
def test_lane_generator_equals_group_generator_on_device(gpu, name):
    _run(name, 1024, 200, gpu)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["SynthS5R2", "PickupLoc", "GoTo", "BossLevel"])
def test_generator_switch_in_mid_run(gpu, name):
    _run(name, 512, 240, gpu, switch=True)


@pytest.mark.gpu
def test_checkpoint_carries_the_canonical_rng_form(gpu):
    """A checkpoint taken from a lane-generator handle continues identically in a lane-group handle, and the other way round."""
    import torch
    for first, second in ((True, False), (False, True)):
        a = _make("SynthS5R2", 256, first, gpu)
        ref = _make("SynthS5R2", 256, first, gpu)
        rng = np.random.RandomState(3)
        acts = [rng.choice(8, size=256, p=[0.13, 0.13, 0.3, 0.1, 0.08, 0.12, 0.02, 0.12]).astype(np.uint8) for _ in range(160)]
        a.reset(); ref.reset()
        for t in range(80):
            a.step(torch.as_tensor(acts[t], device=gpu)); ref.step(torch.as_tensor(acts[t], device=gpu))
        blob = a.save_checkpoint()
        b = _make("SynthS5R2", 256, second, gpu)
        b.load_checkpoint(blob)
        for t in range(80, 160):
            b.step(torch.as_tensor(acts[t], device=gpu)); ref.step(torch.as_tensor(acts[t], device=gpu))
            _same(b.image.cpu().numpy(), ref.image.cpu().numpy(), "step %d" % t)
            _same(b.done.cpu().numpy(), ref.done.cpu().numpy(), "step %d" % t)
        a.close(); b.close(); ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(LEVELS))
def test_device_generator_equals_host_build_at_scale(gpu, name):
    """Every registered level, 1 024 envs x 3 consecutive levels: the records and poses the device's generator (whichever kernel serves the
    kind) leaves behind a reset() against the host build of the lane-group generator, which the CPU suite pins to the oracle and the
    reference.  (The per-level oracle tests use a dozen envs; the compiler fault described above shows up in 3 of 1 024.)"""
    from babyai_amd.engine import BatchedBabyAIEnv
    from hostsim_util import HostEnv
    n = 1024
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % name, n, device=gpu, seeds=4000)
    cfg = make_cfg(name)
    hosts = [HostEnv(cfg, 4000 + i) for i in range(n)]
    for rep in range(3):
        env.reset()
        rec, hot, _ = env.export_state()
        img = env.image.cpu().numpy()
        bad = []
        for i, h in enumerate(hosts):
            himg = h.reset()
            if not (np.array_equal(rec[i], h.rec) and np.array_equal(np.asarray(hot[i]).view(np.uint8)[:8], h.hot[:8]) and np.array_equal(img[i], himg)):
                bad.append(i)
        assert not bad, "%s level %d: %d of %d envs differ from the host build, first %s" % (name, rep, len(bad), n, bad[:8])
    assert env.generator_failures() == 0
    env.close()
