"""Seeding + bit stream: the engine's host seeding (sha512 -> init_by_array) and its MT19937 /
randint / uniform restatement against numpy's legacy RandomState seeded the gym way."""
import ctypes

import numpy as np

from oracle import refenv

refenv.enable_shim()
from gym.utils import seeding  # noqa: E402
from hostsim_util import lib  # noqa: E402


def test_seed_state_matches_numpy():
    L = lib()
    for seed in [0, 1, 2, 7, 1337, 99999, 2 ** 31, 2 ** 40 + 12345, 2 ** 63 + 5]:
        rng, _ = seeding.np_random(seed)
        state = rng.get_state()
        mt = np.zeros(624, np.uint32)
        L.hs_seed(ctypes.c_uint64(seed), mt.ctypes.data)
        assert np.array_equal(mt, state[1]), seed
        assert state[2] == 624


def test_single_word_key_branch():
    """hash words with a zero high word collapse to a 1-word key (_int_list_from_bigint)."""
    from babyai_amd import levels  # noqa: F401  (import check only)
    mt = np.zeros(624, np.uint32)
    rs = np.random.RandomState()
    rs.seed([123456789])
    # not reachable through a seed we can choose cheaply; check init_by_array(len 1) == numpy via the 2-word path's code
    # by seeding numpy with [lo, 0] vs [lo]: they differ, which is why the branch exists.
    rs2 = np.random.RandomState()
    rs2.seed([123456789, 0])
    assert not np.array_equal(rs.get_state()[1], rs2.get_state()[1])
