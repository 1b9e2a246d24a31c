"""k_step's view pipeline in registers (babyai_amd/csrc/bbai_view.hpp: view_cells_perm + encode_cells -- byte permutes on packed dwords,
SWAR opacity, dot-product row masks) against the straightforward per-cell observation (bbai_step.hpp observe_env, itself pinned to the
reference's golden traces): host build of the very same header, the gfx950 builtins emulated bit for bit.  Random grids of every
appearance byte the engine can produce, every pose and direction whose window fits the plane, every byte phase of the window fetch,
every carried object, and the in-flight patch of the front cell (an object action of the same step)."""
import ctypes

import numpy as np
import pytest

from babyai_amd.levels import make_cfg
from hostsim_util import lib

TYPES = [1, 2, 4, 5, 6, 7]


def random_cells(rng, shape):
    t = rng.choice(TYPES, size=shape, p=[0.45, 0.2, 0.1, 0.08, 0.09, 0.08])
    c = rng.randint(0, 6, size=shape)
    s = np.where(t == 4, rng.randint(0, 3, size=shape), 0)
    c = np.where(t == 1, 0, np.where(t == 2, 5, c))
    return (t | (c << 3) | (s << 6)).astype(np.uint8)


@pytest.mark.parametrize("level", ["GoToLocal", "GoTo", "BossLevel", "GoToObjS4"])
def test_register_view_equals_per_cell_observation(level):
    L = lib()
    cfg = make_cfg(level)
    rng = np.random.RandomState(7)
    rec = np.zeros(cfg.rec_bytes, np.uint8)
    hot = np.zeros(16, np.uint8)
    a, b = np.zeros(147, np.uint8), np.zeros(147, np.uint8)
    checked = 0
    for trial in range(60):
        plane = random_cells(rng, (cfg.EH, cfg.ES))
        rec[:cfg.ES * cfg.EH] = plane.reshape(-1)
        rec[cfg.off_app:cfg.off_app + cfg.maxo] = random_cells(rng, (cfg.maxo,))
        for _ in range(120):
            hot[0], hot[1] = rng.randint(1, cfg.W - 1), rng.randint(1, cfg.H - 1)       # every interior cell (the agent never stands on the outer wall:
                                                                                        # the 5-cell margin covers exactly these poses): every byte phase of the fetch
            hot[2] = rng.randint(0, 4)
            hot[3] = 0xFF if rng.rand() < 0.5 else rng.randint(0, cfg.maxo)
            L.hs_observe(ctypes.byref(cfg), rec.ctypes.data, hot.ctypes.data, a.ctypes.data)
            fe2 = L.hs_observe_perm(ctypes.byref(cfg), rec.ctypes.data, hot.ctypes.data, -1, b.ctypes.data)
            assert np.array_equal(a, b), (level, trial, hot[:4], a.reshape(7, 7, 3)[..., 0].T, b.reshape(7, 7, 3)[..., 0].T)
            fx = int(hot[0]) + (1, 0, -1, 0)[hot[2]]
            fy = int(hot[1]) + (0, 1, 0, -1)[hot[2]]
            assert fe2 == plane[fy + 5, fx + 5]
            # an object action changed the front cell after the window was fetched: patched in flight == patched in the plane
            nfe = int(random_cells(rng, (1,))[0])
            fe2 = L.hs_observe_perm(ctypes.byref(cfg), rec.ctypes.data, hot.ctypes.data, nfe, b.ctypes.data)
            old = rec[(fy + 5) * cfg.ES + fx + 5]
            rec[(fy + 5) * cfg.ES + fx + 5] = nfe
            L.hs_observe(ctypes.byref(cfg), rec.ctypes.data, hot.ctypes.data, a.ctypes.data)
            rec[(fy + 5) * cfg.ES + fx + 5] = old
            assert fe2 == nfe and np.array_equal(a, b), (level, trial, hot[:4], nfe)
            checked += 2
    assert checked == 60 * 120 * 2


@pytest.mark.parametrize("level", ["GoToLocal", "PickupLoc", "GoToObjS4", "GoToRedBallGrey", "GoToLocalS5N2", "GoToLocalS7N5", "PutNextLocal", "1RoomS8", "TestLotsOfBlockers", "PickupDist"])
def test_compact_plane_view_equals_per_cell_observation(level):
    """The small single rooms' C plane (bbai_types.hpp): the observation built from the env's 64-byte plane row alone -- wall fill for
    everything outside the grid, three dwords of a virtual row picked and aligned by the window's x origin -- equals observe_env on the
    record, for every pose / direction / carried object and an in-flight patch of the front cell."""
    L = lib()
    cfg = make_cfg(level)
    nb = L.hs_cpl_ok(ctypes.byref(cfg))
    assert nb in (80, 96), (level, cfg.W, cfg.H, cfg.maxo)
    rng = np.random.RandomState(11)
    rec = np.zeros(cfg.rec_bytes, np.uint8)
    hot = np.zeros(16, np.uint8)
    a, b, row = np.zeros(147, np.uint8), np.zeros(147, np.uint8), np.zeros(nb, np.uint8)
    wall = 2 | (5 << 3)
    for trial in range(40):
        plane = np.full((cfg.EH, cfg.ES), wall, np.uint8)
        inner = random_cells(rng, (cfg.H, cfg.W))
        inner[0, :] = inner[-1, :] = wall                  # the outer wall is a wall (everything outside the grid reads as one)
        inner[:, 0] = inner[:, -1] = wall
        plane[5:5 + cfg.H, 5:5 + cfg.W] = inner
        rec[:cfg.ES * cfg.EH] = plane.reshape(-1)
        rec[cfg.off_app:cfg.off_app + cfg.maxo] = random_cells(rng, (cfg.maxo,))
        # object table: a few objects on the grid (id plane says so), one recorded at a cell that another object holds now, one off the grid
        rec[cfg.off_I:cfg.off_I + cfg.W * cfg.H] = 0
        cells = [(x, y) for x in range(1, cfg.W - 1) for y in range(1, cfg.H - 1)]
        rng.shuffle(cells)
        k_on = min(cfg.maxo - 2, len(cells), 5)
        for k in range(k_on):
            x, y = cells[k]
            rec[cfg.off_pos + 2 * k], rec[cfg.off_pos + 2 * k + 1] = x, y
            rec[cfg.off_I + y * cfg.W + x] = k + 2
        rec[cfg.off_pos + 2 * k_on], rec[cfg.off_pos + 2 * k_on + 1] = cells[0]          # stale position (the cell belongs to object 0)
        rec[cfg.off_pos + 2 * (k_on + 1)], rec[cfg.off_pos + 2 * (k_on + 1) + 1] = 255, 255
        for _ in range(80):
            hot[0], hot[1] = rng.randint(1, cfg.W - 1), rng.randint(1, cfg.H - 1)
            hot[2] = rng.randint(0, 4)
            hot[3] = 0xFF if rng.rand() < 0.5 else rng.randint(0, cfg.maxo)
            L.hs_observe(ctypes.byref(cfg), rec.ctypes.data, hot.ctypes.data, a.ctypes.data)
            fe2 = L.hs_observe_cpl(ctypes.byref(cfg), rec.ctypes.data, hot.ctypes.data, -1, b.ctypes.data, row.ctypes.data)
            assert np.array_equal(a, b), (level, trial, hot[:4], a.reshape(7, 7, 3)[..., 0].T, b.reshape(7, 7, 3)[..., 0].T)
            fx = int(hot[0]) + (1, 0, -1, 0)[hot[2]]
            fy = int(hot[1]) + (0, 1, 0, -1)[hot[2]]
            assert fe2 == plane[fy + 5, fx + 5]
            nfe = int(random_cells(rng, (1,))[0])
            fe2 = L.hs_observe_cpl(ctypes.byref(cfg), rec.ctypes.data, hot.ctypes.data, nfe, b.ctypes.data, None)
            old = rec[(fy + 5) * cfg.ES + fx + 5]
            rec[(fy + 5) * cfg.ES + fx + 5] = nfe
            L.hs_observe(ctypes.byref(cfg), rec.ctypes.data, hot.ctypes.data, a.ctypes.data)
            rec[(fy + 5) * cfg.ES + fx + 5] = old
            assert fe2 == nfe and np.array_equal(a, b), (level, trial, hot[:4], nfe)
        # the row itself: plane bytes and the id bytes (on the grid iff the id plane holds the object at its recorded position)
        assert np.array_equal(row[:64].reshape(8, 8)[:cfg.H, :cfg.W], inner) and (row[:64].reshape(8, 8)[cfg.H:] == wall).all() and (row[:64].reshape(8, 8)[:, cfg.W:] == wall).all()
        ids = row[64:]
        for k in range(len(ids)):
            want = (cells[k][1] << 3 | cells[k][0]) if k < k_on else 0xFF
            assert ids[k] == want, (level, k, ids)
        # the id lookup: every cell of the grid
        for y in range(cfg.H):
            for x in range(cfg.W):
                got = L.hs_cid_lookup(ids.ctypes.data, len(ids), y << 3 | x)
                want = int(rec[cfg.off_I + y * cfg.W + x])
                assert got == (want if want >= 2 else 0), (level, x, y, got, want)
