"""Host logic check: the engine's C++ core (the same headers the HIP kernels compile), built for
the CPU with one lane, against the stand-alone oracle -- every supported level, generator draw for
draw (grid bytes, agent pose, mission, max_steps) and step for step (obs, reward bits, done)."""
import random

import numpy as np
import pytest

from babyai_amd.levels import LEVELS, make_cfg
from oracle import levels as olevels
from hostsim_util import HostEnv


def grid_bytes(env):
    g = env.grid.encode()
    return (g[:, :, 0] | (g[:, :, 1] << 3) | (g[:, :, 2] << 6)).T


def run(name, seed, episodes, max_len=400):
    ref = olevels.make_env(name)
    ref.seed(seed)
    sim = HostEnv(make_cfg(name), seed)
    rng = random.Random(seed)
    steps = 0
    for ep in range(episodes):
        o = ref.reset()
        img = sim.reset()
        assert sim.agent == (ref.agent_pos[0], ref.agent_pos[1], ref.agent_dir)
        assert sim.mission == ref.mission
        assert sim.max_steps == ref.max_steps
        assert np.array_equal(sim.grid_bytes(), grid_bytes(ref))
        assert np.array_equal(img, o["image"])
        for t in range(max_len):
            a = rng.randint(0, 6)
            o, r, d, _ = ref.step(a)
            img, rew, done = sim.step(a)
            steps += 1
            assert done == bool(d), (name, seed, ep, t)
            assert np.float32(r).view(np.uint32) == rew.view(np.uint32), (name, seed, ep, t)
            assert np.array_equal(img, o["image"]), (name, seed, ep, t, a)
            assert sim.agent[2] == o["direction"]
            if d:
                break
    return steps


@pytest.mark.parametrize("name", sorted(LEVELS))
def test_core_matches_oracle(name):
    for seed in (3, 11):
        run(name, seed, episodes=2)


def test_tables_agree():
    """babyai_amd/levels.py and oracle/levels.py were written independently; they must agree."""
    assert set(LEVELS) == set(olevels.SPECS)
    script_ids = {"goto_redblue_ball": 1, "open_red_door": 2, "open_door": 3, "goto_door": 4, "goto_obj_door": 5,
                  "action_obj_door": 6, "unlock_local": 7, "key_in_box": 8, "unlock_pickup": 9, "blocked_unlock_pickup": 10,
                  "unlock_to_unlock": 11, "pickup_dist": 12, "pickup_above": 13, "open_two_doors": 14, "find_obj": 15,
                  "key_corridor": 16, "one_room": 17, "put_next": 18, "move_two_across": 19, "open_doors_order": 20}
    for name, (fam, kw) in olevels.SPECS.items():
        p = LEVELS[name]
        assert p["kind"] == {"goto": 0, "levelgen": 1, "bonus": 2, "fixed": 2}[fam]
        if fam == "fixed":
            fixed_ids = {"goto_blocked": 21, "putnext_blocked": 22, "putnext_door1": 23, "putnext_door2": 24,
                         "putnext_identical": 25, "unblocking_loop": 26, "putnext_close_door": 27, "lots_of_blockers": 28}
            assert p["script"] == fixed_ids[kw["script"]], name
            for k, dflt in (("room_size", 9), ("num_rows", 1), ("num_cols", 1)):
                assert p[k] == kw.get(k, dflt), (name, k)
            continue
        if fam == "bonus":
            assert p["script"] == script_ids[kw["script"]], name
            for k, dflt in (("room_size", 8), ("num_rows", 3), ("num_cols", 3), ("num_dists", 0)):
                assert p[k] == kw.get(k, dflt), (name, k)
            assert tuple(p["sp"]) == tuple(kw.get("sp", ())), name
            continue
        env_defaults = dict(room_size=8, num_rows=1 if fam == "goto" else 3, num_cols=1 if fam == "goto" else 3)
        for k in ("room_size", "num_rows", "num_cols"):
            assert p[k] == kw.get(k, env_defaults[k]), (name, k)
        assert p["num_dists"] == kw.get("num_dists", 8 if fam == "goto" else 18), name
        if fam == "levelgen":
            assert tuple(p["action_kinds"]) == tuple(kw.get("action_kinds", ("goto", "pickup", "open", "putnext")))
            assert tuple(p["instr_kinds"]) == tuple(kw.get("instr_kinds", ("action", "and", "seq")))
            assert p["locked_room_prob"] == float(kw.get("locked_room_prob", 0.5))
            for k, dflt in (("locations", True), ("unblocking", True), ("implicit_unlock", True)):
                assert bool(p[k]) == bool(kw.get(k, dflt)), (name, k)
        else:
            for k, dflt in (("redball", False), ("connect", False), ("doors_open", False), ("all_unique", False),
                            ("lock", False), ("lock_color_excl", False), ("dists_per_room", False), ("grey_dists", False)):
                assert bool(p[k]) == bool(kw.get(k, dflt)), (name, k)
            assert p["check_reach"] == int(kw.get("check_reach", True)), name
            assert p["instr"] == {"goto": 1, "pickup": 2, "open": 3, "putnext": 4}[kw.get("instr", "goto")], name
            tg = kw.get("target", "redball" if kw.get("redball") else "dist")
            assert p["target"] == {"redball": 0, "dist": 1, "door": 2, "two_dists": 3, "locked_door": 4,
                                   "locked_room_obj": 5}[tg], name


def test_max_mission_tokens_bounds_every_level():
    """babyai_amd.missions.max_mission_tokens(cfg) -- the fixed instruction width DeviceRollout feeds the model -- is an
    upper bound of the token count of generated missions on every registered level, and not a wild one."""
    from babyai_amd.levels import LEVELS, make_cfg
    from babyai_amd.missions import max_mission_tokens, tokenize
    from hostsim_util import HostEnv
    for name in sorted(LEVELS):
        cfg = make_cfg(name)
        bound = max_mission_tokens(cfg)
        longest = 0
        for seed in range(40):
            h = HostEnv(cfg, 9000 + seed)
            for _ in range(3):
                h.reset()
                longest = max(longest, len(tokenize(h.mission)))
        assert longest <= bound <= 72, (name, longest, bound)


def test_row_packer_lays_rows_out_at_the_output_pitch():
    """k_step parks its 256 observation rows in LDS at the 147-byte OUTPUT pitch (bbai_step.hpp RowPacker): every lane
    shifts its 37 dwords by the row's byte phase and writes aligned dwords + byte-sized head / tail pieces.  Lanes run in
    any order without a barrier, so no lane may write a byte outside its own row -- including its window scratch."""
    import ctypes
    import numpy as np
    from hostsim_util import lib
    L = lib()
    L.hs_pack_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    L.hs_row_scratch.argtypes = [ctypes.c_int]
    rng = np.random.RandomState(7)
    n = 256
    for trial in range(8):
        rows = rng.randint(0, 256, size=(n, 148)).astype(np.uint8)
        rows[:, 147] = 0                                   # dword 36 = three bytes + a zero byte
        order = rng.permutation(n).astype(np.int32)
        lds = np.full(16 + n * 147 + 16, 0xA5, np.uint8)
        L.hs_pack_rows(np.ascontiguousarray(rows).view(np.uint32).ctypes.data, order.ctypes.data, n, lds.ctypes.data, trial & 1)
        assert np.array_equal(lds[16:16 + n * 147].reshape(n, 147), rows[:, :147])
        assert (lds[:16] == 0xA5).all() and (lds[16 + n * 147:] == 0xA5).all()     # nothing before row 0 / after row 255
    for r in range(n):
        s = L.hs_row_scratch(r)
        assert s % 4 == 0 and r * 147 <= s and s + 56 <= (r + 1) * 147


@pytest.mark.parametrize("name", sorted(LEVELS))
def test_k_step_order_of_operations_equals_the_reference_order(name):
    """Every level, object-action-heavy random play: the step in k_step's order (pose -> front-cell id and appearance taken
    BEFORE the object actions and corrected by what they wrote -> verifier on those; bbai_step.hpp step_env_prefetch) must
    leave the same record, hot state, stale set, reward and done as the reference order (step_env), step after step."""
    import ctypes
    from babyai_amd.levels import make_cfg
    from hostsim_util import HostEnv
    rng = np.random.RandomState(sum(map(ord, name)))
    cfg = make_cfg(name)
    for seed in (3, 4):
        a, b = HostEnv(cfg, 9000 + seed), HostEnv(cfg, 9000 + seed)
        b.prefetch_order = True
        a.reset()
        b.reset()
        for t in range(260):
            act = int(rng.choice(7, p=[0.12, 0.12, 0.28, 0.16, 0.12, 0.16, 0.04]))
            ia, ra, da = a.step(act)
            ib, rb, db = b.step(act)
            assert da == db and a.last_reward64 == b.last_reward64, (name, seed, t)
            assert np.array_equal(a.rec, b.rec) and np.array_equal(a.hot, b.hot) and a.stale.value == b.stale.value, (name, seed, t)
            assert np.array_equal(ia, ib)
            if da:
                a.reset()
                b.reset()
