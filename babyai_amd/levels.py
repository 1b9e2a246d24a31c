"""Level table: BabyAI level ids -> generator configuration (`LevelCfg`).

Each entry restates the constructor arguments of the reference level class
(file:line under /root/reference/babyai/levels/iclr19_levels.py) as plain data for
the device-side generator (babyai_amd/csrc/bbai_gen.hpp).  Levels are grouped in two
generator families:

  * K_GOTO      GoToRedBall :40-63, GoToObj :75-102, GoToLocal :105-184, GoTo :224-301
  * K_LEVELGEN  every `LevelGen` parameterisation (levelgen.py:256-460): PickupLoc :494-515,
                GoToSeq :518-551, Synth :554-594, SynthLoc :597-614, SynthSeq :617-633,
                MiniBossLevel :636-645, BossLevel :648-652, BossLevelNoUnlock :655-661

  * the hand-written single-instruction levels GoToRedBallGrey :10-37, PutNextLocal :187-221,
    GoToImpUnlock :304-357, Pickup :360-371, UnblockPickup :374-391, Open :394-415, Unlock :418-474,
    PutNext :477-491 are cfg variants of the K_GOTO generator
  * K_BONUS     the 50 levels of bonus_levels.py: one gen_mission script id (BS_*) per level class with
                up to four integer parameters (device twin: Gen::mission_bonus, bbai_gen.hpp)
                and the 8 fixed regression layouts of test_levels.py
=> every level the reference registers (105 ids) is covered; `make_cfg` raises KeyError for anything else.
"""
import ctypes

K_GOTO, K_LEVELGEN, K_BONUS = 0, 1, 2
AK = {"goto": 0, "pickup": 1, "open": 2, "putnext": 3}
IK = {"action": 0, "and": 1, "seq": 2}


class LevelCfg(ctypes.Structure):
    """Mirror of `bbai::LevelCfg` (babyai_amd/csrc/bbai_types.hpp) == `bbai_level_cfg` (include/bbai.h)."""
    _fields_ = [
        ("kind", ctypes.c_int32),
        ("room_size", ctypes.c_int32), ("num_rows", ctypes.c_int32), ("num_cols", ctypes.c_int32),
        ("num_dists", ctypes.c_int32),
        ("redball", ctypes.c_int32), ("connect", ctypes.c_int32), ("check_reach", ctypes.c_int32),
        ("doors_open", ctypes.c_int32), ("all_unique", ctypes.c_int32),
        ("instr", ctypes.c_int32), ("target", ctypes.c_int32), ("lock", ctypes.c_int32),
        ("lock_color_excl", ctypes.c_int32), ("dists_per_room", ctypes.c_int32), ("grey_dists", ctypes.c_int32),
        ("script", ctypes.c_int32), ("sp", ctypes.c_int32 * 4),
        ("locations", ctypes.c_int32), ("unblocking", ctypes.c_int32), ("implicit_unlock", ctypes.c_int32),
        ("n_action_kinds", ctypes.c_int32), ("action_kinds", ctypes.c_int32 * 4),
        ("n_instr_kinds", ctypes.c_int32), ("instr_kinds", ctypes.c_int32 * 3),
        ("locked_room_prob", ctypes.c_double),
        ("W", ctypes.c_int32), ("H", ctypes.c_int32), ("ES", ctypes.c_int32), ("EH", ctypes.c_int32),
        ("maxo", ctypes.c_int32),
        ("off_I", ctypes.c_int32), ("off_app", ctypes.c_int32), ("off_pos", ctypes.c_int32),
        ("off_cont", ctypes.c_int32), ("off_prog", ctypes.c_int32), ("rec_bytes", ctypes.c_int32),
    ]


L_GOTO, L_PICKUP, L_OPEN, L_PUTNEXT = 1, 2, 3, 4
TG_REDBALL, TG_DIST, TG_DOOR, TG_TWO_DISTS, TG_LOCKED_DOOR, TG_LOCKED_ROOM_OBJ = 0, 1, 2, 3, 4, 5


def _goto(room_size=8, num_rows=1, num_cols=1, num_dists=8, redball=0, connect=0,
          check_reach=1, doors_open=0, all_unique=0, instr=L_GOTO, target=None, lock=0,
          lock_color_excl=0, dists_per_room=0, grey_dists=0):
    if target is None:
        target = TG_REDBALL if redball else TG_DIST
    return dict(kind=K_GOTO, room_size=room_size, num_rows=num_rows, num_cols=num_cols,
                num_dists=num_dists, redball=redball, connect=connect, check_reach=check_reach,
                doors_open=doors_open, all_unique=all_unique, instr=instr, target=target, lock=lock,
                lock_color_excl=lock_color_excl, dists_per_room=dists_per_room, grey_dists=grey_dists)


def _maze(**kw):
    """RoomGridLevel defaults of the hand-written maze levels: 3x3 rooms of size 8."""
    return _goto(num_rows=3, num_cols=3, **kw)


def _levelgen(room_size=8, num_rows=3, num_cols=3, num_dists=18, locked_room_prob=0.5,
              locations=True, unblocking=True, implicit_unlock=True,
              action_kinds=("goto", "pickup", "open", "putnext"),
              instr_kinds=("action", "and", "seq")):
    return dict(kind=K_LEVELGEN, room_size=room_size, num_rows=num_rows, num_cols=num_cols,
                num_dists=num_dists, locked_room_prob=float(locked_room_prob),
                locations=int(locations), unblocking=int(unblocking),
                implicit_unlock=int(implicit_unlock),
                action_kinds=tuple(action_kinds), instr_kinds=tuple(instr_kinds))


(BS_GOTO_REDBLUE_BALL, BS_OPEN_RED_DOOR, BS_OPEN_DOOR, BS_GOTO_DOOR, BS_GOTO_OBJ_DOOR, BS_ACTION_OBJ_DOOR,
 BS_UNLOCK_LOCAL, BS_KEY_IN_BOX, BS_UNLOCK_PICKUP, BS_BLOCKED_UNLOCK_PICKUP, BS_UNLOCK_TO_UNLOCK, BS_PICKUP_DIST,
 BS_PICKUP_ABOVE, BS_OPEN_TWO_DOORS, BS_FIND_OBJ, BS_KEY_CORRIDOR, BS_ONE_ROOM, BS_PUT_NEXT, BS_MOVE_TWO_ACROSS,
 BS_OPEN_DOORS_ORDER, BS_TEST_GOTO_BLOCKED, BS_TEST_PUTNEXT_BLOCKED, BS_TEST_PUTNEXT_DOOR1, BS_TEST_PUTNEXT_DOOR2,
 BS_TEST_PUTNEXT_IDENTICAL, BS_TEST_UNBLOCKING_LOOP, BS_TEST_PUTNEXT_CLOSE_DOOR, BS_TEST_LOTS_OF_BLOCKERS) = range(1, 29)
_COLOR_IDX = {"red": 0, "green": 1, "blue": 2, "purple": 3, "yellow": 4, "grey": 5}


def _bonus(script, room_size=8, num_rows=3, num_cols=3, num_dists=0, sp=()):
    """A hand-written gen_mission of bonus_levels.py (device twin: Gen::mission_bonus)."""
    return dict(kind=K_BONUS, script=script, room_size=room_size, num_rows=num_rows, num_cols=num_cols,
                num_dists=num_dists, sp=tuple(sp))


LEVELS = {
    # --- K_GOTO family -----------------------------------------------------------------
    "GoToRedBallGrey": _goto(num_dists=7, redball=1, grey_dists=1),
    "GoToRedBall": _goto(num_dists=7, redball=1),
    "GoToRedBallNoDists": _goto(num_dists=0, redball=1),
    "GoToObj": _goto(num_dists=1, all_unique=1, check_reach=0),
    "GoToObjS4": _goto(room_size=4, num_dists=1, all_unique=1, check_reach=0),
    "GoToObjS6": _goto(room_size=6, num_dists=1, all_unique=1, check_reach=0),
    "GoToLocal": _goto(num_dists=8),
    "GoToLocalS5N2": _goto(room_size=5, num_dists=2),
    "GoToLocalS6N2": _goto(room_size=6, num_dists=2),
    "GoToLocalS6N3": _goto(room_size=6, num_dists=3),
    "GoToLocalS6N4": _goto(room_size=6, num_dists=4),
    "GoToLocalS7N4": _goto(room_size=7, num_dists=4),
    "GoToLocalS7N5": _goto(room_size=7, num_dists=5),
    "GoToLocalS8N2": _goto(num_dists=2),
    "GoToLocalS8N3": _goto(num_dists=3),
    "GoToLocalS8N4": _goto(num_dists=4),
    "GoToLocalS8N5": _goto(num_dists=5),
    "GoToLocalS8N6": _goto(num_dists=6),
    "GoToLocalS8N7": _goto(num_dists=7),
    "GoTo": _goto(num_rows=3, num_cols=3, num_dists=18, connect=1),
    "GoToOpen": _goto(num_rows=3, num_cols=3, num_dists=18, connect=1, doors_open=1),
    "GoToObjMaze": _goto(num_rows=3, num_cols=3, num_dists=1, connect=1),
    "GoToObjMazeOpen": _goto(num_rows=3, num_cols=3, num_dists=1, connect=1, doors_open=1),
    "GoToObjMazeS4R2": _goto(room_size=4, num_rows=2, num_cols=2, num_dists=1, connect=1),
    "GoToObjMazeS4": _goto(room_size=4, num_rows=3, num_cols=3, num_dists=1, connect=1),
    "GoToObjMazeS5": _goto(room_size=5, num_rows=3, num_cols=3, num_dists=1, connect=1),
    "GoToObjMazeS6": _goto(room_size=6, num_rows=3, num_cols=3, num_dists=1, connect=1),
    "GoToObjMazeS7": _goto(room_size=7, num_rows=3, num_cols=3, num_dists=1, connect=1),
    # hand-written single-instruction levels (iclr19_levels.py:187-221, 304-491)
    "PutNextLocal": _goto(num_dists=8, all_unique=1, instr=L_PUTNEXT, target=TG_TWO_DISTS),
    "PutNextLocalS5N3": _goto(room_size=5, num_dists=3, all_unique=1, instr=L_PUTNEXT, target=TG_TWO_DISTS),
    "PutNextLocalS6N4": _goto(room_size=6, num_dists=4, all_unique=1, instr=L_PUTNEXT, target=TG_TWO_DISTS),
    "Pickup": _maze(num_dists=18, connect=1, instr=L_PICKUP),
    "UnblockPickup": _maze(num_dists=20, connect=1, check_reach=2, instr=L_PICKUP),
    "Open": _maze(num_dists=18, connect=1, instr=L_OPEN, target=TG_DOOR),
    "PutNext": _maze(num_dists=18, connect=1, instr=L_PUTNEXT, target=TG_TWO_DISTS),
    "Unlock": _maze(num_dists=3, connect=1, lock=1, lock_color_excl=1, dists_per_room=1, instr=L_OPEN,
                    target=TG_LOCKED_DOOR),
    "GoToImpUnlock": _maze(num_dists=2, connect=1, lock=1, dists_per_room=1, instr=L_GOTO,
                           target=TG_LOCKED_ROOM_OBJ),
    # --- bonus levels (bonus_levels.py; file:line next to each case of Gen::mission_bonus) ------------
    "GoToRedBlueBall": _bonus(BS_GOTO_REDBLUE_BALL, num_rows=1, num_cols=1, num_dists=7),
    "OpenRedDoor": _bonus(BS_OPEN_RED_DOOR, room_size=5, num_rows=1, num_cols=2),
    "OpenDoor": _bonus(BS_OPEN_DOOR, sp=(0, 0)),
    "OpenDoorDebug": _bonus(BS_OPEN_DOOR, sp=(0, 1)),
    "OpenDoorColor": _bonus(BS_OPEN_DOOR, sp=(1, 0)),
    "OpenDoorLoc": _bonus(BS_OPEN_DOOR, sp=(2, 0)),
    "GoToDoor": _bonus(BS_GOTO_DOOR, room_size=7),
    "GoToObjDoor": _bonus(BS_GOTO_OBJ_DOOR, room_size=8),
    "ActionObjDoor": _bonus(BS_ACTION_OBJ_DOOR, room_size=7),
    "UnlockLocal": _bonus(BS_UNLOCK_LOCAL, sp=(0,)),
    "UnlockLocalDist": _bonus(BS_UNLOCK_LOCAL, sp=(1,)),
    "KeyInBox": _bonus(BS_KEY_IN_BOX),
    "UnlockPickup": _bonus(BS_UNLOCK_PICKUP, room_size=6, num_rows=1, num_cols=2, sp=(0,)),
    "UnlockPickupDist": _bonus(BS_UNLOCK_PICKUP, room_size=6, num_rows=1, num_cols=2, sp=(1,)),
    "BlockedUnlockPickup": _bonus(BS_BLOCKED_UNLOCK_PICKUP, room_size=6, num_rows=1, num_cols=2),
    "UnlockToUnlock": _bonus(BS_UNLOCK_TO_UNLOCK, room_size=6, num_rows=1, num_cols=3),
    "PickupDist": _bonus(BS_PICKUP_DIST, room_size=7, num_rows=1, num_cols=1, sp=(0,)),
    "PickupDistDebug": _bonus(BS_PICKUP_DIST, room_size=7, num_rows=1, num_cols=1, sp=(1,)),
    "PickupAbove": _bonus(BS_PICKUP_ABOVE, room_size=6),
    "OpenTwoDoors": _bonus(BS_OPEN_TWO_DOORS, room_size=6, sp=(0, 0, 0)),
    "OpenTwoDoorsDebug": _bonus(BS_OPEN_TWO_DOORS, room_size=6, sp=(0, 0, 1)),
    "OpenRedBlueDoors": _bonus(BS_OPEN_TWO_DOORS, room_size=6, sp=(1 + 0, 1 + 2, 0)),
    "OpenRedBlueDoorsDebug": _bonus(BS_OPEN_TWO_DOORS, room_size=6, sp=(1 + 0, 1 + 2, 1)),
    "FindObjS5": _bonus(BS_FIND_OBJ, room_size=5),
    "FindObjS6": _bonus(BS_FIND_OBJ, room_size=6),
    "FindObjS7": _bonus(BS_FIND_OBJ, room_size=7),
    "KeyCorridorS3R1": _bonus(BS_KEY_CORRIDOR, room_size=3, num_rows=1),
    "KeyCorridorS3R2": _bonus(BS_KEY_CORRIDOR, room_size=3, num_rows=2),
    "KeyCorridorS3R3": _bonus(BS_KEY_CORRIDOR, room_size=3, num_rows=3),
    "KeyCorridorS4R3": _bonus(BS_KEY_CORRIDOR, room_size=4, num_rows=3),
    "KeyCorridorS5R3": _bonus(BS_KEY_CORRIDOR, room_size=5, num_rows=3),
    "KeyCorridorS6R3": _bonus(BS_KEY_CORRIDOR, room_size=6, num_rows=3),
    "1RoomS8": _bonus(BS_ONE_ROOM, room_size=8, num_rows=1, num_cols=1),
    "1RoomS12": _bonus(BS_ONE_ROOM, room_size=12, num_rows=1, num_cols=1),
    "1RoomS16": _bonus(BS_ONE_ROOM, room_size=16, num_rows=1, num_cols=1),
    "1RoomS20": _bonus(BS_ONE_ROOM, room_size=20, num_rows=1, num_cols=1),
    "PutNextS4N1": _bonus(BS_PUT_NEXT, room_size=4, num_rows=1, num_cols=2, num_dists=1, sp=(0,)),
    "PutNextS5N1": _bonus(BS_PUT_NEXT, room_size=5, num_rows=1, num_cols=2, num_dists=1, sp=(0,)),
    "PutNextS5N2": _bonus(BS_PUT_NEXT, room_size=5, num_rows=1, num_cols=2, num_dists=2, sp=(0,)),
    "PutNextS6N3": _bonus(BS_PUT_NEXT, room_size=6, num_rows=1, num_cols=2, num_dists=3, sp=(0,)),
    "PutNextS7N4": _bonus(BS_PUT_NEXT, room_size=7, num_rows=1, num_cols=2, num_dists=4, sp=(0,)),
    "PutNextS5N2Carrying": _bonus(BS_PUT_NEXT, room_size=5, num_rows=1, num_cols=2, num_dists=2, sp=(1,)),
    "PutNextS6N3Carrying": _bonus(BS_PUT_NEXT, room_size=6, num_rows=1, num_cols=2, num_dists=3, sp=(1,)),
    "PutNextS7N4Carrying": _bonus(BS_PUT_NEXT, room_size=7, num_rows=1, num_cols=2, num_dists=4, sp=(1,)),
    "MoveTwoAcrossS5N2": _bonus(BS_MOVE_TWO_ACROSS, room_size=5, num_rows=1, num_cols=2, num_dists=2),
    "MoveTwoAcrossS8N9": _bonus(BS_MOVE_TWO_ACROSS, room_size=8, num_rows=1, num_cols=2, num_dists=9),
    "OpenDoorsOrderN2": _bonus(BS_OPEN_DOORS_ORDER, room_size=6, sp=(2, 0)),
    "OpenDoorsOrderN4": _bonus(BS_OPEN_DOORS_ORDER, room_size=6, sp=(4, 0)),
    "OpenDoorsOrderN2Debug": _bonus(BS_OPEN_DOORS_ORDER, room_size=6, sp=(2, 1)),
    "OpenDoorsOrderN4Debug": _bonus(BS_OPEN_DOORS_ORDER, room_size=6, sp=(4, 1)),
    # --- test_levels.py: hand-built regression layouts for the bot (test_levels.py:13-232) ---------------
    "TestGoToBlocked": _bonus(BS_TEST_GOTO_BLOCKED, room_size=9, num_rows=1, num_cols=1),
    "TestPutNextToBlocked": _bonus(BS_TEST_PUTNEXT_BLOCKED, room_size=9, num_rows=1, num_cols=1),
    "TestPutNextToCloseToDoor1": _bonus(BS_TEST_PUTNEXT_DOOR1, room_size=9, num_rows=2, num_cols=1),
    "TestPutNextToCloseToDoor2": _bonus(BS_TEST_PUTNEXT_DOOR2, room_size=9, num_rows=2, num_cols=1),
    "TestPutNextToIdentical": _bonus(BS_TEST_PUTNEXT_IDENTICAL, room_size=9, num_rows=1, num_cols=1),
    "TestUnblockingLoop": _bonus(BS_TEST_UNBLOCKING_LOOP, room_size=9, num_rows=2, num_cols=2),
    "TestPutNextCloseToDoor": _bonus(BS_TEST_PUTNEXT_CLOSE_DOOR, room_size=9, num_rows=2, num_cols=2),
    "TestLotsOfBlockers": _bonus(BS_TEST_LOTS_OF_BLOCKERS, room_size=8, num_rows=1, num_cols=1),
    # --- K_LEVELGEN family ---------------------------------------------------------------
    "PickupLoc": _levelgen(action_kinds=("pickup",), instr_kinds=("action",), num_rows=1, num_cols=1,
                           num_dists=8, locked_room_prob=0, locations=True, unblocking=False),
    "GoToSeq": _levelgen(action_kinds=("goto",), locked_room_prob=0, locations=False, unblocking=False),
    "GoToSeqS5R2": _levelgen(room_size=5, num_rows=2, num_cols=2, num_dists=4, action_kinds=("goto",),
                             locked_room_prob=0, locations=False, unblocking=False),
    "Synth": _levelgen(instr_kinds=("action",), locations=False, unblocking=True, implicit_unlock=False),
    "SynthS5R2": _levelgen(room_size=5, num_rows=2, num_cols=2, num_dists=7, instr_kinds=("action",),
                           locations=False, unblocking=True, implicit_unlock=False),
    "SynthLoc": _levelgen(instr_kinds=("action",), locations=True, unblocking=True, implicit_unlock=False),
    "SynthSeq": _levelgen(locations=True, unblocking=True, implicit_unlock=False),
    "MiniBossLevel": _levelgen(num_cols=2, num_rows=2, room_size=5, num_dists=7, locked_room_prob=0.25),
    "BossLevel": _levelgen(),
    "BossLevelNoUnlock": _levelgen(locked_room_prob=0, implicit_unlock=False),
}


def level_name(env_id):
    """'BabyAI-GoToLocal-v0' or 'GoToLocal' -> 'GoToLocal' (levelgen.py:480 id scheme)."""
    name = env_id
    if name.startswith("BabyAI-"):
        name = name[len("BabyAI-"):]
        if name.endswith("-v0"):
            name = name[:-3]
    return name


def fill_layout(cfg):
    """Python twin of bbai::fill_layout (bbai_types.hpp); the C library recomputes and checks it."""
    def rup(v, m):
        return (v + m - 1) // m * m
    MARGIN, PROG = 5, 112
    cfg.W = (cfg.room_size - 1) * cfg.num_cols + 1
    cfg.H = (cfg.room_size - 1) * cfg.num_rows + 1
    cfg.ES = rup(cfg.W + 2 * MARGIN, 4)
    cfg.EH = cfg.H + 2 * MARGIN
    ndoors = cfg.num_rows * (cfg.num_cols - 1) + cfg.num_cols * (cfg.num_rows - 1)
    nd = cfg.num_dists * cfg.num_rows * cfg.num_cols if cfg.dists_per_room else cfg.num_dists
    if cfg.kind == K_BONUS:
        nd = 24
    cfg.maxo = rup(nd + 2 + ndoors, 8)
    cfg.off_I = cfg.ES * cfg.EH
    cfg.off_app = rup(cfg.off_I + cfg.W * cfg.H, 4)
    cfg.off_pos = cfg.off_app + cfg.maxo
    cfg.off_cont = cfg.off_pos + 2 * cfg.maxo
    cfg.off_prog = rup(cfg.off_cont + cfg.maxo, 16)
    cfg.rec_bytes = rup(cfg.off_prog + PROG, 64)
    return cfg


def make_cfg(env_id):
    """Build the `LevelCfg` for a level id; KeyError if the level is not covered."""
    name = level_name(env_id)
    if name not in LEVELS:
        raise KeyError("level %r is not supported by the batched engine (supported: %s)"
                       % (env_id, ", ".join(sorted(LEVELS))))
    p = LEVELS[name]
    cfg = LevelCfg()
    for k, v in p.items():
        if k == "action_kinds":
            cfg.n_action_kinds = len(v)
            for i, a in enumerate(v):
                cfg.action_kinds[i] = AK[a]
        elif k == "instr_kinds":
            cfg.n_instr_kinds = len(v)
            for i, a in enumerate(v):
                cfg.instr_kinds[i] = IK[a]
        elif k == "sp":
            for i, a in enumerate(v):
                cfg.sp[i] = int(a)
        else:
            setattr(cfg, k, v)
    return fill_layout(cfg)
