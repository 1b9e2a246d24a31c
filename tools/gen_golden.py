#!/usr/bin/env python3
"""Generate golden traces from the REFERENCE ITSELF (/root/reference/babyai, imported
unmodified on top of the restated gym/gym_minigrid shim in oracle/shim).

Runs only in the build container (the reference tree cannot travel to the GPU box); the
resulting small fixtures under tests/golden/ are committed and pin
  * the stand-alone oracle (oracle/levels.py)            -> tests/test_oracle_golden.py
  * the HIP engine through the C ABI                     -> tests/test_gpu_parity.py
Re-run:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

Protocol recorded = the reference's ParallelEnv worker (babyai/rl/utils/penv.py:4-16):
  env = Level(); env.seed(s); obs = env.reset();  then for each action:
  obs, reward, done, info = env.step(a); if done: obs = env.reset()
`pre_resets` extra reset() calls are recorded first (they exercise the persistent RNG stream).
"""
import os
import signal
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refenv  # noqa: E402

refenv.import_reference()
from babyai.levels import level_dict  # noqa: E402
from gym_minigrid.wrappers import RGBImgPartialObsWrapper  # noqa: E402

# level, n_envs, n_steps, pre_resets, n_pixel_envs, expert
# expert=True: actions come from the reference's own GOFAI expert (babyai/bot.py) with 12% random
# perturbations, so episodes actually succeed and the traces cover keys, locked doors, pick-up /
# drop / put-next verification and every Seq/And combination -- things uniform random actions
# almost never reach.
PLAN = [
    ("GoToRedBall", 8, 160, 2, 0, False),
    ("GoToLocal", 16, 256, 2, 2, False),
    ("PickupLoc", 16, 256, 2, 0, False),
    ("GoTo", 8, 320, 2, 0, False),
    ("GoToSeq", 6, 200, 1, 0, False),
    ("SynthSeq", 8, 200, 3, 0, False),
    ("MiniBossLevel", 8, 320, 3, 0, False),
    ("BossLevel", 16, 384, 3, 2, False),
    ("PickupLoc", 8, 120, 0, 0, True),
    ("GoTo", 6, 300, 0, 0, True),
    ("SynthLoc", 8, 400, 0, 0, True),
    ("MiniBossLevel", 12, 500, 0, 2, True),
    ("BossLevel", 12, 900, 0, 0, True),
    ("GoToRedBallGrey", 6, 100, 1, 0, False),
    ("PutNextLocal", 8, 200, 1, 0, True),
    ("Unlock", 8, 260, 1, 0, True),
    ("GoToImpUnlock", 6, 300, 1, 0, True),
    ("Open", 6, 150, 1, 0, True),
    ("PutNext", 6, 300, 0, 0, True),
    ("UnblockPickup", 6, 250, 1, 0, True),
    ("Pickup", 4, 150, 0, 0, True),
    # bonus levels: strict ("Debug") verifiers, key inside a box, removed walls, start-carrying resets
    ("KeyInBox", 6, 200, 1, 0, True),
    ("OpenDoorsOrderN4Debug", 8, 160, 1, 0, True),
    ("PickupDistDebug", 8, 120, 1, 0, True),
    ("OpenTwoDoorsDebug", 6, 160, 0, 0, True),
    ("PutNextS6N3Carrying", 8, 200, 1, 2, True),
    ("MoveTwoAcrossS5N2", 6, 240, 0, 0, True),
    ("KeyCorridorS4R3", 6, 300, 1, 0, True),
    ("UnlockToUnlock", 6, 300, 0, 0, True),
    ("BlockedUnlockPickup", 6, 200, 0, 0, True),
    ("ActionObjDoor", 8, 120, 1, 0, True),
    ("1RoomS12", 4, 120, 0, 0, True),
    # test_levels.py fixed layouts
    ("TestUnblockingLoop", 4, 200, 1, 0, True),
    ("TestPutNextCloseToDoor", 4, 160, 0, 0, True),
    ("TestPutNextToIdentical", 3, 120, 0, 0, True),
]
SEED_BASE = 1000


class BotTimeout(BaseException):
    pass


def _on_alarm(signum, frame):
    raise BotTimeout()


class Driver(object):
    """Action source for one env: the reference bot with random perturbations, random after a bot failure."""

    def __init__(self, env, rng):
        from babyai.bot import Bot
        self.env, self.rng, self.Bot = env, rng, Bot
        self.new_episode()

    def new_episode(self):
        self.bot = self.Bot(self.env)
        self.last = None

    def act(self):
        a = None
        if self.bot is not None:
            try:
                # the reference bot can spin forever on some bonus levels: give every decision a 2 s budget
                signal.signal(signal.SIGALRM, _on_alarm)
                signal.setitimer(signal.ITIMER_REAL, 2.0)
                try:
                    a = int(self.bot.replan(self.last))
                finally:
                    signal.setitimer(signal.ITIMER_REAL, 0)
            except BaseException as exc:
                if isinstance(exc, KeyboardInterrupt):
                    raise
                self.bot = None
        if a is None or self.rng.rand() < 0.12:
            a = int(self.rng.randint(0, 7))
        self.last = a
        return a


def trace(name, n_envs, n_steps, pre_resets, n_pix, expert=False):
    rng = np.random.RandomState(sum(map(ord, name)) + (77 if expert else 0))
    actions = rng.randint(0, 7, size=(n_steps, n_envs)).astype(np.uint8)
    seeds = np.arange(n_envs, dtype=np.uint64) + SEED_BASE
    envs = []
    for s in seeds:
        env = level_dict[name]()
        # The constructor runs one reset from OS entropy; for LevelGen levels that can leave a
        # random stale `locked_room` behind (levelgen.py:284,325) which leaks into later episodes
        # when implicit_unlock=False (levelgen.py:384).  Clear it so traces are deterministic:
        # the engine and the oracle define seed() as starting from `locked_room = None`.
        if hasattr(env, 'locked_room'):
            env.locked_room = None
        env.seed(int(s))
        envs.append(env)
    pix = [RGBImgPartialObsWrapper(e) for e in envs[:n_pix]]
    pre_image = np.zeros((pre_resets, n_envs, 7, 7, 3), np.uint8)
    pre_mission = []
    for r in range(pre_resets):
        row = []
        for i, e in enumerate(envs):
            o = e.reset()
            pre_image[r, i] = o['image']
            row.append(o['mission'])
        pre_mission.append(row)
    image = np.zeros((n_steps + 1, n_envs, 7, 7, 3), np.uint8)
    direction = np.zeros((n_steps + 1, n_envs), np.uint8)
    reward = np.zeros((n_steps, n_envs), np.float32)
    reward64 = np.zeros((n_steps, n_envs), np.float64)      # the Python float the reference returns, as is
    done = np.zeros((n_steps, n_envs), np.uint8)
    max_steps = np.zeros((n_steps + 1, n_envs), np.int32)
    pixels = np.zeros((n_steps + 1, n_pix, 56, 56, 3), np.uint8)
    events = []          # (t, env, mission) at every episode start; t = index into image[]
    for i, e in enumerate(envs):
        o = e.reset()
        image[0, i] = o['image']; direction[0, i] = o['direction']; max_steps[0, i] = e.max_steps
        events.append((0, i, o['mission']))
        if i < n_pix:
            pixels[0, i] = pix[i].observation(o)['image']
    n_done = 0
    drivers = [Driver(e, rng) for e in envs] if expert else None
    for t in range(n_steps):
        for i, e in enumerate(envs):
            if expert:
                actions[t, i] = drivers[i].act()
            o, r, d, _ = e.step(int(actions[t, i]))
            reward[t, i] = np.float32(r)
            reward64[t, i] = r
            done[t, i] = d
            if d:
                o = e.reset()
                events.append((t + 1, i, o['mission']))
                n_done += 1
                if expert:
                    drivers[i].new_episode()
            image[t + 1, i] = o['image']; direction[t + 1, i] = o['direction']; max_steps[t + 1, i] = e.max_steps
            if i < n_pix:
                pixels[t + 1, i] = pix[i].observation(o)['image']
    out = os.path.join(ROOT, 'tests', 'golden', name + ('_expert' if expert else '') + '.npz')
    np.savez_compressed(
        out, level=name, seeds=seeds, actions=actions, pre_image=pre_image,
        pre_mission=np.array(pre_mission, dtype=object).astype(str) if pre_resets else np.zeros((0, n_envs), dtype=str),
        image=image, direction=direction, reward=reward, reward64=reward64, done=done, max_steps=max_steps, pixels=pixels,
        event_t=np.array([e[0] for e in events], np.int32), event_env=np.array([e[1] for e in events], np.int32),
        event_mission=np.array([e[2] for e in events]).astype(str))
    print('%-14s%s envs=%d steps=%d episodes_finished=%d success=%d -> %d KB' % (
        name, ' (expert)' if expert else '', n_envs, n_steps, n_done, int((reward > 0).sum()), os.path.getsize(out) // 1024))


if __name__ == '__main__':
    only = set(sys.argv[1:])
    for p in PLAN:
        if not only or p[0] in only:
            trace(*p)
