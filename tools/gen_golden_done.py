#!/usr/bin/env python3
"""Golden traces of the reference's DONE-ACTION verifier mode (babyai/levels/verifier.py:17,216-230,543-545), recorded from
the reference itself: /root/reference/babyai imported UNMODIFIED (on the restated gym / gym_minigrid shim of oracle/shim)
with BABYAI_DONE_ACTIONS set before the import, as a user of that mode would run it.

Build container only (the reference tree cannot travel); the fixtures under tests/golden/done_actions/ are committed and pin
  * the oracle's restatement of the mode (oracle/levels.py DONE_ACTIONS)      -> tests/test_done_actions.py
  * the engine's per-env core on the host and the HIP engine through the C ABI -> tests/test_done_actions.py (-m gpu)
Re-run:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_done.py

Protocol = tools/gen_golden.py's (a ParallelEnv worker: seed, reset, step, reset on done).  Actions: the reference's own
expert most of the time -- so that instructions do get completed -- a `done` action with probability 0.22 (right after a
completed instruction it ends the episode with success, anywhere else with failure), uniformly random otherwise.  Actions
are passed as plain ints, like every vectorised caller of the reference does (babyai/rl/utils/penv.py:8).

`--enum`: the same protocol on the levels whose missions contain AndInstr, with every `done` passed as the ENUM MEMBER
`env.actions.done` -- what the reference's own expert returns (babyai/bot.py:593) and scripts/make_agent_demos.py:93-107 feeds to
env.step.  Only then does AndInstr's failure rule (verifier.py:543-545, `action is self.env.actions.done`) fire.  Fixtures:
tests/golden/done_actions_enum/ (pin oracle, host build and -- bbai_set_option "done_action_enum" -- the HIP engine).
"""
import os
import signal
import sys

os.environ["BABYAI_DONE_ACTIONS"] = "1"

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refenv  # noqa: E402

assert refenv.have_reference(), "needs /root/reference"
refenv.enable_shim()
sys.dont_write_bytecode = True
sys.path.insert(1, refenv.REFERENCE_DIR)
import babyai.levels  # noqa: E402,F401
from babyai.levels import level_dict, verifier  # noqa: E402

assert verifier.use_done_actions, "the reference did not pick the variable up"

# level, n_envs, n_steps
PLAN = [
    ("GoToLocal", 8, 300),
    ("PickupLoc", 8, 300),
    ("PutNextLocal", 6, 400),
    ("Open", 6, 240),
    ("GoToSeq", 6, 400),
    ("SynthSeq", 8, 500),
    ("BossLevel", 8, 700),
    ("OpenDoorsOrderN4Debug", 6, 240),       # strict Before / After: the probe of the second part records lastStepMatch too
    ("PickupDistDebug", 6, 200),
    ("ActionObjDoor", 6, 200),
]
PLAN_ENUM = [("GoToSeq", 8, 500), ("SynthSeq", 8, 500), ("BossLevel", 8, 700), ("MiniBossLevel", 8, 500)]
ENUM = "--enum" in sys.argv
SEED_BASE = 7000 + (500 if ENUM else 0)


class BotTimeout(BaseException):
    pass


def _on_alarm(signum, frame):
    raise BotTimeout()


class Driver(object):
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
        u = self.rng.rand()
        if u < 0.22:
            a = 6
        elif a is None or u < 0.30:
            a = int(self.rng.randint(0, 7))
        self.last = a
        return a


def trace(name, n_envs, n_steps):
    rng = np.random.RandomState(sum(map(ord, name)) + 4242)
    seeds = np.arange(n_envs, dtype=np.uint64) + SEED_BASE
    envs = []
    for s in seeds:
        env = level_dict[name]()
        if hasattr(env, 'locked_room'):
            env.locked_room = None          # (see tools/gen_golden.py: seed() starts from locked_room = None)
        env.seed(int(s))
        envs.append(env)
    actions = np.zeros((n_steps, n_envs), np.uint8)
    image = np.zeros((n_steps + 1, n_envs, 7, 7, 3), np.uint8)
    direction = np.zeros((n_steps + 1, n_envs), np.uint8)
    reward64 = np.zeros((n_steps, n_envs), np.float64)
    done = np.zeros((n_steps, n_envs), np.uint8)
    missions = []
    for i, e in enumerate(envs):
        o = e.reset()
        image[0, i] = o['image']; direction[0, i] = o['direction']
        missions.append((0, i, o['mission']))
    drivers = [Driver(e, rng) for e in envs]
    for t in range(n_steps):
        for i, e in enumerate(envs):
            a = drivers[i].act()
            actions[t, i] = a
            o, r, d, _ = e.step(e.actions.done if (ENUM and a == 6) else int(a))
            reward64[t, i] = r
            done[t, i] = d
            if d:
                o = e.reset()
                missions.append((t + 1, i, o['mission']))
                drivers[i].new_episode()
            image[t + 1, i] = o['image']; direction[t + 1, i] = o['direction']
    out_dir = os.path.join(ROOT, 'tests', 'golden', 'done_actions_enum' if ENUM else 'done_actions')
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, name + '.npz')
    np.savez_compressed(out, level=name, seeds=seeds, actions=actions, image=image, direction=direction, reward64=reward64,
                        done=done, event_t=np.array([m[0] for m in missions], np.int32),
                        event_env=np.array([m[1] for m in missions], np.int32),
                        event_mission=np.array([m[2] for m in missions]).astype(str))
    ended_by_done = int(((actions == 6) & (done == 1)).sum())
    print('%-24s envs=%d steps=%d episodes=%d success=%d ended_by_a_done_action=%d (failures %d) -> %d KB' % (
        name, n_envs, n_steps, int(done.sum()), int((reward64 > 0).sum()), ended_by_done,
        ended_by_done - int((reward64 > 0).sum()), os.path.getsize(out) // 1024))


if __name__ == '__main__':
    only = set(a for a in sys.argv[1:] if not a.startswith("--"))
    for p in (PLAN_ENUM if ENUM else PLAN):
        if not only or p[0] in only:
            trace(*p)
