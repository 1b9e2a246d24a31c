#!/usr/bin/env python3
"""Golden decisions of the REFERENCE's GOFAI expert (babyai/bot.py, imported unmodified on the shim), for the
batched bot.  Build container only; fixtures go to tests/golden/bot/.

Per level two rollouts are recorded:
  pure     the bot's suggestion is the action taken (babyai/utils/agent.py:139-146 BotAgent, make_agent_demos.py)
  advised  12 % of the actions are replaced by a random one and the bot is told (`replan(action_taken)`), which
           exercises the undo / replanning logic (bot.py:88-137)
Recorded per step and env: `suggest` = what `Bot.replan` returned (-1 once the bot raised -- assertion,
DisappearedBoxError, ... -- or exceeded a 2 s decision budget; it stays -1 until the next episode) and `action` =
what the env was stepped with.  The env protocol is the ParallelEnv worker's (auto-reset, new Bot per episode).

    PYTHONDONTWRITEBYTECODE=1 timeout 3000 python tools/gen_golden_bot.py [Level ...]
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
from babyai.bot import Bot  # noqa: E402

SEED_BASE = 4000
OUT = os.path.join(ROOT, "tests", "golden", "bot")

BIG = {"BossLevel": (6, 700), "BossLevelNoUnlock": (4, 500), "MiniBossLevel": (6, 400), "Synth": (4, 300), "SynthLoc": (4, 300),
       "SynthSeq": (4, 400), "GoToSeq": (4, 300), "GoToImpUnlock": (4, 300), "Unlock": (6, 300), "UnblockPickup": (6, 300),
       "PutNext": (4, 300), "GoTo": (4, 250), "Pickup": (4, 250), "Open": (4, 200), "KeyCorridor": (4, 400),
       "BlockedUnlockPickup": (6, 250), "UnlockToUnlock": (4, 300), "PutNextLocal": (8, 200), "PickupLoc": (8, 160),
       "GoToLocal": (8, 160), "KeyInBox": (4, 200), "MoveTwoAcrossS8N9": (3, 300)}


class BotTimeout(BaseException):
    pass


def _on_alarm(signum, frame):
    raise BotTimeout()


def rollout(name, n_envs, n_steps, p_random, rng):
    envs = []
    for i in range(n_envs):
        env = level_dict[name]()
        if hasattr(env, "locked_room"):
            env.locked_room = None          # seed() starts from locked_room = None (see tools/gen_golden.py)
        env.seed(SEED_BASE + i)
        env.reset()
        envs.append(env)
    bots = [Bot(e) for e in envs]
    last = [None] * n_envs
    suggest = np.full((n_steps, n_envs), -1, np.int8)
    action = np.zeros((n_steps, n_envs), np.uint8)
    done = np.zeros((n_steps, n_envs), np.uint8)
    reward = np.zeros((n_steps, n_envs), np.float32)
    for t in range(n_steps):
        for i, e in enumerate(envs):
            a = None
            if bots[i] is not None:
                try:
                    signal.signal(signal.SIGALRM, _on_alarm)
                    signal.setitimer(signal.ITIMER_REAL, 2.0)
                    try:
                        a = int(bots[i].replan(last[i]))
                    finally:
                        signal.setitimer(signal.ITIMER_REAL, 0)
                except BaseException as exc:
                    if isinstance(exc, KeyboardInterrupt):
                        raise
                    bots[i] = None
            if a is not None:
                suggest[t, i] = a
            if a is None or rng.rand() < p_random:
                a = int(rng.randint(0, 7))
            action[t, i] = a
            last[i] = a
            _, r, d, _ = e.step(a)
            reward[t, i] = r
            done[t, i] = d
            if d:
                e.reset()
                bots[i] = Bot(e)
                last[i] = None
    return suggest, action, done, reward


def trace(name):
    base = name
    n_envs, n_steps = BIG.get(base, (4, 160))
    rng = np.random.RandomState(sum(map(ord, name)) + 5)
    out = {}
    for mode, p in (("pure", 0.0), ("advised", 0.12)):
        s, a, d, r = rollout(name, n_envs, n_steps, p, rng)
        out[mode + "_suggest"], out[mode + "_action"], out[mode + "_done"] = s, a, d
        print("%-28s %-8s envs=%d steps=%d episodes=%d success=%d bot_dead_steps=%d" % (
            name, mode, n_envs, n_steps, int(d.sum()), int((r > 0).sum()), int((s < 0).sum())), flush=True)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), level=name, seed_base=SEED_BASE, **out)


DEMO_PLAN = [("GoToLocal", 24, 100), ("PutNextLocal", 12, 7), ("PickupDist", 80, 300), ("BossLevel", 6, 41),
             ("UnlockToUnlock", 4, 9)]


def demos():
    """scripts/make_agent_demos.py:71-137 generate_demos with BotAgent, `--on-exception warn`: demo k = first episode of
    stream seed + k the bot solves; a crash or a failed mission means env.reset() on the same stream."""
    import hashlib
    out = {}
    for name, n_episodes, seed in DEMO_PLAN:
        env = level_dict[name]()
        if hasattr(env, "locked_room"):
            env.locked_room = None
        result = []
        just_crashed = False
        retries = 0
        while len(result) < n_episodes:
            if just_crashed:
                obs = None
                retries += 1
            else:
                env.seed(seed + len(result))
                if hasattr(env, "locked_room"):
                    env.locked_room = None
            obs = env.reset()
            bot = Bot(env)
            mission, images, directions, actions = obs["mission"], [], [], []
            done, reward = False, 0
            try:
                while not done:
                    signal.signal(signal.SIGALRM, _on_alarm)
                    signal.setitimer(signal.ITIMER_REAL, 2.0)
                    try:
                        action = int(bot.replan())
                    finally:
                        signal.setitimer(signal.ITIMER_REAL, 0)
                    new_obs, reward, done, _ = env.step(action)
                    actions.append(action)
                    images.append(obs["image"])
                    directions.append(obs["direction"])
                    obs = new_obs
                if reward > 0:
                    result.append((mission, np.array(images), directions, actions))
                    just_crashed = False
                if reward == 0:
                    just_crashed = True
            except BaseException as exc:
                if isinstance(exc, KeyboardInterrupt):
                    raise
                just_crashed = True
                continue
        out[name + "_mission"] = np.array([d[0] for d in result]).astype(str)
        out[name + "_length"] = np.array([len(d[3]) for d in result], np.int32)
        out[name + "_actions"] = np.concatenate([np.array(d[3], np.uint8) for d in result])
        out[name + "_directions"] = np.concatenate([np.array(d[2], np.uint8) for d in result])
        out[name + "_image_sha"] = np.array([hashlib.sha256(d[1].tobytes()).hexdigest() for d in result]).astype(str)
        out[name + "_seed"] = seed
        print("%-16s demos=%d retries=%d mean_len=%.1f" % (name, n_episodes, retries, out[name + "_length"].mean()), flush=True)
    np.savez_compressed(os.path.join(OUT, "..", "demos", "bot_demos.npz"), **out)


if __name__ == "__main__":
    if sys.argv[1:2] == ["demos"]:
        demos()
        sys.exit(0)
    names = sys.argv[1:] or sorted(level_dict)
    for n in names:
        trace(n)
