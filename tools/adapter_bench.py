#!/usr/bin/env python3
"""Host-protocol rates: the list-of-dicts adapter the reference's BaseAlgo consumes (BatchedParallelEnv, penv.py protocol,
obs copied to the host every step) and the device-resident DeviceRollout, at training-sized batches."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from babyai_amd.engine import BatchedBabyAIEnv  # noqa: E402
from babyai_amd.rollout import DeviceRollout  # noqa: E402
from babyai_amd.vec_env import BatchedParallelEnv  # noqa: E402
from rollout_util import ToyACModel  # noqa: E402

out = {}
for procs in (64, 1024):
    penv = BatchedParallelEnv("BabyAI-GoToLocal-v0", procs, seeds=[100 + i for i in range(procs)])
    penv.reset()
    rng = np.random.RandomState(0)
    acts = rng.randint(0, 7, size=(300, procs))
    for t in range(20):
        penv.step(acts[t])
    t0 = time.perf_counter()
    for t in range(20, 300):
        obs, reward, done, info = penv.step(acts[t])
        _ = obs[0]["image"], obs[procs - 1]["mission"]          # what ObssPreprocessor touches
    dt = time.perf_counter() - t0
    out["BatchedParallelEnv_%d" % procs] = {"env_steps_per_s": 280 * procs / dt, "us_per_step_call": dt / 280 * 1e6}
for n in (64, 4096, 65536):
    env = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", n, seeds=7)
    roll = DeviceRollout(env, ToyACModel(), 40, 0.99, 0.99, reward_scale=20.0)
    roll.collect_experiences()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        exps, logs = roll.collect_experiences()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["DeviceRollout_%d" % n] = {"frames_per_s": 5 * 40 * n / dt, "us_per_frame_batch": dt / 200 * 1e6}
    env.close()
print(json.dumps(out))
