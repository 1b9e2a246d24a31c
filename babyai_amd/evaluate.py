"""Device-resident evaluation of a policy on a BabyAI level.

What it computes is what the reference's evaluation driver logs (`batch_evaluate`, babyai/evaluate.py:85-140: per episode
the number of frames, the return and the seed, episode k seeded with `seed + k`, finished envs frozen until everybody is
done, evaluate.py:73-81) -- but not how: the reference walks 256 Python envs per round and re-derives the bookkeeping on
the host every frame.  Here ALL episodes are one batch of engine envs (ManyEnvs semantics = `auto_reset=False`), the policy
sees device tensors, returns / frame counts are reduced on the device as the frames go by, and the host reads them back
once per chunk.  Returns are accumulated from the float64 rewards, as evaluate.py:128 does.

    logs = evaluate_policy(policy, "BabyAI-GoToLocal-v0", seed=10**9, episodes=100_000)

Differences of `evaluate_policy` from the reference's driver, on purpose: exactly `episodes` episodes are logged (the
reference rounds up to whole rounds of min(256, episodes) envs), all of them run as ONE batch (an agent sees up to `chunk`
observations per `act_batch`), and observations / actions per episode are not collected.  `batch_evaluate` below keeps
the reference's signature, round structure and log shape for callers that depend on them.

`policy(obs, t)` gets `obs = {"image": uint8 tensor [N,7,7,3] (or [N,56,56,3]), "direction": uint8 [N], "instr": uint8
[N,72] mission token ids}` and returns an integer tensor [N] of actions on the same device.  A reference-style agent
(`act_batch(list_of_obs_dicts)` / `analyze_feedback`, babyai/utils/agent.py:51-84) plugs in through `AgentPolicy`,
at the price of one host round trip per frame; callers that want the reference's driver itself can also run it unchanged
on `vec_env.BatchedManyEnvs`.
"""
import numpy as np

from .engine import BatchedBabyAIEnv


class AgentPolicy(object):
    """Adapter: a reference-style agent as a tensor policy (host round trip per frame)."""

    def __init__(self, agent, env):
        from .vec_env import ObsList
        self._agent, self._env, self._ObsList = agent, env, ObsList

    def __call__(self, obs, t):
        env = self._env
        # (missions snapshotted: an agent may keep the obs list past the next step, babyai/utils/agent.py:101-137 DemoAgent)
        many = self._ObsList(obs["image"].cpu().numpy(), obs["direction"].cpu().numpy(), env._missions.snapshot(), env.pixel)
        action = self._agent.act_batch(many)["action"]
        return env.torch.as_tensor(np.asarray([int(a) for a in action], dtype=np.uint8), device=env.device)

    def feedback(self, reward64, done):
        self._agent.analyze_feedback(tuple(float(r) for r in reward64.cpu().numpy()),
                                     tuple(bool(d) for d in done.cpu().numpy()))


def evaluate_policy(policy, env_name, seed, episodes, pixel=False, device="cuda:0", chunk=262144, poll_every=16,
                    agent=None):
    """Run `episodes` episodes (seeds seed .. seed+episodes-1) to completion; returns the reference's `logs` dict
    (num_frames_per_episode, return_per_episode, seed_per_episode).  `agent`: wrap a reference-style agent instead of a
    tensor policy.  `poll_every`: frames between two host checks of "is everybody done" (frozen envs cost nothing)."""
    import torch
    logs = {"num_frames_per_episode": [], "return_per_episode": [], "seed_per_episode": []}
    for first in range(0, episodes, chunk):
        n = min(chunk, episodes - first)
        env = BatchedBabyAIEnv(env_name, n, device=device, pixel=pixel, auto_reset=False)
        env.seed(np.arange(first, first + n, dtype=np.uint64) + np.uint64(seed))
        instr = env.enable_instr_tokens()
        obs = env.reset()
        pol = AgentPolicy(agent, env) if agent is not None else policy
        frames = torch.zeros(n, dtype=torch.int64, device=env.device)      # 0 = still running
        returns = torch.zeros(n, dtype=torch.float64, device=env.device)
        t = 0
        while True:
            t += 1
            action = pol({"image": obs["image"], "direction": obs["direction"], "instr": instr}, t)
            obs, _, done, _ = env.step(action)
            if agent is not None:
                pol.feedback(env.reward64, done)
            just = (done != 0) & (frames == 0)
            returns += env.reward64 * just
            frames = torch.where(just, torch.full_like(frames, t), frames)
            # every episode ends by max_steps at the latest; poll the device only now and then before that
            every = 1 if agent is not None else poll_every        # (a stateful agent must not see extra frames)
            if t >= env.max_steps_bound or (t % every == 0 and bool((frames != 0).all())):
                break
        logs["num_frames_per_episode"].extend(frames.cpu().tolist())
        logs["return_per_episode"].extend(returns.cpu().tolist())
        logs["seed_per_episode"].extend(range(seed + first, seed + first + n))
        env.close()
    return logs


def batch_evaluate(agent, env_name, seed, episodes, return_obss_actions=False, pixel=False, device="cuda:0"):
    """The reference's `batch_evaluate(agent, env_name, seed, episodes, return_obss_actions, pixel)`
    (babyai/evaluate.py:85-140) with its signature and log shape, on engine envs: rounds of min(256, episodes) envs under
    the ManyEnvs protocol (`vec_env.BatchedManyEnvs`), round i seeded seed + i*num_envs ..., so -- like the reference --
    ceil(episodes / num_envs) * num_envs episodes are logged; `observations_per_episode` / `actions_per_episode` hold, per
    episode, the obs dicts the agent acted on and its actions until that episode finished (empty lists unless
    `return_obss_actions`).  One host round trip per frame: the price of the list-of-dicts agent interface.  Use
    `evaluate_policy` for large device-resident evaluations."""
    from .vec_env import BatchedManyEnvs
    num_envs = min(256, episodes)
    logs = {"num_frames_per_episode": [], "return_per_episode": [], "observations_per_episode": [], "actions_per_episode": [],
            "seed_per_episode": []}
    env = BatchedManyEnvs(env_name, num_envs, device=device, pixel=pixel)
    try:
        for rnd in range(-(-episodes // num_envs)):
            seeds = range(seed + rnd * num_envs, seed + (rnd + 1) * num_envs)
            env.seed(list(seeds))
            many_obs = env.reset()
            finished_at = np.zeros(num_envs, dtype="int64")            # 0 = still running
            returns = np.zeros(num_envs)
            trace_obs = [[] for _ in range(num_envs)] if return_obss_actions else None
            trace_act = [[] for _ in range(num_envs)] if return_obss_actions else None
            frame = 0
            while not finished_at.all():
                action = agent.act_batch(many_obs)["action"]
                if return_obss_actions:
                    for k in np.flatnonzero(finished_at == 0):
                        trace_obs[k].append(many_obs[k])
                        trace_act[k].append(action[k].item())
                many_obs, reward, done, _ = env.step(np.asarray([int(a) for a in action], dtype=np.uint8))
                agent.analyze_feedback(reward, done)
                frame += 1
                just = np.asarray(done, dtype=bool) & (finished_at == 0)
                returns += np.asarray(reward) * just                  # float64 rewards: the reference's accumulation
                finished_at[just] = frame
            logs["num_frames_per_episode"].extend(list(finished_at))
            logs["return_per_episode"].extend(list(returns))
            logs["seed_per_episode"].extend(list(seeds))
            if return_obss_actions:
                logs["observations_per_episode"].extend(trace_obs)
                logs["actions_per_episode"].extend(trace_act)
    finally:
        env.close()
    return logs
