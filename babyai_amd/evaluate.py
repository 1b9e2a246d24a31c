"""`batch_evaluate` on the batched engine -- the evaluation driver of the reference
(babyai/evaluate.py:85-140) with its `gym.make` loop + `ManyEnvs` replaced by `BatchedManyEnvs`.

Same signature and the same `logs` dict (num_frames_per_episode, return_per_episode, seed_per_episode,
optionally observations/actions per episode), so scripts/evaluate.py-style callers can switch by import.
The agent contract is the reference's: `agent.act_batch(many_obs)['action']` (babyai/utils/agent.py:51-72)
and `agent.analyze_feedback(reward, done)`.
"""
import numpy as np

from .vec_env import BatchedManyEnvs


def batch_evaluate(agent, env_name, seed, episodes, return_obss_actions=False, pixel=False, device="cuda:0"):
    num_envs = min(256, episodes)
    env = BatchedManyEnvs(env_name, num_envs, device=device, pixel=pixel)

    logs = {
        "num_frames_per_episode": [],
        "return_per_episode": [],
        "observations_per_episode": [],
        "actions_per_episode": [],
        "seed_per_episode": [],
    }

    for i in range((episodes + num_envs - 1) // num_envs):
        seeds = range(seed + i * num_envs, seed + (i + 1) * num_envs)
        env.seed(seeds)
        many_obs = env.reset()

        cur_num_frames = 0
        num_frames = np.zeros((num_envs,), dtype='int64')
        returns = np.zeros((num_envs,))
        already_done = np.zeros((num_envs,), dtype='bool')
        if return_obss_actions:
            obss = [[] for _ in range(num_envs)]
            actions = [[] for _ in range(num_envs)]
        while (num_frames == 0).any():
            action = agent.act_batch(many_obs)['action']
            if return_obss_actions:
                for k in range(num_envs):
                    if not already_done[k]:
                        obss[k].append(many_obs[k])
                        actions[k].append(int(action[k]))
            many_obs, reward, done, _ = env.step(np.asarray([int(a) for a in action]))
            agent.analyze_feedback(reward, done)
            done = np.array(done)
            just_done = done & (~already_done)
            returns += np.array(reward) * just_done
            cur_num_frames += 1
            num_frames[just_done] = cur_num_frames
            already_done[done] = True

        logs["num_frames_per_episode"].extend(list(num_frames))
        logs["return_per_episode"].extend(list(returns))
        logs["seed_per_episode"].extend(list(seeds))
        if return_obss_actions:
            logs["observations_per_episode"].extend(obss)
            logs["actions_per_episode"].extend(actions)

    env.close()
    return logs
