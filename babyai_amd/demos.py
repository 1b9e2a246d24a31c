"""Expert demonstrations on the batched engine -- `generate_demos` of the reference
(scripts/make_agent_demos.py:71-137 with `--model BOT`, i.e. babyai/utils/agent.py:139-155 BotAgent) with the
per-episode Python loop replaced by the device expert (`bbai_bot_act`) driving a batch of envs.

Same result as the reference's sequential loop: demo k is the first episode of stream `seed + k` that the bot solves
(`env.seed(seed + len(demos))`, and after a failure or a bot crash `env.reset()` on the SAME stream,
make_agent_demos.py:84-88,112-123), as the tuple `(mission, images, directions, actions)` of
make_agent_demos.py:111-112 -- `images` is uint8[T,7,7,3] (the reference stores `blosc.pack_array` of it; pass
`pack=blosc.pack_array` to get the identical container format).
"""
import numpy as np

from .engine import BatchedBabyAIEnv


def generate_demos(env_name, n_episodes, seed, device="cuda:0", batch=4096, filter_steps=0, pack=None, max_steps=None):
    demos = [None] * n_episodes
    for start in range(0, n_episodes, batch):
        count = min(batch, n_episodes - start)
        _generate_batch(env_name, seed + start, count, device, filter_steps, pack, max_steps, demos, start)
    return demos


def _generate_batch(env_name, seed, n, device, filter_steps, pack, max_steps, demos, offset):
    import torch
    env = BatchedBabyAIEnv(env_name, n, device=device, seeds=[seed + k for k in range(n)], auto_reset=True)
    obs = env.reset()
    missions = [obs["mission"][i] for i in range(n)]
    images = [[] for _ in range(n)]
    directions = [[] for _ in range(n)]
    actions = [[] for _ in range(n)]
    open_ = np.ones(n, dtype=bool)                 # streams still looking for their first solved episode
    budget = max_steps if max_steps is not None else 64 * env.max_steps_bound
    for _ in range(budget):
        if not open_.any():
            break
        img = obs["image"].cpu().numpy()
        dirs = obs["direction"].cpu().numpy()
        act = env.bot_actions(None)
        crashed = (act == env.BOT_GAVE_UP)
        act = torch.where(crashed, torch.full_like(act, env.RESET_ENV), act)
        obs, reward, done, _ = env.step(act)
        act_h, crashed_h = act.cpu().numpy(), crashed.cpu().numpy()
        reward_h, done_h = reward.cpu().numpy(), done.cpu().numpy().astype(bool)
        fresh = None
        for i in np.nonzero(open_)[0]:
            if not crashed_h[i]:
                images[i].append(img[i])
                directions[i].append(int(dirs[i]))
                actions[i].append(int(act_h[i]))
            if done_h[i]:
                if not crashed_h[i] and reward_h[i] > 0 and (filter_steps == 0 or len(images[i]) <= filter_steps):
                    stack = np.array(images[i])
                    demos[offset + i] = (missions[i], pack(stack) if pack else stack, directions[i], actions[i])
                    open_[i] = False
                else:                               # "mission failed" / bot crash: next level of the same stream
                    if fresh is None:
                        fresh = obs["mission"]
                    missions[i] = fresh[i]
                    images[i], directions[i], actions[i] = [], [], []
    env.close()
    if open_.any():
        raise RuntimeError("no solvable episode found for %d stream(s) within the step budget" % int(open_.sum()))
