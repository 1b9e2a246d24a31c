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
    missions = list(obs["mission"])
    # whole-batch history, one row per step; an episode is the slice [ep_start[i], t] of column i
    hist_img, hist_dir, hist_act = [], [], []
    ep_start = np.zeros(n, dtype=np.int64)
    span = np.full((n, 2), -1, dtype=np.int64)     # [first, last] step of the stream's first solved episode
    open_ = np.ones(n, dtype=bool)                 # streams still looking for it
    budget = max_steps if max_steps is not None else 64 * env.max_steps_bound
    reset_cmd = torch.full((n,), env.RESET_ENV, dtype=torch.uint8, device=env.device)
    for t in range(budget):
        if not open_.any():
            break
        hist_img.append(obs["image"].cpu().numpy())
        hist_dir.append(obs["direction"].cpu().numpy())
        act = env.bot_actions(None)
        crashed = act == env.BOT_GAVE_UP
        act = torch.where(crashed, reset_cmd, act)          # bot crash: env.reset() on the same stream
        obs, reward, done, _ = env.step(act)
        hist_act.append(act.cpu().numpy())
        crashed_h = crashed.cpu().numpy()
        reward_h, done_h = reward.cpu().numpy(), done.cpu().numpy().astype(bool)
        length = t - ep_start + 1
        solved = open_ & done_h & ~crashed_h & (reward_h > 0)
        if filter_steps:
            solved &= length <= filter_steps
        span[solved, 0], span[solved, 1] = ep_start[solved], t
        open_ &= ~solved
        again = open_ & done_h                               # "mission failed" / crash: next level of the same stream
        if again.any():
            fresh = obs["mission"]
            for i in np.nonzero(again)[0]:
                missions[i] = fresh[i]
        ep_start[done_h] = t + 1
    env.close()
    if open_.any():
        raise RuntimeError("no solvable episode found for %d stream(s) within the step budget" % int(open_.sum()))
    img, dirs, acts = np.stack(hist_img), np.stack(hist_dir), np.stack(hist_act)
    for i in range(n):
        lo, hi = span[i, 0], span[i, 1] + 1
        stack = np.ascontiguousarray(img[lo:hi, i])
        demos[offset + i] = (missions[i], pack(stack) if pack else stack,
                             [int(v) for v in dirs[lo:hi, i]], [int(v) for v in acts[lo:hi, i]])
