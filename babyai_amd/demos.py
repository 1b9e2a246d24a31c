"""Expert demonstrations on the batched engine -- `generate_demos` of the reference
(scripts/make_agent_demos.py:71-137 with `--model BOT`, i.e. babyai/utils/agent.py:139-155 BotAgent) with the
per-episode Python loop replaced by the device expert (`bbai_bot_act`) driving a batch of envs.

Same result as the reference's sequential loop: demo k is the first episode of stream `seed + k` that the bot solves
(`env.seed(seed + len(demos))`, and after a failure or a bot crash `env.reset()` on the SAME stream,
make_agent_demos.py:84-88,112-123), as the tuple `(mission, images, directions, actions)` of
make_agent_demos.py:111-112 -- `images` is uint8[T,7,7,3] (the reference stores `blosc.pack_array` of it; pass
`pack=blosc.pack_array` to get the identical container format).  The mission text is rebuilt from the token ids the engine
keeps per episode (`missions.detokenize`: the baby language has 32 words and one punctuation mark).
"""
import numpy as np

from . import missions
from .engine import BatchedBabyAIEnv


def generate_demos(env_name, n_episodes, seed, device="cuda:0", batch=32768, filter_steps=0, pack=None, max_steps=None,
                   rollout=None):
    """`batch` streams run side by side on the device; a batch lasts as long as its slowest stream (the expert's decision
    kernel has a latency of about a millisecond whatever the batch size), so large batches are what makes this fast.
    Memory: with the device rollout a batch keeps ~225 bytes per stream and step on the device until its last stream is
    solved (0.95 GB per 128 steps at 32 768 streams, bounded by half of the free device memory: RuntimeError beyond).

    rollout: True = the per-step loop runs on the device (`bbai_bot_rollout`, history kept there, spans gathered there);
    False = one host round trip per step (rounds 1-2).  Same demonstrations either way.  None picks by what was measured
    (profiles/r03/demo_bench_final2.jsonl, one MI355X, demos/s rollout vs stepwise): BossLevel 131 072 streams 27.9 k vs
    17.6 k, 32 768 streams 13.7 k vs 13.5 k; GoToLocal (5-step episodes) 65 536 streams 73-75 k vs 101-126 k -- episodes
    shorter than a rollout chunk are cheaper to follow step by step."""
    from .levels import make_cfg
    if rollout is None:
        cfg = make_cfg(env_name)
        rollout = cfg.num_rows * cfg.num_cols > 1
    run = _generate_batch if rollout else _generate_batch_stepwise
    demos = [None] * n_episodes
    for start in range(0, n_episodes, batch):
        count = min(batch, n_episodes - start)
        run(env_name, seed + start, count, device, filter_steps, pack, max_steps, demos, start)
    return demos


def scan_chunk(done_t, gave_up_t, reward_t, g0, filter_steps, last_done, open_, span):
    """One rollout chunk's results ([chunk, n] arrays as the engine writes them; global step index of row 0 = g0): for every
    stream still open, find the first episode that ENDS in this chunk solved -- done, not a bot crash, reward > 0, at most
    `filter_steps` long if that is set -- and record its [first, last] step in `span`; failed episodes and crashes
    ("mission failed" / RESET_ENV, make_agent_demos.py:84-88,112-123) just move the stream on to its next level.
    Updates last_done (global index of every stream's latest episode end), open_ and span in place."""
    chunk = done_t.shape[0]
    # the scans run along time: [n, chunk] layout, contiguous per stream
    done_t = done_t.astype(bool)
    ok = np.ascontiguousarray((done_t & (gave_up_t == 0) & (reward_t > 0)).T)
    done = np.ascontiguousarray(done_t.T)
    idx = np.arange(g0, g0 + chunk, dtype=np.int32)[None, :]
    ends = np.maximum.accumulate(np.where(done, idx, np.int32(-1)), axis=1)         # latest episode end at or before each step
    start = np.empty_like(ends)                                                      # first step of each step's episode
    start[:, 0] = last_done + 1
    np.maximum(ends[:, :-1], last_done[:, None], out=start[:, 1:])
    start[:, 1:] += 1
    if filter_steps:
        ok &= (idx - start + 1) <= filter_steps
    found = ok.any(axis=1) & open_
    first = ok.argmax(axis=1)
    cols = np.nonzero(found)[0]
    span[cols, 0], span[cols, 1] = start[cols, first[cols]], g0 + first[cols]
    open_ &= ~found
    np.maximum(last_done, ends[:, -1], out=last_done)


def _generate_batch_stepwise(env_name, seed, n, device, filter_steps, pack, max_steps, demos, offset):
    """One batch of streams, one host round trip per step (bbai_bot_act + bbai_step driven from here)."""
    import torch
    env = BatchedBabyAIEnv(env_name, n, device=device, seeds=[seed + k for k in range(n)], auto_reset=True)
    if env.done_actions:
        # BABYAI_DONE_ACTIONS: the actions below are the expert's own, and its `done` is the enum member (babyai/bot.py:593): AndInstr's
        # identity-tested failure rule applies (verifier.py:543-545; include/bbai.h bbai_set_done_actions) -- as it does inside bbai_bot_rollout
        env.set_option("done_action_enum", 1)
    obs = env.reset()
    mission = list(obs["mission"])
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
        solved = open_ & done_h & ~crashed_h & (reward_h > 0)
        if filter_steps:
            solved &= (t - ep_start + 1) <= filter_steps
        span[solved, 0], span[solved, 1] = ep_start[solved], t
        open_ &= ~solved
        again = open_ & done_h                               # "mission failed" / crash: next level of the same stream
        if again.any():
            fresh = obs["mission"]
            for i in np.nonzero(again)[0]:
                mission[i] = fresh[i]
        ep_start[done_h] = t + 1
    env.close()
    if open_.any():
        raise RuntimeError("no solvable episode found for %d stream(s) within the step budget" % int(open_.sum()))
    img, dirs, acts = np.stack(hist_img), np.stack(hist_dir), np.stack(hist_act)
    for i in range(n):
        lo, hi = span[i, 0], span[i, 1] + 1
        stack = np.ascontiguousarray(img[lo:hi, i])
        demos[offset + i] = (mission[i], pack(stack) if pack else stack, dirs[lo:hi, i].tolist(), acts[lo:hi, i].tolist())


def _generate_batch(env_name, seed, n, device, filter_steps, pack, max_steps, demos, offset):
    """One batch of streams.  The per-step loop runs on the device (`bbai_bot_rollout`: decide, step, auto-reset, history
    rows written by the engine, `chunk` steps per call) and the history STAYS there: per chunk the host reads 6 bytes per
    env-step (done, gave-up flag, reward) to find every stream's first solved episode, and at the end the spans of those
    episodes are gathered on the device and cross PCIe once."""
    import torch
    env = BatchedBabyAIEnv(env_name, n, device=device, seeds=[seed + k for k in range(n)], auto_reset=True)
    env.enable_instr_tokens()
    env.reset()
    budget = max_steps if max_steps is not None else 64 * env.max_steps_bound
    chunk = max(1, min(128, max(16, env.max_steps_bound // 4), budget))
    hist = []                                      # the chunks' device tensors, [chunk, n, ...] each
    last_done = np.full(n, -1, dtype=np.int32)     # global index of the stream's latest episode end
    span = np.full((n, 2), -1, dtype=np.int64)     # [first, last] step of the stream's first solved episode
    open_ = np.ones(n, dtype=bool)                 # streams still looking for it
    g0 = 0
    # The history stays on the device until every stream has its episode: 225 bytes per env-step (image 147, tokens 72, 6 flags),
    # ~0.95 GB per 128-step chunk of 32 768 streams.  A stream the expert never solves would otherwise grow it up to the step
    # budget (64 x max_steps) and end in a device out-of-memory error instead of the RuntimeError below: bound it by half of
    # the memory that is free now.
    free_b, _ = torch.cuda.mem_get_info(env.device)
    chunk_bytes = chunk * n * (147 + 72 + 6 + 4)
    max_chunks = max(2, int(free_b // 2 // max(1, chunk_bytes)))
    while open_.any() and g0 < budget:
        if len(hist) >= max_chunks:
            env.close()
            raise RuntimeError("no solvable episode found for %d stream(s) within the history memory budget (%d chunks of %d steps x %d "
                               "streams = %.1f GB on the device): use a smaller `batch`" % (int(open_.sum()), len(hist), chunk, n, len(hist) * chunk_bytes / 1e9))
        r = env.bot_rollout(chunk, tokens=True)
        hist.append(r)
        scan_chunk(r["done"].cpu().numpy(), r["gave_up"].cpu().numpy(), r["reward"].cpu().numpy(), g0, filter_steps, last_done, open_, span)
        g0 += chunk
    if open_.any():
        env.close()
        raise RuntimeError("no solvable episode found for %d stream(s) within the step budget" % int(open_.sum()))
    # gather the spans on the device: entry k of the flat result = step t_idx[k] of stream i_idx[k]
    lens = span[:, 1] - span[:, 0] + 1
    ends_flat = np.cumsum(lens)
    total = int(ends_flat[-1])
    i_idx = np.repeat(np.arange(n, dtype=np.int64), lens)
    t_idx = np.arange(total, dtype=np.int64) - np.repeat(ends_flat - lens, lens) + np.repeat(span[:, 0], lens)
    dev = env.device
    img = torch.empty((total, 7, 7, 3), dtype=torch.uint8, device=dev)
    dirs = torch.empty((total,), dtype=torch.uint8, device=dev)
    acts = torch.empty((total,), dtype=torch.uint8, device=dev)
    toks = torch.empty((n, hist[0]["tokens"].shape[2]), dtype=torch.uint8, device=dev)
    for c, r in enumerate(hist):
        lo = c * chunk
        sel = np.nonzero((t_idx >= lo) & (t_idx < lo + chunk))[0]
        if sel.size:
            at = torch.as_tensor(sel, device=dev)
            src = torch.as_tensor((t_idx[sel] - lo) * n + i_idx[sel], device=dev)
            img[at] = r["image"].view(chunk * n, 7, 7, 3)[src]
            dirs[at] = r["direction"].view(-1)[src]
            acts[at] = r["action"].view(-1)[src]
        sel = np.nonzero((span[:, 0] >= lo) & (span[:, 0] < lo + chunk))[0]      # the mission is the one of the episode's first step
        if sel.size:
            toks[torch.as_tensor(sel, device=dev)] = r["tokens"].view(chunk * n, -1)[torch.as_tensor((span[sel, 0] - lo) * n + sel, device=dev)]
    img, dirs, acts, toks = img.cpu().numpy(), dirs.cpu().numpy(), acts.cpu().numpy(), toks.cpu().numpy()
    env.close()
    # per-demo assembly on plain Python objects (a numpy slice costs more than the few elements it holds)
    text = {}                                      # a batch holds far fewer distinct missions than streams
    width = toks.shape[1]
    tok_bytes, dirs_l, acts_l, ends_l = toks.tobytes(), dirs.tolist(), acts.tolist(), ends_flat.tolist()
    lo = 0
    for i in range(n):
        hi = ends_l[i]
        key = tok_bytes[i * width:(i + 1) * width]
        mission = text.get(key)
        if mission is None:
            mission = text[key] = missions.detokenize(key)
        stack = img[lo:hi]
        demos[offset + i] = (mission, pack(stack) if pack else stack, dirs_l[lo:hi], acts_l[lo:hi])
        lo = hi
