"""Device-resident experience collection (second half).

`BaseAlgo.collect_experiences` (babyai/rl/algos/base.py:131-188) runs, per frame, a list-of-dicts
preprocess on the host, `action.cpu().numpy()`, a pipe round-trip per worker, four `torch.tensor(list)`
uploads and a Python loop over `done` with one `.item()` per finished episode.  On the batched engine
observations, rewards and done flags already are device tensors, so the rollout below is that function
with every per-frame host hop removed -- same recurrences, same `exps` fields in the same
(env-major, frame-minor) order, same `logs` dict -- and ONE device->host copy per rollout (the episode
statistics) instead of several per frame.

    env  = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", 4096, pixel=False, seeds=...)
    roll = DeviceRollout(env, acmodel, num_frames_per_proc=40, discount=0.99, gae_lambda=0.99, reward_scale=20.)
    exps, logs = roll.collect_experiences()      # then PPOAlgo.update_parameters-style code consumes exps

`acmodel(obs, memory)` follows babyai/model.py:217-273: it returns a dict with 'dist' (`.sample()`,
`.log_prob(a)`), 'value', 'memory' and has `.memory_size`.  `env` is any object with the tensor
protocol of `BatchedBabyAIEnv` (`num_envs`, `device`, `reset()`, `step(uint8 actions)`,
`enable_instr_tokens()`); the collector itself holds no environment logic.

Differences from the reference, all deliberate:
  * `exps.obs` is a `TensorDict(image float32, instr int64)` built from raw uint8 frames kept on the
    device (what `preprocess_obss(exps.obs)` yields at base.py:232), never a list of dicts;
  * reward shaping is `reward_scale * reward` (scripts/train_rl.py:104 `reshape_reward`) or any
    tensor-wise callable `(obs, action, reward, done) -> tensor`, not a per-element Python lambda;
  * `aux_info` (ExtraInfoCollector) is not carried: the engine's `info` dict is empty.
"""
from .preprocess import TensorDict


class DeviceRollout(object):
    def __init__(self, env, acmodel, num_frames_per_proc, discount, gae_lambda, reward_scale=None,
                 reshape_reward=None, recurrence=1):
        import torch
        self.torch = torch
        assert num_frames_per_proc % recurrence == 0                 # base.py:73
        assert reward_scale is None or reshape_reward is None
        self.env = env
        self.acmodel = acmodel
        self.num_frames_per_proc = T = int(num_frames_per_proc)
        self.discount = discount
        self.gae_lambda = gae_lambda
        self.reward_scale = reward_scale
        self.reshape_reward = reshape_reward
        self.recurrence = recurrence
        self.device = dev = env.device
        self.num_procs = P = env.num_envs
        self.num_frames = T * P

        self.tokens = env.enable_instr_tokens()                      # uint8[P, L], kept current by the engine
        obs = env.reset()                                            # base.py:79
        self._cur_image = obs["image"]                               # engine-owned buffers, overwritten by step()
        img_shape = tuple(obs["image"].shape[1:])
        self.images = torch.zeros((T, P) + img_shape, dtype=torch.uint8, device=dev)
        self.instrs = torch.zeros((T, P, self.tokens.shape[1]), dtype=torch.uint8, device=dev)

        self.memory = torch.zeros(P, acmodel.memory_size, device=dev)
        self.memories = torch.zeros(T, P, acmodel.memory_size, device=dev)
        self.mask = torch.ones(P, device=dev)
        self.masks = torch.zeros(T, P, device=dev)
        self.actions = torch.zeros(T, P, device=dev, dtype=torch.int)
        self.values = torch.zeros(T, P, device=dev)
        self.rewards = torch.zeros(T, P, device=dev)
        self.advantages = torch.zeros(T, P, device=dev)
        self.log_probs = torch.zeros(T, P, device=dev)

        # per-frame episode statistics, read back once per rollout (base.py:166-178 does it per frame)
        self.dones = torch.zeros(T, P, device=dev, dtype=torch.uint8)
        self.ep_return = torch.zeros(T, P, device=dev)
        self.ep_reshaped = torch.zeros(T, P, device=dev)
        self.ep_frames = torch.zeros(T, P, device=dev)
        self.log_episode_return = torch.zeros(P, device=dev)
        self.log_episode_reshaped_return = torch.zeros(P, device=dev)
        self.log_episode_num_frames = torch.zeros(P, device=dev)
        self.log_done_counter = 0
        self.log_return = [0] * P
        self.log_reshaped_return = [0] * P
        self.log_num_frames = [0] * P

    # -- helpers ---------------------------------------------------------------------------------------
    def _batch(self, image, instr):
        """RawImagePreprocessor + InstructionsPreprocessor (format.py:44-82) on device tensors."""
        torch = self.torch
        length = int((instr != 0).sum(dim=-1).max().item()) if instr.numel() else 0
        return TensorDict(image=image.to(torch.float32), instr=instr[..., :max(length, 1)].to(torch.int64))

    def _model(self, image, instr):
        with self.torch.no_grad():
            return self.acmodel(self._batch(image, instr), self.memory * self.mask.unsqueeze(1))

    # -- base.py:131-260 ------------------------------------------------------------------------------
    def collect_experiences(self):
        torch = self.torch
        T, P = self.num_frames_per_proc, self.num_procs
        for i in range(T):
            self.images[i].copy_(self._cur_image)                    # obss[i] = obs (before the step overwrites it)
            self.instrs[i].copy_(self.tokens)
            res = self._model(self.images[i], self.instrs[i])
            dist, value, memory = res["dist"], res["value"], res["memory"]
            action = dist.sample()

            obs, reward, done, _ = self.env.step(action.to(torch.uint8))
            self._cur_image = obs["image"]

            self.memories[i] = self.memory
            self.memory = memory
            self.masks[i] = self.mask
            self.mask = 1 - done.to(torch.float)
            self.actions[i] = action
            self.values[i] = value
            if self.reshape_reward is not None:
                self.rewards[i] = self.reshape_reward(obs, action, reward, done)
            elif self.reward_scale is not None:
                self.rewards[i] = self.reward_scale * reward
            else:
                self.rewards[i] = reward
            self.log_probs[i] = dist.log_prob(action)

            self.log_episode_return += reward
            self.log_episode_reshaped_return += self.rewards[i]
            self.log_episode_num_frames += 1
            self.dones[i] = done
            self.ep_return[i] = self.log_episode_return
            self.ep_reshaped[i] = self.log_episode_reshaped_return
            self.ep_frames[i] = self.log_episode_num_frames
            self.log_episode_return *= self.mask
            self.log_episode_reshaped_return *= self.mask
            self.log_episode_num_frames *= self.mask

        next_value = self._model(self._cur_image, self.tokens)["value"]          # base.py:192-194

        for i in reversed(range(T)):                                              # base.py:196-202
            next_mask = self.masks[i + 1] if i < T - 1 else self.mask
            next_value = self.values[i + 1] if i < T - 1 else next_value
            next_advantage = self.advantages[i + 1] if i < T - 1 else 0
            delta = self.rewards[i] + self.discount * next_value * next_mask - self.values[i]
            self.advantages[i] = delta + self.discount * self.gae_lambda * next_advantage * next_mask

        exps = TensorDict()                                                        # base.py:207-232
        image = self.images.transpose(0, 1).reshape((-1,) + tuple(self.images.shape[2:]))
        instr = self.instrs.transpose(0, 1).reshape(-1, self.instrs.shape[2])
        exps.obs = self._batch(image, instr)
        exps.memory = self.memories.transpose(0, 1).reshape(-1, *self.memories.shape[2:])
        exps.mask = self.masks.transpose(0, 1).reshape(-1).unsqueeze(1)
        exps.action = self.actions.transpose(0, 1).reshape(-1)
        exps.value = self.values.transpose(0, 1).reshape(-1)
        exps.reward = self.rewards.transpose(0, 1).reshape(-1)
        exps.advantage = self.advantages.transpose(0, 1).reshape(-1)
        exps.returnn = exps.value + exps.advantage
        exps.log_prob = self.log_probs.transpose(0, 1).reshape(-1)

        # episode statistics: one readback, replayed in the reference's (frame, env) append order
        idx = self.dones.reshape(-1).nonzero().reshape(-1)
        stats = torch.stack([self.ep_return.reshape(-1)[idx], self.ep_reshaped.reshape(-1)[idx],
                             self.ep_frames.reshape(-1)[idx]]).cpu().tolist()
        self.log_done_counter += len(stats[0])
        self.log_return.extend(stats[0])
        self.log_reshaped_return.extend(stats[1])
        self.log_num_frames.extend(stats[2])

        keep = max(self.log_done_counter, P)                                       # base.py:236-252
        log = {
            "return_per_episode": self.log_return[-keep:],
            "reshaped_return_per_episode": self.log_reshaped_return[-keep:],
            "num_frames_per_episode": self.log_num_frames[-keep:],
            "num_frames": self.num_frames,
            "episodes_done": self.log_done_counter,
        }
        self.log_done_counter = 0
        self.log_return = self.log_return[-P:]
        self.log_reshaped_return = self.log_reshaped_return[-P:]
        self.log_num_frames = self.log_num_frames[-P:]
        return exps, log
