"""Device-resident experience collection (second half).

What it produces is what `BaseAlgo.collect_experiences` produces (babyai/rl/algos/base.py:131-260): the `exps` fields in
(env-major, frame-minor) order and the `logs` dict -- the contract `PPOAlgo.update_parameters`-style code consumes, pinned
to the reference's own function by tests/test_rollout.py.  How it gets there is laid out for the device instead:

  * every rollout buffer is ENV-MAJOR `[P, T, ...]` from the start -- the order the reference only reaches by
    transposing and copying every `[T, P]` buffer at the end (base.py:207-232); `exps.*` are views, nothing is copied;
  * advantages are ONE reverse scan per env in a small HIP kernel (`bbai_gae`, lane = env) instead of T passes of five
    tensor ops over `[P]` slices (base.py:196-202) -- bit-identical float32 arithmetic;
  * observations, rewards and done flags already are device tensors: no list-of-dicts preprocess, no
    `action.cpu().numpy()`, no pipe round-trips, no `torch.tensor(list)` uploads; the model sees a fixed instruction
    width per level, so there is no host synchronisation per frame at all;
  * episode statistics are written per frame on the device and read back ONCE per rollout (the reference calls
    `.item()` per finished episode per frame, base.py:171-176);
  * rewards are shaped from the float64 reward (`reward_scale * reward64`, then float32), exactly what
    `torch.tensor([reshape_reward(...)])` over Python floats yields (base.py:162-167, scripts/train_rl.py:104).

    env  = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", 4096, pixel=False, seeds=...)
    roll = DeviceRollout(env, acmodel, num_frames_per_proc=40, discount=0.99, gae_lambda=0.99, reward_scale=20.)
    exps, logs = roll.collect_experiences()

ALIASING: `exps.*` are VIEWS of the collector's persistent `[P, T]` buffers (the reference returns fresh transposed copies,
base.py:207-232): the next `collect_experiences()` overwrites them in place.  A caller that keeps experiences across
collections (asynchronous updates, replay or auxiliary buffers, debugging) passes `copy=True` and gets its own tensors.

`acmodel(obs, memory)` follows babyai/model.py:217-273 ('dist' with sample() / log_prob(), 'value', 'memory';
`.memory_size`).  `env` is any object with the tensor protocol of `BatchedBabyAIEnv` (`num_envs`, `device`, `reset()`,
`step(uint8 actions)`, `enable_instr_tokens()`, optionally `reward64`); the collector holds no environment logic.
`reshape_reward` may instead be any tensor-wise callable `(obs, action, reward, done) -> tensor`; `aux_info` is not
carried (the engine's `info` dict is empty).
"""
import ctypes

from .preprocess import TensorDict


def gae_env_major(rewards, values, masks, last_mask, last_value, discount, gae_lambda, advantage, returnn):
    """advantage / returnn [P, T] from env-major float32 buffers.  ROCm tensors: the k_gae kernel of the engine library
    (include/bbai.h bbai_gae).  Host tensors (the CPU pin test against the reference): the same scan in torch ops."""
    P, T = rewards.shape
    if rewards.is_cuda:
        from .engine import load_library, _check
        import torch
        lib = load_library()
        for t in (rewards, values, masks, last_mask, last_value, advantage, returnn):
            assert t.is_contiguous() and t.dtype == torch.float32
        with torch.cuda.device(rewards.device):        # handle-free entry point: launches on the CURRENT device
            stream = ctypes.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream)
            _check(lib, lib.bbai_gae(P, T, rewards.data_ptr(), values.data_ptr(), masks.data_ptr(), last_mask.data_ptr(),
                                     last_value.data_ptr(), float(discount), float(gae_lambda), advantage.data_ptr(),
                                     returnn.data_ptr(), stream), "bbai_gae")
        return
    next_value, next_mask, next_adv = last_value, last_mask, 0
    for i in reversed(range(T)):
        delta = rewards[:, i] + discount * next_value * next_mask - values[:, i]
        advantage[:, i] = delta + discount * gae_lambda * next_adv * next_mask
        next_value, next_mask, next_adv = values[:, i], masks[:, i], advantage[:, i]
    returnn.copy_(values + advantage)


class DeviceRollout(object):
    def __init__(self, env, acmodel, num_frames_per_proc, discount, gae_lambda, reward_scale=None,
                 reshape_reward=None, recurrence=1):
        import torch
        self.torch = torch
        assert num_frames_per_proc % recurrence == 0                 # base.py:73
        assert reward_scale is None or reshape_reward is None
        self.env, self.acmodel = env, acmodel
        self.num_frames_per_proc = T = int(num_frames_per_proc)
        self.discount, self.gae_lambda = discount, gae_lambda
        self.reward_scale, self.reshape_reward, self.recurrence = reward_scale, reshape_reward, recurrence
        self.device = dev = env.device
        self.num_procs = P = env.num_envs
        self.num_frames = T * P

        self.tokens = env.enable_instr_tokens()                      # uint8[P, L], kept current by the engine
        obs = env.reset()                                            # base.py:79
        self._cur_image = obs["image"]                               # engine-owned buffer, overwritten by step()
        # instruction width the model sees per frame: a per-level constant when the env knows one, else the buffer's
        self.width = int(getattr(env, "max_mission_tokens", self.tokens.shape[1]))
        f32 = dict(dtype=torch.float32, device=dev)
        self.images = torch.zeros((P, T) + tuple(obs["image"].shape[1:]), dtype=torch.uint8, device=dev)
        self.instrs = torch.zeros((P, T, self.tokens.shape[1]), dtype=torch.uint8, device=dev)
        self.memory = torch.zeros(P, acmodel.memory_size, **f32)
        self.memories = torch.zeros(P, T, acmodel.memory_size, **f32)
        self.mask = torch.ones(P, **f32)
        self.masks = torch.zeros(P, T, **f32)
        self.actions = torch.zeros(P, T, device=dev, dtype=torch.int)
        self.values = torch.zeros(P, T, **f32)
        self.rewards = torch.zeros(P, T, **f32)
        self.log_probs = torch.zeros(P, T, **f32)
        self.advantages = torch.zeros(P, T, **f32)
        self.returns = torch.zeros(P, T, **f32)
        # running per-env episode sums + their value at every frame (what the reference appends when done)
        self.ep = torch.zeros(3, P, **f32)                            # return, reshaped return, frames
        self.ep_at = torch.zeros(3, T, P, **f32)
        self.dones = torch.zeros(T, P, device=dev, dtype=torch.uint8)
        self.log_done_counter = 0
        self.log_return = [0] * P
        self.log_reshaped_return = [0] * P
        self.log_num_frames = [0] * P

    def _batch(self, image, instr):
        """RawImagePreprocessor + InstructionsPreprocessor (format.py:44-82) on device tensors."""
        torch = self.torch
        return TensorDict(image=image.to(torch.float32), instr=instr.to(torch.int64))

    def _model(self, image, instr):
        with self.torch.no_grad():
            return self.acmodel(self._batch(image, instr[..., :self.width]), self.memory * self.mask.unsqueeze(1))

    def _shaped(self, obs, action, reward, done):
        torch = self.torch
        if self.reshape_reward is not None:
            return self.reshape_reward(obs, action, reward, done)
        if self.reward_scale is not None:
            r64 = getattr(self.env, "reward64", None)
            return (self.reward_scale * r64).to(torch.float32) if r64 is not None else self.reward_scale * reward
        return reward

    def collect_experiences(self, copy=False):
        """One rollout of T frames per env.  `copy=False`: `exps` aliases the collector's buffers until the next call (see
        the module docstring); `copy=True`: every field is cloned, as the reference's fresh tensors are."""
        torch = self.torch
        T, P = self.num_frames_per_proc, self.num_procs
        for i in range(T):
            self.images[:, i].copy_(self._cur_image)                 # the obs the action is chosen on
            self.instrs[:, i].copy_(self.tokens)
            res = self._model(self.images[:, i], self.instrs[:, i])
            action = res["dist"].sample()
            obs, reward, done, _ = self.env.step(action.to(torch.uint8))
            self._cur_image = obs["image"]
            self.memories[:, i] = self.memory
            self.memory = res["memory"]
            self.masks[:, i] = self.mask
            self.mask = 1 - done.to(torch.float)
            self.actions[:, i] = action
            self.values[:, i] = res["value"]
            self.rewards[:, i] = self._shaped(obs, action, reward, done)
            self.log_probs[:, i] = res["dist"].log_prob(action)
            self.ep[0] += reward
            self.ep[1] += self.rewards[:, i]
            self.ep[2] += 1
            self.dones[i] = done
            self.ep_at[:, i] = self.ep
            self.ep *= self.mask
        last_value = self._model(self._cur_image, self.tokens)["value"].to(torch.float32).contiguous()      # base.py:192-194
        gae_env_major(self.rewards, self.values, self.masks, self.mask.contiguous(), last_value, self.discount, self.gae_lambda,
                      self.advantages, self.returns)

        exps = TensorDict()                                          # views of the env-major buffers: (P * T) rows
        instr = self.instrs.reshape(P * T, -1)
        length = int((instr != 0).sum(dim=-1).max().item()) if instr.numel() else 0      # one sync per rollout
        exps.obs = self._batch(self.images.reshape((P * T,) + tuple(self.images.shape[2:])), instr[:, :max(length, 1)])
        exps.memory = self.memories.reshape(P * T, -1)
        exps.mask = self.masks.reshape(P * T, 1)
        exps.action = self.actions.reshape(-1)
        exps.value = self.values.reshape(-1)
        exps.reward = self.rewards.reshape(-1)
        exps.advantage = self.advantages.reshape(-1)
        exps.returnn = self.returns.reshape(-1)
        exps.log_prob = self.log_probs.reshape(-1)
        if copy:
            obs = exps.obs
            exps.obs = TensorDict(image=obs.image.clone(), instr=obs.instr.clone())
            for k in ("memory", "mask", "action", "value", "reward", "advantage", "returnn", "log_prob"):
                setattr(exps, k, getattr(exps, k).clone())

        # episode statistics: one readback, in the reference's (frame, env) append order
        idx = self.dones.reshape(-1).nonzero().reshape(-1)
        stats = self.ep_at.reshape(3, -1)[:, idx].cpu().tolist()
        self.log_done_counter += len(stats[0])
        self.log_return.extend(stats[0])
        self.log_reshaped_return.extend(stats[1])
        self.log_num_frames.extend(stats[2])
        keep = max(self.log_done_counter, P)
        log = {"return_per_episode": self.log_return[-keep:], "reshaped_return_per_episode": self.log_reshaped_return[-keep:],
               "num_frames_per_episode": self.log_num_frames[-keep:], "num_frames": self.num_frames,
               "episodes_done": self.log_done_counter}
        self.log_done_counter = 0
        self.log_return, self.log_reshaped_return = self.log_return[-P:], self.log_reshaped_return[-P:]
        self.log_num_frames = self.log_num_frames[-P:]
        return exps, log
