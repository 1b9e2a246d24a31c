"""Test doubles for tests/test_rollout*.py: a deterministic actor-critic whose outputs are exactly representable
(so CPU and GPU runs can be compared bit for bit) and a CPU tensor-protocol env over oracle envs."""
import numpy as np
import torch

from babyai_amd.missions import tokenize

TOK_MAX = 72


class _Dist(object):
    def __init__(self, action):
        self.action = action

    def sample(self):
        return self.action

    def log_prob(self, a):
        return -(a.to(torch.float32) + 1) / 8


class ToyACModel(torch.nn.Module):
    """Follows the call contract of babyai/model.py:217-273 (obs.image, obs.instr, memory -> dist/value/memory)."""
    memory_size = 4
    # pickup/drop/toggle/done are rarer than moves so that tiny levels see both successes and timeouts
    TABLE = (2, 0, 2, 1, 2, 5, 2, 3, 2, 0, 4, 2, 1, 6)

    def forward(self, obs, memory):
        n = obs.image.shape[0]
        key = obs.image.reshape(n, -1).to(torch.int64).sum(1) + 3 * obs.instr.reshape(n, -1).sum(1) \
            + memory[:, 0].to(torch.int64)
        table = torch.as_tensor(self.TABLE, device=key.device)
        action = table[key % len(self.TABLE)]
        value = (key % 97).to(torch.float32) / 64
        mem = ((memory[:, :1] + (key % 5).to(torch.float32).unsqueeze(1) + 1) % 8).expand(n, self.memory_size).clone()
        return {"dist": _Dist(action), "value": value, "memory": mem, "extra_predictions": {}}


def pad_tokens(missions, width=None):
    toks = [tokenize(m) for m in missions]
    width = width or max(len(t) for t in toks)
    out = np.zeros((len(toks), width), dtype=np.int64)
    for i, t in enumerate(toks):
        out[i, :len(t)] = t
    return out


class OracleTensorEnv(object):
    """The tensor protocol of BatchedBabyAIEnv (auto-reset, penv.py:8-11) on CPU tensors over oracle envs."""

    def __init__(self, level, seeds):
        from oracle import levels as olevels
        self.envs = []
        for s in seeds:
            e = olevels.make_env(level)
            e.seed(int(s))
            self.envs.append(e)
        self.num_envs = len(self.envs)
        self.device = torch.device("cpu")
        self.instr = torch.zeros((self.num_envs, TOK_MAX), dtype=torch.uint8)
        self.image = torch.zeros((self.num_envs, 7, 7, 3), dtype=torch.uint8)
        self.direction = torch.zeros((self.num_envs,), dtype=torch.uint8)

    def enable_instr_tokens(self):
        return self.instr

    def _publish(self, obss):
        self.image.copy_(torch.as_tensor(np.stack([o["image"] for o in obss])))
        self.direction.copy_(torch.as_tensor(np.array([o["direction"] for o in obss], dtype=np.uint8)))
        self.instr.copy_(torch.as_tensor(pad_tokens([o["mission"] for o in obss], TOK_MAX).astype(np.uint8)))
        return {"image": self.image, "direction": self.direction}

    def reset(self):
        return self._publish([e.reset() for e in self.envs])

    def step(self, actions):
        obss, rewards, dones = [], [], []
        for e, a in zip(self.envs, actions.tolist()):
            o, r, d, _ = e.step(int(a))
            if d:
                o = e.reset()
            obss.append(o); rewards.append(r); dones.append(d)
        self.reward64 = torch.as_tensor(np.array(rewards, dtype=np.float64))
        self.done = torch.as_tensor(np.array(dones, dtype=np.uint8))
        return (self._publish(obss), self.reward64.to(torch.float32), self.done, {})


class _MissionList(object):
    """What BatchedBabyAIEnv puts under obs['mission']: indexable, with snapshot() (engine.py Missions)."""

    def __init__(self, texts):
        self._texts = list(texts)

    def snapshot(self):
        return self

    def __len__(self):
        return len(self._texts)

    def __getitem__(self, i):
        return self._texts[i]

    def __iter__(self):
        return iter(self._texts)


class OracleEngine(object):
    """BatchedBabyAIEnv's tensor protocol (seed / reset / step with auto-reset or freeze, reward64, lazy missions) on CPU
    tensors over oracle envs: lets the protocol adapters of babyai_amd/vec_env.py and the reference's consumers on top
    of them run in a container without a GPU.  TEST DOUBLE: the product has no CPU path."""

    def __init__(self, level, num_envs, auto_reset=True):
        from oracle import levels as olevels
        self.envs = [olevels.make_env(level) for _ in range(num_envs)]
        self.num_envs, self.auto_reset, self.pixel = num_envs, auto_reset, False
        self.device = torch.device("cpu")
        self.frozen = [False] * num_envs
        self.last = [None] * num_envs
        self.closed = False

    def seed(self, seeds):
        for e, s in zip(self.envs, seeds):
            e.seed(int(s))

    def _obs(self, obss):
        return {"image": torch.as_tensor(np.stack([o["image"] for o in obss])),
                "direction": torch.as_tensor(np.array([o["direction"] for o in obss], dtype=np.uint8)),
                "mission": _MissionList(o["mission"] for o in obss)}

    def reset(self):
        self.frozen = [False] * self.num_envs
        self.cur = [e.reset() for e in self.envs]
        return self._obs(self.cur)

    def step(self, actions):
        actions = np.asarray(actions.cpu() if hasattr(actions, "cpu") else actions).reshape(-1)
        rewards, dones = [], []
        for k, (e, a) in enumerate(zip(self.envs, actions.tolist())):
            if self.frozen[k]:
                o, r, d = self.last[k]
            else:
                o, r, d, _ = e.step(int(a))
                if d and self.auto_reset:
                    o = e.reset()
                elif d:
                    self.frozen[k] = True
                self.last[k] = (o, r, d)
            self.cur[k] = o
            rewards.append(r); dones.append(d)
        self.reward64 = torch.as_tensor(np.array(rewards, dtype=np.float64))
        self.done = torch.as_tensor(np.array(dones, dtype=np.uint8))
        return self._obs(self.cur), self.reward64.to(torch.float32), self.done, {}

    def close(self):
        self.closed = True
