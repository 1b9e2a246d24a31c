"""Drop-in adapters: the reference's vectorised-env protocols on top of `BatchedBabyAIEnv`.

The reference vectorises with one Python env per OS process:
  * training   `ParallelEnv(envs)`  babyai/rl/utils/penv.py:18-59, built inside
               `BaseAlgo.__init__` (babyai/rl/algos/base.py:54) from the env list of
               scripts/train_rl.py:53-60 (env i seeded with 100*seed + i)
  * evaluation `ManyEnvs(envs)`     babyai/evaluate.py:58-81, driven by batch_evaluate :85-140

Both speak "list of obs dicts in, zip(*results) out".  The classes below honour exactly that
surface (same method names, argument meaning, return shapes and auto-reset / freeze
behaviour) while every env lives on the GPU, so `ObssPreprocessor` (babyai/utils/format.py:100-119),
`ModelAgent.act_batch` (babyai/utils/agent.py:51-72) and `BaseAlgo.collect_experiences`
(babyai/rl/algos/base.py:110-251) consume the results unchanged.  See INTEGRATION.md for
the no-edit route (`babyai_amd.integrate.install()`).

`make(env_id, num_envs, ...)` is the batched twin of `gym.make(env_id)`.
"""
import numpy as np

from .engine import BatchedBabyAIEnv
from .levels import LEVELS, level_name


class _Box(object):
    """Stand-in for gym.spaces.Box: the reference reads `.shape` and `.high` (format.py:124-126)."""

    def __init__(self, shape):
        self.shape = shape
        self.low = np.zeros(shape, np.uint8)
        self.high = np.full(shape, 255, np.uint8)
        self.dtype = np.dtype("uint8")


class _DictSpace(object):
    """Stand-in for gym.spaces.Dict({'image': Box})."""

    def __init__(self, spaces):
        self.spaces = spaces

    def __getitem__(self, key):
        return self.spaces[key]


class _Discrete(object):
    """Stand-in for gym.spaces.Discrete(7) (model.py:154 reads `.n`)."""

    def __init__(self, n):
        self.n = n


def _spaces(pixel):
    return _DictSpace({"image": _Box((56, 56, 3) if pixel else (7, 7, 3))}), _Discrete(7)


class ObsList(object):
    """`list[dict]` view over the batched observation: item i is
    {'image': np.uint8[7,7,3] (or [56,56,3]), 'direction': int, 'mission': str}.
    Images are copied to the host once per step (one D2H of the whole batch), dicts are built lazily."""

    def __init__(self, image_host, direction_host, missions, pixel):
        self._image, self._dir, self._missions, self._pixel = image_host, direction_host, missions, pixel

    def __len__(self):
        return self._image.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        d = {"image": self._image[i], "mission": self._missions[i]}
        if not self._pixel:           # the pixel wrapper's dict has no 'direction' key
            d["direction"] = int(self._dir[i])
        return d

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class _EnvView(object):
    """What the reference reads off `envs[0]` (scripts/train_rl.py:86-99, babyai/rl/utils/penv.py:24-25)."""

    def __init__(self, vec):
        self.observation_space, self.action_space = vec.observation_space, vec.action_space


class _VecBase(object):
    def __init__(self, env_id, num_envs, device="cuda:0", pixel=False, auto_reset=True, seeds=None, engine=None):
        """`engine`: an object with BatchedBabyAIEnv's tensor protocol to run on instead of building one (tests put the
        CPU oracle there to drive the reference's consumers through this adapter code without a GPU)."""
        self.engine = engine if engine is not None else BatchedBabyAIEnv(env_id, num_envs, device=device, pixel=pixel, auto_reset=auto_reset)
        self.num_envs = num_envs
        self.pixel = pixel
        self.observation_space, self.action_space = _spaces(pixel)
        if seeds is not None:
            self.seed(seeds)

    def seed(self, seeds):
        """ManyEnvs.seed(seeds) (evaluate.py:64-65) / per-env env.seed(s) (train_rl.py:59)."""
        return self.engine.seed(seeds)

    # the list operations the reference applies to its `envs` argument: len(envs) (base.py:86), envs[0].observation_space
    # / .action_space (train_rl.py:86-99) -- so that an adapter can stand where the list of gym envs stood
    def __len__(self):
        return self.num_envs

    def __getitem__(self, i):
        if not -self.num_envs <= int(i) < self.num_envs:
            raise IndexError(i)
        return _EnvView(self)

    def _obs_list(self, obs):
        # everything an obs dict may be asked for later is copied out now: images, directions and the mission programs
        return ObsList(obs["image"].cpu().numpy(), obs["direction"].cpu().numpy(), obs["mission"].snapshot(), self.pixel)

    def reset(self):
        return self._obs_list(self.engine.reset())

    def step(self, actions):
        if hasattr(actions, "cpu") and not hasattr(actions, "data_ptr"):
            actions = np.asarray(actions)
        if not hasattr(actions, "data_ptr"):
            # the reference's env asserts on anything outside MiniGridEnv.Actions (gym_minigrid step: "unknown action");
            # these adapters speak the reference's protocol, so the engine's per-env reset command (7) is not an action here
            a = np.asarray(actions)
            if a.size and (int(a.max()) > 6 or int(a.min()) < 0):
                raise AssertionError("unknown action")
        obs, _, done, _ = self.engine.step(actions)
        reward = self.engine.reward64.cpu().numpy()      # Python floats, as the reference returns them (levelgen.py:59-61)
        done = done.cpu().numpy().astype(bool)
        n = self.num_envs
        # the reference returns zip(*results): four tuples (obs...), (reward...), (done...), (info...)
        return iter((self._obs_list(obs), tuple(float(r) for r in reward), tuple(bool(d) for d in done),
                     tuple({} for _ in range(n))))

    def render(self):
        raise NotImplementedError

    def close(self):
        self.engine.close()


class BatchedParallelEnv(_VecBase):
    """`ParallelEnv` protocol (penv.py:18-59): auto-reset -- when an env finishes, the returned obs is the
    first obs of its next episode while reward/done belong to the terminal step (penv.py:8-11,49-50)."""

    def __init__(self, env_id, num_envs, device="cuda:0", pixel=False, seeds=None, engine=None):
        super().__init__(env_id, num_envs, device=device, pixel=pixel, auto_reset=True, seeds=seeds, engine=engine)


class BatchedManyEnvs(_VecBase):
    """`ManyEnvs` protocol (evaluate.py:58-81): no auto-reset; a finished env re-emits its last
    (obs, reward, done, info) until the next reset()."""

    def __init__(self, env_id, num_envs, device="cuda:0", pixel=False, seeds=None, engine=None):
        super().__init__(env_id, num_envs, device=device, pixel=pixel, auto_reset=False, seeds=seeds, engine=engine)
        self.done = [False] * num_envs

    def reset(self):
        self.done = [False] * self.num_envs
        return super().reset()

    def step(self, actions):
        obs, reward, done, info = super().step(actions)
        self.done = list(done)
        return iter((obs, reward, done, info))


class Actions(object):
    """MiniGridEnv.Actions (names used by scripts/manual_control.py:51-74, babyai/utils/agent.py:89)."""
    left, right, forward, pickup, drop, toggle, done = range(7)


class _GridSnapshot(object):
    def __init__(self, enc):
        self._enc = enc
        self.width, self.height = enc.shape[0], enc.shape[1]

    def encode(self):
        return self._enc.copy()

    def __eq__(self, other):
        return np.array_equal(self._enc, other.encode())

    def __ne__(self, other):
        return not self == other


class SingleEnv(object):
    """The single-env protocol (babyai/levels/levelgen.py:35,49; callers babyai/evaluate.py:20-33,
    scripts/enjoy.py:56-60): `seed(int)`, `reset() -> obs`, `step(int) -> (obs, float, bool, dict)` on a batch of one.
    Mostly for tools and debugging -- one env per launch wastes the GPU."""

    actions = Actions

    def __init__(self, env_id, device="cuda:0", pixel=False, seed=None):
        self.engine = BatchedBabyAIEnv(env_id, 1, device=device, pixel=pixel, auto_reset=False)
        self.pixel = pixel
        self.observation_space, self.action_space = _spaces(pixel)
        self.step_count = 0
        self.mission = self.surface = ""
        if seed is not None:
            self.seed(seed)

    def seed(self, seed=1337):
        self.engine.seed([int(seed)])
        return [seed]

    def _one(self, obs):
        d = {"image": obs["image"][0].cpu().numpy(), "mission": obs["mission"][0]}
        if not self.pixel:
            d["direction"] = int(obs["direction"][0])
        self.mission = self.surface = d["mission"]
        return d

    @property
    def max_steps(self):
        return int(self.engine.max_steps()[0])

    @property
    def unwrapped(self):
        return self

    @property
    def grid(self):
        """Snapshot of the full grid; `.encode()` and `==` as used at levelgen.py:534-536."""
        return _GridSnapshot(self.engine.grid_encoding()[0][0])

    @property
    def agent_pos(self):
        return tuple(int(v) for v in self.engine.grid_encoding()[1][0][:2])

    @property
    def agent_dir(self):
        return int(self.engine.grid_encoding()[1][0][2])

    def reset(self):
        self.step_count = 0
        return self._one(self.engine.reset())

    def step(self, action):
        if not 0 <= int(action) <= 6:
            raise AssertionError("unknown action")            # gym_minigrid MiniGridEnv.step
        obs, _, done, _ = self.engine.step(np.array([int(action)], dtype=np.uint8))
        self.step_count += 1
        return self._one(obs), float(self.engine.reward64[0]), bool(done[0]), {}

    def bot_action(self, action_taken=None):
        """`Bot(env).replan(action_taken)` (babyai/bot.py:547-597): the expert's suggestion for the current state, or
        None where the reference bot would raise.  Call once per step (the expert keeps its plan in the engine)."""
        prev = None if action_taken is None else np.array([int(action_taken)], dtype=np.uint8)
        a = int(self.engine.bot_actions(prev)[0])
        return None if a == self.engine.BOT_GAVE_UP else a

    def close(self):
        self.engine.close()


def make(env_id, num_envs, device="cuda:0", pixel=False, auto_reset=True, seeds=None):
    """Batched twin of `gym.make(env_id)` (ids registered at babyai/levels/levelgen.py:467-493).
    Returns the tensor-level `BatchedBabyAIEnv`."""
    if level_name(env_id) not in LEVELS:
        raise KeyError("unknown / unsupported level id %r" % (env_id,))
    return BatchedBabyAIEnv(env_id, num_envs, device=device, seeds=seeds, pixel=pixel, auto_reset=auto_reset)
