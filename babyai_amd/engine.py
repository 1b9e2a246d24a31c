"""Host side of the batched BabyAI engine: ctypes binding of the C ABI (include/bbai.h)
over torch device tensors.

`BatchedBabyAIEnv` exposes the reference's env protocol in batched / tensor form:
`seed(seeds)`, `reset() -> obs`, `step(actions) -> (obs, reward, done, info)` with the same
`{image, direction, mission}` observation dict (reference: babyai/levels/levelgen.py:35-66,
obs keys babyai/utils/demos.py:57-59).  The list-of-dicts adapters that plug into the
reference's `ParallelEnv` / `ManyEnvs` call sites live in babyai_amd/vec_env.py.

PyTorch is used only for device memory, streams and (in bench.py) torch.distributed.
There is NO CPU fallback: if the HIP library is missing or no GPU is visible, construction
raises.
"""
import ctypes
import sys
import os

import numpy as np

from .levels import make_cfg
from . import missions

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BBAI_ENGINE_LIB") or os.path.join(_HERE, "libbbai_hip.so")      # (override: experiment builds)
ATLAS_PATH = os.path.join(_HERE, "data", "tile_atlas_ts8.npz")

OBS_BYTES = 147
PIX = 56
PROG_BYTES = 112
TOK_MAX = 72

_lib = None


class EngineError(RuntimeError):
    pass


def load_library():
    """dlopen the HIP engine; raises EngineError (never falls back to a CPU path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise EngineError("HIP engine %s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    # torch first: the PyTorch-ROCm wheel carries its own libamdhip64; whichever HIP runtime is mapped first serves
    # the whole process, and the engine must share torch's (device pointers and streams cross the boundary).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    P, I64, I32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.bbai_version.restype = ctypes.c_int
    lib.bbai_last_error.restype = ctypes.c_char_p
    lib.bbai_fill_layout.argtypes = [P]
    lib.bbai_create.argtypes = [P, I64, I32, ctypes.POINTER(P)]
    lib.bbai_destroy.argtypes = [P]
    lib.bbai_destroy.restype = None
    lib.bbai_seed.argtypes = [P, P, I64]
    lib.bbai_reset.argtypes = [P, P, P, P]
    lib.bbai_step.argtypes = [P, P, P, P, P, P, P, I32, P]
    if hasattr(lib, "bbai_step_render"):        # (A/B runs against older experiment builds, BBAI_ENGINE_LIB: they lack the newer entries)
        lib.bbai_step_render.argtypes = [P, P, P, P, P, P, P, I32, P, P]
    lib.bbai_set_atlas.argtypes = [P, P, I32, P]
    lib.bbai_render.argtypes = [P, P, P, P]
    lib.bbai_set_token_buffer.argtypes = [P, P]
    lib.bbai_export_state.argtypes = [P, I64, I64, P, P, P]
    lib.bbai_import_state.argtypes = [P, I64, I64, P, P, P]
    lib.bbai_get_programs.argtypes = [P, I64, I64, P]
    lib.bbai_tap.argtypes = [I64, I64, P, P, P, P, P, P, P, P, P, P, P]
    lib.bbai_tap_ids.argtypes = [I64, I64, P, P, P, P, P, P, P, P, P, P, P, P]
    if hasattr(lib, "bbai_step_tapped"):
        lib.bbai_step_tap_set.argtypes = [P, P, I64]
        lib.bbai_step_tapped.argtypes = [P, P, P, P, P, P, P, I32, P, P, P, P, P]
    lib.bbai_gae.argtypes = [I64, I32, P, P, P, P, P, ctypes.c_double, ctypes.c_double, P, P, P]
    lib.bbai_set_call_events.argtypes = [P, I32]
    lib.bbai_profile.argtypes = [P, I32]
    lib.bbai_profile_read.argtypes = [P, P, P]
    lib.bbai_checkpoint_bytes.argtypes = [P]
    lib.bbai_checkpoint_bytes.restype = I64
    lib.bbai_checkpoint_save.argtypes = [P, P, I64]
    lib.bbai_checkpoint_load.argtypes = [P, P, I64]
    lib.bbai_reset_count.argtypes = [P, P]
    lib.bbai_generator_failures.argtypes = [P, P]
    lib.bbai_bot_act.argtypes = [P, P, P, P]
    lib.bbai_bot_stats.argtypes = [P, P, P]
    lib.bbai_bot_rollout.argtypes = [P, I32, P, P, P, P, P, P, P, P, P, P]
    lib.bbai_set_done_actions.argtypes = [P, I32]
    lib.bbai_rollout.argtypes = [P, I32, P, P, P, P, P, P, I32, P, P, P]
    lib.bbai_set_option.argtypes = [P, ctypes.c_char_p, I64]
    lib.bbai_get_option.argtypes = [P, ctypes.c_char_p, ctypes.POINTER(I64)]
    lib.bbai_get_done_actions.argtypes = [P]
    _lib = lib
    return lib


EXPORTED_SYMBOLS = (
    "bbai_version", "bbai_last_error", "bbai_fill_layout", "bbai_create", "bbai_destroy", "bbai_seed",
    "bbai_reset", "bbai_step", "bbai_set_atlas", "bbai_render", "bbai_set_token_buffer", "bbai_export_state", "bbai_import_state",
    "bbai_get_programs", "bbai_reset_count", "bbai_generator_failures", "bbai_bot_act", "bbai_bot_stats",
    "bbai_checkpoint_bytes", "bbai_checkpoint_save", "bbai_checkpoint_load", "bbai_profile", "bbai_profile_read", "bbai_gae", "bbai_tap",
    "bbai_tap_ids", "bbai_set_call_events", "bbai_bot_rollout", "bbai_set_done_actions", "bbai_get_done_actions",
    "bbai_set_option", "bbai_get_option", "bbai_rollout", "bbai_step_render", "bbai_step_tap_set", "bbai_step_tapped",
)


def _check(lib, rc, what):
    if rc != 0:
        raise EngineError("%s failed (%d): %s" % (what, rc, lib.bbai_last_error().decode()))


class TapLog(ctypes.Structure):
    """include/bbai.h bbai_tap_log"""
    _fields_ = [("count", ctypes.c_int64), ("pix_count", ctypes.c_int64), ("ids_dev", ctypes.c_void_p), ("image_out", ctypes.c_void_p),
                ("dir_out", ctypes.c_void_p), ("reward64_out", ctypes.c_void_p), ("done_out", ctypes.c_void_p), ("pixels_out", ctypes.c_void_p),
                ("obs_row0", ctypes.c_int64), ("row0", ctypes.c_int64)]


class Missions(object):
    """Lazy per-env mission strings: compiled programs are fetched from HBM and rendered to
    text only when a consumer indexes them."""

    def __init__(self, env):
        self._env = env
        self._progs = None
        self._cache = {}
        self._version = env._obs_version          # the step / reset this view belongs to

    def _fetch(self):
        if self._progs is None:
            if self._version != self._env._obs_version:
                # the programs in HBM have moved on (an auto-reset may have replaced this env's mission): refuse to hand
                # out the NEXT episode's text for an observation of an earlier step
                raise EngineError("mission view of an earlier step: index obs['mission'] before the next step()/reset(), "
                                  "or take env.missions() when the observation is produced")
            self._progs = self._env.programs()
        return self._progs

    def snapshot(self):
        """Fetch the programs NOW (n x 112 bytes to the host): the view stays valid after later steps.  The list-of-dicts
        adapters do this for every observation they hand out (the reference's collectors read missions of stored
        observations many steps later, babyai/rl/algos/base.py:207-232)."""
        self._fetch()
        return self

    def __len__(self):
        return self._env.num_envs

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        s = self._cache.get(i)
        if s is None:
            s = missions.prog_surface(self._fetch()[i])
            self._cache[i] = s
        return s

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class BatchedBabyAIEnv(object):
    """N environments of one BabyAI level on one MI355X.

    Parameters
    ----------
    env_id : 'BabyAI-<Level>-v0' or '<Level>' (babyai/levels/levelgen.py:480)
    num_envs : number of parallel envs on this device
    device : torch device string / index (a ROCm GPU)
    pixel : apply RGBImgPartialObsWrapper semantics (obs image uint8[N,56,56,3])
    auto_reset : True = ParallelEnv protocol (penv.py:8-11); False = ManyEnvs protocol
                 (evaluate.py:73-81: finished envs freeze until reset())
    done_actions : the reference's BABYAI_DONE_ACTIONS verifier mode (babyai/levels/verifier.py:17,216-230: an instruction
                 succeeds only on a `done` action right after the step that completed it; include/bbai.h
                 bbai_set_done_actions).  None (default) = as the reference decides it: on iff that environment variable is
                 non-empty; True / False force it for this batch
    validate_actions : check every step()'s actions on the device first and raise AssertionError("unknown action") like
                 the reference (gym_minigrid MiniGridEnv.step) instead of treating bytes 8..255 as `done` (include/bbai.h);
                 costs one reduction and a host synchronisation per step, so it is off by default
    """

    def __init__(self, env_id, num_envs, device="cuda:0", seeds=None, pixel=False, auto_reset=True, validate_actions=False,
                 done_actions=None):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise EngineError("no ROCm GPU visible: the batched engine has no CPU path")
        self.lib = load_library()
        self.env_id = env_id
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise EngineError("device must be a ROCm GPU, got %r" % (device,))
        self.dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.pixel = bool(pixel)
        self.auto_reset = bool(auto_reset)
        self.validate_actions = bool(validate_actions)
        self.cfg = make_cfg(env_id)
        self.handle = ctypes.c_void_p()
        _check(self.lib, self.lib.bbai_create(ctypes.byref(self.cfg), self.num_envs, self.dev_index,
                                               ctypes.byref(self.handle)), "bbai_create")
        if done_actions is not None:
            _check(self.lib, self.lib.bbai_set_done_actions(self.handle, 1 if done_actions else 0), "bbai_set_done_actions")
        self.done_actions = bool(self.lib.bbai_get_done_actions(self.handle))
        n = self.num_envs
        with torch.cuda.device(self.dev_index):
            self.image = torch.zeros((n, 7, 7, 3), dtype=torch.uint8, device=self.device)
            self.direction = torch.zeros((n,), dtype=torch.uint8, device=self.device)
            self.reward = torch.zeros((n,), dtype=torch.float32, device=self.device)
            # the reference's own return value: a Python float = float64 (levelgen.py:59-61); `reward` is its f32 rounding
            self.reward64 = torch.zeros((n,), dtype=torch.float64, device=self.device)
            self.done = torch.zeros((n,), dtype=torch.uint8, device=self.device)
            self.pixels = None
            if self.pixel:
                self.pixels = torch.zeros((n, PIX, PIX, 3), dtype=torch.uint8, device=self.device)
                atlas = np.load(ATLAS_PATH)
                tiles = np.ascontiguousarray(atlas["tiles"], dtype=np.uint8)
                lut = np.ascontiguousarray(atlas["lut"], dtype=np.uint8)
                _check(self.lib, self.lib.bbai_set_atlas(self.handle, tiles.ctypes.data, tiles.shape[0],
                                                          lut.ctypes.data), "bbai_set_atlas")
        self._missions = None
        self._obs_version = 0
        self.kernel_events = None      # bench.py: list of (tag, start_event, end_event) when enabled
        self.num_actions = 7
        self.max_mission_tokens = min(TOK_MAX, missions.max_mission_tokens(self.cfg))
        self.max_steps_bound = 8 * self.cfg.room_size ** 2 * self.cfg.num_rows * self.cfg.num_cols
        if seeds is not None:
            self.seed(seeds)

    # ---- protocol ---------------------------------------------------------------------
    def seed(self, seeds):
        """env.seed(s) for every env: an int base (env i gets base + i) or a sequence."""
        if isinstance(seeds, (int, np.integer)):
            seeds = np.arange(self.num_envs, dtype=np.uint64) + np.uint64(seeds)
        seeds = np.ascontiguousarray(np.asarray(list(seeds) if not isinstance(seeds, np.ndarray) else seeds,
                                                dtype=np.uint64))
        if seeds.shape != (self.num_envs,):
            raise ValueError("need %d seeds, got shape %s" % (self.num_envs, seeds.shape))
        _check(self.lib, self.lib.bbai_seed(self.handle, seeds.ctypes.data, self.num_envs), "bbai_seed")
        self.seeds = seeds
        return list(int(s) for s in seeds[:8])

    def _ev_begin(self):
        if self.kernel_events is None:
            return None
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record(self.torch.cuda.current_stream(self.dev_index))   # same stream the kernels launch on
        return ev

    def _ev_end(self, tag, start):
        if start is None:
            return
        end = self.torch.cuda.Event(enable_timing=True)
        end.record(self.torch.cuda.current_stream(self.dev_index))
        self.kernel_events.append((tag, start, end))

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream(self.dev_index).cuda_stream)

    def _obs(self, rendered=False):
        img = self.image
        if self.pixel and rendered:
            img = self.pixels
        elif self.pixel:
            ev = self._ev_begin()
            _check(self.lib, self.lib.bbai_render(self.handle, self.image.data_ptr(), self.pixels.data_ptr(),
                                                   self._stream()), "bbai_render")
            self._ev_end("render", ev)
            img = self.pixels
        self._obs_version += 1
        self._missions = Missions(self)
        return {"image": img, "direction": self.direction, "mission": self._missions}

    def render_encoding(self, image=None, out=None):
        """RGBImgPartialObsWrapper.observation for ANY encoded batch uint8[N,7,7,3] on the device (default: the current
        one) -> uint8[N,56,56,3] (include/bbai.h bbai_render).  step() / reset() already return pixels in pixel mode; this
        is for stored or hand-made encodings."""
        image = self.image if image is None else image
        out = self.pixels if out is None else out
        _check(self.lib, self.lib.bbai_render(self.handle, image.data_ptr(), out.data_ptr(), self._stream()), "bbai_render")
        return out

    def reset(self):
        _check(self.lib, self.lib.bbai_reset(self.handle, self.image.data_ptr(), self.direction.data_ptr(),
                                              self._stream()), "bbai_reset")
        return self._obs()

    def step(self, actions, validate=None):
        torch = self.torch
        if not isinstance(actions, torch.Tensor):
            actions = np.asarray(actions)
            if actions.size and (int(actions.max()) > self.RESET_ENV or int(actions.min()) < 0):      # host data: checked for free
                raise AssertionError("unknown action")
            actions = torch.as_tensor(actions, device=self.device)
        elif self.validate_actions if validate is None else validate:
            # MiniGridEnv.step: `assert False, "unknown action"` (7 = RESET_ENV is this engine's per-env reset command)
            lo, hi = (int(v) for v in torch.stack([actions.min(), actions.max()]).tolist())
            if hi > self.RESET_ENV or lo < 0:
                raise AssertionError("unknown action")
        if actions.dtype != torch.uint8:
            # a wider integer must not WRAP into a valid action on its way to a byte (263 -> 7 = RESET_ENV, 256 -> 0 = left):
            # everything outside 0..255 becomes 255, which include/bbai.h defines -- like every unknown action -- as `done`
            if actions.dtype.is_floating_point or actions.dtype == torch.bool:
                raise TypeError("actions must be an integer tensor, got %s" % actions.dtype)
            # (ONE extra elementwise launch, then the cast: RL callers step with int64 samples on 30-us steps)
            if actions.dtype == torch.int8:
                actions = actions.to(torch.int16)
            actions = torch.where((actions & -256) != 0, 255, actions)
        if actions.dtype != torch.uint8 or actions.device != self.device or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.uint8).contiguous()
        if actions.numel() != self.num_envs:
            raise ValueError("need %d actions" % self.num_envs)
        self._actions = actions     # keep alive until the launch is consumed
        ev = self._ev_begin()
        if self.pixel and self.kernel_events is None and hasattr(self.lib, "bbai_step_render"):
            # the wrapped env's step: transition + render as ONE call (include/bbai.h bbai_step_render)
            _check(self.lib, self.lib.bbai_step_render(self.handle, actions.data_ptr(), self.image.data_ptr(),
                                                        self.direction.data_ptr(), self.reward.data_ptr(), self.reward64.data_ptr(),
                                                        self.done.data_ptr(), 1 if self.auto_reset else 0, self.pixels.data_ptr(),
                                                        self._stream()), "bbai_step_render")
            return self._obs(rendered=True), self.reward, self.done, {}
        _check(self.lib, self.lib.bbai_step(self.handle, actions.data_ptr(), self.image.data_ptr(),
                                             self.direction.data_ptr(), self.reward.data_ptr(), self.reward64.data_ptr(),
                                             self.done.data_ptr(),
                                             1 if self.auto_reset else 0, self._stream()), "bbai_step")
        self._ev_end("step", ev)
        return self._obs(), self.reward, self.done, {}

    def rollout(self, actions, tap=None, obs_row0=0, row0=0, step_tap=False):
        """An open-loop rollout (include/bbai.h bbai_rollout): `actions` uint8[T, N] on the device; T steps (+ the pixel render
        in pixel mode) are enqueued by ONE call, exactly what T calls of step() enqueue.  `tap`: a log dict as bench.py builds
        it -- device tensors "image" [rows, P, 7, 7, 3], "direction" [rows, P], "reward64" [rows, P], "done" [rows, P],
        optionally "pixels" [rows, PP, 56, 56, 3], and "ids" int64[P]: step t logs the listed envs' outputs into obs row
        obs_row0 + t / result row row0 + t.  Afterwards image / direction / reward / done (/ pixels) hold the last step's
        outputs; returns the observation dict of that step.  step_tap=True: the log's rows are written by the stepping lanes for the envs
        listed with set_step_tap() (log row k = listed env k; no "ids" needed, no pixel rows) -- with encoded observations one k_step launch then
        takes every step the look-ahead window has left (include/bbai.h bbai_rollout)."""
        torch = self.torch
        if actions.dtype != torch.uint8 or actions.device != self.device or not actions.is_contiguous() or actions.dim() != 2 \
                or actions.shape[1] != self.num_envs:
            raise ValueError("actions: contiguous uint8[T, %d] on %s" % (self.num_envs, self.device))
        T = int(actions.shape[0])
        log = None
        if tap is not None:
            P_ = int(tap["done"].shape[1])
            pix = tap.get("pixels")
            for k in ("image", "direction", "reward64", "done"):
                if not tap[k].is_contiguous():
                    raise ValueError("tap log tensors must be contiguous")
            if step_tap:
                pix = None
            log = TapLog(P_, int(pix.shape[1]) if pix is not None else 0, None if step_tap else tap["ids"].data_ptr(), tap["image"].data_ptr(), tap["direction"].data_ptr(),
                         tap["reward64"].data_ptr(), tap["done"].data_ptr(), pix.data_ptr() if pix is not None else None, int(obs_row0), int(row0))
            if tap["image"].shape[0] < obs_row0 + T or tap["done"].shape[0] < row0 + T:
                raise ValueError("tap log has too few rows for %d steps" % T)
        self._actions = actions
        _check(self.lib, self.lib.bbai_rollout(self.handle, T, actions.data_ptr(), self.image.data_ptr(), self.direction.data_ptr(),
                                                self.reward.data_ptr(), self.reward64.data_ptr(), self.done.data_ptr(), 1 if self.auto_reset else 0,
                                                self.pixels.data_ptr() if self.pixel else None, ctypes.byref(log) if log is not None else None,
                                                self._stream()), "bbai_rollout")
        self._obs_version += 1
        self._missions = Missions(self)
        return {"image": self.pixels if self.pixel else self.image, "direction": self.direction, "mission": self._missions}

    # ---- state access -------------------------------------------------------------------
    def programs(self, first=0, count=None):
        if not self.handle:
            raise EngineError("engine is closed (mission strings are fetched lazily: read them before close())")
        count = self.num_envs - first if count is None else count
        out = np.zeros((count, PROG_BYTES), dtype=np.uint8)
        _check(self.lib, self.lib.bbai_get_programs(self.handle, first, count, out.ctypes.data), "bbai_get_programs")
        return out

    def missions(self):
        return list(Missions(self))

    def export_state(self, first=0, count=None):
        count = self.num_envs - first if count is None else count
        rec = np.zeros((count, self.cfg.rec_bytes), dtype=np.uint8)
        hot = np.zeros((count, 16), dtype=np.uint8)
        stale = np.zeros((count,), dtype=np.uint64)
        _check(self.lib, self.lib.bbai_export_state(self.handle, first, count, rec.ctypes.data, hot.ctypes.data,
                                                     stale.ctypes.data), "bbai_export_state")
        return rec, hot, stale

    def grid_encoding(self, first=0, count=None):
        """uint8[count, W, H, 3]: the full-grid `env.grid.encode()` (type, colour, state per cell, indexed [x, y]) and
        the agent poses int[count, 3] = (x, y, dir) -- the `unwrapped.grid` / `agent_pos` view that the reference's own
        level test compares between same-seed envs (babyai/levels/levelgen.py:531-537)."""
        rec, hot, _ = self.export_state(first, count)
        c = self.cfg
        e = rec[:, :c.ES * c.EH].reshape(-1, c.EH, c.ES)[:, 5:5 + c.H, 5:5 + c.W].transpose(0, 2, 1)
        return np.stack([e & 7, (e >> 3) & 7, e >> 6], axis=-1).astype(np.uint8), hot[:, :3].astype(np.int64)

    def import_state(self, rec, hot, stale, first=0):
        rec = np.ascontiguousarray(rec, dtype=np.uint8)
        hot = np.ascontiguousarray(hot, dtype=np.uint8)
        stale = np.ascontiguousarray(stale, dtype=np.uint64)
        count = rec.shape[0]
        assert rec.shape == (count, self.cfg.rec_bytes) and hot.shape == (count, 16) and stale.shape == (count,)
        _check(self.lib, self.lib.bbai_import_state(self.handle, first, count, rec.ctypes.data, hot.ctypes.data,
                                                     stale.ctypes.data), "bbai_import_state")

    def save_checkpoint(self):
        """The whole batch as one host blob (np.uint8): live state, RNG streams, look-ahead ring, window bookkeeping,
        counters, the expert's plans -- everything `load_checkpoint` needs to continue bit-identically."""
        nbytes = int(self.lib.bbai_checkpoint_bytes(self.handle))
        blob = np.empty(nbytes, dtype=np.uint8)
        _check(self.lib, self.lib.bbai_checkpoint_save(self.handle, blob.ctypes.data, nbytes), "bbai_checkpoint_save")
        return blob

    def load_checkpoint(self, blob):
        """Continue a saved run in this handle (same level, batch size and look-ahead depth).  The current observation
        is not part of the blob: callers that need it keep `image` / `direction` next to it (or step once)."""
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        _check(self.lib, self.lib.bbai_checkpoint_load(self.handle, blob.ctypes.data, blob.size), "bbai_checkpoint_load")
        if getattr(self, "instr", None) is not None:        # refill the caller-owned token rows from the loaded programs
            _check(self.lib, self.lib.bbai_set_token_buffer(self.handle, self.instr.data_ptr()), "bbai_set_token_buffer")

    def enable_instr_tokens(self):
        """Keep `self.instr` (uint8[N, 72] on the device) filled with the mission token ids of the current
        episodes -- the tensor fast path that replaces per-step regex tokenisation (format.py:59-75)."""
        if getattr(self, "instr", None) is None:
            with self.torch.cuda.device(self.dev_index):
                self.instr = self.torch.zeros((self.num_envs, TOK_MAX), dtype=self.torch.uint8, device=self.device)
            _check(self.lib, self.lib.bbai_set_token_buffer(self.handle, self.instr.data_ptr()), "bbai_set_token_buffer")
        return self.instr

    BOT_GAVE_UP = 255
    RESET_ENV = 7          # step() "action": abandon the episode of that env (include/bbai.h BBAI_ACTION_RESET_ENV)

    def bot_actions(self, prev_actions=None, out=None):
        """One decision of the reference's expert (babyai/bot.py Bot.replan) for every env, on the device:
        uint8[N] suggested actions, BOT_GAVE_UP (255) where the reference bot would have raised.  `prev_actions` = the
        actions the envs were really stepped with since the last call (advising mode), None = the suggestions."""
        torch = self.torch
        if out is None:
            if getattr(self, "_bot_out", None) is None:
                self._bot_out = torch.zeros((self.num_envs,), dtype=torch.uint8, device=self.device)
            out = self._bot_out
        prev_ptr = None
        if prev_actions is not None:
            if not isinstance(prev_actions, torch.Tensor):
                prev_actions = torch.as_tensor(np.asarray(prev_actions), device=self.device)
            prev_actions = prev_actions.to(device=self.device, dtype=torch.uint8).contiguous()
            if prev_actions.numel() != self.num_envs:
                raise ValueError("need %d previous actions" % self.num_envs)
            self._bot_prev = prev_actions        # keep alive until the launch is consumed
            prev_ptr = prev_actions.data_ptr()
        _check(self.lib, self.lib.bbai_bot_act(self.handle, prev_ptr, out.data_ptr(), self._stream()), "bbai_bot_act")
        return out

    def bot_rollout(self, steps, tokens=False):
        """`steps` expert decisions + auto-reset steps for every env with no host round trip in between (include/bbai.h
        bbai_bot_rollout: the inner loop of scripts/make_agent_demos.py:71-137).  Returns device tensors [steps, N, ...]:
        image / direction (and tokens, uint8[steps, N, 72] mission token ids, if asked) = what the expert decided on at
        each step, action = its decision (RESET_ENV where it gave up: gave_up = 1), reward, done = the step's results.
        Afterwards self.image / self.direction hold the current observation as after step()."""
        torch = self.torch
        T, n = int(steps), self.num_envs
        if not self.auto_reset:
            raise EngineError("bot_rollout steps with ParallelEnv semantics: construct the env with auto_reset=True")
        with torch.cuda.device(self.dev_index):
            u8 = dict(dtype=torch.uint8, device=self.device)
            out = {"image": torch.empty((T, n, 7, 7, 3), **u8), "direction": torch.empty((T, n), **u8),
                   "action": torch.empty((T, n), **u8), "reward": torch.empty((T, n), dtype=torch.float32, device=self.device),
                   "done": torch.empty((T, n), **u8), "gave_up": torch.empty((T, n), **u8)}
            tok_ptr = None
            if tokens:
                self.enable_instr_tokens()
                out["tokens"] = torch.empty((T, n, TOK_MAX), **u8)
                tok_ptr = out["tokens"].data_ptr()
        _check(self.lib, self.lib.bbai_bot_rollout(self.handle, T, self.image.data_ptr(), self.direction.data_ptr(),
                                                    out["image"].data_ptr(), out["direction"].data_ptr(), tok_ptr,
                                                    out["action"].data_ptr(), out["reward"].data_ptr(), out["done"].data_ptr(),
                                                    out["gave_up"].data_ptr(), self._stream()), "bbai_bot_rollout")
        self._obs_version += 1
        return out

    def bot_stats(self):
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _check(self.lib, self.lib.bbai_bot_stats(self.handle, ctypes.byref(a), ctypes.byref(b)), "bbai_bot_stats")
        return {"gave_up": int(a.value), "capacity": int(b.value)}

    def tap(self, image_out, dir_out, reward64_out, done_out, pixels_out=None, ids=None):
        """Copy the current outputs of the first len(done_out) envs -- or of the envs `ids` (int64 device tensor, any order)
        -- into the given log rows, and the pixels of the first len(pixels_out) of them, with one launch on the current
        stream (bench.py's in-run parity tap)."""
        pp = 0 if pixels_out is None else int(pixels_out.shape[0])
        if self.torch.cuda.current_device() != self.dev_index:      # handle-free entry point: launches on the CURRENT device
            with self.torch.cuda.device(self.dev_index):
                return self.tap(image_out, dir_out, reward64_out, done_out, pixels_out, ids)
        count = int(done_out.shape[0])
        src = (self.image.data_ptr(), self.direction.data_ptr(), self.reward64.data_ptr(), self.done.data_ptr(),
               self.pixels.data_ptr() if pp else None)
        dst = (image_out.data_ptr(), dir_out.data_ptr(), reward64_out.data_ptr(), done_out.data_ptr(),
               pixels_out.data_ptr() if pp else None, self._stream())
        if ids is None:
            _check(self.lib, self.lib.bbai_tap(count, pp, *src, *dst), "bbai_tap")
        else:
            if ids.dtype != self.torch.int64 or ids.device != self.device or ids.numel() != count or not ids.is_contiguous():
                raise ValueError("ids: contiguous int64[%d] on %s" % (count, self.device))
            _check(self.lib, self.lib.bbai_tap_ids(count, pp, ids.data_ptr(), *src, *dst), "bbai_tap_ids")

    def set_step_tap(self, ids):
        """The envs step_tapped() logs (include/bbai.h bbai_step_tap_set): any order, no duplicates; log row k = env ids[k].  None / empty clears."""
        ids = np.ascontiguousarray(np.asarray([] if ids is None else ids, dtype=np.int64))
        _check(self.lib, self.lib.bbai_step_tap_set(self.handle, ids.ctypes.data if len(ids) else None, len(ids)), "bbai_step_tap_set")
        self._step_tap = len(ids)

    def step_tapped(self, actions, image_out, dir_out, reward64_out, done_out):
        """step(actions) + the listed envs' outputs of this step into the given log rows (uint8[count, 7, 7, 3], uint8[count], float64[count],
        uint8[count] on the device) -- written by the stepping lanes themselves, no launch behind the step (include/bbai.h bbai_step_tapped).
        `actions`: uint8[num_envs] on the device."""
        if actions.dtype != self.torch.uint8 or actions.device != self.device or not actions.is_contiguous() or actions.numel() != self.num_envs:
            raise ValueError("actions: contiguous uint8[%d] on %s" % (self.num_envs, self.device))
        if int(done_out.shape[0]) != getattr(self, "_step_tap", 0):
            raise ValueError("log rows for %d envs, set_step_tap() listed %d" % (int(done_out.shape[0]), getattr(self, "_step_tap", 0)))
        self._actions = actions
        _check(self.lib, self.lib.bbai_step_tapped(self.handle, actions.data_ptr(), self.image.data_ptr(), self.direction.data_ptr(),
                                                    self.reward.data_ptr(), self.reward64.data_ptr(), self.done.data_ptr(),
                                                    1 if self.auto_reset else 0, image_out.data_ptr(), dir_out.data_ptr(),
                                                    reward64_out.data_ptr(), done_out.data_ptr(), self._stream()), "bbai_step_tapped")
        self._obs_version += 1
        return self.reward, self.done

    def set_call_events(self, enable=True):
        """Record the handle's completion event at the end of every call, so that a caller may destroy a stream it used
        for this env and come back on another one (include/bbai.h bbai_set_call_events; off by default: ~3 us per call)."""
        _check(self.lib, self.lib.bbai_set_call_events(self.handle, 1 if enable else 0), "bbai_set_call_events")

    def set_option(self, name, value):
        """A performance knob of the live handle by name (include/bbai.h bbai_set_option: launch shapes, render input,
        priorities -- never semantics).  What measurements alternate inside one process (tools/ab.py)."""
        _check(self.lib, self.lib.bbai_set_option(self.handle, name.encode(), int(value)), "bbai_set_option(%s)" % name)
    
    def get_option(self, name):
        """A knob of the live handle (the names of set_option), or "lookahead_period" (the refill period chosen at create)."""
        v = ctypes.c_int64(0)
        _check(self.lib, self.lib.bbai_get_option(self.handle, name.encode(), ctypes.byref(v)), "bbai_get_option(%s)" % name)
        return int(v.value)

    def profile(self, enable=True):
        """Bracket every k_step / k_consume / k_render launch with HIP events on its launch stream (bench.py)."""
        _check(self.lib, self.lib.bbai_profile(self.handle, 1 if enable else 0), "bbai_profile")

    def profile_resume(self):
        """Bracket launches again, adding to the totals kept since profile(True)."""
        _check(self.lib, self.lib.bbai_profile(self.handle, 2), "bbai_profile")

    def profile_pause(self):
        """Stop bracketing launches; the totals stay readable."""
        _check(self.lib, self.lib.bbai_profile(self.handle, 0), "bbai_profile")

    def profile_read(self):
        """{kernel: (average ms per launch, launches)} since profile(True)."""
        ms = (ctypes.c_double * 3)()
        cnt = (ctypes.c_int64 * 3)()
        _check(self.lib, self.lib.bbai_profile_read(self.handle, ms, cnt), "bbai_profile_read")
        return {k: (ms[i] / cnt[i] if cnt[i] else None, int(cnt[i])) for i, k in enumerate(("k_step", "k_consume", "k_render"))}

    def reset_count(self):
        v = ctypes.c_uint64(0)
        _check(self.lib, self.lib.bbai_reset_count(self.handle, ctypes.byref(v)), "bbai_reset_count")
        return int(v.value)

    def generator_failures(self):
        v = ctypes.c_uint64(0)
        _check(self.lib, self.lib.bbai_generator_failures(self.handle, ctypes.byref(v)), "bbai_generator_failures")
        return int(v.value)

    def max_steps(self):
        """Per-env max_steps of the current episodes (levelgen.py:42-45)."""
        _, hot, _ = self.export_state()
        return hot[:, 6].astype(np.int32) | (hot[:, 7].astype(np.int32) << 8)

    def gate_timeouts(self):
        """Window gates that gave up waiting for a look-ahead refill (bbai_engine.hip k_gate): must be 0 -- anything else means a
        refill was lost and the batch's results are void.  Synchronises."""
        return self.get_option("gate_timeouts")

    def gate_fault(self):
        """The sticky flag a timed-out window gate leaves in pinned host memory (no synchronisation).  Once it is set, step() / reset() /
        rollout() raise EngineError at the next call (the C entry points return BBAI_ERR_STATE) -- the error surfaces where it happens, not
        at close()."""
        return bool(self.get_option("gate_fault"))

    def close(self):
        """Destroy the handle.  A gate time-out is reported by the stepping calls themselves (gate_fault); close() only warns about one that
        no later call had the chance to report -- it never raises over an exception that is already in flight."""
        if getattr(self, "handle", None) is not None and self.handle:
            fault = False
            try:
                fault = self.gate_fault()
            except Exception:
                pass
            self.lib.bbai_destroy(self.handle)
            self.handle = ctypes.c_void_p()
            if fault and sys.exc_info()[0] is None:
                raise EngineError("a window gate of this batch timed out waiting for a look-ahead refill: the results since are void")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
