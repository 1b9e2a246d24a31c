"""ctypes access to tests/hostsim/libhostsim.so -- the engine's per-env core (generator, step,
verifier, observation) compiled for the host CPU with a single lane.  Test harness only."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostsim", "hostsim.cpp")
LIB = os.path.join(HERE, "hostsim", "libhostsim.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        import sys
        sys.path.insert(0, os.path.dirname(HERE))
        import __graft_entry__
        __graft_entry__.build_hostsim()          # content-stamped: rebuilds iff a source byte changed
        L = ctypes.CDLL(LIB)
        P = ctypes.c_void_p
        L.hs_seed.argtypes = [ctypes.c_uint64, P]
        L.hs_generate.argtypes = [P, P, P, P, P]
        L.hs_generate_lane.argtypes = [P, P, P, P, P]
        L.hs_generate_lane2.argtypes = [P, P, P, P, P, P, P, ctypes.c_int]
        L.hs_untemper.argtypes = [ctypes.c_uint32]
        L.hs_untemper.restype = ctypes.c_uint32
        L.hs_temper.argtypes = [ctypes.c_uint32]
        L.hs_temper.restype = ctypes.c_uint32
        L.hs_step.argtypes = [P, P, P, P, ctypes.c_int, P]
        L.hs_step64.argtypes = [P, P, P, P, ctypes.c_int, P]
        L.hs_step64_prefetch.argtypes = [P, P, P, P, ctypes.c_int, P, P]
        L.hs_observe.argtypes = [P, P, P, P]
        L.hs_observe_perm.argtypes = [P, P, P, ctypes.c_int, P]
        L.hs_bot_set_aligned.argtypes = [ctypes.c_int]
        L.hs_cpl_ok.argtypes = [P]
        L.hs_observe_cpl.argtypes = [P, P, P, ctypes.c_int, P, P]
        L.hs_cid_lookup.argtypes = [P, ctypes.c_int, ctypes.c_int]
        L.hs_fill_layout.argtypes = [P]
        L.hs_start_carry.argtypes = [P, P, P, P]
        L.hs_bot_decide.argtypes = [P, P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.hs_bot_set_eager.argtypes = [ctypes.c_int]
        L.hs_bot_set_lanes.argtypes = [ctypes.c_int]
        L.hs_bot_dead_reason.argtypes = [P]
        L.hs_bot_stack_depth.argtypes = [P]
        _lib = L
    return _lib


class HostEnv(object):
    """One env of the C++ core on the host: seed / reset / step / observe."""

    def __init__(self, cfg, seed):
        self.L = lib()
        self.cfg = cfg
        self.mt = np.zeros(624, np.uint32)
        self.L.hs_seed(int(seed), self.mt.ctypes.data)
        self.mti = ctypes.c_int32(624)
        self.rec = np.zeros(cfg.rec_bytes, np.uint8)
        self.hot = np.zeros(16, np.uint8)
        self.hot[14] = 0xFF              # last_locked = none
        self.stale = ctypes.c_uint64(0)
        self.out = np.zeros(147, np.uint8)

    def reset(self):
        self.L.hs_generate(ctypes.byref(self.cfg), self.mt.ctypes.data, ctypes.byref(self.mti),
                           self.rec.ctypes.data, self.hot.ctypes.data)
        self.stale.value = 0
        img = self.observe()
        # PutNext*Carrying: the object moves into the agent's hands right after the first observation
        self.L.hs_start_carry(ctypes.byref(self.cfg), self.rec.ctypes.data, self.hot.ctypes.data, ctypes.byref(self.stale))
        return img

    def observe(self):
        if self.prefetch_order:     # ... and the observation through k_step's register pipeline (bbai_view.hpp view_cells_perm + encode_cells)
            self.L.hs_observe_perm(ctypes.byref(self.cfg), self.rec.ctypes.data, self.hot.ctypes.data, -1, self.out.ctypes.data)
        else:
            self.L.hs_observe(ctypes.byref(self.cfg), self.rec.ctypes.data, self.hot.ctypes.data, self.out.ctypes.data)
        return self.out.reshape(7, 7, 3).copy()

    prefetch_order = False      # True: step in k_step's order of operations (bbai_step.hpp step_env_prefetch)

    def step(self, action):
        rew = ctypes.c_double(0)
        if self.prefetch_order:
            d = self.L.hs_step64_prefetch(ctypes.byref(self.cfg), self.rec.ctypes.data, self.hot.ctypes.data,
                                          ctypes.byref(self.stale), int(action), ctypes.byref(rew), None)
        else:
            d = self.L.hs_step64(ctypes.byref(self.cfg), self.rec.ctypes.data, self.hot.ctypes.data,
                                 ctypes.byref(self.stale), int(action), ctypes.byref(rew))
        self.last_reward64 = rew.value            # the f64 the kernel core computes; the f32 output is its rounding
        return self.observe(), np.float32(rew.value), bool(d)

    @property
    def agent(self):
        return int(self.hot[0]), int(self.hot[1]), int(self.hot[2])

    @property
    def max_steps(self):
        return int(self.hot[6]) | int(self.hot[7]) << 8

    @property
    def mission(self):
        from babyai_amd.missions import prog_surface
        return prog_surface(self.rec[self.cfg.off_prog:self.cfg.off_prog + 112])

    def grid_bytes(self):
        c = self.cfg
        return self.rec[:c.ES * c.EH].reshape(c.EH, c.ES)[5:5 + c.H, 5:5 + c.W]


class HostBot(object):
    """The expert of babyai_amd/csrc/bbai_bot.hpp on one HostEnv (host build of the same header)."""

    def __init__(self, env, stack_cap=48):
        self.env = env
        self.stack_cap = stack_cap
        self.state = np.zeros(env.L.hs_bot_state_bytes(stack_cap), np.uint8)

    def decide(self, first, action_taken=None):
        """Bot.replan(action_taken): suggested action, or None once the bot gave up (bot.py raises)."""
        e = self.env
        a = e.L.hs_bot_decide(ctypes.byref(e.cfg), e.rec.ctypes.data, e.hot.ctypes.data, ctypes.byref(e.stale),
                              self.state.ctypes.data, self.stack_cap, 1 if first else 0,
                              -1 if action_taken is None else int(action_taken))
        return None if a == 255 else a

    @property
    def dead_reason(self):
        return int(self.env.L.hs_bot_dead_reason(self.state.ctypes.data))

    @property
    def stack_depth(self):
        return int(self.env.L.hs_bot_stack_depth(self.state.ctypes.data))
