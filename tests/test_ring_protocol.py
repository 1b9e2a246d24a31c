"""The look-ahead ring's window protocol as a model (host logic, no GPU): why D = 2B slots are enough for the classic state layout and
D = 2B + 1 for the in-place one, and why the step stream may run AHEAD of the refills as far as k_gate lets it
(babyai_amd/csrc/bbai_engine.hip: window_begin / window_end, k_gate / k_compact / k_mark, consume_env, advance_finish, k_pregen;
DESIGN.md section 5).

The engine generates every env's levels ahead of need into a ring of D slots.  Consume-ticks (one reset() or one auto-resetting step)
are grouped into windows of B ticks; at the end of window w ONE refill launch regenerates the slots the window freed (per env:
`pending` consecutive slots from `first_slot`, kept per window buffer; NWIN buffers).  Refills land in order, at any time after their
launch.  At the start of window x the gate holds the step stream until, with r = the number of refills that have landed,

  (a) x - r <= NWIN - 1                      (the buffer of window x is no longer in use by an unfinished refill), and
  (b) sum of M_v over v = r .. x - 1 <= B    (M_v = the most often ONE env finished in window v, at least 1),

because an env's unrefilled slots are at most that sum, so it still has 2B - B = B ready levels, and a window consumes at most B.
(Rounds 1-4 waited for r >= x - 1: the special case with one open window.)  The model below runs SEVERAL envs with different finish
patterns under refills that land as LATE as the gate allows (only when the gate forces them) or at random moments, and checks

  * an env always finds the next level of ITS sequence in the slot it moves to, and that slot's refill has landed;
  * a refill never writes a slot that still holds a level the env has not played (classic) / the env's live level (in-place);
  * no window starts on a buffer whose previous refill has not landed;
  * the in-place layout's first reset is an ordinary move out of the empty slot D - 1;

and that a ring ONE slot shallower, or a gate that admits a sum of B + 1, breaks exactly these properties (the rule is tight).
"""
import numpy as np
import pytest

NWIN = 34            # bbai_engine.hip NWIN (MAX_PERIOD = 64 since round 6: at most 33 refills outstanding whatever B)


class Env:
    """One env's ring under the engine's bookkeeping."""

    def __init__(self, B, depth, inplace, nwin):
        self.B, self.D, self.inplace = B, depth, inplace
        self.level = [None] * depth          # sequence number of the level a slot holds (None: empty / freed)
        self.busy = [False] * depth          # a refill for this slot has been launched and has not landed yet
        fill = depth - 1 if inplace else depth      # bbai_seed: pending = depth - inplace levels from slot 0
        for s in range(fill):
            self.level[s] = s
        self.gen_seq = fill                  # next sequence number the generator produces for this env
        self.next = 0                        # hot.slot: the slot that holds the env's next level
        self.played = 0                      # levels consumed so far == sequence number of the next level the env must get
        self.pending = [0] * nwin
        self.first_slot = [0] * nwin

    def live(self):
        return (self.next - 1) % self.D     # in-place: the slot the current episode lives in (bbai_engine.hip live_slot)

    def finish(self, wb):
        s = self.next
        assert not self.busy[s], "the env moved to a slot whose refill has not landed"
        assert self.level[s] == self.played, "the env did not get the next level of its sequence"
        if self.inplace:
            freed = self.live()              # the slot the finished episode lived in (the empty slot D - 1 before the first reset)
            self.level[freed] = None         # (slot s stays occupied: it IS the live record now)
        else:
            freed = s                        # the level was copied out: its slot is free at once
            self.level[s] = None
        if self.pending[wb] == 0:
            self.first_slot[wb] = freed
        else:
            assert (self.first_slot[wb] + self.pending[wb]) % self.D == freed, "freed slots of a window are not consecutive"
        self.pending[wb] += 1
        self.next = (s + 1) % self.D
        self.played += 1
        return self.pending[wb]

    def launch_refill(self, wb):
        jobs = []
        for k in range(self.pending[wb]):
            slot = (self.first_slot[wb] + k) % self.D
            assert not self.busy[slot], "two refills in flight for one slot"
            assert self.level[slot] is None, "the refill names a slot that is not free"
            self.busy[slot] = True
            jobs.append((slot, self.gen_seq))
            self.gen_seq += 1
        return jobs

    def land(self, jobs, wb):
        for slot, seq in jobs:
            assert self.busy[slot]
            assert self.level[slot] is None, "a refill landed on a slot that still holds a level"
            if self.inplace:
                assert slot != self.live() or self.played == 0, "a refill landed on the live slot"
            self.level[slot], self.busy[slot] = seq, False
        self.pending[wb] = 0                 # k_pregen clears the env's byte when it is done with it


class Batch:
    """Several envs under ONE window clock, the gate of k_gate and refills that land in order."""

    def __init__(self, B, depth, inplace, n_envs, nwin=NWIN, admit=None, rng=None, land_prob=0.0):
        self.B, self.nwin = B, nwin
        self.envs = [Env(B, depth, inplace, nwin) for _ in range(n_envs)]
        self.M = [0] * nwin                  # meta[0] of every window buffer
        self.in_flight = {}                  # window -> per-env job lists
        self.refilled = 0                    # flow[FLOW_REFILLED]
        self.tick = 0
        self.admit = B if admit is None else admit      # the gate's bound on the sum of M
        self.rng, self.land_prob = rng, land_prob
        self.max_open = 0

    def land_one(self):
        w = self.refilled
        jobs = self.in_flight.pop(w)
        for e, j in zip(self.envs, jobs):
            e.land(j, w % self.nwin)
        self.refilled += 1

    def gate(self, x):
        while True:
            open_ = x - self.refilled
            if open_ < self.nwin and sum(max(1, self.M[v % self.nwin]) for v in range(self.refilled, x)) <= self.admit:
                break
            assert self.refilled in self.in_flight, "the gate waits for a refill that was never launched"
            self.land_one()                  # the LATEST moment the gate allows
        self.max_open = max(self.max_open, x - self.refilled)
        wb = x % self.nwin
        for e in self.envs:
            assert e.pending[wb] == 0, "a window started on a buffer whose refill has not landed"
        self.M[wb] = 0

    def step(self, finished):
        """finished[i]: env i finishes on this tick"""
        w = self.tick // self.B
        if self.tick % self.B == 0:
            self.gate(w)
        wb = w % self.nwin
        for e, f in zip(self.envs, finished):
            if f:
                p = e.finish(wb)
                if p > 1:
                    self.M[wb] = max(self.M[wb], p)      # the atomicMax of the consume paths
        if self.tick % self.B == self.B - 1:
            self.in_flight[w] = [e.launch_refill(wb) for e in self.envs]
        self.tick += 1
        # refills may also land on their own, at any time, in order
        while self.rng is not None and self.refilled in self.in_flight and self.rng.rand() < self.land_prob:
            self.land_one()


def _depth(B, inplace):
    return 2 * B + (1 if inplace else 0)


@pytest.mark.parametrize("inplace", [False, True])
@pytest.mark.parametrize("B", [1, 2, 3, 4, 8, 16, 32])
def test_ring_depth_suffices_when_every_tick_finishes(B, inplace):
    """Worst case: an env finishes on EVERY consume-tick (a reset command per step), for many windows; refills as late as allowed."""
    b = Batch(B, _depth(B, inplace), inplace, 2)
    T = 40 * B + 7
    for t in range(T):
        b.step([True, t % 3 == 0])
    assert b.envs[0].played == T
    assert b.max_open <= 1 + (B == 1)          # M = B per window: the gate degenerates to "the refill before the last has landed"


@pytest.mark.parametrize("inplace", [False, True])
@pytest.mark.parametrize("B", [2, 4, 8, 32])
def test_rare_finishes_let_the_stream_run_far_ahead_of_the_refills(B, inplace):
    """Every env finishes at most once per window (M = 1): up to B windows stay open -- a reset storm's refill runs under the next
    B windows instead of stopping the step stream -- and no env ever meets an unfilled slot."""
    n = 5
    b = Batch(B, _depth(B, inplace), inplace, n)
    rng = np.random.RandomState(B)
    T = 60 * B
    for t in range(T):
        pos = t % B
        b.step([pos == (i * 7 + t // B) % B and rng.rand() < 0.9 for i in range(n)])      # at most one finish per env and window
    assert b.max_open == min(B, NWIN - 1)
    assert all(e.played > 30 for e in b.envs)


@pytest.mark.parametrize("inplace", [False, True])
@pytest.mark.parametrize("B", [1, 2, 4, 8, 32])
def test_random_finish_patterns_and_random_landing_times(B, inplace):
    rng = np.random.RandomState(1000 * B + inplace)
    for trial in range(40):
        n = 4
        T = 30 * B + 11
        ps = rng.choice([0.0, 0.02, 0.3, 0.7, 0.97, 1.0], size=n)
        fin = rng.rand(T, n) < ps
        burst = rng.randint(0, 5 * B + 1)
        fin[burst:burst + 2 * B + 1, rng.randint(n)] = True          # a storm in the middle, longer than two windows
        b = Batch(B, _depth(B, inplace), inplace, n, rng=rng, land_prob=rng.choice([0.0, 0.05, 0.5]))
        for t in range(T):
            b.step(list(fin[t]))
        assert [e.played for e in b.envs] == [int(c) for c in fin.sum(axis=0)]


@pytest.mark.parametrize("inplace", [False, True])
@pytest.mark.parametrize("B", [1, 2, 4, 8, 32])
def test_one_slot_less_is_not_enough(B, inplace):
    """The depths are tight: one slot shallower, the worst case runs into a slot whose refill has not landed (or was never free)."""
    depth = _depth(B, inplace) - 1
    if depth < 2:
        pytest.skip("no ring left")
    with pytest.raises(AssertionError):
        b = Batch(B, depth, inplace, 1)
        for t in range(12 * B + 5):
            b.step([True])


@pytest.mark.parametrize("inplace", [False, True])
@pytest.mark.parametrize("B", [2, 4, 8, 32])
def test_a_gate_that_admits_one_more_is_not_safe(B, inplace):
    """The gate's bound is tight: admitting a sum of M of B + 1 lets an env that then finishes on every tick of the window run out of
    ready levels."""
    b = Batch(B, _depth(B, inplace), inplace, 1, admit=B + 1)
    with pytest.raises(AssertionError):
        # one finish per window for B + 1 windows (all left open by the lax gate), then a window of B finishes
        for w in range(B + 1):
            for pos in range(B):
                b.step([pos == 0])
        for pos in range(B):
            b.step([True])


@pytest.mark.parametrize("B", [4, 32])
def test_buffer_reuse_needs_the_nwin_condition(B):
    """With few window buffers condition (a) is what holds the stream: windows never start on a buffer still in use."""
    b = Batch(B, _depth(B, False), False, 3, nwin=3)
    for t in range(30 * B):
        b.step([t % B == 0, t % (2 * B) == 1, False])
    assert b.max_open <= 2


def test_inplace_first_reset_is_an_ordinary_move():
    """Before the first reset the live slot is the empty slot D - 1 (hot.slot = 0): the first consume-tick frees it like any other
    finished episode's slot, and the refill puts the ring's next level there."""
    B = 4
    b = Batch(B, 2 * B + 1, True, 1)
    e = b.envs[0]
    assert e.live() == 2 * B and e.level[e.live()] is None
    b.step([True])                                # reset(): the env moves to slot 0
    assert e.live() == 0 and e.first_slot[0] == 2 * B and e.pending[0] == 1
    for t in range(1, 3 * B):
        b.step([False])
    while b.refilled in b.in_flight:
        b.land_one()
    assert e.level[2 * B] == 2 * B                # the level after the 2B pre-generated ones
