"""The look-ahead ring's window protocol as a model (host logic, no GPU): why D = 2B slots are enough for the classic state layout and
D = 2B + 1 for the in-place one (babyai_amd/csrc/bbai_engine.hip: window_begin / window_end, consume_env, advance_finish, k_pregen;
DESIGN.md section 5).

The engine generates every env's levels ahead of need into a ring of D slots.  Consume-ticks (one reset() or one auto-resetting step)
are grouped into windows of B ticks; at the end of window w ONE refill launch regenerates the slots the window freed, and the step
stream only waits for it at the start of window w + 2.  The model below lets the refill land at the LATEST moment the engine allows --
exactly at the start of window w + 2 -- and lets the envs finish as often as they like (up to: every env on every tick), and checks,
for both layouts,

  * an env always finds the next level of ITS sequence in the slot it moves to, and that slot's refill has landed;
  * a refill never writes a slot that still holds a level the env has not played (classic) / the env's live level (in-place);
  * the in-place layout's first reset is an ordinary move out of the empty slot D - 1.

and that a ring ONE slot shallower breaks exactly these properties under the worst-case finish pattern (so the depths are tight).
"""
import numpy as np
import pytest


class Ring:
    """One env's ring under the engine's bookkeeping: `pending` / `first_slot` per window buffer (three of them), refills landing two
    windows later."""

    def __init__(self, B, depth, inplace):
        self.B, self.D, self.inplace = B, depth, inplace
        self.level = [None] * depth          # sequence number of the level a slot holds (None: empty / freed)
        self.busy = [False] * depth          # a refill for this slot has been launched and has not landed yet
        fill = depth - 1 if inplace else depth      # bbai_seed: pending = depth - inplace levels from slot 0
        for s in range(fill):
            self.level[s] = s
        self.gen_seq = fill                  # next sequence number the generator produces for this env
        self.next = 0                        # hot.slot: the slot that holds the env's next level
        self.played = 0                      # levels consumed so far == sequence number of the next level the env must get
        self.pending = [0, 0, 0]
        self.first_slot = [0, 0, 0]
        self.in_flight = {}                  # window -> list of (slot, seq) written when the refill of that window lands
        self.tick = 0

    def live(self):
        return (self.next - 1) % self.D     # in-place: the slot the current episode lives in (bbai_engine.hip live_slot)

    def begin_tick(self):
        w = self.tick // self.B
        if self.tick % self.B == 0:
            # window_begin: the stream waits for the refill launched at the end of window w - 2 (latest landing time)
            for slot, seq in self.in_flight.pop(w - 2, []):
                assert self.busy[slot]
                assert self.level[slot] is None, "a refill landed on a slot that still holds a level"
                if self.inplace:
                    assert slot != self.live() or self.played == 0, "a refill landed on the live slot"
                self.level[slot], self.busy[slot] = seq, False
        return w

    def end_tick(self, w):
        if self.tick % self.B == self.B - 1:
            # window_end: ONE refill launch for what the window freed -- `pending` consecutive slots from `first_slot`
            wb = w % 3
            jobs = []
            for k in range(self.pending[wb]):
                slot = (self.first_slot[wb] + k) % self.D
                assert not self.busy[slot], "two refills in flight for one slot"
                assert self.level[slot] is None, "the refill list names a slot that is not free"
                self.busy[slot] = True
                jobs.append((slot, self.gen_seq))
                self.gen_seq += 1
            self.in_flight[w] = jobs
            self.pending[wb] = 0
        self.tick += 1

    def step(self, finished):
        w = self.begin_tick()
        if finished:
            wb = w % 3
            s = self.next
            assert not self.busy[s], "the env moved to a slot whose refill has not landed"
            assert self.level[s] == self.played, "the env did not get the next level of its sequence"
            if self.inplace:
                freed = self.live()          # the slot the finished episode lived in (the empty slot D - 1 before the first reset)
                self.level[freed] = None
                # (slot s stays occupied: it IS the live record now)
            else:
                freed = s                    # the level was copied out: its slot is free at once
                self.level[s] = None
            if self.pending[wb] == 0:
                self.first_slot[wb] = freed
            else:
                assert (self.first_slot[wb] + self.pending[wb]) % self.D == freed, "freed slots of a window are not consecutive"
            self.pending[wb] += 1
            self.next = (s + 1) % self.D
            self.played += 1
        self.end_tick(w)


def _run(B, depth, inplace, pattern, ticks):
    r = Ring(B, depth, inplace)
    for t in range(ticks):
        r.step(pattern(t))
    return r


@pytest.mark.parametrize("inplace", [False, True])
@pytest.mark.parametrize("B", [1, 2, 3, 4, 8, 16, 32])
def test_ring_depth_suffices_when_every_tick_finishes(B, inplace):
    """Worst case: the env finishes on EVERY consume-tick (a reset command per step), for many windows."""
    r = _run(B, 2 * B + (1 if inplace else 0), inplace, lambda t: True, 40 * B + 7)
    assert r.played == 40 * B + 7


@pytest.mark.parametrize("inplace", [False, True])
@pytest.mark.parametrize("B", [1, 2, 4, 8, 32])
def test_ring_depth_suffices_for_random_finish_patterns(B, inplace):
    rng = np.random.RandomState(1000 * B + inplace)
    for trial in range(60):
        p = rng.choice([0.02, 0.3, 0.7, 0.97])
        burst = rng.randint(0, 5 * B + 1)
        fin = (rng.rand(30 * B + 11) < p)
        fin[burst:burst + 2 * B + 1] = True          # a storm in the middle, longer than two windows
        r = _run(B, 2 * B + (1 if inplace else 0), inplace, lambda t: bool(fin[t]), len(fin))
        assert r.played == int(fin.sum())


@pytest.mark.parametrize("inplace", [False, True])
@pytest.mark.parametrize("B", [1, 2, 4, 8, 32])
def test_one_slot_less_is_not_enough(B, inplace):
    """The depths are tight: one slot shallower, the worst case runs into a slot whose refill has not landed (or was never free)."""
    depth = 2 * B + (1 if inplace else 0) - 1
    if depth < 2:
        pytest.skip("no ring left")
    with pytest.raises(AssertionError):
        _run(B, depth, inplace, lambda t: True, 12 * B + 5)


def test_inplace_first_reset_is_an_ordinary_move():
    """Before the first reset the live slot is the empty slot D - 1 (hot.slot = 0): the first consume-tick frees it like any other
    finished episode's slot, and the refill puts the ring's next level there."""
    B = 4
    r = Ring(B, 2 * B + 1, True)
    assert r.live() == 2 * B and r.level[r.live()] is None
    r.step(True)                                  # reset(): the env moves to slot 0
    assert r.live() == 0 and r.first_slot[0] == 2 * B and r.pending[0] == 1
    for t in range(1, 3 * B):
        r.step(False)
    assert r.level[2 * B] == 2 * B                # the level after the 2B pre-generated ones
