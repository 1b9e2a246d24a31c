"""Stand-alone CPU oracle for the BabyAI hot path.  TEST INFRASTRUCTURE ONLY.

A plain-Python restatement of the reference's level layer, written so that it can travel to
the GPU box (where /root/reference does not exist) and serve as the checker in tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg.  It runs on the restated
gym_minigrid core in oracle/shim (see that package's docstring: the real gym_minigrid is an
absent third-party dependency => PARITY UNPINNED against it).

PINNED against the reference itself: tests/test_oracle_golden.py replays the committed
traces in tests/golden/ (recorded by tools/gen_golden.py from the unmodified
/root/reference/babyai running on the same shim) and, in the build container,
tests/test_oracle_reference.py steps this oracle side by side with the reference.

What follows which reference code (all under /root/reference/babyai/levels/):
  Desc.match / Desc.surface        verifier.py:96-161, :64-94   (ObjDesc)
  Clause.verify                    verifier.py:257-274 (open), :296-303 (go to),
                                   :330-350 (pick up), :393-417 (put next), objs_next :379-391
  Combo.verify                     verifier.py:449-471 (before), :490-512 (after), :536-550 (and)
  OracleLevel.reset / step         levelgen.py:35-47, :49-66, refresh-on-drop :68-75
  OracleLevel._gen_grid            levelgen.py:77-102 (rejection loop)
  OracleLevel.validate             levelgen.py:104-155
  OracleLevel.all_reachable        levelgen.py:201-253
  LevelGenOracle.gen_mission etc.  levelgen.py:293-460
  GoToOracle.gen_mission           iclr19_levels.py:40-63, 75-124, 224-257
  BonusOracle.gen_*                bonus_levels.py (line ranges in each method's docstring)
  FixedLayoutOracle.lay_*          test_levels.py (line ranges in each method's docstring)
PutNext / Before / After `strict` (no registered level sets them) are restated and checked against the reference's verifier
classes in tests/test_strict_modes.py.  BABYAI_DONE_ACTIONS (verifier.py:17,216-230,543-545: an action instruction only
succeeds on a `done` action taken right after the step that completed it) is restated behind `DONE_ACTIONS` below and pinned
by tests/golden/done_actions/ (tools/gen_golden_done.py: the reference imported with the variable set).
"""
import os
import sys

_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim")
if _SHIM not in sys.path:
    sys.path.insert(0, _SHIM)

from gym_minigrid.minigrid import COLOR_NAMES, DIR_TO_VEC, Ball, Box, Key  # noqa: E402
from gym_minigrid.roomgrid import RoomGrid  # noqa: E402

# verifier.py:17 `use_done_actions = os.environ.get('BABYAI_DONE_ACTIONS', False)`: any non-empty value, read at import
DONE_ACTIONS = bool(os.environ.get('BABYAI_DONE_ACTIONS', False))

TYPES_ALL = ['box', 'ball', 'key', 'door']
TYPES_MOVABLE = ['box', 'ball', 'key']
LOCS = ['left', 'right', 'front', 'behind']


class Reject(Exception):
    """Rejection-sampling signal (reference: RejectSampling)."""


def _manhattan1(p, q):
    return abs(int(p[0]) - int(q[0])) + abs(int(p[1]) - int(q[1])) == 1


class Desc(object):
    """A set of objects picked out by (type, colour, location-relative-to-start-pose)."""

    def __init__(self, type, color=None, loc=None):
        self.type, self.color, self.loc = type, color, loc
        self.objs, self.poss = [], []

    def match(self, env, use_loc=True):
        if use_loc:
            self.objs = []
        self.poss = []
        ax, ay = env.agent_pos
        room = env.room_from_pos(ax, ay)
        f = DIR_TO_VEC[env.agent_dir]
        r = (-f[1], f[0])
        grid = env.grid
        for x in range(grid.width):
            for y in range(grid.height):
                c = grid.get(x, y)
                if c is None:
                    continue
                if not use_loc and not any(c is o for o in self.objs):
                    continue
                if self.type is not None and c.type != self.type:
                    continue
                if self.color is not None and c.color != self.color:
                    continue
                if use_loc and self.loc is not None:
                    if not room.pos_inside(x, y):
                        continue
                    vx, vy = x - ax, y - ay
                    side = vx * r[0] + vy * r[1]
                    ahead = vx * f[0] + vy * f[1]
                    ok = {'left': side < 0, 'right': side > 0, 'front': ahead > 0, 'behind': ahead < 0}[self.loc]
                    if not ok:
                        continue
                if use_loc:
                    self.objs.append(c)
                self.poss.append((x, y))
        return self.objs, self.poss

    def surface(self, env):
        self.match(env)
        assert len(self.objs) > 0
        words = self.type if self.type else 'object'
        if self.color:
            words = self.color + ' ' + words
        if self.loc == 'front':
            words += ' in front of you'
        elif self.loc == 'behind':
            words += ' behind you'
        elif self.loc:
            words += ' on your ' + self.loc
        return ('a ' if len(self.objs) > 1 else 'the ') + words


class Clause(object):
    """One action instruction: kind in goto / pickup / open / putnext."""
    VERB = {'goto': 'go to ', 'pickup': 'pick up ', 'open': 'open ', 'putnext': 'put '}

    def __init__(self, kind, d1, d2=None, strict=False):
        self.kind, self.d1, self.d2, self.strict = kind, d1, d2, strict
        self.held_before = None

    def descs(self):
        return [self.d1] if self.d2 is None else [self.d1, self.d2]

    def navs(self):
        return 2 if self.kind == 'putnext' else 1

    def surface(self, env):
        s = self.VERB[self.kind] + self.d1.surface(env)
        if self.kind == 'putnext':
            s += ' next to ' + self.d2.surface(env)
        return s

    def start(self, env):
        self.env = env
        self.held_before = None
        self.last_match = False       # ActionInstr.lastStepMatch (verifier.py:213-214)
        for d in self.descs():
            d.match(env)

    def refresh(self):
        for d in self.descs():
            d.match(self.env, use_loc=False)

    def already_adjacent(self):
        return any(_manhattan1(a.cur_pos, p) for a in self.d1.objs for p in self.d2.poss)

    def verify(self, action):
        """ActionInstr.verify (verifier.py:216-230).  With done actions: a `done` succeeds iff the PREVIOUS evaluated action
        completed the instruction and fails otherwise; any other action only records whether it did (and returns None, as
        the reference does -- callers treat that like 'continue')."""
        if not DONE_ACTIONS:
            return self.verify_action(action)
        if action == self.env.actions.done:
            return 'success' if self.last_match else 'failure'
        self.last_match = self.verify_action(action) == 'success'
        return None

    def verify_action(self, action):
        env = self.env
        A = env.actions
        if self.kind == 'goto':
            fx, fy = env.front_pos
            return 'success' if any(p[0] == fx and p[1] == fy for p in self.d1.poss) else 'continue'
        if self.kind == 'open':
            if action != A.toggle:
                return 'continue'
            cell = env.grid.get(*env.front_pos)
            if cell is not None and any(cell is d for d in self.d1.objs) and cell.is_open:
                return 'success'
            if self.strict and cell is not None and cell.type == 'door':
                return 'failure'          # verifier.py:270-272
            return 'continue'
        before, self.held_before = self.held_before, env.carrying
        if self.kind == 'pickup':
            if action != A.pickup:
                return 'continue'
            if before is None and any(env.carrying is o for o in self.d1.objs):
                return 'success'
            if self.strict and env.carrying:
                return 'failure'          # verifier.py:343-346
            return 'continue'
        # putnext
        if self.strict and action == A.pickup and env.carrying:
            return 'failure'              # verifier.py:398-401
        if action != A.drop:
            return 'continue'
        for o in self.d1.objs:
            if before is o and any(_manhattan1(o.cur_pos, p) for p in self.d2.poss):
                return 'success'
        return 'continue'


class Combo(object):
    """Two sub-instructions joined by 'and' / 'before' / 'after'."""
    JOIN = {'and': ' and ', 'before': ', then ', 'after': ' after you '}

    def __init__(self, how, a, b, strict=False):
        self.how, self.a, self.b, self.strict = how, a, b, strict

    def navs(self):
        return self.a.navs() + self.b.navs()

    def surface(self, env):
        return self.a.surface(env) + self.JOIN[self.how] + self.b.surface(env)

    def start(self, env):
        self.a.start(env)
        self.b.start(env)
        self.sa = self.sb = False

    def refresh(self):
        self.a.refresh()
        self.b.refresh()

    def leaves(self):
        out = []
        for s in (self.a, self.b):
            out += s.leaves() if isinstance(s, Combo) else [s]
        return out

    def verify(self, action):
        if self.how == 'and':
            if self.sa != 'success':
                self.sa = self.a.verify(action)
            if self.sb != 'success':
                self.sb = self.b.verify(action)
            # verifier.py:543-545 -- an IDENTITY test against the enum member: true only for callers that pass
            # `env.actions.done` itself, never for the ints / numpy ints every vectorised caller passes (penv.py:8)
            if DONE_ACTIONS and action is self.a.env.actions.done:
                if self.sa == 'failure' and self.sb == 'failure':
                    return 'failure'
            return 'success' if self.sa == 'success' and self.sb == 'success' else 'continue'
        first, second = (self.a, self.b) if self.how == 'before' else (self.b, self.a)
        if not getattr(self, '_first_done', False):
            st = first.verify(action)
            if st == 'failure':
                return 'failure'
            if st != 'success':
                # strict: completing the second part first fails the mission (verifier.py:466-469,507-510); the
                # probe is a real verify() of the second part, side effects included
                if self.strict and second.verify(action) == 'success':
                    return 'failure'
                return 'continue'
            self._first_done = True
        st = second.verify(action)
        return st if st in ('success', 'failure') else 'continue'

    def start_seq(self):
        self._first_done = False


def _leaves(instr):
    return instr.leaves() if isinstance(instr, Combo) else [instr]


class OracleLevel(RoomGrid):
    """RoomGridLevel restated: reset/step glue, rejection-sampled generation, validation."""

    unblocking = None      # only LevelGen-style levels define it

    def __init__(self, room_size=8, **kw):
        super().__init__(room_size=room_size, **kw)

    def reset(self, **kw):
        obs = super().reset(**kw)
        self.instrs.start(self)
        for node in self._combos(self.instrs):
            node.start_seq()
        self.max_steps = self.instrs.navs() * self.room_size ** 2 * self.num_rows * self.num_cols
        return obs

    def _combos(self, instr):
        if isinstance(instr, Combo):
            return [instr] + self._combos(instr.a) + self._combos(instr.b)
        return []

    def step(self, action):
        obs, reward, done, info = super().step(action)
        if action == self.actions.drop:
            self.instrs.refresh()
        status = self.instrs.verify(action)
        if status == 'success':
            done = True
            reward = self._reward()
        elif status == 'failure':
            done = True
            reward = 0
        return obs, reward, done, info

    def _gen_grid(self, width, height):
        while True:
            try:
                super()._gen_grid(width, height)
                self.gen_mission()
                self.validate(self.instrs)
            except (RecursionError, Reject):
                continue
            break
        self.surface = self.instrs.surface(self)
        self.mission = self.surface

    def validate(self, instr):
        locked_colors = []
        if self.unblocking:
            for i in range(self.num_cols):
                for j in range(self.num_rows):
                    for door in self.get_room(i, j).doors:
                        if door and door.is_locked:
                            locked_colors.append(door.color)
        for leaf in _leaves(instr):
            if leaf.kind == 'putnext':
                leaf.start(self)
                if any(a is b for a in leaf.d1.objs for b in leaf.d2.objs):
                    raise Reject('same object on both sides of put-next')
                if leaf.already_adjacent():
                    raise Reject('objects already adjacent')
            if self.unblocking:
                for d in leaf.descs():
                    if d.type == 'key' and d.color in locked_colors:
                        raise Reject('key of a locked door')

    def all_reachable(self):
        """Flood from the agent through empty cells and doors; every object must be touched."""
        seen = set()
        todo = [tuple(int(v) for v in self.agent_pos)]
        W, H = self.grid.width, self.grid.height
        while todo:
            x, y = todo.pop()
            if x < 0 or y < 0 or x >= W or y >= H or (x, y) in seen:
                continue
            seen.add((x, y))
            c = self.grid.get(x, y)
            if c is not None and c.type != 'door':
                continue
            todo += [(x + 1, y), (x - 1, y), (x, y + 1), (x, y - 1)]
        for x in range(W):
            for y in range(H):
                c = self.grid.get(x, y)
                if c is not None and c.type != 'wall' and (x, y) not in seen:
                    raise Reject('unreachable object')


class GoToOracle(OracleLevel):
    """The hand-written single-instruction levels of iclr19_levels.py: GoToRedBall[Grey] :10-63, GoToObj :75-102,
    GoToLocal :105-184, PutNextLocal :187-221, GoTo :224-301, GoToImpUnlock :304-357, Pickup :360-371,
    UnblockPickup :374-391, Open :394-415, Unlock :418-474, PutNext :477-491 -- one parameterised restatement."""

    def __init__(self, room_size=8, num_rows=1, num_cols=1, num_dists=8, redball=False, connect=False,
                 check_reach=True, doors_open=False, all_unique=False, instr='goto', target=None,
                 lock=False, lock_color_excl=False, dists_per_room=False, grey_dists=False, seed=None):
        if target is None:
            target = 'redball' if redball else 'dist'
        self.p = dict(num_dists=num_dists, redball=redball, connect=connect, check_reach=check_reach,
                      doors_open=doors_open, all_unique=all_unique, instr=instr, target=target, lock=lock,
                      lock_color_excl=lock_color_excl, dists_per_room=dists_per_room, grey_dists=grey_dists)
        super().__init__(room_size=room_size, num_rows=num_rows, num_cols=num_cols, seed=seed)

    def _reachable_ok(self):
        try:
            self.all_reachable()
            return True
        except Reject:
            return False

    def _all_doors(self):
        out = []
        for i in range(self.num_cols):
            for j in range(self.num_rows):
                out += [d for d in self.get_room(i, j).doors if d]
        return out

    def _gen_locked(self):
        """Unlock / GoToImpUnlock: locked door + key first, agent last and outside the locked room."""
        p = self.p
        ci = self._rand_int(0, self.num_cols)
        cj = self._rand_int(0, self.num_rows)
        door, _ = self.add_door(ci, cj, locked=True)
        locked = self.get_room(ci, cj)
        while True:
            ki = self._rand_int(0, self.num_cols)
            kj = self._rand_int(0, self.num_rows)
            if (ki, kj) != (ci, cj):
                self.add_object(ki, kj, 'key', door.color)
                break
        if p['lock_color_excl'] and self._rand_bool():
            self.connect_all(door_colors=[c for c in COLOR_NAMES if c != door.color])
        else:
            self.connect_all()
        for i in range(self.num_cols):
            for j in range(self.num_rows):
                if (i, j) != (ci, cj):
                    self.add_distractors(i, j, num_distractors=p['num_dists'], all_unique=False)
        while True:
            self.place_agent()
            if self.room_from_pos(*self.agent_pos) is not locked:
                break
        if p['check_reach']:
            self.all_reachable()
        if p['target'] == 'locked_room_obj':
            obj, = self.add_distractors(ci, cj, num_distractors=1, all_unique=False)
            return [obj]
        return [door]

    def _gen_open(self):
        p = self.p
        self.place_agent()
        ball = None
        if p['redball']:
            ball, _ = self.add_object(0, 0, 'ball', 'red')
        if p['connect']:
            self.connect_all()
        dists = self.add_distractors(num_distractors=p['num_dists'], all_unique=p['all_unique'])
        if p['grey_dists']:
            for d in dists:
                d.color = 'grey'
        if p['check_reach'] == 2:
            if self._reachable_ok():
                raise Reject('everything reachable')
        elif p['check_reach']:
            self.all_reachable()
        if p['target'] == 'redball':
            return [ball]
        if p['target'] == 'dist':
            return [self._rand_elem(dists)]
        if p['target'] == 'two_dists':
            return self._rand_subset(dists, 2)
        return [self._rand_elem(self._all_doors())]

    def gen_mission(self):
        p = self.p
        objs = self._gen_locked() if p['lock'] else self._gen_open()
        descs = [Desc(o.type, o.color) for o in objs]
        self.instrs = Clause(p['instr'], *descs)
        if p['doors_open']:
            for door in self._all_doors():
                door.is_open = True


class BonusOracle(OracleLevel):
    """The hand-written levels of bonus_levels.py, one method per reference class (file:line in each docstring).
    `script` selects the method; `sp` carries the class parameters."""

    def __init__(self, script, room_size=8, num_rows=3, num_cols=3, num_dists=0, sp=(), seed=None):
        self.script, self.n_objs, self.sp = script, num_dists, tuple(sp)
        self.held_at_start = None
        super().__init__(room_size=room_size, num_rows=num_rows, num_cols=num_cols, seed=seed)

    def gen_mission(self):
        self.held_at_start = None
        getattr(self, 'gen_' + self.script)()

    def reset(self, **kw):
        obs = super().reset(**kw)
        if self.held_at_start is not None:        # bonus_levels.py:821-829 -- AFTER the first obs was built
            self.grid.set(*self.held_at_start.init_pos, None)
            self.carrying = self.held_at_start
        return obs

    def _doors(self, n, room=(1, 1), **kw):
        return [self.add_door(room[0], room[1], **kw)[0] for _ in range(n)]

    def gen_goto_redblue_ball(self):
        """bonus_levels.py:7-40"""
        self.place_agent()
        for d in self.add_distractors(num_distractors=self.n_objs, all_unique=False):
            if d.type == 'ball' and d.color in ('blue', 'red'):
                raise Reject('second red/blue ball')
        obj, _ = self.add_object(0, 0, 'ball', self._rand_elem(['red', 'blue']))
        self.all_reachable()
        self.instrs = Clause('goto', Desc(obj.type, obj.color))

    def gen_open_red_door(self):
        """:43-62"""
        self.add_door(0, 0, 0, 'red', locked=False)
        self.place_agent(0, 0)
        self.instrs = Clause('open', Desc('door', 'red'))

    def gen_open_door(self):
        """:65-147  sp = (select_by: 0 random / 1 color / 2 loc, debug)"""
        doors = [self.add_door(1, 1, door_idx=k, color=c, locked=False)[0]
                 for k, c in enumerate(self._rand_subset(COLOR_NAMES, 4))]
        how = self.sp[0] or 1 + ['color', 'loc'].index(self._rand_elem(['color', 'loc']))
        if how == 1:
            desc = Desc('door', doors[0].color)
        else:
            desc = Desc('door', loc=self._rand_elem(LOCS))
        self.place_agent(1, 1)
        self.instrs = Clause('open', desc, strict=bool(self.sp[1]))

    def gen_goto_door(self):
        """:150-171"""
        doors = self._doors(4)
        self.place_agent(1, 1)
        self.instrs = Clause('goto', Desc('door', self._rand_elem(doors).color))

    def gen_goto_obj_door(self):
        """:174-197"""
        self.place_agent(1, 1)
        objs = self.add_distractors(1, 1, num_distractors=8, all_unique=False)
        objs += self._doors(4)
        self.all_reachable()
        o = self._rand_elem(objs)
        self.instrs = Clause('goto', Desc(o.type, o.color))

    def gen_action_obj_door(self):
        """:200-234"""
        objs = self.add_distractors(1, 1, num_distractors=5)
        objs += self._doors(4, locked=False)
        self.place_agent(1, 1)
        o = self._rand_elem(objs)
        kind = 'goto' if self._rand_bool() else ('open' if o.type == 'door' else 'pickup')
        self.instrs = Clause(kind, Desc(o.type, o.color))

    def gen_unlock_local(self):
        """:237-264  sp = (distractors,)"""
        door, _ = self.add_door(1, 1, locked=True)
        self.add_object(1, 1, 'key', door.color)
        if self.sp[0]:
            self.add_distractors(1, 1, num_distractors=3)
        self.place_agent(1, 1)
        self.instrs = Clause('open', Desc('door'))

    def gen_key_in_box(self):
        """:267-287"""
        door, _ = self.add_door(1, 1, locked=True)
        self.place_in_room(1, 1, Box(self._rand_color(), Key(door.color)))
        self.place_agent(1, 1)
        self.instrs = Clause('open', Desc('door'))

    def gen_unlock_pickup(self):
        """:290-329  sp = (distractors,)"""
        box, _ = self.add_object(1, 0, kind='box')
        door, _ = self.add_door(0, 0, 0, locked=True)
        self.add_object(0, 0, 'key', door.color)
        if self.sp[0]:
            self.add_distractors(num_distractors=4)
        self.place_agent(0, 0)
        self.instrs = Clause('pickup', Desc(box.type, box.color))

    def gen_blocked_unlock_pickup(self):
        """:332-361"""
        self.add_object(1, 0, kind='box')
        door, pos = self.add_door(0, 0, 0, locked=True)
        self.grid.set(pos[0] - 1, pos[1], Ball(self._rand_color()))
        self.add_object(0, 0, 'key', door.color)
        self.place_agent(0, 0)
        self.instrs = Clause('pickup', Desc('box'))

    def gen_unlock_to_unlock(self):
        """:364-398"""
        ca, cb = self._rand_subset(COLOR_NAMES, 2)
        self.add_door(0, 0, door_idx=0, color=ca, locked=True)
        self.add_object(2, 0, kind='key', color=ca)
        self.add_door(1, 0, door_idx=0, color=cb, locked=True)
        self.add_object(1, 0, kind='key', color=cb)
        self.add_object(0, 0, kind='ball')
        self.place_agent(1, 0)
        self.instrs = Clause('pickup', Desc('ball'))

    def gen_pickup_dist(self):
        """:401-444  sp = (debug,)"""
        objs = self.add_distractors(num_distractors=5)
        self.place_agent(0, 0)
        o = self._rand_elem(objs)
        how = self._rand_elem(['type', 'color', 'both'])
        desc = Desc(None if how == 'color' else o.type, None if how == 'type' else o.color)
        self.instrs = Clause('pickup', desc, strict=bool(self.sp[0]))

    def gen_pickup_above(self):
        """:447-469"""
        o, _ = self.add_object(1, 0)
        self.add_door(1, 1, 3, locked=False)
        self.place_agent(1, 1)
        self.connect_all()
        self.instrs = Clause('pickup', Desc(o.type, o.color))

    def gen_open_two_doors(self):
        """:472-562  sp = (first colour index + 1 or 0, second colour index + 1 or 0, strict)"""
        names = ['red', 'green', 'blue', 'purple', 'yellow', 'grey']
        drawn = self._rand_subset(COLOR_NAMES, 2)
        c1 = names[self.sp[0] - 1] if self.sp[0] else drawn[0]
        c2 = names[self.sp[1] - 1] if self.sp[1] else drawn[1]
        d1, _ = self.add_door(1, 1, 2, color=c1, locked=False)
        d2, _ = self.add_door(1, 1, 0, color=c2, locked=False)
        self.place_agent(1, 1)
        self.instrs = Combo('before', Clause('open', Desc(d1.type, d1.color), strict=bool(self.sp[2])),
                            Clause('open', Desc(d2.type, d2.color)))

    def gen_find_obj(self):
        """:565-611"""
        i = self._rand_int(0, self.num_rows)
        j = self._rand_int(0, self.num_cols)
        o, _ = self.add_object(i, j)
        self.place_agent(1, 1)
        self.connect_all()
        self.instrs = Clause('pickup', Desc(o.type))

    def gen_key_corridor(self):
        """:614-704"""
        for j in range(1, self.num_rows):
            self.remove_wall(1, j, 3)
        row = self._rand_int(0, self.num_rows)
        door, _ = self.add_door(2, row, 2, locked=True)
        o, _ = self.add_object(2, row, kind='ball')
        self.add_object(0, self._rand_int(0, self.num_rows), 'key', door.color)
        self.place_agent(1, self.num_rows // 2)
        self.connect_all()
        self.instrs = Clause('pickup', Desc(o.type))

    def gen_one_room(self):
        """:707-763"""
        o, _ = self.add_object(0, 0, kind='ball')
        self.place_agent()
        self.instrs = Clause('pickup', Desc(o.type))

    def _two_rooms(self):
        self.place_agent(0, 0)
        left = self.add_distractors(0, 0, self.n_objs)
        right = self.add_distractors(1, 0, self.n_objs)
        self.remove_wall(0, 0, 0)
        return left, right

    def gen_put_next(self):
        """:766-904  sp = (start_carrying,)"""
        left, right = self._two_rooms()
        a = self._rand_elem(left)
        b = self._rand_elem(right)
        if self._rand_bool():
            a, b = b, a
        self.instrs = Clause('putnext', Desc(a.type, a.color), Desc(b.type, b.color))
        if self.sp[0]:
            self.held_at_start = a

    def gen_move_two_across(self):
        """:907-971"""
        left, right = self._two_rooms()
        a, d = self._rand_subset(left, 2)
        b, c = self._rand_subset(right, 2)
        self.instrs = Combo('before', Clause('putnext', Desc(a.type, a.color), Desc(b.type, b.color)),
                            Clause('putnext', Desc(c.type, c.color), Desc(d.type, d.color)))

    def gen_open_doors_order(self):
        """:974-1049  sp = (num_doors, debug)"""
        n, dbg = self.sp[0], bool(self.sp[1])
        doors = [self.add_door(1, 1, color=c, locked=False)[0] for c in self._rand_subset(COLOR_NAMES, n)]
        self.place_agent(1, 1)
        d1, d2 = self._rand_subset(doors, 2)
        mode = self._rand_int(0, 3)
        first = Clause('open', Desc(d1.type, d1.color), strict=dbg)
        if mode == 0:
            self.instrs = first
        else:
            self.instrs = Combo('before' if mode == 1 else 'after', first,
                                Clause('open', Desc(d2.type, d2.color), strict=dbg))


class FixedLayoutOracle(OracleLevel):
    """The hand-built regression layouts of test_levels.py (:13-232): exact cells, assigned agent pose."""

    def __init__(self, script, room_size=9, num_rows=1, num_cols=1, seed=None):
        self.script = script
        super().__init__(room_size=room_size, num_rows=num_rows, num_cols=num_cols, seed=seed)

    def gen_mission(self):
        getattr(self, 'lay_' + self.script)()

    def _pose(self, x, y, d):
        import numpy as np
        self.agent_pos = np.array([x, y])
        self.agent_dir = d

    def _at(self, obj, x, y):
        self.place_obj(obj, (x, y), (1, 1))
        return obj

    def lay_goto_blocked(self):
        """:13-38"""
        self.place_agent()
        self._pose(3, 3, 0)
        target = Ball('yellow')
        self.grid.set(1, 1, target)
        for i in (1, 2, 3):
            for j in (1, 2, 3):
                if (i, j) not in ((1, 1), (3, 3)):
                    self._at(Ball('red'), i, j)
        self.instrs = Clause('goto', Desc('ball', 'yellow'))

    def lay_putnext_blocked(self):
        """:41-66"""
        self.place_agent()
        self._pose(3, 3, 0)
        self._at(Ball('yellow'), 4, 4)
        self._at(Ball('blue'), 1, 1)
        self.grid.set(1, 2, Ball('red'))
        self.grid.set(2, 1, Ball('red'))
        self.instrs = Clause('putnext', Desc('ball', 'yellow'), Desc('ball', 'blue'))

    def _door_and_balls(self):
        self._pose(3, 3, 0)
        door, pos = self.add_door(0, 0, None, 'red', False)
        self._at(Ball('yellow'), 4, 4)
        self._at(Ball('blue'), pos[0], pos[1] + 1)
        return Clause('putnext', Desc('ball', 'yellow'), Desc('ball', 'blue'))

    def lay_putnext_door1(self):
        """:69-96"""
        put = self._door_and_balls()
        self.instrs = Combo('before', Clause('open', Desc('door', 'red')), put)

    def lay_putnext_door2(self):
        """:99-107"""
        self.instrs = self._door_and_balls()

    def lay_putnext_identical(self):
        """:110-138"""
        self._pose(3, 3, 0)
        self._at(Box('yellow'), 1, 1)
        self._at(Ball('blue'), 4, 4)
        self._at(Ball('red'), 2, 2)
        self.instrs = Combo('before', Clause('putnext', Desc('ball', 'blue'), Desc('box', 'yellow')),
                            Clause('putnext', Desc('box', 'yellow'), Desc('ball', None)))

    def _three_doors(self):
        _, p1 = self.add_door(0, 0, 1, 'red', False)
        self.add_door(0, 1, 0, 'red', False)
        self.add_door(1, 1, 3, 'blue', False)
        return p1

    def lay_unblocking_loop(self):
        """:141-168"""
        self._pose(15, 4, 2)
        self._three_doors()
        self._at(Box('yellow'), 9, 1)
        self._at(Ball('blue'), 5, 3)
        self._at(Ball('yellow'), 6, 2)
        self._at(Key('blue'), 15, 15)
        self.instrs = Combo('before', Clause('putnext', Desc('key', 'blue'), Desc('door', 'blue')),
                            Combo('and', Clause('goto', Desc('ball', 'yellow')), Clause('goto', Desc('box', 'yellow'))))

    def lay_putnext_close_door(self):
        """:171-201"""
        self._pose(5, 10, 2)
        px, py = self._three_doors()
        self._at(Ball('blue'), px, py - 1)
        self._at(Ball('blue'), px, py - 2)
        if px - 1 >= 1:
            self._at(Box('green'), px - 1, py - 1)
        if px + 1 < 8:
            self._at(Box('green'), px + 1, py - 1)
        self._at(Box('yellow'), 3, 15)
        self.instrs = Clause('putnext', Desc('box', 'yellow'), Desc('ball', 'blue'))

    def lay_lots_of_blockers(self):
        """:204-232"""
        self._pose(5, 5, 0)
        for x, y in ((2, 1), (2, 2), (2, 3), (3, 4), (2, 6), (1, 3)):
            self._at(Box('yellow'), x, y)
        self._at(Ball('blue'), 1, 2)
        self._at(Ball('red'), 3, 6)
        self.instrs = Clause('putnext', Desc('ball', 'red'), Desc('ball', 'blue'))


class LevelGenOracle(OracleLevel):
    """The general mission sampler (LevelGen) and all its parameterisations."""

    def __init__(self, room_size=8, num_rows=3, num_cols=3, num_dists=18, locked_room_prob=0.5,
                 locations=True, unblocking=True, implicit_unlock=True,
                 action_kinds=('goto', 'pickup', 'open', 'putnext'),
                 instr_kinds=('action', 'and', 'seq'), seed=None):
        self.num_dists = num_dists
        self.locked_room_prob = locked_room_prob
        self.locations = locations
        self.unblocking = unblocking
        self.implicit_unlock = implicit_unlock
        self.action_kinds = list(action_kinds)
        self.instr_kinds = list(instr_kinds)
        self.locked_room = None          # survives episodes, exactly like the reference attribute
        super().__init__(room_size=room_size, num_rows=num_rows, num_cols=num_cols, seed=seed)

    def gen_mission(self):
        if self._rand_float(0, 1) < self.locked_room_prob:
            self._locked_room()
        self.connect_all()
        self.add_distractors(num_distractors=self.num_dists, all_unique=False)
        while True:
            self.place_agent()
            if self.room_from_pos(*self.agent_pos) is not self.locked_room:
                break
        if not self.unblocking:
            self.all_reachable()
        self.instrs = self._instr(self.instr_kinds)

    def _locked_room(self):
        while True:
            i = self._rand_int(0, self.num_cols)
            j = self._rand_int(0, self.num_rows)
            k = self._rand_int(0, 4)
            self.locked_room = self.get_room(i, j)
            if self.locked_room.neighbors[k] is None:
                continue
            door, _ = self.add_door(i, j, k, locked=True)
            break
        while True:
            i = self._rand_int(0, self.num_cols)
            j = self._rand_int(0, self.num_rows)
            if self.get_room(i, j) is self.locked_room:
                continue
            self.add_object(i, j, 'key', door.color)
            break

    def _desc(self, types=TYPES_ALL):
        tries = 0
        while True:
            if tries > 100:
                raise RecursionError('no describable object')
            tries += 1
            color = self._rand_elem([None] + COLOR_NAMES)
            type_ = self._rand_elem(types)
            loc = None
            if self.locations and self._rand_bool():
                loc = self._rand_elem(LOCS)
            d = Desc(type_, color, loc)
            objs, poss = d.match(self)
            if not objs:
                continue
            if not self.implicit_unlock and self.locked_room:
                if all(self.locked_room.pos_inside(*p) for p in poss):
                    continue
            return d

    def _clause(self):
        kind = self._rand_elem(self.action_kinds)
        if kind == 'goto':
            return Clause('goto', self._desc())
        if kind == 'pickup':
            return Clause('pickup', self._desc(TYPES_MOVABLE))
        if kind == 'open':
            return Clause('open', self._desc(['door']))
        first = self._desc(TYPES_MOVABLE)
        return Clause('putnext', first, self._desc())

    def _instr(self, kinds):
        kind = self._rand_elem(kinds)
        if kind == 'action':
            return self._clause()
        if kind == 'and':
            a = self._instr(['action'])
            return Combo('and', a, self._instr(['action']))
        a = self._instr(['action', 'and'])
        b = self._instr(['action', 'and'])
        return Combo(self._rand_elem(['before', 'after']), a, b)


def _g(**kw):
    return ('goto', kw)


def _l(**kw):
    return ('levelgen', kw)


def _b(script, **kw):
    return ('bonus', dict(script=script, **kw))


def _t(script, **kw):
    return ('fixed', dict(script=script, **kw))


# Constructor arguments per level (iclr19_levels.py; written out independently of babyai_amd/levels.py,
# tests/test_levels_table.py checks the two tables agree).
SPECS = {
    'GoToRedBallGrey': _g(num_dists=7, redball=True, grey_dists=True),
    'GoToRedBall': _g(num_dists=7, redball=True),
    'GoToRedBallNoDists': _g(num_dists=0, redball=True),
    'GoToObj': _g(num_dists=1, all_unique=True, check_reach=False),
    'GoToObjS4': _g(room_size=4, num_dists=1, all_unique=True, check_reach=False),
    'GoToObjS6': _g(room_size=6, num_dists=1, all_unique=True, check_reach=False),
    'GoToLocal': _g(num_dists=8),
    'GoToLocalS5N2': _g(room_size=5, num_dists=2), 'GoToLocalS6N2': _g(room_size=6, num_dists=2),
    'GoToLocalS6N3': _g(room_size=6, num_dists=3), 'GoToLocalS6N4': _g(room_size=6, num_dists=4),
    'GoToLocalS7N4': _g(room_size=7, num_dists=4), 'GoToLocalS7N5': _g(room_size=7, num_dists=5),
    'GoToLocalS8N2': _g(num_dists=2), 'GoToLocalS8N3': _g(num_dists=3), 'GoToLocalS8N4': _g(num_dists=4),
    'GoToLocalS8N5': _g(num_dists=5), 'GoToLocalS8N6': _g(num_dists=6), 'GoToLocalS8N7': _g(num_dists=7),
    'GoTo': _g(num_rows=3, num_cols=3, num_dists=18, connect=True),
    'GoToOpen': _g(num_rows=3, num_cols=3, num_dists=18, connect=True, doors_open=True),
    'GoToObjMaze': _g(num_rows=3, num_cols=3, num_dists=1, connect=True),
    'GoToObjMazeOpen': _g(num_rows=3, num_cols=3, num_dists=1, connect=True, doors_open=True),
    'GoToObjMazeS4R2': _g(room_size=4, num_rows=2, num_cols=2, num_dists=1, connect=True),
    'GoToObjMazeS4': _g(room_size=4, num_rows=3, num_cols=3, num_dists=1, connect=True),
    'GoToObjMazeS5': _g(room_size=5, num_rows=3, num_cols=3, num_dists=1, connect=True),
    'GoToObjMazeS6': _g(room_size=6, num_rows=3, num_cols=3, num_dists=1, connect=True),
    'GoToObjMazeS7': _g(room_size=7, num_rows=3, num_cols=3, num_dists=1, connect=True),
    'PutNextLocal': _g(num_dists=8, all_unique=True, instr='putnext', target='two_dists'),
    'PutNextLocalS5N3': _g(room_size=5, num_dists=3, all_unique=True, instr='putnext', target='two_dists'),
    'PutNextLocalS6N4': _g(room_size=6, num_dists=4, all_unique=True, instr='putnext', target='two_dists'),
    'Pickup': _g(num_rows=3, num_cols=3, num_dists=18, connect=True, instr='pickup'),
    'UnblockPickup': _g(num_rows=3, num_cols=3, num_dists=20, connect=True, check_reach=2, instr='pickup'),
    'Open': _g(num_rows=3, num_cols=3, num_dists=18, connect=True, instr='open', target='door'),
    'PutNext': _g(num_rows=3, num_cols=3, num_dists=18, connect=True, instr='putnext', target='two_dists'),
    'Unlock': _g(num_rows=3, num_cols=3, num_dists=3, connect=True, lock=True, lock_color_excl=True,
                 dists_per_room=True, instr='open', target='locked_door'),
    'GoToImpUnlock': _g(num_rows=3, num_cols=3, num_dists=2, connect=True, lock=True, dists_per_room=True,
                        instr='goto', target='locked_room_obj'),
    'GoToRedBlueBall': _b('goto_redblue_ball', num_rows=1, num_cols=1, num_dists=7),
    'OpenRedDoor': _b('open_red_door', room_size=5, num_rows=1, num_cols=2),
    'OpenDoor': _b('open_door', sp=(0, 0)), 'OpenDoorDebug': _b('open_door', sp=(0, 1)),
    'OpenDoorColor': _b('open_door', sp=(1, 0)), 'OpenDoorLoc': _b('open_door', sp=(2, 0)),
    'GoToDoor': _b('goto_door', room_size=7), 'GoToObjDoor': _b('goto_obj_door'),
    'ActionObjDoor': _b('action_obj_door', room_size=7),
    'UnlockLocal': _b('unlock_local', sp=(0,)), 'UnlockLocalDist': _b('unlock_local', sp=(1,)),
    'KeyInBox': _b('key_in_box'),
    'UnlockPickup': _b('unlock_pickup', room_size=6, num_rows=1, num_cols=2, sp=(0,)),
    'UnlockPickupDist': _b('unlock_pickup', room_size=6, num_rows=1, num_cols=2, sp=(1,)),
    'BlockedUnlockPickup': _b('blocked_unlock_pickup', room_size=6, num_rows=1, num_cols=2),
    'UnlockToUnlock': _b('unlock_to_unlock', room_size=6, num_rows=1, num_cols=3),
    'PickupDist': _b('pickup_dist', room_size=7, num_rows=1, num_cols=1, sp=(0,)),
    'PickupDistDebug': _b('pickup_dist', room_size=7, num_rows=1, num_cols=1, sp=(1,)),
    'PickupAbove': _b('pickup_above', room_size=6),
    'OpenTwoDoors': _b('open_two_doors', room_size=6, sp=(0, 0, 0)),
    'OpenTwoDoorsDebug': _b('open_two_doors', room_size=6, sp=(0, 0, 1)),
    'OpenRedBlueDoors': _b('open_two_doors', room_size=6, sp=(1, 3, 0)),
    'OpenRedBlueDoorsDebug': _b('open_two_doors', room_size=6, sp=(1, 3, 1)),
    'FindObjS5': _b('find_obj', room_size=5), 'FindObjS6': _b('find_obj', room_size=6),
    'FindObjS7': _b('find_obj', room_size=7),
    'KeyCorridorS3R1': _b('key_corridor', room_size=3, num_rows=1),
    'KeyCorridorS3R2': _b('key_corridor', room_size=3, num_rows=2),
    'KeyCorridorS3R3': _b('key_corridor', room_size=3, num_rows=3),
    'KeyCorridorS4R3': _b('key_corridor', room_size=4, num_rows=3),
    'KeyCorridorS5R3': _b('key_corridor', room_size=5, num_rows=3),
    'KeyCorridorS6R3': _b('key_corridor', room_size=6, num_rows=3),
    '1RoomS8': _b('one_room', room_size=8, num_rows=1, num_cols=1),
    '1RoomS12': _b('one_room', room_size=12, num_rows=1, num_cols=1),
    '1RoomS16': _b('one_room', room_size=16, num_rows=1, num_cols=1),
    '1RoomS20': _b('one_room', room_size=20, num_rows=1, num_cols=1),
    'PutNextS4N1': _b('put_next', room_size=4, num_rows=1, num_cols=2, num_dists=1, sp=(0,)),
    'PutNextS5N1': _b('put_next', room_size=5, num_rows=1, num_cols=2, num_dists=1, sp=(0,)),
    'PutNextS5N2': _b('put_next', room_size=5, num_rows=1, num_cols=2, num_dists=2, sp=(0,)),
    'PutNextS6N3': _b('put_next', room_size=6, num_rows=1, num_cols=2, num_dists=3, sp=(0,)),
    'PutNextS7N4': _b('put_next', room_size=7, num_rows=1, num_cols=2, num_dists=4, sp=(0,)),
    'PutNextS5N2Carrying': _b('put_next', room_size=5, num_rows=1, num_cols=2, num_dists=2, sp=(1,)),
    'PutNextS6N3Carrying': _b('put_next', room_size=6, num_rows=1, num_cols=2, num_dists=3, sp=(1,)),
    'PutNextS7N4Carrying': _b('put_next', room_size=7, num_rows=1, num_cols=2, num_dists=4, sp=(1,)),
    'MoveTwoAcrossS5N2': _b('move_two_across', room_size=5, num_rows=1, num_cols=2, num_dists=2),
    'MoveTwoAcrossS8N9': _b('move_two_across', room_size=8, num_rows=1, num_cols=2, num_dists=9),
    'OpenDoorsOrderN2': _b('open_doors_order', room_size=6, sp=(2, 0)),
    'OpenDoorsOrderN4': _b('open_doors_order', room_size=6, sp=(4, 0)),
    'OpenDoorsOrderN2Debug': _b('open_doors_order', room_size=6, sp=(2, 1)),
    'OpenDoorsOrderN4Debug': _b('open_doors_order', room_size=6, sp=(4, 1)),
    'TestGoToBlocked': _t('goto_blocked'), 'TestPutNextToBlocked': _t('putnext_blocked'),
    'TestPutNextToCloseToDoor1': _t('putnext_door1', num_rows=2), 'TestPutNextToCloseToDoor2': _t('putnext_door2', num_rows=2),
    'TestPutNextToIdentical': _t('putnext_identical'),
    'TestUnblockingLoop': _t('unblocking_loop', num_rows=2, num_cols=2),
    'TestPutNextCloseToDoor': _t('putnext_close_door', num_rows=2, num_cols=2),
    'TestLotsOfBlockers': _t('lots_of_blockers', room_size=8),
    'PickupLoc': _l(action_kinds=('pickup',), instr_kinds=('action',), num_rows=1, num_cols=1, num_dists=8,
                    locked_room_prob=0, locations=True, unblocking=False),
    'GoToSeq': _l(action_kinds=('goto',), locked_room_prob=0, locations=False, unblocking=False),
    'GoToSeqS5R2': _l(room_size=5, num_rows=2, num_cols=2, num_dists=4, action_kinds=('goto',),
                      locked_room_prob=0, locations=False, unblocking=False),
    'Synth': _l(instr_kinds=('action',), locations=False, unblocking=True, implicit_unlock=False),
    'SynthS5R2': _l(room_size=5, num_rows=2, num_cols=2, num_dists=7, instr_kinds=('action',),
                    locations=False, unblocking=True, implicit_unlock=False),
    'SynthLoc': _l(instr_kinds=('action',), locations=True, unblocking=True, implicit_unlock=False),
    'SynthSeq': _l(locations=True, unblocking=True, implicit_unlock=False),
    'MiniBossLevel': _l(num_cols=2, num_rows=2, room_size=5, num_dists=7, locked_room_prob=0.25),
    'BossLevel': _l(),
    'BossLevelNoUnlock': _l(locked_room_prob=0, implicit_unlock=False),
}


def level_name(env_id):
    name = env_id
    if name.startswith('BabyAI-'):
        name = name[len('BabyAI-'):]
        if name.endswith('-v0'):
            name = name[:-3]
    return name


def make_env(env_id, seed=None):
    """Oracle twin of gym.make('BabyAI-<Level>-v0'); `seed` is the constructor seed."""
    fam, kw = SPECS[level_name(env_id)]
    cls = {'goto': GoToOracle, 'levelgen': LevelGenOracle, 'bonus': BonusOracle, 'fixed': FixedLayoutOracle}[fam]
    env = cls(seed=seed, **kw)
    if seed is None and fam == 'levelgen':
        # the constructor's entropy-seeded reset must not leak a stale locked_room into the seeded
        # stream (the reference has this leak: levelgen.py:284,325,384); seed() starts from None.
        env.locked_room = None
    return env
