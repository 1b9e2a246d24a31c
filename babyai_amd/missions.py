"""Mission strings from the compiled instruction program (`bbai::Prog`).

The engine never builds strings on the device: the generator stores, per leaf and
descriptor, (type, colour, location, match count) and the surface form is produced here
on demand, following the reference grammar (babyai/levels/verifier.py:64-94 ObjDesc.surface,
:248-249 open, :287-288 go to, :318-319 pick up, :366-367 put..next to, :439-440 ', then ',
:480-481 ' after you ', :526-527 ' and ').
"""
import numpy as np

PROG_BYTES = 112
TYPE_NAME = {0: "object", 4: "door", 5: "key", 6: "ball", 7: "box"}    # 0: ObjDesc(type=None)
COLOR_NAME = {0: "red", 1: "green", 2: "blue", 3: "purple", 4: "yellow", 5: "grey"}
L_GOTO, L_PICKUP, L_OPEN, L_PUTNEXT = 1, 2, 3, 4
R_ACTION, R_AND, R_BEFORE, R_AFTER = 0, 1, 2, 3

# the 32-word baby-language vocabulary, fixed ids 1..32, 0 = padding
VOCAB = ("go to pick up open put next the a object red green blue purple yellow grey box ball key "
         "door in front of you behind on your left right and then after").split()
WORD_TO_ID = {w: i + 1 for i, w in enumerate(VOCAB)}


def _desc_surface(d):
    type_, color, loc, count = int(d[0]), int(d[1]), int(d[2]), int(d[3])
    s = TYPE_NAME[type_]
    if color != 7:
        s = COLOR_NAME[color] + " " + s
    if loc == 3:
        s += " in front of you"
    elif loc == 4:
        s += " behind you"
    elif loc == 1:
        s += " on your left"
    elif loc == 2:
        s += " on your right"
    return ("a " if count > 1 else "the ") + s


def _leaf_surface(kind, descs):
    if kind == L_GOTO:
        return "go to " + _desc_surface(descs[0])
    if kind == L_PICKUP:
        return "pick up " + _desc_surface(descs[0])
    if kind == L_OPEN:
        return "open " + _desc_surface(descs[0])
    if kind == L_PUTNEXT:
        return "put " + _desc_surface(descs[0]) + " next to " + _desc_surface(descs[1])
    raise ValueError("bad leaf kind %d" % kind)


def prog_surface(prog):
    """prog: uint8[112] -> mission string."""
    prog = np.asarray(prog, dtype=np.uint8)
    desc = prog[64:96].reshape(4, 2, 4)
    kind = prog[96:100]
    root, n_a, n_b = int(prog[100]), int(prog[101]), int(prog[102])

    def side(base, n):
        parts = [_leaf_surface(int(kind[base + q]), desc[base + q]) for q in range(n)]
        return " and ".join(parts)

    if root in (R_ACTION, R_AND):
        return side(0, n_a)
    a, b = side(0, n_a), side(2, n_b)
    return a + (", then " if root == R_BEFORE else " after you ") + b


def tokenize(mission):
    """Fixed-vocabulary token ids of a mission string (same split as
    babyai/utils/format.py:64 `re.findall("([a-z]+)", mission.lower())`)."""
    import re
    return [WORD_TO_ID[w] for w in re.findall("([a-z]+)", mission.lower())]


def detokenize(ids):
    """Mission string of a row of fixed-vocabulary token ids (0 = padding): the inverse of `tokenize` on the baby language,
    whose only punctuation is the comma of BeforeInstr's ', then ' (babyai/levels/verifier.py:439-440)."""
    words = [VOCAB[int(t) - 1] for t in ids if int(t)]
    return " ".join(words).replace(" then ", ", then ")


def max_mission_tokens(cfg):
    """Upper bound on the token count of any mission of a level (so that a model can be fed a fixed instruction width,
    no host synchronisation per frame): per descriptor article + colour + type (+ up to 4 location words); per leaf the
    verb (go to / pick up / open / put ... next to); " and " inside a side, ", then " / " after you " between sides."""
    kind = int(cfg.kind)
    if kind == 2:                      # bonus scripts: hand-written missions, at most two leaves without locations
        return 28
    desc = 3 + (4 if (kind == 1 and cfg.locations) else 0)
    leaf = {L_GOTO: 2 + desc, L_PICKUP: 2 + desc, L_OPEN: 1 + desc, L_PUTNEXT: 3 + 2 * desc}
    if kind == 0:
        return leaf[int(cfg.instr)]
    kinds = [int(cfg.action_kinds[i]) + 1 for i in range(int(cfg.n_action_kinds))]      # AK_* -> L_*
    one = max(leaf[k] for k in kinds)
    instr = [int(cfg.instr_kinds[i]) for i in range(int(cfg.n_instr_kinds))]           # 0 action, 1 and, 2 seq
    if 2 in instr:
        return 4 * one + 2 + 2         # two And-pairs joined by " after you "
    if 1 in instr:
        return 2 * one + 1
    return one
