"""`strict` verifier modes that no registered level switches on (babyai/levels/verifier.py:398-401 PutNextInstr,
:466-469 BeforeInstr, :507-510 AfterInstr): flipped on by hand after every reset in the REFERENCE's own instruction
objects, in the stand-alone oracle and in the kernel core (host build of bbai_step.hpp), then stepped side by side.
Build-container only (needs /root/reference)."""
import random

import numpy as np
import pytest

from oracle import refenv
from oracle import levels as olevels
from babyai_amd.levels import make_cfg
from hostsim_util import HostEnv

pytestmark = pytest.mark.skipif(not refenv.have_reference(), reason="/root/reference not present")

PROG_STRICT = 103          # byte offset of Prog.strict inside the 112-byte program (bbai_types.hpp)


@pytest.fixture(scope="module")
def level_dict():
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        refenv.import_reference()
        from babyai.levels import level_dict
    return level_dict


def _arm(ref, ora, host, leaf_bits, seq):
    """Switch strict on in all three after a reset."""
    from babyai.levels.verifier import SeqInstr, AndInstr
    if seq and isinstance(ref.instrs, SeqInstr) and not isinstance(ref.instrs, AndInstr):
        ref.instrs.strict = True
        ora.instrs.strict = True
        host.rec[host.cfg.off_prog + PROG_STRICT] |= 16
    if leaf_bits:
        def leaves(i):
            return leaves(i.instr_a) + leaves(i.instr_b) if isinstance(i, SeqInstr) else [i]
        for leaf in leaves(ref.instrs):
            leaf.strict = True
        for leaf in olevels._leaves(ora.instrs):
            leaf.strict = True
        host.rec[host.cfg.off_prog + PROG_STRICT] |= 15


@pytest.mark.parametrize("name,leaf_bits,seq", [
    ("PutNextLocal", True, False),           # PutNext strict: any pickup that leaves the agent holding something fails
    ("PutNextS5N2", True, False),
    ("GoToSeqS5R2", False, True),            # Before / After strict: completing the second part first fails
    ("SynthSeq", False, True),               # ... with And sides: the probe's side effects (progress bits, preCarrying)
    ("MiniBossLevel", True, True),           # everything at once
])
def test_strict_modes_three_way(level_dict, name, leaf_bits, seq):
    strict_failures = 0
    for seed in range(12):
        ref = level_dict[name]()
        if hasattr(ref, "locked_room"):
            ref.locked_room = None
        ref.seed(seed)
        ora = olevels.make_env(name)
        ora.seed(seed)
        host = HostEnv(make_cfg(name), seed)
        rng = random.Random(seed * 7 + 1)
        for ep in range(3):
            a, b = ref.reset(), ora.reset()
            img = host.reset()
            assert np.array_equal(a["image"], b["image"]) and np.array_equal(a["image"], img)
            _arm(ref, ora, host, leaf_bits, seq)
            for t in range(400):
                act = rng.choice([0, 1, 2, 2, 2, 3, 3, 4, 5, 6])
                (a, ra, da, _), (b, rb, db, _) = ref.step(act), ora.step(act)
                img, rh, dh = host.step(act)
                assert np.array_equal(a["image"], b["image"]) and np.array_equal(a["image"], img), (name, seed, ep, t)
                assert ra == rb and da == db, (name, seed, ep, t)
                assert np.float32(ra) == rh and da == dh, (name, seed, ep, t)
                if da:
                    if ra == 0 and ref.step_count < ref.max_steps:
                        strict_failures += 1
                    break
    assert strict_failures > 0, "the strict branches were never taken"
