"""A reference-trained model's vocabulary on the tensor fast path: the reference's own `InstructionsPreprocessor`
(babyai/utils/format.py:44-75, ids in first-seen order, saved as vocab.json) against the engine's fixed token ids pushed
through `babyai_amd.preprocess.remap_table` -- on 10^4 missions drawn from every level.  Build container only."""
import json

import numpy as np
import pytest

from oracle import refenv
from babyai_amd.levels import LEVELS, make_cfg
from babyai_amd.missions import tokenize
from babyai_amd.preprocess import remap_table
from hostsim_util import HostEnv

pytestmark = pytest.mark.skipif(not refenv.have_reference(), reason="/root/reference not present")


def test_remap_table_reproduces_the_reference_preprocessor(tmp_path, monkeypatch):
    monkeypatch.setenv("BABYAI_STORAGE", str(tmp_path))
    refenv.import_reference()
    from babyai.utils.format import InstructionsPreprocessor, get_vocab_path
    missions = []
    names = sorted(LEVELS)
    k = 0
    while len(missions) < 10000:
        h = HostEnv(make_cfg(names[k % len(names)]), 31000 + k)
        for _ in range(3):
            h.reset()
            missions.append(h.mission)
        k += 1
    # a "trained model": its vocabulary grew in first-seen order over the first 300 missions only
    trained = InstructionsPreprocessor("trained")
    trained([{"mission": m} for m in missions[:300]])
    trained.vocab.save()
    vocab = json.load(open(get_vocab_path("trained")))
    assert 5 < len(vocab) <= 32
    lut, extended = remap_table(vocab)
    assert all(extended[w] == vocab[w] for w in vocab)
    # the reference preprocessor, continuing from the saved vocabulary, on all 10^4 missions in one batch per 500
    ref = InstructionsPreprocessor("trained")
    for lo in range(0, len(missions), 500):
        batch = missions[lo:lo + 500]
        want = ref([{"mission": m} for m in batch]).numpy()
        for row, m in zip(want, batch):
            ids = [lut[t] for t in tokenize(m)]
            assert list(row[:len(ids)]) == ids and not row[len(ids):].any(), m
    # a vocabulary that has seen only a few words: known words keep their ids exactly, unseen ones get the next free
    # ids (distinct, above the loaded ones), as Vocabulary.__getitem__ would hand them out
    small = InstructionsPreprocessor("small")
    small([{"mission": m} for m in missions[:2]])
    v2 = dict(small.vocab.vocab)
    assert len(v2) < 12
    lut2, ext2 = remap_table(v2)
    assert all(ext2[w] == v2[w] for w in v2)
    new_ids = sorted(ext2[w] for w in ext2 if w not in v2)
    assert new_ids == list(range(len(v2) + 1, 33)) and len(set(lut2[1:])) == 32
    known_only = [m for m in missions if all(w in v2 for w in m.replace(",", " ").split())]
    assert known_only
    again = InstructionsPreprocessor("small")
    again.vocab.vocab = dict(v2)
    for m in known_only[:200]:
        assert list(again([{"mission": m}]).numpy()[0]) == [lut2[t] for t in tokenize(m)]
