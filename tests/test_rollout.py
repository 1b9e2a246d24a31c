"""babyai_amd.rollout.DeviceRollout vs the reference's own BaseAlgo.collect_experiences
(babyai/rl/algos/base.py:131-260), run UNMODIFIED on the shim over reference envs in ParallelEnv workers.
Needs /root/reference (build container only)."""
import numpy as np
import pytest
import torch

from oracle import refenv
from rollout_util import OracleTensorEnv, ToyACModel, pad_tokens

pytestmark = pytest.mark.skipif(not refenv.have_reference(), reason="reference tree not present")

FIELDS = ("memory", "mask", "action", "value", "reward", "advantage", "returnn", "log_prob")


def _reference_algo(level, seeds, T, scale):
    refenv.import_reference()
    import gym
    from babyai.rl.algos.base import BaseAlgo
    from babyai.rl.utils import DictList

    class Algo(BaseAlgo):
        def update_parameters(self):
            pass

    def preprocess(obss, device=None):
        return DictList({"image": torch.tensor(np.array([o["image"] for o in obss]), dtype=torch.float),
                         "instr": torch.tensor(pad_tokens([o["mission"] for o in obss]))})

    envs = []
    for s in seeds:
        e = gym.make("BabyAI-%s-v0" % level)
        e.seed(int(s))
        envs.append(e)
    return Algo(envs, ToyACModel(), T, 0.99, 7e-4, 0.95, 0.01, 0.5, 0.5, 1, preprocess,
                lambda _0, _1, reward, _2: scale * reward, None)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("level,scale", [("GoToObjS4", 16.0), ("GoToLocalS5N2", 16.0), ("GoToObjS4", 20.0)])
def test_device_rollout_equals_reference_collect_experiences(level, scale):
    from babyai_amd.rollout import DeviceRollout
    seeds = [500 + i for i in range(5)]
    T = 24
    algo = _reference_algo(level, seeds, T, scale)
    roll = DeviceRollout(OracleTensorEnv(level, seeds), ToyACModel(), T, 0.99, 0.95, reward_scale=scale)
    total_done = 0
    for it in range(3):                                    # state carries across rollouts (memory, masks, log tails)
        exps_ref, log_ref = algo.collect_experiences()
        exps, log = roll.collect_experiences()
        exact = True               # rewards are shaped from the float64 reward, like the reference: exact for any scale
        for f in FIELDS:
            a, b = getattr(exps_ref, f), getattr(exps, f)
            assert a.shape == b.shape and a.dtype == b.dtype, f
            if exact:
                assert torch.equal(a, b), (level, it, f)
            else:
                assert torch.allclose(a, b, rtol=1e-6, atol=1e-6), (level, it, f)
        assert torch.equal(exps_ref.obs.image, exps.obs.image)
        assert torch.equal(exps_ref.obs.instr, exps.obs.instr)
        assert log_ref["num_frames"] == log["num_frames"] and log_ref["episodes_done"] == log["episodes_done"]
        assert log_ref["num_frames_per_episode"] == log["num_frames_per_episode"]
        for k in ("return_per_episode", "reshaped_return_per_episode"):
            assert np.allclose(log_ref[k], log[k], rtol=0 if exact else 1e-6, atol=0 if exact else 1e-6), k
        total_done += log["episodes_done"]
    assert total_done >= 5                                 # the comparison crossed auto-resets


def test_collect_experiences_copy_survives_the_next_collection():
    """`exps` of collect_experiences() alias the collector's buffers (module docstring); copy=True hands out tensors of
    the caller's own, like the reference's fresh transposed copies (base.py:207-232)."""
    from babyai_amd.rollout import DeviceRollout
    seeds = [900 + i for i in range(4)]
    roll = DeviceRollout(OracleTensorEnv("GoToObjS4", seeds), ToyACModel(), 12, 0.99, 0.95, reward_scale=20.0)
    views, _ = roll.collect_experiences()
    own, _ = roll_b = DeviceRollout(OracleTensorEnv("GoToObjS4", seeds), ToyACModel(), 12, 0.99, 0.95, reward_scale=20.0).collect_experiences(copy=True)
    for f in FIELDS:
        assert torch.equal(getattr(views, f), getattr(own, f)), f
    keep = {f: getattr(own, f).clone() for f in FIELDS}
    keep_img = own.obs.image.clone()
    assert views.value.data_ptr() == roll.values.data_ptr() and own.value.data_ptr() != roll.values.data_ptr()
    before = views.action.clone()
    roll.collect_experiences()                             # overwrites what `views` points at ...
    assert not torch.equal(views.action, before) or not torch.equal(views.value, keep["value"])
    for f in FIELDS:                                       # ... and leaves the copies alone
        assert torch.equal(getattr(own, f), keep[f]), f
    assert torch.equal(own.obs.image, keep_img)
