"""babyai_amd.integrate: the engine's adapters under the reference's REAL training stack, with no edit to the reference.

north_star: "so babyai/rl and babyai/model.py consume it unchanged".  `BaseAlgo.__init__` hard-codes `ParallelEnv(envs)`
and `len(envs)` (babyai/rl/algos/base.py:54,86); `integrate.install()` swaps the name `ParallelEnv` inside that module for
a factory that lets an adapter through.  Here the reference's own `PPOAlgo` (babyai/rl/algos/ppo.py), `ACModel`
(babyai/model.py) and `ObssPreprocessor` (babyai/utils/format.py:100-119) run two full updates
  (A) over the reference's `ParallelEnv` on the reference's gym envs (worker processes, the shim underneath), and
  (B) over `BatchedParallelEnv` -- the adapter code of babyai_amd/vec_env.py -- with the engine slot filled by an
      engine-protocol object over oracle envs (this container has no GPU; on the GPU box the slot holds BatchedBabyAIEnv,
      whose outputs the -m gpu suite pins to the same oracle),
from the same seeds and the same initial weights: every log value and every model parameter must be bit-equal.
Needs /root/reference (build container only)."""
import copy

import numpy as np
import pytest
import torch

from oracle import refenv
from rollout_util import OracleEngine

pytestmark = pytest.mark.skipif(not refenv.have_reference(), reason="reference tree not present")

LEVEL, PROCS, SEED, T = "GoToObjS4", 6, 3, 16


def _ppo(envs, acmodel, pre):
    import babyai.rl
    return babyai.rl.PPOAlgo(envs, acmodel, T, 0.99, 1e-3, 0.9, 0.999, 0.99, 0.01, 0.5, 0.5, 4, 1e-5, 0.2, 2, 32, pre,
                             lambda _0, _1, reward, _2: 20.0 * reward)


def _train(envs, acmodel, pre, updates=2):
    import babyai.utils as utils
    utils.seed(SEED)                       # random, numpy (PPO's batch permutations) and torch (dist.sample)
    algo = _ppo(envs, acmodel, pre)
    logs = [algo.update_parameters() for _ in range(updates)]
    return algo, logs


@pytest.mark.timeout(900)
def test_reference_ppo_runs_unchanged_over_the_adapter(monkeypatch, tmp_path):
    monkeypatch.setenv("BABYAI_STORAGE", str(tmp_path))
    refenv.import_reference()
    import gym
    import babyai.utils as utils
    import babyai.rl.algos.base as base
    from babyai.model import ACModel
    from babyai_amd import integrate
    from babyai_amd.vec_env import BatchedParallelEnv

    seeds = [100 * SEED + i for i in range(PROCS)]             # scripts/train_rl.py:59
    ref_envs = []
    for s in seeds:
        e = gym.make("BabyAI-%s-v0" % LEVEL)
        e.seed(s)
        ref_envs.append(e)
    pre_a = utils.ObssPreprocessor("bbai_integration_a", ref_envs[0].observation_space)
    torch.manual_seed(7)
    # (arch without "res": the reference's residual `out += x` (model.py:248) is an in-place add on a ReLU output, which the
    #  autograd of the torch in this image rejects -- a reference-vs-modern-torch matter, not an env one)
    model_a = ACModel(pre_a.obs_space, ref_envs[0].action_space, 128, 128, 128, True, "gru", True, "bow_endpool")
    model_b = copy.deepcopy(model_a)

    original = base.ParallelEnv
    integrate.install()
    try:
        assert base.ParallelEnv is not original and base.ParallelEnv._bbai_original is original
        integrate.install()                                        # idempotent
        assert base.ParallelEnv._bbai_original is original
        # (A) a list of gym envs still becomes the reference's own ParallelEnv
        algo_a, logs_a = _train(ref_envs, model_a, pre_a)
        assert type(algo_a.env) is original
        # (B) the adapter stands where the list stood: len(), [0].observation_space, and passes through the factory
        adapter = BatchedParallelEnv("BabyAI-%s-v0" % LEVEL, PROCS, seeds=seeds, engine=OracleEngine(LEVEL, PROCS))
        assert len(adapter) == PROCS and adapter[0].action_space.n == 7 and adapter[0].observation_space.spaces["image"].shape == (7, 7, 3)
        pre_b = utils.ObssPreprocessor("bbai_integration_b", adapter[0].observation_space)
        algo_b, logs_b = _train(adapter, model_b, pre_b)
        assert algo_b.env is adapter and algo_b.num_procs == PROCS
    finally:
        integrate.uninstall()
    assert base.ParallelEnv is original

    assert len(logs_a) == len(logs_b) == 2
    for la, lb in zip(logs_a, logs_b):
        assert set(la) == set(lb)
        for k in la:
            va, vb = la[k], lb[k]
            if isinstance(va, list):
                assert np.array_equal(np.asarray(va), np.asarray(vb)), k
            else:
                assert va == vb, k
    assert sum(l["episodes_done"] for l in logs_b) >= PROCS        # the run crossed auto-resets
    sa, sb = model_a.state_dict(), model_b.state_dict()
    assert set(sa) == set(sb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k                        # two PPO updates later: identical weights
    assert pre_a.vocab.vocab == pre_b.vocab.vocab                  # same missions in the same first-seen order


def test_install_swaps_batch_evaluate_and_restores_it():
    refenv.import_reference()
    import babyai.evaluate as ref_eval
    from babyai_amd import integrate, evaluate as ours
    original = ref_eval.batch_evaluate
    integrate.install(evaluation=True)
    try:
        assert ref_eval.batch_evaluate is ours.batch_evaluate
        import inspect
        ref_params = list(inspect.signature(original).parameters)
        assert list(inspect.signature(ours.batch_evaluate).parameters)[:len(ref_params)] == ref_params       # same call
    finally:
        integrate.uninstall()
    assert ref_eval.batch_evaluate is original
