"""No-edit integration with the reference's training and evaluation code.

`BaseAlgo.__init__` builds its vectorised env itself -- `self.env = ParallelEnv(envs)` (babyai/rl/algos/base.py:54) -- from
the list of gym envs that scripts/train_rl.py:53-60 makes, and sizes its buffers with `len(envs)` (base.py:86).  To put
the engine underneath WITHOUT touching the reference's files:

    import babyai_amd.integrate as bbai
    bbai.install()                                              # once, before the algorithm is built
    envs = bbai.make_envs(args.env, args.procs, args.seed, pixel=use_pixel)      # instead of the gym.make loop
    algo = babyai.rl.PPOAlgo(envs, acmodel, ...)                 # the reference's class, unchanged

`install()` replaces the NAME `ParallelEnv` inside `babyai.rl.algos.base` (the only place the reference constructs one)
by a factory that hands an engine adapter through and still builds the reference's own `ParallelEnv` for a list of gym
envs; `make_envs` returns a `BatchedParallelEnv` that also answers the few list operations the reference applies to
`envs` (`len(envs)`, `envs[0].observation_space`, `envs[0].action_space`: base.py:86, train_rl.py:86-99), seeded like
train_rl.py:59 (`100 * seed + i`).  With `evaluation=True` it also swaps `babyai.evaluate.batch_evaluate` (called by
train_rl.py / evaluate.py / imitation.py for validation) for the engine's twin of the same signature
(babyai_amd.evaluate.batch_evaluate).  `uninstall()` restores both.

Everything downstream -- `ObssPreprocessor` (babyai/utils/format.py:100-119), `ACModel.forward` (babyai/model.py:217-273),
`collect_experiences` / `update_parameters` (base.py:110-260, ppo.py:33-160) -- runs as the reference wrote it;
tests/test_integration.py drives the real `PPOAlgo` + `ACModel` + `ObssPreprocessor` this way and compares with the
same run over the reference's own `ParallelEnv`, parameter for parameter.
"""
from .vec_env import BatchedParallelEnv, _VecBase

_saved = {}


def _parallel_env_factory(original):
    def ParallelEnv(envs):
        """babyai.rl.utils.ParallelEnv for a list of gym envs; an engine adapter passes through as it is."""
        if isinstance(envs, _VecBase):
            return envs
        return original(envs)
    ParallelEnv._bbai_original = original
    return ParallelEnv


def install(evaluation=True):
    """Idempotent.  Needs the reference's `babyai` package importable (it is the thing being integrated with)."""
    import babyai.rl.algos.base as base
    if not hasattr(base.ParallelEnv, "_bbai_original"):
        _saved["ParallelEnv"] = base.ParallelEnv
        base.ParallelEnv = _parallel_env_factory(base.ParallelEnv)
    if evaluation:
        import babyai.evaluate as ref_eval
        from . import evaluate as ours
        if ref_eval.batch_evaluate is not ours.batch_evaluate:
            _saved["batch_evaluate"] = ref_eval.batch_evaluate
            ref_eval.batch_evaluate = ours.batch_evaluate
    return True


def uninstall():
    import babyai.rl.algos.base as base
    if "ParallelEnv" in _saved:
        base.ParallelEnv = _saved.pop("ParallelEnv")
    if "batch_evaluate" in _saved:
        import babyai.evaluate as ref_eval
        ref_eval.batch_evaluate = _saved.pop("batch_evaluate")


def make_envs(env_name, procs, seed, pixel=False, device="cuda:0"):
    """The env list of scripts/train_rl.py:53-60 as ONE engine batch: env i seeded with 100 * seed + i."""
    return BatchedParallelEnv(env_name, procs, device=device, pixel=pixel, seeds=[100 * seed + i for i in range(procs)])
