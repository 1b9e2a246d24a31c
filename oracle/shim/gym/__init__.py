"""Minimal restatement of the `gym` (<=0.21 API generation) surface that
mila-iqia/babyai touches.  TEST INFRASTRUCTURE ONLY (oracle shim).

The real `gym` package is an un-vendored third-party dependency of the
reference (setup.py:9-15 `gym>=0.9.6`) and is absent from this image, so the
handful of names the reference uses are restated here from the published
gym 0.9-0.21 behaviour:  Env / Wrapper / ObservationWrapper, spaces.{Box,
Discrete,Dict}, envs.registration.register + make, utils.seeding.np_random.
Reference call sites: babyai/levels/levelgen.py:4,483; babyai/rl/utils/penv.py:18;
babyai/evaluate.py:58,90.
"""
from . import spaces, error
from .core import Env, Wrapper, ObservationWrapper
from . import core
from .envs.registration import make, register, registry
from . import envs, utils

__version__ = "0.21.0-shim"
