"""babyai_amd -- MI355X-native batched BabyAI environment engine (step / reset /
7x7 egocentric observation / instruction verifier / pixel render as HIP kernels)."""
from .levels import LEVELS, make_cfg, level_name  # noqa: F401

__version__ = "0.1.0"
