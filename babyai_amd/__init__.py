"""babyai_amd -- MI355X-native batched BabyAI environment engine (step / reset /
7x7 egocentric observation / instruction verifier / pixel render as HIP kernels)."""
import os as _os

# Kernel arguments in device memory instead of host memory read across PCIe at every dispatch: the small shards' steps are short kernels
# back to back on one stream, and this is worth 4-5 % of a step there (profiles/r06/kernarg_placement_ab.jsonl: GoToLocal 65 536 envs 0.0250 /
# 0.0259 -> 0.0241 / 0.0243 ms, PickupLoc 262 144 0.0615 -> 0.0588).  A HIP runtime switch, read once when the runtime initialises: it only
# takes effect when this package is imported before the process first touches the GPU, and a value the caller set is left alone.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .levels import LEVELS, make_cfg, level_name  # noqa: F401

__version__ = "0.1.0"
