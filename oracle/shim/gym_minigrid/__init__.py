"""Restatement of the `gym_minigrid` surface used by mila-iqia/babyai.
TEST INFRASTRUCTURE ONLY (oracle shim) -- never imported by the product.

gym_minigrid (maximecb/gym-minigrid, pinned `gym_minigrid>=1.2.0` at
/root/reference/setup.py:14) is an un-vendored dependency that is absent from
this image, so its published algorithm is restated here.  PARITY UNPINNED
against the real package (no copy exists to diff against); the restatement is
anchored on the reference's own call sites and semantic leaks, listed in
SURVEY.md section 8c, and validated by running the reference's unmodified
`babyai.levels.test()` and the `babyai.bot.Bot` expert on top of it
(tests/test_oracle_reference.py).
"""
from . import minigrid, roomgrid, wrappers, rendering  # noqa: F401
