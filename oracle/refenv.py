"""Import helper: put the oracle shim (restated gym / gym_minigrid / blosc) on
sys.path and, when available, the read-only reference tree, so that
`/root/reference/babyai` runs UNMODIFIED on top of the shim.

TEST INFRASTRUCTURE ONLY.  /root/reference exists only in the build container
(never on the GPU box): callers must check `have_reference()`.
"""
import os
import sys

SHIM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "shim")
REFERENCE_DIR = os.environ.get("BABYAI_REFERENCE", "/root/reference")


def enable_shim():
    if SHIM_DIR not in sys.path:
        sys.path.insert(0, SHIM_DIR)


def have_reference():
    return os.path.isfile(os.path.join(REFERENCE_DIR, "babyai", "levels", "levelgen.py"))


def import_reference():
    """Import the reference's babyai package on top of the shim."""
    if not have_reference():
        raise ImportError("reference tree not present at %s" % REFERENCE_DIR)
    enable_shim()
    sys.dont_write_bytecode = True      # never write __pycache__ into the reference
    os.environ.pop("BABYAI_DONE_ACTIONS", None)
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(1, REFERENCE_DIR)
    import babyai  # noqa: F401
    import babyai.levels  # noqa: F401
    return babyai
