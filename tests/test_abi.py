"""The C-ABI library loads and exports every symbol include/bbai.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "bbai.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bbai_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from babyai_amd import engine
    assert declared_symbols() == sorted(engine.EXPORTED_SYMBOLS)


def test_library_exports_all_symbols():
    import __graft_entry__
    __graft_entry__.build()
    import torch  # noqa: F401  (same load order as the product: torch's HIP runtime first)
    lib = ctypes.CDLL(os.path.join(ROOT, "babyai_amd", "libbbai_hip.so"))
    for sym in declared_symbols():
        assert hasattr(lib, sym), sym
    lib.bbai_version.restype = ctypes.c_int
    assert lib.bbai_version() >= 100


def test_layout_python_twin_matches_native():
    from babyai_amd.levels import LEVELS, LevelCfg, make_cfg
    import __graft_entry__
    __graft_entry__.build()
    import torch  # noqa: F401
    lib = ctypes.CDLL(os.path.join(ROOT, "babyai_amd", "libbbai_hip.so"))
    lib.bbai_fill_layout.argtypes = [ctypes.c_void_p]
    for name in LEVELS:
        cfg = make_cfg(name)
        twin = LevelCfg.from_buffer_copy(cfg)
        for f in ("W", "H", "ES", "EH", "maxo", "off_I", "off_app", "off_pos", "off_prog", "rec_bytes"):
            setattr(twin, f, 0)
        assert lib.bbai_fill_layout(ctypes.byref(twin)) == 0
        assert bytes(twin) == bytes(cfg), name


def test_no_cpu_fallback():
    """Without a GPU the product must refuse to run, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from babyai_amd.engine import BatchedBabyAIEnv, EngineError
    with pytest.raises(EngineError):
        BatchedBabyAIEnv("BabyAI-GoToLocal-v0", 4)
