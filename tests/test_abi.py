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


def test_header_is_plain_c():
    """include/bbai.h must be consumable from C (the drop-in boundary is a C ABI, not a C++ one)."""
    import subprocess
    subprocess.check_call(["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-pedantic",
                           os.path.join(ROOT, "include", "bbai.h")])


@pytest.mark.gpu
def test_c_abi_demo_runs_without_python_or_torch(gpu, tmp_path):
    """examples/c_abi_demo.c: a plain-C host that drives the engine through include/bbai.h only.  Its digest must
    match the same rollout driven from Python (same seeds, same LCG action stream)."""
    import subprocess
    import numpy as np
    import torch
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call(["gcc", "-std=c99", "-O2", os.path.join(ROOT, "examples", "c_abi_demo.c"),
                           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           "-L" + os.path.join(ROOT, "babyai_amd"), "-lbbai_hip", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(ROOT, "babyai_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.check_output([exe]).decode()
    assert "generator_failures=0" in out
    from babyai_amd.engine import BatchedBabyAIEnv
    n = 4096
    env = BatchedBabyAIEnv("BabyAI-GoToLocal-v0", n, device=gpu, seeds=1000)
    env.reset()
    lcg = 12345
    for t in range(200):
        a = np.empty(n, np.uint8)
        for i in range(n):
            lcg = (lcg * 1664525 + 1013904223) & 0xFFFFFFFF
            a[i] = (lcg >> 24) % 7
        env.step(torch.as_tensor(a, device=gpu))
    solved = episodes = 0
    for t in range(100):                       # second phase of the demo: the device expert drives
        _, r, d, _ = env.step(env.bot_actions(None))
        solved += int((r > 0).sum())
        episodes += int(d.sum())
    assert ("expert: episodes=%d solved=%d gave_up=0 capacity=0" % (episodes, solved)) in out, out
    # unsolved = episodes that were already close to max_steps when the expert took over from the random policy
    assert episodes - solved <= n // 8 and solved > 10 * n, out
    torch.cuda.synchronize()
    digest = 1469598103934665603
    for b in env.image.cpu().numpy().reshape(-1):
        digest = ((digest ^ int(b)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert ("obs_digest=%016x" % digest) in out, out
    env.close()


def test_product_never_touches_the_oracle_or_the_reference():
    """oracle/ is test infrastructure: nothing under babyai_amd/ may import it or read /root/reference, and bench.py may
    only reach it inside the cpu_baseline leg."""
    import re
    pkg = os.path.join(ROOT, "babyai_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f
                assert not re.search(r"""["']/root/reference""", text), f          # (citations in comments are fine)
                if f.endswith(".py"):
                    assert "hostsim" not in text, f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # bench.py reaches the oracle only as the checker / the reported baseline: after the timed region, cpu_baseline module only
    uses = [m.start() for m in re.finditer(r"(from|import) oracle", bench)]
    assert len(uses) == 1 and bench[uses[0]:uses[0] + 40].startswith("from oracle import cpu_baseline")
    # ... and the module is never CALLED beside a timed block: the parity replays come after the last one; the CPU baselines run in ONE
    # function, on a thread that is started before the GPU runtime exists and JOINED before the first warm-up step (after_seed)
    last_block = bench.rindex("shard.timed_blocks(")
    pre = bench.index("def _baselines_before_the_gpu_work():")
    pre_end = bench.index('baselines["thread"].start()')
    for m in re.finditer(r"cpu_baseline\.(parity_replay|run)\(", bench):
        assert m.start() > last_block or pre < m.start() < pre_end, bench[m.start() - 80:m.start() + 40]
    join = bench.index('baselines["thread"].join()')
    assert bench.index("def after_seed():") < join < bench.index("m = measure(ctx, level, pixel"), "the CPU legs must be complete before the GPU is stepped"
