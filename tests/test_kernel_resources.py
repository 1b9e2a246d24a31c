"""The register / spill / scratch / LDS / occupancy figures DESIGN.md quotes are the COMPILER'S for the library that is built
(babyai_amd/kernel_resources.json, written by __graft_entry__.build() from -Rpass-analysis=kernel-resource-usage), not a memory of
an earlier build: the generated table in DESIGN.md must equal what tools/kernel_resources.py renders from that file, and the file
must belong to the current kernel sources."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _load():
    import __graft_entry__ as g
    g.build()                                   # (a no-op when the library is current)
    with open(g.RESOURCES) as f:
        return g, json.load(f)


def test_resource_report_belongs_to_the_current_sources():
    g, rep = _load()
    assert rep["built_from"] == g._digest(g._csrc(), g.hip_command(g.HIP_LIB)), "kernel_resources.json is from other sources: rebuild"
    k = rep["kernels"]
    for name in ("k_step<true, 1, false>", "k_step<false, 3, true>", "k_step<false, 3, false>", "k_pregen<1, 32, false>", "k_pregen<0, 32, true>", "k_render_q<8, 1024, 1, 1>", "k_compact", "k_gate"):
        assert name in k, name
    # what the design relies on: the step kernels keep four waves per SIMD and touch no scratch memory; the render one block per CU
    for name, r in k.items():
        if name.startswith("k_step<"):
            assert r["occupancy_waves_per_simd"] >= 4 and r["scratch_bytes_per_lane"] == 0 and r["vgpr_spills"] == 0, (name, r)
    assert k["k_render_q<8, 1024, 1, 1>"]["vgprs"] <= 64


def test_design_quotes_the_compilers_figures():
    import kernel_resources
    g, rep = _load()
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    a, b = text.index(kernel_resources.BEGIN), text.index(kernel_resources.END) + len(kernel_resources.END)
    assert text[a:b] == kernel_resources.design_block(rep["kernels"]), "DESIGN.md's kernel table is stale: python tools/kernel_resources.py --design"
