"""bench.py's host-side arithmetic, and the rule that ties the committed rocprofv3 counters to the committed kernel sources: every
workload the judged line quotes `roofline.traffic` for must have an entry in profiles/pmc_latest.json taken on EXACTLY these sources
(`csrc_sha`), or the line would print `traffic: null` -- a kernel edit after the last evidence pass fails here, not silently there."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_block_stats_are_sustained_means():
    st = bench.block_stats([1.0, 1.0, 1.0, 5.0], K=10, E=100, world=2)
    assert st["mean"] == 2.0 and st["median"] == 1.0 and st["max"] == 5.0 and st["min"] == 1.0
    assert st["value_mean"] == 10 * 100 * 2 / 2.0 and st["value_median"] == 10 * 100 * 2 / 1.0
    assert st["mean_over_median"] == 2.0 and st["max_over_median"] == 5.0 and st["p90"] == 5.0


def test_every_benched_workload_has_counters_of_the_committed_sources():
    with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
        pmc = json.load(f)
    assert pmc["csrc_sha"] == bench.csrc_sha(), "kernel sources changed since the last rocprofv3 evidence pass (tools/lease.sh ... profcfg / rejudged)"
    workloads = [("BossLevel", bench.HEADLINE_ENVS, True)] + [(c["level"], c["total"], c["pixel"]) for _, c in bench.EXTRA_CONFIGS]
    for level, envs, pixel in workloads:
        dom = "k_render" if pixel else "k_step"
        t = bench.traffic_of(level, envs, pixel, dom)
        assert t is not None and t["current"] and t["bytes"] > 0, (level, envs, pixel)
        alg = envs * (147 + 9408) if pixel else envs * 235
        assert 0.99 < t["bytes"] / alg < 3.0, (level, envs, pixel, t["bytes"] / alg)       # the render writes every byte once; k_step's gathers cost whole lines


def test_extra_configs_cover_baseline_json():
    names = [n for n, _ in bench.EXTRA_CONFIGS]
    for want in ("C2", "C3", "C4", "C4-shard", "C5-encoded", "C5-shard-131072", "C5-shard-262144", "C5-shard-524288"):
        assert want in names
    shards = {c["of_gpus"]: c["total"] for _, c in bench.EXTRA_CONFIGS if c.get("of_gpus")}
    assert shards == {8: bench.HEADLINE_ENVS // 8, 4: bench.HEADLINE_ENVS // 4, 2: bench.HEADLINE_ENVS // 2}
