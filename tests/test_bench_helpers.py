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
        alg = envs * (147 + 9408) if pixel else envs * 235 * t["steps_per_launch"]        # (bbai_rollout: a k_step launch takes a look-ahead window's steps)
        # the render writes every byte once; k_step's gathers cost whole lines -- and a k_step_ticks launch that rewrites a small batch's outputs tick after
        # tick keeps part of them in the memory-side cache (GoToLocal 65 536: below its algorithmic bytes)
        assert (0.99 if pixel else 0.5) < t["bytes"] / alg < 3.0, (level, envs, pixel, t["bytes"] / alg)


def test_extra_configs_cover_baseline_json():
    names = [n for n, _ in bench.EXTRA_CONFIGS]
    for want in ("C2", "C3", "C4", "C4-shard", "C5-encoded", "C5-shard-131072", "C5-shard-262144", "C5-shard-524288"):
        assert want in names
    shards = {c["of_gpus"]: c["total"] for _, c in bench.EXTRA_CONFIGS if c.get("of_gpus")}
    assert shards == {8: bench.HEADLINE_ENVS // 8, 4: bench.HEADLINE_ENVS // 4, 2: bench.HEADLINE_ENVS // 2}


def _synthetic_record(n_configs=9):
    """A full record shaped like a real run's (every optional part present, long strings where round 5's line had them)."""
    cfg = {
        "envs": 1048576, "gate_timeouts": 0, "workload": "BabyAI-BossLevel-v0 56x56x3 pixel (RGBImgPartialObsWrapper) obs, 1048576 envs in total" * 2,
        "reference": "x" * 200, "value": 6.5e8, "ms_per_step": 0.123456789, "ms_per_step_median": 0.12, "mean_over_median": 1.01, "max_over_median": 6.5,
        "block_ms_list": [1.2345] * 64, "kernel_avg_ms": {"k_step": 0.1, "k_render": 1.5},
        "loop": "one bbai_rollout call per block: ONE k_step launch per look-ahead window (64 steps), the parity tap's rows written by the stepping lanes",
        "roofline": {"bound": "hbm", "kernel": "k_render", "alg_bytes_per_launch": 10019143680, "avg_launch_ms": 1.5, "steps_per_launch": 64.0, "achieved": 6646.7, "unit": "GB/s",
                     "peak": 8000.0, "frac": 0.83, "traffic": 10038800000.0, "traffic_provenance": {"source": "y" * 300}, "whole_step_alg_GBs": 6197.3},
        "parity": {"envs": 256, "steps": 3584, "mismatches": 0, "env_selection": "z" * 100},
        "cpu_baseline": {"value": 58149.6, "sample": "s" * 400}, "cpu_reference_over_port": {"provenance": "p" * 600},
    }
    return {
        "metric": "env-steps/sec", "value": 652624123.456, "unit": "env-steps/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 1.606712345,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "BabyAI-BossLevel-v0 56x56x3 pixel (RGBImgPartialObsWrapper) obs, 1048576 envs in total = 1048576 per GPU x 1, random actions, auto-reset",
                   "envs_per_gpu": 1048576, "total_envs": 1048576, "resets_in_timed_region": 67735, "parallelism": "env-shards x1, no collective", "actions": "a" * 100},
        "rccl": {"world": 1, "backend": None, "allreduce_of_ones": 1, "distinct_devices": 1, "ranks": [{"rank": 0, "device": "d" * 80}] * 8,
                 "per_rank_ms_per_step": [1.6] * 8, "per_rank_ms_per_step_min": 1.6, "per_rank_ms_per_step_max": 1.6, "launched_by": "single process"},
        "timing": {"blocks": 10, "block_ms": {"min": 32.0, "median": 32.1, "mean": 32.13, "p90": 32.5, "max": 32.75}, "block_ms_list": [32.1234] * 64,
                   "timed_seconds": 0.61, "mean_over_median": 1.002, "profiled_ms_per_step": 1.62, "clock": "c" * 400, "note": "n" * 200},
        "setup_ms": {"create": 2137.2, "seed": 746.8, "first_reset": 5.5, "note": "n" * 300},
        "roofline": dict(cfg["roofline"], achievable={"fill_GBs": 6880.0, "copy_GBs": 5200.0}, frac_of_achievable=0.966, achievable_ceiling="fill_GBs",
                         kernel_avg_ms={"k_step": 0.0945, "k_render": 1.507}, kernel_launches={"k_step": 100, "k_render": 100}),
        "parity": {"envs": 1024, "envs_all_ranks": 1024, "steps": 405, "pixel_envs": 64, "mismatches": 0, "mismatches_all_ranks": 0, "checker_errors_all_ranks": 0,
                   "env_selection": "e" * 200, "steps_checked": "s" * 100},
        "cpu_baseline": {"value": 58149.6, "unit": "env-steps/s", "cores": 16, "kind": "port", "sample": "s" * 400, "single_core_value": 3794.8,
                         "reference_over_port": 0.992, "reference_over_port_provenance": {"file": "f" * 100}},
        "configs": {"C%d-some-long-config-name" % i: dict(cfg) for i in range(n_configs)},
        "scaling_implied": {"basis": "b" * 150, "one_gpu_ms_per_step": 1.6067,
                            "gpus": {str(g): {"config": "C5-shard", "envs_per_gpu": 1048576 // g, "ms_per_step_shard": 0.223, "implied_value": 4.7e9, "implied_efficiency": 0.9} for g in (8, 4, 2)},
                            "C4": {"gpus": 8, "envs_per_gpu": 131072, "one_gpu_ms_per_step": 0.196, "ms_per_step_shard": 0.0396, "implied_value": 2.6e10, "implied_efficiency": 0.62}},
        "gate_timeouts": 0, "state_layout": "classic (live record per env, k_consume / in-wave copy on reset)", "build": {"commit": "abcdef0", "csrc_sha": "0123456789abcdef"},
        "wall_seconds": 75.2,
    }


def test_judged_line_is_small():
    """BENCH_r05.parsed was null because the one stdout line was 32 KB.  Whatever a run measured, the line stays under 6 KB and
    keeps what the harness and the judge read."""
    rec = _synthetic_record(9)
    assert len(json.dumps(rec)) > 12000                   # (the full record is what round 5 printed)
    s = bench.compact_line(rec, "gpurun_out/bench_full.json")
    assert "\n" not in s and len(s.encode()) < 6000, len(s)
    line = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "parity", "configs", "scaling_implied", "build", "gate_timeouts", "full_record"):
        assert k in line, k
    assert line["config"]["workload"].startswith("BabyAI-BossLevel-v0") and line["config"]["total_envs"] == 1048576
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes_per_launch", "avg_launch_ms", "frac_of_achievable"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample", "reference_over_port"):
        assert k in line["cpu_baseline"], k
    assert line["parity"]["mismatches"] == 0 and line["build"]["csrc_sha"] == "0123456789abcdef"
    assert len(line["configs"]) == 9 and all("ms_per_step" in c and "frac" in c and "traffic_ratio" in c and "mismatches" in c for c in line["configs"].values())
    assert all(c["loop"] == "rollout" and c["steps_per_launch"] == 64.0 for c in line["configs"].values()) and line["timing"]["loop"] in ("step", "rollout")
    assert set(line["scaling_implied"]["gpus"]) == {"8", "4", "2"} and line["scaling_implied"]["C4"]["gpus"] == 8
    assert abs(line["value"] - rec["value"]) / rec["value"] < 1e-6 and abs(line["ms_per_step"] - rec["ms_per_step"]) / rec["ms_per_step"] < 1e-6
    # a run with far more configs than any real one still fits: the optional parts are shed, the required ones stay
    s = bench.compact_line(_synthetic_record(60), None)
    line = json.loads(s)
    assert len(s.encode()) < 6000 and line["roofline"]["frac"] == 0.83 and line["cpu_baseline"]["value"] == 58149.6 and len(line["configs"]) == 60


def test_full_record_goes_to_a_side_file(tmp_path):
    path = bench.write_full_record(_synthetic_record(2), str(tmp_path / "sub" / "full.json"))
    assert path == str(tmp_path / "sub" / "full.json") and json.load(open(path))["n_gpus"] == 1
