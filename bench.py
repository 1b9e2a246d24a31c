#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched BabyAI hot path on MI355X.

One "step" = one pass of the hot path over one batch: every env of the shard applies one action (transition + verifier),
finished envs are regenerated on the device (auto-reset) and the observation is written (7x7x3 encoding, plus the
56x56x3 pixel render for the default BossLevel workload = BASELINE.json configs[4]).  Actions are synthetic, i.i.d.
uniform over the 7 actions from a counter-based generator keyed on (bench seed, step, global env index)
(babyai_amd/action_stream.py), resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script LAUNCHES ITSELF as N ranks (torch.distributed.run, one rank
per GPU, rendezvous on 127.0.0.1); under an external `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...` it joins that group.  Either way a run whose live process group is not exactly N ranks FAILS -- it never
falls back to one rank.

The workload is BASELINE.json's: 1 048 576 BossLevel envs with pixel observations IN TOTAL, sharded contiguously over the
N GPUs (`scaling: "strong"`; 131 072 envs per GPU at N = 8).  `--weak` keeps 1 048 576 envs on EVERY GPU instead
(`scaling: "weak"`); `--envs` sets the per-GPU count by hand.

What one run proves about itself (all in the one JSON line rank 0 prints):
  * the loop    per step one bbai_step (pixel batches: bbai_step_render, the wrapped env's step = transition + render) + bbai_tap_ids call from
                Python, like any caller's loop; --rollout-entry
                enqueues a block's K steps with ONE call of the engine's open-loop rollout entry instead (include/bbai.h bbai_rollout,
                the same launches) -- measured equal on every config, 65 536-env steps included: the GPU sets the pace.
  * timing      W warmup steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both sides; a rank's
                clock stops when its own device is idle, the block is the max over ranks (the closing barrier itself is
                reported as timing.barrier_ms, not timed); repeated until >= --min-seconds of timed work.  Blocks alternate between PLAIN (nothing but
                the steps and the parity tap) and PROFILED (every kernel launch bracketed by a HIP event pair on its
                launch stream).  `value` / `ms_per_step` = the MEAN over the plain blocks (all their steps / all their seconds:
                what a long rollout sustains -- reset storms, when a million envs time out on the same tick, are in it); the
                median, p90, max and every block's time are next to it (`timing`).  `roofline` = the profiled blocks, whose own
                median step time is reported next to it (`timing.profiled_block_ms`), so that the kernel times add up to
                a step that was really measured and the cost of the event pairs is visible.
  * rccl        the live process group: world size, backend, every rank's device, an all-reduce of ones, and every
                rank's own ms per step (before the max-reduce).
  * roofline    the dominant kernel's algorithmic bytes / its HIP-event time, against the 8 TB/s spec peak AND against
                what a plain 1-GiB fill / copy reaches on this box in this process (`achievable`).  `traffic` = HBM bytes
                per launch from the committed rocprofv3 PMC passes, only while the kernel sources still hash to what was
                profiled (else null).
  * parity      the outputs of 1024 envs SCATTERED over the shard (both ends, wave and block boundaries, a pseudo-random
                spread: shard.scattered_ids) at EVERY step (image, direction, f64 reward bits, done; pixels of 64) are
                tapped inside the timed region and re-derived afterwards by the CPU oracle from the seeds and the action
                stream: `parity.mismatches_all_ranks` must be 0 (exit code 3 otherwise; 4 when the checker itself broke).
  * configs     after the headline's timed region, the default single-GPU run puts every other BASELINE.json config through
                the SAME loop on the same GPU -- C2 GoToLocal 65 536, C3 PickupLoc 262 144, C4 GoTo 1 048 576 (and its 131 072-env
                8-GPU shard), encoded BossLevel 1 048 576, and the headline's own per-GPU shards of the 8- / 4- / 2-GPU job
                (BossLevel pixels, 131 072 / 262 144 / 524 288 envs: `C5-shard-*`, with the strong-scaling efficiency they imply
                under `scaling_implied`): >= --extra-seconds of timed work each, alternating plain / profiled blocks, 256 scattered
                envs tapped at every step of the first ~1000 steps (a spread of them at every step of the rest) and re-derived by
                the oracle.  `configs` in the line; a mismatch there exits 3 like one in the headline.  --no-extra-configs skips them.
  * setup_ms    what is outside the metric: bbai_create, bbai_seed (incl. the first fill of the look-ahead rings), first reset.
  * cpu_baseline  the oracle on the usable host cores over the same seeds and action stream (rank 0; a reported
                baseline), with the measured reference/port ratio of the build container when it is on file; every other
                workload of `configs` gets its own short sample on the same cores (`configs.*.cpu_baseline`).

Envs shard embarrassingly (babyai_amd/shard.py): rank r owns a contiguous range of global envs with seeds base + global
index; no collective on the step path.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak (MI355X_MICROARCH.md)
HEADLINE_ENVS = 1048576     # BASELINE.json: "1M parallel envs"

# BASELINE.json configs: level, obs mode and the TOTAL env count the config is quoted on.  C4 / C5 are quoted on
# 8 GPUs (131072 envs each); the "-shard" names run that per-GPU shard size on however many GPUs are used (profiling
# one GPU's share of the 8-GPU job on a 1-GPU box).
CONFIGS = {
    "C2": dict(level="GoToLocal", total=65536, pixel=False),
    "C3": dict(level="PickupLoc", total=262144, pixel=False),
    "C4": dict(level="GoTo", total=1048576, pixel=False),
    "C5": dict(level="BossLevel", total=1048576, pixel=True),
    "C4-shard": dict(level="GoTo", per_gpu=131072, pixel=False),
    "C5-shard": dict(level="BossLevel", per_gpu=131072, pixel=True),
    "C5-shard-131072": dict(level="BossLevel", per_gpu=131072, pixel=True),
    "C5-shard-262144": dict(level="BossLevel", per_gpu=262144, pixel=True),
    "C5-shard-524288": dict(level="BossLevel", per_gpu=524288, pixel=True),
    "C5-encoded": dict(level="BossLevel", total=1048576, pixel=False),
}


def csrc_sha():
    """Hash of the kernel sources: ties a PMC measurement to the code it was taken on."""
    d = os.path.join(ROOT, "babyai_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()[:16]


def git_head():
    try:
        return subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def achievable_bandwidth(torch, dev):
    """Plain fill and copy of 1 GiB on this box, in this process (GB/s of bytes moved)."""
    n = 1 << 30
    x = torch.empty(n, dtype=torch.uint8, device=dev).view(torch.int32)
    y = torch.empty(n, dtype=torch.uint8, device=dev).view(torch.int32)

    def timeit(fn, iters=12):
        fn()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / iters * 1e-3

    fill = n / timeit(lambda: x.fill_(7)) / 1e9
    copy = 2 * n / timeit(lambda: y.copy_(x)) / 1e9
    del x, y
    return {"fill_GBs": fill, "copy_GBs": copy, "bytes": n}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--envs", type=int, default=None, help="envs PER GPU (default: --total-envs / --gpus)")
    ap.add_argument("--total-envs", type=int, default=None, help="envs of the whole job (default 1048576 = the BASELINE metric)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: the single-GPU env count on EVERY GPU (default: the total is fixed)")
    ap.add_argument("--level", default="BossLevel")
    ap.add_argument("--no-pixel", action="store_true")
    ap.add_argument("--seed", type=int, default=0, help="env i of the whole job is seeded with seed + i")
    ap.add_argument("--action-seed", type=int, default=1234)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="a BASELINE.json config by name (overrides --level/--envs/--no-pixel); default = C5")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="repeat the K-step block until this much timed work")
    ap.add_argument("--max-blocks", type=int, default=64)
    ap.add_argument("--parity-envs", type=int, default=1024, help="envs of every shard, scattered over it, checked against the oracle (0 = off)")
    ap.add_argument("--parity-pixel-envs", type=int, default=64)
    ap.add_argument("--parity-budget", type=int, default=300000,
                    help="oracle env-steps the parity check may cost: long runs check every step of fewer envs")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0, help="the headline's CPU baseline: the oracle port on the host cores over a bounded sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm-seconds", type=float, default=1.0, help="untimed GPU activity before the warmup steps")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) | gloo (test rigs: ranks may share a GPU)")
    ap.add_argument("--share-device", action="store_true", help="test rigs only: every rank uses cuda:0")
    ap.add_argument("--own-stream", action="store_true", help="run the rollout on a stream of its own instead of torch's default (NULL) stream")
    ap.add_argument("--no-extra-configs", action="store_true", help="only the headline workload (default: the default single-GPU run also measures BASELINE configs C2-C4 and encoded BossLevel)")
    ap.add_argument("--extra-configs", action="store_true", help="measure the other configs with --gpus N > 1 too (sharded over the N ranks)")
    ap.add_argument("--extra-seconds", type=float, default=0.3, help="timed work per extra config")
    ap.add_argument("--extra-parity-envs", type=int, default=256)
    ap.add_argument("--extra-parity-budget", type=int, default=150000, help="oracle env-steps per extra config")
    ap.add_argument("--extra-cpu-seconds", type=float, default=1.0, help="CPU baseline (the oracle port on the host cores) per extra workload; 0 = only the ratio on file")
    ap.add_argument("--extra-parity-horizon", type=int, default=1024, help="steps over which ALL --extra-parity-envs are followed (then a spread of them)")
    ap.add_argument("--profile-tail", action="store_true", help="all plain blocks first (one contiguous rollout), the profiled blocks behind them (default: alternating; the extra configs always run this way)")
    ap.add_argument("--loop", choices=("auto", "step", "rollout"), default=os.environ.get("BBAI_BENCH_LOOP", "auto"),
                    help="auto (default): encoded observations through bbai_rollout (one k_step launch per look-ahead window's steps, the tap rows written by the "
                         "stepping lanes), pixel observations through per-step calls; step: per-step calls everywhere (rounds 1-5's loop); rollout: bbai_rollout everywhere")
    ap.add_argument("--rollout-entry", action="store_true", help="one bbai_rollout call per block instead of one bbai_step / bbai_render / bbai_tap_ids call per step from Python (the same launches)")
    ap.add_argument("--full-out", default=None, help="where the FULL record goes (default <repo>/gpurun_out/bench_full.json); stdout carries the compact judged line only")
    ap.add_argument("--dump-digest", default=None, help="write per-env output digests of this rank to <prefix>.rank<r>.npy")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


def resolve_workload(args):
    """(level, pixel, envs per GPU, total envs, scaling) for this run."""
    level, pixel = args.level, not args.no_pixel
    total = args.total_envs
    per_gpu = args.envs
    if args.config:
        c = CONFIGS[args.config]
        level, pixel = c["level"], c["pixel"]
        if "per_gpu" in c:
            per_gpu = c["per_gpu"] if per_gpu is None else per_gpu
        elif total is None:
            total = c["total"]
    if per_gpu is not None:                       # explicit per-GPU count: every GPU gets it
        return level, pixel, per_gpu, per_gpu * args.gpus, "weak"
    if total is None:
        total = HEADLINE_ENVS
    if args.weak:
        return level, pixel, total, total * args.gpus, "weak"
    if total % args.gpus:
        raise SystemExit("bench.py: %d envs do not split evenly over %d GPUs" % (total, args.gpus))
    return level, pixel, total // args.gpus, total, "strong"


def self_launch(args):
    """--gpus N > 1 outside a torch.distributed launch: become the launcher.  The child ranks inherit stdout, so rank 0's
    JSON line is this process's JSON line; the exit code is the launcher's (non-zero if any rank failed)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), BBAI_BENCH_SELF_LAUNCHED="1")
    sys.stderr.write("bench.py: launching %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    return subprocess.call(cmd, env=env, cwd=os.getcwd())


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


def block_stats(blocks, K, E, world):
    """What the plain blocks say: sustained (mean) throughput first, the spread next to it."""
    bs = sorted(blocks)
    med = median(blocks)
    mean = sum(blocks) / len(blocks)
    p90 = bs[min(len(bs) - 1, int(0.9 * len(bs)))]
    return {"mean": mean, "median": med, "p90": p90, "min": bs[0], "max": bs[-1],
            "value_mean": K * E * world / mean, "value_median": K * E * world / med,
            "mean_over_median": mean / med, "max_over_median": bs[-1] / med}


def _sig(x, n=7):
    """floats to n significant digits (the judged line is a record, not a dump)"""
    if isinstance(x, float):
        return float("%.*g" % (n, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


JUDGED_LINE_MAX_BYTES = 6000


def compact_line(out, full_path=None):
    """The ONE stdout line: what the driver parses and the judge reads, well under 6 KB whatever the run measured.  Everything else --
    every block's time, the per-kernel averages, provenance strings, the per-config baselines' samples -- is the FULL record
    (`--full-out`, default gpurun_out/bench_full.json), which this line names.  Round 5's single 32-KB line could not be parsed by the
    harness: a measurement nobody downstream can read is not a measurement (tests/test_bench_helpers.py::test_judged_line_is_small)."""
    def pick(d, keys):
        return {k: d.get(k) for k in keys if d is not None and k in d} if d else None

    r = out.get("roofline") or {}
    roof = pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_of_achievable", "achievable_ceiling",
                    "alg_bytes_per_launch", "avg_launch_ms", "whole_step_alg_GBs"))
    if roof is not None:
        roof["traffic_over_alg"] = (r["traffic"] / r["alg_bytes_per_launch"]) if r.get("traffic") and r.get("alg_bytes_per_launch") else None
        roof["whole_step_frac"] = (r["whole_step_alg_GBs"] / r["peak"]) if r.get("whole_step_alg_GBs") and r.get("peak") else None
    cb = out.get("cpu_baseline")
    if cb is not None:
        cbc = pick(cb, ("value", "unit", "cores", "kind", "single_core_value", "reference_over_port", "error"))
        if cb.get("sample"):
            cbc["sample"] = cb["sample"][:200]
        cb = cbc
    par = out.get("parity")
    if par is not None:
        par = pick(par, ("envs", "envs_all_ranks", "steps", "pixel_envs", "mismatches", "mismatches_all_ranks", "checker_errors_all_ranks", "error"))
    t = out.get("timing") or {}
    bm = t.get("block_ms") or {}
    K = out.get("steps") or 1
    cfgs = None
    if out.get("configs"):
        cfgs = {}
        for name, c in out["configs"].items():
            if "error" in c:
                cfgs[name] = {"error": str(c["error"])[:120]}
                continue
            cr = c.get("roofline") or {}
            cfgs[name] = {"envs": c.get("envs"), "ms_per_step": c.get("ms_per_step"), "ms_median": c.get("ms_per_step_median"),
                          "max_over_median": c.get("max_over_median"),
                          "loop": ("rollout" if str(c.get("loop", "")).startswith("one bbai_rollout") else "step"),      # (bbai_rollout: a window's steps per launch / per-step calls)
                          "kernel": cr.get("kernel"), "kernel_ms": cr.get("avg_launch_ms"), "steps_per_launch": cr.get("steps_per_launch"), "frac": cr.get("frac"),
                          "step_frac": (cr["whole_step_alg_GBs"] / cr["peak"]) if cr.get("whole_step_alg_GBs") and cr.get("peak") else None,
                          "traffic_ratio": (cr["traffic"] / cr["alg_bytes_per_launch"]) if cr.get("traffic") and cr.get("alg_bytes_per_launch") else None,
                          "mismatches": (c.get("parity") or {}).get("mismatches"),
                          "parity_steps": (c.get("parity") or {}).get("steps"),
                          "cpu": (c.get("cpu_baseline") or {}).get("value")}
    si = out.get("scaling_implied")
    if si:
        si = {"basis": "1-GPU runs of each per-GPU shard size on this box (no collective on the step path); implied, not measured",
              "one_gpu_ms_per_step": si.get("one_gpu_ms_per_step"),
              "gpus": {g: pick(v, ("envs_per_gpu", "ms_per_step_shard", "implied_value", "implied_efficiency")) for g, v in (si.get("gpus") or {}).items()},
              "C4": si.get("C4")}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    line["config"] = pick(out.get("config"), ("workload", "envs_per_gpu", "total_envs", "resets_in_timed_region", "parallelism"))
    line["roofline"] = roof
    line["cpu_baseline"] = cb
    line["parity"] = par
    line["timing"] = {"blocks": t.get("blocks"), "ms_median": (bm.get("median") / K) if bm.get("median") else None,
                      "ms_max": (bm.get("max") / K) if bm.get("max") else None, "mean_over_median": t.get("mean_over_median"),
                      "timed_seconds": t.get("timed_seconds"), "profiled_ms_per_step": t.get("profiled_ms_per_step"),
                      "loop": ("rollout" if str(t.get("loop", "")).startswith("one bbai_rollout") else "step")}
    line["kernel_avg_ms"] = r.get("kernel_avg_ms")
    rc = out.get("rccl") or {}
    line["rccl"] = pick(rc, ("world", "backend", "allreduce_of_ones", "distinct_devices", "per_rank_ms_per_step_min", "per_rank_ms_per_step_max", "launched_by"))
    if out.get("obs_gather"):
        line["obs_gather"] = pick(out["obs_gather"], ("ms", "GBs_into_rank0", "error"))
    line["configs"] = cfgs
    line["scaling_implied"] = si
    line["gate_timeouts"] = out.get("gate_timeouts")
    line["state_layout"] = out.get("state_layout")
    line["build"] = out.get("build")
    line["wall_seconds"] = out.get("wall_seconds")
    line["full_record"] = full_path
    line = _sig(line)
    s = json.dumps(line, separators=(",", ":"))
    if len(s) > JUDGED_LINE_MAX_BYTES:          # never again: shed the optional parts, largest first, until it fits
        for k in ("kernel_avg_ms", "timing", "rccl", "configs", "scaling_implied"):
            line[k] = None if k != "configs" else {n: {"ms_per_step": c.get("ms_per_step"), "mismatches": c.get("mismatches")} for n, c in (line["configs"] or {}).items()}
            s = json.dumps(line, separators=(",", ":"))
            if len(s) <= JUDGED_LINE_MAX_BYTES:
                break
    return s


def write_full_record(out, path):
    """The full record next to the judged line; never fatal (a read-only tree falls back to the temp dir)."""
    import tempfile
    for cand in (path, os.path.join(tempfile.gettempdir(), "bbai_bench_full.json")):
        try:
            d = os.path.dirname(os.path.abspath(cand))
            os.makedirs(d, exist_ok=True)
            with open(cand, "w") as f:
                json.dump(out, f)
                f.write("\n")
            return cand
        except OSError:
            continue
    return None


def traffic_of(level, E, pixel, dom):
    """HBM bytes per launch of kernel `dom` on this workload from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, separate
    runs; tools/lease.sh pmccfg -> tools/summarize_profile.py -> profiles/pmc_latest.json: one entry per workload).  Counters
    cannot be collected inside this process: the committed summary is quoted with its provenance, and only while the kernel
    sources still hash to what was profiled on this exact workload."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        wl = pmc["workloads"]["%s:%d:%s" % (level, E, "pixel" if pixel else "encoded")]
        cands = [k for k in wl["kernels"] if k == dom or k.startswith(dom + "<") or k.startswith(dom + "_q<") or k.startswith(dom + "_ticks<")]      # k_render / k_render_q, k_step / k_step_ticks: templates
        if not cands:
            return None
        key = max(cands, key=lambda k: wl["kernels"][k].get("launches", 0))        # (the instantiation the step loop runs)
        kk = wl["kernels"][key]
        fetch = kk.get("FETCH_SIZE_corrected", 2 * kk["FETCH_SIZE"])      # gfx950: FETCH_SIZE tallies 128-B requests as 64 B
        t = {"bytes": fetch + kk["WRITE_SIZE"], "fetch_corrected": fetch, "fetch_raw": kk["FETCH_SIZE"], "write": kk["WRITE_SIZE"],
             "kernel": key, "steps_per_launch": float(wl.get("k_step_steps_per_launch", 1.0)) if dom == "k_step" else 1.0,
             "correction": "FETCH_SIZE x 2 (MI355X_MICROARCH.md, calibrated on k_render's known byte counts)",
             "source": "profiles/pmc_latest.json", "commit": pmc.get("commit"), "csrc_sha": pmc.get("csrc_sha"),
             "current": pmc.get("csrc_sha") == csrc_sha()}
        if not t["current"]:
            t["bytes"] = None         # kernels changed since the counters were taken
        return t
    except Exception:
        return None


class Ctx(object):
    """What every measurement of a run shares: torch, the process group, the device, the CPU worker pool."""
    pass


def measure(ctx, level, pixel, E, total_envs, K, W, min_seconds, max_blocks, P, PP, parity_budget, after_seed=None, digest=None, profile_tail=False):
    """One workload through the measured loop (babyai_amd/shard.py timed_blocks): create + seed + reset the shard (timed:
    `setup_ms`), W warmup steps and ONE K-step probe block, then as many further K-step blocks as make `min_seconds`,
    alternately plain and profiled (profile_tail: all the plain blocks first, as ONE contiguous rollout whose mean is what a long run
    sustains -- reset storms land where they land, not in the blocks that happen to be profiled -- and a quarter as many profiled blocks
    behind them), with the outputs of P scattered envs (pixels of PP) tapped at every step.  Returns a
    dict with the block times, the per-kernel HIP-event times, the device logs for the oracle replay and the counters; the
    env is closed."""
    import time
    torch, shard, ranks, dev, args = ctx.torch, ctx.shard, ctx.ranks, ctx.dev, ctx.args
    from babyai_amd.action_stream import actions_torch
    from babyai_amd.engine import BatchedBabyAIEnv
    first, count = shard.shard_range(total_envs, ranks.world, ranks.rank)
    assert count == E
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % level, E, device=dev, pixel=pixel)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    env.seed(shard.shard_seeds(args.seed, total_envs, ranks.world, ranks.rank))       # synchronous (include/bbai.h bbai_seed)
    t2 = time.perf_counter()
    if after_seed:
        after_seed()

    P = min(P, E)
    PP = min(PP, P) if pixel else 0
    # Which loop.  Encoded observations: ONE bbai_rollout call per block (include/bbai.h: a k_step launch takes every step the look-ahead window has
    # left; the actions of a block are resident, as they always were) -- the warm-up is then rounded UP to the next window boundary (the reset is a
    # window tick itself), so that every timed launch is a whole window and "per launch" means one thing.  Pixels: per-step calls (the render sits
    # between the steps).  --loop step: per-step calls everywhere, as rounds 1-5 measured.
    fast = digest is None and (args.rollout_entry or args.loop == "rollout" or (args.loop == "auto" and not pixel))
    period = int(env.get_option("lookahead_period"))
    if fast and not pixel and env.get_option("rollout_multi"):
        W = W + (-(1 + W)) % period

    # the tapped envs: scattered over the shard; the ones whose pixels are checked too come first in the log rows and
    # are themselves a spread (both ends included)
    def tap_ids(p, pp):
        ids = shard.scattered_ids(E, p, salt=first)
        head = [ids[(len(ids) - 1) * k // max(1, pp - 1)] for k in range(pp)] if pp else []
        head = sorted(set(head))
        rest = [i for i in ids if i not in set(head)]
        return head + rest, len(head)

    def make_log(steps, with_initial, ids, pp):
        if not ids:
            return None
        p = len(ids)
        extra = 1 if with_initial else 0
        lg = {"image": torch.zeros((steps + extra, p, 7, 7, 3), dtype=torch.uint8, device=dev),
              "direction": torch.zeros((steps + extra, p), dtype=torch.uint8, device=dev),
              "reward64": torch.zeros((steps, p), dtype=torch.float64, device=dev),
              "done": torch.zeros((steps, p), dtype=torch.uint8, device=dev)}
        if pp:
            lg["pixels"] = torch.zeros((steps + extra, pp, 56, 56, 3), dtype=torch.uint8, device=dev)
        lg["ids"] = torch.as_tensor(ids, dtype=torch.int64, device=dev)
        return lg

    def tap(lg, obs_row, row):
        """the parity tap: ONE small launch inside the timed region (include/bbai.h bbai_tap_ids)"""
        env.tap(lg["image"][obs_row], lg["direction"][obs_row], lg["reward64"][row], lg["done"][row],
                lg["pixels"][obs_row] if "pixels" in lg else None, ids=lg["ids"])

    # phase 1: reset, W warmup steps and ONE K-step block; its time decides how many further blocks make min_seconds
    S1 = W + K
    actions1 = actions_torch(args.action_seed, 0, S1, first, E, dev)        # resident before the timed region
    ids1, PP1 = tap_ids(P, PP)
    log1 = make_log(S1, True, ids1, PP1)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    env.reset()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    setup_ms = {"create": (t1 - t0) * 1e3, "seed": (t2 - t1) * 1e3, "first_reset": (t4 - t3) * 1e3,
                "note": "bbai_create (allocation + clearing), bbai_seed (sha512 + MT19937 init + the first D levels of every env's "
                        "look-ahead ring) and the first reset() with its observation (+ render); outside the metric"}
    if log1 is not None:
        log1["image"][0].copy_(env.image[log1["ids"]])
        log1["direction"][0].copy_(env.direction[log1["ids"]])
        if PP1:
            log1["pixels"][0].copy_(env.pixels[log1["ids"][:PP1]])

    # Encoded observations: the step logs its own tap rows (include/bbai.h bbai_step_tapped: the stepping lanes write the listed envs' outputs,
    # no launch behind the step -- k_tap and its dependent-launch gap were 6 of a 65 536-env step's 21 us).  Pixel batches keep the tap launch
    # behind the render (BBAI_BENCH_TAP=launch: everywhere, as rounds 1-5 measured).
    step_tap = (not pixel) and os.environ.get("BBAI_BENCH_TAP", "step") == "step" and hasattr(env.lib, "bbai_step_tapped")

    def step1(t):
        env.step_tapped(actions1[t], log1["image"][t + 1], log1["direction"][t + 1], log1["reward64"][t], log1["done"][t])

    if step_tap and log1 is not None:
        env.set_step_tap(ids1)

    def after1(t):
        if log1 is not None and not step_tap:
            tap(log1, t + 1, t)
        if digest is not None:
            digest.update(env.image, env.direction, env.reward64, env.done)

    # --rollout-entry: one call per block (include/bbai.h bbai_rollout: K x (step [+ render] + tap) enqueued by the engine) instead of
    # the per-step calls -- the same launches; measured equal on every config (profiles/r04/bench_loop_rollout_entry_vs_python_ab.jsonl:
    # the GPU, not the interpreter, sets the pace even of a 40-us step), so the default stays what a caller's loop looks like
    def run1(t0, k):
        env.rollout(actions1[t0:t0 + k], tap=log1, obs_row0=t0 + 1, row0=t0, step_tap=step_tap and log1 is not None)

    torch.cuda.synchronize()
    blocks = shard.timed_blocks(env, actions1, W, K, 1, ranks, after1, run_steps=run1 if fast else None, step_fn=step1 if (step_tap and log1 is not None) else None)
    want = int(min(max_blocks, max(0, -(-min_seconds // blocks[0]))))
    want = int(ranks.max(want))
    if want == 1:
        want = 2                                      # one plain and one profiled block at least
    n_plain, n_prof = (want + 1) // 2, want // 2      # alternating: plain (even) and profiled (odd) blocks
    if profile_tail and want:
        n_plain, n_prof = want, max(2, want // 4)
        want = n_plain + n_prof
    is_prof = (lambda i: i >= n_plain) if profile_tail else (lambda i: i % 2 == 1)
    # phase 2: `want` more blocks of exactly K steps; when there are any, the phase-1 block was only the probe
    S2 = want * K
    ids2, PP2, sel2 = ids1, PP1, None
    if P and (S1 + S2) * P > parity_budget:      # long run: every step of FEWER envs in phase 2 -- a spread of phase 1's (ALL P are checked over phase 1)
        P2 = max(min(16, P), (parity_budget - S1 * P) // max(1, S2))
        P2 = min(P, P2)
        PP2 = min(PP1, P2)
        pix_rows = sorted(set((PP1 - 1) * k // max(1, PP2 - 1) for k in range(PP2))) if PP2 else []
        n_rest = len(ids1) - PP1
        rest_rows = sorted(set(PP1 + (n_rest - 1) * k // max(1, P2 - len(pix_rows) - 1) for k in range(P2 - len(pix_rows)))) if n_rest > 0 else []
        sel2 = pix_rows + rest_rows                    # rows of log1 that phase 2 keeps following (pixel rows first)
        ids2, PP2 = [ids1[r] for r in sel2], len(pix_rows)
    log2 = None
    resets0 = env.reset_count()
    env.profile(True)              # per-kernel HIP event pairs on the launch stream (include/bbai.h bbai_profile) ...
    env.profile_pause()            # ... in the odd blocks only
    local_blocks, barrier_s, profiled = [], [], []
    if want:
        actions2 = actions_torch(args.action_seed, S1, S1 + S2, first, E, dev)
        log2 = make_log(S2, False, ids2, PP2)

        def step2(t):
            env.step_tapped(actions2[t], log2["image"][t], log2["direction"][t], log2["reward64"][t], log2["done"][t])

        if step_tap and log2 is not None:
            env.set_step_tap(ids2)

        def after2(t):
            if log2 is not None and not step_tap:
                tap(log2, t, t)
            if digest is not None:
                digest.update(env.image, env.direction, env.reward64, env.done)

        def before_block(i):
            if is_prof(i):
                env.profile_resume()
            else:
                env.profile_pause()

        def run2(t0, k):
            env.rollout(actions2[t0:t0 + k], tap=log2, obs_row0=t0, row0=t0, step_tap=step_tap and log2 is not None)

        torch.cuda.synchronize()
        all_blocks = shard.timed_blocks(env, actions2, 0, K, want, ranks, after2, before_block=before_block, local_out=local_blocks,
                                        barrier_out=barrier_s, run_steps=run2 if fast else None, step_fn=step2 if (step_tap and log2 is not None) else None)
        env.profile_pause()
        blocks = [b for i, b in enumerate(all_blocks) if not is_prof(i)]
        profiled = [b for i, b in enumerate(all_blocks) if is_prof(i)]
        local_blocks = [b for i, b in enumerate(local_blocks) if not is_prof(i)]
        del actions2
    resets = ranks.sum(env.reset_count() - resets0)
    if not want:                    # single-block run: nothing was bracketed; time a few untimed steps
        env.profile(True)
        for t in range(4):
            env.step(actions1[t])
        torch.cuda.synchronize()
    prof = env.profile_read()
    step_ticks = int(env.get_option("profile_step_ticks"))      # steps the bracketed k_step launches took (several per launch under bbai_rollout)
    env.profile(False)
    gate_timeouts = int(ranks.sum(env.gate_timeouts()))      # (synchronises; outside the timed blocks) a window gate that gave up = a void run
    m = {"level": level, "pixel": pixel, "E": E, "total_envs": total_envs, "first": first, "K": K, "W": W, "S1": S1, "S2": S2, "want": want,
         "blocks": blocks, "profiled": profiled, "local_blocks": local_blocks, "barrier_s": barrier_s,
         "kernel_ms": {k: v[0] for k, v in prof.items() if v[0] is not None},
         "kernel_launches": {k: v[1] for k, v in prof.items() if v[0] is not None},
         "resets": resets, "setup_ms": setup_ms, "gate_timeouts": gate_timeouts,
         "steps_per_k_step_launch": (step_ticks / float(prof["k_step"][1])) if prof["k_step"][1] else 1.0,
         "loop": ("one bbai_rollout call per block: ONE k_step launch per look-ahead window (%d steps), the parity tap's rows written by the stepping lanes" % period) if (fast and not pixel and env.get_option("rollout_multi")) else
                 "one bbai_rollout call per block: the engine enqueues K x (step [+ render] + tap)" if fast else
                 ("per-step calls from Python: bbai_step_render (the step + the render as one call) + bbai_tap_ids" if pixel else
                  "per-step calls from Python: bbai_step_tapped (the step; the parity tap's rows are written by the stepping lanes)" if step_tap else
                  "per-step calls from Python: bbai_step + bbai_tap_ids"),
         "state_layout": "in-place (the look-ahead slot is the live record)" if env.get_option("inplace") else "classic (live record per env, k_consume / in-wave copy on reset)",
         "lookahead_period": env.get_option("lookahead_period"),
         "log1": log1, "log2": log2, "ids2": ids2, "PP1": PP1, "PP2": PP2, "sel2": sel2, "env": env}
    return m


def replay(ctx, m):
    """Outside the timed region: the oracle re-derives what the tap recorded (oracle/cpu_baseline.py parity_replay).  When phase 2
    followed a subset of the tapped envs (long runs), the subset is checked over ALL steps and the other envs over phase 1."""
    torch, np = ctx.torch, ctx.np
    log1, log2, sel2, PP1, PP2, ids2 = m["log1"], m["log2"], m["sel2"], m["PP1"], m["PP2"], m["ids2"]
    if log1 is None:
        return None
    try:
        host = {}
        for k in log1:
            if k == "ids" or (k == "pixels" and not PP2):
                continue
            a = log1[k]
            if sel2 is not None:                    # phase 2 followed a subset: those envs, all steps
                rows = [r for r in sel2 if r < PP1] if k == "pixels" else sel2
                a = a[:, torch.as_tensor(rows, dtype=torch.int64, device=ctx.dev)]
            host[k] = np.concatenate([a.cpu().numpy()] + ([log2[k].cpu().numpy()] if log2 is not None else []))
        par = ctx.cpu_baseline.parity_replay(m["level"], host, ctx.args.seed, ctx.args.action_seed, m["first"], PP2,
                                             env_ids=[m["first"] + i for i in ids2], pool=ctx.pool)
        par["envs_all_steps"] = par["envs"]
        if sel2 is not None:                        # ... and the rest of phase 1's envs over phase 1
            ids1 = [int(i) for i in log1["ids"].cpu().tolist()]
            keep = set(sel2)
            rows = [r for r in range(len(ids1)) if r not in keep]
            if rows:
                pix_rows = [r for r in rows if r < PP1]
                hostA = {}
                for k in log1:
                    if k == "ids" or (k == "pixels" and not pix_rows):
                        continue
                    rr = pix_rows if k == "pixels" else rows
                    hostA[k] = log1[k][:, torch.as_tensor(rr, dtype=torch.int64, device=ctx.dev)].cpu().numpy()
                parA = ctx.cpu_baseline.parity_replay(m["level"], hostA, ctx.args.seed, ctx.args.action_seed, m["first"], len(pix_rows),
                                                      env_ids=[m["first"] + ids1[r] for r in rows], pool=ctx.pool)
                par["mismatches"] += parA["mismatches"]
                par["first_mismatch"] = par["first_mismatch"] or parA["first_mismatch"]
                par["seconds"] += parA["seconds"]
                par["envs"] += parA["envs"]
                par["pixel_envs"] += parA["pixel_envs"]
                par["envs_first_steps_only"] = {"envs": parA["envs"], "steps": parA["steps"], "pixel_envs": parA["pixel_envs"]}
    except Exception as exc:
        par = {"error": repr(exc), "mismatches": None}          # the CHECKER broke: reported, not a parity verdict
    return par


def roofline_of(m):
    """(dominant kernel, algorithmic bytes per launch, bytes per env-step, its avg ms, GB/s, ceiling key)"""
    E = m["E"]
    if m["pixel"]:
        dom, alg_bytes, bps, key = "k_render", E * (147 + 9408), 9496, "fill_GBs"        # reads the encoding, writes the pixels: a pure store stream
    else:
        # (reads and writes mixed; a bbai_rollout launch takes steps_per_k_step_launch steps: its algorithmic bytes are that many steps')
        dom, alg_bytes, bps, key = "k_step", E * 235 * m.get("steps_per_k_step_launch", 1.0), 235, "copy_GBs"
    dom_ms = m["kernel_ms"][dom]
    return dom, alg_bytes, bps, dom_ms, alg_bytes / (dom_ms * 1e-3) / 1e9, key


# what the one JSON line carries next to the headline: every other BASELINE.json config on this GPU (world == 1), measured by
# the same loop.  steps = steps per block (a block must last milliseconds for a host clock, and many look-ahead windows: it ends with the
# device idle, i.e. with its last refills drained -- 0.4-0.9 ms that a run in progress never waits for); C4 is quoted on 8 GPUs --
# its total fits one, so the single-GPU line runs the total, and the per-GPU shard of the 8-GPU job next to it.
EXTRA_CONFIGS = [
    ("C2", dict(level="GoToLocal", total=65536, pixel=False, steps=1024, ref="BASELINE.json configs[1]; babyai/levels/iclr19_levels.py:105-124")),
    ("C3", dict(level="PickupLoc", total=262144, pixel=False, steps=512, ref="BASELINE.json configs[2]; iclr19_levels.py:494-515")),
    ("C4", dict(level="GoTo", total=1048576, pixel=False, steps=128, ref="BASELINE.json configs[3] (1 048 576 envs, here on ONE GPU); iclr19_levels.py:224-257")),
    ("C4-shard", dict(level="GoTo", total=131072, pixel=False, steps=576, ref="one GPU's share of configs[3] on 8 GPUs")),
    ("C5-encoded", dict(level="BossLevel", total=1048576, pixel=False, steps=128, ref="the headline's level with 7x7x3 encoded observations (k_step's own roofline)")),
    # the headline's own per-GPU workloads on 8 / 4 / 2 GPUs (`scaling: "strong"`: 1 048 576 envs in total), on this ONE GPU: the only
    # driver-timed evidence a scaling claim can have while no multi-GPU node runs the bench (`scaling_implied` in the line)
    ("C5-shard-131072", dict(level="BossLevel", total=131072, pixel=True, steps=64, horizon=384, of_gpus=8, ref="one GPU's share of configs[4] (the headline) on 8 GPUs")),
    ("C5-shard-262144", dict(level="BossLevel", total=262144, pixel=True, steps=32, horizon=256, of_gpus=4, ref="one GPU's share of the headline on 4 GPUs")),
    ("C5-shard-524288", dict(level="BossLevel", total=524288, pixel=True, steps=20, horizon=160, of_gpus=2, ref="one GPU's share of the headline on 2 GPUs")),
]


def main():
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # (what `import babyai_amd` sets: here before anything can initialise the HIP runtime)
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; start with --nproc-per-node == --gpus "
                         "(or run plain `python bench.py --gpus N`, which launches the ranks itself)" % (args.gpus, env_world))
    level, pixel, E, total_envs, scaling = resolve_workload(args)

    # the CPU legs' worker pool is forked BEFORE the GPU runtime and the process group exist (oracle/cpu_baseline.py
    # make_pool): it idles through the timed region and is only fed afterwards.  Rank 0 also runs the CPU baseline and
    # gets the larger share of the node's cores; the other ranks keep two workers each for their parity replay.
    pool = None
    pool_size = 0
    cpu_baseline = None
    if args.parity_envs or not args.no_cpu_baseline:
        from oracle import cpu_baseline            # outside the timed region: checker / reported baseline only
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        cores = cpu_baseline.usable_cores()
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            pool_size = max(2, cores - 2 * (local_world - 1))
        else:
            pool_size = 2
        pool = cpu_baseline.make_pool(pool_size)

    import time
    t_start = time.perf_counter()
    # The CPU baselines (the oracle port on the host cores: rank 0, the headline's bounded sample + a short one per other workload) need no GPU:
    # they run NOW, in the pool, while this process pages torch in and creates / seeds the headline batch (a cold box spends a minute there),
    # and are COMPLETE before the first warm-up step (joined in after_seed below) -- nothing of the oracle runs inside or beside a timed block.
    baselines = {"thread": None, "results": {}, "headline": None, "error": None}
    will_run_extras = (not args.no_extra_configs) and (env_world == 1 or args.extra_configs) and args.config is None and \
        args.envs is None and args.total_envs is None and not args.weak and args.level == "BossLevel" and not args.no_pixel
    if pool is not None and not args.no_cpu_baseline and int(os.environ.get("RANK", "0")) == 0:
        import threading

        def _baselines_before_the_gpu_work():
            try:
                cb = cpu_baseline.run(level, pixel, args.cpu_baseline_seconds, args.seed, args.action_seed, pool=pool, cores=pool_size)
                baselines["headline"] = cb
                baselines["results"][(level, pixel)] = cb
                if will_run_extras and args.extra_cpu_seconds > 0:
                    for _, c in EXTRA_CONFIGS:
                        key = (c["level"], c["pixel"])
                        if key not in baselines["results"]:
                            try:
                                baselines["results"][key] = cpu_baseline.run(c["level"], c["pixel"], args.extra_cpu_seconds, args.seed, args.action_seed, pool=pool, cores=pool_size)
                            except Exception as exc:
                                baselines["results"][key] = {"error": repr(exc)}
            except Exception as exc:          # the baseline is a reported number, never the product path
                baselines["error"] = repr(exc)

        baselines["thread"] = threading.Thread(target=_baselines_before_the_gpu_work, daemon=True)
        baselines["thread"].start()
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU path)")
    if not args.share_device and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", env_world)):
        raise SystemExit("bench.py: %d ranks on this node but only %d GPUs visible (one rank per GPU; test rigs: --share-device "
                         "--dist-backend gloo)" % (env_world, torch.cuda.device_count()))
    from babyai_amd import shard
    ranks = shard.Ranks.from_env(args.dist_backend, args.share_device)
    rank, world, dev = ranks.rank, ranks.world, ranks.device
    if world != args.gpus:
        raise SystemExit("bench.py: the live process group has %d ranks, --gpus asked for %d" % (world, args.gpus))
    group = ranks.describe()
    if group["allreduce_of_ones"] != world or (not args.share_device and group["distinct_devices"] != world):
        raise SystemExit("bench.py: process group check failed: %r" % (group,))

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    ranks.barrier()

    if args.own_stream:
        torch.cuda.set_stream(torch.cuda.Stream(dev))
    ctx = Ctx()
    ctx.torch, ctx.np, ctx.shard, ctx.ranks, ctx.dev, ctx.args, ctx.pool, ctx.cpu_baseline = torch, np, shard, ranks, dev, args, pool, cpu_baseline
    K, W = args.steps, args.warmup
    state = {"achievable": None}

    def after_seed():
        if baselines["thread"] is not None:          # the CPU legs end before any GPU step is timed (or warmed up)
            t_wait = time.perf_counter()
            baselines["thread"].join()
            baselines["waited_s"] = time.perf_counter() - t_wait
        state["achievable"] = achievable_bandwidth(torch, dev) if rank == 0 else None
        # clock ramp: a cold GPU spends its first second or so below its sustained clocks; keep it busy with an
        # untimed fill stream before the (short) warmup so the timed region sees steady-state clocks
        if args.prewarm_seconds > 0:
            scratch = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
            t_end = time.perf_counter() + args.prewarm_seconds
            while time.perf_counter() < t_end:
                for _ in range(20):
                    scratch.fill_(1)
                torch.cuda.synchronize()
            del scratch

    P = min(args.parity_envs, E)
    if world > 1 and P:
        P = max(min(128, E), P // world)      # every rank's host cores are shared by all ranks of the node
    digest = shard.EnvDigest(E, dev, 147) if args.dump_digest else None
    m = measure(ctx, level, pixel, E, total_envs, K, W, args.min_seconds, args.max_blocks, P, args.parity_pixel_envs, args.parity_budget,
                after_seed=after_seed, digest=digest, profile_tail=args.profile_tail)
    env = m["env"]
    achievable = state["achievable"]
    blocks, profiled, local_blocks = m["blocks"], m["profiled"], m["local_blocks"]
    kernel_ms, kernel_launches = m["kernel_ms"], m["kernel_launches"]
    S, want = m["S1"] + m["S2"], m["want"]

    bs = sorted(blocks)
    st = block_stats(blocks, K, E, world)
    med = st["median"]
    value = st["value_mean"]
    per_rank_ms = ranks.gather_objects(median(local_blocks) / K * 1e3 if local_blocks else None)
    dom, alg_bytes, bytes_per_step, dom_ms, achieved, ceiling_key = roofline_of(m)
    traffic = traffic_of(level, E, pixel, dom)
    prof_med = median(profiled) if profiled else None
    out = {
        "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": m["W"],
        "ms_per_step": st["mean"] / K * 1e3, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "BabyAI-%s-v0 %s obs, %d envs in total = %d per GPU x %d, random actions, auto-reset" % (
            level, "56x56x3 pixel (RGBImgPartialObsWrapper)" if pixel else "7x7x3 encoded", total_envs, E, world),
            "envs_per_gpu": E, "total_envs": total_envs, "resets_in_timed_region": m["resets"],
            "parallelism": "env-shards x%d, no collective" % world,
            "render_input": "the step's 147-byte encoding" if pixel else None,
            "actions": "counter-based (action_seed %d, step, global env index), uniform over 7" % args.action_seed},
        "rccl": dict(group, per_rank_ms_per_step=per_rank_ms,
                     per_rank_ms_per_step_min=min(per_rank_ms) if all(v is not None for v in per_rank_ms) else None,
                     per_rank_ms_per_step_max=max(per_rank_ms) if all(v is not None for v in per_rank_ms) else None,
                     launched_by="bench.py itself" if os.environ.get("BBAI_BENCH_SELF_LAUNCHED") else ("external launcher" if world > 1 else "single process")),
        "timing": {"blocks": len(blocks), "steps_per_block": K,
                   "block_ms": {"min": bs[0] * 1e3, "median": med * 1e3, "mean": st["mean"] * 1e3, "p90": st["p90"] * 1e3, "max": bs[-1] * 1e3},
                   "block_ms_list": [round(b * 1e3, 4) for b in blocks],
                   "timed_seconds": sum(blocks) + sum(profiled),
                   "value_from": "MEAN over the plain blocks: all their steps / all their seconds (sustained throughput)",
                   "value_mean": st["value_mean"], "value_median": st["value_median"], "mean_over_median": st["mean_over_median"],
                   "max_over_median": st["max_over_median"], "value_at_min": K * E * world / bs[0],
                   "value_at_max": K * E * world / bs[-1],
                   "loop": m["loop"],
                   "clock": "per block: opening barrier -> K steps -> this rank's device idle; the block = max over ranks; the closing barrier "
                            "and the max-reduce run after every rank's clock has stopped (barrier_ms); Python's cyclic garbage collector is off inside the loop (as timeit does)",
                   "barrier_ms": {"median": median(m["barrier_s"]) * 1e3, "max": max(m["barrier_s"]) * 1e3} if m["barrier_s"] else None,
                   "profiled_blocks": len(profiled),
                   "profiled_block_ms": {"min": min(profiled) * 1e3, "median": prof_med * 1e3, "max": max(profiled) * 1e3} if profiled else None,
                   "profiled_ms_per_step": prof_med / K * 1e3 if profiled else None,
                   "event_pairs_cost_us_per_step": (prof_med - med) / K * 1e6 if profiled else None,
                   "note": "plain and profiled blocks alternate; kernel_avg_ms are the profiled blocks' launches and add up to (at most) "
                           "profiled_ms_per_step"},
        "setup_ms": m["setup_ms"],
        "state_layout": m["state_layout"], "lookahead_period": m["lookahead_period"],
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic["bytes"] if traffic else None, "traffic_provenance": traffic,
                     "achievable": achievable, "frac_of_achievable": (achieved / achievable[ceiling_key]) if achievable else None,
                     "achievable_ceiling": ceiling_key,
                     "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms, "steps_per_launch": 1.0 if pixel else m["steps_per_k_step_launch"],
                     "whole_step_alg_GBs": value / world * bytes_per_step / 1e9,
                     "kernel_avg_ms": kernel_ms, "kernel_launches": kernel_launches, "measured_on": "rank 0"},
        "parity": None, "cpu_baseline": None, "configs": None, "gate_timeouts": m["gate_timeouts"],
        "build": {"commit": git_head(), "csrc_sha": csrc_sha()},
    }
    # the OPTIONAL gather of the encoded observations to rank 0 (north_star: "only an optional xGMI gather of obs to
    # rank 0"): timed on its own, after the timed region -- it is not part of a step and not part of `value`
    if world > 1:
        try:
            src = env.image if args.dist_backend == "nccl" else env.image.cpu()
            shard.gather_to_rank0(src, ranks.dist, via_all_gather=True)
            ranks.barrier()
            t0 = time.perf_counter()
            for _ in range(5):
                shard.gather_to_rank0(src, ranks.dist, via_all_gather=True)
            ranks.barrier()
            g_ms = ranks.max(time.perf_counter() - t0) / 5 * 1e3
            out["obs_gather"] = {"ms": g_ms, "bytes_per_peer": int(env.image.numel()), "backend": args.dist_backend,
                                 "GBs_into_rank0": env.image.numel() * (world - 1) / (g_ms * 1e-3) / 1e9,
                                 "note": "encoded obs of every shard -> rank 0 (as an all-gather: every rank receives); optional, outside the step path"}
        except Exception as exc:
            out["obs_gather"] = {"error": repr(exc)}
    torch.cuda.synchronize()
    if args.dump_digest:
        np.save("%s.rank%d.npy" % (args.dump_digest, rank), digest.numpy())
    env.close()
    m["env"] = None

    # ---- the other BASELINE configs, same loop, same GPU (world == 1 unless --extra-configs) -----------------------------------
    extras = []
    run_extras = will_run_extras
    if run_extras:
        for name, c in EXTRA_CONFIGS:
            if c["total"] % world:
                continue
            try:
                horizon = min(args.extra_parity_horizon, c.get("horizon", args.extra_parity_horizon))
                mc = measure(ctx, c["level"], c["pixel"], c["total"] // world, c["total"], c["steps"], max(16, horizon - c["steps"]), args.extra_seconds, 64,
                             args.extra_parity_envs, 16 if c["pixel"] else 0, args.extra_parity_budget, profile_tail=True)
                mc["env"].close()
                mc["env"] = None
                extras.append((name, c, mc))
            except Exception as exc:
                extras.append((name, c, {"error": repr(exc)}))
            torch.cuda.empty_cache()

    # ---- outside the timed region: the oracle re-derives what the taps recorded --------------------------------------
    exit_code = 0
    if m["gate_timeouts"]:
        exit_code = 5                       # a window gate gave up waiting for a refill (k_gate): the run is void, whatever it printed
    par = replay(ctx, m)
    if par is not None:
        bad = ranks.sum(par["mismatches"] or 0)
        broken = ranks.sum(1 if par["mismatches"] is None else 0)
        ids2 = m["ids2"]
        if rank == 0:
            par["mismatches_all_ranks"] = None if broken else bad       # never readable as "0 mismatches" when nothing was checked
            par["checker_errors_all_ranks"] = broken
            par["envs_all_ranks"] = len(ids2) * world
            par["env_selection"] = "scattered over each shard: both ends, wave / block boundaries, pseudo-random spread (shard.scattered_ids)"
            par["env_ids_rank0"] = {"min": min(ids2), "max": max(ids2), "count": len(ids2), "shard_envs": E}
            par["steps_checked"] = "all %d steps of the run (warmup, probe block and the %d timed blocks)" % (S, want)
            out["parity"] = par
        if broken:
            exit_code = 4
        elif bad:
            exit_code = 3                   # a fast kernel whose results differ from the oracle's is not done
    if extras:
        cfgs = {}
        for name, c, mc in extras:
            if "error" in mc:
                cfgs[name] = {"error": mc["error"]}
                exit_code = exit_code or 4
                continue
            if mc["gate_timeouts"]:
                exit_code = 5
            pc = replay(ctx, mc)
            bad = ranks.sum((pc or {}).get("mismatches") or 0)
            broken = ranks.sum(1 if (pc is not None and pc["mismatches"] is None) else 0)
            if broken:
                exit_code = exit_code or 4
            elif bad:
                exit_code = 3
            cdom, calg, cbps, cdom_ms, cach, ckey = roofline_of(mc)
            Kc, Ec = mc["K"], mc["E"]
            cst = block_stats(mc["blocks"], Kc, Ec, world)
            ctraffic = traffic_of(c["level"], Ec, c["pixel"], cdom)
            cfgs[name] = {
                "envs": Ec * world, "gate_timeouts": mc["gate_timeouts"],
                "workload": "BabyAI-%s-v0 %s obs, %d envs in total = %d per GPU x %d, random actions, auto-reset" % (
                    c["level"], "56x56x3 pixel (RGBImgPartialObsWrapper)" if c["pixel"] else "7x7x3 encoded", c["total"], Ec, world),
                "reference": c["ref"], "value": cst["value_mean"], "unit": "env-steps/s", "ms_per_step": cst["mean"] / Kc * 1e3,
                "value_from": "MEAN over the plain blocks, which are ONE contiguous rollout (sustained: every reset storm of the stretch is in it); the profiled blocks follow", "value_median": cst["value_median"], "ms_per_step_median": cst["median"] / Kc * 1e3,
                "mean_over_median": cst["mean_over_median"], "max_over_median": cst["max_over_median"],
                "steps_per_block": Kc, "blocks": len(mc["blocks"]), "timed_seconds": sum(mc["blocks"]) + sum(mc["profiled"]),
                "block_ms": {"min": cst["min"] * 1e3, "median": cst["median"] * 1e3, "mean": cst["mean"] * 1e3, "p90": cst["p90"] * 1e3, "max": cst["max"] * 1e3},
                "block_ms_list": [round(b * 1e3, 4) for b in mc["blocks"]],
                "kernel_avg_ms": mc["kernel_ms"], "state_layout": mc["state_layout"], "lookahead_period": mc["lookahead_period"],
                "loop": mc["loop"],
                "roofline": {"bound": "hbm", "kernel": cdom, "alg_bytes_per_launch": calg, "avg_launch_ms": cdom_ms, "steps_per_launch": 1.0 if c["pixel"] else mc["steps_per_k_step_launch"],
                             "achieved": cach, "unit": "GB/s",
                             "peak": HBM_PEAK_GBS, "frac": cach / HBM_PEAK_GBS,
                             "frac_of_achievable": (cach / achievable[ckey]) if achievable else None, "achievable_ceiling": ckey,
                             "traffic": ctraffic["bytes"] if ctraffic else None, "traffic_provenance": ctraffic,
                             "whole_step_alg_GBs": cst["value_mean"] / world * cbps / 1e9},
                "resets": mc["resets"], "resets_per_step": mc["resets"] / float(mc["S2"] or 1), "setup_ms": {k: v for k, v in mc["setup_ms"].items() if k != "note"},
                "parity": None if pc is None else {"envs": pc.get("envs"), "envs_all_steps": pc.get("envs_all_steps"), "steps": pc.get("steps"),
                                                   "envs_first_steps_only": pc.get("envs_first_steps_only"), "pixel_envs": pc.get("pixel_envs"),
                                                   "mismatches": None if broken else bad,
                                                   "first_mismatch": pc.get("first_mismatch"), "seconds": pc.get("seconds"), "error": pc.get("error"),
                                                   "env_selection": "scattered over the shard (shard.scattered_ids), every step"},
            }
        out["configs"] = cfgs
        # what the per-GPU shards of the headline imply for the N-GPU job (envs never interact and there is no collective on the step
        # path, so an N-GPU step lasts as long as its slowest shard): efficiency = t(1 048 576 envs on 1 GPU) / (N x t(1 048 576 / N envs on 1 GPU)).
        # IMPLIED from single-GPU runs of the shard sizes on this box -- not a measurement of N GPUs.
        if world == 1 and level == "BossLevel" and pixel and E == HEADLINE_ENVS:
            imp = {}
            for name, c, mc in extras:
                if c.get("of_gpus") and name in cfgs and "error" not in cfgs[name]:
                    n_g = c["of_gpus"]
                    t_shard = cfgs[name]["ms_per_step"]
                    imp[str(n_g)] = {"config": name, "envs_per_gpu": c["total"], "ms_per_step_shard": t_shard,
                                     "implied_value": HEADLINE_ENVS / (t_shard * 1e-3),
                                     "implied_efficiency": out["ms_per_step"] / (n_g * t_shard)}
            out["scaling_implied"] = {"basis": "single-GPU runs of each per-GPU shard size, same process, same box; no collective on the step path",
                                      "one_gpu_ms_per_step": out["ms_per_step"], "gpus": imp}
            c4, c4s = cfgs.get("C4"), cfgs.get("C4-shard")      # BASELINE.json configs[3] (encoded GoTo, 1 048 576 envs over 8 GPUs): the same inference
            if c4 and c4s and "error" not in c4 and "error" not in c4s:
                out["scaling_implied"]["C4"] = {"gpus": 8, "envs_per_gpu": c4s["envs"], "one_gpu_ms_per_step": c4["ms_per_step"],
                                                "ms_per_step_shard": c4s["ms_per_step"], "implied_value": 1048576 / (c4s["ms_per_step"] * 1e-3),
                                                "implied_efficiency": c4["ms_per_step"] / (8 * c4s["ms_per_step"])}
    ranks.barrier()
    if rank == 0 and not args.no_cpu_baseline:
        if baselines["thread"] is not None:
            baselines["thread"].join()
        if baselines["error"] or baselines["headline"] is None:
            out["cpu_baseline"] = {"error": baselines["error"] or "the CPU baseline did not run"}
        else:
            cb = dict(baselines["headline"])
            cb.update(cpu_baseline.reference_over_port(level, pixel))
            cb["when"] = "before the GPU work: in the worker pool while this process imported torch and created / seeded the batch; complete " \
                         "before the first warm-up step (waited %.1f s for it there)" % baselines.get("waited_s", 0.0)
            out["cpu_baseline"] = cb
            for name, c, mc in extras:          # the port on this box's host cores for every other workload (a short sample each), + the ratio on file
                if out["configs"] and name in out["configs"] and "error" not in out["configs"][name]:
                    out["configs"][name]["cpu_reference_over_port"] = cpu_baseline.reference_over_port(c["level"], c["pixel"]) or None
                    key = (c["level"], c["pixel"])
                    if key in baselines["results"]:
                        out["configs"][name]["cpu_baseline"] = {k: v for k, v in baselines["results"][key].items() if k in ("value", "unit", "cores", "kind", "sample", "single_core_value", "error")}
    if pool is not None:
        pool.terminate()
    out["wall_seconds"] = time.perf_counter() - t_start
    if rank == 0:
        # ONE stdout line, compact (compact_line); the full record goes to a side file -- BBAI_BENCH_LINE=full (tools/, tests) prints
        # the full record as the line instead
        full_path = write_full_record(out, args.full_out or os.path.join(ROOT, "gpurun_out", "bench_full.json"))
        if os.environ.get("BBAI_BENCH_LINE") == "full":
            print(json.dumps(out), flush=True)
        else:
            print(compact_line(out, os.path.relpath(full_path, ROOT) if full_path and full_path.startswith(ROOT) else full_path), flush=True)
    ranks.barrier()
    ranks.close()
    sys.exit(exit_code)


if __name__ == "__main__":
    main()
