#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched BabyAI hot path on MI355X.

One "step" = one pass of the hot path over one batch: every env of the shard applies one action (transition + verifier),
finished envs are regenerated on the device (auto-reset) and the observation is written (7x7x3 encoding, plus the
56x56x3 pixel render for the default BossLevel workload = BASELINE.json configs[4] on one GPU).  Actions are synthetic,
i.i.d. uniform over the 7 actions from a counter-based generator keyed on (bench seed, step, global env index)
(babyai_amd/action_stream.py), resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

What one run proves about itself (all in the one JSON line rank 0 prints):
  * timing      W warmup steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize and max-reduced
                over ranks, repeated until >= --min-seconds of timed work: `ms_per_step` / `value` come from the MEDIAN
                block, `timing` holds min / median / max (box-to-box and run-to-run variance is ~10 %).
  * roofline    the dominant kernel's algorithmic bytes / its HIP-event time on the launch stream, against the 8 TB/s
                spec peak AND against what a plain 1-GiB fill / copy reaches on this box in this process
                (`achievable`, `frac_of_achievable`).  `traffic` = HBM bytes per launch from the committed rocprofv3 PMC
                passes, only while the kernel sources still hash to what was profiled (else null).
  * parity      the outputs of the shard's first 1024 envs at EVERY timed step (image, direction, f64 reward bits, done;
                pixels of the first 64) are tapped inside the timed region and re-derived afterwards by the CPU oracle
                from the seeds and the action stream: `parity.mismatches` must be 0.
  * cpu_baseline  the oracle on the usable host cores over the same seeds and action stream (a reported baseline).

Envs shard embarrassingly (babyai_amd/shard.py): rank r owns global envs [r*E, (r+1)*E) with seeds base + global index;
no collective on the step path.  scaling = weak (E envs per GPU fixed).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak (MI355X_MICROARCH.md)

# BASELINE.json configs (per-GPU env counts; C4/C5 are quoted on 8 GPUs with 131072 envs each, the default
# bench runs C5's level and obs mode with all 1 048 576 envs on ONE GPU, which is the headline metric's shape)
CONFIGS = {
    "C2": dict(level="GoToLocal", envs=65536, pixel=False),
    "C3": dict(level="PickupLoc", envs=262144, pixel=False),
    "C4": dict(level="GoTo", envs=131072, pixel=False),
    "C5": dict(level="BossLevel", envs=131072, pixel=True),
    "C5-1gpu": dict(level="BossLevel", envs=1048576, pixel=True),
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--envs", type=int, default=1048576, help="envs per GPU")
    ap.add_argument("--level", default="BossLevel")
    ap.add_argument("--no-pixel", action="store_true")
    ap.add_argument("--seed", type=int, default=0, help="env i of the whole job is seeded with seed + i")
    ap.add_argument("--action-seed", type=int, default=1234)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="a BASELINE.json config by name (overrides --level/--envs/--no-pixel); default = C5 on one GPU")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="repeat the K-step block until this much timed work")
    ap.add_argument("--max-blocks", type=int, default=64)
    ap.add_argument("--parity-envs", type=int, default=1024, help="first envs of every shard checked against the oracle (0 = off)")
    ap.add_argument("--parity-pixel-envs", type=int, default=64)
    ap.add_argument("--parity-budget", type=int, default=600000,
                    help="oracle env-steps the parity check may cost: long runs check every step of fewer envs")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm-seconds", type=float, default=1.0, help="untimed GPU activity before the warmup steps")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, one rank per GPU) | gloo (test rigs: ranks may share a GPU)")
    ap.add_argument("--share-device", action="store_true", help="test rigs only: every rank uses cuda:0")
    ap.add_argument("--dump-digest", default=None, help="write per-env output digests of this rank to <prefix>.rank<r>.npy")
    args = ap.parse_args()
    if args.config:
        cfgsel = CONFIGS[args.config]
        args.level, args.envs, args.no_pixel = cfgsel["level"], cfgsel["envs"], not cfgsel["pixel"]

    # the CPU legs' worker pool is forked BEFORE the GPU runtime and the process group exist (oracle/cpu_baseline.py
    # make_pool): it idles through the timed region and is only fed afterwards
    pool = None
    if args.parity_envs or not args.no_cpu_baseline:
        from oracle import cpu_baseline            # outside the timed region: checker / reported baseline only
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
        pool = cpu_baseline.make_pool(max(2, cpu_baseline.usable_cores() // local_world))

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU path)")
    from babyai_amd import shard
    from babyai_amd.action_stream import actions_torch
    ranks = shard.Ranks.from_env(args.dist_backend, args.share_device)
    rank, world, dev = ranks.rank, ranks.world, ranks.device
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    ranks.barrier()
    from babyai_amd.engine import BatchedBabyAIEnv

    pixel = not args.no_pixel
    E = args.envs
    total_envs = E * world
    first, count = shard.shard_range(total_envs, world, rank)
    assert count == E
    env = BatchedBabyAIEnv("BabyAI-%s-v0" % args.level, E, device=dev, pixel=pixel)
    env.seed(shard.shard_seeds(args.seed, total_envs, world, rank))
    K, W = args.steps, args.warmup

    achievable = achievable_bandwidth(torch, dev) if rank == 0 else None

    # clock ramp: a cold GPU spends its first second or so below its sustained clocks; keep it busy with an
    # untimed fill stream before the (short) warmup so the timed region sees steady-state clocks
    if args.prewarm_seconds > 0:
        import time
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
    PP = min(args.parity_pixel_envs, P) if pixel else 0
    digest = shard.EnvDigest(E, dev, 147) if args.dump_digest else None

    def make_log(steps, with_initial, p, pp):
        if not p:
            return None
        extra = 1 if with_initial else 0
        lg = {"image": torch.zeros((steps + extra, p, 7, 7, 3), dtype=torch.uint8, device=dev),
              "direction": torch.zeros((steps + extra, p), dtype=torch.uint8, device=dev),
              "reward64": torch.zeros((steps, p), dtype=torch.float64, device=dev),
              "done": torch.zeros((steps, p), dtype=torch.uint8, device=dev)}
        if pp:
            lg["pixels"] = torch.zeros((steps + extra, pp, 56, 56, 3), dtype=torch.uint8, device=dev)
        return lg

    def tap(lg, obs_row, row):
        """the parity tap: ONE small launch inside the timed region (include/bbai.h bbai_tap)"""
        env.tap(lg["image"][obs_row], lg["direction"][obs_row], lg["reward64"][row], lg["done"][row],
                lg["pixels"][obs_row] if "pixels" in lg else None)

    # phase 1: reset, W warmup steps and ONE K-step block; its time decides how many further blocks make --min-seconds
    S1 = W + K
    actions1 = actions_torch(args.action_seed, 0, S1, first, E, dev)        # resident before the timed region
    log1 = make_log(S1, True, P, PP)
    env.reset()
    if log1 is not None:
        log1["image"][0].copy_(env.image[:P])
        log1["direction"][0].copy_(env.direction[:P])
        if PP:
            log1["pixels"][0].copy_(env.pixels[:PP])

    def after1(t):
        if log1 is not None:
            tap(log1, t + 1, t)
        if digest is not None:
            digest.update(env.image, env.direction, env.reward64, env.done)

    torch.cuda.synchronize()
    blocks = shard.timed_blocks(env, actions1, W, K, 1, ranks, after1)
    want = int(min(args.max_blocks, max(0, -(-args.min_seconds // blocks[0]))))
    want = int(ranks.max(want))
    # phase 2: `want` more blocks of exactly K steps; when there are any, the phase-1 block was only the probe
    S2 = want * K
    if P and (S1 + S2) * P > args.parity_budget:      # long run: every step of fewer envs
        P = max(min(16, P), args.parity_budget // (S1 + S2))
        PP = min(PP, P)
    log2 = None
    resets0 = env.reset_count()
    env.profile(True)              # per-kernel HIP event pairs on the launch stream (include/bbai.h bbai_profile)
    if want:
        actions2 = actions_torch(args.action_seed, S1, S1 + S2, first, E, dev)
        log2 = make_log(S2, False, P, PP)

        def after2(t):
            if log2 is not None:
                tap(log2, t, t)
            if digest is not None:
                digest.update(env.image, env.direction, env.reward64, env.done)

        torch.cuda.synchronize()
        # the per-kernel event pairs ride along in the first of these blocks only (two event records per launch are
        # not free at 65 536 envs); `value` comes from the median block
        blocks = shard.timed_blocks(env, actions2, 0, K, want, ranks, after2,
                                    after_block=lambda i: env.profile_pause() if i == 0 else None)
    resets = ranks.sum(env.reset_count() - resets0)
    S = S1 + S2

    # per-kernel durations: HIP event pairs recorded by the engine on the launch stream around each launch (bbai_profile)
    if not want:                    # single-block run: nothing was bracketed; time a few untimed steps
        env.profile(True)
        for t in range(4):
            env.step(actions1[t])
        torch.cuda.synchronize()
    kernel_ms = {k: v[0] for k, v in env.profile_read().items() if v[0] is not None}
    kernel_launches = {k: v[1] for k, v in env.profile_read().items() if v[0] is not None}
    env.profile(False)

    bs = sorted(blocks)
    med = bs[len(bs) // 2] if len(bs) % 2 else 0.5 * (bs[len(bs) // 2 - 1] + bs[len(bs) // 2])
    value = K * E * world / med
    if pixel:
        dom, alg_bytes = "k_render", E * (147 + 9408)          # reads the encoding, writes the pixels
        dom_ms = kernel_ms["k_render"]
        bytes_per_step = 9496
        ceiling_key = "fill_GBs"                                # a pure store stream
    else:
        dom, alg_bytes = "k_step", E * 235
        dom_ms = kernel_ms["k_step"]
        bytes_per_step = 235
        ceiling_key = "copy_GBs"                                # reads and writes mixed
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    # HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, separate
    # runs; tools/gpu_profile.sh -> tools/summarize_profile.py -> profiles/pmc_latest.json).  Counters cannot be
    # collected inside this process: the committed summary is quoted with its provenance, and only while the kernel
    # sources still hash to what was profiled on this exact workload.
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        key = next((k for k in pmc["kernels"] if k == dom or k.startswith(dom + "<")), None)      # k_render is a template
        if pmc.get("level") == args.level and pmc.get("envs") == E and key:
            kk = pmc["kernels"][key]
            fetch = kk.get("FETCH_SIZE_corrected", 2 * kk["FETCH_SIZE"])      # gfx950: FETCH_SIZE tallies 128-B requests as 64 B
            traffic = {"bytes": fetch + kk["WRITE_SIZE"], "fetch_corrected": fetch, "fetch_raw": kk["FETCH_SIZE"], "write": kk["WRITE_SIZE"],
                       "correction": "FETCH_SIZE x 2 (MI355X_MICROARCH.md, calibrated on k_render's known byte counts)",
                       "source": "profiles/pmc_latest.json", "commit": pmc.get("commit"), "csrc_sha": pmc.get("csrc_sha"),
                       "current": pmc.get("csrc_sha") == csrc_sha()}
            if not traffic["current"]:
                traffic["bytes"] = None         # kernels changed since the counters were taken
    except Exception:
        traffic = None
    out = {
        "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": med / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "BabyAI-%s-v0 %s obs, %d envs/GPU, random actions, auto-reset" % (
            args.level, "56x56x3 pixel (RGBImgPartialObsWrapper)" if pixel else "7x7x3 encoded", E),
            "envs_per_gpu": E, "total_envs": total_envs, "resets_in_timed_region": resets,
            "parallelism": "env-shards x%d, no collective" % world,
            "actions": "counter-based (action_seed %d, step, global env index), uniform over 7" % args.action_seed},
        "timing": {"blocks": len(blocks), "steps_per_block": K, "block_ms": {"min": bs[0] * 1e3, "median": med * 1e3, "max": bs[-1] * 1e3},
                   "timed_seconds": sum(blocks), "value_from": "median block", "value_at_min": K * E * world / bs[0],
                   "value_at_max": K * E * world / bs[-1]},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic["bytes"] if traffic else None, "traffic_provenance": traffic,
                     "achievable": achievable, "frac_of_achievable": (achieved / achievable[ceiling_key]) if achievable else None,
                     "achievable_ceiling": ceiling_key,
                     "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms,
                     "whole_step_alg_GBs": value / world * bytes_per_step / 1e9,
                     "kernel_avg_ms": kernel_ms, "kernel_launches": kernel_launches},
        "parity": None, "cpu_baseline": None,
        "build": {"commit": git_head(), "csrc_sha": csrc_sha()},
    }
    # the OPTIONAL gather of the encoded observations to rank 0 (north_star: "only an optional xGMI gather of obs to
    # rank 0"): timed on its own, after the timed region -- it is not part of a step and not part of `value`
    if world > 1:
        try:
            src = env.image if args.dist_backend == "nccl" else env.image.cpu()
            shard.gather_to_rank0(src, ranks.dist, via_all_gather=True)
            ranks.barrier()
            import time
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
    # ---- outside the timed region: the oracle re-derives what the tap recorded --------------------------------------
    if log1 is not None:
        host = {k: np.concatenate([log1[k][:, :(PP if k == "pixels" else P)].cpu().numpy()]
                                  + ([log2[k].cpu().numpy()] if log2 is not None else [])) for k in log1 if (k != "pixels" or PP)}
        try:
            par = cpu_baseline.parity_replay(args.level, host, args.seed, args.action_seed, first, PP, pool=pool)
        except Exception as exc:
            par = {"error": repr(exc), "mismatches": None}          # the CHECKER broke: reported, not a parity verdict
        bad = ranks.sum(par["mismatches"] or 0)
        broken = ranks.sum(1 if par["mismatches"] is None else 0)
        if rank == 0:
            par["mismatches_all_ranks"] = bad
            par["checker_errors_all_ranks"] = broken
            par["envs_all_ranks"] = P * world
            par["steps_checked"] = "all %d steps of the run (warmup, probe block and the %d timed blocks)" % (S, len(blocks))
            out["parity"] = par
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline.run(args.level, pixel, args.cpu_baseline_seconds, args.seed, args.action_seed, pool=pool)
        except Exception as exc:      # the baseline is a reported number, never the product path
            out["cpu_baseline"] = {"error": repr(exc)}
    if pool is not None:
        pool.terminate()
    if rank == 0:
        print(json.dumps(out))
    env.close()
    ranks.close()
    if rank == 0 and out["parity"] and (out["parity"].get("mismatches_all_ranks") or 0) != 0:
        sys.exit(3)                     # a fast kernel whose results differ from the oracle's is not done


if __name__ == "__main__":
    main()
