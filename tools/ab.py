#!/usr/bin/env python3
"""tools/ab.py -- alternate engine settings (include/bbai.h bbai_set_option) inside ONE process on ONE box.

Boxes of the pool differ by a few per cent on the pure store stream, and so do two runs on one box: settings are only
compared when they alternate inside one process.  One batch is created and stepped with random actions; for every
repetition, for every setting: apply the options, a few untimed steps, then `--blocks` blocks of `--steps` steps, plain and
profiled blocks alternating as in bench.py (the per-kernel times come from the profiled ones).  One JSON line per
(repetition, setting); the last line ranks the settings by their median over repetitions.

    python tools/ab.py --level BossLevel --envs 1048576 --pixel --settings render_queue=0 render_queue=1 render_queue=2
    python tools/ab.py --level GoToLocal --envs 65536 --steps 256 --settings consume_fused=0 consume_fused=1

A setting is a comma-separated list of name=value pairs ("render_queue=2,render_queue_bpc=1").  Names starting with "env:"
are environment variables and need a fresh batch: they are applied by re-creating the env (slower; kept for knobs that are
read at bbai_create only, e.g. env:BBAI_LOOKAHEAD=8).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_setting(text):
    opts, envs = [], {}
    for kv in text.split(","):
        if not kv:
            continue
        k, v = kv.split("=")
        if k.startswith("env:") or k.startswith("env."):      # ("env." for tools/lease.sh, whose job syntax owns the colon)
            envs[k[4:]] = v
        else:
            opts.append((k, int(v)))
    return opts, envs


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", default="BossLevel")
    ap.add_argument("--envs", type=int, default=1048576)
    ap.add_argument("--pixel", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=10, help="timed blocks per setting per repetition (half plain, half profiled)")
    ap.add_argument("--settle", type=int, default=4, help="untimed steps after a switch")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--settings", nargs="+", required=True)
    ap.add_argument("--base", default="", help="options applied before every setting (so that a setting only names what it changes)")
    ap.add_argument("--tag", default=None)
    args = ap.parse_args()

    import torch
    import __graft_entry__
    __graft_entry__.build()
    from babyai_amd.engine import BatchedBabyAIEnv
    from babyai_amd.action_stream import actions_torch
    dev = torch.device("cuda", 0)
    settings = [(s,) + parse_setting(s) for s in args.settings]
    need_fresh = any(envs for _, _, envs in settings)
    E, K = args.envs, args.steps
    pool = actions_torch(1234, 0, 64, 0, E, dev)            # 64 rows of i.i.d. actions, cycled
    state = {"env": None, "envs": None, "t": 0}

    def get_env(envs):
        if state["env"] is not None and (not need_fresh or state["envs"] == envs):
            return state["env"]
        if state["env"] is not None:
            state["env"].close()
        for k in list(os.environ):
            if k.startswith("BBAI_") and k not in ("BBAI_ENGINE_LIB",):
                del os.environ[k]
        os.environ.update(envs)
        env = BatchedBabyAIEnv("BabyAI-%s-v0" % args.level, E, device=dev, pixel=args.pixel, seeds=args.seed)
        env.reset()
        for _ in range(16):
            step(env)
        state["env"], state["envs"] = env, dict(envs)
        return env

    def step(env):
        env.step(pool[state["t"] % 64])
        state["t"] += 1

    results = {}
    for rep in range(args.reps):
        for name, opts, envs in settings:
            env = get_env(envs)
            for k, v in parse_setting(args.base)[0] + opts:
                env.set_option(k, v)
            for _ in range(args.settle):
                step(env)
            env.profile(True)
            env.profile_pause()
            plain, prof = [], []
            r0 = env.reset_count()
            for b in range(args.blocks):
                if b % 2:
                    env.profile_resume()
                else:
                    env.profile_pause()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(K):
                    step(env)
                torch.cuda.synchronize()
                (prof if b % 2 else plain).append((time.perf_counter() - t0) / K * 1e3)
            env.profile_pause()
            km = {k: (round(v[0], 5) if v[0] is not None else None) for k, v in env.profile_read().items()}
            env.profile(False)
            line = {"tag": args.tag, "rep": rep, "setting": name, "level": args.level, "envs": E, "pixel": args.pixel,
                    "ms_per_step": round(median(plain), 5), "ms_per_step_min": round(min(plain), 5),
                    "profiled_ms_per_step": round(median(prof), 5) if prof else None, "kernel_avg_ms": km,
                    "resets_per_step": (env.reset_count() - r0) / (args.blocks * K)}
            print(json.dumps(line), flush=True)
            results.setdefault(name, []).append(line)
    rank = sorted(((median([l["ms_per_step"] for l in ls]), n) for n, ls in results.items()))
    print(json.dumps({"tag": args.tag, "ranking_ms_per_step": [[n, round(v, 5)] for v, n in rank],
                      "kernels_median": {n: {k: median([l["kernel_avg_ms"][k] for l in ls if l["kernel_avg_ms"].get(k) is not None] or [0])
                                             for k in ("k_step", "k_consume", "k_render")} for n, ls in results.items()}}), flush=True)


if __name__ == "__main__":
    main()
