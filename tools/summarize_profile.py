#!/usr/bin/env python3
"""Condense the rocprofv3 runs of one lease (gpurun_out/<tag>/, written by tools/lease.sh trace / pmc / sq / profcfg) into the tracked
evidence under profiles/<tag>/ and into profiles/pmc_latest.json, which bench.py quotes as `roofline.traffic`.

    python tools/summarize_profile.py <tag>                   the headline run (lease.sh trace + pmc [+ sq]: BossLevel pixels, 1 048 576 envs)
    python tools/summarize_profile.py <tag> --config C2       one BASELINE workload (lease.sh profcfg:C2)
    python tools/summarize_profile.py --line <bench.json>     one-line digest of a bench.py JSON line

profiles/pmc_latest.json holds ONE ENTRY PER WORKLOAD ("<level>:<envs per GPU>:<pixel|encoded>"), all taken on the kernel sources
whose hash it names (`csrc_sha`; an entry of other sources is dropped when a new one arrives): per kernel the steady-state (median
over launches) FETCH_SIZE / WRITE_SIZE in bytes per launch, the number of launches seen, and -- when the SQ pass ran -- the SQ
counters per launch.  FETCH_SIZE on gfx950 tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md; calibrated on k_render's known
byte counts in round 3): FETCH_SIZE_corrected = 2 x FETCH_SIZE.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")


def line_digest(path):
    d = json.loads(open(path).read().strip().splitlines()[-1])
    print(round(d["value"] / 1e6, 1), "M steps/s", round(d["ms_per_step"], 4), "ms/step (mean; median",
          round(d["timing"]["block_ms"]["median"] / d["steps"], 4), ") frac", round(d["roofline"]["frac"], 3),
          "of achievable", d["roofline"].get("frac_of_achievable"), "parity", (d.get("parity") or {}).get("mismatches_all_ranks"),
          "kernels", d["roofline"]["kernel_avg_ms"], "cpu", (d.get("cpu_baseline") or {}).get("value"), "wall", round(d.get("wall_seconds", 0), 1))
    for name, c in sorted((d.get("configs") or {}).items()):
        if "error" in c:
            print("  ", name, "ERROR", c["error"])
            continue
        print("   %-17s ms/step mean %.4f median %.4f (mean/med %.2f, max/med %.2f)  %s %.4f ms  frac %.3f  traffic %s  parity %s/%s envs %s" % (
            name, c["ms_per_step"], c["ms_per_step_median"], c["mean_over_median"], c["max_over_median"], c["roofline"]["kernel"],
            c["roofline"]["avg_launch_ms"], c["roofline"]["frac"], c["roofline"].get("traffic"), (c.get("parity") or {}).get("mismatches"),
            (c.get("parity") or {}).get("envs"), c["state_layout"].split(" ")[0]))
    if d.get("scaling_implied"):
        print("   implied scaling:", {k: round(v["implied_efficiency"], 3) for k, v in d["scaling_implied"]["gpus"].items()})


def csrc_sha():
    import hashlib
    d = os.path.join("babyai_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()[:16]


def git_head():
    import subprocess
    try:
        return subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def find_csv(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def trace_summary(stats_dir, out_path, what):
    """The --stats CSV reports MEANS, which the launches that overlap the seed-time generation skew: medians and percentiles of
    every kernel from the kernel trace of the same run."""
    kt = find_csv(stats_dir, "kernel_trace.csv")
    if not kt:
        return None
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(out_path, "w") as f:
        f.write("rocprofv3 --kernel-trace of `%s`, durations in microseconds per launch\n" % what)
        f.write("%-34s %6s %10s %10s %10s %10s %10s\n" % ("kernel", "calls", "median", "mean", "p10", "p90", "max"))
        for k, v in sorted(dur.items()):
            if not k.startswith("k_"):
                continue
            v = sorted(v)
            f.write("%-34s %6d %10.1f %10.1f %10.1f %10.1f %10.1f\n" % (k, len(v), v[len(v) // 2], sum(v) / len(v), v[len(v) // 10], v[len(v) * 9 // 10], v[-1]))
    return {k: sorted(v)[len(v) // 2] for k, v in dur.items() if k.startswith("k_")}


def counters(d, names):
    """{kernel: {counter: (median per launch, launches)}} from a counter_collection.csv"""
    path = find_csv(d, "counter_collection.csv")
    out = collections.defaultdict(dict)
    if not path:
        return out
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] in names:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if not k.startswith("k_"):
            continue
        for c, v in cs.items():
            v = sorted(v)
            out[k][c] = (v[len(v) // 2], len(v), v[0], v[-1])
    return out


SQ_NAMES = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
            "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--line":
        line_digest(sys.argv[2])
        return
    tag = sys.argv[1]
    cfg = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--config" else None
    src = os.path.join("gpurun_out", tag)
    dst = os.path.join("profiles", tag)
    os.makedirs(dst, exist_ok=True)
    sfx = "_" + cfg if cfg else ""
    name = cfg or "boss_pixel_1M"
    if cfg:
        bench_json = os.path.join(src, "bench_%s_under_rocprof.json" % cfg)
    else:
        bench_json = os.path.join(src, "bench_boss_pixel_1M_under_rocprof.json")
        for f in os.listdir(src):
            if f.endswith(".json") and f.startswith("bench_"):
                shutil.copy(os.path.join(src, f), dst)
    level, envs, pixel, what = "BossLevel", 1048576, True, "python bench.py --steps 20 --warmup 5"
    steps_per_launch = 1.0        # steps a k_step launch of this workload's loop takes (bbai_rollout: a look-ahead window's worth)
    try:
        d = json.loads(open(bench_json).read().strip().splitlines()[-1])
        steps_per_launch = float((d.get("roofline") or {}).get("steps_per_launch") or 1.0)
        shutil.copy(bench_json, os.path.join(dst, "bench_%s_under_rocprof.json" % name))
        w = d["config"]["workload"]
        level = w.split("-")[1]
        envs = int(d["config"]["envs_per_gpu"])
        pixel = "pixel" in w
        what = "python bench.py --config %s (%s)" % (cfg, w) if cfg else what
    except Exception as exc:
        print("no bench line under rocprof (%r): workload taken as %s:%d" % (exc, level, envs))
    key = "%s:%d:%s" % (level, envs, "pixel" if pixel else "encoded")

    stats_dir = os.path.join(src, "stats" + sfx)
    ks = find_csv(stats_dir, "kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(dst, "rocprofv3_kernel_stats_%s.csv" % name))
    med_us = trace_summary(stats_dir, os.path.join(dst, "rocprofv3_kernel_trace_summary_%s.txt" % name), what) or {}

    fetch = counters(os.path.join(src, "pmc_fetch" + sfx), ("FETCH_SIZE",))
    write = counters(os.path.join(src, "pmc_write" + sfx), ("WRITE_SIZE",))
    sq = counters(os.path.join(src, "pmc_sq" + sfx), SQ_NAMES)
    kernels, lines = {}, []
    for k in sorted(set(fetch) | set(write) | set(sq)):
        e = kernels.setdefault(k, {})
        if k in fetch:
            m, n, lo, hi = fetch[k]["FETCH_SIZE"]
            e["FETCH_SIZE"] = m * 1024.0            # (the counter is KB per dispatch)
            e["FETCH_SIZE_corrected"] = 2 * m * 1024.0
            e["launches"] = n
            lines.append("%-34s %-11s launches=%4d median=%14.1f KB  min=%14.1f  max=%14.1f" % (k, "FETCH_SIZE", n, m, lo, hi))
        if k in write:
            m, n, lo, hi = write[k]["WRITE_SIZE"]
            e["WRITE_SIZE"] = m * 1024.0
            e["launches"] = max(e.get("launches", 0), n)
            lines.append("%-34s %-11s launches=%4d median=%14.1f KB  min=%14.1f  max=%14.1f" % (k, "WRITE_SIZE", n, m, lo, hi))
        if k in sq:
            e["sq"] = {c: v[0] for c, v in sq[k].items()}
        if k in med_us:
            e["trace_median_us"] = med_us[k]
    note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel-trace only); counters are KB per dispatch; steady state = "
            "median over launches (the first k_pregen / k_consume launches cover all envs).  Correction per MI355X_MICROARCH.md (gfx950: FETCH_SIZE "
            "tallies 128-byte requests at half their size): FETCH_SIZE_corrected = 2 x FETCH_SIZE, calibrated on k_render (reads envs x 147 B, writes "
            "envs x 9408 B exactly); WRITE_SIZE needs none.")
    if lines:
        with open(os.path.join(dst, "rocprofv3_pmc_hbm_%s.txt" % name), "w") as f:
            f.write("workload %s (%s)\n%s\n\n%s\n" % (key, what, note, "\n".join(lines)))
    if sq:
        with open(os.path.join(dst, "rocprofv3_pmc_sq_%s.txt" % name), "w") as f:
            f.write("workload %s (%s): SQ counters, median per launch\n" % (key, what))
            for k in sorted(sq):
                c = {n: v[0] for n, v in sq[k].items()}
                wc = c.get("SQ_WAVE_CYCLES") or 1.0
                f.write("%-34s launches=%4d %s\n" % (k, max(v[1] for v in sq[k].values()), json.dumps({n: round(v) for n, v in c.items()})))
                f.write("%-34s   waiting %.0f %% of wave cycles, issuing %.0f %%, VALU %.0f %% of the issuing cycles%s\n" % (
                    "", 100 * c.get("SQ_WAIT_ANY", 0) / wc, 100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                    100 * c.get("SQ_ACTIVE_INST_VALU", 0) / (c.get("SQ_ACTIVE_INST_ANY") or 1.0),
                    (", %.0f VALU instructions per wave-launch unit" % c["SQ_INSTS_VALU"]) if "SQ_INSTS_VALU" in c else ""))

    # merge into profiles/pmc_latest.json
    latest_path = os.path.join("profiles", "pmc_latest.json")
    sha = csrc_sha()
    try:
        latest = json.load(open(latest_path))
        if latest.get("csrc_sha") != sha or "workloads" not in latest:
            latest = None
    except Exception:
        latest = None
    if latest is None:
        latest = {"unit": "bytes per launch", "csrc_sha": sha, "note": note, "workloads": {}}
    latest["commit"] = git_head()
    latest["profile_tag"] = tag
    if kernels:
        latest["workloads"][key] = {"what": what, "kernels": kernels, "k_step_steps_per_launch": steps_per_launch}
        json.dump(latest, open(latest_path, "w"), indent=1, sort_keys=True)
    print("workload", key)
    print("\n".join(lines))
    for k in sorted(sq):
        c = {n: v[0] for n, v in sq[k].items()}
        print("SQ %-30s wait %.0f %% issue %.0f %% valu/issue %.0f %%" % (k, 100 * c.get("SQ_WAIT_ANY", 0) / (c.get("SQ_WAVE_CYCLES") or 1),
                                                                      100 * c.get("SQ_ACTIVE_INST_ANY", 0) / (c.get("SQ_WAVE_CYCLES") or 1),
                                                                      100 * c.get("SQ_ACTIVE_INST_VALU", 0) / (c.get("SQ_ACTIVE_INST_ANY") or 1)))


if __name__ == "__main__":
    main()
