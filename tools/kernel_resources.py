#!/usr/bin/env python3
"""Register / LDS / scratch figures of every kernel of the engine, from the compiler itself.

    python tools/kernel_resources.py [--log build.log] [--json out.json] [--markdown]

Without --log the engine is compiled once more with `-Rpass-analysis=kernel-resource-usage` (the flags of
__graft_entry__.build(); about two minutes, no GPU needed) into a scratch file.  `__graft_entry__.build()` runs this on every
real build and keeps the result next to the library (babyai_amd/kernel_resources.json); DESIGN.md section 4 quotes that file and
tests/test_kernel_resources.py checks the quotes against it -- the figures in the documentation are the binary's, not a memory of
an earlier build.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = {
    "VGPRs": "vgprs", "AGPRs": "agprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
    "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "SGPRs Spill": "sgpr_spills", "VGPRs Spill": "vgpr_spills",
    "LDS Size [bytes/block]": "lds_bytes_per_block", "Dynamic Stack": "dynamic_stack",
}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return [short_name(o) for o in out]


def short_name(sig):
    """`void k_step<true, 1>(bbai::LevelCfg, ...)` -> `k_step<true, 1>`"""
    sig = sig.strip()
    if sig.startswith("void "):
        sig = sig[5:]
    depth = 0
    for i, ch in enumerate(sig):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return sig[:i]
    return sig


def parse(log_text):
    kernels, cur = [], None
    for line in log_text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"mangled": m.group(1)}
            kernels.append(cur)
            continue
        if cur is None:
            continue
        m = re.search(r"remark:\s+(.+?):\s+(\d+|True|False)\s+\[-Rpass-analysis", line)
        if m and m.group(1) in FIELDS:
            v = m.group(2)
            cur[FIELDS[m.group(1)]] = int(v) if v.isdigit() else (v == "True")
    names = demangle([k["mangled"] for k in kernels])
    out = {}
    for k, name in zip(kernels, names):
        k.pop("mangled")
        out[name] = k
    return out


def compile_log():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    with tempfile.TemporaryDirectory() as d:
        cmd = g.hip_command(os.path.join(d, "probe.so")) + ["-Rpass-analysis=kernel-resource-usage"]
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
        if p.returncode:
            raise SystemExit(p.stderr[-4000:])
        return p.stderr


def markdown(res, only=None):
    rows = ["| kernel | VGPRs | SGPRs | SGPR spills | scratch B/lane | LDS B/block | waves / SIMD |", "|---|---|---|---|---|---|---|"]
    for name in sorted(res):
        if only and not any(name.startswith(o) for o in only):
            continue
        r = res[name]
        rows.append("| `%s` | %d | %d | %d | %d | %d | %d |" % (name, r.get("vgprs", 0), r.get("sgprs", 0), r.get("sgpr_spills", 0),
                                                             r.get("scratch_bytes_per_lane", 0), r.get("lds_bytes_per_block", 0),
                                                             r.get("occupancy_waves_per_simd", 0)))
    return "\n".join(rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log")
    ap.add_argument("--json")
    ap.add_argument("--markdown", action="store_true")
    ap.add_argument("--only", nargs="*", default=None, help="kernel name prefixes")
    a = ap.parse_args()
    text = open(a.log).read() if a.log else compile_log()
    res = parse(text)
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
    if a.markdown:
        print(markdown(res, a.only))
    elif not a.json:
        print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
