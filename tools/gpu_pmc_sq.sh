#!/bin/bash
# SQ counter pass over the headline bench (own run, --kernel-trace only): where do the waves of k_render / k_step spend
# their cycles?  usage: tools/gpu_pmc_sq.sh <round-tag>
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU \
  --kernel-trace --output-format csv -d $OUT/pmc_sq -o boss -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/rocprof_sq.log 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_sq/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in f:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if k in ("k_render", "k_step", "k_consume"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sorted(v)[len(v) // 2]) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
