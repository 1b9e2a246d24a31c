#!/bin/bash
# SQ / SQC counter passes over the level generator alone (k_pregen filling the look-ahead rings of 262144 BossLevel envs):
# what does a generator wave spend its cycles on?  usage: tools/gpu_pmc_gen.sh <round-tag> [lib.so]
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG/gen_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export BBAI_LOOKAHEAD=2
cat > /tmp/gen_only.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from babyai_amd.engine import BatchedBabyAIEnv
env = BatchedBabyAIEnv("BabyAI-BossLevel-v0", 262144, device="cuda:0")
env.seed(0); env.seed(7); torch.cuda.synchronize(); env.close()
PY
pass() {  # name lib counters...
  name=$1; lib=$2; shift 2
  if [ -n "$lib" ]; then export BBAI_ENGINE_LIB=$lib; else unset BBAI_ENGINE_LIB; fi
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o gen -- python /tmp/gen_only.py > $OUT/$name.log 2>&1
}
pass p1 "" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES
pass p2 "" SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_INST_LDS
pass p3 "" SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC
if [ -n "$2" ]; then pass p3_old "$2" SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC; fi
python - <<PY > $OUT/summary.txt
import csv, glob, collections, os
for p in sorted(glob.glob("$OUT/p*")):
    if not os.path.isdir(p): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("k_pregen") or k.startswith("void k_pregen"):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(os.path.basename(p), k, {c: [round(x) for x in v] for c, v in d.items()})
PY
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +5M -delete
