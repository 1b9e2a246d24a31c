#!/bin/bash
# Round 3, lease C: full GPU suite on the fused tile plane + grouped generator, render A/B (fused plane vs encoding, launch
# shapes) inside the real loop, step-kernel wave priority A/B on the reset-heavy configs, k_step layout microbenchmarks.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03c
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict($1, ms_per_step=d['ms_per_step'], value=d['value'], profiled_ms_per_step=d['timing']['profiled_ms_per_step'], parity=(d['parity'] or {}).get('mismatches_all_ranks'), kernels=d['roofline']['kernel_avg_ms'], fill_GBs=d['roofline']['achievable']['fill_GBs'])))"; }
# render: fused plane vs encoding x launch shape, alternated twice (1M envs = the headline)
for rep in 1 2; do
  for fused in 1 0; do
    for shape in "0 0" "2 512" "4 512" "8 1024"; do
      set -- $shape
      BBAI_RENDER_FUSED=$fused BBAI_RENDER_GROUP=$1 BBAI_RENDER_TPB=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-envs 128 --min-seconds 0.6 2>>$OUT/render_ab.err | line "fused=$fused, group=$1, tpb=$2, envs=1048576" >> $OUT/render_fused_ab_1M.jsonl
    done
  done
done
cat $OUT/render_fused_ab_1M.jsonl
for rep in 1 2; do
  for fused in 1 0; do
    for shape in "0 0" "4 512" "8 1024"; do
      set -- $shape
      BBAI_RENDER_FUSED=$fused BBAI_RENDER_GROUP=$1 BBAI_RENDER_TPB=$2 timeout 300 python bench.py --config C5-shard --steps 64 --warmup 8 --no-cpu-baseline --parity-envs 128 --min-seconds 0.5 2>>$OUT/render_ab.err | line "fused=$fused, group=$1, tpb=$2, envs=131072" >> $OUT/render_fused_ab_131072.jsonl
    done
  done
done
cat $OUT/render_fused_ab_131072.jsonl
# wave priority of the step-path kernels
for rep in 1 2; do
  for prio in 0 1; do
    for cfg in C2 C3 C4-shard; do
      BBAI_STEP_PRIO=$prio timeout 300 python bench.py --config $cfg --steps 256 --warmup 16 --no-cpu-baseline --parity-envs 128 --min-seconds 0.5 2>>$OUT/prio.err | line "step_prio=$prio, config='$cfg'" >> $OUT/step_prio_ab.jsonl
    done
    BBAI_STEP_PRIO=$prio timeout 300 python bench.py --no-pixel --steps 64 --warmup 8 --no-cpu-baseline --parity-envs 128 --min-seconds 0.5 2>>$OUT/prio.err | line "step_prio=$prio, config='boss_encoded_1M'" >> $OUT/step_prio_ab.jsonl
  done
done
cat $OUT/step_prio_ab.jsonl
timeout 300 tools/ubench_gather > $OUT/ubench_gather.jsonl 2> $OUT/ubench_gather.err
grep -E "envtile|ztile|vline|\"chain\"|\"soa\"" $OUT/ubench_gather.jsonl
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/ubench_pmc_$c -o ug -- $REPO/tools/ubench_gather > $OUT/ubench_pmc_$c.log 2>&1
done
cd $REPO
python - <<PY
import csv, glob, collections, json
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = collections.defaultdict(list)
    for f in glob.glob("$OUT/ubench_pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                rows[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in rows.items():
        res.setdefault(k, {})[c + "_KB_median"] = sorted(v)[len(v) // 2]
        res[k]["launches"] = len(v)
json.dump(res, open("$OUT/ubench_gather_counters.json", "w"), indent=1)
for k, v in res.items():
    print(k[:80], v)
PY
find $OUT -name "*kernel_trace.csv" -size +5M -delete
find $OUT -name "*.csv" -size +20M -delete
