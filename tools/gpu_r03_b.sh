#!/bin/bash
# Round 3, lease B: the lane-group generator (k_pregen<KIND, G>): device == host build exhaustively, generator rate per
# group width, the reset-heavy configs per group width, the new k_step layout microbenchmarks (+ their HBM counters).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03b
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
for g in 16 32 64; do
  BBAI_PREGEN_GROUP=$g timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_termination_guard.py -m gpu -x -q -k "generator_equals_host or golden_trace or every_registered_level or termination or reseed or very_short" > $OUT/pytest_gen_g$g.log 2>&1; echo "G=$g pytest rc=$?" | tee -a $OUT/pytest_gen_g$g.log
  tail -3 $OUT/pytest_gen_g$g.log
done
for g in 64 32 16 64 16; do
  BBAI_PREGEN_GROUP=$g timeout 300 python tools/gen_rate.py >> $OUT/gen_rate.jsonl 2>> $OUT/gen_rate.err
done
cat $OUT/gen_rate.jsonl
for g in 64 16 32 64 16; do
  for cfg in C2 C3 C4-shard; do
    BBAI_PREGEN_GROUP=$g timeout 300 python bench.py --config $cfg --steps 256 --warmup 16 --no-cpu-baseline --parity-envs 256 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'pregen_group': $g, 'config': '$cfg', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'parity': d['parity']['mismatches_all_ranks'], 'resets': d['config']['resets_in_timed_region'], 'kernels': d['roofline']['kernel_avg_ms']}))" >> $OUT/pregen_group_bench.jsonl
  done
done
cat $OUT/pregen_group_bench.jsonl
timeout 300 tools/ubench_gather > $OUT/ubench_gather.jsonl 2> $OUT/ubench_gather.err
cat $OUT/ubench_gather.jsonl
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
    print(k, v)
PY
find $OUT -name "*kernel_trace.csv" -size +5M -delete
