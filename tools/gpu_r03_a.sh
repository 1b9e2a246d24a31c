#!/bin/bash
# Round 3, lease A (on the GPU box via gpurun): GPU test suite on the new bench / engine plumbing, the judged bench line,
# the per-call completion event A/B, the self-launched multi-rank line, the FETCH_SIZE calibration.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03a
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="python bench.py --steps 20 --warmup 5"
timeout 600 $B > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
for ce in 1 0 1 0; do
  BBAI_CALL_EVENTS=$ce timeout 300 python bench.py --config C2 --steps 256 --warmup 16 --no-cpu-baseline --parity-envs 0 2>>$OUT/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'call_events': $ce, 'config': 'C2', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'profiled_ms_per_step': d['timing']['profiled_ms_per_step'], 'kernels': d['roofline']['kernel_avg_ms']}))" >> $OUT/call_events_ab.jsonl
  BBAI_CALL_EVENTS=$ce timeout 300 python bench.py --config C5-shard --steps 64 --warmup 8 --no-cpu-baseline --parity-envs 0 2>>$OUT/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'call_events': $ce, 'config': 'C5-shard', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'profiled_ms_per_step': d['timing']['profiled_ms_per_step'], 'kernels': d['roofline']['kernel_avg_ms']}))" >> $OUT/call_events_ab.jsonl
done
cat $OUT/call_events_ab.jsonl
timeout 600 python bench.py --gpus 4 --share-device --dist-backend gloo --total-envs 262144 --steps 32 --warmup 8 --min-seconds 0.3 --no-cpu-baseline --parity-envs 256 > $OUT/bench_selflaunch_4ranks_one_gpu.json 2> $OUT/bench_selflaunch.err; echo "selflaunch rc=$?"
timeout 300 python bench.py --config C3 --steps 128 --warmup 16 --no-cpu-baseline > $OUT/bench_pickuploc_262144.json 2>> $OUT/bench.err
timeout 300 python bench.py --config C4-shard --steps 128 --warmup 16 --no-cpu-baseline > $OUT/bench_goto_131072.json 2>> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 120 $REPO/tools/ubench_fetchcal > $OUT/fetchcal.jsonl 2> $OUT/fetchcal.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetchcal_pmc -o cal -- $REPO/tools/ubench_fetchcal > $OUT/fetchcal_pmc.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections, json
rows = collections.defaultdict(list)
for f in glob.glob("$OUT/fetchcal_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") == "FETCH_SIZE":
            rows[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
with open("$OUT/fetchcal_counters.json", "w") as f:
    json.dump({k: {"launches": len(v), "FETCH_SIZE_last": v[-1], "FETCH_SIZE_all": v} for k, v in rows.items()}, f, indent=1)
print(open("$OUT/fetchcal_counters.json").read()[:3000])
PY
cat $OUT/fetchcal.jsonl
find $OUT -name "*kernel_trace.csv" -size +5M -delete
for f in $OUT/bench_*.json; do echo "== $f"; python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print(round(d['value']/1e6,1),'M steps/s', round(d['ms_per_step'],4),'ms/step; profiled', d['timing']['profiled_ms_per_step'], 'frac', round(d['roofline']['frac'],3), 'of achievable', d['roofline']['frac_of_achievable'], 'parity', (d['parity'] or {}).get('mismatches_all_ranks'), 'kernels', d['roofline']['kernel_avg_ms'], 'world', d['rccl']['world'], d['scaling'])
except Exception as e: print('ERR', e)
"; done
