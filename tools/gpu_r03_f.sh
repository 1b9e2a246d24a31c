#!/bin/bash
# Round 3, lease F: the look-ahead generator confined to k CUs (CU mask on its stream): reset-heavy shards and the headline;
# SQ counters of k_step on the encoded 1M workload (what are its waves waiting for?).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03f
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_trace or odd_batch or autoreset or checkpoint or very_short or reseed" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
BBAI_PREGEN_CUS=64 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_trace or autoreset or very_short or reseed" > $OUT/pytest_cus64.log 2>&1; echo "pytest (64 generator CUs) rc=$?" >> $OUT/pytest_cus64.log
tail -3 $OUT/pytest_cus64.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict($1, ms_per_step=d['ms_per_step'], value=d['value'], parity=(d['parity'] or {}).get('mismatches_all_ranks'), kernels=d['roofline']['kernel_avg_ms'])))"; }
for rep in 1 2; do
  for cus in 0 32 64 96 128; do
    for cfg in C2 C3 C4-shard; do
      BBAI_PREGEN_CUS=$cus timeout 300 python bench.py --own-stream --config $cfg --steps 256 --warmup 16 --no-cpu-baseline --parity-envs 128 --min-seconds 0.5 2>>$OUT/ab.err | line "pregen_cus=$cus, config='$cfg', own_stream=1" >> $OUT/pregen_cus_ab.jsonl
    done
  done
done
for cus in 0 64; do
  BBAI_PREGEN_CUS=$cus timeout 300 python bench.py --config C2 --steps 256 --warmup 16 --no-cpu-baseline --parity-envs 128 --min-seconds 0.5 2>>$OUT/ab.err | line "pregen_cus=$cus, config='C2', own_stream=0" >> $OUT/pregen_cus_ab.jsonl
  BBAI_PREGEN_CUS=$cus timeout 300 python bench.py --own-stream --config C5-shard --steps 64 --warmup 8 --no-cpu-baseline --parity-envs 128 --min-seconds 0.5 2>>$OUT/ab.err | line "pregen_cus=$cus, config='C5-shard', own_stream=1" >> $OUT/pregen_cus_ab.jsonl
  BBAI_PREGEN_CUS=$cus timeout 300 python bench.py --own-stream --steps 20 --warmup 5 --no-cpu-baseline --parity-envs 128 --min-seconds 0.6 2>>$OUT/ab.err | line "pregen_cus=$cus, config='boss_pixel_1M', own_stream=1" >> $OUT/pregen_cus_ab.jsonl
done
cat $OUT/pregen_cus_ab.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/sq1 -o enc -- python $REPO/bench.py --no-pixel --steps 16 --warmup 4 --no-cpu-baseline --parity-envs 0 --min-seconds 0 > $OUT/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq2 -o enc -- python $REPO/bench.py --no-pixel --steps 16 --warmup 4 --no-cpu-baseline --parity-envs 0 --min-seconds 0 > $OUT/sq2.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for p in ("sq1", "sq2"):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_step" in k or "k_consume" in k:
                rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in rows.items():
        for c, v in d.items():
            res[k][c] = sorted(v)[len(v) // 2]
json.dump(res, open("$OUT/step_sq_counters_boss_encoded_1M.json", "w"), indent=1)
for k, v in res.items():
    print(k[:60], v)
PY
find $OUT -name "*.csv" -size +5M -delete
