#!/bin/bash
# Round 3, lease I (last): GPU suite + judged line + rocprofv3 trace + HBM counter passes on the FINAL kernel sources; the expert
# loop against the generator's lane-group width.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests_final.log 2>&1; echo "tests rc=$?" >> $OUT/gpu_tests_final.log
tail -3 $OUT/gpu_tests_final.log
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 20 --warmup 5"
timeout 900 $B > $OUT/bench_boss_pixel_1M.json 2> $OUT/bench_boss_pixel_1M.err
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o boss -- $B --no-cpu-baseline --parity-envs 0 > $OUT/bench_boss_pixel_1M_under_rocprof.json 2> $OUT/rocprof_stats.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o boss -- $B --no-cpu-baseline --parity-envs 0 --min-seconds 0 > $OUT/rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o boss -- $B --no-cpu-baseline --parity-envs 0 --min-seconds 0 > $OUT/rocprof_write.log 2>&1
cd $REPO
for g in 32 64 32 64; do
  for job in "BossLevel 1048576 40" "BossLevel 262144 100" "GoToLocal 65536 200"; do
    BBAI_PREGEN_GROUP=$g timeout 300 python tools/bot_bench.py $job 2>> $OUT/bot_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['pregen_group']=$g; print(json.dumps(d))" >> $OUT/bot_bench_by_group_width.jsonl
  done
done
cat $OUT/bot_bench_by_group_width.jsonl
find $OUT -name "*kernel_trace.csv" -size +20M -delete
python -c "
import json
d=json.loads(open('$OUT/bench_boss_pixel_1M.json').read().strip().splitlines()[-1])
print(round(d['value']/1e6,1),'M steps/s', round(d['ms_per_step'],4),'ms/step', 'frac', round(d['roofline']['frac'],3), 'of achievable', d['roofline']['frac_of_achievable'], 'parity', d['parity']['mismatches_all_ranks'], 'kernels', d['roofline']['kernel_avg_ms'])"
