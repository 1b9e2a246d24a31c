#!/bin/bash
# Round 3, lease P (last): the judged line, its rocprofv3 trace and the HBM counter passes on the final DEFAULT path
# (render from the encoding; the fused tile plane is opt-in since lease N / O).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 20 --warmup 5"
timeout 400 $B > $OUT/bench_boss_pixel_1M.json 2> $OUT/bench_boss_pixel_1M.err
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o boss -- $B --no-cpu-baseline --parity-envs 0 > $OUT/bench_boss_pixel_1M_under_rocprof.json 2> $OUT/rocprof_stats.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o boss -- $B --no-cpu-baseline --parity-envs 0 --min-seconds 0 > $OUT/rocprof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o boss -- $B --no-cpu-baseline --parity-envs 0 --min-seconds 0 > $OUT/rocprof_write.log 2>&1
find $OUT -name "*.csv" -size +20M -delete
python -c "
import json
d=json.loads(open('$OUT/bench_boss_pixel_1M.json').read().strip().splitlines()[-1])
print(round(d['value']/1e6,1),'M steps/s', round(d['ms_per_step'],4),'ms/step', 'frac', round(d['roofline']['frac'],3), 'of achievable', d['roofline']['frac_of_achievable'], 'parity', d['parity']['mismatches_all_ranks'], 'kernels', d['roofline']['kernel_avg_ms'])"
