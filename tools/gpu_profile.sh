#!/bin/bash
# Run on the GPU box (via gpurun), ONE lease: the judged bench line, the rocprofv3 kernel trace of the same command, the
# HBM counter passes, the BASELINE configs at their per-GPU sizes, and the microbenchmarks the design notes quote.
# usage: tools/gpu_profile.sh <round-tag>      then, in the build container:  python tools/summarize_profile.py <round-tag>
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --steps 20 --warmup 5"
timeout 900 $B > $OUT/bench_boss_pixel_1M.json 2> $OUT/bench_boss_pixel_1M.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o boss -- $B --no-cpu-baseline --parity-envs 0 > $OUT/bench_boss_pixel_1M_under_rocprof.json 2> $OUT/rocprof_stats.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o boss -- $B --no-cpu-baseline --parity-envs 0 --min-seconds 0 > $OUT/rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o boss -- $B --no-cpu-baseline --parity-envs 0 --min-seconds 0 > $OUT/rocprof_write.log 2>&1
timeout 600 python $REPO/bench.py --steps 64 --warmup 8 --no-pixel --no-cpu-baseline > $OUT/bench_boss_encoded_1M.json 2>> $OUT/bench.err
timeout 600 python $REPO/bench.py --config C2 --steps 256 --warmup 16 --no-cpu-baseline > $OUT/bench_gotolocal_65536.json 2>> $OUT/bench.err
timeout 600 python $REPO/bench.py --config C3 --steps 128 --warmup 16 --no-cpu-baseline > $OUT/bench_pickuploc_262144.json 2>> $OUT/bench.err
timeout 600 python $REPO/bench.py --config C4 --steps 128 --warmup 16 --no-cpu-baseline > $OUT/bench_goto_131072.json 2>> $OUT/bench.err
timeout 600 python $REPO/bench.py --config C5 --steps 64 --warmup 8 --no-cpu-baseline > $OUT/bench_boss_pixel_131072.json 2>> $OUT/bench.err
cd $REPO
timeout 120 python tools/membw.py > $OUT/membw.json 2> $OUT/membw.err
timeout 200 tools/ubench_gather > $OUT/ubench_gather.jsonl 2> $OUT/ubench_gather.err
timeout 300 tools/ubench_render > $OUT/ubench_render.jsonl 2> $OUT/ubench_render.err
timeout 200 tools/ubench_store > $OUT/ubench_store.txt 2> $OUT/ubench_store.err
# keep the merged payload small: traces can be large
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT $OUT/stats 2>/dev/null | head -40
for f in $OUT/bench_*.json; do echo "== $f"; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
print(round(d['value']/1e6,1),'M steps/s', round(d['ms_per_step'],4),'ms/step', 'frac', round(d['roofline']['frac'],3), 'of achievable', d['roofline']['frac_of_achievable'], 'parity', (d['parity'] or {}).get('mismatches'), 'kernels', d['roofline']['kernel_avg_ms'])
"; done
