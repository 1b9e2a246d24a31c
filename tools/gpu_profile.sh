#!/bin/bash
# Run on the GPU box (via gpurun): headline bench + rocprofv3 kernel stats + HBM counter passes.
# usage: tools/gpu_profile.sh <round-tag>
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $REPO/bench.py --steps 32 --warmup 8 > $OUT/bench_boss_pixel_1M.json 2> $OUT/bench_boss_pixel_1M.err
timeout 300 python $REPO/bench.py --steps 64 --warmup 8 --no-pixel --no-cpu-baseline > $OUT/bench_boss_encoded_1M.json 2>> $OUT/bench.err
timeout 300 python $REPO/bench.py --level GoToLocal --envs 65536 --steps 256 --warmup 16 --no-pixel --no-cpu-baseline > $OUT/bench_gotolocal_65536.json 2>> $OUT/bench.err
timeout 300 python $REPO/bench.py --level PickupLoc --envs 262144 --steps 128 --warmup 16 --no-pixel --no-cpu-baseline > $OUT/bench_pickuploc_262144.json 2>> $OUT/bench.err
timeout 300 python $REPO/bench.py --level GoTo --envs 131072 --steps 128 --warmup 16 --no-pixel --no-cpu-baseline > $OUT/bench_goto_131072.json 2>> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o boss -- python $REPO/bench.py --steps 32 --warmup 8 --no-cpu-baseline > $OUT/rocprof_stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o boss -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/rocprof_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o boss -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/rocprof_write.log 2>&1
# keep the merged payload small: traces can be large
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT $OUT/stats 2>/dev/null | head -40
cat $OUT/*.json
