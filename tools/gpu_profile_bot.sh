#!/bin/bash
# Run on the GPU box (via gpurun): expert (k_bot) throughput + rocprofv3 kernel stats.  usage: tools/gpu_profile_bot.sh <round-tag>
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "GoToLocal 65536 200" "PickupLoc 262144 100" "GoTo 131072 100" "BossLevel 262144 60" "BossLevel 1048576 30"; do
  set -- $cfg
  timeout 300 python $REPO/tools/bot_bench.py $1 $2 $3 2>> $OUT/bot_bench.err | tail -1 >> $OUT/bot_bench.jsonl
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bot_stats -o bot -- python $REPO/tools/bot_bench.py BossLevel 262144 40 > $OUT/rocprof_bot.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +20M -delete
cat $OUT/bot_bench.jsonl
find $OUT/bot_stats -name "*kernel_stats.csv" | head -1 | xargs head -8
