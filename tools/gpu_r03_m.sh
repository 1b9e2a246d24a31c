#!/bin/bash
# Round 3, lease M: k_step block size on the PIXEL workloads (the fused tile-plane pass runs in the same kernel): 256 / 128 / 64.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
ab() {
  tag=$1; shift
  for lib in new sb64 sb128 new sb64 sb128; do
    if [ $lib = new ]; then unset BBAI_ENGINE_LIB; else export BBAI_ENGINE_LIB=$REPO/tools/libbbai_$lib.so; fi
    timeout 300 python bench.py "$@" --no-cpu-baseline --parity-envs 256 --parity-pixel-envs 16 --min-seconds 0.8 2>>$OUT/ab_m.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib': '$lib', 'config': '$tag', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'parity': d['parity']['mismatches_all_ranks'], 'kernels': d['roofline']['kernel_avg_ms']}))" >> $OUT/step_block_pixel_ab.jsonl
  done
  unset BBAI_ENGINE_LIB
}
ab boss_pixel_1M --steps 20 --warmup 5
ab boss_pixel_131072 --config C5-shard --steps 64 --warmup 8
cat $OUT/step_block_pixel_ab.jsonl
