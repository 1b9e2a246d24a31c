#!/bin/bash
# Round 3, lease O: is the fused tile plane still worth it?  Render input (plane / encoding) x step kernel (this session's /
# the previous build, tools/libbbai_base.so), same box, alternated.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2; do
  for lib in new base; do
    for fused in 1 0; do
      if [ $lib = new ]; then unset BBAI_ENGINE_LIB; else export BBAI_ENGINE_LIB=$REPO/tools/libbbai_base.so; fi
      BBAI_RENDER_FUSED=$fused timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-envs 128 --parity-pixel-envs 8 --min-seconds 0.8 2>>$OUT/ab_o.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib': '$lib', 'render_fused': $fused, 'config': 'boss_pixel_1M', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'parity': d['parity']['mismatches_all_ranks'], 'kernels': d['roofline']['kernel_avg_ms']}))" >> $OUT/render_fused_by_step_build.jsonl
    done
  done
done
unset BBAI_ENGINE_LIB
cat $OUT/render_fused_by_step_build.jsonl
