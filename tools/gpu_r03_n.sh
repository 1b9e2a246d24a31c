#!/bin/bash
# Round 3, lease N: the headline's render input on the FINAL build, another box: fused tile plane (default) vs the encoding.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
for rep in 1 2 3; do
  for fused in 1 0; do
    BBAI_RENDER_FUSED=$fused timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-envs 256 --parity-pixel-envs 16 --min-seconds 1.0 2>>$OUT/ab_n.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'render_fused': $fused, 'config': 'boss_pixel_1M', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'parity': d['parity']['mismatches_all_ranks'], 'kernels': d['roofline']['kernel_avg_ms'], 'fill_GBs': d['roofline']['achievable']['fill_GBs']}))" >> $OUT/render_fused_ab_final.jsonl
  done
done
cat $OUT/render_fused_ab_final.jsonl
