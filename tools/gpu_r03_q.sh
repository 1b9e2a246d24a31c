#!/bin/bash
# Round 3, lease Q (the last seconds of the budget): the ticket-queue render (tools/next/render_queue.patch, built as
# tools/libbbai_rq.so) INSIDE the step loop against the shipped one-shot shape, same library, same box.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
export BBAI_ENGINE_LIB=$REPO/tools/libbbai_rq.so
for q in 1 0 2 1 0 2; do
  BBAI_RENDER_QUEUE=$q timeout 60 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-envs 64 --parity-pixel-envs 16 --min-seconds 0.3 --prewarm-seconds 0.2 2>>$OUT/ab_q.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'render_queue': $q, 'config': 'boss_pixel_1M', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'parity': d['parity']['mismatches_all_ranks'], 'kernels': d['roofline']['kernel_avg_ms']}))" >> $OUT/render_queue_in_loop.jsonl
  cat $OUT/render_queue_in_loop.jsonl | tail -1
done
