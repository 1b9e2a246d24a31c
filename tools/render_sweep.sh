#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sweep; mkdir -p $OUT; cd $REPO
for mode in 0 2 0 2; do
  r=$(BBAI_RENDER_MODE=$mode timeout 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_ms'], round(d['value']/1e6))")
  echo "mode=$mode $r" | tee -a $OUT/render_sweep4.txt
done
BBAI_RENDER_MODE=2 timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -k render 2>&1 | tail -3
