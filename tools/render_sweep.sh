#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sweep; mkdir -p $OUT; cd $REPO
for nt in 1 0; do for gpb in 1 2 4 8 16 64 512; do
  r=$(BBAI_RENDER_NT=$nt BBAI_RENDER_GPB=$gpb timeout 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_ms'], round(d['value']/1e6))")
  echo "nt=$nt gpb=$gpb $r" | tee -a $OUT/render_sweep.txt
done; done
