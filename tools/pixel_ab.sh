#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for b in 2 1 2 1 4; do
  r=$(BBAI_LOOKAHEAD=$b timeout 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:round(v,3) for k,v in d['roofline']['avg_ms'].items()}, round(d['ms_per_step'],3), round(d['value']/1e6))")
  echo "B=$b $r"
done
