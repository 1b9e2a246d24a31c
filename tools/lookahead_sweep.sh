#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sweep; mkdir -p $OUT; cd $REPO
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4)
for d in 1 2 4; do
  for cfg in "--level BossLevel --envs 1048576 --steps 64" "--level GoTo --envs 131072 --steps 128" "--level GoToLocal --envs 65536 --steps 256" "--level PickupLoc --envs 262144 --steps 128"; do
    r=$(BBAI_LOOKAHEAD=$d timeout 200 python bench.py $cfg --warmup 16 --no-pixel --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value']/1e6))")
    echo "depth=$d $cfg -> $r" | tee -a $OUT/lookahead.txt
  done
done
BBAI_LOOKAHEAD=1 timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pixel depth1', round(d['ms_per_step'],4), round(d['value']/1e6))"
BBAI_LOOKAHEAD=2 timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pixel depth2', round(d['ms_per_step'],4), round(d['value']/1e6))"
