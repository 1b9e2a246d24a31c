#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for b in 4 8; do for cfg in "--level GoTo --envs 131072 --steps 128" "--level GoToLocal --envs 65536 --steps 256" "--level PickupLoc --envs 262144 --steps 128" "--level BossLevel --envs 1048576 --steps 64"; do
  r=$(BBAI_LOOKAHEAD=$b timeout 200 python bench.py $cfg --warmup 16 --no-pixel --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value']/1e6))")
  echo "B=$b $cfg -> $r"
done; done
