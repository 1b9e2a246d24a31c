#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/small; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "PickupLoc 262144 128" "GoToLocal 65536 256" "GoTo 131072 128" "BossLevel 1048576 64"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1 -o p -- python $REPO/bench.py --level $1 --envs $2 --steps $3 --warmup 16 --no-pixel --no-cpu-baseline > $OUT/$1.log 2>&1
  echo "== $1 $2"; grep '^{' $OUT/$1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value']/1e6))"
  python - <<PY
import csv
rows=list(csv.DictReader(open('$OUT/$1/p_kernel_stats.csv')))
for r in rows[:7]:
    print('   %-14s calls=%5s avg_us=%9.1f total_ms=%8.2f' % (r['Name'].split('(')[0][:14], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
  rm -f $OUT/$1/p_kernel_trace.csv
done
