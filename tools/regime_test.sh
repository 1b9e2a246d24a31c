#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() { python $REPO/bench.py --steps 32 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:round(v,3) for k,v in d['roofline']['avg_ms'].items()}, round(d['value']/1e6))"; }
cd /tmp; export TMPDIR=/tmp; echo "first (cold, cwd=/tmp): $(run)"
cd $REPO; echo "second (cwd=repo): $(run)"
cd /tmp; echo "third (cwd=/tmp): $(run)"
python $REPO/tools/membw.py > /dev/null 2>&1
cd $REPO; echo "after membw: $(run)"
cd /tmp; echo "again /tmp: $(run)"
