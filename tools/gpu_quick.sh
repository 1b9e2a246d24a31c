#!/bin/bash
# quick GPU check: parity tests + headline bench (+ optional extra configs)
TAG=${1:-quick}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > $OUT/pytest.log
cat $OUT/pytest.log
timeout 120 python tools/membw.py > $OUT/membw.json 2>$OUT/membw.err; cat $OUT/membw.json
timeout 600 python bench.py --steps 32 --warmup 8 --no-cpu-baseline > $OUT/bench_boss_pixel_1M.json 2> $OUT/bench.err; cat $OUT/bench_boss_pixel_1M.json
timeout 300 python bench.py --steps 64 --warmup 8 --no-pixel --no-cpu-baseline > $OUT/bench_boss_encoded_1M.json 2>> $OUT/bench.err; cat $OUT/bench_boss_encoded_1M.json
timeout 300 python bench.py --level GoToLocal --envs 65536 --steps 256 --warmup 16 --no-pixel --no-cpu-baseline > $OUT/bench_gotolocal_65536.json 2>> $OUT/bench.err; cat $OUT/bench_gotolocal_65536.json
timeout 300 python bench.py --level GoTo --envs 131072 --steps 128 --warmup 16 --no-pixel --no-cpu-baseline > $OUT/bench_goto_131072.json 2>> $OUT/bench.err; cat $OUT/bench_goto_131072.json
tail -5 $OUT/bench.err
