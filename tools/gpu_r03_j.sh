#!/bin/bash
# Round 3, lease J (second session): the GPU suite on the reworked k_step (LDS rows at the output pitch, one-round-trip SoA
# loads, window fetch before the object actions), bbai_bot_rollout and the done-action mode; k_step A/B against the previous
# build (tools/libbbai_base.so = c861a17's sources) per workload; demonstration throughput, rollout vs the stepwise host loop.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests_lease_j.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests_lease_j.log
tail -4 $OUT/gpu_tests_lease_j.log
ab() {   # tag, bench args...
  tag=$1; shift
  for lib in base new base new; do
    if [ $lib = base ]; then export BBAI_ENGINE_LIB=$REPO/tools/libbbai_base.so; else unset BBAI_ENGINE_LIB; fi
    timeout 300 python bench.py "$@" --no-cpu-baseline --parity-envs 256 --parity-pixel-envs 16 --min-seconds 0.8 2>>$OUT/ab_j.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib': '$lib', 'config': '$tag', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'parity': d['parity']['mismatches_all_ranks'], 'kernels': d['roofline']['kernel_avg_ms']}))" >> $OUT/step_rows_ab.jsonl
  done
  unset BBAI_ENGINE_LIB
}
ab boss_encoded_1M --no-pixel --steps 64 --warmup 8
ab boss_pixel_1M --steps 20 --warmup 5
ab pickuploc_262144 --config C3 --steps 128 --warmup 16
ab goto_131072 --config C4-shard --steps 128 --warmup 16
ab gotolocal_65536 --config C2 --steps 256 --warmup 16
cat $OUT/step_rows_ab.jsonl
timeout 300 python tools/demo_bench.py BossLevel 8192 4096 > $OUT/demo_bench.jsonl 2>> $OUT/demo_bench.err
timeout 200 python tools/demo_bench.py GoToLocal 16384 8192 >> $OUT/demo_bench.jsonl 2>> $OUT/demo_bench.err
cat $OUT/demo_bench.jsonl; tail -3 $OUT/demo_bench.err
