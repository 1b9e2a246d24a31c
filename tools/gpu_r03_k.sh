#!/bin/bash
# Round 3, lease K: generate_demos on the device-resident history (tests + throughput against the stepwise host loop, large
# batches); k_step experiment builds against the shipped one: 128 / 64 envs per block (BBAI_STEP_BLOCK) and the front cell's
# id fetched with the window (BBAI_PREFETCH_ID), per workload, in-run oracle parity on every line.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $REPO
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "generate_demos or bot_rollout or unaligned" > $OUT/gpu_tests_lease_k.log 2>&1; echo "pytest rc=$?" >> $OUT/gpu_tests_lease_k.log
tail -3 $OUT/gpu_tests_lease_k.log
ab() {   # tag, bench args...
  tag=$1; shift
  for lib in new sb128 sb64 pfid pfid_sb128; do
    if [ $lib = new ]; then unset BBAI_ENGINE_LIB; else export BBAI_ENGINE_LIB=$REPO/tools/libbbai_$lib.so; fi
    timeout 300 python bench.py "$@" --no-cpu-baseline --parity-envs 256 --parity-pixel-envs 16 --min-seconds 0.8 2>>$OUT/ab_k.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'lib': '$lib', 'config': '$tag', 'ms_per_step': d['ms_per_step'], 'value': d['value'], 'parity': d['parity']['mismatches_all_ranks'], 'kernels': d['roofline']['kernel_avg_ms']}))" >> $OUT/step_variants_ab.jsonl
  done
  unset BBAI_ENGINE_LIB
}
ab boss_encoded_1M --no-pixel --steps 64 --warmup 8
ab pickuploc_262144 --config C3 --steps 128 --warmup 16
ab goto_131072 --config C4-shard --steps 128 --warmup 16
ab gotolocal_65536 --config C2 --steps 256 --warmup 16
cat $OUT/step_variants_ab.jsonl
timeout 300 python tools/demo_bench.py BossLevel 32768 32768 > $OUT/demo_bench_large.jsonl 2>> $OUT/demo_bench.err
timeout 200 python tools/demo_bench.py GoToLocal 65536 65536 >> $OUT/demo_bench_large.jsonl 2>> $OUT/demo_bench.err
timeout 200 python tools/demo_bench.py BossLevel 8192 4096 >> $OUT/demo_bench_large.jsonl 2>> $OUT/demo_bench.err
cat $OUT/demo_bench_large.jsonl; tail -3 $OUT/demo_bench.err
