#!/bin/bash
# Round 3, lease D: the window plane (V plane + front-cell cache) in k_step: full GPU suite, A/B against the record-only path
# on every BASELINE workload, render policy at the mid sizes.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03d
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict($1, ms_per_step=d['ms_per_step'], value=d['value'], parity=(d['parity'] or {}).get('mismatches_all_ranks'), kernels=d['roofline']['kernel_avg_ms'], frac=d['roofline']['frac'], copy_GBs=d['roofline']['achievable']['copy_GBs'])))"; }
for rep in 1 2; do
  for vp in 1 0; do
    BBAI_VPLANE=$vp timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-envs 256 --min-seconds 0.6 2>>$OUT/ab.err | line "vplane=$vp, config='boss_pixel_1M'" >> $OUT/vplane_ab.jsonl
    BBAI_VPLANE=$vp timeout 300 python bench.py --no-pixel --steps 64 --warmup 8 --no-cpu-baseline --parity-envs 256 --min-seconds 0.5 2>>$OUT/ab.err | line "vplane=$vp, config='boss_encoded_1M'" >> $OUT/vplane_ab.jsonl
    for cfg in C2 C3 C4-shard C5-shard; do
      BBAI_VPLANE=$vp timeout 300 python bench.py --config $cfg --steps 128 --warmup 16 --no-cpu-baseline --parity-envs 256 --min-seconds 0.5 2>>$OUT/ab.err | line "vplane=$vp, config='$cfg'" >> $OUT/vplane_ab.jsonl
    done
  done
done
cat $OUT/vplane_ab.jsonl
for n in 262144 524288; do
  for fused in 1 0; do
    for shape in "2 512" "4 512" "8 1024"; do
      set -- $shape
      BBAI_RENDER_FUSED=$fused BBAI_RENDER_GROUP=$1 BBAI_RENDER_TPB=$2 timeout 300 python bench.py --envs $n --steps 32 --warmup 8 --no-cpu-baseline --parity-envs 128 --min-seconds 0.5 2>>$OUT/render_ab.err | line "fused=$fused, group=$1, tpb=$2, envs=$n" >> $OUT/render_fused_ab_mid.jsonl
    done
  done
done
cat $OUT/render_fused_ab_mid.jsonl
