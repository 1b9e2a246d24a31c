#!/bin/bash
# Round 3, lease H: where does the window plane cost the render?  Non-temporal obs stores in k_step (so that the step's output
# does not sit dirty in the memory-side cache when the render's store stream starts), and window plane on / off per pixel batch size.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03h
mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(dict($1, ms_per_step=d['ms_per_step'], value=d['value'], parity=(d['parity'] or {}).get('mismatches_all_ranks'), kernels=d['roofline']['kernel_avg_ms'])))"; }
for rep in 1 2 3; do
  for mode in "1 0" "1 1" "0 0" "0 1"; do
    set -- $mode
    BBAI_VPLANE=$1 BBAI_NT_OBS=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --parity-envs 128 --min-seconds 1.0 2>>$OUT/ab.err | line "vplane=$1, nt_obs=$2, envs=1048576" >> $OUT/nt_obs_ab.jsonl
  done
done
cat $OUT/nt_obs_ab.jsonl
for n in 524288 262144 131072; do
  for rep in 1 2; do
    for vp in 1 0; do
      BBAI_VPLANE=$vp timeout 300 python bench.py --envs $n --steps 32 --warmup 8 --no-cpu-baseline --parity-envs 128 --min-seconds 0.5 2>>$OUT/ab.err | line "vplane=$vp, envs=$n" >> $OUT/vplane_pixel_by_size.jsonl
    done
  done
done
cat $OUT/vplane_pixel_by_size.jsonl
