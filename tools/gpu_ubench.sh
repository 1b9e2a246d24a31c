#!/bin/bash
# Run on the GPU box (via gpurun): the microbenchmarks the kernel design notes quote, outputs kept under gpurun_out/<tag>/
# usage: tools/gpu_ubench.sh <tag>      (then copy the outputs to profiles/<tag>/)
TAG=${1:-ubench}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
timeout 120 python tools/membw.py > $OUT/membw.json 2> $OUT/membw.err
timeout 200 tools/ubench_gather > $OUT/ubench_gather.jsonl 2> $OUT/ubench_gather.err
timeout 300 tools/ubench_render > $OUT/ubench_render.jsonl 2> $OUT/ubench_render.err
timeout 200 tools/ubench_store > $OUT/ubench_store.txt 2> $OUT/ubench_store.err
cat $OUT/membw.json $OUT/ubench_gather.jsonl $OUT/ubench_render.jsonl
tail -30 $OUT/ubench_store.txt
