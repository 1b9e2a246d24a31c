#!/bin/bash
# experiment: build the engine with the expert's phase timers and print the breakdown (run via gpurun; the build happens on the box)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out/botprof
[ -f tools/libbbai_prof.so ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DBBAI_BOT_PROF -o tools/libbbai_prof.so babyai_amd/csrc/bbai_engine.hip
for cfg in "BossLevel 262144 40" "GoToLocal 65536 100"; do
  BBAI_ENGINE_LIB=$REPO/tools/libbbai_prof.so python tools/bot_prof.py $cfg >> gpurun_out/botprof/bot_prof.jsonl
done
cat gpurun_out/botprof/bot_prof.jsonl
