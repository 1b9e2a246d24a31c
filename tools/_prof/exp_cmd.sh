mkdir -p gpurun_out/r05x; R=$PWD; cd /tmp
for lib in babyai_amd/libbbai_hip.so tools/_prof/libbbai_exp5.so babyai_amd/libbbai_hip.so tools/_prof/libbbai_exp5.so; do
  echo "{\"lib\": \"$lib\"}" | tee -a $R/gpurun_out/r05x/exp5.jsonl
  for cfg in "BossLevel 1048576 30" "GoTo 131072 100" "PickupLoc 262144 100" "GoToLocal 65536 200"; do BBAI_ENGINE_LIB=$R/$lib timeout 200 python $R/tools/bot_bench.py $cfg 2>/dev/null | tail -1 | tee -a $R/gpurun_out/r05x/exp5.jsonl; done
done
