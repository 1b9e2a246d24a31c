cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02l
(timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_fullsize.py -m gpu -x -q -k "multirank or four_ranks or rccl or scattered" 2>&1 | tail -5) > gpurun_out/r02l/pytest.log
cat gpurun_out/r02l/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02l/bench_default.json 2> gpurun_out/r02l/bench_default.err
tail -3 gpurun_out/r02l/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02l/bench_default.json'))
print(d['value'], d['ms_per_step'], d['parity']['mismatches'], d['parity']['seconds'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['roofline']['kernel_avg_ms'])
PY
