cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
(timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q 2>&1 | tail -30) > gpurun_out/r02d/pytest_multirank.log
cat gpurun_out/r02d/pytest_multirank.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02d/bench_default.json 2> gpurun_out/r02d/bench_default.err
tail -3 gpurun_out/r02d/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02d/bench_default.json'))
print(json.dumps({k:d[k] for k in ('value','ms_per_step','timing','parity','cpu_baseline')}, indent=1)[:3000])
print(json.dumps(d['roofline'], indent=1)[:2500])
PY
