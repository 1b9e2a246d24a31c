cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
(timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q --durations=8 -k "scattered" 2>&1 | tail -30) > gpurun_out/r02e/pytest_fullsize.log
cat gpurun_out/r02e/pytest_fullsize.log
