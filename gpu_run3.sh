cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python gpu_diag.py > gpurun_out/diag.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
cat gpurun_out/diag.log; tail -n 12 gpurun_out/pytest_gpu.log
