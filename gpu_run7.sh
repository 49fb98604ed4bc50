cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
NEURAY_NT=2 timeout 600 python gpu_diag2.py > gpurun_out/diag2.log 2>&1
echo "---- NT=1" >> gpurun_out/diag2.log
NEURAY_NT=1 timeout 600 python gpu_diag2.py >> gpurun_out/diag2.log 2>&1
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
cat gpurun_out/diag2.log; tail -n 8 gpurun_out/pytest_gpu.log
