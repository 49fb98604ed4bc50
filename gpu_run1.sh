set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocm-smi --showproductname 2>&1 | head -8 > gpurun_out/gpuinfo.txt; lscpu | head -20 >> gpurun_out/gpuinfo.txt; nproc >> gpurun_out/gpuinfo.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench1.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench1.log
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof1.log 2>&1; echo "prof rc=$?" >> gpurun_out/prof1.log
tail -5 gpurun_out/smoke.log gpurun_out/pytest_gpu.log gpurun_out/bench1.log gpurun_out/prof1.log
