cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench2.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench2.log
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof2.log 2>&1; echo "prof rc=$?" >> gpurun_out/prof2.log
find gpurun_out/prof2 -type f | head; tail -n 3 gpurun_out/bench2.log
find gpurun_out/prof2 -name "*kernel_stats*" -exec cat {} \;
# keep the trace small: drop the per-dispatch trace if large
find gpurun_out/prof2 -name "*kernel_trace*" -size +8M -delete
