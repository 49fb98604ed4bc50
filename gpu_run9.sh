cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -n 4 gpurun_out/pytest_gpu.log
for cfg in "1 0" "2 0" "2 1"; do set -- $cfg; 
  if [ "$2" = "1" ]; then export NEURAY_CAP128=1; else unset NEURAY_CAP128; fi
  NEURAY_NT=$1 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample-rays 1024 > gpurun_out/bench_nt$1_cap$2.log 2>&1
  echo "NT=$1 cap128=$2: $(python -c "
import json,sys
l=[x for x in open('gpurun_out/bench_nt$1_cap$2.log') if x.startswith('{')]
d=json.loads(l[-1]); print(int(d['value']), 'rays/s  frac %.3f  avg_ms %.2f' % (d['roofline']['frac'], d['roofline']['avg_launch_ms']), d['parity'])")"
done
