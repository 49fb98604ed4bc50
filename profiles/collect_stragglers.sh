#!/bin/bash
# Timings of the kernels VERDICT r4 #6 lists as stragglers (spills / scratch / occupancy 1), from rocprofv3 kernel traces:
#   bash profiles/collect_stragglers.sh <tag>   ->  gpurun_out/<tag>/stragglers.txt
#   - rays_backward_kernel<2> (128 samples per ray: cfg fine_depth_use_all) next to <1> (64 samples), in a training step
#   - dr_points_kernel / dr_rays_kernel (use_dr_prediction) next to the point kernel of the same image
set -u
TAG=${1:-stragglers}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
stat() { cp "$(find "$1" -name '*kernel_stats.csv' | head -1)" "$2"; rm -rf "$1"; }
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t1" -o run -- python tools/bench_train.py --steps 10 > "$OUT/train64.log" 2>&1
stat "$OUT/t1" "$OUT/train64_kernel_stats.csv"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t2" -o run -- python tools/bench_train.py --steps 10 --use-all > "$OUT/train128.log" 2>&1
stat "$OUT/t2" "$OUT/train128_kernel_stats.csv"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t3" -o run -- python -c "
import sys, json, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda', 0)
cfg, r, w, que, ref, tq, tr = bench.build_case(dev, 32, 0)
print(json.dumps(bench.direct_rendering_timing(dev, tq, tr)))
" > "$OUT/dr.log" 2>&1
stat "$OUT/t3" "$OUT/dr_kernel_stats.csv"
{
  echo "# calls, total ns, average ns, % of the trace's GPU time"
  for f in train64 train128 dr; do
    echo "== $f: $(tail -1 "$OUT/$f.log" | cut -c1-300)"
    grep -E 'rays_backward_kernel|points_backward|dr_points_kernel|dr_rays_kernel|points_kernel|rays_kernel' "$OUT/${f}_kernel_stats.csv" | cut -d, -f1-5 | cut -c1-200
  done
} > "$OUT/stragglers.txt"
cat "$OUT/stragglers.txt"
