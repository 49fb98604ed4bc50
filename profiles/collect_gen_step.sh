#!/bin/bash
# Kernel trace of whole generalisation-training steps (bench.gen_train_case: cost-volume init net + encoders + per-ray path + render / depth
# loss + Adam; 512 rays, 8 views of 416 x 608, 64+64):
#   bash profiles/collect_gen_step.sh <tag>   ->  gpurun_out/<tag>/{gen_host_profile.txt, gen_step_kernel_stats.csv, gen_step_by_class.txt}
set -u
TAG=${1:-gen}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 python tools/profile_gen_step.py 2>&1 | grep -v amdgpu > "$OUT/gen_host_profile.txt"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python tools/profile_gen_step.py > "$OUT/trace.log" 2>&1
F=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
if [ -n "$F" ]; then
    cp "$F" "$OUT/gen_step_kernel_stats.csv"
    python tools/profile_gen_step.py --classify "$OUT/gen_step_kernel_stats.csv" 25 > "$OUT/gen_step_by_class.txt"
    cat "$OUT/gen_step_by_class.txt"
fi
rm -rf "$OUT/trace"
head -3 "$OUT/gen_host_profile.txt"
