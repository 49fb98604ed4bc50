#!/bin/bash
# Kernel trace of whole fine-tuning steps (NeuralRayFtRenderer.train_step + backward + Adam, 24 views of 800 x 800, 512 rays, 64+64):
#   bash profiles/collect_ft_step.sh <tag>   ->  gpurun_out/<tag>/{ft_host_profile.txt, ft_step_kernel_stats.csv, ft_step_by_class.txt}
set -u
TAG=${1:-ft}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 200 python tools/profile_ft_step.py 2>&1 | grep -v amdgpu > "$OUT/ft_host_profile.txt"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python tools/profile_ft_step.py > "$OUT/trace.log" 2>&1
F=$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)
if [ -n "$F" ]; then
    cp "$F" "$OUT/ft_step_kernel_stats.csv"
    python tools/profile_ft_step.py --classify "$OUT/ft_step_kernel_stats.csv" 25 > "$OUT/ft_step_by_class.txt"
    cat "$OUT/ft_step_by_class.txt"
fi
rm -rf "$OUT/trace"
head -3 "$OUT/ft_host_profile.txt"
