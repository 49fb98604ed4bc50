#!/bin/bash
# Kernel trace of the per-image encoders on 9 x 800 x 800, forward + backward:
#   bash profiles/collect_encoders.sh <tag> [--fused 0|1]   ->  gpurun_out/<tag>/{encoders.json, encoder_kernel_stats.csv}
set -u
TAG=${1:-enc}
shift || true
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 200 python tools/bench_encoders.py "$@" > "$OUT/encoders.log" 2>&1
tail -1 "$OUT/encoders.log" > "$OUT/encoders.json"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python tools/bench_encoders.py --reps 3 "$@" > "$OUT/trace.log" 2>&1
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/encoder_kernel_stats.csv"
rm -rf "$OUT/trace"
cat "$OUT/encoders.json"
