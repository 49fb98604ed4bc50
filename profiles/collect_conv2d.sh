#!/bin/bash
# Kernel trace + PMC passes of tools/ab_conv2d.py (the encoders' 3 x 3 convolution on the split-operand bf16 MFMA against MIOpen):
#   bash profiles/collect_conv2d.sh <tag> [layer substring]   ->  gpurun_out/<tag>/{ab.log, kernel_stats.csv, pmc_summary.json}
set -u
TAG=${1:-conv2d}
ONLY=${2:-}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python tools/ab_conv2d.py --reps 10 --only=$ONLY"
$CMD > "$OUT/ab.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- $CMD > "$OUT/trace.log" 2>&1
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
i=0
for set in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -o run -- $CMD > "$OUT/pmc$i.log" 2>&1
done
python profiles/summarize_pmc.py "$OUT" > "$OUT/summarize.log" 2>&1
rm -rf "$OUT"/trace "$OUT"/pmc[0-9]
ls -la "$OUT"
