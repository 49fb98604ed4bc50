#!/bin/bash
# Texture-path counters of the point kernel (the gather leg): L1 -> L2 read requests, their summed latency, L1 pending-miss
# stalls.  bash profiles/collect_gather.sh <tag> -> gpurun_out/<tag>/pmc_summary.json
# (A pass with the TA_*BUSY / TA_*STALLED counters aborts rocprofv3 7.2 on this image - signal 6 - and is left out.)
set -u
OUT=gpurun_out/${1:-gather}; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-eager-baseline"
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD" \
           "TA_BUFFER_READ_WAVEFRONTS_sum"; do
  i=$((i+1)); timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -o run -- $BENCH > "$OUT/pmc$i.log" 2>&1
done
python profiles/summarize_pmc.py "$OUT" > "$OUT/summarize.log" 2>&1
rm -rf "$OUT"/pmc[0-9]
