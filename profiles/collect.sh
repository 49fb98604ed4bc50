#!/bin/bash
# Profiling recipe of profiles/ (run on the MI355X box from the repo root, e.g. through gpurun):
#   bash profiles/collect.sh <tag>          ->  gpurun_out/<tag>/{bench.json, kernel_stats.csv, pmc_summary.json, traffic.json}
# Kernel trace and every PMC pass are separate rocprofv3 runs (no trace domain is mixed with --pmc).
set -u
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-eager-baseline"

timeout 300 python bench.py --steps 5 --warmup 2 > "$OUT/bench.log" 2>&1
tail -1 "$OUT/bench.log" > "$OUT/bench.json"

timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- $BENCH > "$OUT/trace.log" 2>&1
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"

i=0
for set in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -o run -- $BENCH > "$OUT/pmc$i.log" 2>&1
done
python profiles/summarize_pmc.py "$OUT" > "$OUT/summarize.log" 2>&1
rm -rf "$OUT"/trace "$OUT"/pmc[0-9]      # raw traces are large; the summaries are what gets committed
ls -la "$OUT"
