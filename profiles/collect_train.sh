#!/bin/bash
# PMC passes over one training step (tools/bench_train.py): instruction mix / wait cycles / traffic of the backward kernels.
#   bash profiles/collect_train.sh <tag>   ->  gpurun_out/<tag>/{kernel_stats.csv, pmc_summary.json}
set -u
TAG=${1:-train}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python tools/bench_train.py --steps 2"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- $CMD > "$OUT/trace.log" 2>&1
cp "$(find "$OUT/trace" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_SMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -o run -- $CMD > "$OUT/pmc$i.log" 2>&1
done
python profiles/summarize_pmc.py "$OUT" > "$OUT/summarize.log" 2>&1
rm -rf "$OUT"/trace "$OUT"/pmc[0-9]
ls -la "$OUT"
