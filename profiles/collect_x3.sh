#!/bin/bash
# PMC passes over one image rendered by the AR_X3 point kernel and one by the fp32 one (same process, kernels told apart by their template
# arguments):   bash profiles/collect_x3.sh <tag>   ->  gpurun_out/<tag>/pmc_summary.json
# Counters only (--kernel-trace --pmc), one rocprofv3 run per counter set.
set -u
TAG=${1:-x3prof}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python tools/ab_forward.py --steps 1 x3=neuray_amd/libneuray_hip.so,x3 f32=neuray_amd/libneuray_hip.so"
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -o run -- $CMD > "$OUT/pmc$i.log" 2>&1
done
python profiles/summarize_pmc.py "$OUT" > "$OUT/summarize.log" 2>&1
python - "$OUT" <<'PY'
import json, sys
r = json.load(open(sys.argv[1] + '/pmc_summary.json'))
for k, v in r.items():
    if 'points_kernel' in k:
        print(k)
        for c, x in sorted(v.items()):
            print('   %-28s %.4g' % (c, x))
PY
rm -rf "$OUT"/pmc[0-9]
