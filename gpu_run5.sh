cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/pmc/counters.txt 2>&1
grep -i -E "MFMA|SQ_WAIT|SQ_BUSY|SQ_WAVE|SQ_ACTIVE|SQ_INSTS|FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS|GRBM_GUI|LDS" gpurun_out/pmc/counters.txt | head -150 > gpurun_out/pmc/counters_short.txt
run() { tag=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$tag -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc/$tag.log 2>&1; echo "$tag rc=$?"; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run c FETCH_SIZE
run d WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python - <<'PY'
import csv, glob, collections
for tag in 'abcd':
    for f in glob.glob('gpurun_out/pmc/%s/**/*counter_collection.csv' % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:40]; acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, d in acc.items():
            if 'points' in k or 'rays_kernel' in k:
                print(tag, k, {c: '%.4g' % v for c, v in d.items()})
PY
tail -n 3 gpurun_out/pmc/a.log
# keep only summaries
find gpurun_out/pmc -name "*kernel_trace.csv" -delete
