"""Aggregates the rocprofv3 --pmc passes written by profiles/collect.sh into per-kernel sums.

    python profiles/summarize_pmc.py gpurun_out/<tag>   ->  <tag>/pmc_summary.json, <tag>/traffic.json

traffic.json is what bench.py reports as roofline.traffic: memory-side bytes per point-kernel launch from FETCH_SIZE
and WRITE_SIZE as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: both count KiB; on gfx950
FETCH_SIZE reports half of the bytes of 16-byte-per-lane loads (every load of this kernel is a buffer_load_dwordx4),
so it is doubled; WRITE_SIZE is uncalibrated and taken as reported.  Infinity-Cache hits are included in these
counters, so this is an upper bound of the HBM traffic.
"""
import csv
import glob
import json
import os
import hashlib
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sources the point kernel is compiled from: bench.py quotes a traffic file only while their hash matches the tree it runs on
KERNEL_SOURCES = ('nr_kernels.h', 'nr_device.h', 'nr_layout.h', 'nr_platform.h')


def kernel_source_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, 'neuray_amd', 'csrc', f), 'rb').read())
    return h.hexdigest()[:16]


def main(out):
    sums = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in glob.glob(os.path.join(out, 'pmc*', '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r['Kernel_Name'].split('(')[0]
                sums[k][r['Counter_Name']] += float(r['Counter_Value'])
                disp[(k, f)].add(r['Dispatch_Id'])
    res = {}
    for k, c in sums.items():
        n = max(len(v) for (kk, _), v in disp.items() if kk == k)
        res[k] = dict(sorted(c.items()), dispatches=n)
    json.dump(res, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1)
    pk = [k for k in res if 'points_kernel' in k]
    if pk and 'FETCH_SIZE' in res[pk[0]]:
        r = res[pk[0]]
        traffic = {'kernel': pk[0], 'launches': r['dispatches'],
                   'FETCH_SIZE_KB_sum': r['FETCH_SIZE'], 'WRITE_SIZE_KB_sum': r.get('WRITE_SIZE', 0.0),
                   'correction': 'FETCH_SIZE x2 (gfx950 wide-load under-count, MI355X_MICROARCH.md HBM section); '
                                 'WRITE_SIZE as reported (uncalibrated)'}
        traffic['bytes_per_launch'] = (2.0 * r['FETCH_SIZE'] + r.get('WRITE_SIZE', 0.0)) * 1024.0 / r['dispatches']
        traffic['kernel_source_sha16'] = kernel_source_hash()     # (of KERNEL_SOURCES, on the tree the counters were collected on)
        if 'TCC_HIT_sum' in r:
            traffic['l2_hit_rate'] = r['TCC_HIT_sum'] / max(r['TCC_HIT_sum'] + r['TCC_MISS_sum'], 1.0)
        json.dump(traffic, open(os.path.join(out, 'traffic.json'), 'w'), indent=1)
        print(json.dumps(traffic, indent=1))
    for k in pk:
        r = res[k]
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in r and 'GRBM_GUI_ACTIVE' in r:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
            simd_cycles = r['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0
            print(k, 'MFMA pipe busy fraction (PMC):', r['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles)


if __name__ == '__main__':
    main(sys.argv[1])
