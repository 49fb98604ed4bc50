"""Per-section cycle counts of the resident point backward (points_backward2_kernel), workgroup 0, per wave.

Builds a profile variant of the library (-DNR_B2_PROFILE: clock64 at the B2_MARK points of csrc/nr_kernels_bwd2.h, summed over
the tiles of workgroup 0) into _ab/lib_b2prof.so, runs tools/time_bwd.py's launch with it and prints cycles per tile between
consecutive marks.       python tools/profile_bwd2.py --build   (no GPU needed)      python tools/profile_bwd2.py   (GPU)"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, '_ab', 'lib_b2prof.so')
NAMES = {0: 'DMA of the saved tile + projection + gathers', 1: 'dist decoder outputs (saved)', 2: 'prob, prob_embed, ray_dir_fc, neuray_fc fwd',
         3: 'weights + barrier (DMA visible)', 4: '-', 5: 'base_fc .. rgb_fc fwd', 6: 'softmax weights (saved sums)',
         7: '-', 9: 'geometry: waves 0-3 read hidden, G', 11: '  geometry_fc.2^T', 12: '  geometry_fc.0^T tile + hand-off (+ vec, staging on wave 0)',
         13: '  staging (waves 1, 2)', 14: '  barrier', 15: '  read hand-off + jobs', 16: 'blend bwd (1 all-reduce)', 17: 'rgb_fc chain', 18: '  barrier',
         19: '  staging writes', 20: '  barrier', 21: '  jobs', 22: 'vis_fc2 bwd (whole round)', 23: 'vis_fc bwd (whole round)', 24: 'base_fc chain',
         25: '  barrier', 26: '  staging + barrier + jobs B2', 27: '  staging + 2 barriers + jobs BV', 28: '  two all-reduces of d h64',
         29: '  statistics^T tiles, hand-off, staging, 2 barriers', 30: '  jobs BG (per point)', 31: 'statistics bwd',
         32: 'neuray_fc + ray_dir_fc bwd (whole round)', 33: 'prob_embed bwd (whole round)', 34: 'prob bwd / between heads', 35: 'head: fwd 2 layers',
         36: 'head: vec^T, 2 transposed layers', 37: 'head: barrier', 38: 'head: staging writes', 39: 'head: barrier', 40: 'head: jobs',
         41: '(after last head)', 42: 'scatter'}


def build():
    sys.path.insert(0, ROOT)
    from neuray_amd import build as b
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [b.HIPCC] + b.FLAGS + ['-DNR_B2_PROFILE'] + b.SOURCES + ['-o', LIB]
    subprocess.check_call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build', action='store_true')
    args = ap.parse_args()
    if args.build or not os.path.exists(LIB):
        build()
        if args.build:
            return
    os.environ['NEURAY_HIP_LIB'] = LIB
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    sys.argv = ['time_bwd.py', '--reps', '1']
    import time_bwd
    lib = ctypes.CDLL(LIB)
    buf = (ctypes.c_ulonglong * (64 * 8))()
    time_bwd.main()                                     # warm-up launches + 1 timed launch
    lib.neuray_debug_b2_profile(buf, 1)                 # clear
    time_bwd.main()
    assert lib.neuray_debug_b2_profile(buf, 0) == 0
    import torch
    launches = 3                                        # two warm-up launches + one timed launch per time_bwd.main()
    tiles = 512 * 64 // 16 // 256 * launches
    tot = [sum(buf[m * 8 + w] for m in range(64)) for w in range(8)]
    print('cycles per tile, wave 0: %d' % (tot[0] // tiles))
    print('%-58s' % 'section' + ''.join('%9s' % ('wave %d' % w) for w in range(8)) + '   % (w0)')
    for m in range(64):
        row = [buf[m * 8 + w] // tiles for w in range(8)]
        if any(row):
            print('%-58s' % NAMES.get(m, str(m)) + ''.join('%9d' % v for v in row) + '   %5.1f' % (100.0 * buf[m * 8] / tot[0]))


if __name__ == '__main__':
    main()
