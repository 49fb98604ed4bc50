"""Builds neuray_amd/libneuray_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo).

    python -m neuray_amd.build [-f]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(HERE, 'libneuray_hip.so')
SOURCES = [os.path.join(CSRC, 'neuray_hip.hip'), os.path.join(CSRC, 'nr_pack.cpp')]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in ('nr_kernels.h', 'nr_kernels_bwd.h', 'nr_kernels_bwd2.h', 'nr_kernels_dr.h', 'nr_kernels_norm.h', 'nr_kernels_conv3d.h', 'nr_kernels_conv2d.h', 'nr_device.h', 'nr_layout.h', 'nr_platform.h', 'nr_pack.h')] + \
    [os.path.join(os.path.dirname(HERE), 'include', 'neuray_hip.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -ffp-contract=off: arithmetic is exactly as written (explicit fmaf where fusion is wanted), so a ray's result does
# not depend on which tile slot / lane it lands in (hipcc otherwise SLP-packs the unrolled tile copies and contracts
# packed and scalar leftovers differently) - required for bitwise batching / sharding invariance.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off', '-Wno-unused-value', '-Wno-pass-failed',
         '-Wno-comment']


OUT_BF16 = os.path.join(HERE, 'libneuray_hip_bf16.so')      # same sources with -DNR_BF16_QUADS (bf16 MFMA operands)
OUT_BF16X3 = os.path.join(HERE, 'libneuray_hip_bf16x3.so')  # ... + -DNR_BF16_SPLIT (hi + lo bf16 operands, three bf16 MFMAs per fp32 quad)


def needs_build(out=OUT):
    return not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(d) for d in DEPS)


def build(force=False, verbose=False):
    """-> path of the product library; also (re)builds the two separately reported variants next to it.  The stale ones are
    compiled concurrently (one hipcc process each, ~4 minutes per library from scratch); every output is written next to its
    final name and moved into place when its compiler succeeded, so an interrupted build never leaves a half-written library."""
    jobs = []
    for out, extra in ((OUT, []), (OUT_BF16, ['-DNR_BF16_QUADS']), (OUT_BF16X3, ['-DNR_BF16_QUADS', '-DNR_BF16_SPLIT'])):
        if not force and not needs_build(out):
            continue
        tmp = out + '.building'
        cmd = [HIPCC] + FLAGS + extra + SOURCES + ['-o', tmp]
        if verbose:
            print(' '.join(cmd[:-1] + [out]))
        jobs.append((out, tmp, cmd, subprocess.Popen(cmd)))
    failed = []
    for out, tmp, cmd, proc in jobs:
        if proc.wait() == 0:
            os.replace(tmp, out)
        else:
            failed.append(cmd)
            if os.path.exists(tmp):
                os.remove(tmp)
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    return OUT


if __name__ == '__main__':
    print(build(force='-f' in sys.argv, verbose=True))
