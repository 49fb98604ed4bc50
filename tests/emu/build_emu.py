"""Builds tests/emu/_build/libneuray_emu.so: the product kernel sources compiled for the CPU fiber
emulator (hip_emu.h).  TEST INFRASTRUCTURE ONLY - never loaded by neuray_amd, never timed."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'neuray_amd', 'csrc')
OUT_DIR = os.path.join(HERE, '_build')
OUT = os.path.join(OUT_DIR, 'libneuray_emu.so')
SOURCES = [os.path.join(CSRC, 'neuray_hip.hip'), os.path.join(CSRC, 'nr_pack.cpp'), os.path.join(HERE, 'hip_emu.cpp')]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in ('nr_kernels.h', 'nr_kernels_bwd.h', 'nr_kernels_bwd2.h', 'nr_kernels_dr.h', 'nr_kernels_norm.h', 'nr_kernels_conv3d.h', 'nr_kernels_conv2d.h', 'nr_device.h',
                                              'nr_layout.h', 'nr_platform.h', 'nr_pack.h')] + \
    [os.path.join(HERE, 'hip_emu.h'), os.path.join(ROOT, 'include', 'neuray_hip.h')]


def build(force=False, variant='fp32'):
    """variant 'bf16': the same sources with -DNR_BF16_QUADS (the bf16-operand library on the emulator); 'bf16x3': + -DNR_BF16_SPLIT"""
    os.makedirs(OUT_DIR, exist_ok=True)
    out = OUT if variant == 'fp32' else OUT.replace('.so', '_%s.so' % variant)
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in DEPS):
        return out
    cmd = ['g++', '-std=c++17', '-O2', '-g', '-rdynamic', '-fPIC', '-shared', '-DNEURAY_EMU', '-ffp-contract=off',
           '-fno-strict-aliasing', '-Wno-unused-value', '-I', HERE, '-I', CSRC, '-pthread', '-o', out + '.building.%d' % os.getpid()]
    if variant in ('bf16', 'bf16x3'):
        cmd.append('-DNR_BF16_QUADS')
    if variant == 'bf16x3':
        cmd.append('-DNR_BF16_SPLIT')
    for s in SOURCES:
        cmd += ['-x', 'c++', s]
    subprocess.check_call(cmd)
    os.replace(out + '.building.%d' % os.getpid(), out)        # (a half-written library is never visible under the final name)
    return out


if __name__ == '__main__':
    print(build(force='-f' in sys.argv))
