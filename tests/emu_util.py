"""Helpers for the CPU test-suite: bind the emulator build of the kernels (tests/emu)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))

_EMU = None


def emu_lib():
    global _EMU
    if _EMU is None:
        import build_emu
        from neuray_amd import _lib
        _EMU = _lib.bind(build_emu.build())
        assert _EMU.neuray_is_device_build() == 0
    return _EMU


_EMU_BF16 = None


def emu_lib_bf16():
    """the bf16-operand variant of the kernels on the emulator"""
    global _EMU_BF16
    if _EMU_BF16 is None:
        import build_emu
        from neuray_amd import _lib
        _EMU_BF16 = _lib.bind(build_emu.build(variant='bf16'))
        assert _EMU_BF16.neuray_is_device_build() == 0 and _EMU_BF16.neuray_operand_precision() == 16
    return _EMU_BF16


def emu_engine(**kw):
    from neuray_amd.engine import RenderEngine
    return RenderEngine('cpu', _test_lib=emu_lib(), **kw)


def to_torch(d, device='cpu'):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in d.items()}


_EMU_BF16X3 = None


def emu_lib_bf16x3():
    """the split variant (hi + lo bf16 operands, three bf16 MFMAs per fp32 quad) on the emulator"""
    global _EMU_BF16X3
    if _EMU_BF16X3 is None:
        import build_emu
        from neuray_amd import _lib
        _EMU_BF16X3 = _lib.bind(build_emu.build(variant='bf16x3'))
        assert _EMU_BF16X3.neuray_is_device_build() == 0 and _EMU_BF16X3.neuray_operand_precision() == 48
    return _EMU_BF16X3
