"""Which autograd nodes / Python call sites issue the small element-wise ops (fill_, copy_, add_, ...) of a generalisation training step?
Runs the step on the CPU through the fiber emulator at a small shape (the op sequence does not depend on the shape) under a
TorchDispatchMode: every watched aten call is attributed to the autograd node that is executing (backward) or to the nearest neuray_amd /
bench.py frame (forward).  TEST TOOLING: binds tests/emu, never used by the product.
    python tools/count_step_ops.py            (CPU, emulator)
    python tools/count_step_ops.py --hip      (the bench shape on cuda:0 through libneuray_hip.so; adds MB moved per site)"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench                                   # noqa: E402
from neuray_amd.network import render_ops as ro  # noqa: E402

WATCH = ('fill_', 'zero_', 'zeros', 'zeros_like', 'copy_', 'add', 'add_', 'mul', 'mul_', 'cat', 'sum', 'clone', '_to_copy', 'new_zeros',
         'index', 'sub', 'div', 'select_backward', 'slice_backward', 'constant_pad_nd', 'index_put_', 'empty_like')
counts = collections.Counter()
mbytes = collections.Counter()


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name.split('::')[-1]
        if name in WATCH:
            node = torch._C._current_autograd_node()
            if node is not None:
                where = 'backward of ' + node.name()
            else:
                where = '?'
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if ('neuray_amd' in fr.filename or fr.filename.endswith('bench.py')) and 'count_step_ops' not in fr.filename:
                        where = '%s:%d %s' % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
                        break
            numel = max([a.numel() for a in args if isinstance(a, torch.Tensor)] + [0])
            key = (name, where, 'big' if numel >= 4096 else 'small')
            counts[key] += 1
            mbytes[key] += numel * 4e-6
        return func(*args, **(kwargs or {}))


def main():
    if '--hip' in sys.argv:
        dev = torch.device('cuda', 0)
        model, opt, step = bench.gen_train_case(dev)
        for _ in range(5):
            step()
    else:
        from emu_util import emu_lib
        ro._TEST_LIB = emu_lib()
        ro._ENGINES.clear()
        dev = torch.device('cpu')
        model, opt, step = bench.gen_train_case(dev, h=64, w=96, rfn=3, extra_src=2, rays=24)
        model.cfg['depth_loss_coords_num'] = 64
        model._engine_test_lib = emu_lib()
        step()
    with Mode():
        step()
    total = sum(counts.values())
    print('%d watched aten calls in one step' % total)
    for (name, where, size), c in counts.most_common(70):
        print('%4d  %8.1f MB  %-16s %-6s %s' % (c, mbytes[(name, where, size)], name, size, where))


if __name__ == '__main__':
    main()
