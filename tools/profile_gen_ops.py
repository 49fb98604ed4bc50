"""Which Python call sites of a generalisation training step make COPIES of big tensors (a `.contiguous()` / `.clone()` / `.float()` that
is not a no-op)?  Monkey-patches those three methods, counts per (caller file:line, shape) over a few steps.
    python tools/profile_gen_ops.py"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                      # noqa: E402

counts = collections.Counter()
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'neuray_amd' in fr.filename or fr.filename.endswith('bench.py'):
            return '%s:%d %s' % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
    return '?'


orig_contig, orig_clone = torch.Tensor.contiguous, torch.Tensor.clone


def contiguous(self, *a, **k):
    if ON[0] and self.numel() >= (1 << 16) and not self.is_contiguous(*a, **k):
        counts[('contiguous-copy', site(), tuple(self.shape))] += 1
    return orig_contig(self, *a, **k)


def clone(self, *a, **k):
    if ON[0] and self.numel() >= (1 << 16):
        counts[('clone', site(), tuple(self.shape))] += 1
    return orig_clone(self, *a, **k)


torch.Tensor.contiguous, torch.Tensor.clone = contiguous, clone
dev = torch.device('cuda', 0)
model, opt, step = bench.gen_train_case(dev)
for _ in range(12):
    step()
torch.cuda.synchronize()
ON[0] = True
N = 3
for _ in range(N):
    step()
torch.cuda.synchronize()
for k, v in counts.most_common(40):
    print('%5.1f/step' % (v / N), k)
