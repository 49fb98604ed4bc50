"""host cost per call of the fused norm path vs the PyTorch composition (tiny tensors: the GPU work is negligible)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
from neuray_amd.network import fused_norm
dev = torch.device('cuda', 0)
bn = nn.InstanceNorm2d(32, affine=True).to(dev)
y = torch.randn(2, 32, 16, 16, device=dev)
big = torch.randn(9, 32, 200, 200, device=dev)
def comp(t):
    return F.pad(F.relu(bn(t)), (1, 1, 1, 1), mode='reflect')
for name, fn in (('fused', lambda t: fused_norm.norm_act(bn, t, 'relu', 1)), ('composed', comp)):
    for t, tag in ((y, 'tiny'), (big, '9x32x200x200')):
        with torch.no_grad():
            for _ in range(5): fn(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200): fn(t)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print('%-9s %-14s host %.1f us/call, with drain %.1f us/call' % (name, tag, 1e6 * (t1 - t0) / 200, 1e6 * (t2 - t0) / 200))
import cProfile, pstats
with torch.no_grad():
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): fused_norm.norm_act(bn, y, 'relu', 1)
    pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
