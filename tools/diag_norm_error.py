"""Error of the fused InstanceNorm path and of PyTorch's own fp32 instance norm against a float64 evaluation, on planes with a large
mean (what tests/test_fused_norm.py feeds): which of the two the test tolerance has to absorb.   python tools/diag_norm_error.py"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuray_amd.network import fused_norm      # noqa: E402

dev = torch.device('cuda', 0)
for shape in ((1, 3, 182, 181), (2, 2, 180, 181), (3, 2, 104, 152), (1, 16, 40, 50), (2, 8, 400, 400)):
    n, c, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    bn = nn.InstanceNorm2d(c, affine=True).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.3)
    y = (torch.randn(n, c, h, w, generator=g) * 2 + 3 * torch.randn(1, c, 1, 1, generator=g)).to(dev)
    with torch.no_grad():
        ours = fused_norm.norm_act(bn, y, 'relu', 0)
        theirs = F.relu(bn(y))
        y64 = y.double()
        m = y64.mean((2, 3), keepdim=True)
        v = y64.var((2, 3), unbiased=False, keepdim=True)
        ref = F.relu((y64 - m) / torch.sqrt(v + bn.eps) * bn.weight.double().view(1, c, 1, 1) + bn.bias.double().view(1, c, 1, 1))
    print(shape, 'fused vs f64 %.2e   torch fp32 vs f64 %.2e   fused vs torch %.2e   max |out| %.1f' % (
        float((ours.double() - ref).abs().max()), float((theirs.double() - ref).abs().max()), float((ours - theirs).abs().max()),
        float(ref.abs().max())))
