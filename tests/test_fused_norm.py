"""csrc/nr_kernels_norm.h: fused InstanceNorm2d + activation (+ residual) + reflection padding of the per-image encoders
 against the PyTorch composition it replaces (network/ops.py:43-75,150-230: `conv -> norm -> relu
[-> + skip -> relu]`, `conv -> norm -> elu`, followed by the next convolution's padding_mode='reflect'), forward and backward."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from emu_util import emu_lib
from neuray_amd.network import fused_norm
from neuray_amd.network import render_ops as ro

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


@pytest.fixture
def device(request):
    backend = request.param
    if backend == 'emu':
        ro._TEST_LIB = emu_lib()
        ro._ENGINES.clear()
        yield torch.device('cpu')
        ro._TEST_LIB = None
        ro._ENGINES.clear()
    else:
        yield torch.device('cuda', 0)


def composed(bn, y, act, pad, res):
    z = bn(y)
    if res is not None:
        z = z + res
    z = F.relu(z) if act == 'relu' else (F.elu(z) if act == 'elu' else z)
    return F.pad(z, (pad, pad, pad, pad), mode='reflect') if pad else z


@pytest.mark.parametrize('device', BACKENDS, indirect=True)
@pytest.mark.parametrize('act,pad,with_res,shape', [
    ('relu', 1, False, (2, 5, 9, 11)), ('relu', 1, True, (3, 4, 8, 6)), (None, 0, False, (2, 3, 7, 5)), ('elu', 0, False, (1, 6, 10, 13)),
    ('relu', 0, True, (2, 4, 6, 6)), ('elu', 1, True, (2, 2, 2, 3)), ('relu', 1, False, (1, 16, 40, 50)),
    # planes at the encoders' training resolutions: several chunks per plane, atomically added partial sums
    ('relu', 1, True, (3, 2, 104, 152)), ('elu', 1, True, (2, 2, 180, 181)), ('relu', 0, False, (1, 3, 182, 181)),
])
def test_fused_norm_act_equals_the_pytorch_composition(device, act, pad, with_res, shape):
    g = torch.Generator().manual_seed(sum(shape) + pad)
    n, c, h, w = shape
    bn = nn.InstanceNorm2d(c, affine=True, track_running_stats=False)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.3)
    bn = bn.to(device)
    y0 = (torch.randn(n, c, h, w, generator=g) * 2 + 3 * torch.randn(1, c, 1, 1, generator=g)).to(device)     # planes with a large mean
    base = torch.randn(n, c, h + 2, w + 2, generator=g).to(device)
    dz = torch.randn(n, c, h + 2 * pad, w + 2 * pad, generator=g).to(device)
    outs = []
    for fused in (True, False):
        fused_norm.FUSED_NORM = fused
        try:
            # the composition is evaluated in float64: on a one-image batch with planes far from zero mean MIOpen's own fp32 instance
            # norm is 4e-4 off the float64 value (tools/diag_norm_error.py, profiles/r05_s_norm_error.log; the kernels here: 2e-7)
            dt = torch.float32 if fused else torch.float64
            mod = bn if fused else bn.double()
            y = y0.detach().to(dt).clone().requires_grad_(True)
            b = base.detach().to(dt).clone().requires_grad_(True)
            res = b[:, :, 1:-1, 1:-1] if with_res else None             # a strided view, as the interior of a padded buffer is
            for p_ in mod.parameters():
                p_.grad = None
            out = fused_norm.norm_act(mod, y, act, pad, res) if fused else composed(mod, y, act, pad, res)
            (out * dz.to(dt)).sum().backward()
            outs.append((out.detach().float().cpu(), y.grad.float().cpu(), mod.weight.grad.float().cpu().clone(),
                         mod.bias.grad.float().cpu().clone(), b.grad.float().cpu() if with_res else None))
        finally:
            fused_norm.FUSED_NORM = True
    (o1, gy1, gw1, gb1, gr1), (o0, gy0, gw0, gb0, gr0) = outs
    assert o1.shape == (n, c, h + 2 * pad, w + 2 * pad)
    assert float((o1 - o0).abs().max()) <= 2e-5
    scale = lambda t: max(1.0, float(t.abs().max()))                      # noqa: E731
    assert float((gy1 - gy0).abs().max()) <= 2e-4 * scale(gy0)
    assert float((gw1 - gw0).abs().max()) <= 2e-4 * scale(gw0) and float((gb1 - gb0).abs().max()) <= 2e-4 * scale(gb0)
    if with_res:
        assert float((gr1 - gr0).abs().max()) <= 1e-5 * scale(gr0)


@pytest.mark.parametrize('device', BACKENDS, indirect=True)
def test_encoders_with_and_without_the_fused_kernels(device):
    """the whole image_encoder + vis_encoder, forward and parameter gradients, fused path vs the PyTorch composition"""
    from neuray_amd.network import encoders
    from test_encoders import fill_by_name
    torch.manual_seed(0)
    enc, vis = encoders.ImageEncoder().to(device), encoders.DefaultVisEncoder({}).to(device)
    fill_by_name(enc), fill_by_name(vis)
    imgs = torch.rand(2, 3, 48, 64, device=device)
    ray0 = torch.randn(2, 32, 12, 16, device=device)
    res = []
    for fused in (True, False):
        encoders.set_fused_norm(fused)
        try:
            for p_ in list(enc.parameters()) + list(vis.parameters()):
                p_.grad = None
            f = enc(imgs)
            r = vis(ray0, f)
            (f.square().mean() + r.square().mean()).backward()
            res.append((f.detach().cpu(), r.detach().cpu(), [p_.grad.cpu().clone() for p_ in list(enc.parameters()) + list(vis.parameters())]))
        finally:
            encoders.set_fused_norm(True)
    (f1, r1, g1), (f0, r0, g0) = res
    assert f1.shape == (2, 32, 12, 16) and float((f1 - f0).abs().max()) <= 1e-4 * max(1.0, float(f0.abs().max()))
    assert float((r1 - r0).abs().max()) <= 1e-4 * max(1.0, float(r0.abs().max()))
    # (a convolution bias in front of an InstanceNorm has a mathematically zero gradient - both sides hold rounding noise there -
    # so every tensor is measured against the larger of its own scale and 1e-3 of the largest gradient of the model)
    top = max(float(b.abs().max()) for b in g0)
    worst = max(float((a - b).abs().max()) / max(1e-3 * top, float(b.abs().max())) for a, b in zip(g1, g0))
    assert worst <= 2e-3, worst


@pytest.mark.parametrize('device', BACKENDS, indirect=True)
@pytest.mark.parametrize('pad', [0, 1])
@pytest.mark.parametrize('shape', [(2, 3, 2, 2), (1, 4, 5, 7), (2, 2, 25, 38), (1, 3, 50, 50), (1, 2, 100, 75)])
def test_upsample2x_pad_equals_interpolate_then_reflection_pad(device, pad, shape):
    """network/ops.py:150-230 (upconv3 / upconv2): F.interpolate(scale_factor=2, bilinear, align_corners=True) + the reflection padding
    of the next convolution, forward and gradient (PyTorch's backward sums with atomics: compared to summation-order tolerance)"""
    g = torch.Generator().manual_seed(sum(shape) + pad)
    n, c, h, w = shape
    x0 = torch.randn(n, c, h, w, generator=g).to(device)
    dz = torch.randn(n, c, 2 * h + 2 * pad, 2 * w + 2 * pad, generator=g).to(device)
    outs = []
    for fused in (True, False):
        fused_norm.FUSED_NORM = fused
        try:
            x = x0.clone().requires_grad_(True)
            out = fused_norm.upsample2x_pad(x, pad)
            assert (type(out.grad_fn).__name__ == '_Upsample2xPadFnBackward') == fused
            (out * dz).sum().backward()
            outs.append((out.detach().cpu(), x.grad.cpu()))
        finally:
            fused_norm.FUSED_NORM = True
    (o1, g1), (o0, g0) = outs
    assert o1.shape == (n, c, 2 * h + 2 * pad, 2 * w + 2 * pad)
    assert float((o1 - o0).abs().max()) <= 2e-6 * max(1.0, float(o0.abs().max()))
    assert float((g1 - g0).abs().max()) <= 1e-5 * max(1.0, float(g0.abs().max()))


def test_upsample_gather_tables_are_the_transpose_of_the_forward():
    """every (padded output, input) pair of the forward appears exactly once in the backward tables, with the forward's weight, for
    every size the encoders can meet"""
    for n_in in list(range(2, 70)) + [100, 101, 200, 203, 400, 511]:
        for pad in (0, 1):
            scale, cnt, idx, wgt = fused_norm._up_axis(n_in, pad)
            n_out = 2 * n_in
            dense = np.zeros((n_out + 2 * pad, n_in), np.float64)
            for op in range(n_out + 2 * pad):
                o = abs(op - pad)
                o = 2 * n_out - 2 - o if o >= n_out else o
                src = np.float32(scale) * np.float32(o)
                i0 = int(src)
                l1 = np.float32(src - np.float32(i0))
                dense[op, i0] += float(np.float32(1.0) - l1)
                dense[op, min(i0 + 1, n_in - 1)] += float(l1)
            back = np.zeros_like(dense)
            for i in range(n_in):
                assert 1 <= cnt[i] <= 8
                for k in range(cnt[i]):
                    back[idx[i, k], i] += float(wgt[i, k])
            assert np.array_equal(back, dense), (n_in, pad)
            assert np.allclose(dense.sum(1), 1.0, atol=1e-6)


@pytest.mark.parametrize('device', BACKENDS, indirect=True)
@pytest.mark.parametrize('act,pad,shape,ct', [('elu', 1, (2, 3, 8, 10), 5), ('relu', 0, (1, 4, 6, 7), 2), ('elu', 1, (3, 6, 12, 9), 6)])
def test_norm_act_writes_its_half_of_a_channel_concatenation(device, act, pad, shape, ct):
    """norm_act(..., tail=t) = torch.cat([norm_act(...), t], 1): the decoder joins of the image encoder (network/ops.py:150-230,
    iconv3 / iconv2 inputs), outputs and every gradient (y, gamma, beta, tail)"""
    g = torch.Generator().manual_seed(sum(shape) + ct)
    n, c, h, w = shape
    bn = nn.InstanceNorm2d(c, affine=True, track_running_stats=False)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(c, generator=g) * 0.3)
    bn = bn.to(device)
    y0 = torch.randn(n, c, h, w, generator=g).to(device)
    t0 = torch.randn(n, ct, h + 2 * pad, w + 2 * pad, generator=g).to(device)
    dz = torch.randn(n, c + ct, h + 2 * pad, w + 2 * pad, generator=g).to(device)
    outs = []
    for fused in (True, False):
        fused_norm.FUSED_NORM = fused
        try:
            y, t = y0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
            for p_ in bn.parameters():
                p_.grad = None
            out = fused_norm.norm_act(bn, y, act, pad, None, tail=t * 1.0)
            (out * dz).sum().backward()
            outs.append([v.detach().cpu() for v in (out, y.grad, t.grad, bn.weight.grad.clone(), bn.bias.grad.clone())])
        finally:
            fused_norm.FUSED_NORM = True
    for a, b, tol in zip(outs[0], outs[1], (2e-5, 2e-4, 0.0, 2e-4, 2e-4)):
        assert a.shape == b.shape and float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
