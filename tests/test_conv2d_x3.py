"""The encoders' 3 x 3 stride-1 convolution on the K = 32 bf16 MFMA with exactly split operands (csrc/nr_kernels_conv2d.h, neuray_conv3x3_x3;
reference network/ops.py:86-148,150-230, network/vis_encoder.py:6-21: nn.Conv2d(C_in, C_out, 3, padding_mode='reflect')) against PyTorch's
own convolution evaluated in float64 - forward on a pre-padded input, zero padding 1, and the data gradient as the full correlation with the
flipped pack - and the error of the split arithmetic next to the error of PyTorch's fp32 convolution.  CPU: the kernel on the emulator
(small shapes); `hip`: libneuray_hip.so at the encoder's own shapes."""
import pytest
import torch
import torch.nn.functional as F

from emu_util import emu_lib
from neuray_amd.network import render_ops as ro

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def dev(request):
    ro._ENGINES.clear()
    if request.param == 'emu':
        ro._TEST_LIB = emu_lib()
        yield 'cpu'
        ro._TEST_LIB = None
        ro._ENGINES.clear()
    else:
        ro._TEST_LIB = None
        yield 'cuda:0'


def _case(n, cin, cout, h, w, seed, bias=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g)
    x = x * torch.exp(torch.randn(n, cin, 1, 1, generator=g))          # channels of different scale, as behind an InstanceNorm + skip
    wgt = torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)
    b = torch.randn(cout, generator=g) if bias else None
    return x, wgt, b


def _rel(a, ref):
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max())


# (n, cin, cout, h, w): rows that wrap inside a 16-position tile, several bands (w > 64), more than one channel block and channel group
SMALL = [(2, 32, 32, 7, 9), (1, 64, 32, 5, 70), (1, 32, 64, 19, 6), (3, 64, 32, 4, 4), (2, 32, 128, 6, 8)]      # (128 outputs: eight-wave workgroups)
LARGE = [(9, 128, 128, 52, 52), (9, 64, 64, 102, 102), (9, 32, 32, 202, 202), (9, 128, 64, 102, 102), (9, 64, 32, 202, 202), (2, 96, 64, 191, 254)]


def _shapes(dev):
    return SMALL if dev == 'cpu' else SMALL + LARGE


def test_forward_valid_and_padded(dev):
    eng = ro.engine_for(torch.device(dev))
    for i, (n, cin, cout, h, w) in enumerate(_shapes(dev)):
        x, wgt, b = _case(n, cin, cout, h, w, i, bias=i % 2 == 0)
        xd, wd, bd = x.to(dev), wgt.to(dev), (b.to(dev) if b is not None else None)
        pack = eng.conv3x3_x3_pack(wd)
        for pad in (0, 1):
            ref = F.conv2d(x.double(), wgt.double(), b.double() if b is not None else None, padding=pad)
            got = eng.conv3x3_x3(xd, pack, bd, cout, pad=pad)
            assert tuple(got.shape) == tuple(ref.shape)
            e_x3 = _rel(got, ref)
            e_f32 = _rel(F.conv2d(x, wgt, b, padding=pad), ref)
            # fp32 grade: the split products are exact to 2^-23, the accumulation is fp32 in 32-channel blocks
            assert e_x3 < 3e-6, (n, cin, cout, h, w, pad, e_x3, e_f32)
            assert e_x3 < 4.0 * e_f32 + 1e-7, (n, cin, cout, h, w, pad, e_x3, e_f32)


def test_data_gradient_is_the_full_correlation_with_the_flipped_pack(dev):
    eng = ro.engine_for(torch.device(dev))
    for i, (n, cin, cout, h, w) in enumerate(_shapes(dev)[:7]):
        x, wgt, _ = _case(n, cin, cout, h, w, 10 + i)
        if cout % 32:
            continue                                                   # (the gradient's contraction runs over C_out: a multiple of 32)
        xr = x.double().requires_grad_(True)
        y = F.conv2d(xr, wgt.double())
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(i), dtype=torch.float64)
        ref, = torch.autograd.grad(y, xr, dy)
        pack_t = eng.conv3x3_x3_pack(wgt.to(dev), transpose_flip=True)
        got = eng.conv3x3_x3(dy.float().to(dev).contiguous(), pack_t, None, cin, pad=2)
        assert tuple(got.shape) == tuple(x.shape)
        assert _rel(got, ref) < 3e-6


def test_weight_gradient(dev):
    """neuray_conv3x3_x3_wrw against autograd of a float64 convolution: K blocks that straddle rows and end inside the last one, several (co, ci)
    blocks, more K blocks than one wave takes (the LDS and workspace reductions)"""
    eng = ro.engine_for(torch.device(dev))
    shapes = [(2, 32, 32, 7, 10), (1, 64, 32, 12, 52), (3, 32, 64, 9, 18)] + ([] if dev == 'cpu' else [(9, 128, 128, 52, 52), (9, 64, 32, 202, 202), (4, 128, 64, 102, 102)])
    for i, (n, cin, cout, hp, wp) in enumerate(shapes):
        x, wgt, _ = _case(n, cin, cout, hp, wp, 20 + i)
        wr = wgt.double().requires_grad_(True)
        y = F.conv2d(x.double(), wr)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(i), dtype=torch.float64)
        ref, = torch.autograd.grad(y, wr, dy)
        got = eng.conv3x3_x3_wrw(dy.float().to(dev).contiguous(), x.to(dev))
        assert got is not None and tuple(got.shape) == tuple(wgt.shape)
        lib = torch.ops.aten.convolution_backward(dy.float(), x, wgt, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        e_x3, e_f32 = _rel(got, ref), _rel(lib, ref)
        assert e_x3 < 3e-6 and e_x3 < 4.0 * e_f32 + 2e-7, (n, cin, cout, hp, wp, e_x3, e_f32)
    assert eng.conv3x3_x3_wrw(torch.zeros(1, 32, 5, 9, device=dev), torch.zeros(1, 32, 7, 11, device=dev)) is None      # odd padded width


def test_bad_shapes_are_refused(dev):
    eng = ro.engine_for(torch.device(dev))
    assert eng.lib.neuray_conv3x3_x3_pack_bytes(16, 32) == -1 and eng.lib.neuray_conv3x3_x3_pack_bytes(32, 48) == -1
    assert eng.lib.neuray_conv3x3_x3_pack_bytes(64, 32) == 9 * 64 * 32 * 6
    x = torch.zeros(1, 32, 2, 8, device=dev)
    pack = eng.conv3x3_x3_pack(torch.zeros(32, 32, 3, 3, device=dev))
    with pytest.raises(RuntimeError):
        eng.conv3x3_x3(x, pack, None, 32, pad=0)                      # two rows: no valid output row


def test_conv_prepadded_routes_the_encoder_layers_through_the_kernel(dev):
    """fused_norm.conv_prepadded: the 3 x 3 stride-1 layers with channel counts in multiples of 32 take the kernel - under autograd as one node
    whose data gradient is the kernel again and whose weight / bias gradients are the library's - and everything else PyTorch's convolution"""
    import torch.nn as nn
    from neuray_amd.network import fused_norm as fn
    ro.engine_for(torch.device(dev))
    torch.manual_seed(3)
    conv = nn.Conv2d(32, 64, 3, 1, 1, bias=True, padding_mode='reflect').to(dev)
    xp = torch.randn(2, 32, 9, 11, device=dev, requires_grad=True)
    y = fn.conv_prepadded(conv, xp)
    assert type(y.grad_fn).__name__.startswith('_Conv3x3X3Fn')
    dy = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, (xp, conv.weight, conv.bias), dy)
    xr = xp.detach().double().cpu().requires_grad_(True)
    wr, br = conv.weight.detach().double().cpu().requires_grad_(True), conv.bias.detach().double().cpu().requires_grad_(True)
    yr = F.conv2d(xr, wr, br)
    rx, rw, rb = torch.autograd.grad(yr, (xr, wr, br), dy.double().cpu())
    for got, ref in ((y, yr), (gx, rx), (gw, rw), (gb, rb)):          # (gw: the kernel's weight gradient; an odd padded width takes the library's)
        assert _rel(got.detach(), ref.detach()) < 3e-6
    with torch.no_grad():                                             # inference: the pack of the frozen weight is cached, and dropped when it changes
        y0 = fn.conv_prepadded(conv, xp.detach())
        assert _rel(y0, yr.detach()) < 3e-6 and id(conv.weight) in fn._PACKS
        conv.weight.mul_(2.0)
        y1 = fn.conv_prepadded(conv, xp.detach())
        assert _rel(y1 - conv.bias.view(1, -1, 1, 1), 2.0 * (yr.detach() - br.detach().view(1, -1, 1, 1))) < 3e-6
    strided = nn.Conv2d(32, 64, 3, 2, 1, bias=False, padding_mode='reflect').to(dev)
    assert 'Conv3x3X3' not in type(fn.conv_prepadded(strided, xp).grad_fn).__name__
    narrow = nn.Conv2d(16, 32, 3, 1, 1, bias=False, padding_mode='reflect').to(dev)
    assert 'Conv3x3X3' not in type(fn.conv_prepadded(narrow, torch.randn(1, 16, 8, 8, device=dev, requires_grad=True)).grad_fn).__name__
