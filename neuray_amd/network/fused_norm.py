"""Fused InstanceNorm2d + activation (+ residual) + reflection padding for the per-image encoders, on the
kernels of csrc/nr_kernels_norm.h: `norm_act(bn, y, act, pad, res)` is

    reflect_pad(act(bn(y) [+ res]), pad)              bn: nn.InstanceNorm2d(affine=True, track_running_stats=False)

in two passes over the activation instead of PyTorch's four to six (MIOpen batch-norm, activation, add, activation,
reflection_pad2d), with a two-pass backward.  The result is handed to the next convolution already padded
(`conv_prepadded`); the un-padded activation is the interior view of the same buffer (`interior`).  On a device without the
kernels (plain CPU) the same function composes the PyTorch ops, so the modules have one forward."""

import weakref

import numpy as np
import torch
import torch.nn.functional as F

ACTS = {None: 0, 'relu': 1, 'elu': 2}
FUSED_NORM = True          # False: always compose the PyTorch ops (A/B timing, tests)


def _engine(device):
    from . import render_ops
    if device.type != 'cuda' and render_ops._TEST_LIB is None:
        return None
    return render_ops.engine_for(device)


class _NormActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, res, pad, act, eps, tail):
        eng = _engine(x.device)
        x = x.contiguous().float()
        n, c, h, w = x.shape
        hp, wp = h + 2 * pad, w + 2 * pad
        ct = 0
        if tail is not None:                  # the result becomes the leading channels of [result, tail] (a channel concatenation)
            assert tail.dtype == torch.float32 and tuple(tail.shape[0:1] + tail.shape[2:]) == (n, hp, wp), (tuple(tail.shape), (n, hp, wp))
            ct = tail.shape[1]
        out = torch.empty(n, c + ct, hp, wp, dtype=torch.float32, device=x.device)
        if ct:
            out[:, c:].copy_(tail)
        stats = torch.empty(n * c, 2, dtype=torch.float32, device=x.device)
        raw = eng.zero_scratch(2 * n * c)
        g, b = gamma.detach().contiguous().float(), beta.detach().contiguous().float()
        rs = (0, 0, 0)
        if res is not None:
            if res.dtype != torch.float32 or res.stride(3) != 1:
                res = res.contiguous().float()
            assert tuple(res.shape) == (n, c, h, w), (tuple(res.shape), (n, c, h, w))
            rs = res.stride()[:3]
        eng._check(eng.lib.neuray_inorm_forward(
            x.data_ptr(), g.data_ptr(), b.data_ptr(), res.data_ptr() if res is not None else None, rs[0], rs[1], rs[2],
            n, c, h, w, int(pad), int(act), float(eps), raw.data_ptr(), stats.data_ptr(), out.data_ptr(), (c + ct) * hp * wp, eng._stream()))
        ctx.save_for_backward(x, out, stats, g)
        ctx.meta = (int(pad), int(act), res is not None, ct)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, out, stats, g = ctx.saved_tensors
        pad, act, has_res, ct = ctx.meta
        eng = _engine(x.device)
        n, c, h, w = x.shape
        d_out = d_out.contiguous().float()
        img = (c + ct) * (h + 2 * pad) * (w + 2 * pad)
        raw = eng.zero_scratch(2 * n * c)[:2 * n * c]
        d_affine = torch.empty(2, c, dtype=torch.float32, device=x.device)
        d_gamma, d_beta = d_affine[0], d_affine[1]
        dx = torch.empty_like(x)
        d_res = torch.empty_like(x) if has_res else None
        # (the affine parameters' gradients come out of the apply kernel - the planes' sums added over the images - instead of one more
        # PyTorch reduction per call: 30 launches of ~15 us per encoder pass)
        eng._check(eng.lib.neuray_inorm_backward(
            x.data_ptr(), out.data_ptr(), img, d_out.data_ptr(), img, stats.data_ptr(), g.data_ptr(), n, c, h, w, pad, act, raw.data_ptr(),
            dx.data_ptr(), d_res.data_ptr() if has_res else None, d_gamma.data_ptr(), d_beta.data_ptr(), eng._stream()))
        return dx, d_gamma, d_beta, d_res, None, None, None, (d_out[:, c:] if ct else None)


def norm_act(bn, y, act=None, pad=0, res=None, tail=None):
    """-> reflect_pad(act(bn(y) [+ res]), pad) as one [n, c, h + 2 pad, w + 2 pad] tensor; with `tail` (a tensor of that padded size):
    torch.cat([that, tail], 1), the normalised half written into the concatenation in place"""
    # (a padded plane of >= 2^23 elements - 2896 x 2896 - is beyond the kernels' fast division: such a map takes the composed PyTorch
    # ops below.  A missing library on a GPU still raises: the package never silently swaps its kernels for something else.)
    if (FUSED_NORM and bn.affine and not bn.track_running_stats and (y.shape[2] + 2 * pad) * (y.shape[3] + 2 * pad) < (1 << 23)
            and _engine(y.device) is not None):
        return _NormActFn.apply(y, bn.weight, bn.bias, res, pad, ACTS[act], bn.eps, tail)
    z = bn(y)
    if res is not None:
        z = z + res
    z = F.relu(z) if act == 'relu' else (F.elu(z) if act == 'elu' else z)
    z = F.pad(z, (pad, pad, pad, pad), mode='reflect') if pad else z
    return z if tail is None else torch.cat([z, tail], 1)


class _InteriorFn(torch.autograd.Function):
    """zp[:, :, pad:-pad, pad:-pad] as ONE autograd node whose backward is one zero-padding (fill + copy) - the two slice nodes of the
    plain indexing expression each allocate, zero and fill a full-size gradient (four launches per use, ~11 uses per encoder pass)"""

    @staticmethod
    def forward(ctx, zp, pad):
        ctx.pad = pad
        return zp[:, :, pad:zp.shape[2] - pad, pad:zp.shape[3] - pad]

    @staticmethod
    def backward(ctx, d):
        p = ctx.pad
        return F.pad(d, (p, p, p, p)), None


def interior(zp, pad):
    if not pad:
        return zp
    if zp.requires_grad and torch.is_grad_enabled():
        return _InteriorFn.apply(zp, pad)
    return zp[:, :, pad:zp.shape[2] - pad, pad:zp.shape[3] - pad]


X3_CONV = True             # False: every convolution through PyTorch / MIOpen (A/B timing, tests)
X3_WRW = True              # False: the weight gradient of the kernel's layers through the library
_PACKS = {}                # inference: id(weight) -> (weight._version, data_ptr, pack, weak reference) of the frozen weights


class _Conv3x3X3Fn(torch.autograd.Function):
    """3 x 3 stride-1 convolution of an input that carries its padding, on the K = 32 bf16 MFMA with exactly split operands
    (csrc/nr_kernels_conv2d.h, neuray_conv3x3_x3): forward and data gradient are the same kernel (the gradient is the full correlation with
    the flipped, transposed pack; both packs come out of one launch); the weight gradient is neuray_conv3x3_x3_wrw, straight from the NCHW
    tensors (the library's kernels want NHWC copies of both); the bias gradient is the library's reduction."""

    @staticmethod
    def forward(ctx, xp, weight, bias):
        eng = _engine(xp.device)
        xp = xp.contiguous()
        pack, pack_t = eng.conv3x3_x3_packs(weight, True, ctx.needs_input_grad[0])
        b = bias.detach().contiguous() if bias is not None else None
        ctx.save_for_backward(xp, weight)
        ctx.pack_t, ctx.has_bias = pack_t, bias is not None
        return eng.conv3x3_x3(xp, pack, b, weight.shape[0], 0)

    @staticmethod
    def backward(ctx, d_out):
        xp, weight = ctx.saved_tensors
        eng = _engine(xp.device)
        d_out = d_out.contiguous()
        dx = eng.conv3x3_x3(d_out, ctx.pack_t, None, weight.shape[1], 2) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1]:
            dw = eng.conv3x3_x3_wrw(d_out, xp) if X3_WRW else None          # (None: an odd padded width)
            if dw is None:
                dw = torch.ops.aten.convolution_backward(d_out, xp, weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # rows first: a reduction over the contiguous axis runs at memory speed, the generic (0, 2, 3) reduction of the library's bias
            # gradient at a fifth of it (75 us per layer at 9 x 800 x 800)
            db = d_out.flatten(2).sum(2).sum(0)
        return dx, dw, db


def _x3_conv_ok(conv, xp):
    w = conv.weight
    return (X3_CONV and tuple(w.shape[2:]) == (3, 3) and tuple(conv.stride) == (1, 1) and tuple(conv.dilation) == (1, 1) and conv.groups == 1
            and w.shape[0] % 32 == 0 and w.shape[1] % 32 == 0 and xp.dtype == torch.float32 and w.dtype == torch.float32
            and xp.is_contiguous() and xp.shape[2] >= 3 and xp.shape[3] >= 3 and xp.numel() * 4 < 0x7fffff00 and xp.shape[0] * w.shape[0] * xp.shape[2] * xp.shape[3] * 4 < 0x7fffff00
            and _engine(xp.device) is not None)


def conv_prepadded(conv, xp):
    """`conv` (an nn.Conv2d with padding_mode='reflect') on an input that already carries its reflection padding.  The 3 x 3 stride-1 layers
    with channel counts in multiples of 32 - every residual-block and decoder convolution of the encoders behind layer1's first - run on the
    split-operand bf16 MFMA kernel (fp32 grade; 1.45 x MIOpen's fp32 kernels on these shapes, profiles/r06_zz5_conv2d_probes.log); the strided,
    1 x 1, 7 x 7 and narrow ones stay with the library."""
    if _x3_conv_ok(conv, xp):
        if torch.is_grad_enabled() and (xp.requires_grad or conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad)):
            return _Conv3x3X3Fn.apply(xp, conv.weight, conv.bias)
        eng = _engine(xp.device)
        w = conv.weight
        hit = _PACKS.get(id(w))
        if hit is None or hit[0] != w._version or hit[1] != w.data_ptr() or hit[2].device != xp.device or hit[3]() is not w:
            hit = (w._version, w.data_ptr(), eng.conv3x3_x3_pack(w), weakref.ref(w))
            _PACKS[id(w)] = hit
        b = conv.bias.detach().contiguous() if conv.bias is not None else None
        return eng.conv3x3_x3(xp.contiguous(), hit[2], b, w.shape[0], 0)
    return F.conv2d(xp, conv.weight, conv.bias, conv.stride, 0, conv.dilation, conv.groups)


# ---- bilinear x2 up-sampling (align_corners) + reflection padding in one kernel, gather backward (csrc/nr_kernels_norm.h) ----------
_UP_TAPS = 8
_UP_TABLES = {}


def _up_axis(n_in, pad):
    """fp32 scale of PyTorch's align_corners up-sampling and, per input index, the padded output indices that read it with their
    weights (the transpose of the forward's `source = scale * output, truncated` in the same fp32 arithmetic; mirrored padding rows
    included) -> scale, cnt [n_in], idx [n_in, 8], wgt [n_in, 8]"""
    n_out = 2 * n_in
    scale = np.float32(n_in - 1) / np.float32(n_out - 1)
    op = np.arange(n_out + 2 * pad)
    o = np.abs(op - pad)
    o = np.where(o >= n_out, 2 * n_out - 2 - o, o)
    src = scale * o.astype(np.float32)
    i0 = src.astype(np.int32)
    l1 = src - i0.astype(np.float32)
    l0 = np.float32(1.0) - l1
    i1 = i0 + (i0 < n_in - 1)
    cnt, idx, wgt = np.zeros(n_in, np.int32), np.zeros((n_in, _UP_TAPS), np.int32), np.zeros((n_in, _UP_TAPS), np.float32)
    for k in range(op.shape[0]):
        for i, l in ((i0[k], l0[k]), (i1[k], l1[k])):
            if l != 0.0:
                assert cnt[i] < _UP_TAPS
                idx[i, cnt[i]], wgt[i, cnt[i]] = k, l
                cnt[i] += 1
    return float(scale), cnt, idx, wgt


def _up_tables(n_in, pad, device):
    key = (n_in, pad, str(device))
    hit = _UP_TABLES.get(key)
    if hit is None:
        scale, cnt, idx, wgt = _up_axis(n_in, pad)
        hit = _UP_TABLES[key] = (scale,) + tuple(torch.from_numpy(a).to(device) for a in (cnt, idx, wgt))
    return hit


class _Upsample2xPadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pad):
        eng = _engine(x.device)
        x = x.contiguous().float()
        n, c, h, w = x.shape
        out = torch.empty(n, c, 2 * h + 2 * pad, 2 * w + 2 * pad, dtype=torch.float32, device=x.device)
        sy, sx = _up_tables(h, pad, x.device)[0], _up_tables(w, pad, x.device)[0]
        eng._check(eng.lib.neuray_upsample2x_pad_forward(x.data_ptr(), n * c, h, w, int(pad), sy, sx, out.data_ptr(), eng._stream()))
        ctx.meta = (n, c, h, w, int(pad))
        return out

    @staticmethod
    def backward(ctx, d_out):
        n, c, h, w, pad = ctx.meta
        eng = _engine(d_out.device)
        d_out = d_out.contiguous().float()
        ty, tx = _up_tables(h, pad, d_out.device), _up_tables(w, pad, d_out.device)
        dx = torch.empty(n, c, h, w, dtype=torch.float32, device=d_out.device)
        eng._check(eng.lib.neuray_upsample2x_pad_backward(d_out.data_ptr(), n * c, h, w, pad, ty[1].data_ptr(), ty[2].data_ptr(), ty[3].data_ptr(),
                                                          tx[1].data_ptr(), tx[2].data_ptr(), tx[3].data_ptr(), dx.data_ptr(), eng._stream()))
        return dx, None


def upsample2x_pad(x, pad=0):
    """-> reflect_pad(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True), pad) as one [n, c, 2h + 2 pad, 2w + 2 pad]
    tensor (network/ops.py:150-230: the up-sampling in front of upconv3 / upconv2 and the padding of their 3 x 3 convolutions)"""
    h, w = x.shape[2:]
    if (FUSED_NORM and h >= 2 and 2 <= w <= 2047 and pad in (0, 1) and (2 * h + 2 * pad) * (2 * w + 2 * pad) < (1 << 23) and _engine(x.device) is not None):
        return _Upsample2xPadFn.apply(x, pad)
    z = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
    return F.pad(z, (pad, pad, pad, pad), mode='reflect') if pad else z
