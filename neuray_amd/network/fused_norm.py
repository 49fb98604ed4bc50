"""Fused InstanceNorm2d + activation (+ residual) + reflection padding for the per-image encoders, on the
kernels of csrc/nr_kernels_norm.h: `norm_act(bn, y, act, pad, res)` is

    reflect_pad(act(bn(y) [+ res]), pad)              bn: nn.InstanceNorm2d(affine=True, track_running_stats=False)

in two passes over the activation instead of PyTorch's four to six (MIOpen batch-norm, activation, add, activation,
reflection_pad2d), with a two-pass backward.  The result is handed to the next convolution already padded
(`conv_prepadded`); the un-padded activation is the interior view of the same buffer (`interior`).  On a device without the
kernels (plain CPU) the same function composes the PyTorch ops, so the modules have one forward."""

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
    def forward(ctx, x, gamma, beta, res, pad, act, eps):
        eng = _engine(x.device)
        x = x.contiguous().float()
        n, c, h, w = x.shape
        out = torch.empty(n, c, h + 2 * pad, w + 2 * pad, dtype=torch.float32, device=x.device)
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
            n, c, h, w, int(pad), int(act), float(eps), raw.data_ptr(), stats.data_ptr(), out.data_ptr(), eng._stream()))
        ctx.save_for_backward(x, out, stats, g)
        ctx.meta = (int(pad), int(act), res is not None)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, out, stats, g = ctx.saved_tensors
        pad, act, has_res = ctx.meta
        eng = _engine(x.device)
        n, c, h, w = x.shape
        d_out = d_out.contiguous().float()
        raw = eng.zero_scratch(2 * n * c)[:2 * n * c]
        dx = torch.empty_like(x)
        d_res = torch.empty_like(x) if has_res else None
        eng._check(eng.lib.neuray_inorm_backward(
            x.data_ptr(), out.data_ptr(), d_out.data_ptr(), stats.data_ptr(), g.data_ptr(), n, c, h, w, pad, act, raw.data_ptr(),
            dx.data_ptr(), d_res.data_ptr() if has_res else None, eng._stream()))
        sums = raw.view(n, c, 2).permute(2, 0, 1).sum(1)          # [2, c] straight out of the reduction: rows are the two gradients, no copies
        return dx, sums[1], sums[0], d_res, None, None, None


def norm_act(bn, y, act=None, pad=0, res=None):
    """-> reflect_pad(act(bn(y) [+ res]), pad) as one [n, c, h + 2 pad, w + 2 pad] tensor"""
    # (a padded plane of >= 2^23 elements - 2896 x 2896 - is beyond the kernels' fast division: such a map takes the composed PyTorch
    # ops below.  A missing library on a GPU still raises: the package never silently swaps its kernels for something else.)
    if (FUSED_NORM and bn.affine and not bn.track_running_stats and (y.shape[2] + 2 * pad) * (y.shape[3] + 2 * pad) < (1 << 23)
            and _engine(y.device) is not None):
        return _NormActFn.apply(y, bn.weight, bn.bias, res, pad, ACTS[act], bn.eps)
    z = bn(y)
    if res is not None:
        z = z + res
    z = F.relu(z) if act == 'relu' else (F.elu(z) if act == 'elu' else z)
    return F.pad(z, (pad, pad, pad, pad), mode='reflect') if pad else z


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


def conv_prepadded(conv, xp):
    """`conv` (an nn.Conv2d with padding_mode='reflect') on an input that already carries its reflection padding"""
    return F.conv2d(xp, conv.weight, conv.bias, conv.stride, 0, conv.dilation, conv.groups)
