"""Per-image encoders of the render pipeline: the image feature encoder
(`NeuralRayBaseRenderer.image_encoder`, reference network/ops.py:150-230, built as ResUNetLight(3, [1,2,6,4], 32,
inplanes=16) at renderer.py:58) and the visibility-feature encoder (network/vis_encoder.py:6-21).

These are convolution stacks, not per-ray work: they run through PyTorch's convolutions (MIOpen on ROCm) in
channels-last memory format, whose NHWC storage is exactly what the HIP gathers read, so the maps reach the render
kernels without a relayout pass (engine.prepare_views takes channels-last tensors as they are).  Parameter names and
shapes are the reference's, so its checkpoints load strictly.

Architecture (all convolutions reflect-padded, all norms InstanceNorm2d(affine=True, no running statistics)):
  stem   7x7/2 conv 3->16, IN, ReLU                                            -> 1/2 resolution
  layer1 1 residual block  16->32,  stride 2   (1x1/2 conv + IN on the skip)     -> 1/4
  layer2 2 residual blocks 32->64,  stride 2                                     -> 1/8
  layer3 6 residual blocks 64->128, stride 2                                     -> 1/16
  up3    bilinear x2 (align_corners) + 3x3 conv 128->64 + IN + ELU, concat layer2, 3x3 conv 128->64 + IN + ELU
  up2    bilinear x2 + 3x3 conv 64->32 + IN + ELU, concat layer1, 3x3 conv 64->32 + IN + ELU
  out    1x1 conv 32->32                                                        -> [n,32,h/4,w/4]
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused_norm
from .fused_norm import conv_prepadded, interior, norm_act, upsample2x_pad


# memory format the convolution stacks run in.  True: channels-last end to end (the NHWC storage the HIP gathers read, no relayout
# of the outputs); False: NCHW inside the stacks, one channels-last conversion of the two 32-channel outputs at the end.
CHANNELS_LAST = True


def _fmt(x):
    # (the fused norm kernels are NCHW, the layout MIOpen's fp32 Winograd convolutions run in natively: with them the stacks are
    # NCHW inside and the two 32-channel outputs are converted once)
    fused = fused_norm.FUSED_NORM and fused_norm._engine(x.device) is not None
    return x.contiguous(memory_format=torch.channels_last) if (CHANNELS_LAST and not fused) else x.contiguous()


def set_fused_norm(on):
    fused_norm.FUSED_NORM = bool(on)


def set_x3_conv(on):
    fused_norm.X3_CONV = bool(on)


def _inorm(c):
    return nn.InstanceNorm2d(c, track_running_stats=False, affine=True)


def _conv(cin, cout, k, stride=1, bias=False):
    return nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=bias, padding_mode='reflect')


class _ResBlock(nn.Module):
    """two 3x3 convs with IN, ReLU after the sum; names conv1/bn1/conv2/bn2/downsample as in the reference's BasicBlock"""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, cout, 3, stride), _inorm(cout)
        self.conv2, self.bn2 = _conv(cout, cout, 3), _inorm(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(_conv(cin, cout, 1, stride), _inorm(cout))

    def forward(self, xp, pad_out=1):
        """xp: the block's input carrying 1 pixel of reflection padding -> the block's output padded by `pad_out`"""
        a = norm_act(self.bn1, conv_prepadded(self.conv1, xp), 'relu', 1)
        skip = interior(xp, 1)
        if self.downsample is not None:
            skip = norm_act(self.downsample[1], conv_prepadded(self.downsample[0], skip), None, 0)
        return norm_act(self.bn2, conv_prepadded(self.conv2, a), 'relu', pad_out, res=skip)


class _ConvNormElu(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv, self.bn = _conv(cin, cout, 3, bias=True), _inorm(cout)

    def forward(self, x, prepadded=False, pad_out=0, tail=None):
        """x (with `prepadded`: already carrying the convolution's reflection padding) -> ELU(IN(conv(x))) padded by `pad_out`
        (with `tail`: followed by it on the channel axis)"""
        return norm_act(self.bn, conv_prepadded(self.conv, x) if prepadded else self.conv(x), 'elu', pad_out, tail=tail)


class _Up(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _ConvNormElu(cin, cout)

    def forward(self, x, pad_out=0, tail=None):
        # bilinear x2 (align_corners) written already padded for the 3x3 convolution: one kernel instead of up-sampling + reflection pad
        return self.conv(upsample2x_pad(x, 1), prepadded=True, pad_out=pad_out, tail=tail)


def _stage(cin, cout, n):
    return nn.Sequential(*[_ResBlock(cin if i == 0 else cout, cout, 2 if i == 0 else 1) for i in range(n)])


def _run_stage(stage, xp, pad_out):
    """the blocks of a stage on a padded input; every block hands the next one its output already padded"""
    for i, blk in enumerate(stage):
        xp = blk(xp, 1 if i + 1 < len(stage) else pad_out)
    return xp


def _join(skip, x):
    """pad the skip tensor to the upsampled size (odd sizes) and put it behind x on the channel axis"""
    dy, dx = x.shape[2] - skip.shape[2], x.shape[3] - skip.shape[3]
    if dy or dx:
        skip = F.pad(skip, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
    return torch.cat([x, skip], 1)


def _up_join_padded(up, x, skip_p):
    """[up(x), skip] on the channel axis, carrying 1 pixel of reflection padding for the 3x3 convolution that follows; skip_p is the
    skip tensor with that padding already.  Reflection padding is per channel, so when the sizes agree the padded halves are
    concatenated as they are (no interior view of the skip, no padding pass over the concatenation, and none in the backward); odd
    sizes (the skip is one pixel larger than twice x) take the zero-padded join of the reference and pad afterwards."""
    if 2 * x.shape[2] == skip_p.shape[2] - 2 and 2 * x.shape[3] == skip_p.shape[3] - 2:
        return up(x, pad_out=1, tail=skip_p)        # (the up branch's norm kernel writes its half of the concatenation in place)
    return F.pad(_join(interior(skip_p, 1), up(x)), (1, 1, 1, 1), mode='reflect')


class ImageEncoder(nn.Module):
    """state_dict-compatible with the reference's `image_encoder` (ResUNetLight(3, [1,2,6,4], 32, inplanes=16))."""

    def __init__(self, in_dim=3, blocks=(1, 2, 6), out_dim=32, width=16):
        super().__init__()
        self.conv1, self.bn1 = _conv(in_dim, width, 7, 2), _inorm(width)
        self.layer1 = _stage(width, 32, blocks[0])
        self.layer2 = _stage(32, 64, blocks[1])
        self.layer3 = _stage(64, 128, blocks[2])
        self.upconv3, self.iconv3 = _Up(128, 64), _ConvNormElu(128, 64)
        self.upconv2, self.iconv2 = _Up(64, 32), _ConvNormElu(64, 32)
        self.out_conv = nn.Conv2d(32, out_dim, 1)

    def forward(self, imgs):
        x = _fmt(imgs)
        xp = norm_act(self.bn1, self.conv1(x), 'relu', 1)                 # every activation travels with the next conv's padding
        s1p = _run_stage(self.layer1, xp, 1)
        s2p = _run_stage(self.layer2, s1p, 1)
        s3 = _run_stage(self.layer3, s2p, 0)
        x = self.iconv3(_up_join_padded(self.upconv3, s3, s2p), prepadded=True)
        x = self.iconv2(_up_join_padded(self.upconv2, x, s1p), prepadded=True)
        return self.out_conv(x).contiguous(memory_format=torch.channels_last)     # (no-op when the convs kept the format)


class _PreActBlock(nn.Module):
    """IN, ReLU, 3x3 conv twice, identity skip: the reference's ResidualBlock(32, 32) (network/ops.py:43-75)"""

    def __init__(self, c):
        super().__init__()
        self.conv = nn.Sequential(_inorm(c), nn.ReLU(), _conv(c, c, 3), _inorm(c), nn.ReLU(), _conv(c, c, 3))

    def forward(self, x):
        a = norm_act(self.conv[0], x, 'relu', 1)
        b = norm_act(self.conv[3], conv_prepadded(self.conv[2], a), 'relu', 1)
        return x + conv_prepadded(self.conv[5], b)


class DefaultVisEncoder(nn.Module):
    """network/vis_encoder.py:6-21: [img_feats, ray_feats] (64 ch) -> 3x3 conv -> 2 residual blocks -> 1x1 conv (32 ch)."""
    default_cfg = {}

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        self.out_conv = nn.Sequential(_conv(64, 32, 3), _PreActBlock(32), _PreActBlock(32), _conv(32, 32, 1))

    def forward(self, ray_feats, imgs_feats):
        x = _fmt(torch.cat([imgs_feats, ray_feats], 1))
        return self.out_conv(x).contiguous(memory_format=torch.channels_last)


name2vis_encoder = {'default': DefaultVisEncoder}
