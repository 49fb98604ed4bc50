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

    def forward(self, x):
        y = self.bn2(self.conv2(F.relu(self.bn1(self.conv1(x)))))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class _ConvNormElu(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv, self.bn = _conv(cin, cout, 3, bias=True), _inorm(cout)

    def forward(self, x):
        return F.elu(self.bn(self.conv(x)))


class _Up(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _ConvNormElu(cin, cout)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True))


def _stage(cin, cout, n):
    return nn.Sequential(*[_ResBlock(cin if i == 0 else cout, cout, 2 if i == 0 else 1) for i in range(n)])


def _join(skip, x):
    """pad the skip tensor to the upsampled size (odd sizes) and put it behind x on the channel axis"""
    dy, dx = x.shape[2] - skip.shape[2], x.shape[3] - skip.shape[3]
    if dy or dx:
        skip = F.pad(skip, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
    return torch.cat([x, skip], 1)


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
        x = imgs.contiguous(memory_format=torch.channels_last)
        x = F.relu(self.bn1(self.conv1(x)))
        s1 = self.layer1(x)
        s2 = self.layer2(s1)
        x = self.iconv3(_join(s2, self.upconv3(self.layer3(s2))))
        x = self.iconv2(_join(s1, self.upconv2(x)))
        return self.out_conv(x).contiguous(memory_format=torch.channels_last)     # (no-op when the convs kept the format)


class _PreActBlock(nn.Module):
    """IN, ReLU, 3x3 conv twice, identity skip: the reference's ResidualBlock(32, 32) (network/ops.py:43-75)"""

    def __init__(self, c):
        super().__init__()
        self.conv = nn.Sequential(_inorm(c), nn.ReLU(), _conv(c, c, 3), _inorm(c), nn.ReLU(), _conv(c, c, 3))

    def forward(self, x):
        return x + self.conv(x)


class DefaultVisEncoder(nn.Module):
    """network/vis_encoder.py:6-21: [img_feats, ray_feats] (64 ch) -> 3x3 conv -> 2 residual blocks -> 1x1 conv (32 ch)."""
    default_cfg = {}

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        self.out_conv = nn.Sequential(_conv(64, 32, 3), _PreActBlock(32), _PreActBlock(32), _conv(32, 32, 1))

    def forward(self, ray_feats, imgs_feats):
        x = torch.cat([imgs_feats, ray_feats], 1).contiguous(memory_format=torch.channels_last)
        return self.out_conv(x).contiguous(memory_format=torch.channels_last)


name2vis_encoder = {'default': DefaultVisEncoder}
