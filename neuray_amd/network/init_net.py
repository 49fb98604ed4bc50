"""Initial visibility-feature networks: the reference's `DepthInitNet`
(network/init_net.py:78-101) with its state_dict names, and the `name2init_net` registry (init_net.py:163-166).

forward(ref_imgs_info, src_imgs_info, is_train) -> initial ray_feats [rfn,32,h/4,w/4]:
  depth -> normalised inverse depth (elementwise, :63-76)
  get_diff_feats (:30-61): the cross-view consistency features - every pixel of every view lifted, projected into all
      views, colour / depth differences reduced to masked mean and variance - is ONE HIP kernel (neuray_diff_feats,
      the gather family of the render path) instead of the reference's [rfn, rfn*h*w, .] tensors
  res_net (ops.py:232-330) on [imgs, depth, diff_feats] (12 channels), depth_skip, conv_out: PyTorch convolutions
      (MIOpen), channels-last end to end - the kernel writes NHWC, which is the channels-last storage of [rfn,8,h,w].
`CostVolumeInitNet` (MVSNet, row f-3) is not built.
"""
import torch
import torch.nn as nn

from .encoders import ImageEncoder


class ResEncoder(ImageEncoder):
    """ops.py:232-330: the encoder of network/encoders.py with two blocks per stage, 32-wide, a 12-channel 8x8/2 stem
    (reflect padding 2) - same parameter names."""

    def __init__(self):
        super().__init__(in_dim=12, blocks=(2, 2, 2), out_dim=32, width=32)
        self.conv1 = nn.Conv2d(12, 32, kernel_size=8, stride=2, padding=2, bias=False, padding_mode='reflect')


def extract_depth_for_init(ref_imgs_info):
    """init_net.py:63-76"""
    dr, depth = ref_imgs_info['depth_range'], ref_imgs_info['depth']
    near_inv, far_inv = (-1 / dr[:, 0])[:, None, None, None], (-1 / dr[:, 1])[:, None, None, None]
    return torch.clamp((-1 / torch.clamp(depth, min=1e-5) - near_inv) / (far_inv - near_inv), min=0, max=1.0)


def get_diff_feats(ref_imgs_info, depth_in):
    """init_net.py:30-61 on the HIP kernel.  depth_in: normalised inverse depth [rfn,1,h,w]."""
    from . import render_ops
    dr = ref_imgs_info['depth_range']
    near_inv, far_inv = (-1 / dr[:, 0])[:, None, None, None], (-1 / dr[:, 1])[:, None, None, None]
    depth = -1 / (depth_in * (far_inv - near_inv) + near_inv)
    return render_ops.engine_for(depth.device).diff_feats(ref_imgs_info, depth)


class DepthInitNet(nn.Module):
    default_cfg = {}

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        self.res_net = ResEncoder()
        self.depth_skip = nn.Sequential(nn.Conv2d(1, 8, 2, 2), nn.ReLU(True), nn.Conv2d(8, 16, 2, 2))
        self.conv_out = nn.Conv2d(16 + 32, 32, 1, 1)

    def forward(self, ref_imgs_info, src_imgs_info, is_train):
        depth = extract_depth_for_init(ref_imgs_info)
        diff_feats = get_diff_feats(ref_imgs_info, depth)          # data only: no gradient flows into images / depth
        x = torch.cat([ref_imgs_info['imgs'], depth, diff_feats], 1).contiguous(memory_format=torch.channels_last)
        feats = self.res_net(x)
        depth_feats = self.depth_skip(depth.contiguous(memory_format=torch.channels_last))
        return self.conv_out(torch.cat([depth_feats, feats], 1)).contiguous(memory_format=torch.channels_last)


name2init_net = {'depth': DepthInitNet}
