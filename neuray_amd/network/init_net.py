"""Initial visibility-feature networks: the reference's `DepthInitNet`
(network/init_net.py:78-101) with its state_dict names, and the `name2init_net` registry (init_net.py:163-166).

forward(ref_imgs_info, src_imgs_info, is_train) -> initial ray_feats [rfn,32,h/4,w/4]:
  depth -> normalised inverse depth (elementwise, :63-76)
  get_diff_feats (:30-61): the cross-view consistency features - every pixel of every view lifted, projected into all
      views, colour / depth differences reduced to masked mean and variance - is ONE HIP kernel (neuray_diff_feats,
      the gather family of the render path) instead of the reference's [rfn, rfn*h*w, .] tensors
  res_net (ops.py:232-330) on [imgs, depth, diff_feats] (12 channels), depth_skip, conv_out: PyTorch convolutions
      (MIOpen), channels-last end to end - the kernel writes NHWC, which is the channels-last storage of [rfn,8,h,w].
`CostVolumeInitNet` (row f-3, :204-258) runs a frozen MVSNet (network/mvsnet.py) whose plane-sweep variance volume is the
HIP kernel neuray_warp_variance.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .encoders import ImageEncoder, _PreActBlock, _conv
from .mvsnet import MVSNet, load_ckpt


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


def construct_project_matrix(x_ratio, y_ratio, Ks, poses):
    """init_net.py:103-111: diag(x_ratio, y_ratio, 1) K [R|t] padded to 4x4.  The diagonal scaling is applied to K's rows with Python
    scalars - the same products `diag @ K` forms (its other terms are exact zeros) - instead of through a 3-element tensor uploaded from
    pageable host memory, which cost 1.2 ms of host time per call and a wait for the queue (profiles/r04_n_gen_host_profile.txt)."""
    scaled = torch.stack([Ks[:, 0] * float(x_ratio), Ks[:, 1] * float(y_ratio), Ks[:, 2]], 1)
    prj = scaled @ poses
    pad = torch.zeros(Ks.shape[0], 1, 4, device=Ks.device)
    pad[:, :, 3] = 1.0
    return torch.cat([prj, pad], 1)


def get_depth_vals(depth_range, dn):
    """init_net.py:162-168: dn planes uniform in inverse depth, the last exactly far"""
    near, far = depth_range[:, 0], depth_range[:, 1]
    interval = (1 / far - 1 / near) / (dn - 1)
    vals = 1 / (1 / near[:, None] + torch.arange(0, dn - 1, device=depth_range.device)[None, :] * interval[:, None])
    return torch.cat([vals, far[:, None]], 1)


def construct_cost_volume_with_src(ref_imgs_info, src_imgs_info, mvsnet, cost_volume_sn, imagenet_mean, imagenet_std, is_train,
                                   batch_num=None):
    """init_net.py:113-160 -> (softmax-ed cost volume [rfn,dn,h/4,w/4], regressed depth [rfn,h/4,w/4]).
    batch_num: reference views per pass through the 3-D U-Net.  The reference takes 2 (training) or 1 (evaluation) to fit
    its GPU; the views are independent (frozen batch norm), so with 288 GB of HBM the default here is all of them at once
    (8 x 800 x 800: 1.7 GB of variance volume)."""
    ref_imgs, src_imgs = ref_imgs_info['imgs'], src_imgs_info['imgs']
    rfn, _, h, w = ref_imgs.shape
    ratio, size = 1.0, None
    if not is_train and max(h, w) >= 800:          # evaluation at full size: the volume is built at reduced resolution
        if (h, w) == (768, 1024):
            size, ratio = (576, 768), 576 / 768
        elif (h, w) == (800, 800):
            size, ratio = (640, 640), 640 / 800
    if size is not None:
        ref_imgs, src_imgs = F.interpolate(ref_imgs, size, mode='bilinear'), F.interpolate(src_imgs, size, mode='bilinear')
    with torch.no_grad():
        ref_prj = construct_project_matrix(0.25 * ratio, 0.25 * ratio, ref_imgs_info['Ks'], ref_imgs_info['poses'])
        src_prj = construct_project_matrix(0.25 * ratio, 0.25 * ratio, src_imgs_info['Ks'], src_imgs_info['poses'])
        depth_vals = get_depth_vals(ref_imgs_info['depth_range'], cost_volume_sn)
        mvsnet.eval()
        cost_reg = mvsnet.construct_cost_volume_with_src((ref_imgs - imagenet_mean) / imagenet_std, (src_imgs - imagenet_mean) / imagenet_std,
                                                         ref_imgs_info['nn_ids'], ref_prj, src_prj, depth_vals, batch_num or rfn)
        cost_reg = torch.nan_to_num(cost_reg, nan=0.0, posinf=float('inf'), neginf=float('-inf'))      # cost_reg[isnan] = 0
        if size is not None:
            cost_reg = F.interpolate(cost_reg, (h // 4, w // 4), mode='bilinear')
        cost_reg = F.softmax(cost_reg, 1)
    depth = torch.sum(cost_reg * depth_vals[:, :, None, None], 1)                 # depth_regression, modules.py:66-71
    return cost_reg, depth


def _head(cin):
    """conv3x3 -> pre-activation residual block -> conv1x1 (init_net.py:222-243)"""
    return nn.Sequential(_conv(cin, 32, 3), _PreActBlock(32), _conv(32, 32, 1))


class CostVolumeInitNet(nn.Module):
    """init_net.py:204-258.  `mvsnet_ckpt`: path of the pretrained MVSNet weights the reference loads from
    'network/mvsnet/mvsnet_pl.ckpt' (not shipped here); None keeps the constructor's initialisation."""
    default_cfg = {'cost_volume_sn': 64, 'mvsnet_ckpt': None, 'mvsnet_batch_num': None}

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        self.mvsnet = MVSNet()
        if self.cfg['mvsnet_ckpt'] is not None:
            load_ckpt(self.mvsnet, self.cfg['mvsnet_ckpt'])
        for prm in self.mvsnet.parameters():
            prm.requires_grad = False
        self.register_buffer('imagenet_mean', torch.tensor([0.485, 0.456, 0.406], dtype=torch.float32)[None, :, None, None])
        self.register_buffer('imagenet_std', torch.tensor([0.229, 0.224, 0.225], dtype=torch.float32)[None, :, None, None])
        self.res_net = ImageEncoder(in_dim=3, blocks=(2, 3, 6), out_dim=32, width=32)           # ResUNetLight(out_dim=32)
        self.volume_conv2d = _head(self.cfg['cost_volume_sn'])
        self.depth_conv = _head(1)
        self.out_conv = _head(64 + 32)

    def forward(self, ref_imgs_info, src_imgs_info, is_train):
        cost_reg, depth = construct_cost_volume_with_src(ref_imgs_info, src_imgs_info, self.mvsnet, self.cfg['cost_volume_sn'],
                                                         self.imagenet_mean, self.imagenet_std, is_train, self.cfg['mvsnet_batch_num'])
        ref_feats = self.res_net(ref_imgs_info['imgs'])
        volume_feats = self.volume_conv2d(cost_reg)
        dr = ref_imgs_info['depth_range']
        near_inv, far_inv = (-1 / dr[:, 0])[:, None, None, None], (-1 / dr[:, 1])[:, None, None, None]
        depth = torch.clamp((-1 / torch.clamp(depth.unsqueeze(1), min=1e-5) - near_inv) / (far_inv - near_inv), min=0, max=1.0)
        volume_feats = torch.cat([volume_feats, self.depth_conv(depth)], 1)
        out = self.out_conv(torch.cat([ref_feats, volume_feats], 1)).contiguous(memory_format=torch.channels_last)
        if not is_train and out.is_cuda:
            from . import render_ops
            # evaluation: the verdicts of the input checks that ran on the device (neighbour index range, singular projections) are raised
            # before the features are used - a single inference call must not return a plausible volume built from clamped indices.
            # (Training drains them where the step already reads its loss back: render_ops.check_deferred_inputs.)
            render_ops.check_deferred_inputs(out.device, wait=True)
        return out


name2init_net = {'depth': DepthInitNet, 'cost_volume': CostVolumeInitNet}
