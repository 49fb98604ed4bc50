"""MVSNet as the cost-volume init net uses it (reference network/mvsnet/mvsnet.py,
network/mvsnet/modules.py): frozen, eval mode, `construct_cost_volume_with_src` only.

  feature            8 conv + activated-batch-norm layers, 3 -> 32 channels at 1/4 resolution (mvsnet.py:7-30)
  variance volume    ONE HIP kernel (neuray_warp_variance) instead of a [B,32,D,h,w] warp per source view plus two
                     running sums (mvsnet.py:186-203, modules.py:25-64)
  cost_regularization  3-D U-Net, 32 -> 1 channel (mvsnet.py:32-75): PyTorch Conv3d / ConvTranspose3d (MIOpen), except - frozen and
                     under no_grad, as the init net runs it - its first and last layer, which are HIP kernels (csrc/nr_kernels_conv3d.h)

The reference builds it with `inplace_abn.ABN` (network/init_net.py:121): F.batch_norm with the running statistics
followed by leaky_relu(0.01); `ActivatedBatchNorm` below has that module's parameter and buffer names (`weight`, `bias`,
`running_mean`, `running_var`), so `mvsnet_pl.ckpt` loads through `load_ckpt` as in the reference.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


FUSED_ABN = True          # False: always compose the PyTorch ops (A/B timing, tests)
FAST_CONV3D = True        # False: conv1 ... conv4 of the cost regularisation through their modules (A/B timing)


class ActivatedBatchNorm(nn.Module):
    """inplace_abn.InPlaceABN as the reference instantiates it (network/mvsnet/modules.py:7-23, norm_act = InPlaceABN): batch norm + leaky
    ReLU(0.01), same parameter / buffer names.  Like that class it may work IN PLACE on its input (the fused evaluation path below does; the
    composed PyTorch path returns a new tensor): hand it a tensor you do not need afterwards, as `self.bn(self.conv(x))` does."""

    def __init__(self, num_features, eps=1e-5, slope=0.01):
        super().__init__()
        self.eps, self.slope = eps, slope
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))

    def _folded(self, device):
        """scale = gamma / sqrt(var + eps), shift = beta - mean * scale of the frozen module; rebuilt when a parameter or buffer changes"""
        src = (self.weight, self.bias, self.running_mean, self.running_var)
        stamp = tuple((t.data_ptr(), t._version) for t in src) + (str(device),)
        hit = self.__dict__.get('_fold')
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                scale = (self.weight / torch.sqrt(self.running_var + self.eps)).float().contiguous()
                shift = (self.bias - self.running_mean * scale).float().contiguous()
            hit = self.__dict__['_fold'] = (stamp, scale, shift)
        return hit[1], hit[2]

    def forward(self, x):
        # Evaluation mode with nothing to differentiate (how the cost-volume init net runs MVSNet: frozen, under no_grad,
        # init_net.py:121-160): batch norm on the running statistics + leaky ReLU as ONE in-place pass over the convolution's output
        # (neuray_scale_shift_leaky) instead of two kernels and two round trips.  Everything else takes the PyTorch ops.
        if (FUSED_ABN and not self.training and not torch.is_grad_enabled() and x.dtype == torch.float32 and x.dim() >= 3
                and x.is_contiguous() and x.shape[0] * x.shape[1] <= 65535):
            from . import render_ops
            if x.device.type == 'cuda' or render_ops._TEST_LIB is not None:
                scale, shift = self._folded(x.device)
                return render_ops.engine_for(x.device).scale_shift_leaky_(x, scale, shift, self.slope)
        x = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, self.training, 0.1, self.eps)
        return F.leaky_relu(x, self.slope)


class _ConvAbn(nn.Module):
    def __init__(self, conv, cout):
        super().__init__()
        self.conv, self.bn = conv, ActivatedBatchNorm(cout)

    def forward(self, x):
        return self.bn(self.conv(x))


def _c2(cin, cout, k=3, stride=1, pad=1):
    return _ConvAbn(nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False), cout)


def _c3(cin, cout, stride=1):
    return _ConvAbn(nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False), cout)


def _up3(cin, cout):
    return nn.Sequential(nn.ConvTranspose3d(cin, cout, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
                         ActivatedBatchNorm(cout))


def _mfma_conv_pack(w, shift):
    """folded weights w [C_out, C_in, (dz,) dy, dx] + shift [C_out] -> the per-lane MFMA A operands of csrc/nr_kernels_conv3d.h conv3d_kernel
    ([dz][q][dy][dx][mt][lane], a 2-D kernel in the dz = 1 slice) and the bias, both padded with zeros to C_in % 4 == 0 / C_out % 16 == 0"""
    if w.dim() == 4:
        w3 = torch.zeros(w.shape[0], w.shape[1], 3, 3, 3, device=w.device, dtype=w.dtype)
        w3[:, :, 1] = w
        w = w3
    co_p, ci_p = (w.shape[0] + 15) // 16 * 16, (w.shape[1] + 3) // 4 * 4
    wp = torch.zeros(co_p, ci_p, 3, 3, 3, device=w.device, dtype=torch.float32)
    wp[:w.shape[0], :w.shape[1]] = w
    sp = torch.zeros(co_p, device=w.device, dtype=torch.float32)
    sp[:shift.numel()] = shift
    lane = torch.arange(64, device=w.device)
    co = 16 * torch.arange(co_p // 16, device=w.device)[:, None] + (lane & 15)[None]          # [MT, 64]
    ci = 4 * torch.arange(ci_p // 4, device=w.device)[:, None] + (lane >> 4)[None]            # [NQ, 64]
    return wp[co[None], ci[:, None]].permute(3, 0, 4, 5, 1, 2).contiguous(), sp


class FeatureNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv0, self.conv1 = _c2(3, 8), _c2(8, 8)
        self.conv2, self.conv3, self.conv4 = _c2(8, 16, 5, 2, 2), _c2(16, 16), _c2(16, 16)
        self.conv5, self.conv6 = _c2(16, 32, 5, 2, 2), _c2(32, 32)
        self.feature = nn.Conv2d(32, 32, 3, 1, 1)

    def _mfma2d(self, conv, bn, x):
        """a 3 x 3, stride-1 layer (conv0 / conv1 / conv3 / conv4 / conv6 with their frozen batch norm + leaky ReLU; `feature`: bias only)
        through the 3-D implicit-GEMM kernel on a one-plane volume: with D = 1 the kernel's dz = 0 / 2 taps lie in the zero padding and are skipped, so the [dz = 1] slice of its weight pack IS the 2-D convolution (neuray_conv3d_bn_leaky, csrc/nr_kernels_conv3d.h)."""
        src = (conv.weight,) + ((conv.bias,) if bn is None else (bn.weight, bn.bias, bn.running_mean, bn.running_var))
        stamp = tuple((t.data_ptr(), t._version) for t in src) + (str(x.device),)
        hit = conv.__dict__.get('_mfma_pack')
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                if bn is None:
                    w, shift = conv.weight.float(), conv.bias.float().contiguous()
                else:
                    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                    w = (conv.weight * scale[:, None, None, None]).float()                        # [C_out, C_in, dy, dx]
                    shift = (bn.bias - bn.running_mean * scale).float().contiguous()
                pack, shift = _mfma_conv_pack(w, shift)
            hit = conv.__dict__['_mfma_pack'] = (stamp, pack.to(x.device), shift.to(x.device))
        from . import render_ops
        # per layer, not per net: what the kernel cannot take - another dtype (a library layer ran under autocast in between),
        # not an [n, c, h, w] tensor, an image of 2^31 bytes or more - goes through the module this layer replaces
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] * x.shape[2] * x.shape[3] * 4 >= 0x7fffff00 or torch.is_autocast_enabled():
            y = conv(x)
            return y if bn is None else bn(y)
        y = render_ops.engine_for(x.device).conv3d_bn_leaky(x.contiguous()[:, :, None], hit[1], hit[2], 1.0 if bn is None else bn.slope,
                                                            conv.out_channels, 1)
        return y[:, :, 0]

    def _fast(self, x):
        from . import render_ops
        return (FAST_CONV3D and not self.training and not torch.is_grad_enabled() and x.dtype == torch.float32
                and (x.device.type == 'cuda' or render_ops._TEST_LIB is not None))

    def forward(self, x):
        if not self._fast(x):
            x = self.conv1(self.conv0(x))
            x = self.conv4(self.conv3(self.conv2(x)))
            return self.feature(self.conv6(self.conv5(x)))
        # frozen, no gradients (how the cost-volume init net runs it): the 3 x 3 stride-1 layers on the MFMA kernel (the two full-resolution
        # ones, 3 -> 8 -> 8, were 1.74 of the net's 3.1 ms per 16 x 800 x 800 through the library); the 5 x 5 stride-2 layers stay the library's
        x = self._mfma2d(self.conv1.conv, self.conv1.bn, self._mfma2d(self.conv0.conv, self.conv0.bn, x))
        x = self.conv2(x)
        x = self._mfma2d(self.conv4.conv, self.conv4.bn, self._mfma2d(self.conv3.conv, self.conv3.bn, x))
        x = self._mfma2d(self.conv6.conv, self.conv6.bn, self.conv5(x))
        return self._mfma2d(self.feature, None, x)


class CostRegNet(nn.Module):
    """mvsnet.py:29-69.  Inside CostVolumeInitNet the network is frozen and evaluated under no_grad (init_net.py:121-160), so its two
    layers that MIOpen runs far off any roofline go through hand-written kernels (csrc/nr_kernels_conv3d.h; round 5: also the encoder half's
    interior layers conv1 ... conv4 and the last decoder step conv11): `conv0` (32 -> 8 channels
    on the full-resolution variance volume: 68 % of the U-Net's MACs, 23 of its 37 ms through MIOpen on 8 x 64 x 160 x 160) as an
    implicit GEMM on the fp32 MFMA with the frozen batch norm folded into weights and bias, and `prob` (8 -> 1, memory bound: 6.2 ms
    through MIOpen).  Everything else, and every call that needs gradients or runs in training mode, takes the module path."""

    def __init__(self):
        super().__init__()
        self.conv0 = _c3(32, 8)
        self.conv1, self.conv2 = _c3(8, 16, 2), _c3(16, 16)
        self.conv3, self.conv4 = _c3(16, 32, 2), _c3(32, 32)
        self.conv5, self.conv6 = _c3(32, 64, 2), _c3(64, 64)
        self.conv7, self.conv9, self.conv11 = _up3(64, 32), _up3(32, 16), _up3(16, 8)
        self.prob = nn.Conv3d(8, 1, 3, stride=1, padding=1)

    _MFMA_LAYERS = {(8, 16, 2), (16, 16, 1), (16, 32, 2), (32, 32, 1)}          # (C_in, C_out, stride) built in csrc/nr_kernels_conv3d.h conv3d_kernel

    def _mfma(self, layer, x, fast):
        """an interior layer of the encoder half (conv1 ... conv4): the MFMA kernel with the frozen batch norm folded when `fast`, else the module.
        (conv5 / conv6 work on 8 x 25 x 25 volumes - a couple of hundred wave tasks - where the library's kernel is the faster one.)"""
        conv, bn = layer.conv, layer.bn
        key = (conv.in_channels, conv.out_channels, conv.stride[0])
        if not (fast and FAST_CONV3D and key in self._MFMA_LAYERS and x.shape[1] * x.shape[2] * x.shape[3] * x.shape[4] * 4 < 0x7fffff00):
            return layer(x)
        src = (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        stamp = tuple((t.data_ptr(), t._version) for t in src) + (str(x.device),)
        hit = layer.__dict__.get('_mfma_pack')
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                w = (conv.weight * scale[:, None, None, None, None]).float()                      # [C_out, C_in, dz, dy, dx]
                shift = (bn.bias - bn.running_mean * scale).float().contiguous()
                pack, shift = _mfma_conv_pack(w, shift)
            hit = layer.__dict__['_mfma_pack'] = (stamp, pack.to(x.device), shift.to(x.device))
        return self._engine(x).conv3d_bn_leaky(x.contiguous(), hit[1], hit[2], bn.slope, key[1], key[2])

    def _up(self, layer, x, skip, fast):
        """skip + a decoder step (conv9: 32 -> 16; conv11 goes through _packs / costreg_up11): transposed convolution + frozen batch norm + leaky
        ReLU + the skip add in one kernel when `fast`, else the modules"""
        up, bn = layer[0], layer[1]
        if not (fast and FAST_CONV3D and (up.in_channels, up.out_channels) == (32, 16)):
            return skip + layer(x)
        src = (up.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        stamp = tuple((t.data_ptr(), t._version) for t in src) + (str(x.device),)
        hit = layer.__dict__.get('_up_pack')
        if hit is None or hit[0] != stamp:
            with torch.no_grad():       # ConvTranspose3d weight [C_in, C_out, kz, ky, kx] with the batch norm folded -> [kz][ky][ci][co][kx]
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                pack = (up.weight * scale[None, :, None, None, None]).float().permute(2, 3, 0, 1, 4).contiguous()
                shift = (bn.bias - bn.running_mean * scale).float().contiguous()
            hit = layer.__dict__['_up_pack'] = (stamp, pack.to(x.device), shift.to(x.device))
        return self._engine(x).convtranspose3d_bn_leaky(x.contiguous(), hit[1], hit[2], bn.slope, skip.contiguous())

    def _tail(self, c0, fast=False):
        c2 = self._mfma(self.conv2, self._mfma(self.conv1, c0, fast), fast)
        c4 = self._mfma(self.conv4, self._mfma(self.conv3, c2, fast), fast)
        x = c4 + self.conv7(self.conv6(self.conv5(c4)))
        x = self._up(self.conv9, x, c2, fast)
        if fast:        # c0 + conv11(x): transposed convolution, frozen batch norm, leaky ReLU and the skip add in one kernel
            pack, shift, slope = self._packs(c0.device)[5:8]
            return self._engine(c0).costreg_up11(x.contiguous(), pack, shift, slope, c0.contiguous())
        return c0 + self.conv11(x)

    def forward_modules(self, x):
        return self.prob(self._tail(self.conv0(x)))

    def _engine(self, x):
        from . import render_ops
        if x.device.type != 'cuda' and render_ops._TEST_LIB is None:
            return None
        return render_ops.engine_for(x.device)

    def _fast_ok(self, x):
        # (one image's 32-channel volume must stay below 2^31 bytes for conv0's buffer descriptor: larger ones take the module path)
        # under autograd the kernels (no backward) run only when nothing they replace wants a gradient: the input and every parameter of
        # conv0 / conv11 / prob (the reference freezes MVSNet; a user who unfreezes any of them gets the module path and its gradients)
        fused = (*self.conv0.parameters(), *self.conv11.parameters(), *self.prob.parameters(), *self.conv1.parameters(), *self.conv2.parameters(),
                 *self.conv3.parameters(), *self.conv4.parameters(), *self.conv9.parameters())
        return (not self.training and not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in fused)))
                and x.dtype == torch.float32 and x.dim() == 5 and x.shape[2] * x.shape[3] * x.shape[4] * 128 < 0x7fffff00
                and self._engine(x) is not None)

    def _packs(self, device):
        """conv0's weights with the frozen batch norm folded in, as per-lane MFMA A fragments (csrc/nr_kernels_conv3d.h Conv0Params.wpack),
        and prob's 216 taps ([ky][c][kz][kx]); rebuilt when a parameter or buffer changes"""
        conv, bn = self.conv0.conv, self.conv0.bn
        up, ubn = self.conv11[0], self.conv11[1]
        src = (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.prob.weight, self.prob.bias,
               up.weight, ubn.weight, ubn.bias, ubn.running_mean, ubn.running_var)
        stamp = tuple((t.data_ptr(), t._version) for t in src) + (str(device),)
        hit = self.__dict__.get('_fast_packs')
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                w = (conv.weight * scale[:, None, None, None, None]).float()                      # [8, 32, kz, ky, kx]
                shift = (bn.bias - bn.running_mean * scale).float().contiguous()
                lane = torch.arange(64, device=w.device)
                m, g = lane & 15, lane >> 4
                top, bottom = (m < 8)[:, None, None].float(), (m >= 8)[:, None, None].float()
                # A fragments of the kernel's row slots r = 0 .. 3: MFMA rows 0 .. 7 = tap ky = r of the wave's first output row,
                # rows 8 .. 15 = tap ky = r - 1 of its second one
                pack = torch.zeros(3, 4, 3, 2, 64, 4, device=w.device)
                for r in range(4):
                    for q in range(2):
                        for i in range(4):
                            ch = 8 * g + 4 * q + i
                            vals = torch.zeros(64, 3, 3, device=w.device)                                  # [lane, kz, kx]
                            if r <= 2:
                                vals = vals + w[m.clamp(max=7), ch, :, r, :] * top
                            if r >= 1:
                                vals = vals + w[(m - 8).clamp(min=0), ch, :, r - 1, :] * bottom
                            pack[:, r, :, q, :, i] = vals.permute(1, 2, 0)
                # conv11 (ConvTranspose3d weight [16 in, 8 out, kz, ky, kx]) with its batch norm folded -> [kz][ky][ci][co][kx]
                uscale = ubn.weight / torch.sqrt(ubn.running_var + ubn.eps)
                upack = (up.weight * uscale[None, :, None, None, None]).float().permute(2, 3, 0, 1, 4).contiguous()
                ushift = (ubn.bias - ubn.running_mean * uscale).float().contiguous()
                hit = (stamp, pack.contiguous().to(device), shift.to(device), float(bn.slope),
                       self.prob.weight.detach()[0].permute(2, 0, 1, 3).reshape(216).float().contiguous().to(device),      # [ky][c][kz][kx]: the order the kernel spends them in
                       float(self.prob.bias.detach()[0]),
                       upack.to(device), ushift.to(device), float(ubn.slope))
            self.__dict__['_fast_packs'] = hit
        return hit[1:]

    def conv0_fast(self, x):
        """leaky_relu(batch_norm(conv0(x))) on a [n, 32, D, H, W] volume (any strides; channels-last-3d storage is taken as it is)"""
        pack, shift, slope = self._packs(x.device)[:3]
        xl = x.permute(0, 2, 3, 4, 1)
        if not xl.is_contiguous():
            xl = xl.contiguous()
        return self._engine(x).costreg_conv0(xl, pack, shift, slope)

    def prob_fast(self, x):
        w, b = self._packs(x.device)[3:5]
        return self._engine(x).costreg_prob(x.contiguous(), w, b)

    def forward(self, x):
        if not self._fast_ok(x):
            return self.forward_modules(x)
        return self.prob_fast(self._tail(self.conv0_fast(x), fast=True))


class MVSNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.feature = FeatureNet()
        self.cost_regularization = CostRegNet()

    def variance_volume(self, ref_feats, src_feats, ref_nn_idx, ref_prjs, src_prjs, depth_values):
        from . import render_ops
        # channels-last-3d storage when the volume goes straight into the conv0 kernel (no relayout of 1.7 GB in between)
        fast = not self.training and not torch.is_grad_enabled()
        return render_ops.engine_for(ref_feats.device).warp_variance(ref_feats, src_feats, ref_nn_idx, ref_prjs, src_prjs, depth_values,
                                                                      channels_last=fast)

    def construct_cost_volume_with_src(self, ref_imgs, src_imgs, ref_nn_idx, ref_prjs, src_prjs, depth_values, batch_num=2):
        """mvsnet.py:186-203 -> cost_reg [rfn,dn,h/4,w/4].  `batch_num` reference views go through the 3-D U-Net at a
        time, as in the reference (it bounds the activation memory, not the result)."""
        ref_feats, src_feats = self.feature(ref_imgs), self.feature(src_imgs)
        out = []
        for r0 in range(0, ref_feats.shape[0], batch_num):
            sl = slice(r0, r0 + batch_num)
            var = self.variance_volume(ref_feats[sl], src_feats, ref_nn_idx[sl], ref_prjs[sl], src_prjs, depth_values[sl])
            out.append(self.cost_regularization(var).squeeze(1))
        return torch.cat(out, 0)


def extract_model_state_dict(ckpt_path, prefixes_to_ignore=()):
    """mvsnet.py:205-229: a pytorch-lightning checkpoint ('state_dict' with a 'model.' prefix) or bare model weights"""
    # the reference's mvsnet_pl.ckpt is a pytorch-lightning checkpoint (hyper-parameters, callbacks, ... next to the
    # tensors): a trusted file that torch >= 2.6's default weights_only=True refuses to unpickle
    ckpt = torch.load(ckpt_path, map_location='cpu', weights_only=False)
    if 'state_dict' in ckpt:
        items = ((k[6:], v) for k, v in ckpt['state_dict'].items() if k.startswith('model.'))
    else:
        items = ckpt.items()
    return {k: v for k, v in items if not any(k.startswith(p) for p in prefixes_to_ignore)}


def load_ckpt(model, ckpt_path, prefixes_to_ignore=()):
    """mvsnet.py:231-235"""
    sd = model.state_dict()
    sd.update(extract_model_state_dict(ckpt_path, prefixes_to_ignore))
    model.load_state_dict(sd)
