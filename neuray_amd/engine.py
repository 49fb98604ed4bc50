"""RenderEngine: drives the HIP kernels of the per-ray render path from torch tensors.

This is host plumbing (device buffers, streams, argument marshalling) around the C ABI of
include/neuray_hip.h; all arithmetic of the hot path happens in libneuray_hip.so.  One engine per
process / device.  The reference functions each step replaces are cited in include/neuray_hip.h.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_DIST_HEADS = ('mean_decoder', 'var_decoder', 'aw_decoder', 'vis_decoder')


import functools


@functools.lru_cache(maxsize=None)
def pass_tensor_keys(dist_prefix, agg_prefix):
    """state_dict keys of one pass in the order of enum nr::PassTensor (csrc/nr_layout.h)."""
    keys = []
    for head in _DIST_HEADS:
        for i in (0, 2, 4):
            keys += ['%s%s.%d.weight' % (dist_prefix, head, i), '%s%s.%d.bias' % (dist_prefix, head, i)]
    for i in (0, 2):
        keys += ['%sprob_embed.%d.weight' % (agg_prefix, i), '%sprob_embed.%d.bias' % (agg_prefix, i)]
    impl = agg_prefix + 'agg_impl.'
    for name in ('ray_dir_fc', 'base_fc', 'vis_fc', 'vis_fc2', 'geometry_fc'):
        for i in (0, 2):
            keys += ['%s%s.%d.weight' % (impl, name, i), '%s%s.%d.bias' % (impl, name, i)]
    keys += [impl + 'ray_attention.w_qs.weight', impl + 'ray_attention.w_ks.weight', impl + 'ray_attention.w_vs.weight',
             impl + 'ray_attention.fc.weight', impl + 'ray_attention.layer_norm.weight', impl + 'ray_attention.layer_norm.bias']
    for i in (0, 2):
        keys += ['%sout_geometry_fc.%d.weight' % (impl, i), '%sout_geometry_fc.%d.bias' % (impl, i)]
    for i in (0, 2, 4):
        keys += ['%srgb_fc.%d.weight' % (impl, i), '%srgb_fc.%d.bias' % (impl, i)]
    for i in (0, 2):
        keys += ['%sneuray_fc.%d.weight' % (impl, i), '%sneuray_fc.%d.bias' % (impl, i)]
    assert len(keys) == _lib.PASS_TENSORS
    return tuple(keys)


def posenc_table(d_hid, n_samples):
    """Sinusoid table of IBRNetWithNeuRay.posenc (network/ibrnet.py:305-313), numpy float64 -> float32."""
    pos = np.arange(n_samples, dtype=np.float64)[:, None]
    j = np.arange(d_hid)[None, :]
    t = pos / np.power(10000, 2 * (j // 2) / d_hid)
    t[:, 0::2] = np.sin(t[:, 0::2])
    t[:, 1::2] = np.cos(t[:, 1::2])
    return torch.from_numpy(t.astype(np.float32))


_K_INV_CACHE = {}


def host_inverse(Ks):
    """torch.inverse of [n,3,3] intrinsics on the host in fp32 (see RenderEngine.prepare_query).  Memoised on the matrix bytes: a
    training run or a render loop sees a handful of distinct cameras, and the LAPACK call is the one host-side op of a step whose cost
    depends on the process's thread settings (25 ms per call after torch.set_num_threads(128): MKL then threads even a 3 x 3 getrf)."""
    k = torch.as_tensor(Ks).detach().to('cpu', torch.float32).contiguous()
    key = (tuple(k.shape), k.numpy().tobytes())
    inv = _K_INV_CACHE.get(key)
    if inv is None:
        if len(_K_INV_CACHE) >= 4096:
            _K_INV_CACHE.clear()
        inv = _K_INV_CACHE[key] = torch.inverse(k)
    return inv.clone()


class PackedPass:
    """Device-resident packed weights of one pass (dist decoder + aggregation net).  folded: prob_embed.2 is multiplied into
    neuray_fc.0 / base_fc.0 (neuray_pack_pass_weights_folded): inference packs; the backward kernels and the training forward
    take the unfolded form."""

    def __init__(self, dev_tensor, has_vis_head, folded=False, dev_x3=None):
        self.dev = dev_tensor
        self.has_vis_head = has_vis_head
        self.folded = folded
        # the point kernel's layers once more in the split-operand form of NEURAY_ARITH_X3 (neuray_pack_pass_weights_x3): present when
        # the engine renders with arith='x3'; every other kernel keeps reading `dev`
        self.dev_x3 = dev_x3


class ViewSet:
    """Per-render() constants of the reference views: camera constants + channels-last maps."""

    def __init__(self, view_const, ray_feats, img_feats, rgba, rfn, h, w, fh, fw):
        self.view_const, self.ray_feats, self.img_feats, self.rgba = view_const, ray_feats, img_feats, rgba
        self.rfn, self.h, self.w, self.fh, self.fw = rfn, h, w, fh, fw


class RenderEngine:
    def __init__(self, device, _test_lib=None, views_per_wave=0, variant='fp32', arith='f32'):
        """device: torch device of the HIP GPU.  `_test_lib` is for the CPU test-suite only (binds the
        emulator build of the same kernels); the product path always uses libneuray_hip.so.
        variant: 'fp32' (the product) or 'bf16' (libneuray_hip_bf16.so: bf16 MFMA operands, fp32 accumulation;
        inference only, reported separately - DESIGN.md section 4.8).
        arith: 'f32' = the MLP contractions of the inference point kernel on the fp32 MFMA; 'x3' = on the K = 32 bf16 MFMA with every
        operand split exactly into three bf16 parts (six products, fp32 accumulation: each product within 2^-23 of exact - DESIGN.md
        section 4.12; the fp32 library only).  Training forwards / backwards always run on the fp32 MFMA."""
        if arith not in ('f32', 'x3'):
            raise ValueError("neuray_amd: arith=%r (use 'f32' or 'x3')" % (arith,))
        if arith == 'x3' and variant != 'fp32':
            raise ValueError("neuray_amd: arith='x3' is an arithmetic of the fp32 library (variant=%r is its own operand format)" % (variant,))
        self.arith = arith
        self.device = torch.device(device)
        if _test_lib is None:
            if self.device.type != 'cuda':
                raise RuntimeError("neuray_amd.RenderEngine needs a HIP device (got %s): the render path has no CPU "
                                   "fallback" % self.device)
            self.lib = _lib.load(variant)
        else:
            self.lib = _test_lib
        self.views_per_wave = views_per_wave
        self.variant = variant
        self._posenc = {}
        # optional kernel timing: set to a list and every point/ray launch appends
        # (name, start_event, end_event, n_points) recorded on the launch stream (HIP events)
        self.timing = None
        self.max_backward_samples = _lib.MAX_BACKWARD_SAMPLES
        self.slot_stats = None                     # optional int64 device tensor [2]: every point-kernel launch adds (view slots run, view slots)

    # ------------------------------------------------------------------------------------------
    def _stream(self):
        if self.device.type == 'cuda':
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return C.c_void_p(0)

    def _f32(self, t):
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def _check(self, rc):
        _lib.check(self.lib, rc)

    def _event_pair(self):
        if self.timing is None or self.device.type != 'cuda':
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(self.device))
        return e0, e1

    def _event_done(self, ev, name, n):
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(self.device))
            self.timing.append((name, ev[0], ev[1], n))

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------------------------------
    def pack_pass(self, state_dict, dist_prefix, agg_prefix, fold=False):
        """state_dict (tensors or ndarrays, reference key names) -> PackedPass on the device.  fold: the inference form with
        prob_embed.2 folded into its consumers (one 32 x 32 layer less per (point, view); same function up to fp32 rounding)."""
        keys = pass_tensor_keys(dist_prefix, agg_prefix)
        host, ptrs = [], (C.c_void_p * _lib.PASS_TENSORS)()
        has_vis = (dist_prefix + 'vis_decoder.0.weight') in state_dict
        for i, k in enumerate(keys):
            if k not in state_dict:
                if '.vis_decoder.' in k and not has_vis:
                    ptrs[i] = None
                    continue
                raise KeyError("neuray_amd: missing weight %s" % k)
            v = state_dict[k]
            t = torch.as_tensor(v).detach().to('cpu', torch.float32).contiguous()
            host.append(t)
            ptrs[i] = t.data_ptr()
        n = self.lib.neuray_packed_pass_floats()
        packed = torch.empty(n, dtype=torch.float32)
        pack = self.lib.neuray_pack_pass_weights_folded if fold else self.lib.neuray_pack_pass_weights
        self._check(pack(ptrs, C.c_void_p(packed.data_ptr())))
        dev_x3 = None
        if fold and self.arith == 'x3':
            x3 = torch.empty(int(self.lib.neuray_packed_points_floats_x3()), dtype=torch.float32)
            self._check(self.lib.neuray_pack_pass_weights_x3(ptrs, C.c_void_p(x3.data_ptr())))
            dev_x3 = x3.to(self.device)
        return PackedPass(packed.to(self.device), has_vis, folded=bool(fold), dev_x3=dev_x3)

    def posenc(self, dn):
        if dn not in self._posenc:
            self._posenc[dn] = posenc_table(16, dn).to(self.device)
        return self._posenc[dn]

    # ------------------------------------------------------------------------------------------
    def prepare_views(self, ref_imgs_info):
        """ref_imgs_info: dict with imgs [rfn,3,h,w], poses, Ks, depth_range, ray_feats, img_feats (NCHW)."""
        imgs = self._f32(ref_imgs_info['imgs'])
        rfn, _, h, w = imgs.shape
        rf, imf = ref_imgs_info['ray_feats'], ref_imgs_info['img_feats']
        assert rf.shape == imf.shape and rf.shape[0] == rfn and rf.shape[1] == 32, (rf.shape, imf.shape)
        fh, fw = rf.shape[-2:]
        if rfn > _lib.MAX_VIEWS:
            raise RuntimeError("neuray_amd: %d reference views > %d" % (rfn, _lib.MAX_VIEWS))
        s = self._stream()
        vc = self.empty(rfn, _lib.VIEW_CONST)
        poses, Ks, dr = self._f32(ref_imgs_info['poses']), self._f32(ref_imgs_info['Ks']), self._f32(ref_imgs_info['depth_range'])
        self._check(self.lib.neuray_setup_views(poses.data_ptr(), Ks.data_ptr(), dr.data_ptr(), rfn, vc.data_ptr(), s))
        rgba = self.empty(rfn, h, w, 4)
        self._check(self.lib.neuray_relayout_nhwc(imgs.data_ptr(), rgba.data_ptr(), rfn, 3, h, w, 4, s))
        return ViewSet(vc, self._nhwc32(rf, s), self._nhwc32(imf, s), rgba, rfn, h, w, fh, fw)

    def _nhwc32(self, t, s):
        """[n,32,fh,fw] feature maps -> channels-last storage [n,fh,fw,32].  A channels-last tensor (what the encoders of
        network/encoders.py emit) already IS that storage: no relayout pass."""
        n, c, fh, fw = t.shape
        t = t.detach()
        if t.dtype == torch.float32 and t.device == self.device and t.is_contiguous(memory_format=torch.channels_last):
            return t.permute(0, 2, 3, 1)
        t = self._f32(t)
        out = self.empty(n, fh, fw, 32)
        self._check(self.lib.neuray_relayout_nhwc(t.data_ptr(), out.data_ptr(), n, 32, fh, fw, 32, s))
        return out

    def diff_feats(self, ref_imgs_info, depth):
        """network/init_net.py:30-61 `get_diff_feats` given the metric depth maps [rfn,1,h,w] (the clamped depths the
        reference recovers from its normalised input): -> [rfn,8,h,w] (channels-last storage) of
        [rgb_mean 3, rgb_var 3, dpt_mean, dpt_var]."""
        imgs = self._f32(ref_imgs_info['imgs'])
        rfn, _, h, w = imgs.shape
        if rfn > _lib.MAX_VIEWS:
            raise RuntimeError("neuray_amd: %d views > %d" % (rfn, _lib.MAX_VIEWS))
        s = self._stream()
        poses, Ks, dr = self._f32(ref_imgs_info['poses']), self._f32(ref_imgs_info['Ks']), self._f32(ref_imgs_info['depth_range'])
        vc = self.empty(rfn, _lib.VIEW_CONST)
        self._check(self.lib.neuray_setup_views(poses.data_ptr(), Ks.data_ptr(), dr.data_ptr(), rfn, vc.data_ptr(), s))
        kinv = self._f32(host_inverse(Ks))                    # as depth2pts3d (init_net.py:23)
        lift = self.empty(rfn, _lib.QUERY_CONST)
        for v in range(rfn):
            self._check(self.lib.neuray_setup_query(poses[v].data_ptr(), kinv[v].data_ptr(), dr[v].data_ptr(), lift[v].data_ptr(), s))
        rgbd_nchw = torch.cat([imgs, self._f32(depth)], 1).contiguous()
        rgbd = self.empty(rfn, h, w, 4)
        self._check(self.lib.neuray_relayout_nhwc(rgbd_nchw.data_ptr(), rgbd.data_ptr(), rfn, 4, h, w, 4, s))
        out = self.empty(rfn, h, w, 8)
        self._check(self.lib.neuray_diff_feats(vc.data_ptr(), lift.data_ptr(), rgbd.data_ptr(), rfn, h, w, out.data_ptr(), s))
        return out.permute(0, 3, 1, 2)

    def costreg_conv0(self, x_ndhwc, wpack, bias, slope):
        """MVSNet CostRegNet.conv0 with the frozen batch norm folded (neuray_conv3d_c32_c8): x [n,d,h,w,32] contiguous -> [n,8,d,h,w]"""
        n, d, h, w, c = x_ndhwc.shape
        assert c == 32 and x_ndhwc.is_contiguous() and x_ndhwc.dtype == torch.float32
        out = self.empty(n, 8, d, h, w)
        self._check(self.lib.neuray_conv3d_c32_c8(x_ndhwc.data_ptr(), wpack.data_ptr(), bias.data_ptr(), float(slope), n, d, h, w,
                                                  out.data_ptr(), self._stream()))
        return out

    def convtranspose3d_bn_leaky(self, x, wpack, bias, slope, skip):
        """skip + leaky_relu(batch_norm(ConvTranspose3d(C_in, C_out, 3, stride 2, padding 1, output_padding 1)(x))) with the frozen batch norm
        folded (neuray_convtranspose3d_bn_leaky: conv11 16 -> 8, conv9 32 -> 16): x [n,C_in,d,h,w] -> [n,C_out,2d,2h,2w]"""
        n, c, d, h, w = x.shape
        cout = bias.numel()
        assert x.is_contiguous() and x.dtype == torch.float32
        assert skip is None or (tuple(skip.shape) == (n, cout, 2 * d, 2 * h, 2 * w) and skip.is_contiguous() and skip.dtype == torch.float32)
        out = self.empty(n, cout, 2 * d, 2 * h, 2 * w)
        self._check(self.lib.neuray_convtranspose3d_bn_leaky(x.data_ptr(), wpack.data_ptr(), bias.data_ptr(), float(slope),
                                                             skip.data_ptr() if skip is not None else None, n, c, cout, d, h, w, out.data_ptr(),
                                                             self._stream()))
        return out

    def costreg_up11(self, x, wpack, bias, slope, skip):
        """MVSNet CostRegNet: skip + conv11(x) with the frozen batch norm folded (neuray_convtranspose3d_c16_c8): x [n,16,d,h,w] -> [n,8,2d,2h,2w]"""
        n, c, d, h, w = x.shape
        assert c == 16 and x.is_contiguous() and x.dtype == torch.float32
        assert skip is None or (tuple(skip.shape) == (n, 8, 2 * d, 2 * h, 2 * w) and skip.is_contiguous() and skip.dtype == torch.float32)
        out = self.empty(n, 8, 2 * d, 2 * h, 2 * w)
        self._check(self.lib.neuray_convtranspose3d_c16_c8(x.data_ptr(), wpack.data_ptr(), bias.data_ptr(), float(slope),
                                                           skip.data_ptr() if skip is not None else None, n, d, h, w, out.data_ptr(), self._stream()))
        return out

    def conv3d_bn_leaky(self, x, wpack, bias, slope, cout, stride):
        """leaky_relu(batch_norm(Conv3d(C_in, cout, 3, stride, padding=1)(x))) with the frozen batch norm folded (neuray_conv3d_bn_leaky):
        x [n,C_in,d,h,w] -> [n,cout,(d-1)//stride+1,(h-1)//stride+1,(w-1)//stride+1]"""
        n, c, d, h, w = x.shape
        assert x.is_contiguous() and x.dtype == torch.float32 and bias.numel() == (cout + 15) // 16 * 16        # (pack / bias padded to the kernel's counts)
        out = self.empty(n, cout, (d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1)
        self._check(self.lib.neuray_conv3d_bn_leaky(x.data_ptr(), wpack.data_ptr(), bias.data_ptr(), float(slope), n, c, cout, int(stride), d, h, w,
                                                    out.data_ptr(), self._stream()))
        return out

    def conv3x3_x3_packs(self, weight, forward=True, gradient=False):
        """weight [C_out, C_in, 3, 3] -> (pack, pack_t): the split-operand packs of neuray_conv3x3_x3 for the layer and for its data gradient
        (None where not asked for), one launch"""
        co, ci = weight.shape[:2]
        w = weight.detach().contiguous().float()
        nbytes = self.lib.neuray_conv3x3_x3_pack_bytes(ci, co)
        assert nbytes > 0 and tuple(weight.shape[2:]) == (3, 3), tuple(weight.shape)
        pack = torch.empty(nbytes // 4, dtype=torch.int32, device=self.device) if forward else None
        pack_t = torch.empty(nbytes // 4, dtype=torch.int32, device=self.device) if gradient else None
        self._check(self.lib.neuray_conv3x3_x3_pack(w.data_ptr(), co, ci, pack.data_ptr() if forward else None, pack_t.data_ptr() if gradient else None,
                                                    self._stream()))
        return pack, pack_t

    def conv3x3_x3_pack(self, weight, transpose_flip=False):
        return self.conv3x3_x3_packs(weight, not transpose_flip, transpose_flip)[1 if transpose_flip else 0]

    def conv3x3_x3(self, x, pack, bias, cout, pad=0):
        """x [n, C_in, h, w] contiguous fp32 with `pad` rings of zeros -> the 3 x 3 correlation [n, cout, h + 2 pad - 2, w + 2 pad - 2] on the
        K = 32 bf16 MFMA with exactly split operands (neuray_conv3x3_x3)"""
        n, c, h, w = x.shape
        assert x.is_contiguous() and x.dtype == torch.float32 and (bias is None or (bias.is_contiguous() and bias.dtype == torch.float32 and bias.numel() == cout))
        out = self.empty(n, cout, h + 2 * pad - 2, w + 2 * pad - 2)
        self._check(self.lib.neuray_conv3x3_x3(x.data_ptr(), pack.data_ptr(), bias.data_ptr() if bias is not None else None, n, c, cout, h, w, int(pad),
                                               out.data_ptr(), self._stream()))
        return out

    def conv3x3_x3_wrw(self, d_out, xp):
        """d_out [n, C_out, h, w], xp [n, C_in, h + 2, w + 2] (the layer's pre-padded input), both contiguous fp32 -> the weight gradient
        [C_out, C_in, 3, 3] (neuray_conv3x3_x3_wrw), or None where the shape is not supported (odd padded width)"""
        n, cout, oh, ow = d_out.shape
        cin, hp, wp = xp.shape[1:]
        assert d_out.is_contiguous() and xp.is_contiguous() and d_out.dtype == xp.dtype == torch.float32 and (hp, wp) == (oh + 2, ow + 2)
        nws = self.lib.neuray_conv3x3_x3_wrw_workspace_floats(n, cin, cout, hp, wp)
        if nws < 0:
            return None
        ws, dw = self.empty(nws), self.empty(cout, cin, 3, 3)
        self._check(self.lib.neuray_conv3x3_x3_wrw(d_out.data_ptr(), xp.data_ptr(), n, cin, cout, hp, wp, ws.data_ptr(), dw.data_ptr(), self._stream()))
        return dw

    def scale_shift_leaky_(self, x, scale, shift, slope):
        """x [n,c,...] contiguous fp32 <- leaky_relu(x * scale[c] + shift[c], slope), in place (neuray_scale_shift_leaky: MVSNet's frozen
        activated batch norm as one pass)"""
        assert x.is_contiguous() and x.dtype == torch.float32 and x.dim() >= 3
        n, c = x.shape[:2]
        inner = x.numel() // (n * c)
        self._check(self.lib.neuray_scale_shift_leaky(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), n, c, inner, float(slope), self._stream()))
        return x

    def costreg_prob(self, x, w27, bias):
        """MVSNet CostRegNet.prob (neuray_conv3d_c8_c1): x [n,8,d,h,w] contiguous -> [n,1,d,h,w]"""
        n, c, d, h, w = x.shape
        assert c == 8 and x.is_contiguous() and x.dtype == torch.float32
        out = self.empty(n, 1, d, h, w)
        self._check(self.lib.neuray_conv3d_c8_c1(x.data_ptr(), w27.data_ptr(), float(bias), n, d, h, w, out.data_ptr(), self._stream()))
        return out

    def warp_variance(self, ref_feats, src_feats, nn_ids, ref_prjs, src_prjs, depth_vals, channels_last=False):
        """network/mvsnet/mvsnet.py:186-203: feature maps [n,32,fh,fw], nn_ids [rfn,n_num] (rows of src_feats), 4x4
        projections, depth_vals [rfn,dn] -> variance volume [rfn,32,dn,fh,fw] (channels_last: the same tensor in channels-last-3d
        storage, which CostRegNet's conv0 kernel reads as it is)."""
        rfn, c, fh, fw = ref_feats.shape
        assert c == 32 and src_feats.shape[1:] == ref_feats.shape[1:]
        sn, n_num, dn = src_feats.shape[0], nn_ids.shape[1], depth_vals.shape[1]
        if nn_ids.is_cuda:
            # A host-side `int(nn_ids.max())` is a device -> host copy: the host waits for everything the previous training step still has
            # queued (three such waits sat at the top of every generalisation step).  The range is checked on the device instead, the
            # verdict travels to pinned memory behind the stream and is raised by check_deferred() - at the next call of this method
            # at the latest - while the kernel itself only ever sees clamped (in-range) indices.
            self.check_deferred(wait=False)
            if not torch.cuda.is_current_stream_capturing():       # (a captured graph replays the clamped kernel; nothing to report to)
                self._defer_check(((nn_ids < 0) | (nn_ids >= sn)).any(),
                                  "warp_variance: a neighbour index in nn_ids lies outside the %d source views" % sn)
            nn_ids = nn_ids.clamp(0, sn - 1)
        else:
            assert int(nn_ids.max()) < sn and int(nn_ids.min()) >= 0
        s = self._stream()
        nhwc = lambda t: self._f32(t).permute(0, 2, 3, 1).contiguous()
        rf, sf = nhwc(ref_feats), nhwc(src_feats)
        ids = nn_ids.to(device=self.device, dtype=torch.int32).contiguous()
        # transform = src_proj @ ref_proj_inv, per (reference view, neighbour) as homo_warp computes it (modules.py:36)
        inv_res = torch.linalg.inv_ex(self._f32(ref_prjs))             # torch.inverse without its host-side singularity check (a sync) ...
        inv = inv_res.inverse
        if inv_res.info.is_cuda and not torch.cuda.is_current_stream_capturing():
            # ... whose verdict is deferred like the index range: torch.inverse would have raised on a singular projection matrix
            self._defer_check((inv_res.info != 0).any(), "warp_variance: a reference projection matrix is singular (torch.inverse would have raised)")
        tr = torch.stack([self._f32(src_prjs)[nn_ids[:, j].to(self.device).long()] @ inv for j in range(n_num)], 1)[:, :, :3, :].contiguous()
        dv = self._f32(depth_vals)
        out = self.empty(rfn, dn, fh, fw, 32) if channels_last else self.empty(rfn, 32, dn, fh, fw)
        self._check(self.lib.neuray_warp_variance_layout(rf.data_ptr(), sf.data_ptr(), ids.data_ptr(), tr.data_ptr(), dv.data_ptr(),
                                                         rfn, sn, n_num, dn, fh, fw, int(bool(channels_last)), out.data_ptr(), s))
        return out.permute(0, 4, 1, 2, 3) if channels_last else out

    def _defer_check(self, bad, message):
        """bad: 0-dim bool tensor on the device (True = the input was invalid).  Recorded without waiting for it."""
        flag = torch.empty((), dtype=torch.bool, pin_memory=True)
        flag.copy_(bad, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        n = self.__dict__['_deferred_calls'] = self.__dict__.get('_deferred_calls', 0) + 1
        self.__dict__.setdefault('_deferred', []).append((ev, flag, '%s [input check #%d of this engine]' % (message, n)))

    def check_deferred(self, wait=True):
        """Raise the AssertionError of an input check that ran on the device (warp_variance's nn_ids range, its singular-matrix check).
        wait=True: after the stream has drained; wait=False: only verdicts that have already arrived.  Product code drains at the sync points
        it already has: CostVolumeInitNet.forward in evaluation, NeuralRayGenRenderer / Ft train_step callers through
        render_ops.check_deferred_inputs() where the loss is read back."""
        pending = self.__dict__.get('_deferred')
        if not pending:
            return
        if wait:
            torch.cuda.current_stream(self.device).synchronize()
        keep, failed = [], None
        for ev, flag, message in pending:
            if not ev.query():
                keep.append((ev, flag, message))
            elif bool(flag) and failed is None:
                failed = message
        self.__dict__['_deferred'] = keep
        if failed is not None:
            raise AssertionError("neuray_amd: " + failed)

    def prepare_query(self, que_imgs_info):
        """-> query constant block.  K^-1 is `torch.inverse(Ks)` as in the reference (render_ops.py:20), evaluated on the
        HOST (LAPACK, fp32) wherever the tensor lives: a 3x3 LU on the GPU goes through a different solver whose last bits
        differ, and the reference-generated golden vectors (and the bit-exact geometry contract) are pinned
        to the CPU result.  36 bytes D2H once per query view (cached by HipRenderPath._query); a caller that wants no host
        round trip at all hands `Ks_inv` over."""
        pose = self._f32(que_imgs_info['poses'])
        assert pose.shape[0] == 1, "one query view per render() call (qn = 1)"
        # torch.inverse returns a column-major tensor: make it contiguous and keep it bound until the launch
        kinv = self._f32(que_imgs_info['Ks_inv'] if 'Ks_inv' in que_imgs_info else host_inverse(que_imgs_info['Ks']))
        dr = self._f32(que_imgs_info['depth_range'])
        qc = self.empty(_lib.QUERY_CONST)
        self._check(self.lib.neuray_setup_query(pose.data_ptr(), kinv.data_ptr(), dr.data_ptr(), qc.data_ptr(), self._stream()))
        return qc

    # ------------------------------------------------------------------------------------------
    def sample_coarse_depth(self, que_depth_range, rn, dn, uniforms=None):
        """uniforms: None, or the [rn, dn-2] draws of sample_depth(random_sample=True) (render_ops.py:160-161)"""
        dr = self._f32(que_depth_range).reshape(-1)[:2].contiguous()
        depth = self.empty(rn, dn)
        if uniforms is None:
            self._check(self.lib.neuray_sample_coarse_depth(dr.data_ptr(), rn, dn, depth.data_ptr(), self._stream()))
        else:
            uniforms = self._f32(uniforms).reshape(rn, dn - 2)
            self._check(self.lib.neuray_sample_coarse_depth_jittered(dr.data_ptr(), uniforms.data_ptr(), rn, dn, depth.data_ptr(), self._stream()))
        return depth

    def sample_fine_depth(self, qconst, depth, hit_prob, fdn, use_all=False, u=None, sort=True, inv_mode=True, trace=False):
        """trace: -> (depths, searchsorted bins int32 [rn, fdn] in the order of u, cdf [rn, dn + 1]) (tests/test_fine_index.py)"""
        rn, dn = depth.shape
        out = self.empty(rn, fdn + (dn if use_all else 0))
        u_ptr = None
        if u is not None:
            u = self._f32(u).reshape(rn, fdn)
            u_ptr = u.data_ptr()
        flags = int(use_all) | (0 if sort else 2) | (0 if inv_mode else 4)
        if trace:
            idx, cdf = self.empty(rn, fdn, dtype=torch.int32), self.empty(rn, dn + 1)
            self._check(self.lib.neuray_sample_fine_depth_traced(qconst.data_ptr(), depth.data_ptr(), hit_prob.data_ptr(), u_ptr, rn, dn, fdn,
                                                                 flags, out.data_ptr(), idx.data_ptr(), cdf.data_ptr(), self._stream()))
            return out, idx, cdf
        self._check(self.lib.neuray_sample_fine_depth(qconst.data_ptr(), depth.data_ptr(), hit_prob.data_ptr(), u_ptr,
                                                      rn, dn, fdn, flags, out.data_ptr(), self._stream()))
        return out

    def setup_views(self, poses, Ks, depth_range):
        poses, Ks, dr = self._f32(poses), self._f32(Ks), self._f32(depth_range)
        vc = self.empty(poses.shape[0], _lib.VIEW_CONST)
        self._check(self.lib.neuray_setup_views(poses.data_ptr(), Ks.data_ptr(), dr.data_ptr(), poses.shape[0], vc.data_ptr(),
                                                self._stream()))
        return vc

    def rays_points(self, qconst, coords, depth=None):
        """coords2rays / depth2points for one query view: -> centers, dirs [rn,3] (, pts, que_dir [rn,dn,3])"""
        coords = self._f32(coords)
        rn = coords.shape[0]
        centers, dirs = self.empty(rn, 3), self.empty(rn, 3)
        if depth is None:
            self._check(self.lib.neuray_rays_points(qconst.data_ptr(), coords.data_ptr(), None, rn, 1, centers.data_ptr(),
                                                    dirs.data_ptr(), None, None, self._stream()))
            return centers, dirs
        depth = self._f32(depth)
        dn = depth.shape[1]
        pts, qdir = self.empty(rn, dn, 3), self.empty(rn, dn, 3)
        self._check(self.lib.neuray_rays_points(qconst.data_ptr(), coords.data_ptr(), depth.data_ptr(), rn, dn, centers.data_ptr(),
                                                dirs.data_ptr(), pts.data_ptr(), qdir.data_ptr(), self._stream()))
        return centers, dirs, pts, qdir

    def depth_dists(self, depth, que_depth_range):
        depth = self._f32(depth)
        out = torch.empty_like(depth)
        dr = self._f32(que_depth_range).reshape(-1)[:2].contiguous() if que_depth_range is not None else None
        rows = depth.numel() // depth.shape[-1]
        self._check(self.lib.neuray_depth_dists(depth.data_ptr(), dr.data_ptr() if dr is not None else None,
                                                int(dr is not None), rows, depth.shape[-1], out.data_ptr(), self._stream()))
        return out

    def project_points(self, view_const, pts, rfn, h, w):
        pts = self._f32(pts)
        pn = pts.shape[0]
        d, p2, z = self.empty(rfn, pn, 3), self.empty(rfn, pn, 2), self.empty(rfn, pn)
        m = self.empty(rfn, pn, dtype=torch.uint8)
        self._check(self.lib.neuray_project_points(view_const.data_ptr(), pts.data_ptr(), rfn, pn, int(h), int(w), d.data_ptr(),
                                                   p2.data_ptr(), z.data_ptr(), m.data_ptr(), self._stream()))
        return d, p2, z, m.bool()

    def alpha2hit_prob(self, alpha):
        alpha = self._f32(alpha)
        out = torch.empty_like(alpha)
        self._check(self.lib.neuray_alpha2hit_prob(alpha.data_ptr(), alpha.numel() // alpha.shape[-1], alpha.shape[-1],
                                                   out.data_ptr(), self._stream()))
        return out

    def dist_decoder_rows(self, feats, packed, var_bias=0.05):
        """MixtureLogisticsDistDecoder.forward on [..., 32] rows -> mean [...,2], var [...,2], vis [...,1] or None, aw [...,1]"""
        feats = self._f32(feats)
        lead = feats.shape[:-1]
        n = feats.numel() // 32
        mean, var, aw = self.empty(n, 2), self.empty(n, 2), self.empty(n)
        vis = self.empty(n) if packed.has_vis_head else None
        self._check(self.lib.neuray_dist_decoder_rows(feats.data_ptr(), packed.dev.data_ptr(), n, int(packed.has_vis_head),
                                                      float(var_bias), mean.data_ptr(), var.data_ptr(), aw.data_ptr(),
                                                      vis.data_ptr() if vis is not None else None, self._stream()))
        return (mean.view(*lead, 2), var.view(*lead, 2), vis.view(*lead, 1) if vis is not None else None, aw.view(*lead, 1))

    def flat_pass_device(self, named_params, dist_prefix, agg_prefix, allow_missing_agg=False):
        """Flat natural-layout weights from DEVICE tensors without a host round trip (training: weights change every step).
        named_params: dict key -> tensor.  -> (flat device tensor, has_vis)"""
        keys = pass_tensor_keys(dist_prefix, agg_prefix)
        has_vis = (dist_prefix + 'vis_decoder.0.weight') in named_params
        parts = []
        for i, k in enumerate(keys):
            if k in named_params:
                parts.append(named_params[k].detach().reshape(-1).to(device=self.device, dtype=torch.float32))
            else:
                if not (('.vis_decoder.' in k and not has_vis) or (allow_missing_agg and k.startswith(agg_prefix))):
                    raise KeyError("neuray_amd: missing weight %s" % k)
                off = self._flat_offsets()
                parts.append(self._zero_const(off[i + 1] - off[i]))
        return torch.cat(parts), has_vis

    def _zero_const(self, n):
        """n zero floats as a view of one cached, never written buffer (a decoder-only pack asks for ~26 absent tensors on every call: one
        fill launch each as torch.zeros - 112 launches per generalisation step, tools/count_step_ops.py)"""
        z = self.__dict__.get('_zero_const_buf')
        if z is None or z.numel() < n:
            z = torch.zeros(max(int(n), 1 << 16), dtype=torch.float32, device=self.device)
            self.__dict__['_zero_const_buf'] = z
        return z[:n]

    def pack_pass_device(self, flat, has_vis):
        """PackedPass from the flat natural layout, on the device: packed = flat[index] * scale (neuray_pack_pass_index_map)."""
        cache = self.__dict__.setdefault('_pack_maps', {})
        if has_vis not in cache:
            n = int(self.lib.neuray_packed_pass_floats())
            idx = torch.empty(n, dtype=torch.int32)
            scale = torch.empty(n, dtype=torch.float32)
            self._check(self.lib.neuray_pack_pass_index_map(int(has_vis), C.c_void_p(idx.data_ptr()), C.c_void_p(scale.data_ptr())))
            cache[has_vis] = (idx.clamp(min=0).long().to(self.device), scale.to(self.device))
        idx, scale = cache[has_vis]
        return PackedPass(self._split_quads(flat[idx] * scale, 0), has_vis)

    def _split_quads(self, packed, transposed):
        """The split library (variant 'bf16x3') keeps a quad's four weights as (hi, hi | lo, lo) bf16 pairs in the slot's 16 bytes
        (nr_pack.cpp): after the fp32 gather the quad ranges are converted on the device - hi = bf16(w), lo = bf16(w - hi).  Other
        variants: unchanged."""
        if self.variant != 'bf16x3':
            return packed
        cache = self.__dict__.setdefault('_quad_ranges', {})
        if transposed not in cache:
            buf = np.zeros(2 * 128, np.int32)
            n = -int(self.lib.neuray_packed_quad_ranges(int(transposed), C.c_void_p(buf.ctypes.data), 128))
            if n <= 0:
                raise RuntimeError("neuray_packed_quad_ranges failed: " + self.lib.neuray_last_error().decode())
            mask = torch.zeros(packed.numel() // 4, dtype=torch.bool)
            for b, e in buf[:2 * n].reshape(n, 2):
                mask[b // 4:e // 4] = True
            cache[transposed] = mask.to(self.device)
        mask = cache[transposed]
        q = packed.view(-1, 4)
        hi = q.to(torch.bfloat16)
        lo = (q - hi.float()).to(torch.bfloat16)
        both = torch.cat([hi, lo], 1).view(torch.float32)                 # [slots, 8] bf16 = 16 bytes -> [slots, 4] fp32 bit patterns
        return torch.where(mask[:, None], both, q).reshape(-1)

    def flat_pass(self, state_dict, dist_prefix, agg_prefix):
        """Flat natural-layout weights of a pass (include/neuray_hip.h, backward kernels) -> (device tensor, has_vis)."""
        keys = pass_tensor_keys(dist_prefix, agg_prefix)
        n = int(self.lib.neuray_flat_pass_floats())
        flat = torch.zeros(n, dtype=torch.float32)
        has_vis = (dist_prefix + 'vis_decoder.0.weight') in state_dict
        for i, k in enumerate(keys):
            if k not in state_dict:
                if '.vis_decoder.' in k and not has_vis:
                    continue
                raise KeyError("neuray_amd: missing weight %s" % k)
            t = torch.as_tensor(state_dict[k]).detach().to('cpu', torch.float32).reshape(-1)
            off = int(self.lib.neuray_flat_tensor_offset(i))
            assert off + t.numel() <= n and int(self.lib.neuray_flat_tensor_offset(i + 1)) - off == t.numel(), k
            flat[off:off + t.numel()] = t
        return flat.to(self.device), has_vis

    def _flat_offsets(self):
        if '_flat_off' not in self.__dict__:
            self._flat_off = [int(self.lib.neuray_flat_tensor_offset(i)) for i in range(_lib.PASS_TENSORS + 1)]
        return self._flat_off

    def unflatten_pass_grads(self, d_flat, state_dict, dist_prefix, agg_prefix):
        """flat gradient buffer -> {state_dict key: grad tensor shaped like the parameter} (views of d_flat)"""
        off = self._flat_offsets()
        out = {}
        for i, k in enumerate(pass_tensor_keys(dist_prefix, agg_prefix)):
            v = state_dict.get(k)
            if v is not None:
                out[k] = d_flat[off[i]:off[i + 1]].view(v.shape)
        return out

    def pack_pass_t_device(self, flat, has_vis):
        """The transposed layers of the backward pass (nr_layout.h LT_*) from the flat natural layout, on the device:
        packed_t = flat[index] (neuray_pack_pass_t_index_map); true weights, no factors."""
        cache = self.__dict__.setdefault('_pack_t_maps', {})
        if has_vis not in cache:
            n = int(self.lib.neuray_packed_t_floats())
            idx = torch.empty(n, dtype=torch.int32)
            self._check(self.lib.neuray_pack_pass_t_index_map(int(has_vis), C.c_void_p(idx.data_ptr())))
            cache[has_vis] = ((idx >= 0).to(self.device), idx.clamp(min=0).long().to(self.device))
        ok, idx = cache[has_vis]
        return self._split_quads(flat[idx] * ok, 1)

    def zeroed(self, *shapes):
        """Zero-initialised fp32 device tensors of the given shapes carved out of ONE buffer (one fill kernel instead of one per
        tensor: a backward pass needs d_flat, the ray weights' gradient and two or three 15 MB map gradients)."""
        sizes = [int(np.prod(sh)) for sh in shapes]
        pad = [(n + 63) // 64 * 64 for n in sizes]                       # 256-byte aligned pieces (float4 / dwordx4 access)
        buf = torch.zeros(sum(pad), dtype=torch.float32, device=self.device)
        out, off = [], 0
        for sh, n, pn in zip(shapes, sizes, pad):
            out.append(buf[off:off + n].view(*sh))
            off += pn
        return out

    def zero_scratch(self, n):
        """n zeroed floats carved out of a 1 M-float chunk that is filled ONCE (a bump allocator: slices are handed out once and
        never reused, a new chunk is made when the current one is used up) - the fused norm kernels need 2 floats per plane of
        zeroed scratch on every call, 60 calls per encoder pass: one 4 MB fill per ~8 passes instead of 60 tiny fills"""
        n = (int(n) + 63) // 64 * 64
        if self.device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            return torch.zeros(n, dtype=torch.float32, device=self.device)     # a replayed graph must zero its scratch itself, every replay
        st = self.__dict__.get('_zero_pool')
        if st is None or st[1] + n > st[0].numel():
            st = [torch.zeros(max(1 << 20, n), dtype=torch.float32, device=self.device), 0]
            self.__dict__['_zero_pool'] = st
        out = st[0][st[1]:st[1] + n]
        st[1] += n
        return out

    def draw_uniforms(self, shape):
        """torch.rand(shape) on the CPU generator (the reference draws the fine-sampling uniforms there, render_ops.py:205, so a
        seeded run consumes the RNG stream identically) -> device tensor.  On a GPU the draw goes into a pinned staging buffer
        and the upload is asynchronous: a pageable-memory H2D copy would block the host until everything queued before it (the
        coarse pass) has run, which stalls the launch pipeline once per training step."""
        if self.device.type != 'cuda':
            return torch.rand(shape)
        n = int(np.prod(shape))
        st = self.__dict__.get('_u_stage')
        if st is None or st[0].numel() < n:
            st = (torch.empty(n, dtype=torch.float32).pin_memory(), torch.cuda.Event())
            self.__dict__['_u_stage'] = st
        else:
            st[1].synchronize()                                           # the previous upload has left the staging buffer
        host = st[0][:n].view(*shape)
        drawn = torch.rand(shape, out=host)
        if drawn.data_ptr() != host.data_ptr():       # (a replaced torch.rand that ignores `out`: tests feed recorded uniforms)
            host.copy_(drawn)
        dev = host.to(self.device, non_blocking=True)
        st[1].record(torch.cuda.current_stream(self.device))
        return dev

    def render_points_backward(self, qconst, views, coords, depth, flat, has_vis_head, use_vis, d_point_rec, var_bias=0.05,
                               packed=None, saved=None, out=None):
        """Backward of the point kernel: -> (d_flat [flat pass floats], d_ray_feats NHWC [rfn,fh,fw,32], d_img_feats NHWC).
        The register / LDS resident kernel of csrc/nr_kernels_bwd2.h, run as its two halves (tail, front) with a hand-over buffer in
        between; at most 8 reference views (no shipped configuration trains with more: dataset/train_dataset.py:73-74,
        renderer.py:350 - the forward kernels take up to 16).  packed: the forward's PackedPass of the same weights (built from `flat`
        here if absent).  saved: render_pass(save=True)['saved'] of the same inputs (produced here by one more forward if absent)."""
        coords, depth, d_point_rec = self._f32(coords), self._f32(depth), self._f32(d_point_rec)
        rn, dn = depth.shape
        if views.rfn > 8:
            raise NotImplementedError("neuray_amd: the point backward covers at most 8 reference views (got %d); render under "
                                      "torch.no_grad() or train with <= 8 working views (INTEGRATION.md 'Shape limits')" % views.rfn)
        # out: (d_flat, d_ray_feats, d_img_feats) already zeroed by the caller (engine.zeroed: one fill for the whole pass)
        d_flat, d_rf, d_if = out if out is not None else self.zeroed(flat.shape, views.ray_feats.shape, views.img_feats.shape)
        if packed is not None and packed.folded:
            raise ValueError("neuray_amd: the backward kernels take the unfolded pack (pack_pass(fold=False) / pack_pass_device)")
        ho = self.empty(int(self.lib.neuray_points_backward_handover_floats(rn * dn)))
        packed = packed if packed is not None else self.pack_pass_device(flat, bool(has_vis_head))
        pt = self.pack_pass_t_device(flat, bool(has_vis_head))
        if saved is None:
            saved = self.render_points_saved(qconst, views, coords, depth, packed, use_vis, var_bias)
        a = _lib.NeurayPointsBwdArgs(
            qconst.data_ptr(), views.view_const.data_ptr(), coords.data_ptr(), depth.data_ptr(), views.ray_feats.data_ptr(),
            views.img_feats.data_ptr(), views.rgba.data_ptr(), flat.data_ptr(), d_point_rec.data_ptr(), d_flat.data_ptr(),
            d_rf.data_ptr(), d_if.data_ptr(), views.rfn, rn, dn, views.h, views.w,
            views.fh, views.fw, int(has_vis_head), int(bool(use_vis)), float(var_bias),
            packed.dev.data_ptr(), pt.data_ptr(), saved.data_ptr(), ho.data_ptr())
        ev = self._event_pair()
        self._check(self.lib.neuray_render_points_backward(C.byref(a), self._stream()))
        self._event_done(ev, 'points_backward', rn * dn)
        return d_flat, d_rf, d_if

    def self_hit_prob_backward(self, qconst, depth, feats, flat, has_vis_head, use_vis, d_hit, var_bias=0.05, packed=None, d_flat=None):
        """Backward of dist_decoder_rows + self_hit_prob: -> (d_feats [rn,32], d_flat).  One wave per 16 rays, the decoder in registers on
        the packed / transposed packs.  d_flat: an existing flat gradient buffer to accumulate into (the kernel adds atomically) instead
        of a fresh zeroed one."""
        depth, feats, d_hit = self._f32(depth), self._f32(feats), self._f32(d_hit)
        rn, dn = depth.shape
        d_feats = self.empty(rn, 32)
        d_flat = torch.zeros_like(flat) if d_flat is None else d_flat
        if packed is not None and packed.folded:
            raise ValueError("neuray_amd: the backward kernels take the unfolded pack (pack_pass(fold=False) / pack_pass_device)")
        pk = (packed if packed is not None else self.pack_pass_device(flat, bool(has_vis_head))).dev
        pt = self.pack_pass_t_device(flat, bool(has_vis_head))
        self._check(self.lib.neuray_self_hit_prob_backward(
            qconst.data_ptr(), depth.data_ptr(), feats.data_ptr(), pk.data_ptr(), pt.data_ptr(), int(has_vis_head), int(bool(use_vis)),
            float(var_bias), d_hit.data_ptr(), rn, dn, d_feats.data_ptr(), d_flat.data_ptr(), self._stream()))
        return d_feats, d_flat

    def dist_decoder_rows_backward(self, feats, flat, has_vis_head, var_bias, d_mean=None, d_var=None, d_aw=None, d_vis=None, packed=None):
        """Backward of dist_decoder_rows: -> (d_feats [n,32], d_flat).  One wave per 16 rows, the heads in registers on the packed /
        transposed packs; heads without an incoming gradient are skipped."""
        feats = self._f32(feats).reshape(-1, 32)
        n = feats.shape[0]
        g = [self._f32(t).reshape(-1) if t is not None else None for t in (d_mean, d_var, d_aw, d_vis)]
        d_feats = self.empty(n, 32)
        d_flat = torch.zeros_like(flat)
        ptr = lambda t: t.data_ptr() if t is not None else None       # noqa: E731
        pk = (packed if packed is not None else self.pack_pass_device(flat, bool(has_vis_head))).dev
        pt = self.pack_pass_t_device(flat, bool(has_vis_head))
        self._check(self.lib.neuray_dist_decoder_rows_backward(
            feats.data_ptr(), pk.data_ptr(), pt.data_ptr(), n, int(has_vis_head), float(var_bias), ptr(g[0]), ptr(g[1]), ptr(g[2]), ptr(g[3]),
            d_feats.data_ptr(), d_flat.data_ptr(), self._stream()))
        return d_feats, d_flat

    def interpolate_feats_backward(self, d_out, feats_shape, points, h=None, w=None, align_corners=False, mask=None, out=None, staged=None):
        """Backward of interpolate_feats w.r.t. the feature maps: -> d_feats [b,c,fh,fw] (out: a zeroed buffer to add into)"""
        d_out, points = self._f32(d_out), self._f32(points)
        b, c, fh, fw = feats_shape
        n = points.shape[1]
        if h is None and w is None:
            h, w = fh, fw
        d_feats = out if out is not None else torch.zeros(b, c, fh, fw, dtype=torch.float32, device=self.device)
        m = self._f32(mask) if mask is not None else None
        if staged is None:            # many points per map: scatter into a channels-last staging map (coalesced atomics), then transpose-add
            # ... when the points are dense enough to pay for the zero-filled staging map and its transpose-add (1 024 points on
            # 8 x 32 x 800 x 800 maps would allocate and sweep 655 MB for a 128 KB scatter), and the map stays below 256 MB
            staged = n >= 1024 and n * 8 >= fh * fw and b * fh * fw * c * 4 <= (1 << 28)
        if staged:
            tmp = torch.zeros(b, fh, fw, c, dtype=torch.float32, device=self.device)
            self._check(self.lib.neuray_interpolate_feats_backward_staged(
                d_out.data_ptr(), points.data_ptr(), m.data_ptr() if m is not None else None, b, n, c, fh, fw, int(h), int(w),
                int(bool(align_corners)), tmp.data_ptr(), d_feats.data_ptr(), self._stream()))
            return d_feats
        self._check(self.lib.neuray_interpolate_feats_backward(d_out.data_ptr(), points.data_ptr(), m.data_ptr() if m is not None else None,
                                                               b, n, c, fh, fw, int(h), int(w), int(bool(align_corners)),
                                                               d_feats.data_ptr(), self._stream()))
        return d_feats

    def render_rays_backward(self, point_rec, depth, packed, d_pixel, d_hit_prob=None, d_render_depth=None, att_saved=None, d_w=None):
        """Backward of the ray kernel (attention, sigma head, compositing): gradients of a scalar loss w.r.t. the
        per-point records [rn,dn,POINT_REC] (geometry feature 0..15, colour 16..18) and the ray-part weights.
        att_saved: render_pass(save=True)['att_saved'] (the forward's softmax statistics; recomputed if absent).
        -> (d_point_rec [rn,dn,POINT_REC], {state_dict suffix: grad})   (suffixes under `agg_net.agg_impl.`)"""
        point_rec, depth, d_pixel = self._f32(point_rec), self._f32(depth), self._f32(d_pixel)
        rn, dn = depth.shape
        assert point_rec.shape == (rn, dn, _lib.POINT_REC) and d_pixel.shape == (rn, 3)
        dh = self._f32(d_hit_prob) if d_hit_prob is not None else None
        dd = self._f32(d_render_depth) if d_render_depth is not None else None
        d_rec = self.empty(rn, dn, _lib.POINT_REC)
        if d_w is None:
            d_w = torch.zeros(_lib.PACKED_RAY_FLOATS, dtype=torch.float32, device=self.device)
        a = _lib.NeurayRaysBwdArgs(
            point_rec.data_ptr(), depth.data_ptr(), self.posenc(dn).data_ptr(), packed.dev.data_ptr(), d_pixel.data_ptr(),
            dh.data_ptr() if dh is not None else None, dd.data_ptr() if dd is not None else None,
            d_rec.data_ptr(), d_w.data_ptr(), rn, dn, att_saved.data_ptr() if att_saved is not None else None)
        self._check(self.lib.neuray_render_rays_backward(C.byref(a), self._stream()))
        grads = {name: d_w[off:off + int(np.prod(shape))].view(*shape) for name, off, shape in _lib.RAY_WEIGHT_SLOTS}
        return d_rec, grads

    def self_hit_prob(self, qconst, depth, mean, var, aw, vis):
        depth = self._f32(depth)
        rn, dn = depth.shape
        out = self.empty(rn, dn)
        self._check(self.lib.neuray_self_hit_prob(qconst.data_ptr(), depth.data_ptr(), mean.data_ptr(), var.data_ptr(), aw.data_ptr(),
                                                  vis.data_ptr() if vis is not None else None, rn, dn, out.data_ptr(), self._stream()))
        return out

    def interpolate_feats(self, feats, points, h=None, w=None, align_corners=False, mask=None):
        """network/ops.py:14-34 (padding_mode='border' as used on the render path) -> [b,n,c]"""
        feats, points = self._f32(feats), self._f32(points)
        b, c, fh, fw = feats.shape
        n = points.shape[1]
        if h is None and w is None:
            h, w = fh, fw
        out = self.empty(b, n, c)
        m = self._f32(mask) if mask is not None else None
        self._check(self.lib.neuray_interpolate_feats(feats.data_ptr(), points.data_ptr(), m.data_ptr() if m is not None else None,
                                                      b, n, c, fh, fw, int(h), int(w), int(bool(align_corners)), out.data_ptr(),
                                                      self._stream()))
        return out

    def render_pass(self, qconst, views, coords, depth, packed, use_vis, var_bias=0.05, ray_mask_view_num=2,
                    ray_mask_point_num=8, want_depth=False, want_density=False, want_dbg=False, save=False, slot_stats=None):
        """One pass (coarse or fine) over rays `coords` [rn,2] at sample depths `depth` [rn,dn].
        -> dict(hit_prob [rn,dn], pixel [rn,3], ray_mask [rn] bool, render_depth?, density?, dbg?, saved?)
        save (training forward, rfn <= 8): also return 'saved', the cross-view quantities render_points_backward reads instead of
        recomputing them (include/neuray_hip.h NeurayPointsArgs.saved_dev).
        slot_stats: optional int64 device tensor [2] the point kernel adds (view slots run, view slots) to (slot skipping)."""
        coords, depth = self._f32(coords), self._f32(depth)
        rn, dn = depth.shape
        assert coords.shape == (rn, 2)
        if save and packed.folded:
            raise ValueError("neuray_amd: the training forward (save=True) takes the unfolded pack")
        if slot_stats is None:
            slot_stats = self.slot_stats
        s = self._stream()
        rec = self.empty(rn * dn, _lib.POINT_REC)
        dbg = self.empty(rn * dn, views.rfn, _lib.DBG_FIELDS) if want_dbg else None
        saved = self.points_saved_buffer(rn * dn) if save and views.rfn <= 8 else None
        # NEURAY_ARITH_X3: the inference point kernel with two views per wave (a single reference view stays on the fp32 MFMA)
        x3 = packed.dev_x3 is not None and not save and views.rfn >= 2 and self.views_per_wave in (0, 2)
        a = _lib.NeurayPointsArgs(
            qconst.data_ptr(), views.view_const.data_ptr(), coords.data_ptr(), depth.data_ptr(),
            views.ray_feats.data_ptr(), views.img_feats.data_ptr(), views.rgba.data_ptr(),
            packed.dev_x3.data_ptr() if x3 else packed.dev.data_ptr(),
            rec.data_ptr(), dbg.data_ptr() if want_dbg else None,
            views.rfn, rn, dn, views.h, views.w, views.fh, views.fw,
            int(packed.has_vis_head), int(bool(use_vis)), float(var_bias), int(self.views_per_wave),
            saved.data_ptr() if saved is not None else None, int(packed.folded),
            slot_stats.data_ptr() if slot_stats is not None else None, _lib.ARITH_X3 if x3 else _lib.ARITH_F32)
        ev = self._event_pair()
        self._check(self.lib.neuray_render_points(C.byref(a), s))
        self._event_done(ev, 'points', rn * dn)
        out = {'hit_prob': self.empty(rn, dn), 'pixel': self.empty(rn, 3),
               'ray_mask': self.empty(rn, dtype=torch.uint8)}
        if want_depth:
            out['render_depth'] = self.empty(rn)
        if want_density:
            out['density'] = self.empty(rn, dn)
        att = self.empty(rn, dn, _lib.RAY_ATT_SAVE) if save else None
        r = _lib.NeurayRaysArgs(
            rec.data_ptr(), depth.data_ptr(), self.posenc(dn).data_ptr(), packed.dev.data_ptr(),
            out['hit_prob'].data_ptr(), out['pixel'].data_ptr(),
            out['render_depth'].data_ptr() if want_depth else None, out['ray_mask'].data_ptr(),
            out['density'].data_ptr() if want_density else None,
            rn, dn, int(ray_mask_view_num), int(ray_mask_point_num), att.data_ptr() if att is not None else None)
        ev = self._event_pair()
        self._check(self.lib.neuray_render_rays(C.byref(r), s))
        self._event_done(ev, 'rays', rn * dn)
        out['ray_mask'] = out['ray_mask'].bool()
        out['point_rec'] = rec.view(rn, dn, _lib.POINT_REC)
        if want_dbg:
            out['dbg'] = dbg.view(rn, dn, views.rfn, _lib.DBG_FIELDS)
        if saved is not None:
            out['saved'] = saved
        if att is not None:
            out['att_saved'] = att
        return out

    def direct_render(self, qconst, views, coords, depth, view_rec, regs, ground=-15.0, point_rec=None):
        """cfg['use_dr_prediction'] (renderer.py:85-125, sph_solver.py): -> dict(hit_prob [rn,dn], pixel [rn,3], alpha [rn,dn],
        colors [rn,dn,3]).  view_rec: render_pass(want_dbg=True)['dbg'] of the same pass; regs [16]: SphericalHarmonicsSolver.regs;
        point_rec given = cfg['use_nr_color_for_dr'] (the aggregation network's per-point colours instead of the SH fit)."""
        coords, depth = self._f32(coords), self._f32(depth)
        rn, dn = depth.shape
        view_rec, regs = self._f32(view_rec), self._f32(regs.to(self.device))
        assert view_rec.numel() == rn * dn * views.rfn * _lib.DBG_FIELDS and regs.numel() == 16
        alpha = self.empty(rn, dn)
        colors = None if point_rec is not None else self.empty(rn, dn, 3)
        s = self._stream()
        self._check(self.lib.neuray_direct_render_points(
            qconst.data_ptr(), views.view_const.data_ptr(), coords.data_ptr(), depth.data_ptr(), views.rgba.data_ptr(),
            view_rec.data_ptr(), regs.data_ptr(), views.rfn, rn, dn, views.h, views.w, float(ground), alpha.data_ptr(),
            colors.data_ptr() if colors is not None else None, s))
        hit, pix = self.empty(rn, dn), self.empty(rn, 3)
        if colors is not None:
            src, stride, first = colors, 3, 0
        else:
            src, stride, first = self._f32(point_rec), _lib.POINT_REC, 16
            colors = src.view(rn, dn, _lib.POINT_REC)[..., 16:19]
        self._check(self.lib.neuray_direct_render_rays(alpha.data_ptr(), src.data_ptr(), stride, first, rn, dn, hit.data_ptr(),
                                                       pix.data_ptr(), s))
        return {'hit_prob': hit, 'pixel': pix, 'alpha': alpha, 'colors': colors}

    def render_rays(self, point_rec, depth, packed, save=False, ray_mask_view_num=2, ray_mask_point_num=8):
        """The ray kernel alone on per-point records [rn,dn,POINT_REC]: -> dict(hit_prob, pixel, att_saved?)"""
        point_rec, depth = self._f32(point_rec), self._f32(depth)
        rn, dn = depth.shape
        out = {'hit_prob': self.empty(rn, dn), 'pixel': self.empty(rn, 3)}
        att = self.empty(rn, dn, _lib.RAY_ATT_SAVE) if save else None
        r = _lib.NeurayRaysArgs(point_rec.data_ptr(), depth.data_ptr(), self.posenc(dn).data_ptr(), packed.dev.data_ptr(),
                                out['hit_prob'].data_ptr(), out['pixel'].data_ptr(), None, None, None,
                                rn, dn, int(ray_mask_view_num), int(ray_mask_point_num), att.data_ptr() if att is not None else None)
        self._check(self.lib.neuray_render_rays(C.byref(r), self._stream()))
        if att is not None:
            out['att_saved'] = att
        return out

    def points_saved_buffer(self, npts):
        return self.empty(int(self.lib.neuray_points_saved_floats(int(npts))))

    def render_points_saved(self, qconst, views, coords, depth, packed, use_vis, var_bias=0.05):
        """The point kernel alone in its training form: -> the saved buffer of render_pass(save=True) (for callers of
        render_points_backward that did not keep the forward's)."""
        coords, depth = self._f32(coords), self._f32(depth)
        rn, dn = depth.shape
        rec = self.empty(rn * dn, _lib.POINT_REC)
        saved = self.points_saved_buffer(rn * dn)
        a = _lib.NeurayPointsArgs(
            qconst.data_ptr(), views.view_const.data_ptr(), coords.data_ptr(), depth.data_ptr(),
            views.ray_feats.data_ptr(), views.img_feats.data_ptr(), views.rgba.data_ptr(), packed.dev.data_ptr(),
            rec.data_ptr(), None, views.rfn, rn, dn, views.h, views.w, views.fh, views.fw,
            int(packed.has_vis_head), int(bool(use_vis)), float(var_bias), 0, saved.data_ptr(), 0, None, _lib.ARITH_F32)
        self._check(self.lib.neuray_render_points(C.byref(a), self._stream()))
        return saved
