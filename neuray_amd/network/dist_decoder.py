"""Mirror of network/dist_decoder.py (state_dict names and shapes are the reference's:
{mean,var,aw,vis}_decoder.{0,2,4}.{weight,bias}).  On the render path the decoder runs fused inside the HIP point
kernel (csrc/nr_kernels.h); called on its own - `forward`, `predict_mean`, `predict_aw`, as the reference's
Gen renderer does for its depth loss (renderer.py:280-316) - it runs the HIP rows kernel (neuray_dist_decoder_rows) and,
under autograd, its backward kernel (neuray_dist_decoder_rows_backward)."""
import torch
import torch.nn as nn


class AddBias(nn.Module):
    """network/ops.py:78-84"""

    def __init__(self, val):
        super().__init__()
        self.val = val

    def forward(self, x):
        return x + self.val


def _head(dim, out, final):
    return nn.Sequential(nn.Linear(dim, dim), nn.ELU(), nn.Linear(dim, dim), nn.ELU(), nn.Linear(dim, out), *final)


class _RowsFn(torch.autograd.Function):
    """feats [..., 32], *decoder params -> mean [...,2], var [...,2], aw [...,1], vis [...,1] (zeros without a vis head)"""

    @staticmethod
    def forward(ctx, dec, feats, *params):
        eng = dec._engine(feats.device)
        named = {'d.' + k: v for k, v in dec.named_parameters()}
        flat, has_vis = eng.flat_pass_device(named, 'd.', 'a.', allow_missing_agg=True)
        packed = eng.pack_pass_device(flat, has_vis)
        mean, var, vis, aw = eng.dist_decoder_rows(feats, packed, dec.cfg['bias_val'])
        ctx.dec, ctx.flat, ctx.has_vis, ctx.packed = dec, flat, has_vis, packed
        ctx.set_materialize_grads(False)          # an output the loss does not use arrives as None: its head's backward is skipped
        ctx.save_for_backward(feats.detach())
        if vis is None:
            vis = torch.zeros_like(aw)
        return mean, var, aw, vis

    @staticmethod
    def backward(ctx, d_mean, d_var, d_aw, d_vis):
        dec = ctx.dec
        feats, = ctx.saved_tensors
        eng = dec._engine(feats.device)
        d_feats, d_flat = eng.dist_decoder_rows_backward(feats, ctx.flat, ctx.has_vis, dec.cfg['bias_val'], d_mean, d_var, d_aw,
                                                         d_vis if ctx.has_vis else None, packed=ctx.packed)
        sd = {'d.' + k: v.detach() for k, v in dec.named_parameters()}
        grads = eng.unflatten_pass_grads(d_flat, sd, 'd.', 'a.')
        # (views of the freshly allocated d_flat, as RenderPassFn returns them: no per-parameter copy)
        return (None, d_feats.view_as(feats)) + tuple(grads['d.' + k] for k, _ in dec.named_parameters())


class MixtureLogisticsDistDecoder(nn.Module):
    default_cfg = {'feats_dim': 32, 'bias_val': 0.05, 'use_vis': True}   # network/dist_decoder.py:54-58

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        dim = self.cfg['feats_dim']
        assert dim == 32, "the HIP path is built for feats_dim = 32"
        self.mean_decoder = _head(dim, 2, [nn.Softplus()])
        self.var_decoder = _head(dim, 2, [nn.Softplus(), AddBias(self.cfg['bias_val'])])
        self.aw_decoder = _head(dim, 1, [nn.Sigmoid()])
        if self.cfg['use_vis']:
            self.vis_decoder = _head(dim, 1, [nn.Sigmoid()])
        self._eng = None
        self._engine_test_lib = None     # CPU test-suite hook (emulator build of the kernels)

    def _engine(self, device):
        from ..engine import RenderEngine
        from . import render_ops
        if self._engine_test_lib is None:
            return render_ops.engine_for(device)       # one engine per device, shared with the free functions
        if self._eng is None or self._eng.device != torch.device(device):
            self._eng = RenderEngine(device, _test_lib=self._engine_test_lib)
        return self._eng

    def _rows(self, feats):
        return _RowsFn.apply(self, feats, *[p for _, p in self.named_parameters()])

    def forward(self, feats):
        """network/dist_decoder.py:99-107 -> (mean, var, vis or None, aw)"""
        mean, var, aw, vis = self._rows(feats)
        return mean, var, (vis if self.cfg['use_vis'] else None), aw

    def predict_mean(self, prj_ray_feats):
        """network/dist_decoder.py:146-148"""
        return self._rows(prj_ray_feats)[0]

    def predict_aw(self, prj_ray_feats):
        """network/dist_decoder.py:150-151"""
        return self._rows(prj_ray_feats)[2]

    def decode_alpha_value(self, alpha_value):
        """network/dist_decoder.py:142-144"""
        return torch.sigmoid(alpha_value)

    def compute_prob(self, *args, **kwargs):
        raise NotImplementedError("neuray_amd: compute_prob runs fused inside the HIP point kernel (render path) and inside "
                                  "neuray_self_hit_prob (a19); there is no stand-alone tensor version")


name2dist_decoder = {'mixture_logistics': MixtureLogisticsDistDecoder}
