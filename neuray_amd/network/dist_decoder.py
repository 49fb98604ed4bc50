"""Parameter container mirroring network/dist_decoder.py (state_dict names and shapes are the
reference's: {mean,var,aw,vis}_decoder.{0,2,4}.{weight,bias}).  The arithmetic of the decoder on the
render path runs inside the fused HIP point kernel (csrc/nr_kernels.h); this module only owns the
parameters and the reference's default_cfg."""
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


name2dist_decoder = {'mixture_logistics': MixtureLogisticsDistDecoder}
