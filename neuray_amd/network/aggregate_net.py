"""Parameter containers mirroring network/aggregate_net.py and IBRNetWithNeuRay (network/ibrnet.py:239-300):
same submodule names, shapes and initialisation (kaiming_normal_ on the listed MLPs, ibrnet.py:295-300), so
reference checkpoints load with load_state_dict(strict=True).  The forward arithmetic runs in the HIP point
and ray kernels."""
import torch.nn as nn


def _weights_init(m):
    """network/ibrnet.py:104-109"""
    if isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight.data)
        if m.bias is not None:
            nn.init.zeros_(m.bias.data)


class MultiHeadAttention(nn.Module):
    """parameters of network/ibrnet.py:52-70 (4 heads, d_model 16, d_k = d_v = 4, no biases)"""

    def __init__(self, n_head, d_model, d_k, d_v):
        super().__init__()
        self.w_qs = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_ks = nn.Linear(d_model, n_head * d_k, bias=False)
        self.w_vs = nn.Linear(d_model, n_head * d_v, bias=False)
        self.fc = nn.Linear(n_head * d_v, d_model, bias=False)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)


class IBRNetWithNeuRay(nn.Module):
    def __init__(self, neuray_in_dim=32, in_feat_ch=32, n_samples=64):
        super().__init__()
        assert neuray_in_dim == 32 and in_feat_ch == 32, "the HIP path is built for 32-channel features"
        act = nn.ELU(inplace=True)
        self.n_samples = n_samples
        self.ray_dir_fc = nn.Sequential(nn.Linear(4, 16), act, nn.Linear(16, in_feat_ch + 3), act)
        self.base_fc = nn.Sequential(nn.Linear((in_feat_ch + 3) * 5 + neuray_in_dim, 64), act, nn.Linear(64, 32), act)
        self.vis_fc = nn.Sequential(nn.Linear(32, 32), act, nn.Linear(32, 33), act)
        self.vis_fc2 = nn.Sequential(nn.Linear(32, 32), act, nn.Linear(32, 1), nn.Sigmoid())
        self.geometry_fc = nn.Sequential(nn.Linear(32 * 2 + 1, 64), act, nn.Linear(64, 16), act)
        self.ray_attention = MultiHeadAttention(4, 16, 4, 4)
        self.out_geometry_fc = nn.Sequential(nn.Linear(16, 16), act, nn.Linear(16, 1), nn.ReLU())
        self.rgb_fc = nn.Sequential(nn.Linear(32 + 1 + 4, 16), act, nn.Linear(16, 8), act, nn.Linear(8, 1))
        self.neuray_fc = nn.Sequential(nn.Linear(neuray_in_dim, 8), act, nn.Linear(8, 1))
        for m in (self.base_fc, self.vis_fc2, self.vis_fc, self.geometry_fc, self.rgb_fc, self.neuray_fc):
            m.apply(_weights_init)


class DefaultAggregationNet(nn.Module):
    default_cfg = {'sample_num': 64, 'neuray_dim': 32, 'use_img_feats': False}   # aggregate_net.py:17-21

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.default_cfg, **cfg}
        dim = self.cfg['neuray_dim']
        self.agg_impl = IBRNetWithNeuRay(dim, n_samples=self.cfg['sample_num'])
        self.prob_embed = nn.Sequential(nn.Linear(2 + 32, dim), nn.ReLU(), nn.Linear(dim, dim))


name2agg_net = {'default': DefaultAggregationNet}
