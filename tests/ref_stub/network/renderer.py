"""TEST INFRASTRUCTURE: a stand-in for the reference's `network.renderer` on boxes where /root/reference does not exist
(the GPU box), so that neuray_amd.integrate.patch_reference() and neuray_amd.launch can be driven end to end there.

It is shaped like the reference module as far as the drop-in boundary can see it and nothing more:
`NeuralRayBaseRenderer(cfg)` whose __init__ sets ONLY what the reference's sets (cfg, vis_encoder, dist_decoder,
image_encoder, agg_net, fine_*, sph_fitter - network/renderer.py:53-65; no engine / packed-weight attributes),
whose un-patched per-ray methods raise (there is no eager path on that box), a `render()` ray-batch loop, a
`NeuralRayGenRenderer` subclass with `forward(data)`, and `name2network`.  The parameter-holding sub-modules come from
neuray_amd.network (same state_dict names as the reference's, tests/test_c_abi.py)."""
import torch
import torch.nn as nn

from neuray_amd.network.aggregate_net import name2agg_net
from neuray_amd.network.dist_decoder import name2dist_decoder
from neuray_amd.network.encoders import ImageEncoder, name2vis_encoder


class NeuralRayBaseRenderer(nn.Module):
    base_cfg = {
        'vis_encoder_type': 'default', 'vis_encoder_cfg': {}, 'dist_decoder_type': 'mixture_logistics', 'dist_decoder_cfg': {},
        'agg_net_type': 'default', 'agg_net_cfg': {}, 'use_hierarchical_sampling': False, 'fine_agg_net_cfg': {},
        'fine_dist_decoder_cfg': {}, 'fine_depth_sample_num': 64, 'fine_depth_use_all': False, 'ray_batch_num': 2048,
        'depth_sample_num': 64, 'alpha_value_ground_state': -15, 'use_dr_prediction': False, 'use_nr_color_for_dr': False,
        'use_self_hit_prob': False, 'use_ray_mask': True, 'ray_mask_view_num': 2, 'ray_mask_point_num': 8, 'render_depth': False,
    }

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.base_cfg, **cfg}
        self.vis_encoder = name2vis_encoder[self.cfg['vis_encoder_type']](self.cfg['vis_encoder_cfg'])
        self.dist_decoder = name2dist_decoder[self.cfg['dist_decoder_type']](self.cfg['dist_decoder_cfg'])
        self.image_encoder = ImageEncoder()
        self.agg_net = name2agg_net[self.cfg['agg_net_type']](self.cfg['agg_net_cfg'])
        if self.cfg['use_hierarchical_sampling']:
            self.fine_dist_decoder = name2dist_decoder[self.cfg['dist_decoder_type']](self.cfg['fine_dist_decoder_cfg'])
            self.fine_agg_net = name2agg_net[self.cfg['agg_net_type']](self.cfg['fine_agg_net_cfg'])

    def render_by_depth(self, que_depth, que_imgs_info, ref_imgs_info, is_train, is_fine):
        raise RuntimeError("stub: the eager per-ray path of the reference is not available here")

    def fine_render_impl(self, coarse_render_info, que_imgs_info, ref_imgs_info, is_train):
        raise RuntimeError("stub: the eager per-ray path of the reference is not available here")

    def render_impl(self, que_imgs_info, ref_imgs_info, is_train):
        raise RuntimeError("stub: the eager per-ray path of the reference is not available here")

    def predict_self_hit_prob(self, que_imgs_info, que_depth, que_dists, is_fine):
        raise RuntimeError("stub: the eager per-ray path of the reference is not available here")

    def render(self, que_imgs_info, ref_imgs_info, is_train):
        feats = self.image_encoder(ref_imgs_info['imgs'])
        ref_imgs_info['img_feats'] = feats
        ref_imgs_info['ray_feats'] = self.vis_encoder(ref_imgs_info['ray_feats'], feats)
        if is_train and self.cfg['use_self_hit_prob']:
            que_imgs_info['ray_feats'] = self.vis_encoder(que_imgs_info['ray_feats'], self.image_encoder(que_imgs_info['imgs']))
        step, coords, acc = self.cfg['ray_batch_num'], que_imgs_info['coords'], {}
        for start in range(0, coords.shape[1], step):
            que_imgs_info['coords'] = coords[:, start:start + step]
            for k, v in self.render_impl(que_imgs_info, ref_imgs_info, is_train).items():
                if is_train or not k.startswith('hit_prob'):
                    acc.setdefault(k, []).append(v)
        return {k: torch.cat(v, 1) for k, v in acc.items()}


class NeuralRayGenRenderer(NeuralRayBaseRenderer):
    def forward(self, data):
        ref, que = data['ref_imgs_info'].copy(), data['que_imgs_info'].copy()
        return self.render(que, ref, 'eval' not in data)


name2network = {'neuray_gen': NeuralRayGenRenderer}
