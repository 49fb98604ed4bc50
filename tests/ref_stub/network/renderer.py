"""TEST INFRASTRUCTURE: a stand-in for the reference's `network.renderer` on boxes where /root/reference does not exist
(the GPU box), so that neuray_amd.integrate.patch_reference() and neuray_amd.launch can be driven end to end there.

It is shaped like the reference module as far as the drop-in boundary can see it and nothing more:
`NeuralRayBaseRenderer(cfg)` whose __init__ sets ONLY what the reference's sets (cfg, vis_encoder, dist_decoder,
image_encoder, agg_net, fine_*, sph_fitter - network/renderer.py:53-65; no engine / packed-weight attributes),
whose un-patched per-ray methods raise (there is no eager path on that box), a `render()` ray-batch loop, a
`NeuralRayGenRenderer` subclass with `forward(data)`, a `NeuralRayFtRenderer` subclass (the scene-resident fine-tuning
renderer, network/renderer.py:331-546: `default_cfg`, a `ray_feats` ParameterList and NO `touched_views`, `slice_imgs_info`
going through the module-level `to_cuda` / `imgs_info_slice`, `train_step` with the reference's np.random draw order,
`validate_step`; its constructor does not read a dataset - the tests set the scene attributes, as
tests/golden/make_golden.py does for the real class), and `name2network`.  The parameter-holding sub-modules come from
neuray_amd.network (same state_dict names as the reference's, tests/test_c_abi.py)."""
import numpy as np
import torch
import torch.nn as nn

from neuray_amd.network.renderer import sample_train_coords

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


def to_cuda(info):
    """utils/base_utils.py `to_cuda`: every tensor of a dict moved to the GPU (tests swap it for identity on CPU legs)"""
    return {k: v.cuda() if torch.is_tensor(v) else v for k, v in info.items()}


def imgs_info_slice(info, idx):
    """utils/imgs_info.py `imgs_info_slice`: the per-view tensors of an imgs_info dict at `idx`"""
    return {k: v[idx] for k, v in info.items() if torch.is_tensor(v)}


class NeuralRayFtRenderer(NeuralRayBaseRenderer):
    default_cfg = {
        'database_name': 'nerf_synthetic/lego/black_400', 'database_split': 'val_all', 'ref_pad_interval': 16,
        'use_consistent_depth_range': True, 'gen_cfg': None, 'use_validation': True, 'validate_initialization': True,
        'init_view_num': 8, 'init_src_view_num': 3, 'include_self_prob': 0.01, 'neighbor_view_num': 8, 'neighbor_pool_ratio': 2,
        'train_ray_num': 512, 'foreground_ratio': 0.5, 'ray_feats_res': [200, 200], 'ray_feats_dim': 32,
    }

    def __init__(self, cfg):
        raise RuntimeError("stub: no dataset layer here - build with __new__ + NeuralRayBaseRenderer.__init__ and set "
                           "ref_ids / ref_imgs_info / val_imgs_info / ref_dist_idx / val_dist_idx / ray_feats")

    def slice_imgs_info(self, ref_idx, val_idx, is_train):
        ref = to_cuda(imgs_info_slice(self.ref_imgs_info, torch.from_numpy(np.asarray(ref_idx)).long()))
        ref['ray_feats'] = torch.cat([self.ray_feats[int(i)] for i in ref_idx], 0)
        one = torch.from_numpy(np.asarray([val_idx])).long()
        if is_train:
            que = imgs_info_slice(self.ref_imgs_info, one)
            fg = que['masks'][0, 0].cpu().numpy() > 0
            coords = sample_train_coords(fg, self.cfg['train_ray_num'], self.cfg['foreground_ratio']).reshape(1, -1, 2)
        else:
            que = imgs_info_slice(self.val_imgs_info, one)
            hn, wn = que['imgs'].shape[-2:]
            coords = np.stack(np.meshgrid(np.arange(wn), np.arange(hn)), -1).reshape(1, -1, 2).astype(np.float32)
        que['coords'] = torch.from_numpy(coords)
        que = to_cuda(que)
        if is_train and self.cfg['use_self_hit_prob']:
            que['ray_feats'] = self.ray_feats[int(val_idx)]
        return ref, que

    def validate_step(self, val_idx):
        ref, que = self.slice_imgs_info(self.val_dist_idx[val_idx][:self.cfg['neighbor_view_num']], val_idx, False)
        with torch.no_grad():
            out = self.render(que, ref, False)
        ref.pop('ray_feats'), ref.pop('img_feats')
        out.update({'ref_imgs_info': ref, 'que_imgs_info': que})
        return out

    def train_step(self):
        # the reference's draw order: query view, include-self coin, shuffle of the neighbour pool, then the ray coordinates
        que_i = np.random.randint(0, len(self.ref_ids))
        pool = self.ref_dist_idx[que_i]
        if np.random.random() > self.cfg['include_self_prob']:
            pool = pool[1:]
        pool = pool[:self.cfg['neighbor_view_num'] * self.cfg['neighbor_pool_ratio']]
        np.random.shuffle(pool)
        ref, que = self.slice_imgs_info(pool[:self.cfg['neighbor_view_num']], que_i, True)
        out = self.render(que.copy(), ref.copy(), True)
        for d in (ref, que):
            d.pop('ray_feats', None), d.pop('img_feats', None)
        out['que_imgs_info'] = que
        return out

    def forward(self, data):
        return self.train_step() if 'eval' not in data else self.validate_step(data['index'])


name2network = {'neuray_gen': NeuralRayGenRenderer, 'neuray_ft': NeuralRayFtRenderer}
