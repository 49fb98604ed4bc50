"""Mirror of network/renderer.py's NeuralRayBaseRenderer call surface on top of the HIP render path.

Same cfg keys (base_cfg, network/renderer.py:25-52), same state_dict names for the hot-path modules, same
`render_impl(que_imgs_info, ref_imgs_info, is_train) -> dict` contract (output keys/shapes: SURVEY.md 8(b)).
What differs is underneath: instead of ~700 small PyTorch kernels per ray batch, each pass is three HIP
launches (point kernel, ray kernel, fine-sampling kernel) through include/neuray_hip.h.

Under autograd (training) each pass is a torch.autograd.Function whose backward runs the backward kernels
(network/autograd.py).  Anything the HIP path does not implement raises - it never silently switches to an eager
implementation.
"""
import torch
import torch.nn as nn

from ..engine import RenderEngine
from .autograd import PassRun, RenderPassFn, SelfHitFn
from .aggregate_net import name2agg_net
from .dist_decoder import name2dist_decoder
from .encoders import ImageEncoder, name2vis_encoder


class NeuralRayBaseRenderer(nn.Module):
    base_cfg = {   # network/renderer.py:25-52
        'vis_encoder_type': 'default', 'vis_encoder_cfg': {},
        'dist_decoder_type': 'mixture_logistics', 'dist_decoder_cfg': {},
        'agg_net_type': 'default', 'agg_net_cfg': {},
        'use_hierarchical_sampling': False, 'fine_agg_net_cfg': {}, 'fine_dist_decoder_cfg': {},
        'fine_depth_sample_num': 64, 'fine_depth_use_all': False,
        'ray_batch_num': 2048, 'depth_sample_num': 64, 'alpha_value_ground_state': -15,
        'use_dr_prediction': False, 'use_nr_color_for_dr': False, 'use_self_hit_prob': False,
        'use_ray_mask': True, 'ray_mask_view_num': 2, 'ray_mask_point_num': 8, 'render_depth': False,
        # not a reference key: also build image_encoder / vis_encoder (network/encoders.py), which makes the state_dict
        # equal to the reference base renderer's and lets render() start from images + initial ray_feats
        'build_encoders': False,
    }

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.base_cfg, **cfg}
        if self.cfg['use_dr_prediction']:
            raise NotImplementedError("neuray_amd: use_dr_prediction (direct rendering, renderer.py:85-125) is outside "
                                      "the HIP render path; every shipped config has it off")
        self.dist_decoder = name2dist_decoder[self.cfg['dist_decoder_type']](self.cfg['dist_decoder_cfg'])
        self.agg_net = name2agg_net[self.cfg['agg_net_type']](self.cfg['agg_net_cfg'])
        if self.cfg['use_hierarchical_sampling']:
            self.fine_dist_decoder = name2dist_decoder[self.cfg['dist_decoder_type']](self.cfg['fine_dist_decoder_cfg'])
            self.fine_agg_net = name2agg_net[self.cfg['agg_net_type']](self.cfg['fine_agg_net_cfg'])
        if self.cfg['build_encoders']:       # renderer.py:56-58
            self.vis_encoder = name2vis_encoder[self.cfg['vis_encoder_type']](self.cfg['vis_encoder_cfg'])
            self.image_encoder = ImageEncoder()
            # the reference registers the (unused unless use_dr_prediction) spherical-harmonics regulariser as a buffer
            # (renderer.py:65, sph_solver.py:10-12): kept so that its checkpoints load strictly
            self.sph_fitter = nn.Module()
            self.sph_fitter.register_buffer('regs', torch.tensor([0.0] + [0.001] * 3 + [0.005] * 5 + [0.05] * 7, dtype=torch.float32))
        self._engine = None
        self._engine_test_lib = None     # CPU test-suite hook (emulator build of the kernels)
        self._packed = {}

    # ---- engine / weight plumbing ---------------------------------------------------------------
    def engine(self, device):
        if self._engine is None or self._engine.device != torch.device(device):
            self._engine = RenderEngine(device, _test_lib=self._engine_test_lib)
            self._packed = {}
        return self._engine

    def _packed_pass(self, eng, is_fine):
        dist = self.fine_dist_decoder if is_fine else self.dist_decoder
        agg = self.fine_agg_net if is_fine else self.agg_net
        params = list(dist.parameters()) + list(agg.parameters())
        stamp = tuple((p.data_ptr(), p._version) for p in params)
        hit = self._packed.get(is_fine)
        if hit is None or hit[0] != stamp:
            sd = {'d.' + k: v for k, v in dist.state_dict().items()}
            sd.update({'a.' + k: v for k, v in agg.state_dict().items()})
            self._packed[is_fine] = (stamp, eng.pack_pass(sd, 'd.', 'a.'))
        return self._packed[is_fine][1]

    def _views(self, eng, ref_imgs_info):
        key = tuple((id(ref_imgs_info[k]), ref_imgs_info[k]._version) for k in ('imgs', 'ray_feats', 'img_feats', 'poses', 'Ks'))
        hit = ref_imgs_info.get('_neuray_views')
        if hit is None or hit[0] != key:
            ref_imgs_info['_neuray_views'] = (key, eng.prepare_views(ref_imgs_info))
        return ref_imgs_info['_neuray_views'][1]

    # ---- render path ---------------------------------------------------------------------------------
    def render_by_depth(self, que_depth, que_imgs_info, ref_imgs_info, is_train, is_fine):
        """network/renderer.py:168-203.  que_depth [1,rn,dn]."""
        coords = que_imgs_info['coords']
        assert coords.shape[0] == 1 and que_depth.shape[0] == 1, "one query view per call (qn = 1)"
        eng = self.engine(coords.device)
        views = self._views(eng, ref_imgs_info)
        qconst = que_imgs_info.get('_neuray_qconst')
        if qconst is None:
            qconst = eng.prepare_query(que_imgs_info)
            que_imgs_info['_neuray_qconst'] = qconst
        packed = None
        dist = self.fine_dist_decoder if is_fine else self.dist_decoder
        agg = self.fine_agg_net if is_fine else self.agg_net
        use_vis = self.dist_decoder.cfg['use_vis']                              # renderer.py:75: always the coarse decoder
        run = PassRun(eng, qconst, views, coords[0].contiguous(), que_depth[0].detach().contiguous(), dist, agg, use_vis,
                      dist.cfg['bias_val'], self.cfg['ray_mask_view_num'], self.cfg['ray_mask_point_num'], self.cfg['render_depth'])
        diff = [p for _, p in run.named_params()] + [ref_imgs_info['ray_feats'], ref_imgs_info['img_feats']]
        self._grad_pass = torch.is_grad_enabled() and any(t.requires_grad for t in diff)
        if self._grad_pass:
            if que_depth.shape[-1] > 64:
                raise NotImplementedError("neuray_amd: the backward kernels take at most 64 samples per ray and pass "
                                          "(fine_depth_use_all with 64 + 64 is forward-only)")
            pix, hitp, rmask, rdepth = RenderPassFn.apply(run, ref_imgs_info['ray_feats'], ref_imgs_info['img_feats'],
                                                          *[p for _, p in run.named_params()])
            res = {'pixel': pix, 'hit_prob': hitp, 'ray_mask': rmask, 'render_depth': rdepth}
        else:
            packed = self._packed_pass(eng, is_fine)
            res = eng.render_pass(qconst, views, run.coords, run.depth, packed, use_vis=use_vis, var_bias=run.var_bias,
                                  ray_mask_view_num=self.cfg['ray_mask_view_num'], ray_mask_point_num=self.cfg['ray_mask_point_num'],
                                  want_depth=self.cfg['render_depth'])
        outputs = {'pixel_colors_nr': res['pixel'][None], 'hit_prob_nr': res['hit_prob'][None]}
        if is_train and self.cfg['use_self_hit_prob']:
            outputs['hit_prob_self'] = self.predict_self_hit_prob(que_imgs_info, que_depth, is_fine, run, packed)
        if 'imgs' in que_imgs_info:
            outputs['pixel_colors_gt'] = eng.interpolate_feats(que_imgs_info['imgs'], coords, align_corners=True)
        if self.cfg['use_ray_mask']:
            outputs['ray_mask'] = res['ray_mask'][None]
        if self.cfg['render_depth']:
            outputs['render_depth'] = res['render_depth'][None]
        return outputs

    def predict_self_hit_prob(self, que_imgs_info, que_depth, is_fine, run=None, packed=None):
        """network/renderer.py:137-155: decode the query view's own visibility feature along its rays."""
        coords = que_imgs_info['coords']
        eng = self.engine(coords.device)
        _, _, h, w = que_imgs_info['imgs'].shape
        if run is not None and torch.is_grad_enabled() and (
                que_imgs_info['ray_feats'].requires_grad or any(p.requires_grad for p in run.dist.parameters())):
            dec = run.dist
            srun = PassRun(eng, run.qconst, None, run.coords, run.depth, run.dist, run.agg, dec.cfg['use_vis'], dec.cfg['bias_val'],
                           0, 0, False)
            return SelfHitFn.apply(srun, h, w, que_imgs_info['ray_feats'], *[p for _, p in srun.named_params()])[None]
        feats = eng.interpolate_feats(que_imgs_info['ray_feats'], coords, h, w, align_corners=False)      # [1,rn,32]
        dec = self.fine_dist_decoder if is_fine else self.dist_decoder
        mean, var, vis, aw = eng.dist_decoder_rows(feats[0], packed if packed is not None else self._packed_pass(eng, is_fine),
                                                   dec.cfg['bias_val'])
        vis = vis if dec.cfg['use_vis'] else None
        return eng.self_hit_prob(que_imgs_info['_neuray_qconst'], que_depth[0], mean, var, aw, vis)[None]

    def fine_render_impl(self, coarse_render_info, que_imgs_info, ref_imgs_info, is_train):
        """network/renderer.py:205-215"""
        depth, hit = coarse_render_info['depth'], coarse_render_info['hit_prob']
        eng = self.engine(depth.device)
        fdn = self.cfg['fine_depth_sample_num']
        u = None
        if is_train:   # the reference draws the uniforms on the CPU (render_ops.py:205)
            u = torch.rand(list(depth.shape[:-1]) + [fdn])[0]
        que_depth = eng.sample_fine_depth(que_imgs_info['_neuray_qconst'], depth[0].contiguous(), hit[0].detach().contiguous(),
                                          fdn, use_all=self.cfg['fine_depth_use_all'], u=u)
        return self.render_by_depth(que_depth[None], que_imgs_info, ref_imgs_info, is_train, True)

    def render_impl(self, que_imgs_info, ref_imgs_info, is_train):
        """network/renderer.py:217-226"""
        coords = que_imgs_info['coords']
        eng = self.engine(coords.device)
        rn = coords.shape[1]
        que_depth = eng.sample_coarse_depth(que_imgs_info['depth_range'], rn, self.cfg['depth_sample_num'])[None]
        outputs = self.render_by_depth(que_depth, que_imgs_info, ref_imgs_info, is_train, False)
        if self.cfg['use_hierarchical_sampling']:
            coarse = {'depth': que_depth, 'hit_prob': outputs['hit_prob_nr']}
            for k, v in self.fine_render_impl(coarse, que_imgs_info, ref_imgs_info, is_train).items():
                outputs[k + '_fine'] = v
        return outputs

    def render(self, que_imgs_info, ref_imgs_info, is_train):
        """network/renderer.py:228-254.  ref_imgs_info either carries 'img_feats' and the encoded 'ray_feats' already, or
        (cfg['build_encoders']) 'imgs' and the initial 'ray_feats', which go through image_encoder / vis_encoder first."""
        if 'img_feats' not in ref_imgs_info:
            # renderer.py:229-235: encode the reference images, refine the initial ray_feats with them
            if not self.cfg['build_encoders']:
                raise NotImplementedError("neuray_amd: render() needs ref_imgs_info['img_feats'] (and the encoded 'ray_feats'), "
                                          "or a renderer built with cfg['build_encoders'] = True")
            ref_img_feats = self.image_encoder(ref_imgs_info['imgs'])
            ref_imgs_info['img_feats'] = ref_img_feats
            ref_imgs_info['ray_feats'] = self.vis_encoder(ref_imgs_info['ray_feats'], ref_img_feats)
            if is_train and self.cfg['use_self_hit_prob']:
                que_imgs_info['ray_feats'] = self.vis_encoder(que_imgs_info['ray_feats'], self.image_encoder(que_imgs_info['imgs']))
        if 'ray_feats' not in ref_imgs_info:
            raise NotImplementedError("neuray_amd: render() needs ref_imgs_info['ray_feats']")
        ray_batch_num = self.cfg['ray_batch_num']
        coords = que_imgs_info['coords']
        ray_num = coords.shape[1]
        render_info_all = {}
        for ray_id in range(0, ray_num, ray_batch_num):
            que_imgs_info['coords'] = coords[:, ray_id:ray_id + ray_batch_num]
            render_info = self.render_impl(que_imgs_info, ref_imgs_info, is_train)
            for k, v in render_info.items():
                if is_train or (not k.startswith('hit_prob')):
                    render_info_all.setdefault(k, []).append(v)
        que_imgs_info['coords'] = coords
        return {k: torch.cat(v, 1) for k, v in render_info_all.items()}


name2network = {'neuray_base': NeuralRayBaseRenderer}
