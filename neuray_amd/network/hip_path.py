"""The per-ray hot path of NeuralRayBaseRenderer on the HIP kernels, as a mixin that needs nothing from its host class
but what the REFERENCE's own `NeuralRayBaseRenderer.__init__` (network/renderer.py:53-65) sets: `self.cfg`,
`self.dist_decoder`, `self.agg_net` (+ `fine_*` with hierarchical sampling) - modules whose `cfg` dicts and parameter
names are the reference's.  Every piece of state the HIP path adds (engine, packed weights) is created lazily, so the
methods can be grafted onto an already constructed reference instance (neuray_amd/integrate.py) as well as inherited
by the mirror classes of network/renderer.py.

Replaced reference methods (same signatures): `render_by_depth` (renderer.py:168-203), `fine_render_impl` (:205-215),
`render_impl` (:217-226), `predict_self_hit_prob` (:147-155).  Under autograd each pass is a torch.autograd.Function
whose backward runs the backward kernels (network/autograd.py).  Anything the HIP path does not implement raises - it
never silently switches to an eager implementation.
"""
import os

import torch

from ..engine import RenderEngine
from .autograd import PassRun, RenderPassFn, RenderPassSelfFn, SelfHitFn

HOT_PATH_METHODS = ('engine', '_packed_pass', '_same_tensors', '_views', '_query', '_self_hit_prob', '_direct_rendering',
                    'render_by_depth', 'predict_self_hit_prob', 'fine_render_impl', 'render_impl')


class HipRenderPath:
    # ---- engine / weight plumbing ---------------------------------------------------------------
    def engine(self, device):
        eng = self.__dict__.get('_engine')
        if eng is None or eng.device != torch.device(device):
            eng = RenderEngine(device, _test_lib=self.__dict__.get('_engine_test_lib'),
                               variant=self.cfg.get('hip_variant', 'fp32'),
                               arith=os.environ.get('NEURAY_HIP_ARITH') or self.cfg.get('hip_arith', 'f32'))      # (the environment wins: scripts run unchanged)
            self.__dict__['_engine'] = eng
            self.__dict__['_packed'] = {}
        return eng

    def _packed_pass(self, eng, is_fine):
        dist = self.fine_dist_decoder if is_fine else self.dist_decoder
        agg = self.fine_agg_net if is_fine else self.agg_net
        params = list(dist.parameters()) + list(agg.parameters())
        stamp = tuple((p.data_ptr(), p._version) for p in params)
        cache = self.__dict__.setdefault('_packed', {})
        hit = cache.get(is_fine)
        if hit is None or hit[0] != stamp:
            sd = {'d.' + k: v for k, v in dist.state_dict().items()}
            sd.update({'a.' + k: v for k, v in agg.state_dict().items()})
            cache[is_fine] = hit = (stamp, eng.pack_pass(sd, 'd.', 'a.', fold=self.cfg.get('hip_fold_prob_embed', True)))
        return hit[1]

    @staticmethod
    def _same_tensors(entry, info, keys):
        """cache validity: the very same tensor objects (the entry holds references, so an id cannot be recycled) at the
        same in-place version"""
        return entry is not None and all(info.get(k) is t and (t is None or t._version == v) for k, (t, v) in zip(keys, entry))

    def _views(self, eng, ref_imgs_info):
        keys = ('imgs', 'ray_feats', 'img_feats', 'poses', 'Ks', 'depth_range')
        hit = ref_imgs_info.get('_neuray_views')
        if hit is None or hit[2] is not eng or not self._same_tensors(hit[0], ref_imgs_info, keys):
            stamp = [(ref_imgs_info.get(k), None if ref_imgs_info.get(k) is None else ref_imgs_info[k]._version) for k in keys]
            ref_imgs_info['_neuray_views'] = hit = (stamp, eng.prepare_views(ref_imgs_info), eng)
        return hit[1]

    def _query(self, eng, que_imgs_info):
        keys = ('poses', 'Ks', 'depth_range', 'Ks_inv')
        hit = que_imgs_info.get('_neuray_qconst_entry')
        if hit is None or hit[2] is not eng or not self._same_tensors(hit[0], que_imgs_info, keys):
            stamp = [(que_imgs_info.get(k), None if que_imgs_info.get(k) is None else que_imgs_info[k]._version) for k in keys]
            que_imgs_info['_neuray_qconst_entry'] = hit = (stamp, eng.prepare_query(que_imgs_info), eng)
        que_imgs_info['_neuray_qconst'] = hit[1]
        return hit[1]

    # ---- render path ---------------------------------------------------------------------------------
    def render_by_depth(self, que_depth, que_imgs_info, ref_imgs_info, is_train, is_fine):
        """network/renderer.py:168-203.  que_depth [1,rn,dn]."""
        coords = que_imgs_info['coords']
        assert coords.shape[0] == 1 and que_depth.shape[0] == 1, "one query view per call (qn = 1)"
        eng = self.engine(coords.device)
        views = self._views(eng, ref_imgs_info)
        qconst = self._query(eng, que_imgs_info)
        packed = None
        dist = self.fine_dist_decoder if is_fine else self.dist_decoder
        agg = self.fine_agg_net if is_fine else self.agg_net
        use_vis = self.dist_decoder.cfg['use_vis']                              # renderer.py:75: always the coarse decoder
        cfg = self.cfg
        run = PassRun(eng, qconst, views, coords[0].contiguous(), que_depth[0].detach().contiguous(), dist, agg, use_vis,
                      dist.cfg['bias_val'], cfg['ray_mask_view_num'], cfg['ray_mask_point_num'], cfg['render_depth'])
        diff = [p for _, p in run.named_params()] + [ref_imgs_info['ray_feats'], ref_imgs_info['img_feats']]
        if torch.is_grad_enabled() and any(t.requires_grad for t in diff):
            if eng.variant not in ('fp32', 'bf16x3'):
                raise NotImplementedError("neuray_amd: cfg['hip_variant'] = %r is inference only (the training forward's saved "
                                          "quantities and the backward kernels exist in the fp32 and the split 'bf16x3' libraries); "
                                          "render under torch.no_grad() or use one of those" % eng.variant)
            if views.rfn > 8:
                raise NotImplementedError("neuray_amd: the backward kernels cover at most 8 reference views (got %d; no shipped configuration "
                                          "trains with more: dataset/train_dataset.py:73-74); render under torch.no_grad() or reduce the "
                                          "working views" % views.rfn)
            if que_depth.shape[-1] > eng.max_backward_samples:
                raise NotImplementedError("neuray_amd: the backward kernels take at most %d samples per ray and pass"
                                          % eng.max_backward_samples)
            if is_train and cfg['use_self_hit_prob'] and 'ray_feats' in que_imgs_info and 'imgs' in que_imgs_info:
                # the pass and the query view's own hit probabilities (renderer.py:137-155) as one autograd node: one gradient per parameter
                _, _, qh, qw = que_imgs_info['imgs'].shape
                pix, hitp, rmask, rdepth, hit_self = RenderPassSelfFn.apply(
                    run, (qh, qw), ref_imgs_info['ray_feats'], ref_imgs_info['img_feats'], que_imgs_info['ray_feats'],
                    *[p for _, p in run.named_params()])
            else:
                hit_self = None
                pix, hitp, rmask, rdepth = RenderPassFn.apply(run, ref_imgs_info['ray_feats'], ref_imgs_info['img_feats'],
                                                              *[p for _, p in run.named_params()])
            res = {'pixel': pix, 'hit_prob': hitp, 'ray_mask': rmask, 'render_depth': rdepth, 'hit_self': hit_self}
        else:
            packed = self._packed_pass(eng, is_fine)
            res = eng.render_pass(qconst, views, run.coords, run.depth, packed, use_vis=use_vis, var_bias=run.var_bias,
                                  ray_mask_view_num=cfg['ray_mask_view_num'], ray_mask_point_num=cfg['ray_mask_point_num'],
                                  want_depth=cfg['render_depth'], want_dbg=bool(cfg.get('use_dr_prediction', False)))
        outputs = {'pixel_colors_nr': res['pixel'][None], 'hit_prob_nr': res['hit_prob'][None]}
        if cfg.get('use_dr_prediction', False):           # renderer.py:181-185
            outputs.update(self._direct_rendering(eng, qconst, views, run, res, packed, is_fine))
        if is_train and cfg['use_self_hit_prob']:
            outputs['hit_prob_self'] = res['hit_self'][None] if res.get('hit_self') is not None else \
                self._self_hit_prob(que_imgs_info, que_depth, is_fine, run, packed)
        if 'imgs' in que_imgs_info:
            outputs['pixel_colors_gt'] = eng.interpolate_feats(que_imgs_info['imgs'], coords, align_corners=True)
        if cfg['use_ray_mask']:
            outputs['ray_mask'] = res['ray_mask'][None]
        if cfg['render_depth']:
            outputs['render_depth'] = res['render_depth'][None]
        return outputs

    def _direct_rendering(self, eng, qconst, views, run, res, packed, is_fine):
        """renderer.py:85-125 (+ sph_solver.py) on the dr kernels -> {'pixel_colors_dr', 'hit_prob_dr'}.  The per-view hit
        probabilities / visibilities come from the point kernel's per-view record; under autograd the pass itself ran as an
        autograd.Function without that record, so the point kernel is run once more for it, and the dr outputs are returned
        DETACHED: the backward kernels cover the losses of every shipped config (loss.py use_dr_loss: false everywhere)."""
        cfg = self.cfg
        rec = res.get('dbg')
        point_rec = res.get('point_rec')
        if rec is None:
            if cfg.get('use_dr_loss') or cfg.get('use_dr_fine_loss'):
                # the reference back-propagates those losses through direct_rendering (loss.py: RenderLoss reads pixel_colors_dr); the
                # backward kernels do not, and a loss term that silently contributes no gradient is worse than an error
                raise NotImplementedError("neuray_amd: cfg use_dr_loss / use_dr_fine_loss need gradients through direct_rendering "
                                          "(renderer.py:85-125), which the HIP backward kernels do not provide; the prediction-only "
                                          "use_dr_prediction (no dr loss) is supported")
            if not self.__dict__.get('_dr_grad_warned'):
                import warnings
                warnings.warn("neuray_amd: use_dr_prediction under autograd - pixel_colors_dr / hit_prob_dr are computed but carry "
                              "no gradient (cfg use_dr_loss is off in every shipped config)")
                self.__dict__['_dr_grad_warned'] = True
            with torch.no_grad():
                packed = packed if packed is not None else self._packed_pass(eng, is_fine)
                again = eng.render_pass(qconst, views, run.coords, run.depth, packed, use_vis=run.use_vis, var_bias=run.var_bias,
                                        want_dbg=True)
            rec, point_rec = again['dbg'], again['point_rec']
        fitter = getattr(self, 'sph_fitter', None)
        regs = fitter.regs if fitter is not None and hasattr(fitter, 'regs') else \
            torch.tensor([0.0] + [0.001] * 3 + [0.005] * 5 + [0.05] * 7, dtype=torch.float32)       # sph_solver.py:6-12, degree 3
        with torch.no_grad():
            dr = eng.direct_render(qconst, views, run.coords, run.depth, rec, regs, ground=float(cfg['alpha_value_ground_state']),
                                   point_rec=point_rec if cfg.get('use_nr_color_for_dr', False) else None)
        return {'pixel_colors_dr': dr['pixel'][None], 'hit_prob_dr': dr['hit_prob'][None]}

    def predict_self_hit_prob(self, que_imgs_info, que_depth, que_dists, is_fine):
        """network/renderer.py:147-155, the reference's signature (`que_dists` is recomputed inside the kernel from
        `que_depth` and the query depth range, as depth2inv_dists does, so the argument is not read)."""
        return self._self_hit_prob(que_imgs_info, que_depth, is_fine)

    def _self_hit_prob(self, que_imgs_info, que_depth, is_fine, run=None, packed=None):
        """decode the query view's own visibility feature along its rays (renderer.py:137-155)"""
        coords = que_imgs_info['coords']
        eng = self.engine(coords.device)
        qconst = self._query(eng, que_imgs_info)
        _, _, h, w = que_imgs_info['imgs'].shape
        dec = self.fine_dist_decoder if is_fine else self.dist_decoder
        agg = self.fine_agg_net if is_fine else self.agg_net
        if torch.is_grad_enabled() and (que_imgs_info['ray_feats'].requires_grad or any(p.requires_grad for p in dec.parameters())):
            srun = PassRun(eng, qconst, None, coords[0].contiguous(), que_depth[0].detach().contiguous(), dec, agg,
                           dec.cfg['use_vis'], dec.cfg['bias_val'], 0, 0, False)
            return SelfHitFn.apply(srun, h, w, que_imgs_info['ray_feats'], *[p for _, p in srun.dist_params()])[None]
        feats = eng.interpolate_feats(que_imgs_info['ray_feats'], coords, h, w, align_corners=False)      # [1,rn,32]
        mean, var, vis, aw = eng.dist_decoder_rows(feats[0], packed if packed is not None else self._packed_pass(eng, is_fine),
                                                   dec.cfg['bias_val'])
        vis = vis if dec.cfg['use_vis'] else None
        return eng.self_hit_prob(qconst, que_depth[0], mean, var, aw, vis)[None]

    def fine_render_impl(self, coarse_render_info, que_imgs_info, ref_imgs_info, is_train):
        """network/renderer.py:205-215"""
        depth, hit = coarse_render_info['depth'], coarse_render_info['hit_prob']
        eng = self.engine(depth.device)
        fdn = self.cfg['fine_depth_sample_num']
        u = None
        if is_train:   # the reference draws the uniforms on the CPU (render_ops.py:205)
            u = coarse_render_info.get('_neuray_u')           # render_impl drew them before launching the coarse pass
            u = (u if u is not None else eng.draw_uniforms(list(depth.shape[:-1]) + [fdn]))[0]
        qconst = self._query(eng, que_imgs_info)
        if '_neuray_fine_range' in que_imgs_info:            # view q > 0 of a multi-view query: view 0's range (quirk A.9.6)
            qconst = eng.prepare_query({**que_imgs_info, 'depth_range': que_imgs_info['_neuray_fine_range']})
        que_depth = eng.sample_fine_depth(qconst, depth[0].contiguous(), hit[0].detach().contiguous(),
                                          fdn, use_all=self.cfg['fine_depth_use_all'], u=u)
        return self.render_by_depth(que_depth[None], que_imgs_info, ref_imgs_info, is_train, True)

    def render_impl(self, que_imgs_info, ref_imgs_info, is_train):
        """network/renderer.py:217-226.  qn > 1 query views (no shipped caller has them, the tensors carry the dimension) go
        through the fused kernels one view at a time; the fine sampling of every view is normalised with view 0's depth
        range, as in the reference (render_ops.py:183,225)."""
        coords = que_imgs_info['coords']
        qn = coords.shape[0]
        if qn > 1:
            per_view = [k for k, v in que_imgs_info.items() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == qn]
            outs = []
            for q in range(qn):
                sub = {k: (v[q:q + 1] if k in per_view else v) for k, v in que_imgs_info.items() if not k.startswith('_')}
                sub['_neuray_fine_range'] = que_imgs_info['depth_range'][0:1]
                outs.append(self.render_impl(sub, ref_imgs_info, is_train))
            return {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}
        eng = self.engine(coords.device)
        rn = coords.shape[1]
        # the fine-sampling uniforms (render_ops.py:205: torch.rand on the CPU generator, the only draw of render_impl) are taken
        # BEFORE the coarse pass is launched and uploaded asynchronously, so the host never waits for the coarse pass
        u = eng.draw_uniforms([1, rn, self.cfg['fine_depth_sample_num']]) if (is_train and self.cfg['use_hierarchical_sampling']) else None
        # (the pass's only draw from the CPU generator is done: a host class whose NEXT draw can be made ahead of time - the generalisation
        # renderer's depth-loss pixel permutation, renderer.py:272-278 - starts it now, in the reference's order, while the kernels are queued)
        hook = self.__dict__.pop('_neuray_after_fine_draw', None)
        if hook is not None:
            hook()
        que_depth = eng.sample_coarse_depth(que_imgs_info['depth_range'], rn, self.cfg['depth_sample_num'])[None]
        outputs = self.render_by_depth(que_depth, que_imgs_info, ref_imgs_info, is_train, False)
        if self.cfg['use_hierarchical_sampling']:
            coarse = {'depth': que_depth, 'hit_prob': outputs['hit_prob_nr'], '_neuray_u': u}
            for k, v in self.fine_render_impl(coarse, que_imgs_info, ref_imgs_info, is_train).items():
                outputs[k + '_fine'] = v
        return outputs
