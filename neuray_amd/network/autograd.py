"""torch.autograd glue of the HIP render path: the forward runs the forward kernels, the backward the backward kernels
(include/neuray_hip.h: neuray_render_rays_backward, neuray_render_points_backward, neuray_self_hit_prob_backward,
neuray_interpolate_feats_backward).  Nothing here computes with PyTorch ops besides layout permutes and slicing."""
import torch

from .. import _lib


class PassRun:
    """Everything one render pass needs besides the differentiable tensors."""

    def __init__(self, eng, qconst, views, coords, depth, dist, agg, use_vis, var_bias, mask_view_num, mask_point_num, want_depth):
        self.eng, self.qconst, self.views, self.coords, self.depth = eng, qconst, views, coords, depth
        self.dist, self.agg, self.use_vis, self.var_bias = dist, agg, use_vis, var_bias
        self.mask_view_num, self.mask_point_num, self.want_depth = mask_view_num, mask_point_num, want_depth

    def named_params(self):
        # (the walk over the module tree is cached on the decoder: parameters keep their identity across steps)
        hit = getattr(self.dist, '_neuray_named', None)
        if hit is None or hit[0] is not self.agg:
            named = [('d.' + k, v) for k, v in self.dist.named_parameters()] + [('a.' + k, v) for k, v in self.agg.named_parameters()]
            hit = (self.agg, named)
            self.dist._neuray_named = hit
        return hit[1]

    def dist_params(self):
        """the dist decoder's parameters only (the self-hit path of renderer.py:137-155 does not touch the aggregation net)"""
        return [(k, v) for k, v in self.named_params() if k.startswith('d.')]

    def state(self):
        return dict(self.named_params())       # only keys and shapes are used (unflatten_pass_grads)

    def device_weights(self):
        """(flat natural-layout weights, packed weights, has_vis) built on the device from the current parameters;
        cached on the decoder module until a parameter changes (the render pass and the self-hit path of one step,
        and gradient-accumulation steps, share it)"""
        named = self.named_params()
        stamp = tuple((p.data_ptr(), p._version) for _, p in named)
        hit = getattr(self.dist, '_neuray_devw', None)
        if hit is None or hit[0] != stamp or hit[1] is not self.agg:
            flat, has_vis = self.eng.flat_pass_device(dict(named), 'd.', 'a.')
            hit = (stamp, self.agg, (flat, self.eng.pack_pass_device(flat, has_vis), has_vis))
            self.dist._neuray_devw = hit
        return hit[2]


class RenderPassFn(torch.autograd.Function):
    """(ray_feats NCHW, img_feats NCHW, *params) -> pixel [rn,3], hit_prob [rn,dn], ray_mask [rn], render_depth [rn]

    Memory: the training forward keeps, until the backward has run, the per-point records (80 B per sample point), the ray
    kernel's softmax statistics (96 B per sample point) and - for <= 8 views - the point kernel's cross-view quantities
    (`saved`, neuray_points_saved_floats: 29 KB per 16-point tile = 1.9 KB per sample point).  At the training shapes (512 rays
    x 64 + 64 samples) that is 120 MB per step; a grad-enabled pass over an inference-sized batch (32 768 rays x 64 samples)
    would hold 3.9 GB, so evaluation belongs under torch.no_grad(), as the reference's own callers do it
    (renderer.py:244-246, :513)."""

    @staticmethod
    def forward(ctx, run, ray_feats, img_feats, *params):
        flat, packed, has_vis = run.device_weights()
        ctx.flat, ctx.has_vis = flat, has_vis
        res = run.eng.render_pass(run.qconst, run.views, run.coords, run.depth, packed, use_vis=run.use_vis,
                                  var_bias=run.var_bias, ray_mask_view_num=run.mask_view_num,
                                  ray_mask_point_num=run.mask_point_num, want_depth=True, save=True)
        ctx.run, ctx.packed = run, packed
        ctx.point_saved, ctx.att_saved = res.get('saved'), res.get('att_saved')                # cross-view quantities of the point kernel, read by its backward
        ctx.save_for_backward(res['point_rec'])
        ctx.mark_non_differentiable(res['ray_mask'])
        return res['pixel'], res['hit_prob'], res['ray_mask'], res['render_depth']

    @staticmethod
    def backward(ctx, d_pixel, d_hit, _d_mask, d_depth):
        run, eng = ctx.run, ctx.run.eng
        point_rec, = ctx.saved_tensors
        rn, dn = run.depth.shape
        if d_pixel is None:
            d_pixel = torch.zeros(rn, 3, device=point_rec.device)
        # every gradient buffer of the pass out of one zero-filled allocation (one fill kernel)
        d_flat, d_w, d_rf, d_if = eng.zeroed(ctx.flat.shape, (_lib.PACKED_RAY_FLOATS,), run.views.ray_feats.shape, run.views.img_feats.shape)
        d_rec, g_ray = eng.render_rays_backward(point_rec, run.depth, ctx.packed, d_pixel.contiguous(),
                                                d_hit.contiguous() if d_hit is not None else None,
                                                d_depth.contiguous() if d_depth is not None else None, att_saved=ctx.att_saved, d_w=d_w)
        sd = run.state()
        d_flat, d_rf, d_if = eng.render_points_backward(run.qconst, run.views, run.coords, run.depth, ctx.flat, ctx.has_vis,
                                                        run.use_vis, d_rec, var_bias=run.var_bias, packed=ctx.packed,
                                                        saved=ctx.point_saved, out=(d_flat, d_rf, d_if))
        grads = eng.unflatten_pass_grads(d_flat, sd, 'd.', 'a.')
        for name, g in g_ray.items():
            grads['a.agg_impl.' + name] = g
        # (the NHWC gradient buffers viewed as NCHW = channels-last tensors: what the channels-last encoders' backward wants,
        # and no relayout pass here)
        return (None, d_rf.permute(0, 3, 1, 2), d_if.permute(0, 3, 1, 2)) + \
            tuple(grads[k] for k, _ in run.named_params())      # views of one buffer: no per-parameter copies


class RenderPassSelfFn(torch.autograd.Function):
    """RenderPassFn and SelfHitFn of the same pass as ONE node: (ray_feats, img_feats, que ray_feats, *params) -> pixel, hit_prob,
    ray_mask, render_depth, hit_prob_self.  Both backward kernels accumulate into one flat gradient buffer, so every parameter receives a
    single gradient (as two nodes the dist decoder's 18 tensors arrive twice per pass and autograd adds them: 36 small kernels a step)."""

    @staticmethod
    def forward(ctx, run, hw, ray_feats, img_feats, que_ray_feats, *params):
        eng = run.eng
        flat, packed, has_vis = run.device_weights()
        ctx.flat, ctx.has_vis = flat, has_vis
        res = eng.render_pass(run.qconst, run.views, run.coords, run.depth, packed, use_vis=run.use_vis,
                              var_bias=run.var_bias, ray_mask_view_num=run.mask_view_num,
                              ray_mask_point_num=run.mask_point_num, want_depth=True, save=True)
        feats = eng.interpolate_feats(que_ray_feats, run.coords[None], hw[0], hw[1], align_corners=False)      # [1,rn,32]
        mean, var, vis, aw = eng.dist_decoder_rows(feats[0], packed, run.var_bias)
        self_use_vis = run.dist.cfg['use_vis']
        hit_self = eng.self_hit_prob(run.qconst, run.depth, mean, var, aw, vis if self_use_vis else None)
        ctx.run, ctx.packed, ctx.hw, ctx.que_shape, ctx.self_use_vis = run, packed, hw, tuple(que_ray_feats.shape), self_use_vis
        ctx.point_saved, ctx.att_saved = res.get('saved'), res.get('att_saved')
        ctx.save_for_backward(res['point_rec'], feats)
        ctx.mark_non_differentiable(res['ray_mask'])
        return res['pixel'], res['hit_prob'], res['ray_mask'], res['render_depth'], hit_self

    @staticmethod
    def backward(ctx, d_pixel, d_hit, _d_mask, d_depth, d_hit_self):
        run, eng = ctx.run, ctx.run.eng
        point_rec, feats = ctx.saved_tensors
        rn, dn = run.depth.shape
        if d_pixel is None:
            d_pixel = torch.zeros(rn, 3, device=point_rec.device)
        # every gradient buffer of the pass - flat weights, ray-part weights, both reference maps, the query view's map - out of
        # one zero-filled allocation: one fill kernel instead of five
        want_map = d_hit_self is not None and ctx.needs_input_grad[4]
        d_flat, d_w, d_rf, d_if, d_map = eng.zeroed(ctx.flat.shape, (_lib.PACKED_RAY_FLOATS,), run.views.ray_feats.shape,
                                                    run.views.img_feats.shape, ctx.que_shape if want_map else (1,))
        d_rec, g_ray = eng.render_rays_backward(point_rec, run.depth, ctx.packed, d_pixel.contiguous(),
                                                d_hit.contiguous() if d_hit is not None else None,
                                                d_depth.contiguous() if d_depth is not None else None, att_saved=ctx.att_saved, d_w=d_w)
        d_flat, d_rf, d_if = eng.render_points_backward(run.qconst, run.views, run.coords, run.depth, ctx.flat, ctx.has_vis,
                                                        run.use_vis, d_rec, var_bias=run.var_bias, packed=ctx.packed,
                                                        saved=ctx.point_saved, out=(d_flat, d_rf, d_if))
        if d_hit_self is not None:
            d_feats, _ = eng.self_hit_prob_backward(run.qconst, run.depth, feats[0], ctx.flat, ctx.has_vis, ctx.self_use_vis,
                                                    d_hit_self.contiguous(), var_bias=run.var_bias, packed=ctx.packed, d_flat=d_flat)
            if want_map:
                eng.interpolate_feats_backward(d_feats[None], ctx.que_shape, run.coords[None], ctx.hw[0], ctx.hw[1], align_corners=False,
                                               out=d_map)
        if not want_map:
            d_map = None
        grads = eng.unflatten_pass_grads(d_flat, run.state(), 'd.', 'a.')
        for name, g in g_ray.items():
            grads['a.agg_impl.' + name] = g
        return (None, None, d_rf.permute(0, 3, 1, 2), d_if.permute(0, 3, 1, 2), d_map) + \
            tuple(grads[k] for k, _ in run.named_params())


class SelfHitFn(torch.autograd.Function):
    """(que ray_feats NCHW [1,32,fh,fw], *run.dist_params()) -> hit_prob_self [rn,dn]   (renderer.py:137-155)"""

    @staticmethod
    def forward(ctx, run, h, w, que_ray_feats, *params):
        eng = run.eng
        flat, packed, has_vis = run.device_weights()
        ctx.flat, ctx.has_vis, ctx.packed = flat, has_vis, packed
        feats = eng.interpolate_feats(que_ray_feats, run.coords[None], h, w, align_corners=False)      # [1,rn,32]
        mean, var, vis, aw = eng.dist_decoder_rows(feats[0], packed, run.var_bias)
        vis = vis if run.use_vis else None
        out = eng.self_hit_prob(run.qconst, run.depth, mean, var, aw, vis)
        ctx.run, ctx.hw, ctx.shape = run, (h, w), tuple(que_ray_feats.shape)
        ctx.save_for_backward(feats)
        return out

    @staticmethod
    def backward(ctx, d_hit):
        run, eng = ctx.run, ctx.run.eng
        feats, = ctx.saved_tensors
        sd = run.state()
        d_feats, d_flat = eng.self_hit_prob_backward(run.qconst, run.depth, feats[0], ctx.flat, ctx.has_vis, run.use_vis,
                                                     d_hit.contiguous(), var_bias=run.var_bias, packed=ctx.packed)
        d_map = eng.interpolate_feats_backward(d_feats[None], ctx.shape, run.coords[None], ctx.hw[0], ctx.hw[1], align_corners=False)
        grads = eng.unflatten_pass_grads(d_flat, sd, 'd.', 'a.')
        return (None, None, None, d_map) + tuple(grads[k] for k, _ in run.dist_params())
