"""Backward kernels against PyTorch autograd of the same formulas (the eager port of the reference's op sequence,
oracle/torch_eager_port.py, whose end-to-end gradients are pinned to the reference's own autograd by
tests/golden/case_g_grads.npz in test_oracle_golden.py).

First kernel: the ray kernel's backward (positional encoding, ray attention, LayerNorm, sigma head, compositing):
gradients w.r.t. the per-point records and the ray-part weights."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_weights
from emu_util import emu_lib
from oracle import torch_eager_port as tep

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]
IP = 'agg_net.agg_impl.'


def ray_part_torch(w, g, colors, nvalid, depth):
    """ibrnet.py:356-360 + renderer.py:157-166 + render_ops.py:72-80 on (geometry feature, colour, #valid views)."""
    rn, dn, _ = g.shape
    g = g + tep._posenc(dn, g.device)
    lin = lambda x, name: F.linear(x, w[IP + name + '.weight'], w.get(IP + name + '.bias'))
    q = lin(g, 'ray_attention.w_qs').view(rn, dn, 4, 4).transpose(1, 2)
    k = lin(g, 'ray_attention.w_ks').view(rn, dn, 4, 4).transpose(1, 2)
    v = lin(g, 'ray_attention.w_vs').view(rn, dn, 4, 4).transpose(1, 2)
    att = (q / 2) @ k.transpose(2, 3)
    att = att.masked_fill(((nvalid > 1).float().view(rn, 1, dn, 1)) == 0, -1e9)
    o = (F.softmax(att, -1) @ v).transpose(1, 2).reshape(rn, dn, 16)
    o = F.layer_norm(lin(o, 'ray_attention.fc') + g, (16,), w[IP + 'ray_attention.layer_norm.weight'],
                     w[IP + 'ray_attention.layer_norm.bias'], 1e-6)
    sigma = F.relu(lin(F.elu(lin(o, 'out_geometry_fc.0')), 'out_geometry_fc.2'))[..., 0].masked_fill(nvalid < 1, 0.)
    alpha = 1 - torch.exp(-sigma)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]
    hit = alpha * T
    return (hit.unsqueeze(-1) * colors).sum(1), hit, (hit * depth).sum(1)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('rn,dn,with_aux', [(5, 16, True), (9, 64, False), (3, 7, True), (3, 128, True), (5, 96, False), (2, 65, True)])
def test_rays_backward_matches_autograd(rn, dn, with_aux, backend):
    from neuray_amd.engine import RenderEngine
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    eng = RenderEngine(dev, _test_lib=emu_lib() if backend == 'emu' else None)
    weights = load_weights(False)
    packed = eng.pack_pass(weights, 'dist_decoder.', 'agg_net.')
    rng = np.random.RandomState(rn * 100 + dn)
    rec = np.zeros((rn, dn, 20), np.float32)
    rec[..., :16] = rng.randn(rn, dn, 16) * 0.7
    rec[..., 16:19] = rng.rand(rn, dn, 3)
    rec[..., 19] = rng.randint(0, 4, size=(rn, dn))           # 0 -> sigma forced to 0, <= 1 -> masked attention row
    depth = np.sort(rng.rand(rn, dn).astype(np.float32) * 4 + 2, -1)
    d_pixel = rng.randn(rn, 3).astype(np.float32)
    d_hit = rng.randn(rn, dn).astype(np.float32) if with_aux else None
    d_dep = rng.randn(rn).astype(np.float32) if with_aux else None

    w = {k: torch.from_numpy(v.copy()).double().requires_grad_(k.startswith(IP + 'ray_attention') or k.startswith(IP + 'out_geometry_fc'))
         for k, v in weights.items() if k.startswith(IP)}
    g = torch.from_numpy(rec[..., :16]).double().requires_grad_(True)
    col = torch.from_numpy(rec[..., 16:19]).double().requires_grad_(True)
    pix, hit, dep = ray_part_torch(w, g, col, torch.from_numpy(rec[..., 19]).double(), torch.from_numpy(depth).double())
    loss = (pix * torch.from_numpy(d_pixel).double()).sum()
    if with_aux:
        loss = loss + (hit * torch.from_numpy(d_hit).double()).sum() + (dep * torch.from_numpy(d_dep).double()).sum()
    loss.backward()

    t = lambda a: torch.from_numpy(a).to(dev) if a is not None else None
    d_rec, gw = eng.render_rays_backward(t(rec), t(depth), packed, t(d_pixel), t(d_hit), t(d_dep))
    d_rec = d_rec.cpu().numpy()
    scale = lambda ref: max(1.0, float(np.abs(ref).max()))
    want_g, want_c = g.grad.numpy(), col.grad.numpy()
    assert np.abs(d_rec[..., :16] - want_g).max() <= 2e-4 * scale(want_g)
    assert np.abs(d_rec[..., 16:19] - want_c).max() <= 1e-5 * scale(want_c)
    assert np.all(d_rec[..., 19] == 0)
    for name, grad in gw.items():
        want = w[IP + name].grad.numpy()
        assert grad.shape == want.shape, name
        assert np.abs(grad.cpu().numpy() - want).max() <= 3e-4 * scale(want), name
    # the same with the attention statistics the training forward leaves behind instead of recomputing them
    if dn <= 128:
        fwd = eng.render_rays(t(rec), t(depth), packed, save=True)
        assert np.abs(fwd['pixel'].cpu().numpy() - pix.detach().numpy()).max() <= 2e-5 * scale(pix.detach().numpy())
        d_rec2, gw2 = eng.render_rays_backward(t(rec), t(depth), packed, t(d_pixel), t(d_hit), t(d_dep), att_saved=fwd['att_saved'])
        assert np.abs(d_rec2.cpu().numpy() - d_rec).max() <= 1e-5 * scale(d_rec)
        for name, grad in gw2.items():
            assert np.abs((grad - gw[name]).cpu().numpy()).max() <= 1e-5 * scale(gw[name].cpu().numpy()), name


@pytest.mark.parametrize('backend', BACKENDS)
def test_rays_backward_rejects_long_rays(backend):
    from neuray_amd.engine import RenderEngine
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    eng = RenderEngine(dev, _test_lib=emu_lib() if backend == 'emu' else None)
    packed = eng.pack_pass(load_weights(False), 'dist_decoder.', 'agg_net.')
    z = torch.zeros(1, 129, 20, device=dev)
    with pytest.raises(RuntimeError, match='dn=129'):
        eng.render_rays_backward(z, torch.ones(1, 129, device=dev), packed, torch.zeros(1, 3, device=dev))


# ---- whole pass: ray backward chained into the point backward, against autograd of the eager port -----------------
def _pass_case(rfn, rn, dn, use_vis_head, seed):
    from neuray_amd import synthetic
    que, ref = synthetic.make_scene(48, 64, rfn, seed=seed)
    rng = np.random.RandomState(seed + 1)
    que['coords'] = (rng.rand(1, rn, 2) * np.array([63, 47])).astype(np.float32)
    que['Ks_inv'] = torch.inverse(torch.from_numpy(que['Ks'])).numpy()
    return que, ref, load_weights(use_vis_head), rng


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('rfn,rn,dn,vis_head', [(3, 5, 8, False), (8, 3, 6, True), (2, 4, 5, False), (1, 3, 5, False),
                                                  (7, 2, 4, True), (5, 7, 9, True)])
def test_pass_backward_matches_autograd(rfn, rn, dn, vis_head, backend):
    from neuray_amd.engine import RenderEngine
    from oracle import neuray_oracle as orc
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    eng = RenderEngine(dev, _test_lib=emu_lib() if backend == 'emu' else None)
    que, ref, weights, rng = _pass_case(rfn, rn, dn, vis_head, seed=40 + rfn)
    lw_pix = rng.randn(rn, 3).astype(np.float32)
    lw_hit = rng.randn(rn, dn).astype(np.float32)
    depth = orc.sample_depth(que['depth_range'], rn, dn)                                            # [1,rn,dn]
    depth = (depth * (1.0 + 0.02 * rng.rand(1, rn, dn))).astype(np.float32); depth.sort(-1)            # de-regularised

    # ---- autograd of the eager port (float32, CPU)
    w = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in weights.items()
         if k.startswith(('dist_decoder.', 'agg_net.'))}
    tq = {k: torch.from_numpy(v) for k, v in que.items()}
    tr = {k: torch.from_numpy(v.copy()) for k, v in ref.items()}
    tr['ray_feats'].requires_grad_(True); tr['img_feats'].requires_grad_(True)
    cfg = {'coarse_use_vis': vis_head, 'fine_use_vis': True}
    out = tep.render_pass(w, cfg, torch.from_numpy(depth), tq, tr, False)
    loss = (out['pixel_colors_nr'][0] * torch.from_numpy(lw_pix)).sum() + (out['hit_prob_nr'][0] * torch.from_numpy(lw_hit)).sum()
    loss.backward()

    # ---- ours
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    views = eng.prepare_views({k: t(v) for k, v in ref.items()})
    qc = eng.prepare_query({k: t(v) for k, v in que.items()})
    packed = eng.pack_pass(weights, 'dist_decoder.', 'agg_net.')
    fwd = eng.render_pass(qc, views, t(que['coords'][0]), t(depth[0]), packed, use_vis=vis_head)
    assert np.abs(fwd['pixel'].cpu().numpy() - out['pixel_colors_nr'][0].detach().numpy()).max() <= 2e-4
    d_rec, g_ray = eng.render_rays_backward(fwd['point_rec'], t(depth[0]), packed, t(lw_pix), t(lw_hit))
    flat, has_vis = eng.flat_pass(weights, 'dist_decoder.', 'agg_net.')
    d_flat, d_rf, d_if = eng.render_points_backward(qc, views, t(que['coords'][0]), t(depth[0]), flat, has_vis, vis_head, d_rec)
    grads = eng.unflatten_pass_grads(d_flat, weights, 'dist_decoder.', 'agg_net.')
    for name, g in g_ray.items():
        grads['agg_net.agg_impl.' + name] = g

    def close(got, want, name, rel=2e-3):
        want = want.detach().numpy() if torch.is_tensor(want) else want
        got = got.detach().cpu().numpy()
        tol = rel * max(1e-3, float(np.abs(want).max()))
        assert got.shape == want.shape, name
        assert np.abs(got - want).max() <= tol, (name, float(np.abs(got - want).max()), float(np.abs(want).max()))

    for k, p_ in w.items():
        want = p_.grad if p_.grad is not None else torch.zeros_like(p_)
        close(grads[k], want, k)
    close(d_rf.permute(0, 3, 1, 2), tr['ray_feats'].grad, 'ray_feats')
    close(d_if.permute(0, 3, 1, 2), tr['img_feats'].grad, 'img_feats')


# ---- training mode end to end: render_impl under autograd against the REFERENCE's own autograd -------------------
@pytest.mark.parametrize('backend', BACKENDS)
def test_training_gradients_match_reference_autograd(backend):
    """tests/golden/case_g_grads.npz was produced by the reference: NeuralRayBaseRenderer.render_impl(is_train=True),
    loss = sum(w_k * output_k) over pixel_colors_nr[_fine] and hit_prob_self[_fine], loss.backward().  The mirror renderer
    runs the same step through the HIP forward + backward kernels; the fine-sampling uniforms come from the same seeded
    CPU generator (render_ops.py:205)."""
    import os
    from conftest import GOLDEN_DIR
    from neuray_amd.network.renderer import NeuralRayBaseRenderer
    z = np.load(os.path.join(GOLDEN_DIR, 'case_g_grads.npz'))
    cfg = __import__('ast').literal_eval(str(z['cfg_json']))
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    r = NeuralRayBaseRenderer(cfg)
    sd = {k: torch.from_numpy(v) for k, v in load_weights(False).items()}
    r.load_state_dict(sd, strict=True)
    r.train()
    if backend == 'emu':
        r._engine_test_lib = emu_lib()
    r = r.to(dev)
    que = {k[4:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith('que.') and k != 'que.Ks_inv'}
    ref = {k[4:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith('ref.')}
    for t_ in (ref['ray_feats'], ref['img_feats'], que['ray_feats']):
        t_.requires_grad_(True)
    torch.manual_seed(4321)
    out = r.render_impl(que, ref, True)
    keys = ('pixel_colors_nr', 'pixel_colors_nr_fine', 'hit_prob_self', 'hit_prob_self_fine')
    for k in ('pixel_colors_nr', 'hit_prob_self'):
        np.testing.assert_allclose(out[k].detach().cpu().numpy(), z['out.' + k], atol=2e-4)
    loss = sum((torch.from_numpy(z['lw.' + k]).to(dev) * out[k]).sum() for k in keys)
    assert abs(float(loss.detach()) - float(z["loss"])) <= 5e-3
    loss.backward()

    def close(got, want, name, rel=5e-3):
        scale = max(1e-3, float(np.abs(want).max()))
        err = float(np.max(np.abs(got - want)))
        assert err <= rel * scale, (name, err, scale)

    for k, p_ in r.named_parameters():
        g = p_.grad.cpu().numpy() if p_.grad is not None else np.zeros(tuple(p_.shape), np.float32)
        close(g, z['grad.' + k], k)
    close(ref['ray_feats'].grad.cpu().numpy(), z['grad.ref.ray_feats'], 'ref.ray_feats')
    close(ref['img_feats'].grad.cpu().numpy(), z['grad.ref.img_feats'], 'ref.img_feats')
    close(que['ray_feats'].grad.cpu().numpy(), z['grad.que.ray_feats'], 'que.ray_feats')


@pytest.mark.parametrize('vis_head', [False, True])
def test_device_packing_matches_host_packer(vis_head):
    """Training packs the forward weights on the device as packed = flat[index] * scale (neuray_pack_pass_index_map);
    that must be the host packer's result up to the rounding of one fp32 multiply."""
    from neuray_amd.engine import RenderEngine
    eng = RenderEngine('cpu', _test_lib=emu_lib())
    w = load_weights(vis_head)
    host = eng.pack_pass(w, 'dist_decoder.', 'agg_net.').dev
    flat, hv = eng.flat_pass_device({k: torch.from_numpy(v) for k, v in w.items()}, 'dist_decoder.', 'agg_net.')
    assert hv == vis_head
    dev = eng.pack_pass_device(flat, hv).dev
    assert torch.equal(host != 0, dev != 0)
    rel = ((host - dev).abs() / host.abs().clamp_min(1e-30))[host != 0]
    assert float(rel.max()) <= 1.2e-7
    flat_h, _ = eng.flat_pass(w, 'dist_decoder.', 'agg_net.')
    assert torch.equal(flat_h, flat)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('use_vis', [False, True])
def test_dist_decoder_module_forward_backward(use_vis, backend):
    """The mirror MixtureLogisticsDistDecoder called on its own (predict_mean for the Gen renderer's depth loss,
    renderer.py:280-316): HIP rows kernel + its backward against the module's own nn.Sequential heads in PyTorch."""
    from neuray_amd.network.dist_decoder import MixtureLogisticsDistDecoder
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    torch.manual_seed(5)
    dec = MixtureLogisticsDistDecoder({'use_vis': use_vis})
    if backend == 'emu':
        dec._engine_test_lib = emu_lib()
    dec = dec.to(dev)
    feats = torch.randn(3, 37, 32, device=dev, requires_grad=True)
    wm, wv, wa = torch.randn(3, 37, 2, device=dev), torch.randn(3, 37, 2, device=dev), torch.randn(3, 37, 1, device=dev)
    mean, var, vis, aw = dec(feats)
    loss = (mean * wm).sum() + (var * wv).sum() + (aw * wa).sum() + (vis.sum() * 0.5 if use_vis else 0.0) + dec.predict_mean(feats)[..., 0].sum()
    loss.backward()
    got = {k: p_.grad.clone() for k, p_ in dec.named_parameters()}
    got_f = feats.grad.clone()
    dec.zero_grad(); feats.grad = None
    mean_t, var_t, aw_t = dec.mean_decoder(feats), dec.var_decoder(feats), dec.aw_decoder(feats)
    loss_t = (mean_t * wm).sum() + (var_t * wv).sum() + (aw_t * wa).sum() + mean_t[..., 0].sum()
    if use_vis:
        loss_t = loss_t + dec.vis_decoder(feats).sum() * 0.5
    loss_t.backward()
    assert torch.allclose(mean, mean_t, atol=1e-5) and torch.allclose(var, var_t, atol=1e-5) and torch.allclose(aw, aw_t, atol=1e-5)
    for k, p_ in dec.named_parameters():
        scale = max(1e-3, float(p_.grad.abs().max()))
        assert float((got[k] - p_.grad).abs().max()) <= 2e-3 * scale, k
    assert float((got_f - feats.grad).abs().max()) <= 2e-3 * max(1e-3, float(feats.grad.abs().max()))


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('use_vis', [False, True])
def test_rows_backward_skips_the_heads_without_a_gradient(use_vis, backend):
    """neuray_dist_decoder_rows_backward (one wave per 16 rows, heads in registers): a head whose gradient pointer is NULL contributes
    exactly what a zero gradient contributes - nothing (predict_mean's backward passes the mean head only, renderer.py:280-316); the
    full gradient against autograd is test_predict_mean_backward_matches_autograd / test_self_hit_prob_backward_matches_autograd"""
    from neuray_amd.network.dist_decoder import MixtureLogisticsDistDecoder
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    torch.manual_seed(6)
    dec = MixtureLogisticsDistDecoder({'use_vis': use_vis})
    if backend == 'emu':
        dec._engine_test_lib = emu_lib()
    dec = dec.to(dev)
    eng = dec._engine(torch.device(dev))
    flat, has_vis = eng.flat_pass_device({'d.' + k: v for k, v in dec.named_parameters()}, 'd.', 'a.', allow_missing_agg=True)
    n = 53                                                       # not a multiple of 16: a partly filled tile
    feats = torch.randn(n, 32, device=dev)
    gm, gv, ga, gs = torch.randn(n, 2, device=dev), torch.randn(n, 2, device=dev), torch.randn(n, 1, device=dev), torch.randn(n, 1, device=dev)
    z2, z1 = torch.zeros(n, 2, device=dev), torch.zeros(n, 1, device=dev)
    for heads, zeros in (((gm, None, None, None), (gm, z2, z1, z1 if use_vis else None)),
                         ((None, gv, ga, None), (z2, gv, ga, z1 if use_vis else None))):
        f1, w1 = eng.dist_decoder_rows_backward(feats, flat, has_vis, 0.05, *zeros)
        f2, w2 = eng.dist_decoder_rows_backward(feats, flat, has_vis, 0.05, *heads)
        assert float((f1 - f2).abs().max()) <= 1e-5 * max(1.0, float(f1.abs().max()))
        assert float((w1 - w2).abs().max()) <= 2e-5 * max(1.0, float(w1.abs().max()))
        assert float(w1.abs().max()) > 0


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('shape', [(2, 32, 13, 17, 300), (3, 5, 8, 40, 77)])
def test_staged_interpolate_backward_equals_the_direct_scatter_and_autograd(shape, backend):
    """neuray_interpolate_feats_backward_staged (scatter into a channels-last map, transpose-add) against the direct channel-major scatter
    and against autograd of F.grid_sample as interpolate_feats calls it (network/ops.py:14-34), incl. masked points, border taps, a
    channel count that is not a multiple of 32 and accumulation into a non-zero buffer"""
    import torch.nn.functional as F
    from neuray_amd.engine import RenderEngine
    b, c, fh, fw, n = shape
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    eng = RenderEngine(torch.device(dev), _test_lib=emu_lib() if backend == 'emu' else None)
    g = torch.Generator().manual_seed(4)
    h, w = 4 * fh, 4 * fw
    pts = (torch.rand(b, n, 2, generator=g) * torch.tensor([w + 3.0, h + 3.0]) - 1.5)          # some outside: border padding
    mask = (torch.rand(b, n, generator=g) > 0.2).float()
    d_out = torch.randn(b, n, c, generator=g)
    base = torch.randn(b, c, fh, fw, generator=g)
    feats = torch.zeros(b, c, fh, fw, requires_grad=True)
    norm = torch.stack([pts[..., 0] / (w - 1), pts[..., 1] / (h - 1)], -1) * 2 - 1
    out = F.grid_sample(feats, norm[:, :, None], mode='bilinear', padding_mode='border', align_corners=False)[..., 0].permute(0, 2, 1)
    (out * mask[..., None] * d_out).sum().backward()
    got = {}
    for staged in (True, False):
        buf = base.clone().to(dev)
        got[staged] = eng.interpolate_feats_backward(d_out.to(dev), (b, c, fh, fw), pts.to(dev), h, w, align_corners=False, mask=mask.to(dev),
                                                     out=buf, staged=staged).cpu() - base
    scale = float(feats.grad.abs().max())
    assert float((got[True] - got[False]).abs().max()) <= 1e-5 * scale
    assert float((got[True] - feats.grad).abs().max()) <= 1e-4 * scale


@pytest.mark.parametrize('backend', BACKENDS)
def test_adam_steps_reduce_the_render_loss(backend):
    """A few optimiser steps through the HIP forward + backward kernels (weights AND the per-view ray_feats, as the
    reference's fine-tuning does, renderer.py:404-437) must reduce a fixed render loss."""
    from neuray_amd import synthetic
    from neuray_amd.network.renderer import NeuralRayBaseRenderer
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': 8,
           'fine_depth_sample_num': 8, 'agg_net_cfg': {'sample_num': 8}, 'fine_agg_net_cfg': {'sample_num': 8}}
    torch.manual_seed(3)
    r = NeuralRayBaseRenderer(cfg).train()
    if backend == 'emu':
        r._engine_test_lib = emu_lib()
    r = r.to(dev)
    que, ref = synthetic.make_scene(32, 48, 3, seed=9)
    rng = np.random.RandomState(9)
    que['coords'] = (rng.rand(1, 12, 2) * np.array([47, 31])).astype(np.float32)
    tq = {k: torch.from_numpy(v).to(dev) for k, v in que.items()}
    tr = {k: torch.from_numpy(v).to(dev) for k, v in ref.items()}
    ray_feats = torch.nn.Parameter(tr['ray_feats'].clone())
    target = torch.rand(1, 12, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    opt = torch.optim.Adam(list(r.parameters()) + [ray_feats], lr=2e-3)
    losses = []
    for step in range(6):
        opt.zero_grad(set_to_none=True)
        torch.manual_seed(100)                      # same fine-sampling uniforms every step: a fixed objective
        out = r.render_impl(dict(tq), dict(tr, ray_feats=ray_feats), True)
        loss = ((out['pixel_colors_nr'] - target) ** 2).mean() + ((out['pixel_colors_nr_fine'] - target) ** 2).mean()
        loss.backward()
        assert ray_feats.grad is not None and torch.isfinite(ray_feats.grad).all()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.9 * losses[0], losses


@pytest.mark.parametrize('backend', BACKENDS)
def test_training_with_fine_depth_use_all(backend):
    """fine_depth_use_all (renderer.py:210-213): the fine pass renders the 64 coarse + 64 fine depths merged = 128 samples per
    ray; forward and backward kernels (two samples per lane in the ray kernels) against autograd of the eager port with the
    same uniforms."""
    from neuray_amd.network.renderer import NeuralRayBaseRenderer
    from neuray_amd import synthetic
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'fine_depth_use_all': True,
           'fine_agg_net_cfg': {'sample_num': 128}}
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    r = NeuralRayBaseRenderer(cfg)
    weights = load_weights(False)
    r.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=True)
    r.train()
    if backend == 'emu':
        r._engine_test_lib = emu_lib()
    r = r.to(dev)
    que, ref = synthetic.make_scene(48, 64, 2, seed=8)
    rng = np.random.RandomState(9)
    que['coords'] = (rng.rand(1, 5, 2) * np.array([63, 47])).astype(np.float32)
    tq = {k: torch.from_numpy(v).to(dev) for k, v in que.items()}
    tr = {k: torch.from_numpy(v).to(dev) for k, v in ref.items()}
    tr['ray_feats'].requires_grad_(True)
    lw = torch.from_numpy(rng.randn(1, 5, 3).astype(np.float32))
    torch.manual_seed(77)
    u = torch.rand(1, 5, 64)
    torch.manual_seed(77)
    out = r.render_impl(tq, tr, True)
    assert out['hit_prob_nr_fine'].shape == (1, 5, 128)
    (out['pixel_colors_nr_fine'] * lw.to(dev)).sum().backward()
    w = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in weights.items()}
    cq = {k: torch.from_numpy(v) for k, v in que.items()}
    cr = {k: torch.from_numpy(v.copy()) for k, v in ref.items()}
    cr['ray_feats'].requires_grad_(True)
    ocfg = {'depth_sample_num': 64, 'fine_depth_sample_num': 64, **cfg, 'coarse_use_vis': False, 'fine_use_vis': True}
    want = tep.render_impl(w, ocfg, cq, cr, is_train=True, u=u)
    assert np.abs(out['pixel_colors_nr_fine'].detach().cpu().numpy() - want['pixel_colors_nr_fine'].detach().numpy()).max() <= 5e-4
    (want['pixel_colors_nr_fine'] * lw).sum().backward()
    for k, p_ in r.named_parameters():
        if k.startswith('fine_'):
            g = w[k].grad.numpy() if w[k].grad is not None else np.zeros(tuple(p_.shape), np.float32)
            assert np.abs(p_.grad.cpu().numpy() - g).max() <= 1e-2 * max(1e-3, float(np.abs(g).max())), k
    g = cr['ray_feats'].grad.numpy()
    assert np.abs(tr['ray_feats'].grad.cpu().numpy() - g).max() <= 1e-2 * float(np.abs(g).max())


# ---- a19: the query view's own hit probabilities (renderer.py:137-155), backward on the resident scheme -------------------------
@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('rn,dn,vis_head', [(5, 8, False), (37, 16, True), (16, 5, True), (70, 64, False),
                                            (8300, 3, False)])      # (more tiles than workgroups: the persistent loop)
def test_self_hit_backward_matches_autograd_and_the_first_version(rn, dn, vis_head, backend):
    from neuray_amd import synthetic
    from neuray_amd.engine import RenderEngine
    from oracle import neuray_oracle as orc
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    eng = RenderEngine(dev, _test_lib=emu_lib() if backend == 'emu' else None)
    que, _ = synthetic.make_scene(48, 64, 2, seed=70 + rn)
    rng = np.random.RandomState(71 + rn)
    que['coords'] = (rng.rand(1, rn, 2) * np.array([63, 47])).astype(np.float32)
    que['ray_feats'] = (0.5 * rng.randn(1, 32, 24, 32)).astype(np.float32)          # the query view's own vis-encoder output
    que.setdefault('imgs', np.zeros((1, 3, 48, 64), np.float32))
    weights = load_weights(vis_head)
    depth = orc.sample_depth(que['depth_range'], rn, dn)
    depth = (depth * (1.0 + 0.02 * rng.rand(1, rn, dn))).astype(np.float32); depth.sort(-1)
    lw = rng.randn(rn, dn).astype(np.float32)

    # ---- autograd of the eager port
    w = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in weights.items() if k.startswith('dist_decoder.')}
    tq = {k: torch.from_numpy(v.copy()) for k, v in que.items()}
    tq['ray_feats'].requires_grad_(True)
    hp = tep.self_hit_prob(w, {'coarse_use_vis': vis_head, 'fine_use_vis': True}, torch.from_numpy(depth), tq, False)
    (hp[0] * torch.from_numpy(lw)).sum().backward()

    # ---- ours: gathered feature -> both backward kernels
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    qc = eng.prepare_query({k: t(v) for k, v in que.items()})
    feats = eng.interpolate_feats(t(que['ray_feats']), t(que['coords']), 48, 64, align_corners=False)[0]      # [rn,32]
    flat, has_vis = eng.flat_pass(weights, 'dist_decoder.', 'agg_net.')
    got = {}
    d_feats, d_flat = eng.self_hit_prob_backward(qc, t(depth[0]), feats, flat, has_vis, vis_head, t(lw))
    d_map = eng.interpolate_feats_backward(d_feats[None], tuple(tq['ray_feats'].shape), t(que['coords']), 48, 64, align_corners=False)
    got['auto'] = (eng.unflatten_pass_grads(d_flat, weights, 'dist_decoder.', 'agg_net.'), d_map, d_feats)

    def close(a, b, name, rel):
        a, b = a.detach().cpu().numpy(), (b.detach().cpu().numpy() if torch.is_tensor(b) else b)
        tol = rel * max(1e-3, float(np.abs(b).max()))
        assert a.shape == b.shape and np.abs(a - b).max() <= tol, (name, float(np.abs(a - b).max()), float(np.abs(b).max()))

    for k, p_ in w.items():
        want = p_.grad if p_.grad is not None else torch.zeros_like(p_)
        close(got['auto'][0][k], want, k, 2e-3)
    close(got['auto'][1], tq['ray_feats'].grad, 'que ray_feats', 2e-3)
    for k, g in got['auto'][0].items():                      # nothing but the dist decoder is touched
        if not k.startswith('dist_decoder.'):
            assert float(g.abs().max()) == 0.0, k


@pytest.mark.parametrize('backend', BACKENDS)
def test_resident_point_backward_uses_the_forwards_saved_quantities(backend):
    """the C ABI refuses the resident kernel without NeurayPointsBwdArgs.saved_dev; the engine produces the buffer itself when the
    caller did not keep the forward's, and both routes give the same gradients"""
    import ctypes as C
    from neuray_amd import _lib
    from neuray_amd.engine import RenderEngine
    from oracle import neuray_oracle as orc
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    eng = RenderEngine(dev, _test_lib=emu_lib() if backend == 'emu' else None)
    que, ref, weights, rng = _pass_case(4, 6, 7, False, seed=91)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    views = eng.prepare_views({k: t(v) for k, v in ref.items()})
    qc = eng.prepare_query({k: t(v) for k, v in que.items()})
    depth = t(orc.sample_depth(que['depth_range'], 6, 7)[0])
    coords = t(que['coords'][0])
    packed = eng.pack_pass(weights, 'dist_decoder.', 'agg_net.')
    flat, has_vis = eng.flat_pass(weights, 'dist_decoder.', 'agg_net.')
    fwd = eng.render_pass(qc, views, coords, depth, packed, use_vis=False, save=True)
    assert fwd['saved'].numel() == int(eng.lib.neuray_points_saved_floats(6 * 7)) > 0
    d_rec = t(rng.randn(6, 7, 20).astype(np.float32))
    a = eng.render_points_backward(qc, views, coords, depth, flat, has_vis, False, d_rec, packed=packed, saved=fwd['saved'])
    b = eng.render_points_backward(qc, views, coords, depth, flat, has_vis, False, d_rec, packed=packed)      # engine reruns the forward
    for x, y in zip(a, b):
        assert float((x - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max()))
    # straight through the C ABI with saved_dev = NULL
    pt = eng.pack_pass_t_device(flat, has_vis)
    d_flat, d_rf, d_if = torch.zeros_like(flat), torch.zeros_like(views.ray_feats), torch.zeros_like(views.img_feats)
    args = _lib.NeurayPointsBwdArgs(
        qc.data_ptr(), views.view_const.data_ptr(), coords.data_ptr(), depth.data_ptr(), views.ray_feats.data_ptr(),
        views.img_feats.data_ptr(), views.rgba.data_ptr(), flat.data_ptr(), d_rec.data_ptr(), d_flat.data_ptr(), d_rf.data_ptr(),
        d_if.data_ptr(), views.rfn, 6, 7, views.h, views.w, views.fh, views.fw, int(has_vis), 0, 0.05,
        packed.dev.data_ptr(), pt.data_ptr(), None, None)
    assert eng.lib.neuray_render_points_backward(C.byref(args), eng._stream()) != 0
    assert b'saved_dev' in eng.lib.neuray_last_error()


@pytest.mark.parametrize('backend', BACKENDS)
def test_point_backward_limits_are_refused_cleanly(backend):
    """round 6 removed the first-version point backward (rfn 9..16) and the one-launch form of the resident one: more than 8 views under
    autograd raise, and the C entry point names the buffer it misses"""
    from neuray_amd.engine import RenderEngine
    from neuray_amd import _lib
    import ctypes as C
    from oracle import neuray_oracle as orc
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    eng = RenderEngine(dev, _test_lib=emu_lib() if backend == 'emu' else None)
    que, ref, weights, rng = _pass_case(9, 2, 4, False, seed=3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)         # noqa: E731
    views = eng.prepare_views({k: t(v) for k, v in ref.items()})
    qc = eng.prepare_query({k: t(v) for k, v in que.items()})
    depth = t(orc.sample_depth(que['depth_range'], 2, 4)[0])
    coords = t(que['coords'][0])
    flat, has_vis = eng.flat_pass(weights, 'dist_decoder.', 'agg_net.')
    d_rec = t(rng.randn(2, 4, 20).astype(np.float32))
    with pytest.raises(NotImplementedError, match='8 reference views'):
        eng.render_points_backward(qc, views, coords, depth, flat, has_vis, False, d_rec)
    assert int(eng.lib.neuray_points_backward_handover_floats(5 * 9)) == ((5 * 9 + 15) // 16) * 8 * 20 * 64
    packed = eng.pack_pass(weights, 'dist_decoder.', 'agg_net.')
    pt = eng.pack_pass_t_device(flat, has_vis)
    d_flat, d_rf, d_if = torch.zeros_like(flat), torch.zeros_like(views.ray_feats), torch.zeros_like(views.img_feats)
    saved = eng.points_saved_buffer(8)

    def call(rfn, saved_, ho):
        args = _lib.NeurayPointsBwdArgs(
            qc.data_ptr(), views.view_const.data_ptr(), coords.data_ptr(), depth.data_ptr(), views.ray_feats.data_ptr(),
            views.img_feats.data_ptr(), views.rgba.data_ptr(), flat.data_ptr(), d_rec.data_ptr(), d_flat.data_ptr(), d_rf.data_ptr(),
            d_if.data_ptr(), rfn, 2, 4, views.h, views.w, views.fh, views.fw, int(has_vis), 0, 0.05,
            packed.dev.data_ptr(), pt.data_ptr(), saved_, ho)
        return eng.lib.neuray_render_points_backward(C.byref(args), eng._stream()), eng.lib.neuray_last_error()
    rc, msg = call(9, saved.data_ptr(), None)
    assert rc != 0 and b'at most 8' in msg
    rc, msg = call(8, saved.data_ptr(), None)
    assert rc != 0 and b'handover_dev' in msg
