"""Generate golden vectors by running the REFERENCE NeuRay modules on CPU.

Run in the build container only (needs /root/reference, read-only):

    python tests/golden/make_golden.py

Writes tests/golden/weights_seed0.npz and tests/golden/case_*.npz.  The reference's
own tests hold no golden vectors for the render path, so these are
produced from the reference itself: seeded random weights built by the reference
constructors (so they carry its kaiming init), the seeded synthetic scene of
oracle/neuray_oracle.make_scene, and NeuralRayBaseRenderer.render_impl plus the
step-by-step functions render_by_depth calls (network/renderer.py:168-203).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_harness  # noqa: E402
from oracle import neuray_oracle as orc  # noqa: E402


def to_t(d):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}


def build_renderer(ns, cfg, seed=0):
    torch.manual_seed(seed)
    r = ns.renderer.NeuralRayBaseRenderer(cfg)
    r.eval()
    return r


def hot_weights(renderer):
    sd = renderer.state_dict()
    keep = ('dist_decoder.', 'agg_net.', 'fine_dist_decoder.', 'fine_agg_net.')
    return {k: v.detach().numpy().copy() for k, v in sd.items() if k.startswith(keep)}


def coords_for_case(rng, h, w, rn, integer):
    if integer:
        xs = rng.randint(0, w, size=rn)
        ys = rng.randint(0, h, size=rn)
        return np.stack([xs, ys], -1)[None].astype(np.float32)
    return (rng.rand(1, rn, 2) * np.array([w - 1, h - 1])).astype(np.float32)


def run_case(ns, name, cfg, h, w, rfn, rn, seed, is_train, integer_coords, depth_range=(2.0, 6.0),
             save_intermediates=False, tweak=None, que_imgs=False):
    que, ref = orc.make_scene(h, w, rfn, seed=seed, depth_range=depth_range, que_imgs=que_imgs)
    if tweak is not None:
        tweak(que, ref)
    rng = np.random.RandomState(seed + 1000)
    que['coords'] = coords_for_case(rng, h, w, rn, integer_coords)
    renderer = build_renderer(ns, cfg, seed=0)
    # Ks_inv exactly as the reference computes it (torch.inverse, render_ops.py:20)
    que['Ks_inv'] = torch.inverse(torch.from_numpy(que['Ks'])).numpy()

    tq, tr = to_t(que), to_t(ref)
    tq.pop('Ks_inv')
    captured = {}
    real_rand = torch.rand

    def rand_capture(*a, **k):
        out = real_rand(*a, **k)
        captured.setdefault('u', out.clone())
        return out

    torch.rand = rand_capture
    try:
        torch.manual_seed(1234)
        with torch.no_grad():
            out = renderer.render_impl(tq, tr, is_train)
    finally:
        torch.rand = real_rand

    save = {'cfg_json': np.array(repr(cfg)), 'h': h, 'w': w, 'rfn': rfn, 'rn': rn, 'seed': seed,
            'is_train': int(is_train)}
    for k, v in que.items():
        save['que.' + k] = v
    for k, v in ref.items():
        save['ref.' + k] = v
    for k, v in out.items():
        save['out.' + k] = v.numpy()
    if 'u' in captured:
        save['u'] = captured['u'].numpy()

    if save_intermediates:
        ro = ns.render_ops
        with torch.no_grad():
            que_depth, _ = ro.sample_depth(tq['depth_range'], tq['coords'], renderer.cfg['depth_sample_num'], False)
            que_dists = ro.depth2inv_dists(que_depth, tq['depth_range'])
            que_pts, que_dir = ro.depth2points(tq, que_depth)
            prj = ro.project_points_dict(tr, que_pts)
            prj = renderer.predict_proj_ray_prob(prj, tr, que_dists, False)
            prj = renderer.get_img_feats(tr, prj)
            mean, var, vis, aw = renderer.dist_decoder(prj['ray_feats'])
            density, colors = renderer.agg_net(prj, que_dir)
        save.update({'mid.que_depth': que_depth.numpy(), 'mid.que_dists': que_dists.numpy(),
                     'mid.que_pts': que_pts.numpy(), 'mid.que_dir': que_dir.numpy(),
                     'mid.density': density.numpy(), 'mid.colors': colors.numpy(),
                     'mid.mean': mean.numpy(), 'mid.var': var.numpy(), 'mid.aw': aw.numpy()})
        if vis is not None:
            save['mid.vis_dec'] = vis.numpy()
        for k, v in prj.items():
            save['mid.prj.' + k] = v.numpy()
    path = os.path.join(HERE, 'case_%s.npz' % name)
    np.savez_compressed(path, **save)
    print('wrote', path, {k: tuple(v.shape) for k, v in out.items()})
    return renderer


def main():
    ns = ref_harness.import_reference()
    torch.set_num_threads(4)

    base = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}}

    # A: small, with every intermediate; fractional coords; 3 views, 16+16 samples
    cfg_a = {**base, 'depth_sample_num': 16, 'fine_depth_sample_num': 16,
             'agg_net_cfg': {'sample_num': 16}, 'fine_agg_net_cfg': {'sample_num': 16}, 'render_depth': True}
    r = run_case(ns, 'a_small', cfg_a, 48, 48, 3, 40, seed=0, is_train=False, integer_coords=False,
                 save_intermediates=True)
    # weights are identical for every case (same constructors, same seed, sample_num does not
    # change parameters); the vis-decoder variant below needs its own file.
    np.savez_compressed(os.path.join(HERE, 'weights_seed0.npz'), **hot_weights(r))

    # B: reference defaults 64+64, 8 views, non-square image, integer pixel coords (gen config 2 shape)
    cfg_b = {**base}
    run_case(ns, 'b_default', cfg_b, 48, 64, 8, 24, seed=1, is_train=False, integer_coords=True)

    # C: adversarial geometry - wide depth range so many samples leave every frustum, one reference
    # camera placed so that samples fall behind it (quirk A.9.1: no z>0 test), one far away.
    def tweak_c(que, ref):
        ref['poses'][1] = orc.look_at_pose(orc.sphere_pos(2.5, 30.0, 25.0), target=orc.sphere_pos(8.0, 30.0, 25.0))
        ref['poses'][2] = orc.look_at_pose(orc.sphere_pos(9.0, 200.0, -40.0))
        ref['depth_range'][2] = np.array([5.0, 13.0], np.float32)
    cfg_c = {**base, 'depth_sample_num': 32, 'fine_depth_sample_num': 32,
             'agg_net_cfg': {'sample_num': 32}, 'fine_agg_net_cfg': {'sample_num': 32}}
    run_case(ns, 'c_adversarial', cfg_c, 48, 48, 4, 32, seed=2, is_train=False, integer_coords=False,
             depth_range=(0.8, 9.0), save_intermediates=True, tweak=tweak_c)

    # D: training mode (random u drawn by the reference on CPU), self hit prob, coarse decoder WITH vis
    cfg_d = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': True}, 'use_self_hit_prob': True,
             'depth_sample_num': 8, 'fine_depth_sample_num': 32, 'render_depth': True,
             'agg_net_cfg': {'sample_num': 8}, 'fine_agg_net_cfg': {'sample_num': 32}}
    r = run_case(ns, 'd_train_vis', cfg_d, 48, 48, 5, 16, seed=3, is_train=True, integer_coords=True,
                 que_imgs=True)
    np.savez_compressed(os.path.join(HERE, 'weights_seed0_vis.npz'), **hot_weights(r))

    # E: fine_depth_use_all (64 coarse + 64 fine = 128 samples in the fine pass)
    cfg_e = {**base, 'depth_sample_num': 64, 'fine_depth_sample_num': 64, 'fine_depth_use_all': True,
             'fine_agg_net_cfg': {'sample_num': 128}}
    run_case(ns, 'e_use_all', cfg_e, 48, 48, 2, 8, seed=4, is_train=False, integer_coords=True)


def gradient_case(ns):
    """Reference autograd through render_impl (training mode): d(loss)/d(weights, ray_feats, img_feats) for
    loss = sum(w_c * pixel_colors_nr) + sum(w_f * pixel_colors_nr_fine) + sum(w_s * hit_prob_self[_fine]).
    Pins the backward kernels (next round) and the autograd of oracle/torch_eager_port.py."""
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'use_self_hit_prob': True,
           'depth_sample_num': 16, 'fine_depth_sample_num': 16, 'agg_net_cfg': {'sample_num': 16},
           'fine_agg_net_cfg': {'sample_num': 16}}
    que, ref = orc.make_scene(48, 48, 3, seed=5, que_imgs=True)
    rng = np.random.RandomState(55)
    que['coords'] = coords_for_case(rng, 48, 48, 24, False)
    que['Ks_inv'] = torch.inverse(torch.from_numpy(que['Ks'])).numpy()
    renderer = build_renderer(ns, cfg, seed=0)
    renderer.train()
    tq, tr = to_t(que), to_t(ref)
    tq.pop('Ks_inv')
    for k in ('ray_feats', 'img_feats'):
        tr[k].requires_grad_(True)
    tq['ray_feats'].requires_grad_(True)
    lw = {k: torch.from_numpy(rng.randn(*shape).astype(np.float32)) for k, shape in
          (('pixel_colors_nr', (1, 24, 3)), ('pixel_colors_nr_fine', (1, 24, 3)), ('hit_prob_self', (1, 24, 16)),
           ('hit_prob_self_fine', (1, 24, 16)))}
    captured = {}
    real_rand = torch.rand

    def rand_capture(*a, **k):
        out = real_rand(*a, **k)
        captured.setdefault('u', out.clone())
        return out

    torch.rand = rand_capture
    try:
        torch.manual_seed(4321)
        out = renderer.render_impl(tq, tr, True)
    finally:
        torch.rand = real_rand
    loss = sum((lw[k] * out[k]).sum() for k in lw)
    loss.backward()
    save = {'cfg_json': np.array(repr(cfg)), 'u': captured['u'].numpy(), 'loss': loss.detach().numpy()}
    for k, v in que.items():
        save['que.' + k] = v
    for k, v in ref.items():
        save['ref.' + k] = v
    for k, v in lw.items():
        save['lw.' + k] = v.numpy()
    for k, v in out.items():
        save['out.' + k] = v.detach().numpy()
    keep = ('dist_decoder.', 'agg_net.', 'fine_dist_decoder.', 'fine_agg_net.')
    for k, p_ in renderer.named_parameters():
        if k.startswith(keep):
            save['grad.' + k] = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).numpy()
    save['grad.ref.ray_feats'] = tr['ray_feats'].grad.numpy()
    save['grad.ref.img_feats'] = tr['img_feats'].grad.numpy()
    save['grad.que.ray_feats'] = tq['ray_feats'].grad.numpy()
    path = os.path.join(HERE, 'case_g_grads.npz')
    np.savez_compressed(path, **save)
    print('wrote', path, 'loss', float(loss))


def fill_by_name(module, scale=0.25):
    """Deterministic weights that do not depend on the order in which a module creates its parameters: every tensor is
    drawn from a generator seeded by the CRC of its name (norm scales around 1)."""
    import zlib
    with torch.no_grad():
        for name, prm in module.named_parameters():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            v = torch.randn(prm.shape, generator=g) * scale
            if name.endswith('.weight') and prm.dim() == 1:     # InstanceNorm scale
                v = v * 0.4 + 1.0
            prm.copy_(v)


def encoder_case(ns):
    """image_encoder (ResUNetLight(3,[1,2,6,4],32,inplanes=16), renderer.py:58) and vis_encoder of the reference on a
    small odd-sized batch, weights from fill_by_name; also the reference base renderer's state_dict surface."""
    import importlib
    import json
    ops, ve = importlib.import_module('network.ops'), importlib.import_module('network.vis_encoder')
    enc = ops.ResUNetLight(3, [1, 2, 6, 4], 32, inplanes=16).eval()
    vis = ve.DefaultVisEncoder({}).eval()
    fill_by_name(enc)
    fill_by_name(vis)
    rng = np.random.RandomState(77)
    imgs = rng.rand(2, 3, 52, 70).astype(np.float32)
    with torch.no_grad():
        feats = enc(torch.from_numpy(imgs))            # odd sizes: the decoder's upsampled size (16 x 20), not h/4 x w/4
        ray_in = rng.randn(*feats.shape).astype(np.float32)
        out = vis(torch.from_numpy(ray_in), feats)
    np.savez_compressed(os.path.join(HERE, 'case_enc.npz'), imgs=imgs, ray_in=ray_in, img_feats=feats.numpy(), ray_feats=out.numpy())
    full = ns.renderer.NeuralRayBaseRenderer({'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': True}})
    json.dump({k: list(v.shape) for k, v in full.state_dict().items()},
              open(os.path.join(HERE, 'ref_base_renderer_state_dict.json'), 'w'), indent=0, sort_keys=True)
    print('wrote case_enc.npz', feats.shape, out.shape)


if __name__ == '__main__':
    main()
    ns_ = ref_harness.import_reference()
    gradient_case(ns_)
    encoder_case(ns_)
