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
import zlib

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


def scene_case(ns):
    """The scene-level renderers of the reference (network/renderer.py:256-545) on a tiny synthetic scene:
      gen:  NeuralRayGenRenderer.forward(data) in eval with a pass-through init_net (the real ones need depth maps / MVSNet)
      ft:   NeuralRayFtRenderer.train_step / validate_step / render_pose on an instance whose scene attributes are set
            directly (its constructor reads a dataset from disk), `to_cuda` patched to the identity
      helpers: get_coords_mask, compute_nearest_camera_indices, select_working_views, pad_imgs_info
    Weights: fill_by_name (reproducible from the parameter names)."""
    import importlib
    import torch.nn as nn
    from neuray_amd import synthetic
    R = ns.renderer
    out = {}
    # ---- helpers
    vs, bu, ii = importlib.import_module('utils.view_select'), importlib.import_module('utils.base_utils'), importlib.import_module('utils.imgs_info')
    rng = np.random.RandomState(21)
    poses = np.stack([synthetic.look_at_pose(synthetic.sphere_pos(4.0, a, e)) for a, e in rng.rand(7, 2) * np.array([360, 60])]).astype(np.float32)
    qposes = np.stack([synthetic.look_at_pose(synthetic.sphere_pos(4.0, a, e)) for a, e in rng.rand(3, 2) * np.array([360, 60])]).astype(np.float32)

    class DB:
        def get_pose(self, i):
            return (poses if i >= 0 else qposes)[i if i >= 0 else -i - 1]
    out['h_poses'], out['h_qposes'] = poses, qposes
    out['h_nearest_self'] = vs.compute_nearest_camera_indices(DB(), list(range(7)))
    out['h_nearest_que'] = vs.compute_nearest_camera_indices(DB(), [-1, -2, -3], list(range(7)))
    out['h_working'] = vs.select_working_views(poses, qposes, 4, True)
    mask = rng.rand(20, 30) > 0.7
    out['h_mask'] = mask
    for tag, (num, ratio) in {'a': (64, 0.5), 'b': (16, 1.0), 'c': (400, 0.5)}.items():
        np.random.seed(5)
        out['h_coords_' + tag] = bu.get_coords_mask(mask, num, ratio)
    info = {'imgs': rng.rand(2, 3, 21, 35).astype(np.float32), 'depth': rng.rand(2, 1, 21, 35).astype(np.float32),
            'masks': (rng.rand(2, 1, 21, 35) > 0.5).astype(np.float32)}
    out['h_pad_in_imgs'], out['h_pad_in_depth'], out['h_pad_in_masks'] = info['imgs'], info['depth'], info['masks']
    padded = ii.pad_imgs_info(dict(info), 16)
    out['h_pad_imgs'], out['h_pad_depth'], out['h_pad_masks'] = padded['imgs'], padded['depth'], padded['masks']

    # ---- gen
    class PassThrough(nn.Module):
        def __init__(self, cfg):
            super().__init__()

        def forward(self, ref_imgs_info, src_imgs_info, is_train):
            return ref_imgs_info['ray_feats']
    R.name2init_net['passthrough'] = PassThrough
    small = {'use_hierarchical_sampling': True, 'depth_sample_num': 8, 'fine_depth_sample_num': 8,
             'agg_net_cfg': {'sample_num': 8}, 'fine_agg_net_cfg': {'sample_num': 8}, 'ray_batch_num': 16}
    gen = R.NeuralRayGenRenderer({**small, 'init_net_type': 'passthrough', 'depth_loss_coords_num': 40}).eval()
    fill_by_name(gen)
    que, ref = synthetic.make_scene(48, 64, 3, seed=5)
    que['coords'] = (np.random.RandomState(6).rand(1, 23, 2) * np.array([63, 47])).astype(np.float32)
    ref.pop('img_feats')
    torch.manual_seed(11)
    with torch.no_grad():
        g = gen({'que_imgs_info': to_t(que), 'ref_imgs_info': to_t(ref), 'eval': True})
    for k, v in g.items():
        out['gen_' + k] = v.numpy()

    # ---- ft
    h, w, n = 48, 64, 6
    _, sref = synthetic.make_scene(h, w, n, seed=9)
    sref.pop('img_feats')
    init = sref.pop('ray_feats')
    srng = np.random.RandomState(10)
    sref['masks'] = (srng.rand(n, 1, h, w) > 0.6).astype(np.float32)
    sref['depth'] = (2 + 4 * srng.rand(n, 1, h, w)).astype(np.float32)
    sval = {'imgs': srng.rand(2, 3, h, w).astype(np.float32), 'masks': np.ones((2, 1, h, w), np.float32),
            'poses': np.stack([synthetic.look_at_pose(synthetic.sphere_pos(4.03, 33.0, 22.0)),
                               synthetic.look_at_pose(synthetic.sphere_pos(4.03, 20.0, 30.0))]).astype(np.float32),
            'Ks': sref['Ks'][:2].copy(), 'depth_range': sref['depth_range'][:2].copy()}
    ft_cfg = {**small, 'use_self_hit_prob': True, 'neighbor_view_num': 3, 'neighbor_pool_ratio': 1, 'train_ray_num': 12,
              'foreground_ratio': 0.5, 'include_self_prob': 0.01, 'use_validation': True}
    ft = R.NeuralRayFtRenderer.__new__(R.NeuralRayFtRenderer)
    R.NeuralRayBaseRenderer.__init__(ft, {**R.NeuralRayFtRenderer.default_cfg, **ft_cfg})
    fill_by_name(ft)
    R.to_cuda = lambda d: d
    ft.ref_ids = np.arange(n)
    ft.ref_imgs_info, ft.val_imgs_info = to_t(sref), to_t(sval)
    cen = lambda P: np.asarray([-p[:, :3].T @ p[:, 3] for p in P])
    ft.ref_dist_idx = np.argsort(np.linalg.norm(cen(sref['poses'])[None] - cen(sref['poses'])[:, None], 2, 2), 1)
    ft.val_dist_idx = np.argsort(np.linalg.norm(cen(sref['poses'])[None] - cen(sval['poses'])[:, None], 2, 2), 1)
    ft.ray_feats = nn.ParameterList([nn.Parameter(torch.from_numpy(init[i:i + 1].copy())) for i in range(n)])
    for k, v in {'ref': sref, 'val': sval}.items():
        for kk, vv in v.items():
            out['ft_%s_%s' % (k, kk)] = vv
    out['ft_init_ray_feats'] = init
    ft.eval()
    v = ft.validate_step(1)
    out['ft_val_pixel_colors_nr_fine'] = v['pixel_colors_nr_fine'].numpy()
    out['ft_val_ray_mask_fine'] = v['ray_mask_fine'].numpy()
    pose_info = {'poses': torch.from_numpy(sval['poses'][:1]), 'Ks': torch.from_numpy(sval['Ks'][:1]),
                 'depth_range': torch.from_numpy(sval['depth_range'][:1]),
                 'coords': torch.from_numpy((np.random.RandomState(12).rand(1, 19, 2) * np.array([63, 47])).astype(np.float32))}
    out['ft_pose_coords'] = pose_info['coords'].numpy()
    out['ft_pose_pixel_colors_nr_fine'] = ft.render_pose(pose_info)['pixel_colors_nr_fine'].numpy()
    ft.train()
    np.random.seed(3)
    torch.manual_seed(4)
    t = ft.train_step()
    out['ft_train_coords'] = t['que_imgs_info']['coords'].numpy()
    for k in ('pixel_colors_nr', 'pixel_colors_nr_fine', 'hit_prob_self', 'hit_prob_self_fine', 'pixel_colors_gt'):
        out['ft_train_' + k] = t[k].detach().numpy()
    loss = ((t['pixel_colors_nr_fine'] - t['pixel_colors_gt']) ** 2).mean() + t['hit_prob_self_fine'].mean()
    loss.backward()
    out['ft_train_touched'] = np.array([i for i in range(n) if ft.ray_feats[i].grad is not None and float(ft.ray_feats[i].grad.abs().max()) > 0])
    for i in out['ft_train_touched'][:2]:
        out['ft_train_grad_%d' % i] = ft.ray_feats[int(i)].grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'case_scene.npz'), **out)
    print('wrote case_scene.npz', out['ft_train_touched'], out['gen_depth_mean'].shape)


def init_net_case(ns):
    """extract_depth_for_init, get_diff_feats and the whole DepthInitNet (network/init_net.py:13-112)
    of the reference on a small scene with depth maps (smooth surface + noise + a few out-of-range values for the clamps)."""
    import importlib
    from neuray_amd import synthetic
    init_net = importlib.import_module('network.init_net')
    h, w, n = 48, 64, 3
    _, ref = synthetic.make_scene(h, w, n, seed=13)
    rng = np.random.RandomState(14)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    depth = np.stack([3.6 + 0.7 * np.sin(xx / 9.0 + v) * np.cos(yy / 7.0 - v) + 0.05 * rng.randn(h, w) for v in range(n)])[:, None]
    depth[0, 0, :2, :5] = 0.0
    depth[1, 0, 5, 5:9] = 9.0
    depth[2, 0, -1, -4:] = 1.0
    depth = depth.astype(np.float32)
    info = {k: torch.from_numpy(ref[k]) for k in ('imgs', 'poses', 'Ks', 'depth_range')}
    info['depth'] = torch.from_numpy(depth)
    net = init_net.DepthInitNet({}).eval()
    fill_by_name(net)
    with torch.no_grad():
        dn = init_net.extract_depth_for_init(info)
        diff = init_net.get_diff_feats(info, dn)
        out = net(info, None, False)
    np.savez_compressed(os.path.join(HERE, 'case_init_depth.npz'), imgs=ref['imgs'], poses=ref['poses'], Ks=ref['Ks'],
                        depth_range=ref['depth_range'], depth=depth, depth_norm=dn.numpy(), diff_feats=diff.numpy(),
                        ray_feats=out.numpy())
    import json
    json.dump({k: list(v.shape) for k, v in net.state_dict().items()},
              open(os.path.join(HERE, 'ref_depth_init_net_state_dict.json'), 'w'), indent=0, sort_keys=True)
    print('wrote case_init_depth.npz', diff.shape, out.shape, float(diff.abs().max()))


def fill_buffers_by_name(module):
    """running statistics of the (frozen, eval-mode) MVSNet batch norms, seeded by name"""
    with torch.no_grad():
        for name, buf in module.named_buffers():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            if name.endswith('running_mean'):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
            elif name.endswith('running_var'):
                buf.copy_(torch.rand(buf.shape, generator=g) * 0.5 + 0.75)


def cost_volume_case(ns):
    """the cost-volume init net (network/init_net.py:113-160,204-258; network/mvsnet/*) on a tiny
    scene, training path (no resize, no torch.cuda calls), MVSNet weights from fill_by_name instead of the pretrained
    mvsnet_pl.ckpt.  inplace_abn.ABN and kornia.create_meshgrid are the restatements of tests/golden/ref_harness.py."""
    import importlib
    import json
    from neuray_amd import synthetic
    init_net = importlib.import_module('network.init_net')
    mods = importlib.import_module('network.mvsnet.modules')
    h, w, rfn, sn, dn = 64, 64, 2, 3, 8
    _, views = synthetic.make_scene(h, w, rfn + sn, seed=17, depth_range=(2.5, 5.5))
    pick = lambda lo, hi: {k: torch.from_numpy(views[k][lo:hi].copy()) for k in ('imgs', 'poses', 'Ks', 'depth_range')}
    ref, src = pick(0, rfn), pick(rfn, rfn + sn)
    ref['nn_ids'] = torch.tensor([[0, 2], [1, 0]])
    real_cuda, real_load = torch.Tensor.cuda, init_net.load_ckpt
    torch.Tensor.cuda = lambda self, *a, **k: self
    init_net.load_ckpt = lambda *a, **k: None
    try:
        net = init_net.CostVolumeInitNet({'cost_volume_sn': dn})
    finally:
        torch.Tensor.cuda, init_net.load_ckpt = real_cuda, real_load
    fill_by_name(net)
    fill_buffers_by_name(net.mvsnet)
    net.eval()
    out = {}
    with torch.no_grad():
        depth_vals = init_net.get_depth_vals(ref['depth_range'], dn)
        ref_prj = init_net.construct_project_matrix(0.25, 0.25, ref['Ks'], ref['poses'])
        src_prj = init_net.construct_project_matrix(0.25, 0.25, src['Ks'], src['poses'])
        rn_ = (ref['imgs'] - net.imagenet_mean) / net.imagenet_std
        sn_ = (src['imgs'] - net.imagenet_mean) / net.imagenet_std
        rf, sf = net.mvsnet.feature(rn_), net.mvsnet.feature(sn_)
        warped = mods.homo_warp(sf[ref['nn_ids'][:, 0]], src_prj[ref['nn_ids'][:, 0]], torch.inverse(ref_prj), depth_vals)
        cost_reg, depth = init_net.construct_cost_volume_with_src(ref, src, net.mvsnet, dn, net.imagenet_mean, net.imagenet_std, True)
        feats = net(ref, src, True)
    out.update(ref_feats=rf.numpy(), src_feats=sf.numpy(), depth_vals=depth_vals.numpy(), ref_prj=ref_prj.numpy(), src_prj=src_prj.numpy(),
               warped0=warped.numpy(), cost_reg=cost_reg.numpy(), depth=depth.numpy(), ray_feats=feats.numpy(), nn_ids=ref['nn_ids'].numpy())
    for tag, d in (('ref', ref), ('src', src)):
        for k in ('imgs', 'poses', 'Ks', 'depth_range'):
            out['%s_%s' % (tag, k)] = d[k].numpy()
    np.savez_compressed(os.path.join(HERE, 'case_cost_volume.npz'), **out)
    json.dump({k: list(v.shape) for k, v in net.state_dict().items()},
              open(os.path.join(HERE, 'ref_cost_volume_init_net_state_dict.json'), 'w'), indent=0, sort_keys=True)
    print('wrote case_cost_volume.npz', cost_reg.shape, depth.shape, feats.shape)


def pipeline_case(ns):
    """build_imgs_info / build_render_imgs_info / select_working_views_db / colour mapping of the
    reference (utils/imgs_info.py, utils/view_select.py, utils/base_utils.py) on an in-memory database."""
    import importlib
    from neuray_amd import synthetic
    ii, vs, bu = importlib.import_module('utils.imgs_info'), importlib.import_module('utils.view_select'), importlib.import_module('utils.base_utils')
    db = synthetic.MemoryDatabase(7, 37, 53, seed=31)
    out = {}
    ids = [4, 0, 6]
    a = ii.build_imgs_info(db, ids, 16, True, False, True, True)
    for k, v in a.items():
        out['aligned_' + k] = v
    b = ii.build_imgs_info(db, ids, -1, True, True, False)
    for k, v in b.items():
        out['nodepth_' + k] = v
    ragged = synthetic.MemoryDatabase(3, 30, 41, seed=32, ragged=True)
    c = ii.build_imgs_info(ragged, [0, 1, 2], -1, False)
    for k, v in c.items():
        out['ragged_' + k] = v
    r = ii.build_render_imgs_info(db.get_pose(2), db.get_K(2), (37, 53), (2.0, 6.0))
    for k in ('poses', 'Ks', 'coords', 'depth_range'):
        out['render_' + k] = r[k]
    qp = np.stack([db.get_pose(1), db.get_pose(5)])
    out['working'] = vs.select_working_views_db(db, None, qp, 3, False)
    out['working_excl'] = vs.select_working_views_db(db, [6, 5, 4, 3, 2], qp, 2, True)
    x = np.random.RandomState(33).rand(5, 7, 3).astype(np.float32) * 1.2 - 0.1
    out['cmap_in'], out['cmap_back'] = x, bu.color_map_backward(x)
    np.savez_compressed(os.path.join(HERE, 'case_pipeline.npz'), **out)
    print('wrote case_pipeline.npz', a['imgs'].shape, c['imgs'].shape)


def multi_query_case(ns):
    """qn = 2 query views in one render_impl call (every shipped caller uses qn = 1; the tensors carry the dimension):
    second view with its own pose / intrinsics / depth range - the fine sampling normalises every view with view 0's
    range (quirk A.9.6, render_ops.py:183,225).  Weights: weights_seed0.npz (the default-cfg renderer under seed 0)."""
    from neuray_amd import synthetic
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': 8, 'fine_depth_sample_num': 8,
           'agg_net_cfg': {'sample_num': 8}, 'fine_agg_net_cfg': {'sample_num': 8}, 'use_self_hit_prob': True, 'render_depth': True}
    que, ref = orc.make_scene(40, 56, 3, seed=21, que_imgs=True)
    pose2 = synthetic.look_at_pose(synthetic.sphere_pos(4.03, 41.0, 17.0)).astype(np.float32)
    rng = np.random.RandomState(22)
    q2 = {'poses': np.stack([que['poses'][0], pose2]), 'Ks': np.stack([que['Ks'][0], que['Ks'][0] * np.array([[1.1], [1.05], [1.0]], np.float32)]),
          'depth_range': np.array([[2.0, 6.0], [2.5, 5.0]], np.float32),
          'coords': (rng.rand(2, 19, 2) * np.array([55, 39])).astype(np.float32),
          'imgs': rng.rand(2, 3, 40, 56).astype(np.float32), 'ray_feats': rng.randn(2, 32, 10, 14).astype(np.float32)}
    renderer = build_renderer(ns, cfg, seed=0)
    with torch.no_grad():
        out = renderer.render_impl(to_t(q2), to_t(ref), False)
    save = {'cfg_json': np.array(repr(cfg))}
    for k, v in q2.items():
        save['que.' + k] = v
    for k, v in ref.items():
        save['ref.' + k] = v
    for k, v in out.items():
        save['out.' + k] = v.numpy()
    for k, v in hot_weights(renderer).items():
        save['w.' + k] = v
    np.savez_compressed(os.path.join(HERE, 'case_h_two_queries.npz'), **save)
    print('wrote case_h_two_queries.npz', {k: tuple(v.shape) for k, v in out.items()})


def sampling_branches_case(ns):
    """the two branches of network/render_ops.py the renderer never takes: sample_depth(random_sample=True) (:160-161) and
    sample_fine_depth(inv_mode=False) (:181-186,224-228), for the render_ops module surface"""
    ro = ns.render_ops
    rng = np.random.RandomState(41)
    dr = torch.tensor([[2.0, 6.0], [1.5, 9.0]])
    coords = torch.zeros(2, 11, 2)
    torch.manual_seed(77)
    u = torch.rand(2, 11, 14)
    torch.manual_seed(77)
    depth, dists = ro.sample_depth(dr, coords, 16, True)
    hit = torch.from_numpy((rng.rand(2, 11, 16) ** 3).astype(np.float32))
    sorted_depth = torch.sort(depth, -1)[0]
    fine_lin = ro.sample_fine_depth(sorted_depth, hit, dr, 12, False, inv_mode=False)
    torch.manual_seed(78)
    uf = torch.rand(2, 11, 12)
    torch.manual_seed(78)
    fine_lin_rand = ro.sample_fine_depth(sorted_depth, hit, dr, 12, True, inv_mode=False)
    np.savez_compressed(os.path.join(HERE, 'case_sampling_branches.npz'), depth_range=dr.numpy(), u=u.numpy(), depth=depth.numpy(),
                        dists=dists.numpy(), hit=hit.numpy(), sorted_depth=sorted_depth.numpy(), fine_lin=fine_lin.numpy(),
                        uf=uf.numpy(), fine_lin_rand=fine_lin_rand.numpy())
    print('wrote case_sampling_branches.npz', depth.shape, fine_lin.shape)


def direct_rendering_cases(ns):
    """cfg['use_dr_prediction'] (renderer.py:85-125 + sph_solver.py): the `*_dr` outputs of both passes, with the spherical-
    harmonics colours (f_dr) and with use_nr_color_for_dr (f_dr_nr).  Geometry of case C (a camera with samples behind it, one far
    away, wide depth range: masked views and empty rays exercise the `ground` / `insufficient` branches).  Each case is also
    run through the reference in float64 (`out64.*`): the 16 x 16 torch.inverse of a regularised rank-<= rfn normal matrix is
    ill conditioned, so |ref32 - ref64| is the honest tolerance of the SH colours."""
    def tweak(que, ref):
        ref['poses'][1] = orc.look_at_pose(orc.sphere_pos(2.5, 30.0, 25.0), target=orc.sphere_pos(8.0, 30.0, 25.0))
        ref['poses'][2] = orc.look_at_pose(orc.sphere_pos(9.0, 200.0, -40.0))
        ref['depth_range'][2] = np.array([5.0, 13.0], np.float32)
    base = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': 16,
            'fine_depth_sample_num': 16, 'agg_net_cfg': {'sample_num': 16}, 'fine_agg_net_cfg': {'sample_num': 16},
            'use_dr_prediction': True}
    for name, extra in (('f_dr', {}), ('f_dr_nr', {'use_nr_color_for_dr': True})):
        cfg = {**base, **extra}
        r = run_case(ns, name, cfg, 48, 48, 5, 48, seed=6, is_train=False, integer_coords=False, depth_range=(0.8, 9.0), tweak=tweak)
        z = dict(np.load(os.path.join(HERE, 'case_%s.npz' % name)))
        r64 = r.double()
        tq = {k[4:]: torch.from_numpy(v).double() for k, v in z.items() if k.startswith('que.') and k != 'que.Ks_inv'}
        tr = {k[4:]: torch.from_numpy(v).double() for k, v in z.items() if k.startswith('ref.')}
        with torch.no_grad():
            out64 = r64.render_impl(tq, tr, False)
        for k in ('pixel_colors_dr', 'hit_prob_dr', 'pixel_colors_dr_fine', 'hit_prob_dr_fine', 'pixel_colors_nr'):
            z['out64.' + k] = out64[k].numpy()
        np.savez_compressed(os.path.join(HERE, 'case_%s.npz' % name), **z)
        print('   ref32 vs ref64: pixel_colors_dr %.2e, hit_prob_dr %.2e' % (
            np.abs(z['out.pixel_colors_dr'] - z['out64.pixel_colors_dr']).max(), np.abs(z['out.hit_prob_dr'] - z['out64.hit_prob_dr']).max()))


if __name__ == '__main__':
    if sys.argv[1:] == ['dr']:
        direct_rendering_cases(ref_harness.import_reference())
        sys.exit(0)
    main()
    ns_ = ref_harness.import_reference()
    direct_rendering_cases(ns_)
    gradient_case(ns_)
    encoder_case(ns_)
    scene_case(ns_)
    init_net_case(ns_)
    cost_volume_case(ns_)
    pipeline_case(ns_)
    multi_query_case(ns_)
    sampling_branches_case(ns_)
