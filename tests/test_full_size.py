"""GPU-only checks at BASELINE.json's full sizes (the bench workload: 800 x 800 image, 8 reference views, 200 x 200 x 32
maps, 32768-ray batches, 64 coarse + 32 / 64 fine samples) through size-independent properties, plus an oracle
comparison on a 256-ray sample of a batch (training path) and on 8 192 strided rays of the image (coarse pass)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def case():
    import bench
    dev = torch.device('cuda', 0)
    cfg, renderer, weights, que, ref, tq, tr = bench.build_case(dev, 32, seed=0)
    renderer.cfg['render_depth'] = True
    return cfg, renderer, weights, que, ref, tq, tr


def render(renderer, tq, tr, coords, is_train=True):
    q = {k: v for k, v in tq.items() if not k.startswith('_')}
    q['coords'] = coords
    with torch.no_grad():
        return renderer.render(q, {k: v for k, v in tr.items() if not k.startswith('_')}, is_train)


def test_full_batch_invariants(case):
    cfg, renderer, weights, que, ref, tq, tr = case
    coords = tq['coords'][:, 300 * 800:300 * 800 + 32768]            # one full ray batch from the middle of the image
    full = render(renderer, tq, tr, coords)
    assert full['pixel_colors_nr_fine'].shape == (1, 32768, 3) and full['hit_prob_nr'].shape == (1, 32768, 64)
    for k in ('hit_prob_nr', 'hit_prob_nr_fine'):
        hp = full[k]
        assert torch.isfinite(hp).all() and float(hp.min()) >= 0.0 and float(hp.sum(-1).max()) <= 1 + 1e-5
    for k in ('pixel_colors_nr', 'pixel_colors_nr_fine'):          # convex combinations of colours in [0,1], times sum(hit) <= 1
        assert float(full[k].min()) >= 0.0 and float(full[k].max()) <= 1 + 1e-5
    near, far = float(que['depth_range'][0, 0]), float(que['depth_range'][0, 1])
    d = full['render_depth_fine']
    assert float(d.min()) >= 0.0 and float(d.max()) <= far * (1 + 1e-5)
    # batching invariance at full size (eval mode: the training path draws its fine-sampling uniforms per call):
    # the same rays in 8 batches of 4096, and a permutation of them, bit for bit
    full = render(renderer, tq, tr, coords, is_train=False)
    renderer.cfg['ray_batch_num'], renderer.cfg['hip_min_ray_batch'] = 4096, 0        # (exactly 4096 rays per launch: render() merges batches otherwise)
    try:
        parts = render(renderer, tq, tr, coords, is_train=False)
    finally:
        renderer.cfg['ray_batch_num'] = 32768
    for k in full:
        assert torch.equal(parts[k], full[k]), k
    perm = torch.randperm(32768, generator=torch.Generator().manual_seed(0)).to(coords.device)
    shuffled = render(renderer, tq, tr, coords[:, perm], is_train=False)
    for k in full:
        assert torch.equal(shuffled[k], full[k][:, perm]), k
    # a second identical call is bitwise repeatable (no atomics / race in the forward path)
    again = render(renderer, tq, tr, coords, is_train=False)
    for k in full:
        assert torch.equal(again[k], full[k]), k


def test_fine_depths_are_sorted_and_in_range(case):
    cfg, renderer, weights, que, ref, tq, tr = case
    eng = renderer.engine(torch.device('cuda', 0))
    coords = tq['coords'][:, 300 * 800:300 * 800 + 32768]
    q = {k: v for k, v in tq.items() if not k.startswith('_')}
    q['coords'] = coords
    qc = eng.prepare_query(q)
    depth = eng.sample_coarse_depth(q['depth_range'], 32768, 64)
    near, far = float(que['depth_range'][0, 0]), float(que['depth_range'][0, 1])
    # (the last depth is 1 / (1/near + (1/far - 1/near)) as the reference rounds it: far to within an ulp or two)
    assert torch.equal(depth[:, 0], torch.full_like(depth[:, 0], near)) and float((depth[:, -1] - far).abs().max()) <= 2e-6 * far
    assert bool((depth[:, 1:] > depth[:, :-1]).all())
    hit = torch.rand(32768, 64, device=depth.device) ** 4
    for fdn in (32, 64):
        fine = eng.sample_fine_depth(qc, depth, hit, fdn)
        assert fine.shape == (32768, fdn) and bool((fine[:, 1:] >= fine[:, :-1]).all())
        assert float(fine.min()) >= near * (1 - 1e-6) and float(fine.max()) <= far * (1 + 1e-6)
    both = eng.sample_fine_depth(qc, depth, hit, 64, use_all=True)
    assert both.shape == (32768, 128) and bool((both[:, 1:] >= both[:, :-1]).all())


def test_reference_view_order_does_not_matter(case):
    """the cross-view statistics are symmetric in the views: permuting them changes the result only by summation order"""
    cfg, renderer, weights, que, ref, tq, tr = case
    coords = tq['coords'][:, 300 * 800:300 * 800 + 4096]
    a = render(renderer, tq, tr, coords, is_train=False)
    perm = torch.tensor([3, 7, 0, 5, 1, 6, 2, 4], device=coords.device)
    tr2 = {k: (v[perm] if torch.is_tensor(v) and v.shape[0] == 8 else v) for k, v in tr.items() if not k.startswith('_')}
    b = render(renderer, tq, tr2, coords, is_train=False)
    assert float((a['pixel_colors_nr'] - b['pixel_colors_nr']).abs().max()) <= 1e-4
    d = (a['pixel_colors_nr_fine'] - b['pixel_colors_nr_fine']).abs().max(-1)[0]
    assert float((d <= 2e-4).float().mean()) >= 0.97            # chained coarse -> fine: DESIGN.md 2.4


def test_sample_of_the_full_batch_against_the_oracle(case):
    from oracle import neuray_oracle as orc
    cfg, renderer, weights, que, ref, tq, tr = case
    idx = np.arange(300 * 800, 300 * 800 + 32768, 128)             # 256 rays of the batch
    torch.manual_seed(5)
    u = torch.rand(1, len(idx), 32)                                # the uniforms the training path is about to draw
    torch.manual_seed(5)
    got = render(renderer, tq, tr, tq['coords'][:, torch.from_numpy(idx).to('cuda:0')])
    ocfg = {**orc.DEFAULT_CFG, **cfg, 'coarse_use_vis': False, 'fine_use_vis': True}
    q = dict(que)
    q['coords'] = que['coords'][:, idx]
    want = orc.render_impl(weights, ocfg, q, ref, is_train=True, u=u.numpy())
    assert float(np.abs(got['pixel_colors_nr'].cpu().numpy() - want['pixel_colors_nr']).max()) <= 2e-4
    assert float(np.abs(got['hit_prob_nr'].cpu().numpy() - want['hit_prob_nr']).max()) <= 1e-4
    d = np.abs(got['pixel_colors_nr_fine'].cpu().numpy() - want['pixel_colors_nr_fine']).max(-1)[0]
    assert np.mean(d <= 2e-4) >= 0.95


def test_coarse_pass_of_8192_rays_of_the_image_against_the_oracle(case):
    """the coarse pass on IDENTICAL inputs, sampled with 8 192 strided rays of the 640 000 (32 x the
    round-2 sample) against the numpy oracle.  Gate: SURVEY 8(c)'s 2e-4 / 1e-4; round 2 sat at 1.6e-5 .. 1.6e-4 there because
    the texel coordinates took u * RN(1 / (W - 1)) for the reference's correctly rounded u / (W - 1) (1 ulp = 1.2e-5 texels on
    a 200-wide white-noise map); with the residual-corrected quotient (nr_device.h nr_div_refined) the worst ray of the
    sample is ~1.4e-6, asserted with 15 x headroom so that a regression of the feature-path geometry cannot hide below 2e-4."""
    from oracle import neuray_oracle as orc
    cfg, renderer, weights, que, ref, tq, tr = case
    n = 8192
    idx = np.linspace(0, que['coords'].shape[1] - 1, n).astype(np.int64)
    got = render(renderer, tq, tr, tq['coords'][:, torch.from_numpy(idx).to('cuda:0')], is_train=False)
    ocfg = {**orc.DEFAULT_CFG, **cfg, 'coarse_use_vis': False, 'fine_use_vis': True, 'use_hierarchical_sampling': False}
    pix, hit = [], []
    for i in range(0, n, 1024):
        q = dict(que)
        q['coords'] = que['coords'][:, idx[i:i + 1024]]
        o = orc.render_impl(weights, ocfg, q, ref)
        pix.append(o['pixel_colors_nr'])
        hit.append(o['hit_prob_nr'])
    ep = np.abs(got['pixel_colors_nr'].cpu().numpy() - np.concatenate(pix, 1)).max(-1)[0]
    # (render() drops hit_prob* in eval - renderer.py:244; the hit probabilities are compared through render_impl below)
    print('coarse pixels, %d rays vs oracle: max %.2e, p99.9 %.2e, median %.2e (gate 2e-4)' % (n, ep.max(), np.percentile(ep, 99.9), np.median(ep)))
    assert ep.max() <= 2e-5
    q = {k: v for k, v in tq.items() if not k.startswith('_')}
    q['coords'] = tq['coords'][:, torch.from_numpy(idx[:2048]).to('cuda:0')]
    with torch.no_grad():
        h = renderer.render_impl(q, {k: v for k, v in tr.items() if not k.startswith('_')}, False)['hit_prob_nr'].cpu().numpy()
    eh = np.abs(h - np.concatenate(hit, 1)[:, :2048]).max()
    print('coarse hit_prob, 2048 rays vs oracle: max %.2e (gate 1e-4)' % eh)
    assert eh <= 2e-5


def test_init_net_kernels_at_full_size():
    """f-2 / f-3 kernels on 8 x 800 x 800 / 160 x 160 x 64: consistent inputs give zeros, outputs are finite"""
    from neuray_amd import synthetic
    from neuray_amd.network import init_net, render_ops
    dev = torch.device('cuda', 0)
    _, ref = synthetic.make_scene(800, 800, 8, seed=0)
    img = torch.rand(1, 3, 800, 800, generator=torch.Generator().manual_seed(1))
    info = {'imgs': img.repeat(8, 1, 1, 1).to(dev), 'poses': torch.from_numpy(np.repeat(ref['poses'][:1], 8, 0)).to(dev),
            'Ks': torch.from_numpy(ref['Ks']).to(dev), 'depth_range': torch.from_numpy(ref['depth_range']).to(dev),
            'depth': torch.full((8, 1, 800, 800), 3.7, device=dev)}
    out = init_net.get_diff_feats(info, init_net.extract_depth_for_init(info))
    assert out.shape == (8, 8, 800, 800) and float(out.abs().max()) <= 1e-3       # (re-projection lands within ~1e-4 px of the pixel; the image is white noise)
    info2 = {k: torch.from_numpy(ref[k]).to(dev) for k in ('imgs', 'poses', 'Ks', 'depth_range')}
    info2['depth'] = 2.5 + 3 * torch.rand(8, 1, 800, 800, device=dev)
    out2 = init_net.get_diff_feats(info2, init_net.extract_depth_for_init(info2))
    assert torch.isfinite(out2).all() and float(out2[:, 6].max()) <= 1.5 + 1e-6 and float(out2.min()) >= 0.0
    eng = render_ops.engine_for(dev)
    f = torch.randn(8, 32, 160, 160, device=dev)
    prj = init_net.construct_project_matrix(0.2, 0.2, info2['Ks'], info2['poses'])
    dv = init_net.get_depth_vals(info2['depth_range'], 64)
    ids = torch.tensor([[(v + 1) % 8, (v + 2) % 8, (v + 3) % 8] for v in range(8)], device=dev)
    var = eng.warp_variance(f, f, ids, prj, prj, dv)
    assert var.shape == (8, 32, 64, 160, 160) and torch.isfinite(var).all() and float(var.min()) >= -1e-4
    same = eng.warp_variance(f[:2], f[:2], torch.tensor([[0, 0, 0], [1, 1, 1]], device=dev), prj[:2], prj[:2], dv[:2])
    assert float(same[:, :, :, 2:-2, 2:-2].abs().max()) <= 1e-3
