"""BASELINE.json config 2 at full size on the MI355X: 800x800 query, 8 reference views, 64 coarse + 32 fine
samples.  The oracle cannot render 640k rays in seconds, so full-size parity goes through size-independent
properties (ray independence / batching invariance, compositing invariants, sortedness) plus a direct
comparison with the oracle on a strided sample of rays."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def full_render():
    import bench
    dev = torch.device('cuda', 0)
    cfg, renderer, weights, que, ref, tq, tr = bench.build_case(dev, 32, seed=0)
    out = bench.render_image(renderer, tq, tr)
    torch.cuda.synchronize()
    return cfg, renderer, weights, que, ref, tq, tr, {k: v.cpu().numpy() for k, v in out.items()}


def test_outputs_finite_and_shaped(full_render):
    cfg, renderer, weights, que, ref, tq, tr, out = full_render
    n = 800 * 800
    assert out['pixel_colors_nr_fine'].shape == (1, n, 3) and out['ray_mask_fine'].shape == (1, n)
    for k in ('pixel_colors_nr', 'pixel_colors_nr_fine'):
        assert np.isfinite(out[k]).all()
        assert out[k].min() >= -1e-5 and out[k].max() <= 1 + 1e-4     # convex blend of image colours in [0,1)


def test_batching_invariance_bitwise(full_render):
    """Rays are independent: a different ray_batch_num (different tiling of the grid) gives identical bits."""
    cfg, renderer, weights, que, ref, tq, tr, out = full_render
    renderer.cfg['ray_batch_num'], renderer.cfg['hip_min_ray_batch'] = 4096, 0        # (exactly 4096 rays per launch: render() merges batches otherwise)
    q = dict(tq)
    q['coords'] = tq['coords'][:, 100000:100000 + 3 * 4096 + 123]
    with torch.no_grad():
        sub = renderer.render(q, {k: v for k, v in tr.items() if not k.startswith('_')}, False)
    renderer.cfg['ray_batch_num'] = cfg['ray_batch_num']
    for k in ('pixel_colors_nr', 'pixel_colors_nr_fine'):
        assert np.array_equal(sub[k].cpu().numpy(), out[k][:, 100000:100000 + 3 * 4096 + 123]), k


def test_sample_against_oracle(full_render):
    """Strided sample of rays against the oracle, stage by stage on identical inputs (tight), then chained.

    The chained coarse->fine comparison is statistical by necessity: sample_fine_depth (render_ops.py:218-219)
    replaces denominators below 1e-5 by 1, and pdf entries of empty samples sit at 1e-5/sum(hit_prob) ~ 1.2e-5,
    so fp32-level noise on the coarse hit_prob flips that branch for a few samples and moves them by up to a
    bin width.  The oracle perturbed by 4e-6 noise shows the same 1-2% of rays beyond 2e-4 (parity)."""
    from oracle import neuray_oracle as orc
    cfg, renderer, weights, que, ref, tq, tr, out = full_render
    dev = tq['coords'].device
    idx = np.linspace(0, 800 * 800 - 1, 768).astype(np.int64)
    q = dict(que)
    q['coords'] = que['coords'][:, idx]
    ocfg = dict(cfg, coarse_use_vis=False, fine_use_vis=True)
    want = orc.render_impl(weights, ocfg, q, ref)
    # (1) coarse pass: direct, tight
    err_c = np.abs(out['pixel_colors_nr'][:, idx] - want['pixel_colors_nr']).max()
    assert err_c <= 2e-4, err_c
    tq2 = {k: v for k, v in tq.items() if not k.startswith('_')}
    tq2['coords'] = tq['coords'][:, torch.from_numpy(idx).to(dev)]
    eng = renderer.engine(dev)
    with torch.no_grad():
        tq2['_neuray_qconst'] = eng.prepare_query(tq2)
        # (2) fine sampling on the oracle's coarse hit_prob (identical inputs)
        fd = eng.sample_fine_depth(tq2['_neuray_qconst'], torch.from_numpy(want['_coarse_depth'][0]).to(dev),
                                   torch.from_numpy(want['hit_prob_nr'][0]).to(dev), 32).cpu().numpy()
        assert np.mean(np.abs(fd - want['_fine_depth'][0]) <= 2e-5) >= 0.999
        assert np.all(np.diff(fd, axis=-1) >= 0)
        # (3) fine pass on the oracle's fine depths (identical inputs): tight
        got = renderer.render_by_depth(torch.from_numpy(want['_fine_depth']).to(dev), tq2, tr, False, True)
    assert np.abs(got['pixel_colors_nr'].cpu().numpy() - want['pixel_colors_nr_fine']).max() <= 2e-4
    assert np.abs(got['hit_prob_nr'].cpu().numpy() - want['hit_prob_nr_fine']).max() <= 1e-4
    assert np.array_equal(got['ray_mask'].cpu().numpy(), want['ray_mask_fine'])
    # (4) chained end to end: statistical.  The few displaced fine samples of DESIGN.md 2.4 dominate the PSNR of a small sample
    # (white-noise images / features are the pathological input for them), so the gate is taken on 8192 strided rays, where
    # bench.py measures 99.7 % within 2e-4 and 75.5 dB: BASELINE.md B4's 60 dB with margin.
    err = np.abs(out['pixel_colors_nr_fine'][:, idx] - want['pixel_colors_nr_fine']).max(-1)
    assert np.mean(err <= 2e-4) >= 0.97, np.mean(err <= 2e-4)
    idx8 = np.linspace(0, 800 * 800 - 1, 8192).astype(np.int64)
    q8 = dict(que)
    q8['coords'] = que['coords'][:, idx8]
    want8 = np.concatenate([orc.render_impl(weights, ocfg, dict(q8, coords=q8['coords'][:, i:i + 1024]), ref)['pixel_colors_nr_fine']
                            for i in range(0, 8192, 1024)], 1)
    err8 = np.abs(out['pixel_colors_nr_fine'][:, idx8] - want8).max(-1)
    psnr = orc.psnr_uint8(out['pixel_colors_nr_fine'][:, idx8], want8)
    assert np.mean(err8 <= 2e-4) >= 0.99 and psnr >= 60.0, (float(np.mean(err8 <= 2e-4)), psnr)
