"""Pins the numpy oracle (oracle/neuray_oracle.py) against golden vectors produced by the
reference's own modules (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import CASES, case_uses_vis_weights, load_case, load_weights, oracle_cfg
from oracle import neuray_oracle as orc

# fp32 tolerances: pixel colours 2e-4 abs, hit_prob 1e-4 abs
TOL_PIXEL = 2e-4
TOL_HIT = 1e-4


@pytest.mark.parametrize('name', CASES)
def test_render_impl_matches_reference(name):
    cfg, que, ref, out, mid, extra = load_case(name)
    weights = load_weights(case_uses_vis_weights(name))
    # stage-wise: the fine samples are placed from the reference's own coarse hit_prob, so both
    # passes are compared on identical inputs (see oracle.render_impl docstring)
    got = orc.render_impl(weights, oracle_cfg(cfg), que, ref, is_train=extra['is_train'], u=extra['u'],
                          coarse_hit_prob=out['hit_prob_nr'])
    for k, v in out.items():
        assert k in got, k
        g = got[k]
        assert g.shape == v.shape, (k, g.shape, v.shape)
        if v.dtype == np.bool_:
            assert np.array_equal(g, v), k
        else:
            tol = TOL_HIT if k.startswith('hit_prob') else TOL_PIXEL
            if k.startswith('render_depth'):
                tol = 2e-3  # metres-scale quantity (depth range up to 13)
            assert np.max(np.abs(g - v)) <= tol, (k, float(np.max(np.abs(g - v))))


@pytest.mark.parametrize('name', ['a_small', 'c_adversarial'])
def test_intermediates_match_reference(name):
    cfg, que, ref, out, mid, extra = load_case(name)
    weights = load_weights(False)
    c = {**orc.DEFAULT_CFG, **oracle_cfg(cfg)}
    rn = que['coords'].shape[1]
    que_depth = orc.sample_depth(que['depth_range'], rn, c['depth_sample_num'])
    np.testing.assert_allclose(que_depth, mid['que_depth'], rtol=1e-6, atol=0)
    _, aux = orc.render_by_depth(weights, c, que_depth, que, ref, False, False, return_aux=True)
    np.testing.assert_allclose(aux['que_dists'], mid['que_dists'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(aux['que_pts'], mid['que_pts'], rtol=0, atol=2e-6)
    np.testing.assert_allclose(aux['que_dir'], mid['que_dir'], rtol=0, atol=1e-6)
    prj = aux['prj']
    assert np.array_equal(prj['mask'], mid['prj.mask'])
    np.testing.assert_allclose(prj['pts'], mid['prj.pts'], rtol=1e-5, atol=2e-3)
    np.testing.assert_allclose(prj['depth'], mid['prj.depth'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(prj['dir'], mid['prj.dir'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(prj['ray_feats'], mid['prj.ray_feats'], rtol=0, atol=2e-4)
    np.testing.assert_allclose(prj['rgb'], mid['prj.rgb'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(prj['img_feats'], mid['prj.img_feats'], rtol=0, atol=2e-4)
    np.testing.assert_allclose(prj['_mean'], mid['mean'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(prj['_var'], mid['var'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(prj['_aw'], mid['aw'], rtol=0, atol=1e-5)
    np.testing.assert_allclose(prj['vis'], mid['prj.vis'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(prj['hit_prob'], mid['prj.hit_prob'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(prj['alpha'], mid['prj.alpha'], rtol=0, atol=5e-3)
    np.testing.assert_allclose(aux['density'], mid['density'], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(aux['colors'], mid['colors'], rtol=0, atol=1e-4)


def test_fine_depths_sorted_and_in_range():
    cfg, que, ref, out, mid, extra = load_case('b_default')
    weights = load_weights(False)
    got = orc.render_impl(weights, oracle_cfg(cfg), que, ref)
    d = got['_fine_depth']
    assert np.all(np.diff(d, axis=-1) >= 0)
    near, far = que['depth_range'][0]
    assert d.min() >= near * (1 - 1e-5) and d.max() <= far * (1 + 1e-5)
    assert np.all(np.sum(got['hit_prob_nr'], -1) <= 1 + 1e-5)


@pytest.mark.parametrize('name', CASES)
def test_render_impl_chained_end_to_end(name):
    """Fully chained coarse->fine run.  Fine-sample placement amplifies fp32 noise on near-empty
    rays (sum(hit_prob) ~ 1e-3), so the bound is: 95% of rays within the stage tolerance, every ray
    within 5e-3, PSNR(ours, reference) >= 60 dB."""
    cfg, que, ref, out, mid, extra = load_case(name)
    weights = load_weights(case_uses_vis_weights(name))
    got = orc.render_impl(weights, oracle_cfg(cfg), que, ref, is_train=extra['is_train'], u=extra['u'])
    for k in ('pixel_colors_nr', 'pixel_colors_nr_fine'):
        err = np.max(np.abs(got[k] - out[k]), -1)
        assert np.mean(err <= TOL_PIXEL) >= 0.95, (k, float(np.mean(err <= TOL_PIXEL)))
        assert err.max() <= 5e-3, (k, float(err.max()))
        assert orc.psnr_uint8(got[k], out[k]) >= 60.0
    assert np.array_equal(got['ray_mask_fine'], out['ray_mask_fine'])


@pytest.mark.parametrize('name', ['a_small', 'b_default', 'c_adversarial', 'e_use_all'])
def test_torch_eager_port_matches_reference(name):
    """The eager-PyTorch port used as the 'stock PyTorch-ROCm' baseline in bench.py is the reference's computation."""
    import torch
    from oracle import torch_eager_port as tep
    cfg, que, ref, out, mid, extra = load_case(name)
    w = {k: torch.from_numpy(v) for k, v in load_weights(False).items()}
    tq = {k: torch.from_numpy(v) for k, v in que.items()}
    tr = {k: torch.from_numpy(v) for k, v in ref.items()}
    with torch.no_grad():
        got = tep.render_impl(w, {**orc.DEFAULT_CFG, **oracle_cfg(cfg)}, tq, tr)
    assert np.max(np.abs(got['pixel_colors_nr'].numpy() - out['pixel_colors_nr'])) <= TOL_PIXEL
    assert np.max(np.abs(got['hit_prob_nr'].numpy() - out['hit_prob_nr'])) <= TOL_HIT
    assert np.array_equal(got['ray_mask'].numpy(), out['ray_mask'])
    err = np.max(np.abs(got['pixel_colors_nr_fine'].numpy() - out['pixel_colors_nr_fine']), -1)
    assert np.mean(err <= TOL_PIXEL) >= 0.95 and err.max() <= 5e-3


def test_torch_eager_port_gradients_match_reference():
    """Autograd through the eager port against the reference's own autograd (tests/golden/case_g_grads.npz): pins the
    gradient oracle that the backward kernels (next round) will be checked against."""
    import os
    import torch
    from conftest import GOLDEN_DIR
    from oracle import torch_eager_port as tep
    z = np.load(os.path.join(GOLDEN_DIR, 'case_g_grads.npz'))
    cfg = __import__('ast').literal_eval(str(z['cfg_json']))
    que = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('que.')}
    ref = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('ref.')}
    w = {k: torch.from_numpy(v).requires_grad_(True) for k, v in load_weights(False).items()}
    for t in (ref['ray_feats'], ref['img_feats'], que['ray_feats']):
        t.requires_grad_(True)
    out = tep.render_impl(w, {**orc.DEFAULT_CFG, **oracle_cfg(cfg)}, que, ref, is_train=True, u=torch.from_numpy(z['u']))
    for k in ('pixel_colors_nr', 'hit_prob_self', 'hit_prob_self_fine'):
        np.testing.assert_allclose(out[k].detach().numpy(), z['out.' + k], atol=1e-4)
    loss = sum((torch.from_numpy(z['lw.' + k]) * out[k]).sum() for k in ('pixel_colors_nr', 'pixel_colors_nr_fine',
                                                                         'hit_prob_self', 'hit_prob_self_fine'))
    assert abs(float(loss) - float(z['loss'])) <= 1e-3
    loss.backward()

    def close(got, want, name):
        scale = max(1e-3, float(np.abs(want).max()))
        assert np.max(np.abs(got - want)) <= 2e-3 * scale, (name, float(np.max(np.abs(got - want))), scale)

    for k, p in w.items():
        close(p.grad.numpy() if p.grad is not None else np.zeros_like(p.detach().numpy()), z['grad.' + k], k)
    close(ref['ray_feats'].grad.numpy(), z['grad.ref.ray_feats'], 'ref.ray_feats')
    close(ref['img_feats'].grad.numpy(), z['grad.ref.img_feats'], 'ref.img_feats')
    close(que['ray_feats'].grad.numpy(), z['grad.que.ray_feats'], 'que.ray_feats')


@pytest.mark.parametrize('name', ['c2_tile_32', 'c2_smooth'])
def test_eager_port_in_float64_reproduces_the_references_float64_run(name):
    """bench.py's whole-image float64 leg runs oracle/torch_eager_port.py in float64.  Pinned here: on the tiles whose float64 evaluation
    by the REFERENCE ITSELF is committed (case_c2_*_f64.npz), the port in float64 lands on it - coarse pixels to 1e-7, chained fine
    pixels to 1e-5 (the chained quantity amplifies the last bits of the coarse hit probabilities: DESIGN.md 2.4)."""
    import os
    import torch
    from conftest import GOLDEN_DIR, load_weights, oracle_cfg
    from test_baseline_shapes import load_tile
    from oracle import torch_eager_port as tep
    z0, cfg, que, ref, out, mid = load_tile(name)
    que['coords'] = z0['coords']
    z = np.load(os.path.join(GOLDEN_DIR, 'case_%s_f64.npz' % name))
    up = lambda v: torch.from_numpy(np.asarray(v)).double() if np.asarray(v).dtype == np.float32 else torch.from_numpy(np.asarray(v))      # noqa: E731
    w = {k: torch.from_numpy(v).double() for k, v in load_weights(False).items()}
    with torch.no_grad():
        o = tep.render_impl(w, oracle_cfg(cfg), {k: up(v) for k, v in que.items()}, {k: up(v) for k, v in ref.items()})
    assert o['pixel_colors_nr'].dtype == torch.float64
    assert np.abs(o['pixel_colors_nr'].numpy() - z['out.pixel_colors_nr']).max() <= 1e-7
    assert np.abs(o['hit_prob_nr'].numpy() - z['out.hit_prob_nr']).max() <= 1e-7
    assert np.abs(o['pixel_colors_nr_fine'].numpy() - z['out.pixel_colors_nr_fine']).max() <= 1e-5
