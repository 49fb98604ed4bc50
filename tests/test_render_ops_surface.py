"""Function-level parity of the network.render_ops mirror against the oracle,
on the emulator (CPU) and on the MI355X (gpu)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, load_case
from emu_util import emu_lib, to_torch
from oracle import neuray_oracle as orc
from neuray_amd.network import render_ops as ro

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def dev(request):
    ro._ENGINES.clear()
    if request.param == 'emu':
        ro._TEST_LIB = emu_lib()
        yield 'cpu'
        ro._TEST_LIB = None
        ro._ENGINES.clear()
    else:
        ro._TEST_LIB = None
        yield 'cuda:0'


def test_ray_and_depth_functions(dev):
    cfg, que, ref, out, mid, extra = load_case('c_adversarial')
    tq = to_torch(que, dev)
    depth, dists = ro.sample_depth(tq['depth_range'], tq['coords'], 32, False)
    assert np.array_equal(depth.cpu().numpy(), mid['que_depth'])
    c, d = ro.coords2rays(tq['coords'], tq['poses'], tq['Ks'])
    oc, od = orc.coords2rays(que['coords'], que['poses'], que['Ks_inv'])
    assert np.array_equal(c.cpu().numpy(), oc) and np.array_equal(d.cpu().numpy(), od)
    pts, qdir = ro.depth2points(tq, depth)
    assert np.array_equal(pts.cpu().numpy(), orc.depth2points(que['coords'], que['poses'], que['Ks_inv'], mid['que_depth'])[0])
    np.testing.assert_allclose(pts.cpu().numpy(), mid['que_pts'], atol=2e-6)
    np.testing.assert_allclose(qdir.cpu().numpy(), mid['que_dir'], atol=1e-6)
    assert np.array_equal(ro.depth2inv_dists(depth, tq['depth_range']).cpu().numpy(), orc.depth2inv_dists(mid['que_depth'], que['depth_range']))
    np.testing.assert_allclose(ro.depth2inv_dists(depth, tq['depth_range']).cpu().numpy(), mid['que_dists'], atol=1e-6)
    assert np.array_equal(ro.depth2dists(depth).cpu().numpy(), orc.depth2dists(mid['que_depth']))


def test_projection_and_gather_functions(dev):
    cfg, que, ref, out, mid, extra = load_case('c_adversarial')
    tr = to_torch(ref, dev)
    pts = torch.from_numpy(mid['que_pts']).to(dev)
    prj = ro.project_points_dict(tr, pts)
    for k, tol in (('pts', 2e-3), ('depth', 1e-5), ('dir', 1e-6), ('ray_feats', 2e-4), ('rgb', 2e-5)):
        np.testing.assert_allclose(prj[k].cpu().numpy(), mid['prj.' + k], rtol=1e-5, atol=tol)
    assert np.array_equal(prj['mask'].cpu().numpy(), mid['prj.mask'])
    flat = pts.reshape(-1, 3)
    p2, valid, z = ro.project_points_coords(flat, tr['poses'], tr['Ks'])
    H = orc.compute_H(ref['poses'], ref['Ks'])
    op2, ovalid, oz = orc.project_points_coords(mid['que_pts'].reshape(-1, 3), H)
    assert np.array_equal(p2.cpu().numpy(), op2) and np.array_equal(valid.cpu().numpy(), ovalid) and np.array_equal(z.cpu().numpy(), oz)
    d = ro.project_points_directions(tr['poses'], flat)
    assert np.array_equal(d.cpu().numpy(), orc.project_points_directions(ref['poses'], mid['que_pts'].reshape(-1, 3)))


def test_compositing_and_fine_sampling_functions(dev):
    cfg, que, ref, out, mid, extra = load_case('a_small')
    tq = to_torch(que, dev)
    alpha = 1 - np.exp(-np.maximum(mid['density'], 0))
    hp = ro.alpha_values2hit_prob(torch.from_numpy(alpha).to(dev)).cpu().numpy()
    assert np.array_equal(hp, orc.alpha_values2hit_prob(alpha))            # sequential cumprod order: bit-exact
    np.testing.assert_allclose(hp, out['hit_prob_nr'], atol=2e-6)
    depth = torch.from_numpy(mid['que_depth']).to(dev)
    fine = ro.sample_fine_depth(depth, torch.from_numpy(out['hit_prob_nr']).to(dev), tq['depth_range'], 16, False)
    want = orc.sample_fine_depth(mid['que_depth'], out['hit_prob_nr'], que['depth_range'], 16)
    np.testing.assert_allclose(fine.cpu().numpy(), want, rtol=2e-5, atol=2e-5)


def test_dist_decoder_rows(dev):
    """a9 stand-alone on arbitrary rows (predict_mean / forward), both decoder variants."""
    from conftest import load_weights
    from neuray_amd.engine import RenderEngine
    eng = ro.engine_for(dev)
    rng = np.random.RandomState(0)
    feats = rng.randn(3, 37, 32).astype(np.float32)
    for vis in (False, True):
        w = load_weights(vis)
        pk = eng.pack_pass(w, 'dist_decoder.', 'agg_net.')
        mean, var, v, aw = eng.dist_decoder_rows(torch.from_numpy(feats).to(dev), pk)
        om, ov, ovis, oaw = orc.dist_decoder_forward(w, 'dist_decoder.', feats)
        np.testing.assert_allclose(mean.cpu().numpy(), om, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(var.cpu().numpy(), ov, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(aw.cpu().numpy(), oaw, atol=2e-6)
        assert (v is None) == (ovis is None)
        if v is not None:
            np.testing.assert_allclose(v.cpu().numpy(), ovis, atol=2e-6)


def test_sampling_branches_the_renderer_never_takes(dev):
    """sample_depth(random_sample=True) and sample_fine_depth(inv_mode=False) against the reference
    (tests/golden/case_sampling_branches.npz); the uniforms are drawn as the reference draws them, so a seeded call
    consumes the generator identically (checked on the CPU generator)."""
    z = np.load(os.path.join(GOLDEN_DIR, 'case_sampling_branches.npz'))
    dr = torch.from_numpy(z['depth_range']).to(dev)
    coords = torch.zeros(2, 11, 2, device=dev)
    if dev == 'cpu':
        torch.manual_seed(77)
        depth, dists = ro.sample_depth(dr, coords, 16, True)
    else:                      # the device generator differs from the CPU one the golden was drawn with: feed the draws
        eng = ro.engine_for(dev)
        u = torch.from_numpy(z['u']).to(dev)
        depth = torch.stack([eng.sample_coarse_depth(dr[q], 11, 16, u[q]) for q in range(2)], 0)
        dists = torch.cat([depth[..., 1:], torch.full_like(depth[..., :1], 1e6)], -1) - depth
    assert np.max(np.abs(depth.cpu().numpy() - z['depth']) / z['depth']) <= 2e-7
    assert np.max(np.abs(dists.cpu().numpy()[..., :-1] - z['dists'][..., :-1])) <= 2e-6 and float(dists[..., -1].min()) > 9e5
    sd, hit = torch.from_numpy(z['sorted_depth']).to(dev), torch.from_numpy(z['hit']).to(dev)
    fine = ro.sample_fine_depth(sd, hit, dr, 12, False, inv_mode=False)
    assert fine.shape == (2, 11, 12) and np.max(np.abs(fine.cpu().numpy() - z['fine_lin']) / z['fine_lin']) <= 1e-5
    torch.manual_seed(78)
    fine_r = ro.sample_fine_depth(sd, hit, dr, 12, True, inv_mode=False)        # (these uniforms are CPU draws on every device)
    assert np.max(np.abs(fine_r.cpu().numpy() - z['fine_lin_rand']) / z['fine_lin_rand']) <= 1e-5
