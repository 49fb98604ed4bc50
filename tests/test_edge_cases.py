"""Edge cases of the render path against the oracle (seeded random weights from the mirror's constructors):
minimum / maximum sample counts, a single ray, ragged ray counts (not a multiple of the 16-point tile), the maximum
number of reference views, non-square images whose feature maps are not exactly 1/4 resolution, and argument
validation of the C ABI."""
import os

import numpy as np
import pytest
import torch

from emu_util import emu_lib, to_torch
from oracle import neuray_oracle as orc
from neuray_amd import synthetic
from neuray_amd.network.renderer import NeuralRayBaseRenderer

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


def build(cfg, backend, seed=0):
    torch.manual_seed(seed)
    r = NeuralRayBaseRenderer(cfg).eval()
    weights = {k: v.detach().numpy().copy() for k, v in r.state_dict().items()}
    if backend == 'emu':
        r._engine_test_lib = emu_lib()
        return r, weights, 'cpu'
    return r.cuda(), weights, 'cuda:0'


def scene(h, w, rfn, rn, seed, fh=None, fw=None):
    que, ref = synthetic.make_scene(h, w, rfn, seed=seed)
    if fh is not None:      # feature maps that are not exactly h/4 x w/4
        rng = np.random.RandomState(seed + 7)
        ref['ray_feats'] = rng.randn(rfn, 32, fh, fw).astype(np.float32)
        ref['img_feats'] = rng.randn(rfn, 32, fh, fw).astype(np.float32)
    rng = np.random.RandomState(seed + 1)
    que['coords'] = (rng.rand(1, rn, 2) * np.array([w - 1, h - 1])).astype(np.float32)
    que['Ks_inv'] = torch.inverse(torch.from_numpy(que['Ks'])).numpy()
    return que, ref


def check(cfg, que, ref, backend, tol=2e-4, tweak=None):
    r, weights, dev = build(cfg, backend)
    if tweak is not None:
        with torch.no_grad():
            tweak(r)
        weights = {k: v.detach().cpu().numpy().copy() for k, v in r.state_dict().items()}
    with torch.no_grad():
        got = r.render_impl(to_torch(que, dev), to_torch(ref, dev), False)
    ocfg = dict(cfg, coarse_use_vis=cfg.get('dist_decoder_cfg', {}).get('use_vis', True), fine_use_vis=True)
    want = orc.render_impl(weights, ocfg, que, ref)
    assert np.max(np.abs(got['pixel_colors_nr'].cpu().numpy() - want['pixel_colors_nr'])) <= tol
    assert np.max(np.abs(got['hit_prob_nr'].cpu().numpy() - want['hit_prob_nr'])) <= 1e-4
    assert np.array_equal(got['ray_mask'].cpu().numpy(), want['ray_mask'])
    return got, want


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('dn,rn,rfn', [(3, 1, 1), (5, 7, 2), (17, 33, 3), (128, 5, 2)])
def test_sample_and_ray_count_extremes(dn, rn, rfn, backend):
    cfg = {'depth_sample_num': dn, 'agg_net_cfg': {'sample_num': dn}, 'dist_decoder_cfg': {'use_vis': dn % 2 == 1}}
    que, ref = scene(40, 56, rfn, rn, seed=dn)
    check(cfg, que, ref, backend)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('scale', [1.0, 12.0])
def test_attention_softmax_shift_paths(scale, backend):
    """The ray kernel shifts the attention softmax by the bound |q| max|k| (one pass over the keys) and falls back to
    the exact two-pass row maximum when the bound exceeds 40.  scale = 12 on the query / key projections makes the
    logits ~144x larger (bounds far above 40, near one-hot rows): the fallback must agree with the oracle too."""
    cfg = {'depth_sample_num': 24, 'agg_net_cfg': {'sample_num': 24}, 'dist_decoder_cfg': {'use_vis': False}}
    que, ref = scene(40, 56, 3, 19, seed=11)

    def tweak(r):
        for name, prm in r.named_parameters():
            if name.endswith('ray_attention.w_qs.weight') or name.endswith('ray_attention.w_ks.weight'):
                prm.mul_(scale)
    check(cfg, que, ref, backend, tweak=tweak)


@pytest.mark.parametrize('backend', BACKENDS)
def test_sixteen_reference_views(backend):
    cfg = {'depth_sample_num': 8, 'agg_net_cfg': {'sample_num': 8}, 'dist_decoder_cfg': {'use_vis': False}}
    que, ref = scene(32, 32, 16, 9, seed=3)
    check(cfg, que, ref, backend)


@pytest.mark.parametrize('backend', BACKENDS)
def test_feature_maps_not_quarter_resolution(backend):
    """ref_pad_interval changes the map / image ratio; the texel mapping must follow interpolate_feats (ops.py:28-29)."""
    cfg = {'use_hierarchical_sampling': True, 'depth_sample_num': 16, 'fine_depth_sample_num': 16,
           'agg_net_cfg': {'sample_num': 16}, 'fine_agg_net_cfg': {'sample_num': 16}, 'dist_decoder_cfg': {'use_vis': False}}
    que, ref = scene(44, 60, 4, 20, seed=5, fh=13, fw=17)
    got, want = check(cfg, que, ref, backend)
    err = np.max(np.abs(got['pixel_colors_nr_fine'].cpu().numpy() - want['pixel_colors_nr_fine']), -1)
    assert np.mean(err <= 2e-4) >= 0.9


@pytest.mark.parametrize('backend', BACKENDS)
def test_abi_rejects_bad_arguments(backend):
    from neuray_amd.engine import RenderEngine
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    eng = RenderEngine(dev, _test_lib=emu_lib() if backend == 'emu' else None)
    with pytest.raises(RuntimeError, match='dn'):
        eng.sample_coarse_depth(torch.tensor([[2.0, 6.0]], device=dev), 4, 2)          # assert(dn > 2), render_ops.py:157
    que, ref = scene(32, 32, 2, 4, seed=1)
    with pytest.raises(RuntimeError, match='views'):
        ref17 = {k: np.repeat(v[:1], 17, 0) for k, v in ref.items()}
        eng.prepare_views(to_torch(ref17, dev))
    with pytest.raises(RuntimeError, match='outside'):
        qc = eng.prepare_query(to_torch(que, dev))
        eng.sample_fine_depth(qc, torch.ones(4, 200, device=dev), torch.ones(4, 200, device=dev), 16)
    # the init-net kernels
    lib, f = eng.lib, torch.zeros(64, device=dev)
    assert lib.neuray_diff_feats(f.data_ptr(), f.data_ptr(), f.data_ptr(), 17, 8, 8, f.data_ptr(), None) != 0
    assert b'rfn=17' in lib.neuray_last_error()
    assert lib.neuray_diff_feats(f.data_ptr(), None, f.data_ptr(), 2, 8, 8, f.data_ptr(), None) != 0
    assert b'null' in lib.neuray_last_error()
    assert lib.neuray_warp_variance(f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), f.data_ptr(), 1, 1, 0, 8, 4, 4, f.data_ptr(), None) != 0
    assert b'n_num=0' in lib.neuray_last_error()
    with pytest.raises(AssertionError):            # a neighbour index beyond the source views never reaches the kernel
        eng.warp_variance(torch.zeros(1, 32, 4, 4, device=dev), torch.zeros(2, 32, 4, 4, device=dev), torch.tensor([[0, 2]], device=dev),
                          torch.eye(4, device=dev)[None], torch.eye(4, device=dev)[None].repeat(2, 1, 1), torch.ones(1, 8, device=dev))
        # (device-resident indices are range-checked on the device without stalling the host: the kernel ran on clamped indices and the
        # verdict is raised here, or by the next call)
        eng.check_deferred()
    if backend == 'hip':
        # a singular reference projection: torch.inverse would have raised at the call; inv_ex does not - its verdict is deferred the same way
        with pytest.raises(AssertionError, match='singular'):
            eng.warp_variance(torch.zeros(1, 32, 4, 4, device=dev), torch.zeros(2, 32, 4, 4, device=dev), torch.tensor([[0, 1]], device=dev),
                              torch.zeros(1, 4, 4, device=dev), torch.eye(4, device=dev)[None].repeat(2, 1, 1), torch.ones(1, 8, device=dev))
            eng.check_deferred()
        # product code drains the verdicts where it waits for the device anyway (network/render_ops.py check_deferred_inputs)
        from neuray_amd.network import render_ops
        render_ops.check_deferred_inputs(dev, wait=True)          # nothing pending: no error


# ---- the shapes of BASELINE.json's other configurations (parity cases, not bench lines) -------------------------
@pytest.mark.parametrize('backend', BACKENDS)
def test_config1_shape_coarse_only(backend):
    """configs[0]: 400x400, 3 reference views, 32 coarse samples, no hierarchical sampling."""
    cfg = {'use_hierarchical_sampling': False, 'depth_sample_num': 32, 'agg_net_cfg': {'sample_num': 32},
           'dist_decoder_cfg': {'use_vis': False}}
    que, ref = synthetic.make_scene(400, 400, 3, seed=21)
    rng = np.random.RandomState(22)
    que['coords'] = np.stack([rng.randint(0, 400, 37), rng.randint(0, 400, 37)], -1)[None].astype(np.float32)
    que['Ks_inv'] = torch.inverse(torch.from_numpy(que['Ks'])).numpy()
    got, want = check(cfg, que, ref, backend)
    assert 'pixel_colors_nr_fine' not in got and 'pixel_colors_nr_fine' not in want


@pytest.mark.parametrize('backend', BACKENDS)
def test_config3_shape_llff_padded_refs(backend):
    """configs[2] (LLFF fern/high): query 756x1008, reference images padded to 768x1024 (ref_pad_interval 32), feature
    maps 192x256, 8 views, wide depth range: the query intrinsics / size differ from the reference views'."""
    cfg = {'use_hierarchical_sampling': True, 'depth_sample_num': 64, 'fine_depth_sample_num': 64,
           'agg_net_cfg': {'sample_num': 64}, 'fine_agg_net_cfg': {'sample_num': 64}, 'dist_decoder_cfg': {'use_vis': False}}
    que, ref = synthetic.make_scene(768, 1024, 8, seed=23, depth_range=(1.2, 12.0), radius=5.0)
    # the query camera keeps the un-padded 756x1008 frame: same focal length, principal point of the smaller image
    que['Ks'][0, 0, 2] = 1008 / 2.0
    que['Ks'][0, 1, 2] = 756 / 2.0
    rng = np.random.RandomState(24)
    que['coords'] = np.stack([rng.randint(0, 1008, 21), rng.randint(0, 756, 21)], -1)[None].astype(np.float32)
    que['Ks_inv'] = torch.inverse(torch.from_numpy(que['Ks'])).numpy()
    r, weights, dev = build(cfg, backend)
    with torch.no_grad():
        got = r.render_impl(to_torch(que, dev), to_torch(ref, dev), False)
    ocfg = dict(cfg, coarse_use_vis=False, fine_use_vis=True)
    want = orc.render_impl(weights, ocfg, que, ref)
    # coarse pass on identical inputs: tight; the chained fine pass carries the reference's own resampling
    # discontinuity, so it is bounded like the other chained comparisons
    assert np.max(np.abs(got['pixel_colors_nr'].cpu().numpy() - want['pixel_colors_nr'])) <= 2e-4
    assert np.max(np.abs(got['hit_prob_nr'].cpu().numpy() - want['hit_prob_nr'])) <= 1e-4
    assert np.array_equal(got['ray_mask'].cpu().numpy(), want['ray_mask'])
    d = np.abs(got['pixel_colors_nr_fine'].cpu().numpy() - want['pixel_colors_nr_fine'])
    assert np.median(d) <= 2e-4 and np.mean(d <= 2e-3) >= 0.9


@pytest.mark.parametrize('backend', BACKENDS)
def test_two_query_views_in_one_call(backend):
    """qn = 2 against the reference (tests/golden/case_h_two_queries.npz): per-view pose / intrinsics / depth range, fine
    sampling normalised with view 0's range for both (render_ops.py:183,225)."""
    from conftest import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'case_h_two_queries.npz'))
    cfg = __import__('ast').literal_eval(str(z['cfg_json']))
    r = NeuralRayBaseRenderer(cfg).eval()
    r.load_state_dict({k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('w.')}, strict=True)
    if backend == 'emu':
        r._engine_test_lib = emu_lib()
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    r = r.to(dev)
    tq = {k[4:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith('que.')}
    tr = {k[4:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith('ref.')}
    with torch.no_grad():
        out = r.render_impl(tq, tr, False)
    want = {k[4:]: z[k] for k in z.files if k.startswith('out.')}
    assert set(out) == set(want)
    for k in ('pixel_colors_nr', 'pixel_colors_gt', 'render_depth'):
        assert out[k].shape == want[k].shape and np.max(np.abs(out[k].cpu().numpy() - want[k])) <= 2e-4, k
    assert np.max(np.abs(out['hit_prob_nr'].cpu().numpy() - want['hit_prob_nr'])) <= 1e-4
    assert np.array_equal(out['ray_mask'].cpu().numpy(), want['ray_mask'])
    d = np.abs(out['pixel_colors_nr_fine'].cpu().numpy() - want['pixel_colors_nr_fine']).max(-1)
    assert np.mean(d <= 2e-4) >= 0.9 and d.max() < 0.1            # chained coarse -> fine
