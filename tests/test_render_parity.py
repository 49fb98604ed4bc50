"""Parity of the HIP render path against the oracle and the reference-generated golden vectors.

Every test body runs on two backends through the same C ABI and the same Python host code:
  * 'emu' - the kernel sources compiled for the CPU fiber emulator (tests/emu); runs in the CPU container
  * 'hip' - libneuray_hip.so on a real MI355X (marked gpu)
Tolerances (fp32): pixel colours 2e-4, hit_prob 1e-4; geometry is bit-exact by the
rounding contract.
"""
import numpy as np
import pytest
import torch

from conftest import CASES, case_uses_vis_weights, load_case, load_weights, oracle_cfg
from emu_util import emu_lib, to_torch
from oracle import neuray_oracle as orc
from neuray_amd.network.renderer import NeuralRayBaseRenderer

TOL_PIXEL = 2e-4
TOL_HIT = 1e-4

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


def make_renderer(cfg, weights, backend):
    cfg = {k: v for k, v in cfg.items()}
    r = NeuralRayBaseRenderer(cfg)
    sd = {k: torch.from_numpy(v) for k, v in weights.items()}
    missing, unexpected = r.load_state_dict(sd, strict=True), None
    r.eval()
    if backend == 'emu':
        r._engine_test_lib = emu_lib()
        return r, 'cpu'
    assert torch.cuda.is_available(), "gpu test without a GPU"
    return r.cuda(), 'cuda:0'


def run_case(name, backend, cfg_override=None, coarse_hit_prob=None):
    cfg, que, ref, out, mid, extra = load_case(name)
    if cfg_override:
        cfg = {**cfg, **cfg_override}
    weights = load_weights(case_uses_vis_weights(name))
    r, dev = make_renderer(cfg, weights, backend)
    tq, tr = to_torch(que, dev), to_torch(ref, dev)
    torch.manual_seed(1234)   # make_golden.py seeds the CPU generator the same way before render_impl
    with torch.no_grad():
        got = r.render_impl(tq, tr, extra['is_train'])
    return cfg, que, ref, out, mid, extra, weights, {k: v.cpu().numpy() for k, v in got.items()}, (r, tq, tr)


@pytest.mark.parametrize('backend', BACKENDS)
def test_mfma_operand_layout(backend):
    """A = asymmetric, B = asymmetric: catches swapped operands / transposed C (cdna guide section 3)."""
    from neuray_amd.engine import RenderEngine
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    eng = RenderEngine(dev, _test_lib=emu_lib() if backend == 'emu' else None)
    g = torch.Generator().manual_seed(0)
    A = torch.randn(16, 4, generator=g).to(dev)
    B = torch.randn(4, 16, generator=g).to(dev)
    D = torch.zeros(16, 16, device=dev)
    assert eng.lib.neuray_mfma_selftest(A.data_ptr(), B.data_ptr(), D.data_ptr(), eng._stream()) == 0
    ref = (A.double() @ B.double()).float()
    assert torch.allclose(D.cpu(), ref.cpu(), atol=1e-6)


@pytest.mark.parametrize('backend', BACKENDS)
def test_lane_group_sum(backend):
    """The vector rows sum their partial dot products over the four 16-lane groups with two permlane swaps; the
    result must be (g0 + g1) + (g2 + g3) bit-exactly, in every lane (hipcc mis-compiles the obvious spelling:
    neuray_amd/csrc/nr_platform.h nr_group_sum, tests/hw/permlane_probe.hip)."""
    from neuray_amd.engine import RenderEngine
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    eng = RenderEngine(dev, _test_lib=emu_lib() if backend == 'emu' else None)
    x = torch.randn(64, generator=torch.Generator().manual_seed(3))
    xd, yd = x.to(dev), torch.zeros(64, device=dev)
    assert eng.lib.neuray_group_sum_selftest(xd.data_ptr(), yd.data_ptr(), eng._stream()) == 0
    g = x.view(4, 16)
    want = ((g[0] + g[1]) + (g[2] + g[3])).repeat(4)
    assert torch.equal(yd.cpu(), want)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('name', ['a_small', 'c_adversarial'])
def test_coarse_pass_stagewise(name, backend):
    """Every intermediate of the coarse pass against the oracle: geometry bit-exact, MLP stages ~1e-6."""
    cfg, que, ref, out, mid, extra = load_case(name)
    weights = load_weights(False)
    r, dev = make_renderer(cfg, weights, backend)
    eng = r.engine(dev)
    c = {**orc.DEFAULT_CFG, **oracle_cfg(cfg)}
    rn, dn = que['coords'].shape[1], c['depth_sample_num']
    tq, tr = to_torch(que, dev), to_torch(ref, dev)
    depth = eng.sample_coarse_depth(tq['depth_range'], rn, dn)
    o_depth = orc.sample_depth(que['depth_range'], rn, dn)
    assert np.array_equal(depth.cpu().numpy(), o_depth[0])
    assert np.array_equal(o_depth, mid['que_depth'])       # and bit-equal to the reference itself
    qc = eng.prepare_query(tq)
    views = eng.prepare_views(tr)
    res = eng.render_pass(qc, views, tq['coords'][0], depth, r._packed_pass(eng, False), use_vis=c['coarse_use_vis'],
                          want_depth=True, want_density=True, want_dbg=True)
    res = {k: v.cpu().numpy() for k, v in res.items()}
    o, aux = orc.render_by_depth(weights, c, o_depth, que, ref, False, False, return_aux=True)
    prj = aux['prj']
    tr_ = lambda t: t[:, 0].transpose(1, 2, 0)          # [rfn,qn,rn,dn] -> [rn,dn,rfn]
    dbg = res['dbg']
    assert np.array_equal(dbg[..., 0], tr_(prj['mask'][..., 0]))
    assert np.array_equal(dbg[..., 1], tr_(prj['pts'][..., 0]))      # u, v, z bit-exact (rounding contract)
    assert np.array_equal(dbg[..., 2], tr_(prj['pts'][..., 1]))
    assert np.array_equal(dbg[..., 3], tr_(prj['depth'][..., 0]))
    np.testing.assert_allclose(dbg[..., 4], tr_(prj['hit_prob'][..., 0]), atol=2e-6)
    np.testing.assert_allclose(dbg[..., 5], tr_(prj['vis'][..., 0]), atol=2e-6)
    np.testing.assert_allclose(dbg[..., 6], tr_(prj['_mean'][..., 0]), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(dbg[..., 9], tr_(prj['_var'][..., 1]), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(dbg[..., 10], tr_(prj['_aw'][..., 0]), atol=2e-6)
    rec = res['point_rec']
    np.testing.assert_allclose(rec[..., :16], aux['geo_feat'], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(rec[..., 16:19], aux['colors'][0], atol=5e-6)
    assert np.array_equal(rec[..., 19], aux['num_valid'])
    np.testing.assert_allclose(res['density'], aux['density'][0], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(res['hit_prob'], o['hit_prob_nr'][0], atol=1e-5)
    np.testing.assert_allclose(res['pixel'], o['pixel_colors_nr'][0], atol=1e-5)
    assert np.array_equal(res['ray_mask'], o['ray_mask'][0])
    # and against the reference's own intermediates (golden)
    np.testing.assert_allclose(res['density'], mid['density'][0], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(rec[..., 16:19], mid['colors'][0], atol=1e-4)
    np.testing.assert_allclose(res['hit_prob'], out['hit_prob_nr'][0], atol=TOL_HIT)
    np.testing.assert_allclose(res['pixel'], out['pixel_colors_nr'][0], atol=TOL_PIXEL)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('name', CASES)
def test_fine_sampling_matches_reference(name, backend):
    """a17 on the reference's own coarse result: sample_fine_depth + sort."""
    cfg, que, ref, out, mid, extra = load_case(name)
    r, dev = make_renderer(cfg, load_weights(case_uses_vis_weights(name)), backend)
    eng = r.engine(dev)
    c = {**orc.DEFAULT_CFG, **oracle_cfg(cfg)}
    rn, dn, fdn = que['coords'].shape[1], c['depth_sample_num'], c['fine_depth_sample_num']
    o_depth = orc.sample_depth(que['depth_range'], rn, dn)
    u = extra['u'] if extra['is_train'] else None
    want = orc.sample_fine_depth(o_depth, out['hit_prob_nr'], que['depth_range'], fdn, u)
    if c['fine_depth_use_all']:
        want = np.concatenate([o_depth, want], -1)
    want = np.sort(want, -1)[0]
    qc = eng.prepare_query(to_torch(que, dev))
    got = eng.sample_fine_depth(qc, torch.from_numpy(o_depth[0]).to(dev), torch.from_numpy(out['hit_prob_nr'][0]).to(dev),
                                fdn, use_all=c['fine_depth_use_all'], u=None if u is None else torch.from_numpy(u[0]))
    got = got.cpu().numpy()
    assert np.all(np.diff(got, axis=-1) >= 0)
    near, far = que['depth_range'][0]
    assert got.min() >= near * (1 - 1e-5) and got.max() <= far * (1 + 1e-5)
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('name', CASES)
def test_render_impl_matches_reference(name, backend):
    """Full coarse+fine render_impl against the golden outputs of the reference renderer."""
    cfg, que, ref, out, mid, extra, weights, got, _ = run_case(name, backend)
    for k, v in out.items():
        assert k in got, k
        assert got[k].shape == v.shape, (k, got[k].shape, v.shape)
    # coarse pass: direct comparison
    assert np.max(np.abs(got['pixel_colors_nr'] - out['pixel_colors_nr'])) <= TOL_PIXEL
    assert np.max(np.abs(got['hit_prob_nr'] - out['hit_prob_nr'])) <= TOL_HIT
    assert np.array_equal(got['ray_mask'], out['ray_mask'])
    # fine pass, chained: fine-sample placement amplifies fp32 noise on near-empty rays (see
    # tests/test_oracle_golden.py), hence the robust bound + PSNR
    err = np.max(np.abs(got['pixel_colors_nr_fine'] - out['pixel_colors_nr_fine']), -1)
    assert np.mean(err <= TOL_PIXEL) >= 0.95 and err.max() <= 5e-3, (float(np.mean(err <= TOL_PIXEL)), float(err.max()))
    assert orc.psnr_uint8(got['pixel_colors_nr_fine'], out['pixel_colors_nr_fine']) >= 60.0
    assert np.array_equal(got['ray_mask_fine'], out['ray_mask_fine'])
    if 'render_depth' in out:
        np.testing.assert_allclose(got['render_depth'], out['render_depth'], atol=2e-3)
    if 'hit_prob_self' in out:      # a19, training mode (case d): the coarse one sees identical inputs
        np.testing.assert_allclose(got['hit_prob_self'], out['hit_prob_self'], atol=TOL_HIT)
        err_s = np.max(np.abs(got['hit_prob_self_fine'] - out['hit_prob_self_fine']), -1)
        assert np.mean(err_s <= TOL_HIT) >= 0.9
    if 'pixel_colors_gt' in out:
        np.testing.assert_allclose(got['pixel_colors_gt'], out['pixel_colors_gt'], atol=1e-6)


@pytest.mark.parametrize('backend', BACKENDS)
def test_fine_pass_on_reference_fine_depths(backend):
    """Fine pass fed the oracle's fine depths (identical inputs on both sides) -> tight tolerance."""
    name = 'b_default'
    cfg, que, ref, out, mid, extra = load_case(name)
    weights = load_weights(False)
    r, dev = make_renderer(cfg, weights, backend)
    c = {**orc.DEFAULT_CFG, **oracle_cfg(cfg)}
    want = orc.render_impl(weights, c, que, ref, coarse_hit_prob=out['hit_prob_nr'])
    tq, tr = to_torch(que, dev), to_torch(ref, dev)
    with torch.no_grad():
        tq['_neuray_qconst'] = r.engine(dev).prepare_query(tq)
        got = r.render_by_depth(torch.from_numpy(want['_fine_depth']).to(dev), tq, tr, False, True)
    np.testing.assert_allclose(got['pixel_colors_nr'].cpu().numpy(), out['pixel_colors_nr_fine'], atol=TOL_PIXEL)
    np.testing.assert_allclose(got['hit_prob_nr'].cpu().numpy(), out['hit_prob_nr_fine'], atol=TOL_HIT)
    np.testing.assert_allclose(got['pixel_colors_nr'].cpu().numpy(), want['pixel_colors_nr_fine'], atol=1e-5)


@pytest.mark.parametrize('backend', BACKENDS)
def test_rays_are_independent_of_batching(backend):
    """Size-independent property: rendering a subset / a permutation of the rays gives bit-identical pixels."""
    cfg, que, ref, out, mid, extra, weights, full, (r, tq, tr) = run_case('a_small', backend)
    rn = que['coords'].shape[1]
    perm = np.random.RandomState(0).permutation(rn)[:17]
    tq2 = {k: v for k, v in tq.items() if not k.startswith('_')}
    tq2['coords'] = tq['coords'][:, torch.from_numpy(perm).to(tq['coords'].device)]
    with torch.no_grad():
        sub = r.render_impl(tq2, tr, False)
    for k in ('pixel_colors_nr', 'pixel_colors_nr_fine', 'hit_prob_nr_fine'):
        assert np.array_equal(sub[k].cpu().numpy(), full[k][:, perm]), k
    # Sum of hit probabilities never exceeds 1 (compositing invariant)
    assert np.all(full['hit_prob_nr'].sum(-1) <= 1 + 1e-5) and np.all(full['hit_prob_nr_fine'].sum(-1) <= 1 + 1e-5)


@pytest.mark.parametrize('backend', BACKENDS)
def test_render_loop_drops_hit_prob_and_concats(backend):
    """NeuralRayBaseRenderer.render: ray-batch loop, eval drops hit_prob* keys (renderer.py:241-252)."""
    cfg, que, ref, out, mid, extra = load_case('a_small')
    r, dev = make_renderer({**cfg, 'ray_batch_num': 16, 'hip_min_ray_batch': 0}, load_weights(False), backend)      # (exactly 16 rays per launch: no merging)
    tq, tr = to_torch(que, dev), to_torch(ref, dev)
    with torch.no_grad():
        got = r.render(tq, tr, False)
    assert not any(k.startswith('hit_prob') for k in got)
    assert got['pixel_colors_nr_fine'].shape == (1, que['coords'].shape[1], 3)
    err = np.max(np.abs(got['pixel_colors_nr'].cpu().numpy() - out['pixel_colors_nr']))
    assert err <= TOL_PIXEL


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('vpw', [1, 2])
@pytest.mark.parametrize('name', ['a_small', 'd_train_vis', 'e_use_all'])
def test_both_wave_decompositions(name, vpw, backend):
    """views_per_wave = 1 and 2 (odd view counts pad a masked view; rfn 2, 3, 5 exercise OWN = 2 / 4 tiles per
    wave) give the same result up to the summation order of the cross-view reductions."""
    cfg, que, ref, out, mid, extra = load_case(name)
    r, dev = make_renderer(cfg, load_weights(case_uses_vis_weights(name)), backend)
    r.engine(dev).views_per_wave = vpw
    torch.manual_seed(1234)
    with torch.no_grad():
        got = r.render_impl(to_torch(que, dev), to_torch(ref, dev), extra['is_train'])
    assert np.max(np.abs(got['pixel_colors_nr'].cpu().numpy() - out['pixel_colors_nr'])) <= TOL_PIXEL
    assert np.max(np.abs(got['hit_prob_nr'].cpu().numpy() - out['hit_prob_nr'])) <= TOL_HIT
    assert np.array_equal(got['ray_mask'].cpu().numpy(), out['ray_mask'])


@pytest.mark.parametrize('backend', BACKENDS)
def test_single_reference_view(backend):
    """rfn = 1: one wave owns all four tiles of the per-point layers."""
    cfg, que, ref, out, mid, extra = load_case('a_small')
    weights = load_weights(False)
    ref1 = {k: v[:1] for k, v in ref.items()}
    r, dev = make_renderer(cfg, weights, backend)
    with torch.no_grad():
        got = r.render_impl(to_torch(que, dev), to_torch(ref1, dev), False)
    want = orc.render_impl(weights, oracle_cfg(cfg), que, ref1)
    assert np.max(np.abs(got['pixel_colors_nr'].cpu().numpy() - want['pixel_colors_nr'])) <= 1e-5
    assert np.max(np.abs(got['hit_prob_nr'].cpu().numpy() - want['hit_prob_nr'])) <= 1e-5


def test_per_dict_caches_follow_their_tensors():
    """the view / query constant blocks cached inside the imgs_info dicts are rebuilt when a source tensor is replaced or
    modified in place (they are keyed by tensor identity + in-place version, with the entry holding the references)"""
    cfg, que, ref, out, mid, extra, weights, full, (r, tq, tr) = run_case('a_small', 'emu')
    with torch.no_grad():
        a = r.render_impl(tq, tr, False)['pixel_colors_nr'].clone()
        assert torch.equal(r.render_impl(tq, tr, False)['pixel_colors_nr'], a)             # cached blocks reused
        views_before = tr['_neuray_views'][1]
        tq['poses'][0, 0, 3] += 0.05                                                        # in place: version bump
        b = r.render_impl(tq, tr, False)['pixel_colors_nr'].clone()
        assert not torch.equal(a, b) and tr['_neuray_views'][1] is views_before            # only the query block was rebuilt
        tq['poses'][0, 0, 3] -= 0.05
        tr['ray_feats'] = tr['ray_feats'] * 1.0 + 0.25                                      # replaced by a new tensor
        c = r.render_impl(tq, tr, False)['pixel_colors_nr']
        assert not torch.equal(a, c) and tr['_neuray_views'][1] is not views_before
