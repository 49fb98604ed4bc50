"""The split variant of the kernels (libneuray_hip_bf16x3.so, cfg['hip_variant'] = 'bf16x3'):
same sources with -DNR_BF16_QUADS -DNR_BF16_SPLIT - every MFMA operand carried as hi + lo bf16 halves (x = hi + lo to 2^-16),
hi*hi + hi*lo + lo*hi as three bf16 MFMAs per fp32 quad, fp32 accumulation, fp32 everywhere else.  Never the default and never
the headline; it is only worth reporting if it passes the FP32 gates of the product path - 2e-4 on the pixels and 1e-4 on the hit
probabilities of every stage on identical inputs, against the REFERENCE's own outputs - which is what these tests demand."""
import numpy as np
import pytest
import torch

from conftest import load_case, load_weights
from emu_util import emu_lib_bf16x3, to_torch
from neuray_amd.network.renderer import NeuralRayBaseRenderer

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


def split_renderer(cfg, backend):
    r = NeuralRayBaseRenderer({**cfg, 'hip_variant': 'bf16x3'}).eval()
    r.load_state_dict({k: torch.from_numpy(v) for k, v in load_weights(False).items()}, strict=False)
    if backend == 'emu':
        r._engine_test_lib = emu_lib_bf16x3()
        return r, 'cpu'
    return r.cuda(), 'cuda:0'


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('name', ['a_small', 'b_default', 'c_adversarial'])
def test_split_variant_passes_the_fp32_gates_on_the_reference_goldens(name, backend):
    cfg, que, ref, out, mid, extra = load_case(name)
    r, dev = split_renderer(cfg, backend)
    assert r.engine(dev).lib.neuray_operand_precision() == 48
    with torch.no_grad():
        got = {k: v.cpu().numpy() for k, v in r.render_impl(to_torch(que, dev), to_torch(ref, dev), False).items()}
    ep = np.abs(got['pixel_colors_nr'] - out['pixel_colors_nr']).max()
    eh = np.abs(got['hit_prob_nr'] - out['hit_prob_nr']).max()
    print('%s[%s] split variant vs reference: coarse pixels %.2e (gate 2e-4), hit_prob %.2e (gate 1e-4)' % (name, backend, ep, eh))
    assert ep <= 2e-4 and eh <= 1e-4
    assert np.array_equal(got['ray_mask'], out['ray_mask'])
    ef = np.abs(got['pixel_colors_nr_fine'] - out['pixel_colors_nr_fine']).max(-1)
    assert np.mean(ef <= 2e-4) >= 0.9                                       # chained: DESIGN.md 2.4


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['c2_tile_32', 'c2_tile_64', 'c2_smooth', 'c1_tile', 'c3_tile'])
def test_split_variant_passes_the_fp32_gates_on_every_reference_tile(name):
    """stage-wise on identical inputs at BASELINE.json's shapes: coarse pass, fine pass on the reference's fine depths"""
    from test_baseline_shapes import load_tile, ray_err
    z, cfg, que, ref, want, mid = load_tile(name)
    r, dev = split_renderer(cfg, 'hip')
    tq = {k: torch.from_numpy(v).to(dev) for k, v in que.items()}
    tq['coords'] = torch.from_numpy(z['coords']).to(dev)
    tr = {k: torch.from_numpy(v).to(dev) for k, v in ref.items()}
    with torch.no_grad():
        coarse = r.render_by_depth(torch.from_numpy(mid['coarse_depth']).to(dev), tq, tr, False, False)
        fine = r.render_by_depth(torch.from_numpy(mid['fine_depth']).to(dev), tq, tr, False, True)
    res = {}
    for tag, got, sfx in (('coarse', coarse, ''), ('fine', fine, '_fine')):
        ep = ray_err(got['pixel_colors_nr'].cpu().numpy(), want['pixel_colors_nr' + sfx])
        eh = np.abs(got['hit_prob_nr'].cpu().numpy() - want['hit_prob_nr' + sfx]).max()
        res[tag] = (float(ep.max()), float(np.percentile(ep, 99)), float(eh))
        assert np.array_equal(got['ray_mask'].cpu().numpy(), want['ray_mask' + sfx])
    print('%s split variant vs reference (pixel max, pixel p99, hit_prob max): coarse %s fine %s' % (name, res['coarse'], res['fine']))
    for tag in res:
        assert res[tag][0] <= 2e-4 and res[tag][2] <= 1e-4, (name, tag, res[tag])


def test_split_packing_is_hi_plus_lo():
    """every quad slot holds bf16(w) in its first two dwords and bf16(w - bf16(w)) in the last two: hi + lo = w to 2^-16"""
    from emu_util import emu_lib
    from neuray_amd import _lib
    from neuray_amd.engine import RenderEngine
    w = load_weights(False)
    p32 = RenderEngine('cpu', _test_lib=emu_lib()).pack_pass(w, 'dist_decoder.', 'agg_net.').dev.numpy()
    p3 = RenderEngine('cpu', _test_lib=emu_lib_bf16x3()).pack_pass(w, 'dist_decoder.', 'agg_net.').dev.numpy()
    assert p32.shape == p3.shape
    # the first layer's quads: 2 tiles x 2 quads x 64 lanes x 4 floats
    n = 2 * 2 * 64 * 4
    q32 = p32[:n].reshape(-1, 4)
    bits = p3[:n].view(np.uint16).reshape(-1, 8)
    up = lambda h: (h.astype(np.uint32) << 16).view(np.float32)             # noqa: E731
    hi, lo = up(bits[:, :4]), up(bits[:, 4:])
    assert np.abs(hi + lo - q32).max() <= 2.0 ** -15 * np.abs(q32).max() and np.abs(lo).max() <= 2.0 ** -8 * np.abs(q32).max()
    assert np.abs(hi + lo - q32).max() < 0.02 * np.abs(hi - q32).max()      # two orders closer than bf16 alone
    del _lib


@pytest.mark.parametrize('backend', BACKENDS)
def test_split_variant_trains_gradients_match_the_references_autograd(backend):
    """BASELINE.json config 5 names bf16 training: the split library carries the whole training path (training forward with its saved
    quantities, ray / point / self-hit backward kernels on device-packed hi + lo weights, fp32 weight-gradient accumulation, fp32
    master weights).  Gate: the fp32 path's own - 5e-3 of each tensor's largest gradient entry against the REFERENCE's autograd
    (tests/golden/case_g_grads.npz)."""
    import os
    from conftest import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'case_g_grads.npz'))
    cfg = __import__('ast').literal_eval(str(z['cfg_json']))
    r, dev = split_renderer(cfg, backend)
    r.train()
    assert r.engine(dev).variant == 'bf16x3'
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
    assert abs(float(loss.detach()) - float(z['loss'])) <= 5e-3
    loss.backward()
    worst = 0.0
    for k, p_ in r.named_parameters():
        want = z['grad.' + k]
        g = p_.grad.cpu().numpy() if p_.grad is not None else np.zeros_like(want)
        scale = max(1e-3, float(np.abs(want).max()))
        err = float(np.abs(g - want).max()) / scale
        worst = max(worst, err)
        assert err <= 5e-3, (k, err)
    for t_, k in ((ref['ray_feats'], 'grad.ref.ray_feats'), (ref['img_feats'], 'grad.ref.img_feats'), (que['ray_feats'], 'grad.que.ray_feats')):
        assert np.abs(t_.grad.cpu().numpy() - z[k]).max() <= 5e-3 * np.abs(z[k]).max(), k
    print('split variant [%s]: worst relative parameter-gradient error vs the reference autograd %.2e' % (backend, worst))
