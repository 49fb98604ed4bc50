"""Evidence for the CHAINED coarse -> fine comparison.

Every stage on identical inputs is inside SURVEY.md 8(c)'s tolerances (tests/test_baseline_shapes.py).  The chained
output `pixel_colors_nr_fine` is not, for ~1-3 % of the rays of the white-noise synthetic scene - and neither is the fp32
REFERENCE against ITSELF evaluated in float64.  This file pins that down with committed fixtures instead of prose:

  (a) tests/golden/case_c2_*_f64.npz: the reference run in float64 on the C2 tiles (make_golden_full.py tile_case_f64).
      |ref32 - ref64| sets the noise floor of the chained quantity; |ours - ref64| has to be distributed like it.
  (b) attribution.  The inverse-CDF step s* = e_lo + (u - cdf_lo) / mass * w (render_ops.py:211-224) has condition number
      1 / mass: a cdf perturbation D moves a fine sample by <= 3 D / mass bins (first order; derivation in the test).  With
      empty-space bins holding ~1e-5 of the mass, coarse hit_prob differences of 1e-6..1e-5 (both well inside the 1e-4
      gate) move samples by 1e-3..1e-1 bins, and the white-noise maps turn a displaced sample into a different colour.
      The tests check (i) every fine sample of ours is displaced from the reference's by no more than that first-order
      bound computed from the MEASURED coarse hit_prob difference, (ii) every ray beyond 2e-4 has a displaced sample,
      (iii) rays whose samples all sit within 1e-4 bins of the reference's are within 2e-4.
      (Round 2's DESIGN.md attributed the outliers to the `denom < 1e-5 -> 1` branch alone; the float64 fixture shows that
      branch flips on 0 of the reference's own 17 outlier rays - it is the conditioning, of which the branch is the limit.)
  (c) the oracle against itself with uniform +-4e-6 noise on the coarse hit_prob: the same percentage of rays leaves 2e-4.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, load_weights, oracle_cfg
from test_baseline_shapes import load_tile, ray_err, renderer_for

BACKENDS = ['oracle', 'emu', pytest.param('hip', marks=pytest.mark.gpu)]
FDN = 32


def to_s(d, near, far):
    d = np.asarray(d, np.float64)
    return (1.0 / near - 1.0 / d) / (1.0 / near - 1.0 / far)


def cdf_of(hit, dtype):
    """render_ops.py:196-199 in `dtype`"""
    hp = hit.astype(dtype) + dtype(1e-5)
    pdf = hp / np.sum(hp, -1, keepdims=True, dtype=dtype)
    c = np.cumsum(pdf, -1, dtype=dtype)
    return np.concatenate([np.zeros_like(c[..., :1]), c], -1).astype(np.float64)


def neighbourhood_mass(cdf64, fdn):
    """smallest pdf mass among the bin a stratified sample falls in and its two neighbours (a sample within D of a knot
    may land next door)"""
    u = (0.5 + np.arange(fdn)) / fdn
    j = np.clip(np.stack([np.searchsorted(c, u, side='right') for c in cdf64]) - 1, 0, cdf64.shape[-1] - 2)
    pdf = np.diff(cdf64, axis=-1)
    rows = np.arange(len(j))[:, None]
    last = pdf.shape[-1] - 1
    return np.minimum(np.minimum(pdf[rows, np.clip(j - 1, 0, last)], pdf[rows, np.clip(j + 1, 0, last)]), pdf[rows, j])


def displacement_bound(hit_a, hit_b, fdn, dn):
    """first-order bound (in coarse bins) on |s*_a - s*_b| per fine sample: with t = (u - c_lo) / (c_hi - c_lo) and
    perturbed knots c' = c + d, |t' - t| <= (|d_lo| + t' |d_hi - d_lo|) / mass <= 3 D / mass, D = max_k |c_a - c_b|; the fp32
    evaluation of (u - c_lo) adds 2^-22 to D; 2e-5 bins of slack for the fp32 map s -> metric depth and back"""
    ca, cb = cdf_of(hit_a, np.float32), cdf_of(hit_b, np.float32 if hit_b.dtype == np.float32 else np.float64)
    D = np.abs(ca - cb).max(-1)
    mass = neighbourhood_mass(cb, fdn)
    return 3.0 * (D[:, None] + 2.0 ** -22) / np.maximum(mass, 1e-5) + 2e-5, D


def our_chain(name, backend, sel):
    """-> dict of our outputs on the tile's rays[sel]: coarse pixels / hit_prob, OUR fine depths, chained fine pixels"""
    z, cfg, que, ref, want, mid = load_tile(name)
    idx = np.arange(z['coords'].shape[1])[sel]
    if backend == 'oracle':
        from oracle import neuray_oracle as orc
        q = dict(que)
        q['coords'] = z['coords'][:, idx]
        res = orc.render_impl(load_weights(False), oracle_cfg({**orc.DEFAULT_CFG, **cfg}), q, ref)
        got = {k: res[k] for k in ('pixel_colors_nr', 'hit_prob_nr', 'pixel_colors_nr_fine')}
        got['fine_depth'] = res['_fine_depth']
    else:
        r, dev = renderer_for(cfg, backend)
        tq = {k: torch.from_numpy(v).to(dev) for k, v in que.items()}
        tq['coords'] = torch.from_numpy(z['coords'][:, idx]).to(dev)
        tr = {k: torch.from_numpy(v).to(dev) for k, v in ref.items()}
        seen = {}
        inner = r.render_by_depth

        def wrapped(que_depth, *a, **k):
            seen['fine_depth' if a[3] else 'coarse_depth'] = que_depth.detach().cpu().numpy()
            return inner(que_depth, *a, **k)
        r.render_by_depth = wrapped
        with torch.no_grad():
            out = r.render_impl(tq, tr, False)
        got = {k: out[k].cpu().numpy() for k in ('pixel_colors_nr', 'hit_prob_nr', 'pixel_colors_nr_fine')}
        got['fine_depth'] = seen['fine_depth']
    W = {k: v[:, idx] for k, v in want.items()}
    M = {k: v[:, idx] for k, v in mid.items()}
    f64 = np.load(os.path.join(GOLDEN_DIR, 'case_%s_f64.npz' % name))
    W64 = {k[4:]: f64[k][:, idx] for k in f64.files if k.startswith('out.')}
    M64 = {k[4:]: f64[k][:, idx] for k in f64.files if k.startswith('mid.')}
    return got, W, M, W64, M64, que


def quantiles(e):
    return {'median': float(np.median(e)), 'p90': float(np.percentile(e, 90)), 'p99': float(np.percentile(e, 99)), 'max': float(e.max()),
            'beyond_2e-4': float(np.mean(e > 2e-4))}


# ---- (a) the fp32 reference against its own float64 evaluation: the floor --------------------------------------------------
@pytest.mark.parametrize('name,lo,hi', [('c2_tile_32', 0.004, 0.04), ('c2_smooth', 0.0, 0.002)])
def test_fp32_reference_vs_float64_reference_is_the_noise_floor(name, lo, hi):
    a = np.load(os.path.join(GOLDEN_DIR, 'case_%s.npz' % name))
    b = np.load(os.path.join(GOLDEN_DIR, 'case_%s_f64.npz' % name))
    assert np.array_equal(a['ray_index'], b['ray_index'])
    coarse = ray_err(a['out.pixel_colors_nr'], b['out.pixel_colors_nr'])
    chained = ray_err(a['out.pixel_colors_nr_fine'], b['out.pixel_colors_nr_fine'])
    print('%s: ref32 vs ref64  coarse %s\n    chained %s' % (name, quantiles(coarse), quantiles(chained)))
    assert coarse.max() <= 1e-4 and np.abs(a['out.hit_prob_nr'] - b['out.hit_prob_nr']).max() <= 2e-4
    assert np.array_equal(a['out.ray_mask'], b['out.ray_mask']) and np.array_equal(a['out.ray_mask_fine'], b['out.ray_mask_fine'])
    # the reference in fp32 misses the 2e-4 gate against its own float64 evaluation on this share of the chained rays
    assert lo <= np.mean(chained > 2e-4) <= hi
    # ... and NOT because of the `denom < 1e-5 -> 1` branch (render_ops.py:218-219): it flips on none of those rays
    c32 = cdf_of(a['out.hit_prob_nr'][0], np.float32)
    c64 = cdf_of(b['out.hit_prob_nr'][0], np.float64)
    u = (0.5 + np.arange(FDN)) / FDN

    def masses(c):
        j = np.stack([np.searchsorted(row, u, side='right') for row in c])
        return np.take_along_axis(c, np.minimum(j, c.shape[-1] - 1), -1) - np.take_along_axis(c, np.maximum(j - 1, 0), -1)
    flips = ((masses(c32) < 1e-5) != (masses(c64) < 1e-5)).any(-1)
    assert np.sum(flips & (chained > 2e-4)) <= 0.2 * max(1, np.sum(chained > 2e-4))
    # it is the conditioning of the inverse CDF: every fine sample of the fp32 run is within the first-order bound of the f64 run's
    near, far = 2.0, 6.0
    disp = np.abs(to_s(a['mid.fine_depth'][0], near, far) - to_s(b['mid.fine_depth'][0], near, far)) * 63.0
    bound, D = displacement_bound(a['out.hit_prob_nr'][0], b['out.hit_prob_nr'][0], FDN, 64)
    assert np.mean(disp <= bound) >= 0.9995, float(np.mean(disp <= bound))
    worst = disp.max(-1)
    assert np.all(worst[chained > 2e-4] > 1e-4)                      # every outlier has a displaced sample
    assert chained[worst <= 1e-4].max(initial=0.0) <= 2e-4           # and undisplaced rays are inside the gate


# ---- (a) + (b) for OUR chain ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('name', ['c2_tile_32', 'c2_smooth'])
def test_our_chained_error_is_distributed_like_the_references_own_fp32_noise(name, backend):
    sel = slice(None) if backend == 'hip' else (slice(None, None, 16) if backend == 'emu' else slice(None, None, 4))
    got, W, M, W64, M64, que = our_chain(name, backend, sel)
    near, far = (float(x) for x in que['depth_range'][0])
    ours64 = quantiles(ray_err(got['pixel_colors_nr_fine'], W64['pixel_colors_nr_fine']))
    ref64 = quantiles(ray_err(W['pixel_colors_nr_fine'], W64['pixel_colors_nr_fine']))
    ours32 = quantiles(ray_err(got['pixel_colors_nr_fine'], W['pixel_colors_nr_fine']))
    c_ours64 = quantiles(ray_err(got['pixel_colors_nr'], W64['pixel_colors_nr']))
    c_ref64 = quantiles(ray_err(W['pixel_colors_nr'], W64['pixel_colors_nr']))
    print('%s[%s] chained  |ours-ref64| %s\n%s           |ref32-ref64| %s\n%s           |ours-ref32| %s\n    coarse   |ours-ref64| %s\n             |ref32-ref64| %s'
          % (name, backend, ours64, ' ' * len(name), ref64, ' ' * len(name), ours32, c_ours64, c_ref64))
    # coarse: we are as close to the float64 truth as the fp32 reference is (x2 + fp32 epsilon of slack)
    assert c_ours64['max'] <= 2.0 * c_ref64['max'] + 2e-5 and c_ours64['median'] <= 2.0 * c_ref64['median'] + 2e-6
    # chained: same distribution as the reference's own fp32-vs-float64 error
    assert ours64['median'] <= 2.0 * ref64['median'] + 2e-6
    assert ours64['p90'] <= 2.0 * ref64['p90'] + 5e-6
    assert ours64['beyond_2e-4'] <= 2.0 * ref64['beyond_2e-4'] + 0.01
    # attribution of ours-vs-ref32: displacement of OUR fine samples against the reference's, bounded by the conditioning
    disp = np.abs(to_s(got['fine_depth'][0], near, far) - to_s(M['fine_depth'][0], near, far)) * 63.0
    bound, D = displacement_bound(got['hit_prob_nr'][0], W['hit_prob_nr'][0], FDN, 64)
    err = ray_err(got['pixel_colors_nr_fine'], W['pixel_colors_nr_fine'])
    worst = disp.max(-1)
    print('    coarse cdf difference D: max %.2e median %.2e; samples inside the first-order bound: %.5f; outlier rays %d, all displaced: %s'
          % (D.max(), np.median(D), np.mean(disp <= bound), int(np.sum(err > 2e-4)), bool(np.all(worst[err > 2e-4] > 1e-4))))
    assert np.mean(disp <= bound) >= 0.9995
    assert np.all(worst[err > 2e-4] > 1e-4)
    assert err[worst <= 1e-4].max(initial=0.0) <= 2e-4


# ---- (c) the noise experiment DESIGN.md 2.4 quotes ----------------------------------------------------------------------
def test_oracle_against_itself_with_hit_prob_noise_at_the_fp32_level():
    """uniform +-4e-6 on the coarse hit_prob (25x below the 1e-4 gate) -> fine depths -> fine pass, oracle on both sides"""
    from oracle import neuray_oracle as orc
    z, cfg, que, ref, want, mid = load_tile('c2_tile_32')
    idx = np.arange(z['coords'].shape[1])[::2]
    q = dict(que)
    q['coords'] = z['coords'][:, idx]
    ocfg = oracle_cfg({**orc.DEFAULT_CFG, **cfg})
    weights = load_weights(False)
    hit = want['hit_prob_nr'][:, idx]
    noisy = np.clip(hit + np.random.RandomState(0).uniform(-4e-6, 4e-6, hit.shape).astype(np.float32), 0, None)
    pix = []
    for h in (hit, noisy):
        fd = np.sort(orc.sample_fine_depth(mid['coarse_depth'][:, idx], h, que['depth_range'], FDN), -1)
        pix.append(orc.render_by_depth(weights, ocfg, fd, q, ref, False, True)['pixel_colors_nr'])
    err = ray_err(pix[0], pix[1])
    print('oracle vs oracle + 4e-6 noise on hit_prob: %s' % quantiles(err))
    assert 0.003 <= np.mean(err > 2e-4) <= 0.06
