"""GPU diagnostic: full-size scene, strided ray sample, every stage against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import bench
from oracle import neuray_oracle as orc
dev = torch.device('cuda', 0)
cfg, renderer, weights, que, ref, tq, tr = bench.build_case(dev, 32, seed=0)
eng = renderer.engine(dev)
idx = np.linspace(0, 800 * 800 - 1, 256).astype(np.int64)
q = dict(que); q['coords'] = que['coords'][:, idx]
c = {**orc.DEFAULT_CFG, **dict(cfg, coarse_use_vis=False, fine_use_vis=True)}
od = orc.sample_depth(q['depth_range'], len(idx), 64)
o, aux = orc.render_by_depth(weights, c, od, q, ref, False, False, return_aux=True)
prj = aux['prj']
qc = eng.prepare_query(tq); views = eng.prepare_views(tr); pk = renderer._packed_pass(eng, False)
coords = tq['coords'][0][torch.from_numpy(idx).to(dev)]
d = eng.sample_coarse_depth(tq['depth_range'], len(idx), 64)
res = eng.render_pass(qc, views, coords, d, pk, use_vis=False, want_depth=True, want_density=True, want_dbg=True)
torch.cuda.synchronize()
res = {k: v.cpu().numpy() for k, v in res.items()}
tr_ = lambda t: t[:, 0].transpose(1, 2, 0)
dbg = res['dbg']
names = {0: ('mask', prj['mask'][..., 0]), 1: ('u', prj['pts'][..., 0]), 2: ('v', prj['pts'][..., 1]), 3: ('z', prj['depth'][..., 0]),
         4: ('hit', prj['hit_prob'][..., 0]), 5: ('vis', prj['vis'][..., 0]), 6: ('mu0', prj['_mean'][..., 0]), 7: ('mu1', prj['_mean'][..., 1]),
         8: ('s0', prj['_var'][..., 0]), 9: ('s1', prj['_var'][..., 1]), 10: ('aw', prj['_aw'][..., 0])}
for i, (n, want) in names.items():
    e = np.abs(dbg[..., i] - tr_(want))
    print('%-5s maxdiff %.3e  #>1e-4: %d / %d   per-view max: %s' % (n, e.max(), (e > 1e-4).sum(), e.size, np.array2string(e.max((0, 1)), precision=2)))
rec = res['point_rec']
for n, got, want in (('geo', rec[..., :16], aux['geo_feat']), ('rgb', rec[..., 16:19], aux['colors'][0]), ('nvalid', rec[..., 19], aux['num_valid']),
                     ('density', res['density'], aux['density'][0]), ('hit_nr', res['hit_prob'], o['hit_prob_nr'][0]), ('pixel', res['pixel'], o['pixel_colors_nr'][0])):
    e = np.abs(got - want); print('%-8s maxdiff %.3e #>1e-3: %d/%d' % (n, e.max(), (e > 1e-3).sum(), e.size))
e = np.abs(rec[..., :16] - aux['geo_feat']).max(-1)
bad = np.argwhere(e > 1e-3)
print('bad geo points (ray, sample) first 20:', bad[:20].tolist(), ' total', len(bad))
if len(bad):
    r_, s_ = bad[0]
    print('nvalid there', rec[r_, s_, 19], 'masks', dbg[r_, s_, :, 0], 'vis2', dbg[r_, s_, :, 14], 'zlogit', dbg[r_, s_, :, 15])
