"""GPU diagnostic (not a test): run-to-run and placement determinism of the render path, stage by stage."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tests'))
import numpy as np, torch
from conftest import load_case, load_weights, oracle_cfg
from neuray_amd.network.renderer import NeuralRayBaseRenderer
from oracle import neuray_oracle as orc

cfg, que, ref, out, mid, extra = load_case('a_small')
w = load_weights(False)
r = NeuralRayBaseRenderer(cfg); r.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}); r = r.eval().cuda()
dev = 'cuda:0'
eng = r.engine(dev)
tq = {k: torch.from_numpy(v).to(dev) for k, v in que.items()}
tr = {k: torch.from_numpy(v).to(dev) for k, v in ref.items()}
c = {**orc.DEFAULT_CFG, **oracle_cfg(cfg)}
rn, dn = que['coords'].shape[1], c['depth_sample_num']
qc = eng.prepare_query(tq); views = eng.prepare_views(tr); pk = r._packed_pass(eng, False)

def run(coords):
    d = eng.sample_coarse_depth(tq['depth_range'], coords.shape[0], dn)
    res = eng.render_pass(qc, views, coords, d, pk, use_vis=False, want_depth=True, want_density=True, want_dbg=True)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in res.items()}

a = run(tq['coords'][0]); b = run(tq['coords'][0])
print('run-to-run identical:', {k: bool(np.array_equal(a[k], b[k])) for k in a})
perm = np.random.RandomState(0).permutation(rn)[:17]
s = run(tq['coords'][0][torch.from_numpy(perm).to(dev)])
names = ['mask','u','v','z','hit','vis','mu0','mu1','s0','s1','aw','nu','sn','visp','vis2','zlogit']
for i, n in enumerate(names):
    d = np.abs(s['dbg'][..., i] - a['dbg'][perm][..., i])
    print('dbg %-7s maxdiff %.3e  nonzero %d/%d' % (n, d.max(), (d > 0).sum(), d.size))
for k in ('point_rec', 'density', 'hit_prob', 'pixel'):
    d = np.abs(s[k] - a[k][perm]); print('%-10s maxdiff %.3e nonzero %d/%d' % (k, d.max(), (d > 0).sum(), d.size))
d = np.abs(s['point_rec'] - a['point_rec'][perm]).max((1, 2))
print('rays with diffs (sub index -> full index, parity of full/sub slot):', [(int(j), int(perm[j]), int(perm[j]) % 2, j % 2) for j in np.nonzero(d > 0)[0]])
# vs oracle geometry exactness
o, aux = orc.render_by_depth(w, c, orc.sample_depth(que['depth_range'], rn, dn), que, ref, False, False, return_aux=True)
tr_ = lambda t: t[:, 0].transpose(1, 2, 0)
print('u exact vs oracle:', np.array_equal(a['dbg'][..., 1], tr_(aux['prj']['pts'][..., 0])), ' z exact:', np.array_equal(a['dbg'][..., 3], tr_(aux['prj']['depth'][..., 0])))
print('max u diff', np.abs(a['dbg'][..., 1] - tr_(aux['prj']['pts'][..., 0])).max())
