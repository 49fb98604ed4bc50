"""Per-parameter gradient error of the config-4-shape training step against the reference's autograd
(tests/golden/case_c4_train.npz), on the MI355X or (--emu) the CPU emulator.  Prints the worst tensors."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_baseline_shapes import load_tile, renderer_for  # noqa: E402


def main():
    backend = 'emu' if '--emu' in sys.argv else 'hip'
    z, cfg, que, ref, want, mid = load_tile('c4_train')
    r, dev = renderer_for(cfg, backend, train=True)
    lib = [a.split('=', 1)[1] for a in sys.argv if a.startswith('--lib=')]
    if lib:
        from neuray_amd import _lib
        r._engine_test_lib = _lib.bind(os.path.abspath(lib[0]))
    tq = {k: torch.from_numpy(v).to(dev) for k, v in que.items()}
    tq['coords'] = torch.from_numpy(z['coords']).to(dev)
    tr = {k: torch.from_numpy(v).to(dev) for k, v in ref.items()}
    for t in (tr['ray_feats'], tr['img_feats'], tq['ray_feats']):
        t.requires_grad_(True)
    out = r.render_by_depth(torch.from_numpy(mid['coarse_depth']).to(dev), tq, tr, True, False)
    for k, v in r.render_by_depth(torch.from_numpy(mid['fine_depth']).to(dev), tq, tr, True, True).items():
        out[k + '_fine'] = v
    gt = out['pixel_colors_gt'].detach()
    loss = ((out['pixel_colors_nr'] - gt) ** 2).mean() + ((out['pixel_colors_nr_fine'] - gt) ** 2).mean()
    for sfx in ('', '_fine'):
        p, q = out['hit_prob_nr' + sfx].detach(), out['hit_prob_self' + sfx]
        loss = loss + 0.1 * torch.nn.functional.binary_cross_entropy(q.clamp(1e-4, 1 - 1e-4), p.clamp(0, 1))
    print('loss %.7f (reference %.7f)' % (float(loss.detach()), float(z['loss'])))
    loss.backward()
    rows = []
    for k, p in r.named_parameters():
        g = z['grad.' + k]
        got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(g)
        rows.append((np.abs(got - g).max() / max(np.abs(g).max(), 1e-7), k, float(np.abs(g).max()), float(np.abs(got - g).max())))
    rows.sort(reverse=True)
    for rel, k, gm, err in rows[:12]:
        print('%-55s rel %.2e  |g|max %.2e  err %.2e' % (k, rel, gm, err))


if __name__ == '__main__':
    main()
