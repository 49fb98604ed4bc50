"""Which host-side operations launch the small kernels of a training step (copies / fills / adds)?  torch.profiler with python
stacks around one render_impl(is_train=True) + backward (tools/bench_train.py's step), grouped by operator and by the first
neuray_amd / script frame on the stack.     python tools/profile_train_ops.py [--emu]"""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuray_amd import synthetic                                   # noqa: E402
from neuray_amd.network.renderer import NeuralRayBaseRenderer      # noqa: E402


def main():
    emu = '--emu' in sys.argv
    dev = torch.device('cpu') if emu else torch.device('cuda', 0)
    rays, dn = (16, 8) if emu else (512, 64)
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': dn,
           'fine_depth_sample_num': dn, 'agg_net_cfg': {'sample_num': dn}, 'fine_agg_net_cfg': {'sample_num': dn}, 'use_self_hit_prob': True}
    torch.manual_seed(0)
    r = NeuralRayBaseRenderer(cfg).train()
    if emu:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from emu_util import emu_lib
        r._engine_test_lib = emu_lib()
    r = r.to(dev)
    h, w = (48, 64) if emu else (400, 600)
    que, ref = synthetic.make_scene(h, w, 8, seed=0, que_imgs=True)
    que['coords'] = (np.random.RandomState(0).rand(1, rays, 2) * np.array([w - 1, h - 1])).astype(np.float32)
    tq = {k: torch.from_numpy(v).to(dev) for k, v in que.items()}
    tr = {k: torch.from_numpy(v).to(dev) for k, v in ref.items()}
    for t in (tr['ray_feats'], tr['img_feats'], tq['ray_feats']):
        t.requires_grad_(True)
    tgt = torch.rand(1, rays, 3, device=dev)
    opt = torch.optim.Adam(r.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        out = r.render_impl(tq, tr, True)
        loss = ((out['pixel_colors_nr'] - tgt) ** 2).mean() + ((out['pixel_colors_nr_fine'] - tgt) ** 2).mean() + \
            out['hit_prob_self'].mean() + out['hit_prob_self_fine'].mean()
        loss.backward()
        if '--adam' in sys.argv:
            opt.step()
    for _ in range(3):
        step()
    from torch.profiler import ProfilerActivity, profile
    acts = [ProfilerActivity.CPU] + ([] if emu else [ProfilerActivity.CUDA])
    with profile(activities=acts, with_stack=True, experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        step()
        if not emu:
            torch.cuda.synchronize()
    by_op = collections.Counter()
    where = collections.defaultdict(collections.Counter)
    watch = ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::add_', 'aten::add', 'aten::mul', 'aten::index', 'aten::cat', 'aten::_to_copy')
    for e in prof.events():
        if e.name not in watch:
            continue
        if e.cpu_parent is not None and e.cpu_parent.name in ('aten::zero_', 'aten::_to_copy', 'aten::add'):
            continue                                     # counted at the parent
        by_op[e.name] += 1
        st = [s for s in (e.stack or []) if 'profiler' not in s]
        frames = [s for s in st if 'neuray_amd' in s or 'profile_train_ops' in s]
        frame = frames[0] if frames else ('autograd engine' if any('autograd' in s for s in st) else 'other')
        where[e.name][frame.strip()[:150]] += 1
    print('operators that launch a copy / fill / elementwise kernel, per step:', dict(by_op.most_common(20)))
    for op, _ in by_op.most_common(10):
        print(op)
        for fr, n in where[op].most_common(8):
            print('   %3d  %s' % (n, fr))


if __name__ == '__main__':
    main()
