"""Timing of the depth init net front end on the lego-800 shape: 8 views of 800 x 800.
    python tools/bench_init.py [--views 8] [--size 800]
get_diff_feats is one HIP kernel (neuray_diff_feats): rfn * h * w pixels x rfn projections, 4 taps of 16 bytes each."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuray_amd import synthetic                      # noqa: E402
from neuray_amd.network import init_net               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--views', type=int, default=8)
    ap.add_argument('--size', type=int, default=800)
    ap.add_argument('--reps', type=int, default=20)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    h = w = a.size
    _, ref = synthetic.make_scene(h, w, a.views, seed=0)
    info = {k: torch.from_numpy(ref[k]).to(dev) for k in ('imgs', 'poses', 'Ks', 'depth_range')}
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    depth = np.stack([3.6 + 0.7 * np.sin(xx / 90.0 + v) * np.cos(yy / 70.0 - v) for v in range(a.views)])[:, None].astype(np.float32)
    info['depth'] = torch.from_numpy(depth).to(dev)
    net = init_net.DepthInitNet({}).eval().to(dev)

    def timeit(fn):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / a.reps
    with torch.no_grad():
        dn = init_net.extract_depth_for_init(info)
        t_diff = timeit(lambda: init_net.get_diff_feats(info, dn))
        t_net = timeit(lambda: net(info, None, False))
    # cost-volume init net, evaluation path: 800 x 800 is built at 640 x 640 -> 160 x 160 x 64 planes, 3 source views each
    cv = init_net.CostVolumeInitNet({}).eval().to(dev)
    info['nn_ids'] = torch.tensor([[(v + 1) % a.views, (v + 2) % a.views, (v + 3) % a.views] for v in range(a.views)], device=dev)
    with torch.no_grad():
        t_cv = timeit(lambda: cv(info, info, False)) if a.size in (800,) else None
        from neuray_amd.network import render_ops
        eng = render_ops.engine_for(dev)
        f = torch.randn(a.views, 32, 160, 160, device=dev)
        prj = init_net.construct_project_matrix(0.2, 0.2, info['Ks'], info['poses'])
        dv = init_net.get_depth_vals(info['depth_range'], 64)
        t_var = timeit(lambda: eng.warp_variance(f[:1], f, info['nn_ids'][:1], prj[:1], prj, dv[:1]))
    pairs = a.views * a.views * h * w
    print(json.dumps({'views': a.views, 'size': a.size, 'get_diff_feats_ms': t_diff, 'depth_init_net_ms': t_net,
                      'cost_volume_init_net_ms': t_cv, 'warp_variance_ms_per_ref_view_160x160x64x3src': t_var,
                      'warp_variance_out_GB': 32 * 64 * 160 * 160 * 4 / 1e9,
                      'projections': pairs, 'gather_demand_GB': pairs * 64 / 1e9,
                      'gather_demand_GBps': pairs * 64 / 1e9 / (t_diff * 1e-3)}))


if __name__ == '__main__':
    main()
