"""Time the point backward kernel alone (HIP events): 512 rays x 8 views x 64 samples (the training shape).
    python tools/time_bwd.py [--reps 20]        (NEURAY_HIP_LIB=<other .so> for A/B builds)"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuray_amd import synthetic                                   # noqa: E402
from neuray_amd.engine import RenderEngine                         # noqa: E402
from neuray_amd.network.renderer import NeuralRayBaseRenderer      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--rays', type=int, default=512)
    ap.add_argument('--hw', type=int, nargs=2, default=[400, 600], help='image size (the feature maps are 8 x h x w x 32 floats each)')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    r = NeuralRayBaseRenderer({'use_hierarchical_sampling': False, 'dist_decoder_cfg': {'use_vis': False}})
    sd = {k: v for k, v in r.state_dict().items()}
    eng = RenderEngine(dev)
    que, ref = synthetic.make_scene(args.hw[0], args.hw[1], 8, seed=0)
    rng = np.random.RandomState(0)
    coords = torch.from_numpy((rng.rand(args.rays, 2) * np.array([args.hw[1] - 1, args.hw[0] - 1])).astype(np.float32)).to(dev)
    t = lambda a: torch.from_numpy(a).to(dev)                      # noqa: E731
    views = eng.prepare_views({k: t(v) for k, v in ref.items()})
    qc = eng.prepare_query({k: t(v) for k, v in que.items()})
    depth = eng.sample_coarse_depth(t(que['depth_range']), args.rays, 64)
    flat, has_vis = eng.flat_pass(sd, 'dist_decoder.', 'agg_net.')
    packed = eng.pack_pass_device(flat, has_vis)
    d_rec = torch.randn(args.rays, 64, 20, device=dev) * 1e-2
    saved = eng.render_points_saved(qc, views, coords, depth, packed, False)       # what the training forward leaves for the backward
    run = lambda: eng.render_points_backward(qc, views, coords, depth, flat, has_vis, False, d_rec, packed=packed, saved=saved)   # noqa: E731
    run(); run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        e0.record(); out = run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(json.dumps({'lib': os.environ.get('NEURAY_HIP_LIB', 'product'), 'rays': args.rays,
                      'ms_min': min(ts), 'ms_median': float(np.median(ts)), 'd_flat_abs_sum': float(out[0].abs().sum())}))


if __name__ == '__main__':
    main()
