"""the per-image encoders on the training-step shape (9 images of 800 x 800: 8 reference views + the query view,
renderer.py:229-235) - image_encoder + vis_encoder, forward and forward + backward, HIP events.
    python tools/bench_encoders.py [--n 9] [--hw 800 800] [--reps 10] [--fused 0|1]
    rocprofv3 --kernel-trace --stats ... -- python tools/bench_encoders.py --reps 3      (profiles/collect_encoders.sh)"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuray_amd.network import encoders      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=9)
    ap.add_argument('--hw', type=int, nargs=2, default=[800, 800])
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--fused', type=int, default=-1, help='1 / 0: force the fused InstanceNorm kernels on / off (default: the module default)')
    ap.add_argument('--fwd-only', action='store_true', help='time only the forward (kernel traces of the forward alone)')
    ap.add_argument('--cprofile', action='store_true', help='host-side profile of 20 forwards')
    ap.add_argument('--benchmark', action='store_true', help='torch.backends.cudnn.benchmark = True (MIOpen find mode instead of the immediate heuristics)')
    ap.add_argument('--nchw', action='store_true', help='run the stacks in NCHW (encoders.CHANNELS_LAST = False)')
    a = ap.parse_args()
    if a.nchw:
        encoders.CHANNELS_LAST = False
    if a.benchmark:
        torch.backends.cudnn.benchmark = True
    dev = torch.device('cuda', 0)
    if a.fused >= 0 and hasattr(encoders, 'set_fused_norm'):
        encoders.set_fused_norm(bool(a.fused))
    torch.manual_seed(0)
    img_enc = encoders.ImageEncoder().to(dev)
    vis_enc = encoders.DefaultVisEncoder({}).to(dev)
    imgs = torch.rand(a.n, 3, a.hw[0], a.hw[1], device=dev)
    ray0 = torch.randn(a.n, 32, a.hw[0] // 4, a.hw[1] // 4, device=dev, requires_grad=True)

    def fwd():
        f = img_enc(imgs)
        return f, vis_enc(ray0, f)

    def fwd_bwd():
        f, r = fwd()
        (f.square().mean() + r.square().mean()).backward()
        for p in list(img_enc.parameters()) + list(vis_enc.parameters()) + [ray0]:
            p.grad = None

    def time(fn, grad):
        ctx = torch.enable_grad() if grad else torch.no_grad()
        with ctx:
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps
    if a.cprofile:
        import cProfile
        import pstats
        with torch.no_grad():
            fwd(); torch.cuda.synchronize()
            pr = cProfile.Profile(); pr.enable()
            for _ in range(20):
                fwd()
            pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats('tottime').print_stats(18)
    out = {'n': a.n, 'hw': a.hw, 'miopen_find': bool(a.benchmark), 'channels_last': encoders.CHANNELS_LAST, 'fused_norm': encoders.fused_norm.FUSED_NORM,
           'forward_ms': time(fwd, False), 'forward_backward_ms': None if a.fwd_only else time(fwd_bwd, True)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
