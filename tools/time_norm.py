"""Fused InstanceNorm + activation + padding (csrc/nr_kernels_norm.h), forward and backward by HIP events, per library build:

    python tools/time_norm.py name=path/to/lib.so [name=build:-DSOME_FLAG ...]

Shapes: the per-image encoders' planes at the generalisation training resolution (9 x 32 x 104 x 152 ... 9 x 128 x 26 x 38), a
mid-size one and the fine-tuning step's 400 x 400 planes.  Prints us per forward and per backward (autograd node included: below
~100 us this is the host, not the kernels) and the HBM rate on the compulsory bytes (forward: x + out; backward: x, out, d_out + dx
[+ d_res]).  Used for the round-5 one-workgroup-per-plane experiment (profiles/r05_r_norm_ab.log; the kernels of that experiment are
profiles/r05_r_norm_one_workgroup_per_plane_experiment.diff, not in the tree)."""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from ab_forward import bind_compat, build_variant      # noqa: E402
from neuray_amd.network import fused_norm               # noqa: E402
from neuray_amd.network import render_ops as ro         # noqa: E402

SHAPES = [(9, 32, 104, 152, True), (9, 64, 52, 76, True), (9, 128, 26, 38, False), (9, 32, 150, 200, True), (2, 32, 104, 152, False),
          (9, 32, 400, 400, True), (9, 64, 200, 200, True), (9, 128, 100, 100, True), (9, 64, 400, 400, False)]


def main():
    dev = torch.device('cuda', 0)
    libs = []
    for a in sys.argv[1:]:
        name, path = a.split('=', 1)
        libs.append((name, build_variant(name, path[6:].split()) if path.startswith('build:') else path))
    rows = {}
    for name, path in libs:
        ro._TEST_LIB = bind_compat(path)
        ro._ENGINES.clear()
        for (n, c, h, w, with_res) in SHAPES:
            bn = nn.InstanceNorm2d(c, affine=True).to(dev)
            x = torch.randn(n, c, h, w, device=dev, requires_grad=True)
            res = torch.randn(n, c, h, w, device=dev, requires_grad=True) if with_res else None
            dz = torch.randn(n, c, h + 2, w + 2, device=dev)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tf = tb = 0.0
            reps = 30
            for it in range(reps + 5):
                ev[0].record()
                z = fused_norm.norm_act(bn, x, 'relu', 1, res=res)
                ev[1].record()
                z.backward(dz)
                ev[2].record()
                torch.cuda.synchronize()
                if it >= 5:
                    tf += ev[0].elapsed_time(ev[1])
                    tb += ev[1].elapsed_time(ev[2])
                x.grad = None
                if res is not None:
                    res.grad = None
            el = n * c * h * w * 4
            fb = el * (2 + (1 if with_res else 0))
            bb = el * (4 + (1 if with_res else 0))
            rows.setdefault((n, c, h, w, with_res), []).append(
                '%s fwd %.1f us (%.2f TB/s) bwd %.1f us (%.2f TB/s)' % (name, 1e3 * tf / reps, fb / (tf / reps * 1e-3) / 1e12,
                                                                          1e3 * tb / reps, bb / (tb / reps * 1e-3) / 1e12))
    for k, v in rows.items():
        print(k, ' | '.join(v))


if __name__ == '__main__':
    main()
