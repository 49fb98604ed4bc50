"""Quick A/B of the two kernel libraries on the bench workload (800 x 800, 8 views, 64+32), no baselines:
    python tools/bench_variants.py [--steps 3]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                     # noqa: E402
from neuray_amd import synthetic                                  # noqa: E402
from neuray_amd.network.renderer import NeuralRayBaseRenderer    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg, r32, weights, que, ref, tq, tr = bench.build_case(dev, 32, seed=0)
    torch.manual_seed(0)
    r16 = NeuralRayBaseRenderer({**cfg, 'hip_variant': 'bf16'}).eval().to(dev)
    res, pix = {}, {}
    for tag, r in (('fp32', r32), ('bf16', r16)):
        eng = r.engine(dev)
        out = bench.render_image(r, tq, tr)
        eng.timing = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = bench.render_image(r, tq, tr)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pts = [e0.elapsed_time(e1) for name, e0, e1, n in eng.timing if name == 'points']
        eng.timing = None
        pix[tag] = out['pixel_colors_nr_fine'].cpu().numpy()
        res[tag] = {'rays_per_s': a.steps * 640000 / dt, 'point_kernel_ms': float(np.mean(pts))}
    res['bf16']['psnr_vs_fp32_db'] = synthetic.psnr_uint8(np.clip(pix['bf16'], 0, 1), np.clip(pix['fp32'], 0, 1))
    print(json.dumps(res))


if __name__ == '__main__':
    main()
