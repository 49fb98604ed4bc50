"""Host-side profile (cProfile) of NeuralRayFtRenderer.train_step + backward + Adam on a 24-view 800 x 800 in-memory scene:
where the ~45 ms of a fine-tuning step go on the host (the GPU work of the step is ~20 ms).   python tools/profile_ft_step.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuray_amd import pipeline, synthetic                          # noqa: E402
from neuray_amd.network.renderer import NeuralRayFtRenderer         # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    db = synthetic.MemoryDatabase(24, 800, 800, seed=0)
    scene = {'ref_imgs_info': pipeline.build_imgs_info(db, db.get_img_ids(), -1, True, False, True, True)}
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'use_self_hit_prob': True, 'use_validation': False,
           'train_ray_num': 512}
    torch.manual_seed(0); np.random.seed(0)
    ft = NeuralRayFtRenderer(cfg, scene=scene).train().to(dev)
    opt = torch.optim.Adam(ft.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        out = ft.train_step()
        loss = ((out['pixel_colors_nr'] - out['pixel_colors_gt']) ** 2).mean() + ((out['pixel_colors_nr_fine'] - out['pixel_colors_gt']) ** 2).mean() + \
            out['hit_prob_self'].mean() + out['hit_prob_self_fine'].mean()
        loss.backward()
        opt.step()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('10 steps: host %.1f ms/step, with drain %.1f ms/step' % (100 * (t1 - t0), 100 * (t2 - t0)))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)


def classify(path, steps):
    """rocprofv3 kernel_stats.csv of this script's 25 steps -> GPU milliseconds per step by kernel class"""
    import csv
    classes = [('per-ray path (nr:: kernels)', ('nr::',)), ('fused norm (nr::inorm)', ('inorm',)),
               ('MIOpen convolutions', ('conv', 'Conv', 'igemm', 'gemm', 'Sp3', 'winograd', 'Winograd', 'xform', 'naive_conv', 'Cijk')),
               ('MIOpen transposes / batched copies', ('transpose', 'Transpose', 'batched_transpose')),
               ('up-sampling', ('upsample', 'bilinear')), ('optimizer (multi_tensor)', ('multi_tensor',)),
               ('indexing / gathers / cat / pad', ('index', 'Index', 'gather', 'Cat', 'cat_', 'pad', 'Pad')),
               ('reductions', ('reduce', 'Reduce')), ('element-wise / copies / fills', ('elementwise', 'Elementwise', 'copy', 'Copy', 'fill', 'Fill'))]
    rows = list(csv.DictReader(open(path)))
    tot = {}
    for r in rows:
        name, ms, calls = r['Name'], float(r['TotalDurationNs']) / 1e6, int(r['Calls'])
        cls = 'other'
        if 'inorm' in name:
            cls = 'fused norm (nr::inorm)'
        elif 'nr::conv2d_x3' in name:
            cls = 'encoder 3x3 convolutions (nr::conv2d_x3: forward + data gradient + packs)'
        elif 'nr::upsample2x' in name:
            cls = 'up-sampling'
        elif any(k in name for k in ('nr::costreg', 'nr::warp_variance', 'nr::diff_feats', 'nr::conv3d_kernel', 'nr::scale_shift_leaky')):
            cls = 'init net (nr:: cost volume / consistency kernels)'
        else:
            for c, keys in classes:
                if any(k in name for k in keys):
                    cls = c
                    break
        t = tot.setdefault(cls, [0.0, 0])
        t[0] += ms
        t[1] += calls
    whole = sum(v[0] for v in tot.values())
    print('GPU kernel time per step (%d steps in the trace): %.2f ms in %d launches' % (steps, whole / steps, sum(v[1] for v in tot.values()) // steps))
    for c, v in sorted(tot.items(), key=lambda kv: -kv[1][0]):
        print('  %-40s %7.2f ms  %5.1f %%  %5d launches' % (c, v[0] / steps, 100 * v[0] / whole, v[1] // steps))


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--classify':
        classify(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 25)
    else:
        main()
