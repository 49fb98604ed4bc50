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


if __name__ == '__main__':
    main()
