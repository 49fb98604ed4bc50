"""Training-step timing (not the contract bench): one render_impl(is_train=True) + backward on 512 rays, 8 reference
views, 64 + 64 samples (the shape of BASELINE.json configs[3]/[4]) through the HIP forward + backward kernels.
The baseline beside it (the eager-PyTorch port of the reference's op sequence on the same GPU) is bench.py's
`training_step` leg - this tool never touches oracle/.      python tools/bench_train.py [--rays 512] [--steps 5]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuray_amd import synthetic                                   # noqa: E402
from neuray_amd.network.renderer import NeuralRayBaseRenderer      # noqa: E402


def ft_step(args, dev):
    from neuray_amd import pipeline
    from neuray_amd.network.renderer import NeuralRayFtRenderer
    db = synthetic.MemoryDatabase(24, 800, 800, seed=0)
    scene = {'ref_imgs_info': pipeline.build_imgs_info(db, db.get_img_ids(), -1, True, False, True, True)}
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'use_self_hit_prob': True, 'use_validation': False,
           'train_ray_num': args.rays}
    ft = NeuralRayFtRenderer(cfg, scene=scene).train().to(dev)
    opt = torch.optim.Adam(ft.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        out = ft.train_step()
        loss = ((out['pixel_colors_nr'] - out['pixel_colors_gt']) ** 2).mean() + ((out['pixel_colors_nr_fine'] - out['pixel_colors_gt']) ** 2).mean() + \
            out['hit_prob_self'].mean() + out['hit_prob_self_fine'].mean()
        loss.backward()
        opt.step()
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    print(json.dumps({'what': 'NeuralRayFtRenderer.train_step + backward + Adam, %d rays, 8 of 24 views of 800 x 800, 64+64 samples' % args.rays,
                      'ms_per_step': 1e3 * (time.perf_counter() - t0) / args.steps}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, default=512)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--ft', action='store_true', help='a NeuralRayFtRenderer.train_step on a 24-view 800 x 800 in-memory scene '
                                                      '(per-view learnable ray_feats, encoders trained) instead of the bare render_impl step')
    ap.add_argument('--use-all', action='store_true', help="cfg fine_depth_use_all: the fine pass runs on 64 + 64 = 128 samples per ray (rays_backward_kernel<2>)")
    ap.add_argument('--variant', default='fp32', help="'bf16x3': the split library (hi + lo bf16 MFMA operands, fp32 accumulate)")
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    if args.ft:
        return ft_step(args, dev)
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': 64,
           'fine_depth_sample_num': 64, 'agg_net_cfg': {'sample_num': 64}, 'fine_agg_net_cfg': {'sample_num': 64},
           'use_self_hit_prob': True, 'hip_variant': args.variant}
    if args.use_all:
        cfg.update(fine_depth_use_all=True, fine_agg_net_cfg={'sample_num': 128})
    torch.manual_seed(0)
    r = NeuralRayBaseRenderer(cfg).train().to(dev)
    que, ref = synthetic.make_scene(400, 600, 8, seed=0, que_imgs=True)
    rng = np.random.RandomState(0)
    que['coords'] = (rng.rand(1, args.rays, 2) * np.array([599, 399])).astype(np.float32)
    tq = {k: torch.from_numpy(v).to(dev) for k, v in que.items()}
    tr = {k: torch.from_numpy(v).to(dev) for k, v in ref.items()}
    tr['ray_feats'].requires_grad_(True); tr['img_feats'].requires_grad_(True); tq['ray_feats'].requires_grad_(True)
    tgt = torch.rand(1, args.rays, 3, device=dev)

    def step_ours():
        r.zero_grad(set_to_none=True)
        out = r.render_impl(tq, tr, True)
        loss = ((out['pixel_colors_nr'] - tgt) ** 2).mean() + ((out['pixel_colors_nr_fine'] - tgt) ** 2).mean() + \
            out['hit_prob_self'].mean() + out['hit_prob_self_fine'].mean()
        loss.backward()
        return loss

    def timeit(fn):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps

    res = {'rays': args.rays, 'views': 8, 'samples': '64+64', 'hip_ms_per_step': 1e3 * timeit(step_ours)}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
