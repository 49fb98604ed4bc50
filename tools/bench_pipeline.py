"""End-to-end (host buffers in, uint8 image out) rate of the generalisation renderer on the lego-800 shape (SURVEY.md 8(f)
f-4): the reference-style loop - build_imgs_info on the host, fp32 upload of the 8 working views for every pose, fp32
copy back - next to the device-resident view cache of neuray_amd/pipeline.py (uint8 upload once per view, uint8 copy back).
    python tools/bench_pipeline.py [--poses 6] [--views 24]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuray_amd import pipeline, synthetic                      # noqa: E402
from neuray_amd.network import renderer as R                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--poses', type=int, default=6)
    ap.add_argument('--views', type=int, default=24)
    ap.add_argument('--size', type=int, default=800)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    h = w = a.size
    db = synthetic.MemoryDatabase(a.views, h, w, seed=0)
    gen = R.NeuralRayGenRenderer({'use_hierarchical_sampling': True, 'fine_depth_sample_num': 32, 'fine_agg_net_cfg': {'sample_num': 32},
                                  'ray_batch_num': 32768, 'init_net_type': 'depth', 'dist_decoder_cfg': {'use_vis': False}}).eval().to(dev)
    qposes = np.stack([synthetic.look_at_pose(synthetic.sphere_pos(4.03, 360.0 * (i + 0.37) / a.views, 27.0)) for i in range(a.poses)]).astype(np.float32)
    ref_ids = pipeline.select_working_views_db(db, None, qposes, 8, False)
    K, shape, dr = [db.get_K(0)] * a.poses, [(h, w)] * a.poses, [(2.0, 6.0)] * a.poses

    def reference_style():
        up = 0
        for qi in range(a.poses):
            info = pipeline.build_imgs_info(db, list(ref_ids[qi]), 16, True, False, True, True)
            up += sum(v.nbytes for v in info.values())
            ref = {k: torch.from_numpy(v).to(dev) for k, v in info.items()}
            que = pipeline.build_render_imgs_info(qposes[qi], K[qi], shape[qi], dr[qi])
            que.pop('shape')
            with torch.no_grad():
                out = gen({'que_imgs_info': {k: torch.from_numpy(v).to(dev) for k, v in que.items()}, 'ref_imgs_info': ref, 'eval': True})
            pipeline.color_map_backward(out['pixel_colors_nr_fine'].cpu().numpy().reshape(h, w, 3))
        return up

    cache = pipeline.DeviceViewCache(db, dev, pad_interval=16)

    def cached():
        pipeline.render_poses(gen, db, qposes, K, shape, dr, ref_ids, cache=cache, save_fn=lambda qi, im: None)

    def timeit(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r
    reference_style()                      # warm-up (MIOpen algorithm search, allocator)
    t_ref, up_ref = timeit(reference_style)
    t_first, _ = timeit(cached)            # uploads every distinct working view once
    up_first = cache.uploaded_bytes
    t_warm, _ = timeit(cached)             # every view resident
    n = a.poses
    print(json.dumps({
        'poses': n, 'scene_views': a.views, 'image': '%dx%d, 8 working views, 64+32 samples' % (h, w),
        'reference_style': {'images_per_s': n / t_ref, 'rays_per_s': n * h * w / t_ref, 'h2d_MB_per_image': up_ref / n / 1e6, 'd2h_MB_per_image': h * w * 12 / 1e6},
        'cached_first_pass': {'images_per_s': n / t_first, 'rays_per_s': n * h * w / t_first, 'h2d_MB_per_image': up_first / n / 1e6, 'd2h_MB_per_image': h * w * 3 / 1e6},
        'cached_resident': {'images_per_s': n / t_warm, 'rays_per_s': n * h * w / t_warm, 'h2d_MB_per_image': (cache.uploaded_bytes - up_first) / n / 1e6, 'd2h_MB_per_image': h * w * 3 / 1e6}}))


if __name__ == '__main__':
    main()
