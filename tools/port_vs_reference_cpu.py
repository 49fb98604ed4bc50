"""Is `cpu_baseline`'s stand-in fair?  bench.py times oracle/torch_eager_port.py on the GPU box's host because the reference tree
does not travel there; this script - BUILD CONTAINER ONLY, where /root/reference exists - times the REFERENCE ITSELF
(`NeuralRayBaseRenderer.render_impl`, imported read-only through tests/golden/ref_harness.py) and the port on the same CPU, the same
thread count, the same weights and the same 4096-ray batches of the 800 x 800 / 8 views / 64 + 32 workload, and checks that both
produce the same pixels.

    python tools/port_vs_reference_cpu.py [--threads 8] [--batches 3]      -> one JSON line (profiles/r03_y_port_vs_reference_cpu.json)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=os.cpu_count())
    ap.add_argument('--batches', type=int, default=5)
    ap.add_argument('--rays', type=int, default=4096)
    a = ap.parse_args()
    import ref_harness
    from neuray_amd import synthetic
    from oracle import torch_eager_port as tep          # (a tool of the test infrastructure, like the bench's cpu_baseline leg)
    ns = ref_harness.import_reference()
    torch.set_num_threads(a.threads)
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': 64, 'fine_depth_sample_num': 32,
           'agg_net_cfg': {'sample_num': 64}, 'fine_agg_net_cfg': {'sample_num': 32}, 'ray_batch_num': a.rays}
    torch.manual_seed(0)
    ref = ns.renderer.NeuralRayBaseRenderer(cfg).eval()
    weights = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    que, views = synthetic.make_scene(800, 800, 8, seed=0)
    coords = synthetic.meshgrid_coords(800, 800)
    tq = {k: torch.from_numpy(v) for k, v in que.items()}
    tv = {k: torch.from_numpy(v) for k, v in views.items()}
    starts = np.linspace(0, coords.shape[1] - a.rays, a.batches + 1).astype(np.int64)
    ocfg = dict(cfg, coarse_use_vis=False, fine_use_vis=True)

    def run_ref(st):
        q = dict(tq)
        q['coords'] = torch.from_numpy(coords[:, st:st + a.rays])
        with torch.no_grad():
            return ref.render_impl(q, dict(tv), False)

    def run_port(st):
        q = dict(tq)
        q['coords'] = torch.from_numpy(coords[:, st:st + a.rays])
        with torch.no_grad():
            return tep.render_impl(weights, ocfg, q, tv)

    # interleaved (reference batch, port batch, reference batch, ...) so that a noisy neighbour on a shared host hits both alike;
    # rates from the median batch time
    run_ref(int(starts[0])); run_port(int(starts[0]))       # warm-up batch each
    t_ref, t_port = [], []
    for st in starts[1:]:
        t0 = time.perf_counter(); o_ref = run_ref(int(st)); t_ref.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); o_port = run_port(int(st)); t_port.append(time.perf_counter() - t0)
    r_ref, r_port = a.rays / float(np.median(t_ref)), a.rays / float(np.median(t_port))
    d = float((o_ref['pixel_colors_nr_fine'] - o_port['pixel_colors_nr_fine']).abs().max())
    print(json.dumps({'what': 'the reference itself vs the eager-PyTorch port of its op sequence, same CPU / threads / weights / batches; 800 x 800, 8 views, 64 + 32',
                      'threads': a.threads, 'batches_timed': a.batches, 'rays_per_batch': a.rays,
                      'reference_rays_per_s': r_ref, 'port_rays_per_s': r_port, 'batch_seconds_reference': t_ref, 'batch_seconds_port': t_port, 'port_over_reference': r_port / r_ref,
                      'max_abs_pixel_difference_last_batch': d, 'host': os.uname().nodename, 'cpus': os.cpu_count()}))


if __name__ == '__main__':
    main()
