"""bench.py - rays/sec of the NeuRay per-ray render path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (coarse + fine render_impl over every ray batch) over one synthetic
"lego-800" image: 800x800 = 640,000 rays, 8 reference views, 64 coarse + 32 fine samples (the configuration
BASELINE.json's metric is quoted on), seeded random weights and feature maps; inputs are
resident in HBM before the timed region.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL).  Started without a launcher (`WORLD_SIZE` unset),
`python bench.py --gpus N` re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1` and refuses to run if fewer than N GPUs are visible; started under a launcher it insists that
WORLD_SIZE == N.  Images are sharded across ranks (no data-path collective; rays and images are independent), weak
scaling; the timed region is bracketed by barrier + synchronize and the max over ranks is taken.  After the timed
region an N > 1 run adds two side legs to the line: `split_image` (ONE image split over the ranks + one fused all-gather
of the tiles) and `train_ddp` (forward + backward + ONE flattened gradient all-reduce per step); they run after the headline is in
hand and under a watchdog (`--side-leg-timeout`), so a rank lost inside one costs the side legs, not the line.

One JSON line is printed by rank 0.  `roofline` is for the dominant kernel (the MFMA point kernel):
achieved = algorithmic FLOPs per launch / average launch duration (HIP events on the launch stream).
`cpu_baseline` times the golden-checked eager-PyTorch port of the reference's op sequence (oracle/torch_eager_port.py;
the reference tree itself does not exist on the GPU box; where it exists, the port runs the same batches within ~10 % of the reference
itself - 0.83 ... 1.02 x on a shared host - with identical pixels: tools/port_vs_reference_cpu.py) on the host - with torch.set_num_threads(physical cores) and,
on a many-core host, with 16 threads; the faster one is `value` - on a bounded sample of the same workload.  At N = 1 side measurements ride along (reported baselines, not the metric):
`eager_torch_baseline` (the same port on the same GPU - the stand-in for "the reference on stock PyTorch-ROCm"),
`numpy_oracle` (parity of the rendered image against the numpy oracle + its speed; `parity.full_image`, round 5: ALL rays of the image
against that port on the GPU, the rays it disputes re-rendered by the numpy oracle), `extra` (64+64 samples, the
reference CLI's 4096-ray batches), `training_step`, `init_net`, `pipeline_pcie_inclusive` (host buffers in, uint8 image
out), `bf16_variant`.  These legs are the only places this
file touches oracle/.

`--emulator-lib PATH` is a TEST HOOK (tests/test_bench_launcher.py): the same code path - launcher, process group,
sharding, collectives, JSON line - on CPU tensors with the `gloo` backend and the CPU emulator build of the kernels, on a
tiny image; its numbers are not measurements and the line says so.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from neuray_amd import synthetic  # noqa: E402
from neuray_amd.network.renderer import NeuralRayBaseRenderer  # noqa: E402

H = W = 800
RFN = 8
DN_COARSE = 64
RAY_BATCH = 65536      # rays per launch (results do not depend on it: bitwise batching invariance).  32768 in rounds 1-5; with the X3 kernel's shorter
                       # launches the gaps between them weigh more: 3.195 M rays/s at 32768, 3.231 at 65536, 3.235 / 3.242 at 131072 / 262144
                       # (profiles/r06_s_x3_ray_batch.log)
MFMA_F32_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: fp32-in MFMA dense peak


def point_macs(vis_head_used, folded=True):
    """-> (MACs per (sample point, reference view), MACs per sample point) of the point kernel (factored base_fc;
    SURVEY.md 8(d) 'minimal' count minus the attention / sigma head that live in the ray kernel).  folded (the inference packs since
    round 4): prob_embed.2, a Linear with no activation behind it (aggregate_net.py:27-31), is multiplied into neuray_fc.0 and
    base_fc.0 at pack time - 32 x 32 MACs per (point, view) that no longer exist in the network the kernel evaluates."""
    dist = 3 * (32 * 32 * 2) + 32 * 2 + 32 * 2 + 32 * 1 + (32 * 32 * 2 + 32 if vis_head_used else 0)
    per_view = (dist + (34 * 32 + (0 if folded else 32 * 32)) + (4 * 16 + 16 * 35) + (32 * 8 + 8) + ((35 + 32) * 64 + 64 * 32)
                + (32 * 32 + 32 * 33) + (32 * 32 + 32) + (37 * 16 + 16 * 8 + 8))
    per_point = 140 * 64 + (65 * 64 + 64 * 16)
    return per_view, per_point


def algorithmic_macs_per_point(rfn, vis_head_used, folded=True, slots_run_share=1.0):
    """MACs per sample point; slots_run_share < 1: the EXECUTED count when the kernel skipped that share of fully masked
    (16-point tile, view) slots (their per-view layers are multiplied by the zero mask downstream: csrc/nr_kernels.h points_kernel)."""
    per_view, per_point = point_macs(vis_head_used, folded)
    return rfn * per_view * slots_run_share + per_point


def build_case(device, fdn, seed, test_lib=None, arith='f32'):
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False},
           'depth_sample_num': DN_COARSE, 'fine_depth_sample_num': fdn,
           'agg_net_cfg': {'sample_num': DN_COARSE}, 'fine_agg_net_cfg': {'sample_num': fdn},
           'ray_batch_num': RAY_BATCH, 'hip_arith': arith}
    torch.manual_seed(0)
    renderer = NeuralRayBaseRenderer(cfg).eval()
    weights = {k: v.detach().numpy().copy() for k, v in renderer.state_dict().items()}
    renderer = renderer.to(device)
    renderer._engine_test_lib = test_lib
    que, ref = synthetic.make_scene(H, W, RFN, seed=seed)
    que['coords'] = synthetic.meshgrid_coords(H, W)
    # no `Ks_inv` key: the query constants take the drop-in route (K^-1 = torch.inverse on the host, engine.prepare_query)
    tq = {k: torch.from_numpy(v).to(device) for k, v in que.items()}
    tr = {k: torch.from_numpy(v).to(device) for k, v in ref.items()}
    return cfg, renderer, weights, que, ref, tq, tr


def render_image(renderer, tq, tr, split=False):
    q = dict(tq)
    r = {k: v for k, v in tr.items() if not k.startswith('_')}   # fresh dict: relayout is part of the step
    with torch.no_grad():
        if split:       # this rank's ray range + all-gather of the tiles (neuray_amd/parallel.py)
            from neuray_amd import parallel
            return parallel.render_image_sharded(renderer, q, r)
        return renderer.render(q, r, False)


def host_cpu():
    """-> (physical cores, model name) of the box this runs on"""
    model, pairs, phys, core = 'unknown', set(), None, None
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
            elif line.startswith('physical id'):
                phys = line.split(':', 1)[1].strip()
            elif line.startswith('core id'):
                core = line.split(':', 1)[1].strip()
                pairs.add((phys, core))
    except OSError:
        pass
    n = len(pairs)
    if n == 0:
        try:
            import psutil
            n = psutil.cpu_count(logical=False) or 0
        except Exception:                       # noqa: BLE001
            n = 0
    return (n or os.cpu_count() or 1), model


def cpu_baseline(cfg, weights, que, ref, budget_s=20.0, rays_per_batch=4096, max_batches=8):
    """The reference's CPU path as it can be timed on this box: the eager-PyTorch port of the reference's op sequence
    (oracle/torch_eager_port.py, checked against reference-generated goldens by tests/test_oracle_golden.py), fp32,
    torch.no_grad(), the reference CLI's 4096-ray batches (render.py:205), one warm-up batch, then batches of the same 800x800
    image until ~budget_s of CPU work, split between torch.set_num_threads(physical cores) and 16 threads."""
    from oracle import torch_eager_port as tep
    cores, model = host_cpu()
    old = torch.get_num_threads()
    w = {k: torch.from_numpy(v) for k, v in weights.items()}
    tq = {k: torch.from_numpy(v) for k, v in que.items()}
    tr = {k: torch.from_numpy(v) for k, v in ref.items()}
    ocfg = dict(cfg, coarse_use_vis=False, fine_use_vis=True)
    n = tq['coords'].shape[1]
    rays_per_batch = min(rays_per_batch, n)
    starts = np.linspace(0, n - rays_per_batch, max_batches + 1).astype(np.int64)

    def run(st):
        q = dict(tq)
        q['coords'] = tq['coords'][:, st:st + rays_per_batch]
        with torch.no_grad():
            return tep.render_impl(w, ocfg, q, tr)

    def timed(threads, budget, min_batches=3):
        torch.set_num_threads(threads)
        run(int(starts[0]))
        done, t0 = 0, time.perf_counter()
        for st in starts[1:]:
            run(int(st))
            done += 1
            if done >= min_batches and time.perf_counter() - t0 > budget:     # (at least 3 timed batches)
                break
        dt = time.perf_counter() - t0
        return done * rays_per_batch / dt, done, dt

    # all physical cores (what BASELINE.md asks for) and, on a many-core host, 16 threads: an op-by-op eager graph on 4096-ray
    # batches does not scale to 128 threads, and the baseline should be the faster of the two
    tried = {}
    try:
        for threads in ([cores, 16] if cores > 16 else [cores]):
            tried[threads] = timed(threads, budget_s / (2 if cores > 16 else 1))
    finally:
        torch.set_num_threads(old)
    best = max(tried, key=lambda t: tried[t][0])
    v, done, dt = tried[best]
    return {'value': v, 'unit': 'rays/s', 'cores': int(best), 'kind': 'port', 'cpu_model': model, 'physical_cores': int(cores),
            'rays_per_s_by_threads': {str(t): tried[t][0] for t in tried},
            'sample': '%d batches of %d rays of the same 800x800 image (64+%d samples, 8 views), eager-PyTorch port of the '
                      'reference op sequence, fp32, torch.set_num_threads(%d) (the faster of %s threads), %.1f s after one warm-up batch'
                      % (done, rays_per_batch, cfg['fine_depth_sample_num'], best, ' / '.join(str(t) for t in tried), dt)}


_ORACLE_SAMPLE = {}


# the gates of the tests (tests/test_render_parity.py, tests/test_full_size.py), applied to the bench's own sample as well
PARITY_GATES = {'coarse_pixel_max': 2e-4, 'chained_frac_within_2e-4_min': 0.99, 'chained_max': 5e-3, 'chained_psnr_db_min': 70.0}


def numpy_oracle_leg(cfg, weights, que, ref, got, sample_rays, chunk):
    """Parity of the rendered image against the numpy oracle on `sample_rays` strided rays (+ the oracle's own speed):
    the COARSE pixels (identical inputs: SURVEY 8(c)'s 2e-4) and the chained coarse -> fine pixels (distributional gates,
    DESIGN.md 2.4 / tests/test_chained_parity.py).  `got`: {'pixel_colors_nr', 'pixel_colors_nr_fine'} of the whole image."""
    from oracle import neuray_oracle as orc
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get('num_threads', 1) for p in threadpool_info()] + [1])
    except Exception:                       # noqa: BLE001
        cores = os.cpu_count() or 1
    ocfg = dict(cfg, coarse_use_vis=False, fine_use_vis=True)
    n = que['coords'].shape[1]
    idx = np.linspace(0, n - 1, sample_rays).astype(np.int64)
    outs = []
    t0 = time.perf_counter()
    for i in range(0, sample_rays, chunk):
        q = dict(que)
        q['coords'] = que['coords'][:, idx[i:i + chunk]]
        o = orc.render_impl(weights, ocfg, q, ref)
        outs.append((o['pixel_colors_nr'], o['pixel_colors_nr_fine']))
    dt = time.perf_counter() - t0
    want_c, want = (np.concatenate([o[i] for o in outs], 1) for i in (0, 1))
    _ORACLE_SAMPLE.update(idx=idx, coarse=want_c, fine=want)          # (kept for the split-variant leg: the oracle costs 30 s)
    fine = got['pixel_colors_nr_fine'][:, idx]
    err = np.abs(fine - want).max(-1)
    err_c = np.abs(got['pixel_colors_nr'][:, idx] - want_c).max(-1)
    parity = {'psnr_vs_oracle_db': synthetic.psnr_uint8(fine, want), 'max_abs_err_vs_oracle': float(err.max()),
              'frac_rays_within_2e-4': float(np.mean(err <= 2e-4)),
              'coarse_pixels': {'max_abs_err_vs_oracle': float(err_c.max()), 'p99.9': float(np.percentile(err_c, 99.9)),
                                'median': float(np.median(err_c)), 'gate': PARITY_GATES['coarse_pixel_max']},
              'rays': int(sample_rays), 'gates': PARITY_GATES,
              'vs_gt_psnr': 'unmeasurable here: no datasets / checkpoints in the container (north_star asks <= 0.05 dB vs the '
                            'reference on real scenes); ours-vs-oracle PSNR on the synthetic workload is the proxy'}
    parity['pass'] = bool(err_c.max() <= PARITY_GATES['coarse_pixel_max'] and parity['frac_rays_within_2e-4'] >= PARITY_GATES['chained_frac_within_2e-4_min']
                          and err.max() <= PARITY_GATES['chained_max'] and parity['psnr_vs_oracle_db'] >= PARITY_GATES['chained_psnr_db_min'])
    return {'value': sample_rays / dt, 'unit': 'rays/s', 'blas_threads': int(cores), 'kind': 'port (numpy oracle, the parity checker)',
            'sample': '%d strided rays of the same image, %.1f s' % (sample_rays, dt)}, parity


def eager_torch_baseline(cfg, weights, tq, tr, device, batches=6, rays_per_batch=4096):
    """Stand-in for BASELINE.md B2 ("the reference on stock PyTorch-ROCm", the denominator of the north star's >= 10x):
    the reference tree does not exist on the GPU box, so this is the golden-checked eager-PyTorch PORT of the reference's
    op sequence (oracle/torch_eager_port.py: the same ~700 small kernels per ray batch through PyTorch-ROCm) on the same
    device, same workload, the reference's default 4096-ray batches (render.py:205)."""
    from oracle import torch_eager_port as tep
    w = {k: torch.from_numpy(v).to(device) for k, v in weights.items()}
    ocfg = dict(cfg, coarse_use_vis=False, fine_use_vis=True)
    n = tq['coords'].shape[1]
    starts = np.linspace(0, n - rays_per_batch, batches + 1).astype(np.int64)

    def run(st):
        q = dict(tq)
        q['coords'] = tq['coords'][:, st:st + rays_per_batch]
        with torch.no_grad():
            return tep.render_impl(w, ocfg, q, tr)

    run(int(starts[0]))
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for st in starts[1:]:
        run(int(st))
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    return {'value': batches * rays_per_batch / dt, 'unit': 'rays/s',
            'kind': 'port of the reference (not the reference tree itself)',
            'what': 'eager-PyTorch port of the reference op sequence on the same MI355X, %d batches of %d rays'
                    % (batches, rays_per_batch)}


def float64_image(cfg, weights, tq, tr, device, chunk=4096):
    """The eager port (oracle/torch_eager_port.py) in FLOAT64 on the GPU over every ray of the image: weights and every float input cast
    up, as tests/golden/make_golden_full.py tile_case_f64 runs the reference itself (the port in float64 reproduces those fixtures to
    3e-9 / 2e-6: tests/test_oracle_golden.py).  -> (coarse pixels, chained fine pixels) as float64 tensors, seconds"""
    from oracle import torch_eager_port as tep
    up = lambda v: v.double() if v.is_floating_point() else v      # noqa: E731
    w = {k: torch.from_numpy(v).to(device).double() for k, v in weights.items()}
    q64 = {k: up(v) for k, v in tq.items() if not k.startswith('_')}
    r64 = {k: up(v) for k, v in tr.items() if not k.startswith('_')}
    ocfg = dict(cfg, coarse_use_vis=False, fine_use_vis=True)
    n = tq['coords'].shape[1]
    c64 = torch.empty(n, 3, device=device, dtype=torch.float64)
    f64 = torch.empty(n, 3, device=device, dtype=torch.float64)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    with torch.no_grad():
        for st in range(0, n, chunk):
            q = dict(q64)
            q['coords'] = q64['coords'][:, st:st + chunk]
            o = tep.render_impl(w, ocfg, q, r64)
            c64[st:st + chunk] = o['pixel_colors_nr'][0]
            f64[st:st + chunk] = o['pixel_colors_nr_fine'][0]
    torch.cuda.synchronize(device)
    return c64, f64, time.perf_counter() - t0


def against_float64(pixels_c, pixels_f, c64, f64):
    """coarse / chained pixels [n,3] of one evaluation against the float64 image"""
    ec = (pixels_c.double() - c64).abs().amax(-1)
    ef = (pixels_f.double() - f64).abs().amax(-1)
    mse = float(((pixels_f.double().clamp(0, 1) * 255).round() - (f64.clamp(0, 1) * 255).round()).pow(2).mean())
    return {'coarse': {'max': float(ec.max()), 'mean': float(ec.mean()), 'frac_within_2e-4': float((ec <= 2e-4).double().mean()),
                       'rays_beyond_2e-4': int((ec > 2e-4).sum()), 'rays_beyond_5e-3': int((ec > 5e-3).sum())},
            'chained': {'max': float(ef.max()), 'mean': float(ef.mean()), 'frac_within_2e-4': float((ef <= 2e-4).double().mean()),
                        'rays_beyond_2e-4': int((ef > 2e-4).sum()), 'rays_beyond_5e-3': int((ef > 5e-3).sum()),
                        'psnr_db': min(200.0, 10.0 * float(np.log10(255.0 ** 2 / mse))) if mse > 0 else 200.0}}


def float64_parity_leg(cfg, weights, tq, tr, got, others, port_fp32, device):
    """how many rays does the REFERENCE's fp32 arithmetic put beyond 2e-4 / 5e-3 of a float64 evaluation of the same
    image - next to ours?  `got`: the image of the timed loop; `others`: {label: image} of the other arithmetics of this library;
    `port_fp32`: (coarse, chained) of the eager port in fp32 on this GPU (the reference's op sequence in the reference's precision).
    Gate: each of our counts <= 1.25 x the fp32 reference arithmetic's own (+ 8 rays of slack for the small counts)."""
    c64, f64, dt = float64_image(cfg, weights, tq, tr, device)
    res = {'checker': 'eager port of the reference op sequence in FLOAT64 on this GPU, all %d rays, %.1f s' % (c64.shape[0], dt),
           'ours': against_float64(got['pixel_colors_nr'][0], got['pixel_colors_nr_fine'][0], c64, f64),
           'reference_arithmetic_fp32': against_float64(port_fp32[0], port_fp32[1], c64, f64)}
    for k, o in others.items():
        res[k] = against_float64(o['pixel_colors_nr'][0], o['pixel_colors_nr_fine'][0], c64, f64)
    ok = True
    for who in ['ours'] + list(others):
        for tier, key in (('coarse', 'rays_beyond_2e-4'), ('chained', 'rays_beyond_2e-4'), ('chained', 'rays_beyond_5e-3')):
            ok = ok and res[who][tier][key] <= 1.25 * res['reference_arithmetic_fp32'][tier][key] + 8
    res['gate'] = 'rays beyond 2e-4 (coarse, chained) and beyond 5e-3 (chained): ours <= 1.25 x the fp32 reference arithmetic + 8'
    res['pass'] = bool(ok)
    return res


def full_image_parity_leg(cfg, renderer, weights, que, ref, tq, tr, got, device, chunk=4096, ours_chunk=32768, recheck_cap=256, others=None):
    """parity over ALL rays of the bench image (the numpy-oracle leg covers 1.3 % of it: the oracle runs at 300 rays/s).
    Reference path: network/renderer.py:168-226.  `got`: the whole image as the timed loop rendered it.  Two tiers, both outside the
    timed region:
      tier 1  every ray against the golden-checked eager-PyTorch port of the reference's op sequence (oracle/torch_eager_port.py; pinned
              to the reference's own outputs and autograd by tests/test_oracle_golden.py) in fp32 on the SAME GPU, 4096-ray batches.
              That checker is fast (50 k rays/s) but not bit-faithful: PyTorch's GPU kernels contract and reassociate, so ITS coarse
              pixels sit 1e-5 off the numpy oracle (reported as `checker_vs_numpy_oracle`, on the oracle leg's 8 192 rays, next to ours:
              3e-7) and its bounds masks (render_ops.py:100-104,127-128) flip on points within an ulp of an image border.
      tier 2  therefore every ray tier 1 puts beyond the coarse gate, and the worst chained ones (up to `recheck_cap`), are re-rendered by
              the numpy oracle on the CPU - the restatement whose geometry is bit-identical to the reference's - and must be inside the
              gates THERE.  What remains of a chained difference is attributed per ray: how far its fine samples sit from the checker's
              (a chained error is a re-sampling displacement, not an arithmetic one)."""
    from oracle import neuray_oracle as orc
    from oracle import torch_eager_port as tep
    w = {k: torch.from_numpy(v).to(device) for k, v in weights.items()}
    ocfg = dict(cfg, coarse_use_vis=False, fine_use_vis=True)
    n = tq['coords'].shape[1]
    fdn = cfg['fine_depth_sample_num']
    want_c = torch.empty(n, 3, device=device)
    want_f = torch.empty(n, 3, device=device)
    want_fd = torch.empty(n, fdn, device=device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    with torch.no_grad():
        for st in range(0, n, chunk):
            q = dict(tq)
            q['coords'] = tq['coords'][:, st:st + chunk]
            o = tep.render_impl(w, ocfg, q, tr)
            want_c[st:st + chunk] = o['pixel_colors_nr'][0]
            want_f[st:st + chunk] = o['pixel_colors_nr_fine'][0]
            want_fd[st:st + chunk] = o['_fine_depth'][0]
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    # our fine sample depths: render_impl (returns the coarse hit probabilities that render() drops in eval mode) + the fine sampler on
    # them - the same kernels on the same inputs as the timed loop, whose pixels are checked to be reproduced bit for bit
    eng = renderer.engine(device)
    ours_fd = torch.empty(n, fdn, device=device)
    same = True
    with torch.no_grad():
        for st in range(0, n, ours_chunk):
            q = dict(tq)
            q['coords'] = tq['coords'][:, st:st + ours_chunk].contiguous()
            r = {k: v for k, v in tr.items()}
            o = renderer.render_impl(q, r, False)
            qconst = renderer._query(eng, q)
            depth = eng.sample_coarse_depth(q['depth_range'], q['coords'].shape[1], cfg['depth_sample_num'])
            ours_fd[st:st + ours_chunk] = eng.sample_fine_depth(qconst, depth.contiguous(), o['hit_prob_nr'][0].contiguous(), fdn,
                                                                use_all=cfg.get('fine_depth_use_all', False))
            same = same and bool(torch.equal(o['pixel_colors_nr_fine'][0], got['pixel_colors_nr_fine'][0, st:st + ours_chunk]))
    err_c = (got['pixel_colors_nr'][0] - want_c).abs().amax(-1)
    err_f = (got['pixel_colors_nr_fine'][0] - want_f).abs().amax(-1)
    disp = (ours_fd - want_fd).abs().amax(-1)                    # largest displacement of a (sorted) fine sample, metric depth
    moved = disp > 1e-4
    res = {
        'rays': int(n),
        'tier1_checker': 'golden-checked eager-PyTorch port of the reference op sequence, fp32, same GPU, %d-ray batches, %.1f s (%.0f rays/s)'
                         % (chunk, dt, n / dt),
        'coarse_pixels': {'max_abs_err': float(err_c.max()), 'p99.9': float(torch.quantile(err_c[::7].float(), 0.999)),
                          'median': float(err_c[::7].float().median()),
                          'frac_within_2e-4': float((err_c <= 2e-4).float().mean()), 'rays_beyond_2e-4': int((err_c > 2e-4).sum())},
        'chained_fine_pixels': {'max_abs_err': float(err_f.max()), 'p99.9': float(torch.quantile(err_f[::7].float(), 0.999)),
                                'median': float(err_f[::7].float().median()),
                                'frac_within_2e-4': float((err_f <= 2e-4).float().mean()),
                                'psnr_db': min(200.0, synthetic.psnr_uint8(got['pixel_colors_nr_fine'].cpu().numpy(), want_f[None].cpu().numpy())),
                                'rays_beyond_5e-3': int((err_f > PARITY_GATES['chained_max']).sum())},
        'rays_whose_fine_samples_sit_more_than_1e-4_from_the_checkers': int(moved.sum()),
        'fine_pixels_reproduced_bitwise_by_the_probe_render': same,
    }
    if 'idx' in _ORACLE_SAMPLE:       # the checker's own distance from the numpy oracle (the parity leg's rays), next to ours
        sel = torch.from_numpy(_ORACLE_SAMPLE['idx']).to(device)
        oc = torch.from_numpy(_ORACLE_SAMPLE['coarse'][0]).to(device)
        e_chk = (want_c[sel] - oc).abs().amax(-1)
        e_ours = (got['pixel_colors_nr'][0][sel] - oc).abs().amax(-1)
        res['checker_vs_numpy_oracle'] = {'rays': int(sel.numel()),
                                          'coarse_pixels_checker': {'max': float(e_chk.max()), 'p99.9': float(torch.quantile(e_chk, 0.999)), 'median': float(e_chk.median())},
                                          'coarse_pixels_ours': {'max': float(e_ours.max()), 'p99.9': float(torch.quantile(e_ours, 0.999)), 'median': float(e_ours.median())}}
    # tier 2: the outliers against the numpy oracle on the CPU
    bad_c = torch.nonzero(err_c > PARITY_GATES['coarse_pixel_max'])[:, 0]
    order = torch.argsort(err_f, descending=True)
    bad_f = order[:max(0, recheck_cap - min(int(bad_c.numel()), recheck_cap // 2))]
    bad_f = bad_f[err_f[bad_f] > PARITY_GATES['chained_max']]
    pick = torch.unique(torch.cat([bad_c[:recheck_cap // 2], bad_f])).cpu().numpy()
    tier2 = {'rays_rechecked': int(pick.size), 'coarse_outliers': int(bad_c.numel()), 'chained_outliers_beyond_5e-3': int((err_f > PARITY_GATES['chained_max']).sum()),
             'not_rechecked': int(max(0, bad_c.numel() - recheck_cap // 2) + max(0, int((err_f > PARITY_GATES['chained_max']).sum()) - int(bad_f.numel()))),
             'checker': 'numpy oracle on the CPU (geometry bit-identical to the reference)'}
    ok2 = True
    if pick.size:
        q = dict(que)
        q['coords'] = que['coords'][:, pick]
        o = orc.render_impl(weights, ocfg, q, ref)
        gc = got['pixel_colors_nr'][0].cpu().numpy()[pick]
        gf = got['pixel_colors_nr_fine'][0].cpu().numpy()[pick]
        e2c = np.abs(gc - o['pixel_colors_nr'][0]).max(-1)
        e2f = np.abs(gf - o['pixel_colors_nr_fine'][0]).max(-1)
        tier2.update({'coarse_max_abs_err_vs_numpy_oracle': float(e2c.max()), 'chained_max_abs_err_vs_numpy_oracle': float(e2f.max()),
                      'chained_frac_within_2e-4_vs_numpy_oracle': float(np.mean(e2f <= 2e-4)),
                      'chained_rays_beyond_5e-3_vs_numpy_oracle': int(np.sum(e2f > PARITY_GATES['chained_max']))})
        ok2 = bool(e2c.max() <= PARITY_GATES['coarse_pixel_max'])
    res['tier2_outliers_vs_numpy_oracle'] = tier2
    res['gates'] = {'tier1': {'coarse_frac_within_2e-4_min': 0.9999, 'chained_frac_within_2e-4_min': 0.98, 'chained_psnr_db_min': 60.0,
                              'why': "against a checker that is itself 1e-5 from the reference's arithmetic (checker_vs_numpy_oracle) the "
                                     "distributional gates are SURVEY 8(c)'s PSNR >= 60 dB and the shares; the strict gates are tier 2's"},
                    'tier2': {'coarse_pixel_max_vs_numpy_oracle': PARITY_GATES['coarse_pixel_max']}}
    # the float64 yardstick over the whole image: ours and the fp32 reference arithmetic (the tier-1 checker's own pixels) against it
    res['vs_float64'] = float64_parity_leg(cfg, weights, tq, tr, got, others or {}, (want_c, want_f), device)
    res['pass'] = bool(res['coarse_pixels']['frac_within_2e-4'] >= 0.9999 and res['chained_fine_pixels']['frac_within_2e-4'] >= 0.98
                       and res['chained_fine_pixels']['psnr_db'] >= 60.0 and ok2 and same and res['vs_float64']['pass'])
    return res


def training_step_timing(device, steps=3):
    """Side measurement (not the headline metric): one render_impl(is_train=True) + backward through the HIP forward and
    backward kernels on 512 rays x 8 views x 64+64 samples (the shape of BASELINE.json configs[3]/[4]) next to autograd
    of the eager-PyTorch port on the same GPU."""
    from oracle import torch_eager_port as tep
    from oracle import neuray_oracle as orc
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': 64,
           'fine_depth_sample_num': 64, 'agg_net_cfg': {'sample_num': 64}, 'fine_agg_net_cfg': {'sample_num': 64},
           'use_self_hit_prob': True}
    torch.manual_seed(0)
    r = NeuralRayBaseRenderer(cfg).train().to(device)
    weights = {k: v.detach().cpu().numpy().copy() for k, v in r.state_dict().items()}
    que, ref = synthetic.make_scene(400, 600, 8, seed=0, que_imgs=True)
    rng = np.random.RandomState(0)
    que['coords'] = (rng.rand(1, 512, 2) * np.array([599, 399])).astype(np.float32)
    tq = {k: torch.from_numpy(v).to(device) for k, v in que.items()}
    tr = {k: torch.from_numpy(v).to(device) for k, v in ref.items()}
    for t_ in (tr['ray_feats'], tr['img_feats'], tq['ray_feats']):
        t_.requires_grad_(True)
    tgt = torch.rand(1, 512, 3, device=device)

    def loss_of(out):
        return ((out['pixel_colors_nr'] - tgt) ** 2).mean() + ((out['pixel_colors_nr_fine'] - tgt) ** 2).mean() + \
            out['hit_prob_self'].mean() + out['hit_prob_self_fine'].mean()

    def ours():
        r.zero_grad(set_to_none=True)
        loss_of(r.render_impl(tq, tr, True)).backward()

    w = {k: torch.from_numpy(v).to(device).requires_grad_(True) for k, v in weights.items()}
    ocfg = {**orc.DEFAULT_CFG, **cfg, 'coarse_use_vis': False, 'fine_use_vis': True}
    que_t = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in tq.items() if torch.is_tensor(v)}
    ref_t = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in tr.items() if torch.is_tensor(v)}

    def eager():
        for p_ in w.values():
            p_.grad = None
        loss_of(tep.render_impl(w, ocfg, que_t, ref_t, is_train=True)).backward()

    def timeit(fn, warm, n):
        # (the first ~100 ms after the CPU legs' torch.set_num_threads() calls run several times slower on the host side: warm up past it)
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(device)
        return 1e3 * (time.perf_counter() - t0) / n

    # the same policy for both legs: 10 warm-up steps, then the mean over 10 * steps timed ones
    a, b = timeit(ours, 10, 10 * steps), timeit(eager, 10, 10 * steps)
    out = {'what': 'forward + backward, 512 rays x 8 views x 64+64 samples, HIP kernels vs autograd of the eager-PyTorch port',
           'hip_ms_per_step': a, 'eager_torch_ms_per_step': b, 'speedup_vs_eager': b / a,
           'timing_policy': '10 warm-up steps, mean of %d timed steps, both legs' % (10 * steps)}
    # the dominant kernel of the step by HIP events on its launch stream: the point backward, one launch per pass
    eng = r.engine(device)
    eng.timing = []
    for _ in range(5):
        ours()
    torch.cuda.synchronize(device)
    ts = [e0.elapsed_time(e1) for name, e0, e1, _ in eng.timing if name == 'points_backward']
    eng.timing = None
    if ts:
        ms = float(np.median(ts))
        conv = 3 * 2.0 * algorithmic_macs_per_point(8, vis_head_used=False, folded=False) * 512 * 64 / (ms * 1e-3) / 1e12
        out['point_backward'] = {'ms_per_pass': ms, 'launches_timed': len(ts),
                                 'tflops_by_3x_forward_convention': conv, 'frac_of_fp32_mfma_peak': conv / MFMA_F32_PEAK_TFLOPS,
                                 'note': 'a backward pass counted as 3 x the forward algorithmic FLOP of its 512 x 64 points (round-1 judge convention); '
                                         'the kernel reads the cross-view quantities the training forward saved and recomputes only per-view layers; one pass = its two '
                                         'launches (network tail, then front, with a hand-over buffer: nr_kernels_bwd2.h B2Part), timed together'}
    return out


def ft_step_timing(device, steps=20):
    """Side measurement: one whole fine-tuning step as run_training.py's ft configs do it (BASELINE.json configs[3]) -
    NeuralRayFtRenderer.train_step (a random query view of a 24-view 800 x 800 scene, 8 neighbours, 512 rays, 64 + 64 samples:
    encoders on 9 images + per-ray forward) + loss.backward() (HIP backward kernels + encoder backward) + Adam on every parameter
    incl. the touched per-view ray_feats maps."""
    from neuray_amd import pipeline
    from neuray_amd.network.renderer import NeuralRayFtRenderer
    db = synthetic.MemoryDatabase(24, 800, 800, seed=0)
    scene = {'ref_imgs_info': pipeline.build_imgs_info(db, db.get_img_ids(), -1, True, False, True, True)}
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'use_self_hit_prob': True, 'use_validation': False,
           'train_ray_num': 512}
    torch.manual_seed(0)
    np.random.seed(0)
    ft = NeuralRayFtRenderer(cfg, scene=scene).train().to(device)
    opt = torch.optim.Adam(ft.parameters(), lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        out = ft.train_step()
        loss = ((out['pixel_colors_nr'] - out['pixel_colors_gt']) ** 2).mean() + ((out['pixel_colors_nr_fine'] - out['pixel_colors_gt']) ** 2).mean() + \
            out['hit_prob_self'].mean() + out['hit_prob_self_fine'].mean()
        loss.backward()
        opt.step()
    for _ in range(60):             # (also builds the views' cached pixel lists of the ray sampler - 5 ms per first visit of one of the 24 views -
        step()                      # and gets MIOpen's per-shape selections and the allocator's pools settled: 15 warm-up steps left 3 ms of that in the timed ones)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(device)
    return {'what': 'NeuralRayFtRenderer.train_step + backward + Adam: 512 rays, 8 of 24 views of 800 x 800, 64+64 samples, encoders trained',
            'ms_per_step': 1e3 * (time.perf_counter() - t0) / steps}


def gen_train_case(device, h=416, w=608, rfn=8, extra_src=4, rays=512, host_ks_inv=False):
    """One generalisation-training step at the shape of BASELINE.json configs[4] (configs/train/gen/neuray_gen_cost_volume_train.yaml;
    dataset/train_dataset.py:78-102,304-378): NeuralRayGenRenderer with the cost-volume init net (`init_net_type: cost_volume`, frozen
    MVSNet + trained heads), image / visibility encoders, 8 working views of a 400 x 600 crop padded to the ref_pad_interval of 32
    (416 x 608), 3 cost-volume neighbours per view out of 8 + `extra_src` source views, 512 rays, 64 + 64 samples, render + depth loss
    (network/loss.py:46-77,79-130), Adam on every trainable parameter.  -> (model, optimiser, step function)"""
    from neuray_amd.network.renderer import NeuralRayGenRenderer
    cfg = {'init_net_type': 'cost_volume', 'use_hierarchical_sampling': True, 'use_depth_loss': True, 'dist_decoder_cfg': {'use_vis': False},
           'fine_dist_decoder_cfg': {'use_vis': False}, 'ray_batch_num': 2048, 'depth_loss_coords_num': 8192}
    torch.manual_seed(0)
    np.random.seed(0)
    model = NeuralRayGenRenderer(cfg).train().to(device)
    n_src = rfn + extra_src
    que, scene = synthetic.make_scene(h, w, n_src, seed=0, que_imgs=True, smooth=True)
    cams = {k: torch.from_numpy(scene[k]).to(device) for k in ('imgs', 'poses', 'Ks', 'depth_range')}
    src = dict(cams)
    ref = {k: v[:rfn].contiguous() for k, v in cams.items()}
    ref['nn_ids'] = torch.tensor([[(v + 1) % n_src, (v + 2) % n_src, (v + 5) % n_src] for v in range(rfn)], device=device)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    ref['true_depth'] = torch.from_numpy(np.stack([3.6 + 0.7 * np.sin(xx / 90.0 + v) * np.cos(yy / 70.0 - v)
                                                   for v in range(rfn)])[:, None].astype(np.float32)).to(device)
    tq = {k: torch.from_numpy(v).to(device) for k, v in que.items() if k != 'ray_feats'}
    tq['coords'] = torch.from_numpy((np.random.RandomState(0).rand(1, rays, 2) * np.array([w - 1, h - 1])).astype(np.float32)).to(device)
    if host_ks_inv:
        # what a host data pipeline can hand over with the batch: K^-1 of the query view, evaluated where engine.prepare_query would
        # evaluate it (torch.inverse on the host, the same call on the same bytes) BEFORE the upload - the step then has no device ->
        # host copy, i.e. no wait for the previous step's queue (INTEGRATION.md "Training")
        from neuray_amd.engine import host_inverse
        tq['Ks_inv'] = host_inverse(torch.from_numpy(que['Ks'])).to(device)
    opt = torch.optim.Adam([p_ for p_ in model.parameters() if p_.requires_grad], lr=4e-4)
    near, far = -1 / ref['depth_range'][:, 0:1], -1 / ref['depth_range'][:, 1:2]

    def step():
        opt.zero_grad(set_to_none=True)
        out = model({'que_imgs_info': dict(tq), 'ref_imgs_info': dict(ref), 'src_imgs_info': dict(src), 'scene_name': 'synthetic'})
        gt = out['pixel_colors_gt']
        loss = ((out['pixel_colors_nr'] - gt) ** 2).mean() + ((out['pixel_colors_nr_fine'] - gt) ** 2).mean()
        # depth loss (loss.py:79-130, l2): decoded mixture means of every view at random pixels against the normalised inverse true depth
        c = out['depth_coords'].long()                          # [rfn, pn, 2]
        d = ref['true_depth'][torch.arange(rfn, device=device)[:, None], 0, c[..., 1].clamp(max=h - 1), c[..., 0].clamp(max=w - 1)]
        d = (((-1 / d.clamp(min=1e-5)) - near) / (far - near)).clamp(0, 1)
        loss = loss + ((d - out['depth_mean']) ** 2).mean() + ((d - out['depth_mean_fine']) ** 2).mean()
        loss.backward()
        opt.step()
        return loss
    return model, opt, step


def gen_train_step_timing(device, steps=10):
    """Side measurement (VERDICT r3 missing #1 / next #5): the whole config-5-shaped training step (gen_train_case)."""
    model, opt, step = gen_train_case(device)
    for _ in range(6):
        step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize(device)
    ms = 1e3 * (time.perf_counter() - t0) / steps
    del model, opt, step
    _, _, step2 = gen_train_case(device, host_ks_inv=True)
    for _ in range(6):
        step2()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        step2()
    torch.cuda.synchronize(device)
    ms_kinv = 1e3 * (time.perf_counter() - t0) / steps
    # the init net alone, forward only (what an evaluation pays per image at this size)
    from neuray_amd.network import render_ops as _ro
    _ro.check_deferred_inputs(device, wait=True)        # the last step's device-side input checks (warp_variance) are raised here
    return {'what': 'NeuralRayGenRenderer(cost_volume init net + encoders) forward + render/depth loss + backward + Adam: 512 rays, 8 views of '
                    '416 x 608 (+ 4 source views), 64+64 samples, fp32',
            'ms_per_step': ms, 'ms_per_step_with_host_Ks_inv': ms_kinv,
            'note': "ms_per_step: the batch as the reference's loader hands it over (K^-1 of the query view = torch.inverse on the host "
                    "inside the step: a 36-byte device -> host copy that waits for the previous step's queue); with_host_Ks_inv: the batch "
                    "carries que_imgs_info['Ks_inv'] computed by the host pipeline before the upload (the two agree within the run-to-run "
                    "noise of +-2 ms: the copy is taken before anything of the step is queued)",
            'loss_is_finite': bool(torch.isfinite(loss).item()),
            'kernel_classes': 'profiles/r06_zzc_gen_step_by_class.txt (bash profiles/collect_gen_step.sh)'}


def encoder_timing(device, n=9, hw=(800, 800), reps=5):
    """Side measurement: image_encoder + vis_encoder on the fine-tuning step's 9 images of 800 x 800
    (renderer.py:229-235), forward and forward + backward, with the fused InstanceNorm / activation / residual / reflection-pad
    kernels (csrc/nr_kernels_norm.h) and with the PyTorch composition they replace (MIOpen batch-norm + element-wise + pad kernels).
    `fused_norm`: the product path - those kernels plus the 3 x 3 stride-1 convolutions (forward and data gradient) on the split-operand
    bf16 MFMA (csrc/nr_kernels_conv2d.h); `fused_norm_miopen_convs`: the same with every convolution MIOpen's (the round-5 path);
    `pytorch_norm`: MIOpen / PyTorch throughout."""
    from neuray_amd.network import encoders
    torch.manual_seed(0)
    img_enc, vis_enc = encoders.ImageEncoder().to(device), encoders.DefaultVisEncoder({}).to(device)
    imgs = torch.rand(n, 3, hw[0], hw[1], device=device)
    ray0 = torch.randn(n, 32, hw[0] // 4, hw[1] // 4, device=device, requires_grad=True)

    def fwd():
        f = img_enc(imgs)
        return f, vis_enc(ray0, f)

    def fwd_bwd():
        f, r_ = fwd()
        (f.square().mean() + r_.square().mean()).backward()
        for p_ in list(img_enc.parameters()) + list(vis_enc.parameters()) + [ray0]:
            p_.grad = None

    def timeit(fn, grad):
        with (torch.enable_grad() if grad else torch.no_grad()):
            for _ in range(6):
                fn()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(device)
        return 1e3 * (time.perf_counter() - t0) / reps
    out = {'what': 'image_encoder + vis_encoder, %d x 3 x %d x %d, fp32' % (n, hw[0], hw[1])}
    try:
        for tag, on, x3 in (('fused_norm', True, True), ('fused_norm_miopen_convs', True, False), ('pytorch_norm', False, False)):
            encoders.set_fused_norm(on)
            encoders.set_x3_conv(x3)
            out[tag] = {'forward_ms': timeit(fwd, False), 'forward_backward_ms': timeit(fwd_bwd, True)}
    finally:
        encoders.set_fused_norm(True)
        encoders.set_x3_conv(True)
    out['speedup_forward'] = out['pytorch_norm']['forward_ms'] / out['fused_norm']['forward_ms']
    out['speedup_forward_backward'] = out['pytorch_norm']['forward_backward_ms'] / out['fused_norm']['forward_backward_ms']
    return out


def fp32_mfma_leg(device, fdn, tq, tr, steps, headline_value):
    """The SAME workload with the MLP contractions on the fp32 MFMA (cfg['hip_arith'] = 'f32', v_mfma_f32_16x16x4_f32: the
    arithmetic of rounds 1-5), timed exactly like the headline and reported next to it whatever the headline arithmetic is.
    -> (block of the line, the rendered image for the float64 parity leg)"""
    cfg, r, _, _, _, _, _ = build_case(device, fdn, seed=0, arith='f32')
    eng = r.engine(device)
    stats = torch.zeros(2, dtype=torch.int64, device=device)
    eng.slot_stats = stats
    out = render_image(r, tq, tr)
    eng.slot_stats = None
    eng.timing = []
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = render_image(r, tq, tr)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    pts = [(e0.elapsed_time(e1) * 1e-3, n) for name, e0, e1, n in eng.timing if name == 'points']
    eng.timing = None
    st = stats.cpu().numpy()
    share = float(st[0] / st[1]) if st[1] > 0 else 1.0
    t_pts, n_pts = sum(t for t, _ in pts), sum(n for _, n in pts)
    achieved = 2.0 * algorithmic_macs_per_point(RFN, vis_head_used=False, folded=True, slots_run_share=share) * n_pts / t_pts / 1e12
    value = steps * H * W / dt
    return {'what': "the same workload, same steps, with cfg['hip_arith'] = 'f32': every MLP contraction on v_mfma_f32_16x16x4_f32 (the headline "
                    "arithmetic of rounds 1-5)", 'value': value, 'unit': 'rays/s', 'ms_per_step': 1e3 * dt / steps, 'steps': steps,
            'point_kernel_avg_launch_ms': 1e3 * t_pts / max(1, len(pts)),
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / MFMA_F32_PEAK_TFLOPS,
                         'kernel': 'nr::points_kernel<..., AR_F32>', 'view_slots_run_share': share},
            'headline_over_this': headline_value / value}, out


def bf16_variant_timing(device, fdn, tq, tr, fp32_pixels, steps=2, variant='bf16'):
    """Side measurement, reported SEPARATELY from the headline (which stays fp32, the reference's arithmetic): the same
    workload through libneuray_hip_bf16.so - bf16 MFMA operands (weights and activations rounded to bf16 in the quad
    K-steps of every layer), fp32 accumulation, everything else fp32 - with its distance from the fp32 render.
    variant='bf16x3': libneuray_hip_bf16x3.so, the error-compensated split (hi + lo bf16 operands, three bf16 MFMAs per fp32 quad),
    which is additionally held against the numpy oracle and the FP32 gates on the parity leg's ray sample."""
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': DN_COARSE,
           'fine_depth_sample_num': fdn, 'agg_net_cfg': {'sample_num': DN_COARSE}, 'fine_agg_net_cfg': {'sample_num': fdn},
           'ray_batch_num': RAY_BATCH, 'hip_variant': variant}
    torch.manual_seed(0)
    r = NeuralRayBaseRenderer(cfg).eval().to(device)
    eng = r.engine(device)
    out = render_image(r, tq, tr)
    eng.timing = []
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = render_image(r, tq, tr)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    pts_n = [(e0.elapsed_time(e1), n) for name, e0, e1, n in eng.timing if name == 'points']
    pts = [t for t, _ in pts_n]
    eng.timing = None
    got = out['pixel_colors_nr_fine'].cpu().numpy()
    err = np.abs(got - fp32_pixels).max(-1).reshape(-1)
    res = {'what': 'same workload, bf16 MFMA operands / fp32 accumulate (libneuray_hip_bf16.so); NOT the headline', 'dtype': 'bf16 operands, fp32 accumulate',
           'value': steps * H * W / dt, 'unit': 'rays/s', 'point_kernel_ms_per_launch': float(np.mean(pts)),
           'psnr_vs_fp32_render_db': synthetic.psnr_uint8(np.clip(got, 0, 1), np.clip(fp32_pixels, 0, 1)),
           'max_abs_err_vs_fp32_render': float(err.max()), 'frac_rays_within_1e-2': float(np.mean(err <= 1e-2))}
    if variant == 'bf16x3':
        res['what'] = ('same workload, error-compensated split: every MFMA operand as hi + lo bf16 halves, hi*hi + hi*lo + lo*hi = three bf16 '
                       'MFMAs per fp32 quad, fp32 accumulate (libneuray_hip_bf16x3.so); NOT the headline, NOT the default')
        res['dtype'] = 'hi + lo bf16 operands (16 significand bits), fp32 accumulate'
        res['algorithmic_tflops'] = 2.0 * algorithmic_macs_per_point(RFN, vis_head_used=False) * sum(n for _, n in pts_n) / (sum(pts) * 1e-3) / 1e12
        if _ORACLE_SAMPLE:
            idx = _ORACLE_SAMPLE['idx']
            ec = np.abs(out['pixel_colors_nr'].cpu().numpy()[:, idx] - _ORACLE_SAMPLE['coarse']).max(-1)
            ef = np.abs(got[:, idx] - _ORACLE_SAMPLE['fine']).max(-1)
            res['parity_vs_oracle'] = {
                'coarse_pixels_max': float(ec.max()), 'coarse_gate': PARITY_GATES['coarse_pixel_max'], 'coarse_pass': bool(ec.max() <= PARITY_GATES['coarse_pixel_max']),
                'chained_frac_within_2e-4': float(np.mean(ef <= 2e-4)), 'chained_max': float(ef.max()),
                'chained_psnr_db': synthetic.psnr_uint8(np.clip(got[:, idx], 0, 1), np.clip(_ORACLE_SAMPLE['fine'], 0, 1)),
                'note': 'stage-wise (identical inputs) it passes the fp32 gates on every reference tile (tests/test_bf16x3_variant.py); chained, its '
                        '1e-5-level coarse differences displace more fine samples than the fp32 path does (DESIGN.md 2.4)'}
    return res


def init_net_timing(device, reps=10):
    """Side measurement: the depth init net front end on 8 views of 800 x 800 - get_diff_feats as the
    fused neuray_diff_feats kernel next to the eager tensor formulation of the reference (oracle/torch_eager_port.py),
    and the whole DepthInitNet forward."""
    from oracle import torch_eager_port as tep
    from neuray_amd.network import init_net
    h = w = 800
    _, ref = synthetic.make_scene(h, w, 8, seed=0)
    info = {k: torch.from_numpy(ref[k]).to(device) for k in ('imgs', 'poses', 'Ks', 'depth_range')}
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    info['depth'] = torch.from_numpy(np.stack([3.6 + 0.7 * np.sin(xx / 90.0 + v) * np.cos(yy / 70.0 - v)
                                               for v in range(8)])[:, None].astype(np.float32)).to(device)
    net = init_net.DepthInitNet({}).eval().to(device)

    def timeit(fn, n):
        fn()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(device)
        return 1e3 * (time.perf_counter() - t0) / n
    with torch.no_grad():
        dn = init_net.extract_depth_for_init(info)
        got, want = init_net.get_diff_feats(info, dn), tep.get_diff_feats(info, dn)
        res = {'what': 'depth init net front end, 8 x 800 x 800: 41 M lift-project-gather pairs',
               'hip_get_diff_feats_ms': timeit(lambda: init_net.get_diff_feats(info, dn), reps),
               'eager_torch_get_diff_feats_ms': timeit(lambda: tep.get_diff_feats(info, dn), 3),
               'depth_init_net_ms': timeit(lambda: net(info, None, False), reps),
               'frac_within_1e-4_of_eager': float(((got - want).abs() <= 1e-4).float().mean())}
    res['speedup_vs_eager'] = res['eager_torch_get_diff_feats_ms'] / res['hip_get_diff_feats_ms']
    # f-3: plane-sweep variance volume of the cost-volume init net, one reference view of the 800 x 800 evaluation path
    # (built at 640 x 640: 160 x 160 x 64 planes, 3 source views), and the whole CostVolumeInitNet forward
    from neuray_amd.network import render_ops
    eng = render_ops.engine_for(device)
    info['nn_ids'] = torch.tensor([[(v + 1) % 8, (v + 2) % 8, (v + 3) % 8] for v in range(8)], device=device)
    cv = init_net.CostVolumeInitNet({}).eval().to(device)
    with torch.no_grad():
        f = torch.randn(8, 32, 160, 160, generator=torch.Generator().manual_seed(0)).to(device)
        prj = init_net.construct_project_matrix(0.2, 0.2, info['Ks'], info['poses'])
        dv = init_net.get_depth_vals(info['depth_range'], 64)
        ids = info['nn_ids'][:1]
        got, want = eng.warp_variance(f[:1], f, ids, prj[:1], prj, dv[:1]), tep.variance_volume(f[:1], f, ids, prj[:1], prj, dv[:1])
        got_cl = eng.warp_variance(f[:1], f, ids, prj[:1], prj, dv[:1], channels_last=True)
        res.update({'hip_warp_variance_ms_per_ref_view': timeit(lambda: eng.warp_variance(f[:1], f, ids, prj[:1], prj, dv[:1]), reps),
                    # the layout the init net's conv0 kernel reads (warp_variance_cl_kernel: eight lanes per voxel), the same bits
                    'hip_warp_variance_channels_last_ms_per_ref_view': timeit(
                        lambda: eng.warp_variance(f[:1], f, ids, prj[:1], prj, dv[:1], channels_last=True), reps),
                    'warp_variance_channels_last_equals_ncdhw_bitwise': bool(torch.equal(got_cl, got)),
                    'eager_torch_warp_variance_ms_per_ref_view': timeit(lambda: tep.variance_volume(f[:1], f, ids, prj[:1], prj, dv[:1]), 3),
                    'warp_variance_frac_within_1e-3_of_eager': float(((got - want).abs() <= 1e-3).float().mean()),
                    'cost_volume_init_net_ms': timeit(lambda: cv(info, info, False), 3)})
    res['warp_variance_speedup_vs_eager'] = res['eager_torch_warp_variance_ms_per_ref_view'] / res['hip_warp_variance_ms_per_ref_view']
    return res


def pipeline_timing(device, fdn, poses=3, views=16, arith='f32'):
    """Side measurement: host buffers in -> uint8 image out.  An in-memory scene with the reference's
    database accessors (uint8 images, masks, depth maps on the HOST), render.py's loop through neuray_amd.pipeline: working
    views by camera distance, uint8 upload of each view once (DeviceViewCache), depth init net + encoders + the 64+fdn render
    path, quantisation on the GPU, uint8 copy back.  PCIe-inclusive rays/s: first pass (every working view uploaded) and
    resident pass."""
    from neuray_amd import pipeline
    from neuray_amd.network import renderer as R
    db = synthetic.MemoryDatabase(views, H, W, seed=0)
    gen = R.NeuralRayGenRenderer({'use_hierarchical_sampling': True, 'fine_depth_sample_num': fdn, 'fine_agg_net_cfg': {'sample_num': fdn},
                                  'ray_batch_num': RAY_BATCH, 'init_net_type': 'depth', 'dist_decoder_cfg': {'use_vis': False}, 'hip_arith': arith}).eval().to(device)
    qposes = np.stack([synthetic.look_at_pose(synthetic.sphere_pos(4.03, 360.0 * (i + 0.37) / views, 27.0)) for i in range(poses)]).astype(np.float32)
    ref_ids = pipeline.select_working_views_db(db, None, qposes, RFN, False)
    K, shape, dr = [db.get_K(0)] * poses, [(H, W)] * poses, [(2.0, 6.0)] * poses
    cache = pipeline.DeviceViewCache(db, device, pad_interval=16)
    pipeline.render_poses(gen, db, qposes[:1], K[:1], shape[:1], dr[:1], ref_ids[:1], save_fn=lambda qi, im: None)     # warm-up (MIOpen search)

    def timed():
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        pipeline.render_poses(gen, db, qposes, K, shape, dr, ref_ids, cache=cache, save_fn=lambda qi, im: None)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0
    t_first = timed()
    up = cache.uploaded_bytes
    t_warm = timed()
    return {'what': 'host uint8 views in -> uint8 image out: depth init net + encoders + 64+%d render, %d working views of a %d-view %dx%d scene, %d poses'
                    % (fdn, RFN, views, H, W, poses),
            'first_pass_rays_per_s': poses * H * W / t_first, 'h2d_MB_per_image_first_pass': up / poses / 1e6,
            'resident_rays_per_s': poses * H * W / t_warm, 'd2h_MB_per_image': H * W * 3 / 1e6}


def fresh_process_leg(name, device):
    """The whole-step training legs are host-sensitive (1 600 launches per step, host time ~ device time): measured at the end of this
    process - after the CPU baselines' thread-pool changes, the eager port, seven other workloads - the generalisation step reads 47 ms
    where a process that only trains reads 34 (same code, same box; set_num_threads, garbage-collector load and the preceding GPU legs were
    each ruled out on their own).  A training run IS a fresh process, so that is what the leg measures: `python bench.py --leg <name>` on
    the same GPU while this process idles; if the child fails, the leg runs here and says so."""
    import subprocess
    try:
        if (device.index or 0) != 0:
            raise RuntimeError('the fresh-process legs run on cuda:0 of the visible devices')
        res = subprocess.run([sys.executable, os.path.abspath(__file__), '--leg', name], capture_output=True, text=True, timeout=600)
        for line in reversed(res.stdout.splitlines()):
            if line.startswith('{') and '"leg"' in line:
                out = json.loads(line)['result']
                out['measured_in'] = 'a fresh process (python bench.py --leg %s), as a training run is' % name
                return out
        raise RuntimeError('no result line (rc %d): %s' % (res.returncode, res.stderr[-300:]))
    except Exception as e:                          # noqa: BLE001
        out = FRESH_PROCESS_LEGS[name](device)
        out['measured_in'] = 'this process (fresh-process run failed: %s: %s)' % (type(e).__name__, e)
        return out


def side(fn, *a, **k):
    """a side measurement must never cost the headline line: report its failure instead"""
    try:
        return fn(*a, **k)
    except Exception as e:                          # noqa: BLE001
        return {'error': '%s: %s' % (type(e).__name__, e)}


def extra_config_timing(device, fdn, ray_batch, tq, tr, steps=2, arith='f32'):
    """The same image at another sampling / batching configuration (reported next to the headline, never instead of it)."""
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': DN_COARSE,
           'fine_depth_sample_num': fdn, 'agg_net_cfg': {'sample_num': DN_COARSE}, 'fine_agg_net_cfg': {'sample_num': fdn},
           'ray_batch_num': ray_batch, 'hip_min_ray_batch': 0, 'hip_arith': arith}      # (launches of exactly `ray_batch` rays: render() would merge them otherwise)
    torch.manual_seed(0)
    r = NeuralRayBaseRenderer(cfg).eval().to(device)
    eng = r.engine(device)
    render_image(r, tq, tr)
    eng.slot_stats = stats = torch.zeros(2, dtype=torch.int64, device=device)      # (one untimed image counts the skipped slots)
    render_image(r, tq, tr)
    eng.slot_stats = None
    eng.timing = []
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        render_image(r, tq, tr)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    pts = [(e0.elapsed_time(e1) * 1e-3, n) for name, e0, e1, n in eng.timing if name == 'points']
    eng.timing = None
    st = stats.cpu().numpy()
    share = float(st[0] / st[1]) if st[1] > 0 else 1.0
    achieved = 2.0 * algorithmic_macs_per_point(RFN, False, True, share) * sum(n for _, n in pts) / sum(t for t, _ in pts) / 1e12     # executed FLOPs
    return {'samples': '64+%d' % fdn, 'ray_batch': ray_batch, 'value': steps * H * W / dt, 'unit': 'rays/s', 'view_slots_run_share': share,
            'point_kernel_ms_per_launch': 1e3 * sum(t for t, _ in pts) / len(pts), 'point_kernel_frac_of_fp32_mfma_peak': achieved / MFMA_F32_PEAK_TFLOPS}


def llff_fern_cost_volume_timing(device, arith, steps=2, h=756, w=1008, rfn=8, extra_src=4):
    """BASELINE.json configs[2] end to end (VERDICT r5 missing #4): `neuray_gen_cost_volume` (configs/gen/neuray_gen_cost_volume.yaml:
    NeuralRayGenRenderer, init_net_type cost_volume, 64 + 64 samples) on an llff_colmap/fern/high-shaped scene - a 756 x 1008 query view
    (762 048 rays), 8 working views padded to the ref_pad_interval of 32 (768 x 1024, dataset/train_dataset.py / render.py), 3 cost-volume
    neighbours per view out of 8 + `extra_src` source views, depth range [1.2, 12] - through renderer(data) in evaluation: MVSNet feature
    net + plane-sweep variance volume + 3-D U-Net per working view (init_net.py:113-160), `res_net`, image / visibility encoders, then
    the coarse + fine ray-batch loop.  Synthetic smooth images, seeded random weights (no datasets / checkpoints here).
    -> rays/s and ms per image for the whole call, and for the per-image networks alone."""
    from neuray_amd.network.renderer import NeuralRayGenRenderer
    hp, wp = (h + 31) // 32 * 32, (w + 31) // 32 * 32
    cfg = {'init_net_type': 'cost_volume', 'use_hierarchical_sampling': True, 'use_depth_loss': True, 'dist_decoder_cfg': {'use_vis': False},
           'ray_batch_num': RAY_BATCH, 'hip_arith': arith}
    torch.manual_seed(0)
    model = NeuralRayGenRenderer(cfg).eval().to(device)
    n_src = rfn + extra_src
    que, scene = synthetic.make_scene(hp, wp, n_src, seed=3, que_imgs=False, smooth=True, depth_range=(1.2, 12.0))
    cams = {k: torch.from_numpy(scene[k]).to(device) for k in ('imgs', 'poses', 'Ks', 'depth_range')}
    src = dict(cams)
    ref = {k: v[:rfn].contiguous() for k, v in cams.items()}
    ref['nn_ids'] = torch.tensor([[(v + 1) % n_src, (v + 2) % n_src, (v + 5) % n_src] for v in range(rfn)], device=device)
    tq = {k: torch.from_numpy(v).to(device) for k, v in que.items() if k not in ('ray_feats', 'imgs')}
    tq['coords'] = torch.from_numpy(synthetic.meshgrid_coords(h, w)).to(device)
    data = lambda: {'que_imgs_info': dict(tq), 'ref_imgs_info': dict(ref), 'src_imgs_info': dict(src), 'eval': True}      # noqa: E731

    def whole():
        with torch.no_grad():
            return model(data())
    out = whole()
    whole()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = whole()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    # the per-image networks alone: cost-volume init net, then the encoders on its output
    def nets():
        with torch.no_grad():
            r = dict(ref)
            r['ray_feats'] = model.init_net(r, dict(src), False)
            r['img_feats'] = model.image_encoder(r['imgs'])
            r['ray_feats'] = model.vis_encoder(r['ray_feats'], r['img_feats'])
            return r
    nets()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        nets()
    torch.cuda.synchronize(device)
    dn = (time.perf_counter() - t0) / steps
    from neuray_amd.network import render_ops as _ro
    _ro.check_deferred_inputs(device, wait=True)
    px = out['pixel_colors_nr_fine']
    return {'what': 'BASELINE.json configs[2] shape: neuray_gen_cost_volume on an llff fern/high-shaped scene - %d x %d query (%d rays), %d working views '
                    'padded to %d x %d, cost-volume init net (3 neighbours of %d source views, 64 planes) + res_net + encoders + 64+64 render, '
                    'renderer(data) in evaluation; synthetic smooth images, random weights' % (h, w, h * w, rfn, hp, wp, n_src),
            'value': h * w / dt, 'unit': 'rays/s', 'ms_per_image': 1e3 * dt, 'arith': arith,
            'per_image_networks_ms': 1e3 * dn, 'render_loop_ms': 1e3 * (dt - dn),
            'rays_per_s_of_the_render_loop_alone': h * w / max(dt - dn, 1e-9),
            'finite': bool(torch.isfinite(px).all().item()), 'pixels': list(px.shape)}


def direct_rendering_timing(device, tq, tr, steps=2):
    """Side measurement: the image with cfg['use_dr_prediction'] (renderer.py:85-125: the point kernel's
    per-view-record instantiation - which computes every slot - plus dr_points_kernel / dr_rays_kernel in both passes).  Off in every
    shipped config; timed once so that its cost is a number."""
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': DN_COARSE,
           'fine_depth_sample_num': 32, 'agg_net_cfg': {'sample_num': DN_COARSE}, 'fine_agg_net_cfg': {'sample_num': 32},
           'ray_batch_num': RAY_BATCH, 'use_dr_prediction': True}
    torch.manual_seed(0)
    r = NeuralRayBaseRenderer(cfg).eval().to(device)
    out = render_image(r, tq, tr)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = render_image(r, tq, tr)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    return {'what': 'same image with use_dr_prediction (per-view record + SH(16) fit per sample point + second compositing), 64+32',
            'value': steps * H * W / dt, 'unit': 'rays/s', 'ms_per_image': 1e3 * dt / steps,
            'finite': bool(torch.isfinite(out['pixel_colors_dr_fine']).all().item())}


def train_ddp_leg(device, world, test_lib, steps, rays=512):
    """N > 1 side leg: data-parallel training (the reference has none, train/trainer.py:65-70).  Every rank runs
    render_impl(is_train=True) + backward on its OWN `rays` rays through the HIP kernels, then neuray_amd.parallel
    sums the gradients of the shared networks with ONE flattened all-reduce (RCCL over xGMI) and Adam steps."""
    import torch.distributed as dist
    from neuray_amd import parallel
    small = test_lib is not None
    dn = 8 if small else 64
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': dn,
           'fine_depth_sample_num': dn, 'agg_net_cfg': {'sample_num': dn}, 'fine_agg_net_cfg': {'sample_num': dn}}
    torch.manual_seed(0)
    r = NeuralRayBaseRenderer(cfg).train().to(device)
    r._engine_test_lib = test_lib
    h, w = (32, 48) if small else (400, 600)
    rays = 6 if small else rays
    rank = dist.get_rank()
    que, ref = synthetic.make_scene(h, w, 3 if small else 8, seed=rank)
    rng = np.random.RandomState(rank)
    que['coords'] = (rng.rand(1, rays, 2) * np.array([w - 1, h - 1])).astype(np.float32)
    tq = {k: torch.from_numpy(v).to(device) for k, v in que.items()}
    tr = {k: torch.from_numpy(v).to(device) for k, v in ref.items()}
    tgt = torch.rand(1, rays, 3, device=device)
    opt = torch.optim.Adam(r.parameters(), lr=1e-4)
    sync = (lambda: torch.cuda.synchronize(device)) if device.type == 'cuda' else (lambda: None)

    def step():
        return parallel.train_step(r, None, lambda out: ((out['pixel_colors_nr'] - tgt) ** 2).mean() +
                                   ((out['pixel_colors_nr_fine'] - tgt) ** 2).mean(), opt)
    r.forward = lambda _data: r.render_impl(tq, tr, True)
    step()
    sync(); dist.barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync(); dist.barrier(); sync()
    dt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    first = torch.stack([p.detach().double().sum() for p in r.parameters()]).sum().reshape(1)      # checksum of every parameter
    lo, hi = first.clone(), first.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    nparam = sum(p.numel() for p in r.parameters() if p.requires_grad)
    return {'what': 'forward + backward + ONE flattened gradient all-reduce + Adam per step, %d rays per rank, %d+%d samples' % (rays, dn, dn),
            'world_size': world, 'ms_per_step': 1e3 * float(dt.item()) / steps, 'value': world * rays * steps / float(dt.item()),
            'unit': 'rays/s (training)', 'allreduce_bytes_per_step': 4 * nparam, 'replicas_identical_after_steps': bool(lo.item() == hi.item())}


def split_image_leg(renderer, tq, tr, device, world, steps):
    """N > 1 side leg: ONE image split over the ranks (contiguous ray ranges) + one fused all-gather of the tiles."""
    import torch.distributed as dist
    sync = (lambda: torch.cuda.synchronize(device)) if device.type == 'cuda' else (lambda: None)
    render_image(renderer, tq, tr, True)
    sync(); dist.barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = render_image(renderer, tq, tr, True)
    sync(); dist.barrier(); sync()
    dt = torch.tensor([time.perf_counter() - t0], device=device, dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    return {'what': 'one %dx%d image split over %d ranks, one fused all-gather of the rendered tiles per image' % (H, W, world),
            'scaling': 'strong', 'value': steps * H * W / float(dt.item()), 'unit': 'rays/s', 'ms_per_image': 1e3 * float(dt.item()) / steps,
            'gathered_rays': int(out['pixel_colors_nr_fine'].shape[1])}


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this file under torch.distributed.run."""
    import subprocess
    if not args.emulator_lib:
        seen = torch.cuda.device_count()
        if seen < args.gpus:
            sys.exit("bench.py: --gpus %d but only %d GPU(s) visible - refusing to report a line with the wrong n_gpus" % (args.gpus, seen))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % args.gpus,
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    sys.exit(subprocess.call(cmd))


FRESH_PROCESS_LEGS = {'ft_step': ft_step_timing, 'gen_train_step': gen_train_step_timing}


def main(argv=None):
    global H, W, RFN, DN_COARSE, RAY_BATCH
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--fine-samples', type=int, default=32, help='32 = BASELINE.json wording, 64 = reference default')
    ap.add_argument('--arith', default=None, choices=['x3', 'f32'],
                    help="arithmetic of the point kernel's MLP contractions in the timed region (cfg['hip_arith']): x3 = operands split exactly "
                         "into three bf16 parts on the K = 32 bf16 MFMA, fp32 accumulation (each product within 2^-23 of exact); f32 = the "
                         "fp32 MFMA.  Default x3; at N = 1 the fp32-MFMA line is timed next to it (`fp32_mfma`)")
    ap.add_argument('--cpu-sample-rays', type=int, default=8192, help='rays of the numpy-oracle parity leg')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-eager-baseline', action='store_true', help='skip every side leg (baselines, extras)')
    ap.add_argument('--split-image', action='store_true',
                    help='N > 1: make the TIMED region the split of ONE image over the ranks + all-gather of the tiles '
                         '(strong scaling) instead of one image per rank (weak scaling, no collective)')
    ap.add_argument('--no-side-legs', action='store_true', help='N > 1: skip the split_image / train_ddp side legs')
    ap.add_argument('--fail-leg-on-rank', type=int, default=-1, help='TEST HOOK: that rank raises inside the first side leg')
    ap.add_argument('--side-leg-timeout', type=int, default=300, help='N > 1: seconds after which the line is printed without the side legs')
    ap.add_argument('--emulator-lib', default=None, help='TEST HOOK: CPU emulator build of the kernels, gloo, tiny image')
    ap.add_argument('--size', type=int, nargs=2, default=None, metavar=('H', 'W'), help='test hook: image size (default 800 800)')
    ap.add_argument('--leg', default=None, help='internal: run ONE training-step side leg in this (fresh) process and print its JSON')
    args = ap.parse_args(argv)
    if args.arith is None:                                         # (the emulator smoke of the launcher path keeps the cheaper arithmetic)
        args.arith = 'f32' if args.emulator_lib else 'x3'
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # multi-process GPU work: the host driver only supports dmabuf IPC (RCCL)
    if args.leg:
        print(json.dumps({'leg': args.leg, 'result': FRESH_PROCESS_LEGS[args.leg](torch.device('cuda', 0))}), flush=True)
        return

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        self_launch(args, argv)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d (or with no "
                 "launcher at all)" % (args.gpus, world, args.gpus))
    emu = None
    if args.emulator_lib:
        from neuray_amd import _lib
        emu = _lib.bind(args.emulator_lib)
        assert emu.neuray_is_device_build() == 0
        H, W = args.size or (16, 24)
        RFN, DN_COARSE, RAY_BATCH, args.fine_samples = 3, 8, 64, 8
        device = torch.device('cpu')
    else:
        if args.size:
            H, W = args.size
        if torch.cuda.device_count() <= local_rank:
            sys.exit("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()))
        device = torch.device('cuda', local_rank)
        torch.cuda.set_device(device)
    sync = (lambda: torch.cuda.synchronize(device)) if device.type == 'cuda' else (lambda: None)
    if world > 1:
        import torch.distributed as dist
        if emu is not None:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)
        seen = torch.tensor([1.0], device=device)
        dist.all_reduce(seen)                      # every rank that joined the group counts itself: the world size RCCL saw
        world_seen = int(seen.item())
        assert world_seen == world == dist.get_world_size()
    else:
        world_seen = 1

    split = args.split_image and world > 1
    cfg, renderer, weights, que, ref, tq, tr = build_case(device, args.fine_samples, seed=0 if split else rank, test_lib=emu, arith=args.arith)
    eng = renderer.engine(device)
    nrays = H * W

    for _ in range(args.warmup):
        out = render_image(renderer, tq, tr, split)

    def fence():
        sync()
        if world > 1:
            dist.barrier()
            sync()

    # the share of (tile, view) slots the point kernel runs / skips: counted by the kernel itself in ONE extra image outside the timed
    # region (deterministic per image; the counters cost two atomics per workgroup, which the timed launches do not pay)
    slot_stats = torch.zeros(2, dtype=torch.int64, device=device) if device.type == 'cuda' else None
    eng.slot_stats = slot_stats
    render_image(renderer, tq, tr, split)
    eng.slot_stats = None
    eng.timing = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = render_image(renderer, tq, tr, split)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    timing, eng.timing = eng.timing, None

    line = None
    if rank == 0:
        value = (1 if split else world) * args.steps * nrays / dt
        standard = (H, W, RFN, DN_COARSE) == (800, 800, 8, 64) and emu is None
        line = {
            'metric': 'rays/sec (64 coarse+%d fine samples), lego 800x800 synthetic' % args.fine_samples,
            'value': value, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'strong' if split else 'weak',
            'vs_baseline': None,
            'dtype': 'fp32' if args.arith == 'f32' else 'fp32 (MLP contractions: operands split exactly into 3 bf16 parts, 6 bf16-MFMA products per K = 32, fp32 accumulate)',
            'arith': args.arith, 'data': 'synthetic' if emu is None else 'synthetic (CPU EMULATOR smoke of the launcher path - NOT a measurement)',
            'world_size_seen_by_process_group': world_seen,
            'config': {'workload': 'lego-800 synthetic (nerf_synthetic/lego/black_800 shape): %dx%d = %d rays/image, '
                                   '%d ref views, %d coarse + %d fine samples, maps %dx%dx32, 1 image per step per GPU'
                                   % (H, W, nrays, RFN, DN_COARSE, args.fine_samples, H // 4, W // 4) +
                                   ('' if standard else ' [NON-STANDARD SIZE: test hook]'),
                       'ray_batch': cfg['ray_batch_num'], 'parallelism': ('one image split over %d GPUs, all-gather of tiles' % world) if split else
                                      ('images sharded over %d GPU(s), no collective' % world)},
        }
        if device.type == 'cuda':
            # dominant kernel: the point kernel.  Duration from HIP events on the launch stream.
            pts = [(e0.elapsed_time(e1) * 1e-3, n) for name, e0, e1, n in timing if name == 'points']
            rays_k = [e0.elapsed_time(e1) * 1e-3 for name, e0, e1, n in timing if name == 'rays']
            t_pts = sum(t for t, _ in pts)
            n_pts = sum(n for _, n in pts)
            # FLOP accounting: `achieved` counts what the kernel EXECUTES - the folded network (prob_embed.2 folded into
            # its consumers at pack time) on the (tile, view) slots that were not skipped as fully masked (counted by the kernel itself:
            # NeurayPointsArgs.slot_stats_dev) - so that skipping work does not inflate the fraction of the matrix peak; the rate the same
            # launches represent in terms of the reference's unfolded network on every (point, view) is reported next to it.
            st = slot_stats.cpu().numpy() if slot_stats is not None else None
            share = float(st[0] / st[1]) if st is not None and st[1] > 0 else 1.0
            flops_exec = 2.0 * algorithmic_macs_per_point(RFN, vis_head_used=False, folded=True, slots_run_share=share)
            flops_folded = 2.0 * algorithmic_macs_per_point(RFN, vis_head_used=False, folded=True)
            flops_ref = 2.0 * algorithmic_macs_per_point(RFN, vis_head_used=False, folded=False)
            # HBM/fabric traffic of the point kernel per launch: PMC passes are run separately (rocprofv3 --pmc cannot run
            # inside this process); the committed summary of the same command is reported here (profiles/README.md)
            traffic, tsrc = None, None
            import glob
            import re

            def tkey(f):         # r01_traffic < r01_e_traffic < r01_zz_traffic < r02_a_traffic: by round, then by tag (length first)
                m = re.match(r'r(\d+)_?(.*?)_?traffic\.json$', os.path.basename(f))
                return (int(m.group(1)), len(m.group(2)), m.group(2)) if m else (-1, 0, os.path.basename(f))
            tfiles = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')), key=tkey)
            if args.fine_samples == 32 and standard and tfiles:
                tj = json.load(open(tfiles[-1]))
                sys.path.insert(0, os.path.join(ROOT, 'profiles'))
                from summarize_pmc import kernel_source_hash
                if tj.get('kernel_source_sha16') == kernel_source_hash():
                    traffic = tj.get('bytes_per_launch')
                    tsrc = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes/launch; collected on tree %s; the hash of the ' \
                           'kernel sources stored in it matches this tree)' % (os.path.basename(tfiles[-1]), tj.get('commit', 'of that profile run'))
                else:           # the kernel changed since the counters were collected: a stale figure is not reported
                    tsrc = 'null: profiles/%s was collected on other kernel sources (sha16 %s, this tree %s) - re-run profiles/collect.sh' % (
                        os.path.basename(tfiles[-1]), tj.get('kernel_source_sha16'), kernel_source_hash())
            achieved = flops_exec * n_pts / t_pts / 1e12
            line['roofline'] = {'bound': 'mfma', 'achieved': achieved, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                'frac': achieved / MFMA_F32_PEAK_TFLOPS, 'traffic': traffic, 'traffic_source': tsrc,
                                'kernel': 'nr::points_kernel<..., %s>' % ('AR_X3' if args.arith == 'x3' else 'AR_F32'), 'launches': len(pts),
                                **({'issued_bf16_mfma_tflops': 6.0 * achieved, 'bf16_mfma_dense_peak_tflops': 2500.0,
                                    'frac_of_bf16_dense_peak_issued': 6.0 * achieved / 2500.0,
                                    'matrix_pipe_busy_pmc': '0.28 (SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles, profiles/r06_zy_pmc_summary.json; fp32 kernel: 0.51)'}
                                   if args.arith == 'x3' else {}),
                                'peak_note': 'fp32-in MFMA dense peak (MI355X_MICROARCH.md): the peak of the arithmetic GRADE the kernel delivers' +
                                             ('.  With arith = x3 the contractions run on the bf16 pipe (dense peak 2500 TFLOP/s): six bf16 MFMA products per '
                                              'fp32-grade product, i.e. %.0f TFLOP/s of bf16 MFMA work issued = %.3f of that peak; the kernel is bound by VALU '
                                              'issue (operand splits, ELU, blends), not by the matrix pipe (DESIGN.md 4.12)'
                                              % (6.0 * achieved, 6.0 * achieved / 2500.0) if args.arith == 'x3' else ''),
                                'avg_launch_ms': 1e3 * t_pts / max(1, len(pts)),
                                'flops_counted': 'executed: folded network (prob_embed.2 folded into neuray_fc.0 / base_fc.0), per-view layers only on '
                                                 'the (16-point tile, view) slots the kernel ran (fully masked slots are skipped)',
                                'view_slots_run_share': share,
                                'executed_flops_per_point': flops_exec,
                                'algorithmic_flops_per_point': flops_folded,
                                'algorithmic_flops_per_point_unfolded_reference_network': flops_ref,
                                'equivalent_tflops_of_the_unfolded_network_on_every_view': flops_ref * n_pts / t_pts / 1e12,
                                # 1,072 B per (point, view) = 4 taps x (32 + 32 + 3 channels) x 4 B; the kernel's texels are RGBA (a
                                # 4th, unused colour channel: 1,088 B actually requested), reported separately
                                'gather_demand_tb_per_s': 1072.0 * RFN * n_pts * share / t_pts / 1e12,
                                'gather_requested_tb_per_s_rgba_texels': 1088.0 * RFN * n_pts * share / t_pts / 1e12,
                                'gather_demand_note': "north_star's '>= 40 % HBM roofline on feature gather' = 3.2 TB/s of demand = 4.2 M rays/s at 64+32 samples.  The "
                                                      "fp32 MFMA bound of the folded network (33.25 MFLOP/ray at 157.3 TFLOP/s) is ~3.7 M rays/s; the x3 arithmetic lifts "
                                                      "that bound (six bf16 MFMAs per K = 32: 0.28 of the matrix pipe busy) and is bound by VALU issue instead - operand "
                                                      "splits, ELU, blends, statistics (DESIGN.md 4.12).  The maps are L2 / Infinity-Cache resident (roofline.traffic), so "
                                                      "the demand rate is a cache-path figure, not an HBM one",
                                'point_kernel_share_of_step': t_pts / dt, 'ray_kernel_share_of_step': sum(rays_k) / dt}
        else:
            line['roofline'] = None

    # ---- N > 1 side legs: collectives, every rank takes part.  They run AFTER the headline is in hand and under a watchdog: a rank
    # that fails inside a leg leaves the others waiting in a collective, and that must cost the side legs, never the line.
    emitted, emit_lock = threading.Event(), threading.Lock()

    def emit():
        with emit_lock:
            if not emitted.is_set():
                emitted.set()
                if rank == 0:
                    print(json.dumps(line))
                    sys.stdout.flush()

    def put(key, val):
        with emit_lock:
            if line is not None and not emitted.is_set():
                line[key] = val

    watchdog = None
    if world > 1:
        limit = args.side_leg_timeout

        def expired():
            if rank == 0:
                with emit_lock:                      # (the main thread only touches `line` under the same lock: put())
                    if not emitted.is_set():
                        line['side_legs_error'] = 'a multi-rank side leg (or the final barrier) did not finish within %d s; the headline above is complete' % limit
            emit()
            os._exit(0)                              # the headline is out; the launcher must not discard it over a lost side leg
        watchdog = threading.Timer(limit, expired)
        watchdog.daemon = True
        watchdog.start()
    if world > 1 and not args.no_side_legs:
        legs = {}
        if not split:
            c2, r2, _, _, _, tq2, tr2 = build_case(device, args.fine_samples, seed=0, test_lib=emu)
            def split_leg(*a):
                if args.fail_leg_on_rank == rank:          # TEST HOOK (tests/test_bench_launcher.py): the other ranks are left in a collective
                    raise RuntimeError('test hook: rank %d fails inside the side leg' % rank)
                return split_image_leg(*a)
            legs['split_image'] = side(split_leg, r2, tq2, tr2, device, world, max(1, min(args.steps, 3)))
        if args.fail_leg_on_rank == rank:
            time.sleep(10 ** 6)                            # (... and never joins another one: only the watchdog ends this rank)
        legs['train_ddp'] = side(train_ddp_leg, device, world, emu, 2 if emu is not None else 5)
        if rank == 0:
            for k_, v_ in legs.items():
                put(k_, v_)

    others = {}
    if rank == 0 and world == 1 and emu is None and device.type == 'cuda':
        if args.arith != 'f32':       # the fp32-MFMA line, next to the headline either way (VERDICT r5 credit rule)
            res = side(fp32_mfma_leg, device, args.fine_samples, tq, tr, args.steps, value)
            if isinstance(res, tuple):
                put('fp32_mfma', res[0])
                others['ours_fp32_mfma'] = res[1]
            else:
                put('fp32_mfma', res)
    if rank == 0:
        if world == 1 and emu is None and not args.no_eager_baseline and not args.no_cpu_baseline:
            # the whole-step training legs first: they are host-sensitive (host time ~ device time), and after the CPU baselines' minute
            # of all-core work below the host's single-thread speed stays depressed for a while - measured after them the generalisation
            # step read 39-47 ms where a quiet machine gives 33-34 (fresh_process_leg)
            put('ft_step', side(fresh_process_leg, 'ft_step', device))
            put('gen_train_step', side(fresh_process_leg, 'gen_train_step', device))
        if world == 1 and emu is None and not args.no_cpu_baseline:   # baselines: rank 0 at N = 1 only
            put('cpu_baseline', side(cpu_baseline, cfg, weights, que, ref))
            if args.cpu_sample_rays > 0:
                res = side(numpy_oracle_leg, cfg, weights, que, ref,
                           {k: out[k].cpu().numpy() for k in ('pixel_colors_nr', 'pixel_colors_nr_fine')}, args.cpu_sample_rays, 1024)
                if isinstance(res, tuple):
                    put('numpy_oracle', res[0])
                    parity = res[1]
                    if not args.no_eager_baseline:
                        # all 640 000 rays against the eager port on this GPU (the numpy oracle above: 1.3 % of them)
                        parity['full_image'] = side(full_image_parity_leg, cfg, renderer, weights, que, ref, tq, tr, out, device, others=others)
                    put('parity', parity)
                else:
                    put('numpy_oracle', res)
        if world == 1 and emu is None and not args.no_eager_baseline and not args.no_cpu_baseline:
            eb = side(eager_torch_baseline, cfg, weights, tq, tr, device)
            if 'value' in eb:
                eb['speedup_vs_eager'] = value / eb['value']
            put('eager_torch_baseline', eb)
            put('extra', {
                'reference_default_64+64': side(extra_config_timing, device, 64, RAY_BATCH, tq, tr, 2, args.arith),
                'reference_cli_ray_batch_4096': side(extra_config_timing, device, args.fine_samples, 4096, tq, tr, 2, args.arith),
                'llff_fern_cost_volume': side(llff_fern_cost_volume_timing, device, args.arith),
            })
            put('direct_rendering', side(direct_rendering_timing, device, tq, tr))
            put('training_step', side(training_step_timing, device))
            put('encoders', side(encoder_timing, device))
            put('init_net', side(init_net_timing, device))
            put('pipeline_pcie_inclusive', side(pipeline_timing, device, args.fine_samples, 3, 16, args.arith))
            put('bf16_variant', side(bf16_variant_timing, device, args.fine_samples, tq, tr, out['pixel_colors_nr_fine'].cpu().numpy()))
            put('bf16x3_split_variant', side(bf16_variant_timing, device, args.fine_samples, tq, tr, out['pixel_colors_nr_fine'].cpu().numpy(),
                                             2, 'bf16x3'))
    if rank == 0 and line is not None:
        # the scalars a reader of the LAST kilobytes of this (long) line needs: the driver keeps only the tail of stdout
        def pick(d, *path):
            for k_ in path:
                d = d.get(k_) if isinstance(d, dict) else None
            return d
        put('summary', {
            'value_rays_per_s': line['value'], 'arith': args.arith, 'n_gpus': world, 'ms_per_step': line['ms_per_step'],
            'roofline_frac': pick(line, 'roofline', 'frac'), 'point_kernel_avg_launch_ms': pick(line, 'roofline', 'avg_launch_ms'),
            'fp32_mfma_rays_per_s': pick(line, 'fp32_mfma', 'value'), 'fp32_mfma_roofline_frac': pick(line, 'fp32_mfma', 'roofline', 'frac'),
            'fp32_mfma_point_kernel_avg_launch_ms': pick(line, 'fp32_mfma', 'point_kernel_avg_launch_ms'),
            'parity_pass': pick(line, 'parity', 'pass'), 'full_image_pass': pick(line, 'parity', 'full_image', 'pass'),
            'vs_float64_pass': pick(line, 'parity', 'full_image', 'vs_float64', 'pass'),
            'ft_step_ms': pick(line, 'ft_step', 'ms_per_step'), 'gen_train_step_ms': pick(line, 'gen_train_step', 'ms_per_step'),
            'point_backward_ms_per_pass': pick(line, 'training_step', 'point_backward', 'ms_per_pass'),
            'cost_volume_init_net_ms': pick(line, 'init_net', 'cost_volume_init_net_ms'),
            'encoders_forward_backward_ms': pick(line, 'encoders', 'fused_norm', 'forward_backward_ms'),
            'encoders_forward_backward_ms_miopen_convs': pick(line, 'encoders', 'fused_norm_miopen_convs', 'forward_backward_ms'),
            'encoders_speedup_forward_backward': pick(line, 'encoders', 'speedup_forward_backward'),
            'cpu_baseline_rays_per_s': pick(line, 'cpu_baseline', 'value'), 'eager_torch_rays_per_s': pick(line, 'eager_torch_baseline', 'value'),
            'config3_llff_rays_per_s': pick(line, 'extra', 'llff_fern_cost_volume', 'value'),
        })
    emit()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
        watchdog.cancel()


if __name__ == '__main__':
    main()
