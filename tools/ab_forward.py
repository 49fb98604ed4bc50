"""A/B of kernel-library builds on the bench workload (800 x 800, 8 views, 64 + 32), on the MI355X:

    python tools/ab_forward.py [--steps 3] [--parity 8192] name=path/to/lib.so [name=path ...]

For every library: rays/s, point-kernel ms per launch (HIP events of the engine), and - with --parity N - the error of the
COARSE pixels / hit probabilities and of the chained fine pixels on N strided rays of the image against the numpy oracle
(TEST INFRASTRUCTURE: the oracle is the checker here, computed once).  `build:` entries compile a variant first:
    python tools/ab_forward.py base=neuray_amd/libneuray_hip.so norefine=build:-DNR_FEATURE_RCP_REFINE=0
(variants are written to _ab/, which is git-ignored).  A path may carry `,nofold`: that library renders with the unfolded
pack (cfg hip_fold_prob_embed = False; also what a library older than ABI 7 needs); `,x3`: cfg hip_arith = 'x3'."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                     # noqa: E402
from neuray_amd import _lib, build as nbuild                      # noqa: E402
from neuray_amd.network.renderer import NeuralRayBaseRenderer    # noqa: E402


def build_variant(name, flags):
    out = os.path.join(ROOT, '_ab', 'lib_%s.so' % name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call([nbuild.HIPCC] + nbuild.FLAGS + flags + nbuild.SOURCES + ['-o', out])
    return out


def bind_compat(path):
    """_lib.bind that tolerates a library from before the current ABI (symbols it lacks stay unbound; the argument structs only
    grew at their ends, which an older library does not read)"""
    import ctypes as C
    lib = C.CDLL(path)
    for name, (restype, argtypes) in _lib.SYMBOLS.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype, fn.argtypes = restype, argtypes
    return lib


def oracle_sample(cfg, weights, que, ref, n):
    from oracle import neuray_oracle as orc
    idx = np.linspace(0, que['coords'].shape[1] - 1, n).astype(np.int64)
    ocfg = {**orc.DEFAULT_CFG, **cfg, 'coarse_use_vis': False, 'fine_use_vis': True}
    keys = ('pixel_colors_nr', 'hit_prob_nr', 'pixel_colors_nr_fine')
    acc = {k: [] for k in keys}
    for i in range(0, n, 1024):
        q = dict(que)
        q['coords'] = que['coords'][:, idx[i:i + 1024]]
        o = orc.render_impl(weights, ocfg, q, ref)
        for k in keys:
            acc[k].append(o[k])
    return idx, {k: np.concatenate(v, 1) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--parity', type=int, default=0)
    ap.add_argument('--order', default='row', help="ray order of the image: 'row' (meshgrid, the caller's), 'morton' (Z-order), 'tile8x2', 'tile4x4'")
    ap.add_argument('--ray-batch', type=int, default=0, help='rays per render_impl call (default: bench.RAY_BATCH = 32768; render.py uses 4096)')
    ap.add_argument('libs', nargs='+')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    cfg, r0, weights, que, ref, tq, tr = bench.build_case(dev, 32, seed=0)
    if a.order != 'row':             # timing experiment: the same rays in another order (results per ray are order-independent)
        xy = tq['coords'][0].long()
        x, y = xy[:, 0], xy[:, 1]
        if a.order == 'morton':
            def spread(v):
                v = (v | (v << 8)) & 0x00FF00FF
                v = (v | (v << 4)) & 0x0F0F0F0F
                v = (v | (v << 2)) & 0x33333333
                return (v | (v << 1)) & 0x55555555
            key = spread(x) | (spread(y) << 1)
        else:
            bw, bh = (8, 2) if a.order == 'tile8x2' else (4, 4)
            key = ((y // bh) * (bench.W // bw) + x // bw) * (bw * bh) + (y % bh) * bw + x % bw
        perm = torch.argsort(key)
        tq['coords'] = tq['coords'][:, perm].contiguous()
    want = None
    if a.parity:
        t0 = time.perf_counter()
        idx, want = oracle_sample(cfg, weights, que, ref, a.parity)
        print('oracle: %d rays in %.1f s' % (a.parity, time.perf_counter() - t0), flush=True)
    res = {}
    for item in a.libs:
        name, path = item.split('=', 1)
        path, *opts = path.split(',')
        fold = 'nofold' not in opts
        arith = 'x3' if 'x3' in opts else 'f32'           # `,x3`: NEURAY_ARITH_X3 (three-way split bf16 operands on the K = 32 MFMA)
        if path.startswith('build:'):
            path = build_variant(name, path[6:].split())
        torch.manual_seed(0)
        r = NeuralRayBaseRenderer({**cfg, 'hip_fold_prob_embed': fold, 'hip_arith': arith, **({'ray_batch_num': a.ray_batch, 'hip_min_ray_batch': 0} if a.ray_batch else {})}).eval()
        r.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=True)
        r = r.to(dev)
        r._engine_test_lib = bind_compat(os.path.join(ROOT, path) if not os.path.isabs(path) else path)
        eng = r.engine(dev)
        stats = torch.zeros(2, dtype=torch.int64, device=dev)
        eng.slot_stats = stats                       # (the warm-up image counts the skipped slots; the timed ones run without the counters)
        out = bench.render_image(r, tq, tr)
        eng.slot_stats = None
        eng.timing = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = bench.render_image(r, tq, tr)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        pts = [e0.elapsed_time(e1) for nm, e0, e1, n in eng.timing if nm == 'points']
        rays = [e0.elapsed_time(e1) for nm, e0, e1, n in eng.timing if nm == 'rays']
        eng.timing = None
        st = stats.cpu().numpy()
        res[name] = {'rays_per_s': a.steps * 640000 / dt, 'point_kernel_ms': float(np.mean(pts)), 'ray_kernel_ms': float(np.mean(rays)),
                     'fold': fold, 'slots_run_share': float(st[0] / st[1]) if st[1] else None}
        if want is not None:
            sel = torch.from_numpy(idx).to(dev)
            q = {k: v for k, v in tq.items() if not k.startswith('_')}
            got = {}
            with torch.no_grad():
                for i in range(0, len(idx), 4096):
                    q['coords'] = tq['coords'][:, sel[i:i + 4096]]
                    for k, v in r.render_impl(q, tr, False).items():
                        got.setdefault(k, []).append(v.cpu().numpy())
            got = {k: np.concatenate(v, 1) for k, v in got.items()}
            ec = np.abs(got['pixel_colors_nr'] - want['pixel_colors_nr']).max(-1)[0]
            eh = np.abs(got['hit_prob_nr'] - want['hit_prob_nr']).max(-1)[0]
            ef = np.abs(got['pixel_colors_nr_fine'] - want['pixel_colors_nr_fine']).max(-1)[0]
            pc = lambda e, q_: float(np.percentile(e, q_))       # noqa: E731
            res[name]['parity'] = {
                'coarse_pixel': {'max': float(ec.max()), 'p99.9': pc(ec, 99.9), 'p99': pc(ec, 99), 'median': pc(ec, 50)},
                'coarse_hit_prob': {'max': float(eh.max()), 'p99.9': pc(eh, 99.9), 'median': pc(eh, 50)},
                'chained_fine_pixel': {'max': float(ef.max()), 'frac_within_2e-4': float(np.mean(ef <= 2e-4)), 'p99': pc(ef, 99), 'median': pc(ef, 50)}}
        print(name, json.dumps(res[name]), flush=True)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
