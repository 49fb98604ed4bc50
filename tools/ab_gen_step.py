"""Same-process A/B of the generalisation training step (bench.gen_train_case) with cfg switches flipped between blocks of steps:

    python tools/ab_gen_step.py [--blocks 4] [--steps 15] key=a,b [key=a,b ...]

e.g. hip_prefetch_depth_coords=1,0.  Blocks alternate A, B, A, B ...; per variant the median block time per step (host-bound steps
drift with the box: interleaving in one process is what makes two numbers comparable)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--blocks', type=int, default=4)
    ap.add_argument('--steps', type=int, default=15)
    ap.add_argument('--host-ks-inv', action='store_true')
    ap.add_argument('switch', nargs='+')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    model, opt, step = bench.gen_train_case(dev, host_ks_inv=a.host_ks_inv)
    sw = [(s.split('=')[0], [json.loads(v) for v in s.split('=')[1].split(',')]) for s in a.switch]
    nvar = len(sw[0][1])
    for _ in range(25):
        step()
    torch.cuda.synchronize()
    times = [[] for _ in range(nvar)]
    for b in range(a.blocks * nvar):
        v = b % nvar
        for k, vals in sw:
            model.cfg[k] = vals[v]
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        times[v].append(1e3 * (time.perf_counter() - t0) / a.steps)
    for v in range(nvar):
        print({k: vals[v] for k, vals in sw}, 'ms/step median %.2f' % np.median(times[v]), ['%.2f' % t for t in times[v]])


if __name__ == '__main__':
    main()
