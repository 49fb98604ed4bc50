"""One generalisation-training step at the shape of BASELINE.json configs[4] (bench.gen_train_case: NeuralRayGenRenderer + cost-volume
init net + encoders + render / depth loss + Adam, 512 rays, 8 views of 416 x 608): wall time, host cProfile, and - under
`rocprofv3 --kernel-trace --stats` (profiles/collect_gen_step.sh) - the GPU time per step by kernel class.

    python tools/profile_gen_step.py [--steps 25]
    python tools/profile_gen_step.py --classify <kernel_stats.csv> <steps>"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import bench                                   # noqa: E402
from profile_ft_step import classify           # noqa: E402


def main():
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 25
    dev = torch.device('cuda', 0)
    model, opt, step = bench.gen_train_case(dev)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps - 15):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = steps - 15
    print('%d steps: host %.1f ms/step, with drain %.1f ms/step' % (n, 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(24)
    pstats.Stats(pr).sort_stats('tottime').print_stats(30)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--classify':
        classify(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 25)
    else:
        main()
