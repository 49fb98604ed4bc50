"""Where do the small element-wise launches of a generalisation training step come from (fills, copies, adds: ~500 launches of a step's
1 500)?  torch.profiler over a few steps; per (aten op, nearest neuray_amd / bench.py frame, or the autograd node for backward-thread ops):
launches per step and device time per step.
    python tools/profile_gen_small_ops.py [--ft]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                      # noqa: E402

WATCH = ('aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::add', 'aten::add_', 'aten::mul', 'aten::mul_', 'aten::sub', 'aten::div',
         'aten::cat', 'aten::index', 'aten::sum', 'aten::where', 'aten::clamp', 'aten::neg', 'aten::exp', 'aten::abs', 'aten::sqrt',
         'aten::rsub', 'aten::_to_copy', 'aten::clone', 'aten::index_select', 'aten::gather', 'aten::mean', 'aten::pow', 'aten::reciprocal')


def main():
    dev = torch.device('cuda', 0)
    if '--ft' in sys.argv:
        raise SystemExit("the fine-tuning step has no stand-alone case in bench.py: use tools/profile_ft_step.py")
    else:
        step = bench.gen_train_case(dev)[-1]
    for _ in range(12):
        step()
    torch.cuda.synchronize()
    n = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if ev.name not in WATCH or ev.device_time_total <= 0 or not ev.kernels:
            continue
        # the chain of enclosing CPU events: the autograd node being evaluated (backward thread) or the outermost aten op (forward)
        chain, par = [], ev.cpu_parent
        while par is not None:
            chain.append(par.name)
            par = par.cpu_parent
        node = [c for c in chain if c.startswith('autograd::engine::evaluate_function: ')]
        if node:
            where = 'backward of ' + node[-1].split(': ', 1)[1] + (' > ' + chain[0] if chain and not chain[0].startswith('autograd::') else '')
        else:
            where = 'forward' + (' > ' + ' > '.join(reversed(chain[-2:])) if chain else '')
        a = agg[(ev.name, where)]
        a[0] += len(ev.kernels)
        a[1] += sum(k.duration for k in ev.kernels)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
    tot_l = sum(v[0] for v in agg.values()) / n
    tot_t = sum(v[1] for v in agg.values()) / n
    print('watched element-wise ops: %.0f launches, %.2f ms of device time per step' % (tot_l, tot_t / 1e3))
    for (name, where), (cnt, us) in rows[:60]:
        print('%6.1f launches %7.1f us per step  %-18s %s' % (cnt / n, us / n, name, where[:150]))


if __name__ == '__main__':
    main()
