"""A/B of the encoders' 3 x 3 stride-1 convolutions: neuray_conv3x3_x3 (split-operand bf16 MFMA, csrc/nr_kernels_conv2d.h) against PyTorch's
F.conv2d (MIOpen) on the layer shapes of the image / visibility encoders at 9 x 800 x 800 (reference network/ops.py:150-230,
network/vis_encoder.py:6-21): forward on the pre-padded input and the data gradient, per-call milliseconds (HIP events on the current stream,
median of `--reps` after a warm-up), TFLOP/s by the direct MAC count, and both errors against a float64 convolution.

    python tools/ab_conv2d.py [--reps 30] [--n 9]
"""
import argparse
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuray_amd.network import render_ops as ro  # noqa: E402

# (label, C_in, C_out, output size, convolutions of that shape per encoder pass)
LAYERS = [('layer3 128->128 @50', 128, 128, 50, 11), ('layer2 64->64 @100', 64, 64, 100, 3), ('layer1/vis 32->32 @200', 32, 32, 200, 5),
          ('up3/iconv3 128->64 @100', 128, 64, 100, 2), ('up2/iconv2/vis 64->32 @200', 64, 32, 200, 3)]


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--n', type=int, default=9)
    ap.add_argument('--only', default='', help='substring of the layer label')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    eng = ro.engine_for(dev)
    torch.manual_seed(0)
    tot = {'miopen_fwd': 0.0, 'x3_fwd': 0.0, 'miopen_bwd': 0.0, 'x3_bwd': 0.0}
    print('tile overrides: NT=%s MTW=%s TW=%s WC=%s' % tuple(os.environ.get(k, '-') for k in ("NEURAY_CONV2D_NT", "NEURAY_CONV2D_MTW", "NEURAY_CONV2D_TW", "NEURAY_CONV2D_WC")))
    print('%-28s %9s %9s %7s | %9s %9s %7s | %9s %9s' % ('layer (n = %d)' % args.n, 'MIOpen ms', 'x3 ms', 'TF/s x3', 'MIOpen dx', 'x3 dx', 'TF/s', 'err fp32', 'err x3'))
    for label, cin, cout, s, count in LAYERS:
        if args.only not in label:
            continue
        x = torch.randn(args.n, cin, s + 2, s + 2, device=dev)
        w = torch.randn(cout, cin, 3, 3, device=dev) / (3.0 * cin ** 0.5)
        dy = torch.randn(args.n, cout, s, s, device=dev)
        pack, pack_t = eng.conv3x3_x3_pack(w), eng.conv3x3_x3_pack(w, transpose_flip=True)
        flop = 2.0 * 9 * cin * cout * s * s * args.n
        ref = F.conv2d(x.double(), w.double())
        y_lib, y_x3 = F.conv2d(x, w), eng.conv3x3_x3(x, pack, None, cout, 0)
        e_lib = float((y_lib.double() - ref).abs().max() / ref.abs().max())
        e_x3 = float((y_x3.double() - ref).abs().max() / ref.abs().max())
        dref = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double())
        dx3 = eng.conv3x3_x3(dy, pack_t, None, cin, 2)
        e_dx = float((dx3.double() - dref).abs().max() / dref.abs().max())
        t_lib = timed(lambda: F.conv2d(x, w), args.reps)
        t_x3 = timed(lambda: eng.conv3x3_x3(x, pack, None, cout, 0), args.reps)
        t_lib_b = timed(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False]), args.reps)
        t_x3_b = timed(lambda: eng.conv3x3_x3(dy, pack_t, None, cin, 2), args.reps)
        t_lib_w = timed(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]), args.reps)
        t_x3_w = timed(lambda: eng.conv3x3_x3_wrw(dy, x), args.reps)
        wref = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double())
        e_w = float((eng.conv3x3_x3_wrw(dy, x).double() - wref).abs().max() / wref.abs().max())
        print('%-28s %9.4f %9.4f %7.1f | %9.4f %9.4f %7.1f | %9.2e %9.2e (dx %.2e) | dW: MIOpen %.4f x3 %.4f ms (%.1f TF/s, err %.2e)'
              % (label, t_lib, t_x3, flop / t_x3 * 1e-9, t_lib_b, t_x3_b, flop / t_x3_b * 1e-9, e_lib, e_x3, e_dx, t_lib_w, t_x3_w, flop / t_x3_w * 1e-9, e_w))
        tot['miopen_wrw'] = tot.get('miopen_wrw', 0.0) + count * t_lib_w
        tot['x3_wrw'] = tot.get('x3_wrw', 0.0) + count * t_x3_w
        tot['miopen_fwd'] += count * t_lib
        tot['x3_fwd'] += count * t_x3
        tot['miopen_bwd'] += count * t_lib_b
        tot['x3_bwd'] += count * t_x3_b
    print('per encoder pass (image + visibility encoder, weighted by layer count): forward MIOpen %.3f ms, x3 %.3f ms; data gradient MIOpen %.3f ms, x3 %.3f ms'
          % (tot['miopen_fwd'], tot['x3_fwd'], tot['miopen_bwd'], tot['x3_bwd']))
    print('weight gradient: MIOpen (with the NHWC copies it makes) %.3f ms, x3 %.3f ms' % (tot.get('miopen_wrw', 0.0), tot.get('x3_wrw', 0.0)))
    t_pack = timed(lambda: eng.conv3x3_x3_pack(w), args.reps)
    print('pack of one 64 -> 32 layer: %.4f ms' % t_pack)


if __name__ == '__main__':
    main()
