"""Time the stand-alone 3-D kernels of the cost regularisation on the bench volume (8 x 64 x 160 x 160), HIP events:
    python tools/time_costreg.py [lib.so ...]      (default: the product library)
-> per library: conv0 / up11 / prob / frozen batch-norm + leaky pass, ms and GB/s of compulsory traffic; up11 / prob / abn checked against PyTorch."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuray_amd import _lib                          # noqa: E402
from neuray_amd.engine import RenderEngine           # noqa: E402
from neuray_amd.network import mvsnet                # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device('cuda', 0)
    libs = sys.argv[1:] or [_lib.LIB_PATH]
    torch.manual_seed(0)
    net = mvsnet.CostRegNet().eval().to(dev)
    n, d, h, w = 8, 64, 160, 160
    x16 = torch.randn(n, 16, d // 2, h // 2, w // 2, device=dev)
    c0 = torch.randn(n, 8, d, h, w, device=dev)
    with torch.no_grad():
        mvsnet.FUSED_ABN = False
        want_up = c0 + net.conv11(x16)
        want_prob = net.prob(c0)
        want_abn = net.conv0.bn(c0.clone())
        mvsnet.FUSED_ABN = True
    for path in libs:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        from ab_forward import bind_compat              # (tolerates a library from before the current ABI)
        eng = RenderEngine(dev, _test_lib=bind_compat(path) if path != _lib.LIB_PATH else None)
        net._engine = lambda x_, e=eng: e
        net.__dict__.pop('_fast_packs', None)
        packs = net._packs(dev)
        res = {'lib': os.path.basename(path)}
        with torch.no_grad():
            pack, shift, slope = packs[5:8]
            got = eng.costreg_up11(x16, pack, shift, slope, c0)
            res['up11_err'] = float((got - want_up).abs().max())
            ms = timeit(lambda: eng.costreg_up11(x16, pack, shift, slope, c0))
            res['up11_ms'] = ms
            res['up11_gb_s'] = (x16.numel() + 2 * c0.numel()) * 4 / ms / 1e6
            w27, pb = packs[3], packs[4]
            got = eng.costreg_prob(c0, w27, pb)
            res['prob_err'] = float((got - want_prob).abs().max())
            ms = timeit(lambda: eng.costreg_prob(c0, w27, pb))
            res['prob_ms'] = ms
            res['prob_gb_s'] = (c0.numel() + c0.numel() // 8) * 4 / ms / 1e6
            if getattr(eng.lib, 'neuray_scale_shift_leaky', None) is not None and hasattr(eng.lib.neuray_scale_shift_leaky, 'argtypes') and eng.lib.neuray_scale_shift_leaky.argtypes:
                bn = net.conv0.bn
                sc, sh = bn._folded(dev)
                y = c0.clone()
                eng.scale_shift_leaky_(y, sc, sh, bn.slope)
                res['abn_err'] = float((y - want_abn).abs().max())
                ms = timeit(lambda: eng.scale_shift_leaky_(y, sc, sh, bn.slope))
                res['abn_fused_ms'] = ms
                res['abn_fused_gb_s'] = 2 * c0.numel() * 4 / ms / 1e6
                mvsnet.FUSED_ABN = False
                res['abn_pytorch_ms'] = timeit(lambda: bn(y))
                mvsnet.FUSED_ABN = True
            if getattr(eng.lib, 'neuray_conv3d_bn_leaky', None) is not None and getattr(eng.lib.neuray_conv3d_bn_leaky, 'argtypes', None):
                # the stride-1 interior layers at their shapes inside the U-Net of this volume, two reference views per call as the init net runs them
                for name, (c, dd, hh, ww) in (('conv1', (8, d, h, w)), ('conv2', (16, d // 2, h // 2, w // 2)), ('conv3', (16, d // 2, h // 2, w // 2)), ('conv4', (32, d // 4, h // 4, w // 4))):
                    mod = getattr(net, name)
                    xin = torch.randn(2, c, dd, hh, ww, device=dev)
                    mvsnet.FUSED_ABN = False
                    want = mod(xin)
                    mvsnet.FUSED_ABN = True
                    got = net._mfma(mod, xin, True)
                    res[name + '_err'] = float((got - want).abs().max())
                    res[name + '_kernel_ms'] = timeit(lambda: net._mfma(mod, xin, True))
                    res[name + '_module_ms'] = timeit(lambda: mod(xin))
                    res[name + '_kernel_tflops'] = 2 * got.numel() * c * 27 / res[name + '_kernel_ms'] / 1e9
            if getattr(eng.lib, 'neuray_convtranspose3d_bn_leaky', None) is not None and getattr(eng.lib.neuray_convtranspose3d_bn_leaky, 'argtypes', None):
                x9 = torch.randn(2, 32, d // 4, h // 4, w // 4, device=dev)
                c2 = torch.randn(2, 16, d // 2, h // 2, w // 2, device=dev)
                mvsnet.FUSED_ABN = False
                want = c2 + net.conv9(x9)
                mvsnet.FUSED_ABN = True
                res['conv9_err'] = float((net._up(net.conv9, x9, c2, True) - want).abs().max())
                res['conv9_kernel_ms'] = timeit(lambda: net._up(net.conv9, x9, c2, True))
                res['conv9_module_ms'] = timeit(lambda: c2 + net.conv9(x9))
        print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
