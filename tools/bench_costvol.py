"""Where the time of CostVolumeInitNet goes (network/init_net.py:113-160,204-254, network/mvsnet/mvsnet.py:29-69,160-188),
on the evaluation path of an 8 x 800 x 800 image set (volume built at 640 x 640: 8 x 32 x 64 x 160 x 160), and what the frozen 3-D U-Net
costs under the execution variants open to an inference-only network:

    python tools/bench_costvol.py [--views 8] [--reps 5]

Stages are timed with torch.cuda events around the module calls of the forward; the variants re-run `cost_regularization` on the same
variance volume and report their distance from the fp32 NCDHW result."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuray_amd import synthetic                      # noqa: E402
from neuray_amd.network import init_net               # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--views', type=int, default=8)
    ap.add_argument('--reps', type=int, default=5)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    h = w = 800
    _, ref = synthetic.make_scene(h, w, a.views, seed=0, smooth=True)
    info = {k: torch.from_numpy(ref[k]).to(dev) for k in ('imgs', 'poses', 'Ks', 'depth_range')}
    info['nn_ids'] = torch.tensor([[(v + 1) % a.views, (v + 2) % a.views, (v + 3) % a.views] for v in range(a.views)], device=dev)
    torch.manual_seed(0)
    cv = init_net.CostVolumeInitNet({}).eval().to(dev)
    mv = cv.mvsnet
    out = {'views': a.views}
    with torch.no_grad():
        out['whole_forward_ms'] = timeit(lambda: cv(info, info, False), a.reps)
        # ---- stages of the forward (the module's own code path, piece by piece)
        size, ratio = (640, 640), 640 / 800
        imgs = F.interpolate(info['imgs'], size, mode='bilinear')
        x = (imgs - cv.imagenet_mean) / cv.imagenet_std
        prj = init_net.construct_project_matrix(0.25 * ratio, 0.25 * ratio, info['Ks'], info['poses'])
        dv = init_net.get_depth_vals(info['depth_range'], 64)
        feats = mv.feature(x)
        var = mv.variance_volume(feats, feats, info['nn_ids'], prj, prj, dv)
        reg = mv.cost_regularization(var)
        out['stages_ms'] = {
            'resize + normalise': timeit(lambda: (F.interpolate(info['imgs'], size, mode='bilinear') - cv.imagenet_mean) / cv.imagenet_std, a.reps),
            'mvsnet.feature (2-D, 8 views x2: ref + src)': 2 * timeit(lambda: mv.feature(x), a.reps),
            'variance volume (neuray_warp_variance)': timeit(lambda: mv.variance_volume(feats, feats, info['nn_ids'], prj, prj, dv), a.reps),
            'cost_regularization (3-D U-Net)': timeit(lambda: mv.cost_regularization(var), a.reps),
            'nan_to_num + resize + softmax + depth regression': timeit(
                lambda: torch.sum(F.softmax(F.interpolate(torch.nan_to_num(reg.squeeze(1)), (h // 4, w // 4), mode='bilinear'), 1) * dv[:, :, None, None], 1), a.reps),
            'res_net (ResUNetLight on 8 x 800 x 800)': timeit(lambda: cv.res_net(info['imgs']), a.reps),
        }
        cost = F.softmax(F.interpolate(torch.nan_to_num(reg.squeeze(1)), (h // 4, w // 4), mode='bilinear'), 1)
        out['stages_ms']['volume_conv2d + depth_conv + out_conv heads'] = timeit(
            lambda: cv.out_conv(torch.cat([torch.randn(a.views, 32, h // 4, w // 4, device=dev),
                                           cv.volume_conv2d(cost), cv.depth_conv(cost[:, :1])], 1)), a.reps)
        # ---- the 3-D U-Net layer by layer
        net = mv.cost_regularization
        layers = {}
        c0 = net.conv0(var)
        layers['conv0 32->8 @64x160x160'] = timeit(lambda: net.conv0(var), a.reps)
        c1 = net.conv1(c0)
        layers['conv1 8->16 s2'] = timeit(lambda: net.conv1(c0), a.reps)
        c2 = net.conv2(c1)
        layers['conv2 16->16'] = timeit(lambda: net.conv2(c1), a.reps)
        c3 = net.conv3(c2)
        layers['conv3 16->32 s2'] = timeit(lambda: net.conv3(c2), a.reps)
        c4 = net.conv4(c3)
        layers['conv4 32->32'] = timeit(lambda: net.conv4(c3), a.reps)
        c5 = net.conv5(c4)
        layers['conv5 32->64 s2'] = timeit(lambda: net.conv5(c4), a.reps)
        c6 = net.conv6(c5)
        layers['conv6 64->64'] = timeit(lambda: net.conv6(c5), a.reps)
        u7 = c4 + net.conv7(c6)
        layers['conv7 up 64->32'] = timeit(lambda: net.conv7(c6), a.reps)
        u9 = c2 + net.conv9(u7)
        layers['conv9 up 32->16'] = timeit(lambda: net.conv9(u7), a.reps)
        u11 = c0 + net.conv11(u9)
        layers['conv11 up 16->8'] = timeit(lambda: net.conv11(u9), a.reps)
        layers['prob 8->1'] = timeit(lambda: net.prob(u11), a.reps)
        layers['conv0 convolution alone'] = timeit(lambda: net.conv0.conv(var), a.reps)
        layers['conv0 batch-norm + leaky-ReLU alone'] = timeit(lambda: net.conv0.bn(c0), a.reps)
        out['unet_layers_ms'] = layers
        # ---- variants of the whole 3-D U-Net
        base = reg.float()
        scale = float(base.abs().max())
        variants = {}

        def add(tag, fn):
            try:
                got = fn()
                variants[tag] = {'ms': timeit(fn, a.reps), 'max_abs_err_over_max_abs': float((got.float() - base).abs().max()) / scale}
            except Exception as e:                      # noqa: BLE001
                variants[tag] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
        add('fp32 NCDHW (product)', lambda: net(var))
        var_cl = var.contiguous(memory_format=torch.channels_last_3d)
        net_cl = init_net.MVSNet().cost_regularization.eval().to(dev)
        net_cl.load_state_dict(net.state_dict())
        net_cl = net_cl.to(memory_format=torch.channels_last_3d)
        add('fp32 channels_last_3d', lambda: net_cl(var_cl))
        with torch.autocast('cuda', dtype=torch.bfloat16):
            add('bf16 autocast NCDHW', lambda: net(var))
            add('bf16 autocast channels_last_3d', lambda: net_cl(var_cl))
        with torch.autocast('cuda', dtype=torch.float16):
            add('fp16 autocast NCDHW', lambda: net(var))
        torch.backends.cudnn.benchmark = True
        add('fp32 NCDHW, MIOpen find mode (cudnn.benchmark)', lambda: net(var))
        add('fp32 channels_last_3d, MIOpen find mode', lambda: net_cl(var_cl))
        torch.backends.cudnn.benchmark = False
        out['unet_variants'] = variants
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
