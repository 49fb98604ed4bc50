"""The plane-sweep variance volume (neuray_warp_variance_layout) alone, by HIP events around the library call, per library build:
    python tools/time_warp_variance.py name=path/to/lib.so [name=path ...]
Shape of the 800 x 800 evaluation path: 8 reference views, 3 neighbours each, 64 planes of 160 x 160 x 32 (tools/bench_costvol.py's
scene and projections); both output layouts.  GB/s on the 1.68 GB written per call."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from ab_forward import bind_compat                     # noqa: E402
from neuray_amd import synthetic                        # noqa: E402
from neuray_amd.network import init_net                 # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    views, fh, fw, dn = 8, 160, 160, 64
    _, ref = synthetic.make_scene(800, 800, views, seed=0, smooth=True)
    info = {k: torch.from_numpy(ref[k]).to(dev) for k in ('poses', 'Ks', 'depth_range')}
    prj = init_net.construct_project_matrix(0.2, 0.2, info['Ks'], info['poses'])
    dv = init_net.get_depth_vals(info['depth_range'], dn).contiguous()
    nn_ids = torch.tensor([[(v + 1) % views, (v + 2) % views, (v + 3) % views] for v in range(views)], device=dev)
    feats = torch.randn(views, fh, fw, 32, device=dev)
    inv = torch.inverse(prj)
    tr = torch.stack([prj[nn_ids[:, j]] @ inv for j in range(3)], 1)[:, :, :3, :].contiguous()
    ids = nn_ids.int().contiguous()
    out = torch.empty(views * 32 * dn * fh * fw, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for a in sys.argv[1:]:
        name, path = a.split('=', 1)
        lib = bind_compat(path)
        row = []
        for cl in (0, 1):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            def call():
                rc = lib.neuray_warp_variance_layout(feats.data_ptr(), feats.data_ptr(), ids.data_ptr(), tr.data_ptr(), dv.data_ptr(), views, views, 3,
                                                     dn, fh, fw, cl, out.data_ptr(), stream)
                assert rc == 0
            for _ in range(3):
                call()
            ev[0].record()
            for _ in range(10):
                call()
            ev[1].record()
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / 10
            row.append('%s %.3f ms (%.2f TB/s written)' % ('channels-last' if cl else 'NCDHW', ms, out.numel() * 4 / ms / 1e9))
        print(name, ' | '.join(row), ' checksum %.6e' % float(out.double().sum()))


if __name__ == '__main__':
    main()
