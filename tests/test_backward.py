"""Backward kernels against PyTorch autograd of the same formulas (the eager port of the reference's op sequence,
oracle/torch_eager_port.py, whose end-to-end gradients are pinned to the reference's own autograd by
tests/golden/case_g_grads.npz in test_oracle_golden.py).

First kernel: the ray kernel's backward (positional encoding, ray attention, LayerNorm, sigma head, compositing):
gradients w.r.t. the per-point records and the ray-part weights."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_weights
from emu_util import emu_lib
from oracle import torch_eager_port as tep

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]
IP = 'agg_net.agg_impl.'


def ray_part_torch(w, g, colors, nvalid, depth):
    """ibrnet.py:356-360 + renderer.py:157-166 + render_ops.py:72-80 on (geometry feature, colour, #valid views)."""
    rn, dn, _ = g.shape
    g = g + tep._posenc(dn, g.device)
    lin = lambda x, name: F.linear(x, w[IP + name + '.weight'], w.get(IP + name + '.bias'))
    q = lin(g, 'ray_attention.w_qs').view(rn, dn, 4, 4).transpose(1, 2)
    k = lin(g, 'ray_attention.w_ks').view(rn, dn, 4, 4).transpose(1, 2)
    v = lin(g, 'ray_attention.w_vs').view(rn, dn, 4, 4).transpose(1, 2)
    att = (q / 2) @ k.transpose(2, 3)
    att = att.masked_fill(((nvalid > 1).float().view(rn, 1, dn, 1)) == 0, -1e9)
    o = (F.softmax(att, -1) @ v).transpose(1, 2).reshape(rn, dn, 16)
    o = F.layer_norm(lin(o, 'ray_attention.fc') + g, (16,), w[IP + 'ray_attention.layer_norm.weight'],
                     w[IP + 'ray_attention.layer_norm.bias'], 1e-6)
    sigma = F.relu(lin(F.elu(lin(o, 'out_geometry_fc.0')), 'out_geometry_fc.2'))[..., 0].masked_fill(nvalid < 1, 0.)
    alpha = 1 - torch.exp(-sigma)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]
    hit = alpha * T
    return (hit.unsqueeze(-1) * colors).sum(1), hit, (hit * depth).sum(1)


@pytest.mark.parametrize('backend', BACKENDS)
@pytest.mark.parametrize('rn,dn,with_aux', [(5, 16, True), (9, 64, False), (3, 7, True)])
def test_rays_backward_matches_autograd(rn, dn, with_aux, backend):
    from neuray_amd.engine import RenderEngine
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    eng = RenderEngine(dev, _test_lib=emu_lib() if backend == 'emu' else None)
    weights = load_weights(False)
    packed = eng.pack_pass(weights, 'dist_decoder.', 'agg_net.')
    rng = np.random.RandomState(rn * 100 + dn)
    rec = np.zeros((rn, dn, 20), np.float32)
    rec[..., :16] = rng.randn(rn, dn, 16) * 0.7
    rec[..., 16:19] = rng.rand(rn, dn, 3)
    rec[..., 19] = rng.randint(0, 4, size=(rn, dn))           # 0 -> sigma forced to 0, <= 1 -> masked attention row
    depth = np.sort(rng.rand(rn, dn).astype(np.float32) * 4 + 2, -1)
    d_pixel = rng.randn(rn, 3).astype(np.float32)
    d_hit = rng.randn(rn, dn).astype(np.float32) if with_aux else None
    d_dep = rng.randn(rn).astype(np.float32) if with_aux else None

    w = {k: torch.from_numpy(v.copy()).double().requires_grad_(k.startswith(IP + 'ray_attention') or k.startswith(IP + 'out_geometry_fc'))
         for k, v in weights.items() if k.startswith(IP)}
    g = torch.from_numpy(rec[..., :16]).double().requires_grad_(True)
    col = torch.from_numpy(rec[..., 16:19]).double().requires_grad_(True)
    pix, hit, dep = ray_part_torch(w, g, col, torch.from_numpy(rec[..., 19]).double(), torch.from_numpy(depth).double())
    loss = (pix * torch.from_numpy(d_pixel).double()).sum()
    if with_aux:
        loss = loss + (hit * torch.from_numpy(d_hit).double()).sum() + (dep * torch.from_numpy(d_dep).double()).sum()
    loss.backward()

    t = lambda a: torch.from_numpy(a).to(dev) if a is not None else None
    d_rec, gw = eng.render_rays_backward(t(rec), t(depth), packed, t(d_pixel), t(d_hit), t(d_dep))
    d_rec = d_rec.cpu().numpy()
    scale = lambda ref: max(1.0, float(np.abs(ref).max()))
    want_g, want_c = g.grad.numpy(), col.grad.numpy()
    assert np.abs(d_rec[..., :16] - want_g).max() <= 2e-4 * scale(want_g)
    assert np.abs(d_rec[..., 16:19] - want_c).max() <= 1e-5 * scale(want_c)
    assert np.all(d_rec[..., 19] == 0)
    for name, grad in gw.items():
        want = w[IP + name].grad.numpy()
        assert grad.shape == want.shape, name
        assert np.abs(grad.cpu().numpy() - want).max() <= 3e-4 * scale(want), name


@pytest.mark.parametrize('backend', BACKENDS)
def test_rays_backward_rejects_long_rays(backend):
    from neuray_amd.engine import RenderEngine
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    eng = RenderEngine(dev, _test_lib=emu_lib() if backend == 'emu' else None)
    packed = eng.pack_pass(load_weights(False), 'dist_decoder.', 'agg_net.')
    z = torch.zeros(1, 65, 20, device=dev)
    with pytest.raises(RuntimeError, match='dn=65'):
        eng.render_rays_backward(z, torch.ones(1, 65, device=dev), packed, torch.zeros(1, 3, device=dev))
