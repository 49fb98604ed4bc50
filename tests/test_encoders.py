"""the per-image encoders (neuray_amd/network/encoders.py) against the reference's
image_encoder / vis_encoder (tests/golden/case_enc.npz: reference modules with name-seeded weights), the full
base-renderer state_dict surface, and render() starting from images."""
import json
import os
import sys
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from emu_util import emu_lib

BACKENDS = ['cpu', pytest.param('cuda:0', marks=pytest.mark.gpu)]


def fill_by_name(module, scale=0.25):      # as tests/golden/make_golden.py
    with torch.no_grad():
        for name, prm in module.named_parameters():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            v = torch.randn(prm.shape, generator=g) * scale
            if name.endswith('.weight') and prm.dim() == 1:
                v = v * 0.4 + 1.0
            prm.copy_(v)


@pytest.mark.parametrize('dev', BACKENDS)
def test_encoders_match_reference_modules(dev):
    from neuray_amd.network.encoders import DefaultVisEncoder, ImageEncoder
    z = np.load(os.path.join(GOLDEN_DIR, 'case_enc.npz'))
    enc, vis = ImageEncoder().eval(), DefaultVisEncoder({}).eval()
    fill_by_name(enc)
    fill_by_name(vis)
    enc, vis = enc.to(dev), vis.to(dev)
    with torch.no_grad():
        feats = enc(torch.from_numpy(z['imgs']).to(dev))
        out = vis(torch.from_numpy(z['ray_in']).to(dev), feats)
    assert feats.is_contiguous(memory_format=torch.channels_last) and out.is_contiguous(memory_format=torch.channels_last)
    tol = 2e-4 if dev == 'cpu' else 2e-3        # MIOpen picks its own convolution algorithms
    assert float((feats.cpu() - torch.from_numpy(z['img_feats'])).abs().max()) <= tol * max(1.0, float(np.abs(z['img_feats']).max()))
    assert float((out.cpu() - torch.from_numpy(z['ray_feats'])).abs().max()) <= tol * max(1.0, float(np.abs(z['ray_feats']).max()))


def test_full_state_dict_surface_equals_reference_base_renderer():
    from neuray_amd.network.renderer import NeuralRayBaseRenderer
    want = json.load(open(os.path.join(GOLDEN_DIR, 'ref_base_renderer_state_dict.json')))
    r = NeuralRayBaseRenderer({'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': True}, 'build_encoders': True})
    sd = r.state_dict()
    assert sorted(sd) == sorted(want)
    for k, shape in want.items():
        assert list(sd[k].shape) == shape, k


@pytest.mark.parametrize('backend', ['emu', pytest.param('hip', marks=pytest.mark.gpu)])
def test_render_from_images_equals_render_from_precomputed_features(backend):
    """render() with build_encoders: images + initial ray_feats -> encoders (channels-last, consumed by the HIP kernels
    without a relayout) -> render; must equal rendering from the same feature maps handed over as plain NCHW tensors."""
    from neuray_amd import synthetic
    from neuray_amd.network.renderer import NeuralRayBaseRenderer
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': 8,
           'fine_depth_sample_num': 8, 'agg_net_cfg': {'sample_num': 8}, 'fine_agg_net_cfg': {'sample_num': 8},
           'build_encoders': True, 'ray_batch_num': 16}
    r = NeuralRayBaseRenderer(cfg).eval()
    fill_by_name(r)
    if backend == 'emu':
        r._engine_test_lib = emu_lib()
    r = r.to(dev)
    que, ref = synthetic.make_scene(48, 64, 3, seed=5)
    rng = np.random.RandomState(6)
    que['coords'] = (rng.rand(1, 23, 2) * np.array([63, 47])).astype(np.float32)
    tq = {k: torch.from_numpy(v).to(dev) for k, v in que.items()}
    tr = {k: torch.from_numpy(v).to(dev) for k, v in ref.items()}
    init_ray = tr.pop('ray_feats')
    tr.pop('img_feats')
    with torch.no_grad():
        a = r.render(dict(tq), dict(tr, ray_feats=init_ray), False)
        img_feats = r.image_encoder(tr['imgs'])
        ray_feats = r.vis_encoder(init_ray, img_feats)
        assert img_feats.is_contiguous(memory_format=torch.channels_last)
        # same feature tensors, once channels-last (consumed in place) and once as plain NCHW copies (relayout kernel)
        b = r.render(dict(tq), dict(tr, ray_feats=ray_feats, img_feats=img_feats), False)
        c = r.render(dict(tq), dict(tr, ray_feats=ray_feats.contiguous().clone(), img_feats=img_feats.contiguous().clone()), False)
    assert a['pixel_colors_nr_fine'].shape == (1, 23, 3) and float(a['pixel_colors_nr_fine'].abs().max()) > 1e-3
    for k in b:
        assert torch.equal(b[k], c[k]), k
        # (a ran the encoders itself: bitwise equal on CPU; MIOpen convolutions need not repeat bit for bit)
        assert torch.allclose(a[k].float(), b[k].float(), atol=1e-4), k


def test_gradients_reach_the_encoders():
    """Training from images: the map gradients of the HIP backward kernels flow on into image_encoder / vis_encoder
    (and into the initial ray_feats, the learnable per-view parameters of the reference's fine-tuning)."""
    from neuray_amd import synthetic
    from neuray_amd.network.renderer import NeuralRayBaseRenderer
    cfg = {'use_hierarchical_sampling': False, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': 6,
           'agg_net_cfg': {'sample_num': 6}, 'build_encoders': True, 'use_self_hit_prob': True}
    r = NeuralRayBaseRenderer(cfg).train()
    fill_by_name(r)      # (a default-initialised sigma head can sit in its dead ReLU half: zero hit_prob, zero gradients)
    r._engine_test_lib = emu_lib()
    que, ref = synthetic.make_scene(32, 48, 2, seed=8, que_imgs=True)
    que['coords'] = (np.random.RandomState(8).rand(1, 9, 2) * np.array([47, 31])).astype(np.float32)
    tq = {k: torch.from_numpy(v) for k, v in que.items()}
    tr = {k: torch.from_numpy(v) for k, v in ref.items()}
    tr.pop('img_feats')
    init = torch.nn.Parameter(tr.pop('ray_feats'))
    out = r.render(tq, dict(tr, ray_feats=init), True)
    assert float(out['hit_prob_nr'].sum()) > 0.1
    (out['pixel_colors_nr'].square().sum() + out['hit_prob_self'].sum()).backward()
    for name in ('image_encoder.conv1.weight', 'image_encoder.out_conv.weight', 'vis_encoder.out_conv.0.weight'):
        g = dict(r.named_parameters())[name].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0, name
    assert init.grad is not None and float(init.grad.abs().max()) > 0
