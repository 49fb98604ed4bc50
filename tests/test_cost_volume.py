"""the cost-volume init net (network/init_net.py:113-160,204-258, network/mvsnet/*): the fused
plane-sweep variance kernel `neuray_warp_variance` against the reference's homo_warp and the numpy oracle, the MVSNet /
CostVolumeInitNet mirrors against the reference modules (tests/golden/case_cost_volume.npz: fill_by_name weights, the
training path).  inplace_abn.ABN and kornia's create_meshgrid are absent third-party dependencies restated in
tests/golden/ref_harness.py."""
import json
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from emu_util import emu_lib
from test_encoders import fill_by_name
from neuray_amd.network import render_ops as ro
from oracle import neuray_oracle as orc

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


def fill_buffers_by_name(module):      # as tests/golden/make_golden.py
    with torch.no_grad():
        for name, buf in module.named_buffers():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            if name.endswith('running_mean'):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
            elif name.endswith('running_var'):
                buf.copy_(torch.rand(buf.shape, generator=g) * 0.5 + 0.75)


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(GOLDEN_DIR, 'case_cost_volume.npz'))


@pytest.fixture(params=BACKENDS)
def dev(request):
    ro._ENGINES.clear()
    if request.param == 'emu':
        ro._TEST_LIB = emu_lib()
        yield 'cpu'
        ro._TEST_LIB = None
        ro._ENGINES.clear()
    else:
        ro._TEST_LIB = None
        yield 'cuda:0'


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b))) / max(1.0, float(np.max(np.abs(b))))


def test_oracle_homo_warp_matches_reference(gold):
    ids = gold['nn_ids'][:, 0]
    inv = np.stack([np.linalg.inv(p.astype(np.float64)).astype(np.float32) for p in gold['ref_prj']])
    got = orc.homo_warp(gold['src_feats'][ids], gold['src_prj'][ids], inv, gold['depth_vals'])
    assert got.shape == gold['warped0'].shape == (2, 32, 8, 16, 16)
    d = np.abs(got - gold['warped0'])
    # a tap that sits on the border of the source map within fp32 noise is in for one and out for the other (zero padding)
    assert np.mean(d <= 1e-4 * max(1.0, np.abs(gold['warped0']).max())) >= 0.999


def test_warp_variance_kernel_matches_oracle(gold, dev):
    t = lambda k: torch.from_numpy(gold[k]).to(dev)
    eng = ro.engine_for(dev)
    got = eng.warp_variance(t('ref_feats'), t('src_feats'), t('nn_ids'), t('ref_prj'), t('src_prj'), t('depth_vals')).cpu().numpy()
    want = orc.variance_volume(gold['ref_feats'], gold['src_feats'], gold['nn_ids'], gold['ref_prj'], gold['src_prj'], gold['depth_vals'])
    assert got.shape == want.shape == (2, 32, 8, 16, 16)
    d = np.abs(got - want)
    assert np.mean(d <= 1e-4 * max(1.0, np.abs(want).max())) >= 0.999
    # identical source views and a reference view that is its own neighbour: the variance vanishes wherever the warp is
    # the identity (same projection) - a size-independent property
    same = eng.warp_variance(t('ref_feats')[:1], t('ref_feats')[:1], torch.zeros(1, 2, dtype=torch.long), t('ref_prj')[:1],
                             t('ref_prj')[:1], t('depth_vals')[:1]).cpu().numpy()
    assert float(np.abs(same[:, :, :, 1:-1, 1:-1]).max()) <= 1e-3 * float(np.abs(gold['ref_feats']).max()) ** 2
    # the channels-last layout conv0 reads comes from its own kernel (eight lanes per voxel, warp_variance_cl_kernel): every channel's
    # sums are formed by the same operations in the same order, so the two layouts hold the same bits - also on maps whose pixel count
    # is not a multiple of the 32 voxels of a workgroup, and with taps that leave the source map
    cl = eng.warp_variance(t('ref_feats'), t('src_feats'), t('nn_ids'), t('ref_prj'), t('src_prj'), t('depth_vals'), channels_last=True)
    assert cl.shape == (2, 32, 8, 16, 16) and cl.stride(1) == 1
    assert np.array_equal(cl.cpu().numpy(), got)
    g = torch.Generator().manual_seed(5)
    rf, sf = torch.randn(2, 32, 9, 11, generator=g).to(dev), torch.randn(3, 32, 9, 11, generator=g).to(dev)
    ids = torch.tensor([[0, 2, 1], [1, 1, 0]])
    prj = t('ref_prj')[:1].repeat(3, 1, 1).clone()
    prj[1, 0, 3] += 40.0
    prj[2, 1, 3] -= 25.0
    prj[2, 0, 0] *= 1.3
    dv = t('depth_vals')[:1, :5].repeat(2, 1).contiguous()
    a = eng.warp_variance(rf, sf, ids, prj[:2], prj, dv)
    b = eng.warp_variance(rf, sf, ids, prj[:2], prj, dv, channels_last=True)
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy()) and float(a.abs().max()) > 0


def test_cost_volume_init_net_matches_reference(gold, dev):
    from neuray_amd.network.init_net import CostVolumeInitNet, construct_cost_volume_with_src, construct_project_matrix, get_depth_vals
    want_sd = json.load(open(os.path.join(GOLDEN_DIR, 'ref_cost_volume_init_net_state_dict.json')))
    net = CostVolumeInitNet({'cost_volume_sn': 8})
    sd = net.state_dict()
    assert sorted(sd) == sorted(want_sd) and all(list(sd[k].shape) == want_sd[k] for k in want_sd)
    assert not any(p.requires_grad for p in net.mvsnet.parameters())
    fill_by_name(net)
    fill_buffers_by_name(net.mvsnet)
    net = net.eval().to(dev)
    ref = {k[4:]: torch.from_numpy(gold[k]).to(dev) for k in gold.files if k.startswith('ref_') and k not in ('ref_feats', 'ref_prj')}
    src = {k[4:]: torch.from_numpy(gold[k]).to(dev) for k in gold.files if k.startswith('src_') and k not in ('src_feats', 'src_prj')}
    ref['nn_ids'] = torch.from_numpy(gold['nn_ids']).to(dev)
    assert relerr(get_depth_vals(ref['depth_range'], 8).cpu().numpy(), gold['depth_vals']) <= 1e-6
    assert relerr(construct_project_matrix(0.25, 0.25, ref['Ks'], ref['poses']).cpu().numpy(), gold['ref_prj']) <= 1e-6
    tol = 2e-4 if dev == 'cpu' else 3e-3           # (MIOpen picks its own 2-D / 3-D convolution algorithms)
    with torch.no_grad():
        feats = net.mvsnet.feature((ref['imgs'] - net.imagenet_mean) / net.imagenet_std)
        assert relerr(feats.cpu().numpy(), gold['ref_feats']) <= tol
        cost_reg, depth = construct_cost_volume_with_src(ref, src, net.mvsnet, 8, net.imagenet_mean, net.imagenet_std, True)
        out = net(ref, src, True)
    assert relerr(cost_reg.cpu().numpy(), gold['cost_reg']) <= 10 * tol and relerr(depth.cpu().numpy(), gold['depth']) <= 10 * tol
    assert out.shape == (2, 32, 16, 16) and relerr(out.cpu().numpy(), gold['ray_feats']) <= 20 * tol


def test_gen_renderer_builds_with_the_cost_volume_init_net():
    from neuray_amd.network import renderer as R
    gen = R.NeuralRayGenRenderer({'init_net_type': 'cost_volume', 'init_net_cfg': {'cost_volume_sn': 8}})
    assert any(k.startswith('init_net.mvsnet.cost_regularization.') for k in gen.state_dict())


def test_eager_port_of_variance_volume_matches_oracle(gold):
    """the eager tensor formulation bench.py times beside the kernel, pinned through the (reference-pinned) oracle"""
    from oracle import torch_eager_port as tep
    t = lambda k: torch.from_numpy(gold[k])
    got = tep.variance_volume(t('ref_feats'), t('src_feats'), t('nn_ids'), t('ref_prj'), t('src_prj'), t('depth_vals')).numpy()
    want = orc.variance_volume(gold['ref_feats'], gold['src_feats'], gold['nn_ids'], gold['ref_prj'], gold['src_prj'], gold['depth_vals'])
    assert np.mean(np.abs(got - want) <= 1e-4 * max(1.0, np.abs(want).max())) >= 0.999


def test_lightning_style_checkpoint_loads(tmp_path):
    """`mvsnet_pl.ckpt` (network/mvsnet/mvsnet_pl.ckpt) is a pytorch-lightning checkpoint: tensors under
    'state_dict' with a 'model.' prefix next to non-tensor objects that torch's default weights_only=True unpickler
    refuses.  extract_model_state_dict must read such a file (and the reference's real one, where the tree exists)."""
    import argparse
    from neuray_amd.network import mvsnet as mv
    net = mv.MVSNet()
    sd = {'model.' + k: v.clone() for k, v in net.state_dict().items()}
    sd['loss.weight'] = torch.zeros(1)
    path = str(tmp_path / 'pl.ckpt')
    torch.save({'state_dict': sd, 'hyper_parameters': argparse.Namespace(lr=1e-3, levels=[1, 2]), 'epoch': 3,
                'callbacks': {object: 'not a tensor'}}, path)
    got = mv.extract_model_state_dict(path)
    assert set(got) == set(net.state_dict()) and all(torch.equal(got[k], v) for k, v in net.state_dict().items())
    mv.load_ckpt(net, path)
    real = '/root/reference/network/mvsnet/mvsnet_pl.ckpt'
    if os.path.exists(real):
        net2 = mv.MVSNet()
        mv.load_ckpt(net2, real)
        assert set(mv.extract_model_state_dict(real)) >= set(k for k in net2.state_dict() if 'num_batches_tracked' not in k)
