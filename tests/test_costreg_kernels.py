"""The two hand-written 3-D convolutions of MVSNet's cost regularisation (csrc/nr_kernels_conv3d.h; reference
network/mvsnet/mvsnet.py:29-69: `conv0` = ConvBnReLU3D(32, 8) on the variance volume, `prob` = Conv3d(8, 1, 3, padding=1)) against
PyTorch's own Conv3d + frozen batch norm + leaky ReLU on the same weights, and the whole CostRegNet fast path against the module path.
CPU: kernels on the emulator; `hip`: libneuray_hip.so."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from emu_util import emu_lib
from neuray_amd.network import mvsnet
from neuray_amd.network import render_ops as ro

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


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


@pytest.fixture(autouse=True)
def pure_pytorch_module_path(request):
    """the `want` side of every comparison here is PyTorch's own batch_norm + leaky_relu; the fused pass has its own test"""
    if 'fused_abn' in request.keywords:
        yield
        return
    mvsnet.FUSED_ABN = False
    yield
    mvsnet.FUSED_ABN = True


def make_net(dev, seed=0):
    torch.manual_seed(seed)
    net = mvsnet.CostRegNet().eval()
    with torch.no_grad():
        for name, buf in net.named_buffers():
            if name.endswith('running_mean'):
                buf.copy_(torch.randn(buf.shape) * 0.1)
            elif name.endswith('running_var'):
                buf.copy_(torch.rand(buf.shape) * 0.5 + 0.75)
        for name, prm in net.named_parameters():
            if name.endswith('bn.weight'):
                prm.copy_(torch.rand(prm.shape) * 0.5 + 0.75)
            elif name.endswith('bn.bias'):
                prm.copy_(torch.randn(prm.shape) * 0.1)
    return net.to(dev)


@pytest.mark.parametrize('shape', [(1, 4, 6, 16), (2, 8, 5, 23), (1, 3, 9, 40)])
def test_conv0_kernel_matches_conv3d_bn_leaky(dev, shape):
    """odd sizes: rows that are not a multiple of the 16-voxel strip, every face / edge / corner of the zero padding"""
    n, d, h, w = shape
    net = make_net(dev)
    x = torch.randn(n, 32, d, h, w, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        want = net.conv0(x)
        got = net.conv0_fast(x)
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize('shape', [(1, 4, 6, 16), (2, 5, 7, 19)])
def test_prob_kernel_matches_conv3d(dev, shape):
    n, d, h, w = shape
    net = make_net(dev)
    x = torch.randn(n, 8, d, h, w, generator=torch.Generator().manual_seed(2)).to(dev)
    with torch.no_grad():
        want = net.prob(x)
        got = net.prob_fast(x)
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


def test_costregnet_fast_path_equals_the_module_path(dev):
    """the whole 3-D U-Net with the two kernels in place (what construct_cost_volume_with_src runs under no_grad) against the plain
    module composition, on a channels-last volume as warp_variance hands it over and on a plain NCDHW one"""
    net = make_net(dev)
    x = torch.randn(2, 32, 8, 16, 24, generator=torch.Generator().manual_seed(3)).to(dev)
    with torch.no_grad():
        want = net.forward_modules(x)
        for vol in (x, x.contiguous(memory_format=torch.channels_last_3d)):
            got = net(vol)
            assert got.shape == want.shape == (2, 1, 8, 16, 24)
            assert float((got - want).abs().max()) <= 5e-5 * max(1.0, float(want.abs().max()))
    # with gradients enabled towards the volume the module path is taken (the kernels are inference-only)
    xg = x.clone().requires_grad_(True)
    out = net(xg)
    assert out.requires_grad


@pytest.mark.parametrize('shape', [(1, 2, 3, 5), (2, 3, 4, 17), (1, 1, 2, 130)])
@pytest.mark.parametrize('with_skip', [True, False])
def test_up11_kernel_matches_convtranspose_bn_leaky_plus_skip(dev, shape, with_skip):
    """c0 + conv11(x) (mvsnet.py:57-69: ConvTranspose3d(16, 8, 3, stride 2, padding 1, output_padding 1) + frozen batch norm + leaky
    ReLU + the skip add) as one kernel: every parity of the output coordinates, the last odd plane / row / column that has a single tap,
    rows wider than one workgroup"""
    n, d, h, w = shape
    net = make_net(dev, seed=4)
    x = torch.randn(n, 16, d, h, w, generator=torch.Generator().manual_seed(5)).to(dev)
    c0 = torch.randn(n, 8, 2 * d, 2 * h, 2 * w, generator=torch.Generator().manual_seed(6)).to(dev)
    with torch.no_grad():
        want = net.conv11(x) + (c0 if with_skip else 0.0)
        pack, shift, slope = net._packs(x.device)[5:8]
        got = net._engine(x).costreg_up11(x, pack, shift, slope, c0 if with_skip else None)
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


# ---- random small volumes on the emulator: every combination of odd / even sizes, single planes / rows / columns ----------------
from hypothesis import HealthCheck, given, settings, strategies as st     # noqa: E402


@settings(max_examples=14, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(n=st.integers(1, 2), d=st.integers(1, 4), h=st.integers(1, 7), w=st.integers(1, 37), seed=st.integers(0, 10 ** 6))
def test_conv0_and_up11_on_random_volumes_emulator(n, d, h, w, seed):
    ro._ENGINES.clear()
    ro._TEST_LIB = emu_lib()
    try:
        net = make_net('cpu', seed=seed % 7)
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(n, 32, d, h, w, generator=g)
        with torch.no_grad():
            want, got = net.conv0(x), net.conv0_fast(x)
        assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), ('conv0', n, d, h, w)
        y = torch.randn(n, 16, d, h, w, generator=g)
        c0 = torch.randn(n, 8, 2 * d, 2 * h, 2 * w, generator=g)
        with torch.no_grad():
            want = net.conv11(y) + c0
            pack, shift, slope = net._packs(y.device)[5:8]
            got = net._engine(y).costreg_up11(y, pack, shift, slope, c0)
        assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), ('up11', n, d, h, w)
    finally:
        ro._TEST_LIB = None
        ro._ENGINES.clear()


@pytest.mark.fused_abn
@pytest.mark.parametrize('shape', [(2, 8, 5, 23), (3, 16, 7, 12, 20), (1, 32, 1, 3), (12, 8, 40, 52)])
def test_fused_activated_batch_norm_matches_batch_norm_plus_leaky_relu(dev, shape):
    """neuray_scale_shift_leaky (MVSNet's frozen InPlaceABN, modules.py:7-23) against F.batch_norm on the running statistics +
    F.leaky_relu: 2-D and 3-D outputs, plane sizes that are and are not multiples of four, in place"""
    torch.manual_seed(3)
    c = shape[1]
    abn = mvsnet.ActivatedBatchNorm(c).eval()
    with torch.no_grad():
        abn.weight.copy_(torch.rand(c) * 0.5 + 0.75); abn.bias.copy_(torch.randn(c) * 0.1)
        abn.running_mean.copy_(torch.randn(c) * 0.1); abn.running_var.copy_(torch.rand(c) * 0.5 + 0.75)
    abn = abn.to(dev)
    x = torch.randn(*shape).to(dev)
    with torch.no_grad():
        mvsnet.FUSED_ABN = False
        want = abn(x.clone())
        mvsnet.FUSED_ABN = True
        xin = x.clone()
        got = abn(xin)
    assert got.data_ptr() == xin.data_ptr()                    # in place, as the reference's InPlaceABN
    assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
    # with gradients enabled (or in training mode) the module takes the PyTorch ops: differentiable, running statistics untouched in eval
    y = abn(x.clone().requires_grad_(True))
    assert y.requires_grad and torch.allclose(y.detach(), want, atol=1e-6)
    with torch.no_grad():                                      # a changed buffer is picked up (the folded scale / shift are rebuilt)
        abn.running_mean.add_(0.25)
        mvsnet.FUSED_ABN = False
        want2 = abn(x.clone())
        mvsnet.FUSED_ABN = True
        assert float((abn(x.clone()) - want2).abs().max()) <= 2e-6 * max(1.0, float(want2.abs().max()))


@pytest.mark.parametrize('layer,shape', [('conv2', (2, 5, 9, 21)), ('conv4', (1, 3, 6, 16)), ('conv1', (2, 6, 11, 37)), ('conv1', (1, 5, 8, 32)),
                                         ('conv3', (1, 4, 7, 19)), ('conv3', (2, 3, 10, 34))])
def test_interior_layers_match_conv3d_bn_leaky(dev, layer, shape):
    """neuray_conv3d_bn_leaky (conv1 ... conv4: stride 1 and 2, frozen batch norm folded, leaky ReLU) against the module: odd and even
    sizes, sizes that are not multiples of the 16-voxel strip / the 4-row group, every face of the zero padding"""
    n, d, h, w = shape
    net = make_net(dev)
    mod = getattr(net, layer)
    x = torch.randn(n, mod.conv.in_channels, d, h, w, generator=torch.Generator().manual_seed(2)).to(dev)
    with torch.no_grad():
        want = mod(x)
        got = net._mfma(mod, x, True)
    assert got.shape == want.shape and got.data_ptr() != want.data_ptr()
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    assert net._mfma(net.conv6, torch.zeros(1, 64, 2, 3, 3, device=dev), True).shape == (1, 64, 2, 3, 3)      # (not built: the module)


@pytest.mark.parametrize('shape', [(1, 3, 5, 9), (2, 2, 6, 20)])
def test_conv9_decoder_step_matches_the_modules(dev, shape):
    """neuray_convtranspose3d_bn_leaky at (32 -> 16): c2 + conv9(x) (transposed convolution + frozen batch norm + leaky ReLU + skip) against
    the modules, odd and even sizes"""
    n, d, h, w = shape
    net = make_net(dev)
    x = torch.randn(n, 32, d, h, w, generator=torch.Generator().manual_seed(3)).to(dev)
    c2 = torch.randn(n, 16, 2 * d, 2 * h, 2 * w, generator=torch.Generator().manual_seed(4)).to(dev)
    with torch.no_grad():
        want = c2 + net.conv9(x)
        got = net._up(net.conv9, x, c2, True)
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))


def test_feature_net_mfma_layers_match_the_modules(dev):
    """FeatureNet (mvsnet.py:7-30) with its 16 / 32-channel 3 x 3 layers on the implicit-GEMM kernel (a one-plane volume through
    neuray_conv3d_bn_leaky) against the module path: odd sizes, batch of 3"""
    torch.manual_seed(5)
    net = mvsnet.FeatureNet().eval()
    with torch.no_grad():
        for name, buf in net.named_buffers():
            if name.endswith('running_mean'):
                buf.copy_(torch.randn(buf.shape) * 0.1)
            elif name.endswith('running_var'):
                buf.copy_(torch.rand(buf.shape) * 0.5 + 0.75)
        for name, prm in net.named_parameters():
            if name.endswith('bn.weight'):
                prm.copy_(torch.rand(prm.shape) * 0.5 + 0.75)
            elif name.endswith('bn.bias'):
                prm.copy_(torch.randn(prm.shape) * 0.1)
    net = net.to(dev)
    x = torch.randn(3, 3, 44, 68, generator=torch.Generator().manual_seed(6)).to(dev)
    with torch.no_grad():
        mvsnet.FAST_CONV3D = False
        want = net(x)
        mvsnet.FAST_CONV3D = True
        got = net(x)
        assert net._fast(x)
    assert got.shape == want.shape == (3, 32, 11, 17)
    assert float((got - want).abs().max()) <= 3e-5 * max(1.0, float(want.abs().max()))
