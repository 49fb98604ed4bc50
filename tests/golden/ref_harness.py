"""Import harness for the *reference* NeuRay modules (read-only, /root/reference).

TEST INFRASTRUCTURE ONLY.  Used by tests/golden/make_golden.py in the build
container to generate golden vectors; it never runs on the GPU box (the
reference tree does not exist there) and nothing in neuray_amd/ imports it.

Recipe follows SURVEY.md section 8(c): insert dummy third-party modules in
sys.modules (easydict, skimage, cv2, ...) and patch the hard-coded "cuda:0" in
IBRNetWithNeuRay.posenc (network/ibrnet.py:312) so the modules run on CPU.
Nothing under /root/reference is modified or copied.
"""
import sys
import importlib.machinery
import types
import os

REFERENCE_ROOT = os.environ.get("NEURAY_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "network"))


class _Permissive(types.ModuleType):
    """Stub module: any missing attribute resolves to a dummy callable/class."""
    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        dummy = type(item, (), {"__init__": lambda self, *a, **k: None,
                                "__call__": lambda self, *a, **k: None})
        setattr(self, item, dummy)
        return dummy


def _stub(name, **attrs):
    m = _Permissive(name)
    # a module in sys.modules without a __spec__ makes importlib.util.find_spec raise (torch._dynamo's trace rules probe
    # 'sklearn' & co. that way when an optimiser is first built), so the stubs carry one
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _abn_class():
    """Restatement of inplace_abn.ABN (mapillary/inplace_abn 1.x, the activated batch norm the reference's MVSNet is built
    with: network/init_net.py:5,121 `MVSNet(ABN)`; requirements.txt lists `inplace-abn` unpinned): F.batch_norm with the
    module's running statistics followed by leaky_relu(0.01); parameters weight / bias, buffers running_mean /
    running_var (no num_batches_tracked); any number of trailing dimensions (it is used on 4-D and 5-D tensors)."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    class ABN(nn.Module):
        def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu", activation_param=0.01):
            super().__init__()
            self.num_features, self.eps, self.momentum = num_features, eps, momentum
            self.activation, self.activation_param = activation, activation_param
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
            self.register_buffer('running_mean', torch.zeros(num_features))
            self.register_buffer('running_var', torch.ones(num_features))

        def forward(self, x):
            x = F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, self.training, self.momentum, self.eps)
            return F.leaky_relu(x, negative_slope=self.activation_param)
    return ABN


def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=None):
    """Restatement of kornia.utils.create_meshgrid for the one way the reference calls it (mvsnet/modules.py:42,
    normalized_coordinates=False): [1, H, W, 2] pixel coordinates, (x, y) order."""
    import torch
    assert not normalized_coordinates
    xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
    gy, gx = torch.meshgrid(ys, xs, indexing='ij')
    return torch.stack([gx, gy], -1)[None]


def install_stubs():
    import torch.nn as nn
    ABN = _abn_class()
    if "easydict" not in sys.modules:
        _stub("easydict", EasyDict=dict)
    for n in ["skimage", "skimage.io", "skimage.metrics", "h5py", "plyfile", "transforms3d",
              "transforms3d.axangles", "transforms3d.euler", "kornia", "kornia.utils", "tensorboardX",
              "imageio", "sklearn", "sklearn.decomposition", "sklearn.manifold"]:
        if n not in sys.modules:
            _stub(n)
    sys.modules["skimage.io"].imsave = lambda *a, **k: None
    sys.modules["skimage.io"].imread = lambda *a, **k: None
    if "cv2" not in sys.modules:
        _stub("cv2", INTER_LINEAR=1, INTER_NEAREST=0, INTER_AREA=3, INTER_CUBIC=2,
              SOLVEPNP_ITERATIVE=0, SOLVEPNP_EPNP=1, BORDER_CONSTANT=0)
    if "inplace_abn" not in sys.modules:
        _stub("inplace_abn", ABN=ABN, InPlaceABN=ABN)
    sys.modules["kornia.utils"].create_meshgrid = create_meshgrid


def import_reference():
    """Returns a namespace with the reference hot-path modules (CPU-patched)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import numpy as np
    import torch
    import network.ibrnet as ibrnet

    def posenc_cpu(self, d_hid, n_samples):
        # same table as network/ibrnet.py:305-313 without the .to("cuda:0")
        def get_position_angle_vec(position):
            return [position / np.power(10000, 2 * (hid_j // 2) / d_hid) for hid_j in range(d_hid)]
        t = np.array([get_position_angle_vec(pos_i) for pos_i in range(n_samples)])
        t[:, 0::2] = np.sin(t[:, 0::2])
        t[:, 1::2] = np.cos(t[:, 1::2])
        return torch.from_numpy(t).float().unsqueeze(0)

    ibrnet.IBRNetWithNeuRay.posenc = posenc_cpu
    import network.render_ops as render_ops
    import network.ops as ops
    import network.dist_decoder as dist_decoder
    import network.aggregate_net as aggregate_net
    ns = types.SimpleNamespace(render_ops=render_ops, ops=ops, dist_decoder=dist_decoder,
                               aggregate_net=aggregate_net, ibrnet=ibrnet)
    try:
        import network.renderer as renderer
        ns.renderer = renderer
    except Exception as e:  # pragma: no cover
        ns.renderer = None
        ns.renderer_error = e
    return ns
