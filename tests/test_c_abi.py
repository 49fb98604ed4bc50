"""The C-ABI shared library loads and exports every symbol include/neuray_hip.h declares (CPU: no compute)."""
import os
import re

import pytest

from neuray_amd import _lib
from neuray_amd import build as nbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'neuray_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(neuray_[a-z_0-9]+)\s*\(', text)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_product_library_exports_abi():
    path = nbuild.build()          # hipcc cross-compiles gfx950 without a GPU
    lib = _lib.bind(path)          # raises AttributeError on a missing symbol
    assert lib.neuray_abi_version() == 5
    assert lib.neuray_is_device_build() == 1
    assert lib.neuray_packed_pass_floats() > 30000


def test_engine_refuses_cpu_device():
    from neuray_amd.engine import RenderEngine
    with pytest.raises(RuntimeError):
        RenderEngine('cpu')


def test_state_dict_surface_matches_reference():
    """Key names/shapes of the hot-path modules equal the reference's (golden weights were dumped from it)."""
    import torch
    from conftest import load_weights
    from neuray_amd.network.renderer import NeuralRayBaseRenderer
    for vis in (False, True):
        w = load_weights(vis)
        r = NeuralRayBaseRenderer({'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': vis}})
        sd = r.state_dict()
        assert sorted(sd) == sorted(w)
        for k in w:
            assert tuple(sd[k].shape) == tuple(w[k].shape), k
        r.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
