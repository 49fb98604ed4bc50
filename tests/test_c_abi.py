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
    assert lib.neuray_abi_version() == 11
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


def _header_structs():
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import abi_structs
    return abi_structs.parse_structs()


def test_binding_structs_match_the_header_field_by_field():
    """every `typedef struct` of include/neuray_hip.h has a ctypes class of the same name in _lib.py with the same fields,
    types and order (pointers are c_void_p on the Python side)"""
    import ctypes as C
    structs = _header_structs()
    assert len(structs) >= 4
    for name, fields in structs.items():
        cls = getattr(_lib, name)
        got = [(n, 'C.' + t.__name__) for n, t in cls._fields_]
        assert got == fields, name
        assert C.sizeof(cls) % 8 == 0


def test_integration_md_option_b_snippet_matches_the_header():
    """the ctypes stub INTEGRATION.md shows a maintainer stopped at `views_per_wave` while the header's
    struct had grown `saved_dev`.  The document's python block is executed here (with the library load stubbed out) and
    every Structure it defines is compared with the header."""
    import ctypes as C
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    m = re.search(r'```python\n(# network/neuray_hip\.py.*?)```', text, flags=re.S)
    assert m, 'Option B snippet not found'

    abi = int(re.search(r'#define NEURAY_ABI_VERSION (\d+)', open(os.path.join(ROOT, 'include', 'neuray_hip.h')).read()).group(1))

    class FakeFn:
        restype = None

        def __call__(self, *a):
            return abi            # the snippet asserts the version it was generated for

    class FakeLib:
        def __getattr__(self, name):
            return FakeFn()
    real = C.CDLL
    C.CDLL = lambda *_a, **_k: FakeLib()
    try:
        ns = {}
        exec(compile(m.group(1), 'INTEGRATION.md:option-b', 'exec'), ns)          # noqa: S102  our own document
    finally:
        C.CDLL = real
    structs = _header_structs()
    seen = 0
    for name, obj in ns.items():
        if isinstance(obj, type) and issubclass(obj, C.Structure) and obj is not C.Structure:
            assert name in structs, name
            assert [(n, 'C.' + t.__name__) for n, t in obj._fields_] == structs[name], name
            assert C.sizeof(obj) == C.sizeof(getattr(_lib, name))
            seen += 1
    assert seen >= 2
