"""Drop the HIP render path into the REFERENCE's own classes.

The reference has no FFI layer: its "operator API" for the per-ray path is the method surface of
`network.renderer.NeuralRayBaseRenderer` (render_by_depth / fine_render_impl / render_impl, network/renderer.py:168-226),
reached from render.py:90-95,143-144 (`name2network[cfg['network']](cfg)`, `.load_state_dict`, `.cuda()`, `.eval()`,
`renderer(data)`) and train/trainer.py:50,123 (`self.train_network(train_data)`).  `patch_reference()` grafts the methods
of neuray_amd/network/hip_path.py onto that class, so every subclass the reference registers in `name2network`
(NeuralRayGenRenderer, NeuralRayFtRenderer) - with its own constructor, checkpoint format, dataset layer, init nets and
encoders - renders and back-propagates through libneuray_hip.so.  Nothing of the reference is edited or copied.

    import neuray_amd.integrate as nri
    nri.patch_reference()                     # after the reference tree is importable (it is sys.path[0] for render.py)
    from network.renderer import name2network

or, with the reference's scripts untouched:

    python -m neuray_amd.launch render.py --cfg configs/gen/neuray_gen_depth.yaml --database ... --pose_type eval
    python -m neuray_amd.launch run_training.py --cfg configs/train/ft/...yaml

Options swap the neighbouring per-image pieces too (each is the same arithmetic on a HIP kernel / channels-last conv):
  render_ops=True   network.render_ops' 12 free functions -> stand-alone kernels (neuray_amd/network/render_ops.py)
  init_nets=True    DepthInitNet.get_diff_feats -> neuray_diff_feats
  render_loop=True  NeuralRayBaseRenderer.render (renderer.py:228-254) -> the mirror's: the same encoders-once + ray-batch
                    loop, with the query view of the self-hit loss batched into the reference views' encoder pass (half the launches of
                    a training step's encoders) and encoder outputs accepted from the caller (`img_feats` already in ref_imgs_info)
  ft_host=True      NeuralRayFtRenderer.slice_imgs_info (renderer.py:469-491) -> the resident-scene host path: the scene's views are
                    moved to the device once instead of `to_cuda(imgs_info_slice(...))` of 100 MB per step, the ray sampler works from
                    cached pixel lists (same np.random draws, same rays), no synchronous copy is left in a step
"""
import importlib
import sys

from .network.hip_path import HOT_PATH_METHODS, HipRenderPath

_PATCHED = {}


def patch_renderer_class(cls):
    """Graft the HIP hot path onto `cls` (the reference's NeuralRayBaseRenderer or anything shaped like it: instances carry
    cfg / dist_decoder / agg_net [/ fine_dist_decoder / fine_agg_net]).  Idempotent; returns the saved originals."""
    if cls in _PATCHED:
        return _PATCHED[cls]
    saved = {}
    for name in HOT_PATH_METHODS:
        if name in cls.__dict__:
            saved[name] = cls.__dict__[name]
        setattr(cls, name, HipRenderPath.__dict__[name])
    _PATCHED[cls] = saved
    return saved


def unpatch_renderer_class(cls):
    saved = _PATCHED.pop(cls, None)
    if saved is None:
        return
    for name in HOT_PATH_METHODS:
        if name in saved:
            setattr(cls, name, saved[name])
        elif name in cls.__dict__:
            delattr(cls, name)


def patch_reference(renderer_module=None, render_ops=False, init_nets=False, ft_host=False, render_loop=False):
    """Patch the reference's `network.renderer` (imported here if it is not yet; the reference tree must be importable).
    -> the patched module."""
    mod = renderer_module if renderer_module is not None else (
        sys.modules.get('network.renderer') or importlib.import_module('network.renderer'))
    patch_renderer_class(mod.NeuralRayBaseRenderer)
    if render_loop:
        patch_render_loop(mod.NeuralRayBaseRenderer)
    if ft_host:
        patch_ft_host(mod.NeuralRayFtRenderer, cache_encoded_views=render_loop)
    if render_ops:
        from .network import render_ops as hip_ops
        ref_ops = importlib.import_module('network.render_ops')
        for name in hip_ops.__all__:
            if hasattr(ref_ops, name):
                # (setdefault: a second patch_reference() call must not record the installed HIP function as the original)
                _PATCHED.setdefault(ref_ops, {}).setdefault(name, getattr(ref_ops, name))
                setattr(ref_ops, name, getattr(hip_ops, name))
                if hasattr(mod, name):           # `from network.render_ops import *` in network/renderer.py:17
                    setattr(mod, name, getattr(hip_ops, name))
    if init_nets:
        from .network import init_net as hip_init
        ref_init = importlib.import_module('network.init_net')
        _PATCHED.setdefault(ref_init, {}).setdefault('get_diff_feats', ref_init.get_diff_feats)
        ref_init.get_diff_feats = hip_init.get_diff_feats
    return mod


RENDER_LOOP_METHODS = ('render', 'encode_views', '_early_query')
_PATCHED_LOOP = {}


def patch_render_loop(cls):
    """Graft the mirror's render() (row a20) and encode_views() onto `cls` (the reference's NeuralRayBaseRenderer).  Idempotent."""
    if cls in _PATCHED_LOOP:
        return _PATCHED_LOOP[cls]
    from .network.renderer import NeuralRayBaseRenderer as ours
    saved = {name: cls.__dict__[name] for name in RENDER_LOOP_METHODS if name in cls.__dict__}
    for name in RENDER_LOOP_METHODS:
        setattr(cls, name, ours.__dict__[name])
    _PATCHED_LOOP[cls] = saved
    return saved


def unpatch_render_loop(cls):
    saved = _PATCHED_LOOP.pop(cls, None)
    if saved is None:
        return
    for ft_cls in _PATCHED_FT:                       # (their cached encoder outputs were for the loop that is going away)
        if cls in ft_cls.__mro__:
            ft_cls.cache_encoded_views = False
    for name in RENDER_LOOP_METHODS:
        if name in saved:
            setattr(cls, name, saved[name])
        elif name in cls.__dict__:
            delattr(cls, name)


FT_HOST_METHODS = ('slice_imgs_info', '_ref_views', '_resident', '_encoded', '_device')
_PATCHED_FT = {}


def patch_ft_host(ft_cls, cache_encoded_views=False):
    """Graft the resident-scene host path of neuray_amd's NeuralRayFtRenderer onto `ft_cls` (the reference's NeuralRayFtRenderer or
    anything shaped like it: instances carry cfg / ref_imgs_info / val_imgs_info as dicts of host tensors / ray_feats).  The class
    keeps its own train_step / validate_step / render; a seeded run draws the same views and rays.  cache_encoded_views: in eval
    hand render() cached per-view encoder outputs - only with a render() that accepts them (patch_render_loop).  Idempotent."""
    # cached per-view encoder outputs are only sound under a render() that accepts them: the reference's own render() would run
    # vis_encoder on already encoded ray_feats (wrong images, no error), so the flag follows the render loop actually in place
    loop_patched = any(base in _PATCHED_LOOP for base in ft_cls.__mro__)
    if ft_cls in _PATCHED_FT:
        ft_cls.cache_encoded_views = bool(cache_encoded_views) and loop_patched     # a repeat call updates the flag
        return _PATCHED_FT[ft_cls]
    from .network.renderer import NeuralRayFtRenderer as ours
    saved = {}
    for name in FT_HOST_METHODS + ('cache_encoded_views',):
        if name in ft_cls.__dict__:
            saved[name] = ft_cls.__dict__[name]
    for name in FT_HOST_METHODS:
        setattr(ft_cls, name, ours.__dict__[name])
    ft_cls.cache_encoded_views = bool(cache_encoded_views) and loop_patched      # (the reference's own render() runs the encoders unconditionally, renderer.py:229-235)
    _PATCHED_FT[ft_cls] = saved
    return saved


def unpatch_ft_host(ft_cls):
    saved = _PATCHED_FT.pop(ft_cls, None)
    if saved is None:
        return
    for name in FT_HOST_METHODS + ('cache_encoded_views',):
        if name in saved:
            setattr(ft_cls, name, saved[name])
        elif name in ft_cls.__dict__:
            delattr(ft_cls, name)


def unpatch_reference(renderer_module=None):
    mod = renderer_module if renderer_module is not None else sys.modules.get('network.renderer')
    if mod is not None:
        unpatch_renderer_class(mod.NeuralRayBaseRenderer)
        unpatch_render_loop(mod.NeuralRayBaseRenderer)
        if hasattr(mod, 'NeuralRayFtRenderer'):
            unpatch_ft_host(mod.NeuralRayFtRenderer)
    for target in [t for t in _PATCHED if not isinstance(t, type)]:
        for name, fn in _PATCHED.pop(target).items():
            setattr(target, name, fn)
            if mod is not None and hasattr(mod, name) and target.__name__ == 'network.render_ops':
                setattr(mod, name, fn)
