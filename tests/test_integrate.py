"""The drop-in boundary: neuray_amd.integrate.patch_reference() grafts the HIP per-ray methods onto the
REFERENCE's own `network.renderer.NeuralRayBaseRenderer`, so that what render.py:90-95,143-144 and train/trainer.py:50,123
do - `name2network[cfg['network']](cfg)`, `load_state_dict`, `renderer(data)`, `loss.backward()` - runs on the kernels.

Two legs:
  * `ref`  - the real reference tree (/root/reference, imported through tests/golden/ref_harness.py: only third-party
             stubs, nothing of the reference edited), kernels on the CPU emulator.  Skipped where the tree is absent.
  * `stub` - tests/ref_stub/network/renderer.py, a stand-in shaped like the reference module, for the GPU box (where the
             reference tree does not exist): same patch call, kernels on the emulator (CPU) or libneuray_hip.so (GPU).
Expected values are the committed goldens, which the UNPATCHED reference produced (tests/golden/make_golden.py).
"""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, ROOT, load_case, load_weights
from emu_util import emu_lib
from test_encoders import fill_by_name
from neuray_amd import integrate, synthetic

sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import ref_harness  # noqa: E402

STUB_ROOT = os.path.join(ROOT, 'tests', 'ref_stub')
LEGS = [('ref', 'cpu'), ('stub', 'cpu'), pytest.param(('stub', 'cuda:0'), marks=pytest.mark.gpu)]
SMALL = {'use_hierarchical_sampling': True, 'depth_sample_num': 8, 'fine_depth_sample_num': 8,
         'agg_net_cfg': {'sample_num': 8}, 'fine_agg_net_cfg': {'sample_num': 8}, 'ray_batch_num': 16}


def _drop_network_modules():
    for k in [k for k in sys.modules if k == 'network' or k.startswith('network.')]:
        del sys.modules[k]


@pytest.fixture(params=LEGS, ids=lambda p: '%s-%s' % p)
def patched(request):
    """-> (patched `network.renderer` module, device): exactly the two lines INTEGRATION.md tells a maintainer to add"""
    which, dev = request.param
    _drop_network_modules()
    if which == 'ref':
        if not ref_harness.reference_available():
            pytest.skip('reference tree not present')
        ref_harness.import_reference()                       # stubs for cv2 / skimage / ..., reference root on sys.path
        root = ref_harness.REFERENCE_ROOT
    else:
        root = STUB_ROOT
        while ref_harness.REFERENCE_ROOT in sys.path:
            sys.path.remove(ref_harness.REFERENCE_ROOT)
        sys.path.insert(0, root)
        _drop_network_modules()
    mod = integrate.patch_reference()                        # <- the drop-in
    assert os.path.abspath(mod.__file__).startswith(os.path.abspath(root))
    yield mod, dev
    integrate.unpatch_reference(mod)
    if root in sys.path:
        sys.path.remove(root)
    _drop_network_modules()


def place(r, dev):
    if dev == 'cpu':
        r._engine_test_lib = emu_lib()        # CPU test-suite hook: the same kernel sources on the fiber emulator
        return r
    return r.cuda()                            # render.py:93


def frac_within(a, b, tol):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).reshape(-1, a.shape[-1]).max(1)
    return float(np.mean(d <= tol)), float(d.max())


def test_reference_base_renderer_on_hip_path_reproduces_reference_goldens(patched):
    """the reference's class, the reference's constructor, the reference-generated weights and outputs"""
    mod, dev = patched
    cfg, que, ref, want, _, _ = load_case('b_default')
    r = mod.NeuralRayBaseRenderer(cfg)
    for attr in ('_engine', '_packed', '_engine_test_lib'):            # nothing of ours was set by that constructor
        assert attr not in r.__dict__
    missing, unexpected = r.load_state_dict({k: torch.from_numpy(v) for k, v in load_weights().items()}, strict=False)
    assert not unexpected and all(k.startswith(('image_encoder.', 'vis_encoder.', 'sph_fitter.')) for k in missing)
    r = place(r.eval(), dev)
    tq = {k: torch.from_numpy(v).to(dev) for k, v in que.items() if k != 'Ks_inv'}      # the reference has no such key
    tr = {k: torch.from_numpy(v).to(dev) for k, v in ref.items()}
    with torch.no_grad():
        got = r.render_impl(tq, tr, False)                                # renderer.py:217
    assert set(got) == set(want)
    assert np.array_equal(got['ray_mask'].cpu().numpy(), want['ray_mask'])
    assert frac_within(got['pixel_colors_nr'].cpu().numpy(), want['pixel_colors_nr'], 2e-4)[0] == 1.0
    assert frac_within(got['hit_prob_nr'].cpu().numpy(), want['hit_prob_nr'], 1e-4)[0] == 1.0
    f, worst = frac_within(got['pixel_colors_nr_fine'].cpu().numpy(), want['pixel_colors_nr_fine'], 2e-4)
    assert f >= 0.95 and worst < 0.05, (f, worst)                         # chained coarse -> fine: DESIGN.md 2.4


def test_gen_renderer_forward_through_name2network(patched):
    """render.py:90-95,143-144: name2network[...](cfg) -> eval -> renderer(data); encoders are the module's own"""
    mod, dev = patched
    gold = np.load(os.path.join(GOLDEN_DIR, 'case_scene.npz'))
    if hasattr(mod, 'name2init_net'):                     # the real Gen renderer builds an init_net from this registry

        class PassThrough(torch.nn.Module):
            def __init__(self, cfg):
                super().__init__()

            def forward(self, ref_imgs_info, src_imgs_info, is_train):
                return ref_imgs_info['ray_feats']
        mod.name2init_net['passthrough'] = PassThrough
    r = mod.name2network['neuray_gen']({**SMALL, 'init_net_type': 'passthrough', 'depth_loss_coords_num': 40}).eval()
    fill_by_name(r)
    r = place(r, dev)
    que, ref = synthetic.make_scene(48, 64, 3, seed=5)
    que['coords'] = (np.random.RandomState(6).rand(1, 23, 2) * np.array([63, 47])).astype(np.float32)
    ref.pop('img_feats')
    torch.manual_seed(11)
    with torch.no_grad():
        out = r({'que_imgs_info': {k: torch.from_numpy(v).to(dev) for k, v in que.items()},
                 'ref_imgs_info': {k: torch.from_numpy(v).to(dev) for k, v in ref.items()}, 'eval': True})
    tol = 2e-4 if dev == 'cpu' else 1e-3        # (the encoders run on MIOpen on the GPU)
    assert frac_within(out['pixel_colors_nr'].cpu().numpy(), gold['gen_pixel_colors_nr'], tol)[0] == 1.0
    f, worst = frac_within(out['pixel_colors_nr_fine'].cpu().numpy(), gold['gen_pixel_colors_nr_fine'], tol)
    assert f >= 0.95 and worst < 0.1, (f, worst)
    assert np.mean(out['ray_mask_fine'].cpu().numpy() == gold['gen_ray_mask_fine']) >= 0.95
    assert not any(k.startswith('hit_prob') for k in out)                 # eval drops them (renderer.py:244)


def test_training_step_backpropagates_through_the_patched_class(patched):
    """train/trainer.py:123: outputs of the patched class carry a grad_fn; gradients equal the reference's own autograd
    (tests/golden/case_g_grads.npz)"""
    mod, dev = patched
    z = np.load(os.path.join(GOLDEN_DIR, 'case_g_grads.npz'))
    cfg = __import__('ast').literal_eval(str(z['cfg_json']))      # noqa: S307  repr() of a plain dict written by make_golden.py
    r = mod.NeuralRayBaseRenderer(cfg)
    r.load_state_dict({k: torch.from_numpy(v) for k, v in load_weights().items()}, strict=False)
    r = place(r.train(), dev)
    que = {k[4:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith('que.') and k != 'que.Ks_inv'}
    ref = {k[4:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith('ref.')}
    for t in (ref['ray_feats'], ref['img_feats'], que['ray_feats']):
        t.requires_grad_(True)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: torch.from_numpy(z['u']).clone()         # the uniforms the reference drew
    try:
        out = r.render_impl(que, ref, True)
    finally:
        torch.rand = real_rand
    lw = {k[3:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith('lw.')}
    sum((lw[k] * out[k]).sum() for k in lw).backward()
    checked = 0
    for name, p in r.named_parameters():
        if 'grad.' + name in z.files:
            want = z['grad.' + name]
            got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(want)
            assert np.abs(got - want).max() <= 5e-3 * max(np.abs(want).max(), 1e-3), name
            checked += 1
    assert checked >= 60
    for t, k in ((ref['ray_feats'], 'grad.ref.ray_feats'), (ref['img_feats'], 'grad.ref.img_feats'), (que['ray_feats'], 'grad.que.ray_feats')):
        assert np.abs(t.grad.cpu().numpy() - z[k]).max() <= 5e-3 * np.abs(z[k]).max(), k


def test_unpatch_restores_the_original_methods(patched):
    mod, _ = patched
    cls = mod.NeuralRayBaseRenderer
    assert cls.render_impl is integrate.HipRenderPath.__dict__['render_impl']
    integrate.unpatch_reference(mod)
    assert cls.render_impl is not integrate.HipRenderPath.__dict__['render_impl'] and 'engine' not in cls.__dict__
    integrate.patch_reference(mod)
    assert cls.render_impl is integrate.HipRenderPath.__dict__['render_impl']


def test_launcher_runs_an_unmodified_script(tmp_path, patched):
    """python -m neuray_amd.launch <script>: the script only knows the reference's names (as render.py does)"""
    mod, dev = patched
    integrate.unpatch_reference(mod)                  # the launcher has to do the patching itself
    script = str(tmp_path / 'probe.py')               # (never next to the reference's scripts: that tree is not ours to write to)
    with open(script, 'w') as f:
        f.write("from network.renderer import name2network\n"
                "cls = name2network['neuray_gen']\n"
                "RESULT = cls.render_impl.__module__\n")
    try:
        from neuray_amd import launch
        keep = list(sys.path)
        ns = launch.run(script)
        sys.path[:] = keep
        # the option flags of the command line reach patch_reference
        with open(script, 'w') as f:
            f.write("from network.renderer import name2network\n"
                    "RESULT = (name2network['neuray_ft'].slice_imgs_info.__module__, name2network['neuray_gen'].render.__module__,\n"
                    "          name2network['neuray_ft'].cache_encoded_views)\n")
        integrate.unpatch_reference(mod)
        launch_main_ns = {}
        real_run = launch.run
        launch.run = lambda *a, **k: launch_main_ns.update(real_run(*a, **k))
        try:
            os.environ.pop('NEURAY_HIP_ARITH', None)
            launch.main(['--ft-host', '--render-loop', '--arith', 'x3', script])
            assert os.environ.pop('NEURAY_HIP_ARITH') == 'x3'             # (what HipRenderPath.engine reads: the scripts stay unchanged)
        finally:
            launch.run = real_run
        sys.path[:] = keep
    finally:
        os.remove(script)
        integrate.unpatch_reference(mod)
    assert ns['RESULT'] == 'neuray_amd.network.hip_path'
    assert launch_main_ns['RESULT'] == ('neuray_amd.network.renderer', 'neuray_amd.network.renderer', True)
    assert 'slice_imgs_info' in mod.NeuralRayFtRenderer.__dict__ and mod.NeuralRayFtRenderer.slice_imgs_info.__module__ != 'neuray_amd.network.renderer'


def build_ft(mod, gold, extra_cfg=None):
    """an instance of `mod.NeuralRayFtRenderer` on the scene of tests/golden/case_scene.npz, built the way
    tests/golden/make_golden.py scene_case builds the reference's (no dataset on disk: attributes set by hand)"""
    import torch.nn as nn
    ft_cfg = {**SMALL, 'use_self_hit_prob': True, 'neighbor_view_num': 3, 'neighbor_pool_ratio': 1, 'train_ray_num': 12,
              'foreground_ratio': 0.5, 'include_self_prob': 0.01, 'use_validation': True, **(extra_cfg or {})}
    ft = mod.NeuralRayFtRenderer.__new__(mod.NeuralRayFtRenderer)
    mod.NeuralRayBaseRenderer.__init__(ft, {**mod.NeuralRayFtRenderer.default_cfg, **ft_cfg})
    fill_by_name(ft)
    sref = {k[7:]: gold[k] for k in gold.files if k.startswith('ft_ref_')}
    sval = {k[7:]: gold[k] for k in gold.files if k.startswith('ft_val_') and k[7:] in ('imgs', 'masks', 'poses', 'Ks', 'depth_range')}
    n = sref['imgs'].shape[0]
    ft.ref_ids = np.arange(n)
    ft.ref_imgs_info = {k: torch.from_numpy(v) for k, v in sref.items()}
    ft.val_imgs_info = {k: torch.from_numpy(v) for k, v in sval.items()}
    cen = lambda P: np.asarray([-p[:, :3].T @ p[:, 3] for p in P])      # noqa: E731
    ft.ref_dist_idx = np.argsort(np.linalg.norm(cen(sref['poses'])[None] - cen(sref['poses'])[:, None], 2, 2), 1)
    ft.val_dist_idx = np.argsort(np.linalg.norm(cen(sref['poses'])[None] - cen(sval['poses'])[:, None], 2, 2), 1)
    init = gold['ft_init_ray_feats']
    ft.ray_feats = nn.ParameterList([nn.Parameter(torch.from_numpy(init[i:i + 1].copy())) for i in range(n)])
    return ft, n


def test_reference_ft_renderer_train_and_validate_steps(patched):
    """the reference's NeuralRayFtRenderer (train_step / validate_step / render_pose with its own neighbour selection,
    `imgs_info_slice`, `get_coords_mask`) on the patched base class; scene attributes set as tests/golden/make_golden.py
    scene_case sets them (the constructor reads a dataset from disk).  The stub legs run the stand-in ft class of
    tests/ref_stub (same surface, no `touched_views`), on the MI355X with the module's real `to_cuda`."""
    mod, dev = patched
    if dev == 'cpu' and not hasattr(mod, 'compute_nearest_camera_indices') and ref_harness.reference_available():
        pytest.skip('the stand-in on the CPU emulator repeats the reference leg (40 s); it runs where the reference tree is absent and on the GPU')
    gold = np.load(os.path.join(GOLDEN_DIR, 'case_scene.npz'))
    ft, n = build_ft(mod, gold)
    to_cuda = mod.to_cuda
    if dev == 'cpu':
        mod.to_cuda = lambda d: d
    try:
        ft = place(ft, dev)
        ft.eval()
        v = ft.validate_step(1)
        tol = 2e-4 if dev == 'cpu' else 1e-3        # (the encoders run on MIOpen on the GPU)
        f, worst = frac_within(v['pixel_colors_nr_fine'].cpu().numpy(), gold['ft_val_pixel_colors_nr_fine'], tol)
        assert f >= 0.95 and worst < 0.1, (f, worst)
        ft.train()
        np.random.seed(3)
        torch.manual_seed(4)
        t = ft.train_step()
        assert np.array_equal(t['que_imgs_info']['coords'].cpu().numpy(), gold['ft_train_coords'])
        for k in ('pixel_colors_nr', 'hit_prob_self', 'pixel_colors_gt'):
            assert frac_within(t[k].detach().cpu().numpy(), gold['ft_train_' + k], tol)[0] == 1.0, k
        loss = ((t['pixel_colors_nr_fine'] - t['pixel_colors_gt']) ** 2).mean() + t['hit_prob_self_fine'].mean()
        loss.backward()
        touched = [i for i in range(n) if ft.ray_feats[i].grad is not None and float(ft.ray_feats[i].grad.abs().max()) > 0]
        assert touched == list(gold['ft_train_touched'])
        for i in touched[:2]:
            want = gold['ft_train_grad_%d' % i]
            assert np.abs(ft.ray_feats[i].grad.cpu().numpy() - want).max() <= 5e-3 * float(np.abs(want).max()), i
    finally:
        mod.to_cuda = to_cuda


def test_reference_ft_renderer_with_the_resident_scene_host_path(patched):
    """integrate.patch_ft_host: the ft class's slice_imgs_info replaced by the resident-scene host path (no per-step `to_cuda` of the
    sliced views, cached pixel lists in the ray sampler).  Same np.random draws: the rays equal the reference's golden draws and the
    generator is left in the state the unpatched method leaves it in; outputs and gradients pass the same checks as above."""
    mod, dev = patched
    if dev == 'cpu' and not hasattr(mod, 'compute_nearest_camera_indices') and ref_harness.reference_available():
        pytest.skip('the stand-in on the CPU emulator repeats the reference leg; it runs where the reference tree is absent and on the GPU')
    gold = np.load(os.path.join(GOLDEN_DIR, 'case_scene.npz'))
    ft, n = build_ft(mod, gold)
    to_cuda = mod.to_cuda
    if dev == 'cpu':
        mod.to_cuda = lambda d: d
    try:
        ft = place(ft, dev)
        ft.train()
        np.random.seed(3)
        torch.manual_seed(4)
        ft.train_step()
        state_after = np.random.random()                      # where the unpatched host path leaves the generator
        integrate.patch_ft_host(type(ft))
        assert type(ft).cache_encoded_views is False and integrate.patch_ft_host(type(ft)) is not None      # idempotent
        np.random.seed(3)
        torch.manual_seed(4)
        t = ft.train_step()
        assert np.random.random() == state_after
        assert np.array_equal(t['que_imgs_info']['coords'].cpu().numpy(), gold['ft_train_coords'])
        assert 'ray_feats' not in t['que_imgs_info'] and t['que_imgs_info']['imgs'].device.type == torch.device(dev).type
        tol = 2e-4 if dev == 'cpu' else 1e-3
        for k in ('pixel_colors_nr', 'hit_prob_self', 'pixel_colors_gt'):
            assert frac_within(t[k].detach().cpu().numpy(), gold['ft_train_' + k], tol)[0] == 1.0, k
        loss = ((t['pixel_colors_nr_fine'] - t['pixel_colors_gt']) ** 2).mean() + t['hit_prob_self_fine'].mean()
        loss.backward()
        touched = [i for i in range(n) if ft.ray_feats[i].grad is not None and float(ft.ray_feats[i].grad.abs().max()) > 0]
        assert touched == list(gold['ft_train_touched'])
        # ... and with the mirror's render() (row a20) on the base class: the query view rides through the encoders with the reference
        # views, eval reuses cached per-view encoder outputs
        integrate.unpatch_ft_host(type(ft))
        integrate.patch_render_loop(mod.NeuralRayBaseRenderer)
        integrate.patch_ft_host(type(ft), cache_encoded_views=True)
        ft.train()
        for p_ in ft.parameters():
            p_.grad = None
        np.random.seed(3)
        torch.manual_seed(4)
        t2 = ft.train_step()
        assert np.random.random() == state_after
        for k in ('pixel_colors_nr', 'hit_prob_self', 'pixel_colors_gt'):
            assert frac_within(t2[k].detach().cpu().numpy(), gold['ft_train_' + k], tol)[0] == 1.0, k
        ft.eval()
        v2 = ft.validate_step(1)
        f, worst = frac_within(v2['pixel_colors_nr_fine'].cpu().numpy(), gold['ft_val_pixel_colors_nr_fine'], tol)
        assert f >= 0.95 and worst < 0.1, (f, worst)
        assert len(ft._enc_cache) > 0 and 'img_feats' not in v2['ref_imgs_info'] and 'ray_feats' not in v2['ref_imgs_info']
        entries = {i: hit[1] for i, hit in ft._enc_cache.items()}
        ft._encoded(list(entries))                             # a second pose with the same views is served from the cache
        assert all(ft._enc_cache[i][1] is f_ for i, f_ in entries.items())
    finally:
        integrate.unpatch_ft_host(type(ft))
        integrate.unpatch_render_loop(mod.NeuralRayBaseRenderer)
        mod.to_cuda = to_cuda
    assert 'cache_encoded_views' not in type(ft).__dict__ and '_resident' not in type(ft).__dict__
    assert 'encode_views' not in mod.NeuralRayBaseRenderer.__dict__
