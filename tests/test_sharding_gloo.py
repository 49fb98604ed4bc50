"""N > 1 path on CPU: world_size 2, gloo.  Each rank renders its ray shard through the (emulated) HIP kernels, the
tiles are all-gathered, and the result must equal the single-process render bit for bit (ray independence +
batching invariance)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_case, load_weights
from emu_util import emu_lib, to_torch
from neuray_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), NEURAY_EMU_THREADS='2')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from test_render_parity import make_renderer
    cfg, que, ref, out, mid, extra = load_case('a_small')
    r, dev = make_renderer({**cfg, 'ray_batch_num': 8, 'hip_min_ray_batch': 0}, load_weights(False), 'emu')
    with torch.no_grad():
        full = parallel.render_image_sharded(r, to_torch(que), to_torch(ref))
    if rank == 0:
        np.savez(os.path.join(out_dir, 'sharded.npz'), **{k: v.numpy() for k, v in full.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (1, 7, 40, 640000):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1
    assert parallel.images_for_rank(10, 1, 4) == [1, 5, 9]


def test_two_rank_sharded_render_is_bitwise_identical(tmp_path):
    emu_lib()   # build once in the parent
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), 'sharded.npz'))
    from test_render_parity import make_renderer
    cfg, que, ref, out, mid, extra = load_case('a_small')
    r, dev = make_renderer(cfg, load_weights(False), 'emu')
    with torch.no_grad():
        single = r.render(to_torch(que), to_torch(ref), False)
    for k, v in single.items():
        assert np.array_equal(got[k], v.numpy()), k
    assert np.max(np.abs(got['pixel_colors_nr'] - out['pixel_colors_nr'])) <= 2e-4


# ---- data-parallel training step: rays split over 2 ranks, one gradient all-reduce ------------------------------
def _train_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), NEURAY_EMU_THREADS='2')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from conftest import GOLDEN_DIR
    from neuray_amd.network.renderer import NeuralRayBaseRenderer
    z = np.load(os.path.join(GOLDEN_DIR, 'case_g_grads.npz'))
    cfg = __import__('ast').literal_eval(str(z['cfg_json']))
    r = NeuralRayBaseRenderer(cfg)
    r.load_state_dict({k: torch.from_numpy(v) for k, v in load_weights(False).items()}, strict=True)
    r.train()
    r._engine_test_lib = emu_lib()
    rn = z['que.coords'].shape[1]
    s, e = parallel.shard_range(rn, rank, world)
    que = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('que.') and k != 'que.Ks_inv'}
    ref = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith('ref.')}
    que['coords'] = que['coords'][:, s:e].contiguous()
    u = torch.from_numpy(z['u'])[:, s:e].contiguous()
    real_rand = torch.rand
    torch.rand = lambda *a, **k: u.clone()          # this rank's slice of the uniforms the reference drew
    try:
        out = r.render_impl(que, ref, True)
    finally:
        torch.rand = real_rand
    keys = ('pixel_colors_nr', 'pixel_colors_nr_fine', 'hit_prob_self', 'hit_prob_self_fine')
    loss = sum((torch.from_numpy(z['lw.' + k])[:, s:e] * out[k]).sum() for k in keys)
    loss.backward()
    parallel.allreduce_gradients(r.parameters(), average=False)
    if rank == 0:
        np.savez(os.path.join(out_dir, 'grads.npz'), **{k: p.grad.numpy() for k, p in r.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_step_matches_reference_gradients(tmp_path):
    """Each rank back-propagates its half of the rays through the (emulated) HIP backward kernels; after ONE
    all-reduce (sum) the gradients equal the reference's single-process autograd over all rays
    (tests/golden/case_g_grads.npz)."""
    from conftest import GOLDEN_DIR
    emu_lib()
    port = 31500 + os.getpid() % 2000
    mp.spawn(_train_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), 'grads.npz'))
    z = np.load(os.path.join(GOLDEN_DIR, 'case_g_grads.npz'))
    for k in got.files:
        want = z['grad.' + k]
        scale = max(1e-3, float(np.abs(want).max()))
        assert np.max(np.abs(got[k] - want)) <= 5e-3 * scale, k


# ---- fine-tuning mode: per-view feature maps get gradients on the ranks that rendered from them -------------------
def _ft_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), NEURAY_EMU_THREADS='2')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from conftest import GOLDEN_DIR
    from test_scene_renderers import make_ft
    gold = np.load(os.path.join(GOLDEN_DIR, 'case_scene.npz'))
    ft = make_ft(gold, 'cpu').train()                    # identical replicas on every rank
    shared = [p for n, p in ft.named_parameters() if not n.startswith('ray_feats.')]
    opt = torch.optim.Adam(ft.parameters(), lr=1e-2)
    log = []
    for step in range(2):
        np.random.seed(100 * step + rank)                # every rank draws its own query view / neighbours / rays
        torch.manual_seed(100 * step + rank)
        opt.zero_grad(set_to_none=True)
        t = ft.train_step()
        (((t['pixel_colors_nr_fine'] - t['pixel_colors_gt']) ** 2).mean() + t['hit_prob_self_fine'].mean()).backward()
        mine = list(ft.touched_views)
        parallel.allreduce_gradients(shared)
        union = parallel.allreduce_scene_feature_gradients(ft.ray_feats, mine, ft.cfg['neighbor_view_num'] + 1)
        with_grad = [i for i, p in enumerate(ft.ray_feats) if p.grad is not None]
        log.append((mine, union, with_grad))
        opt.step()
    state = torch.cat([p.detach().reshape(-1) for p in ft.parameters()])
    torch.save({'log': log, 'state': state}, os.path.join(out_dir, 'ft_rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_finetuning_exchanges_only_touched_feature_maps(tmp_path):
    """SURVEY.md 8(e) caveat: in ft mode each rank touches <= neighbor_view_num + 1 of the scene's per-view maps; the
    ranks exchange the union's gradients only, views outside it keep grad None (so Adam skips them on every rank), and
    the replicas stay bit-identical after the optimiser steps."""
    emu_lib()
    port = 33500 + os.getpid() % 2000
    mp.spawn(_ft_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'ft_rank0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'ft_rank1.pt'))
    assert torch.equal(a['state'], b['state'])
    differed = False
    for (mine_a, union_a, grad_a), (mine_b, union_b, grad_b) in zip(a['log'], b['log']):
        assert union_a == union_b == sorted(set(mine_a) | set(mine_b)) == grad_a == grad_b
        assert len(union_a) < 6 or mine_a != mine_b
        differed |= mine_a != mine_b
    assert differed            # the ranks did render from different views


# ---- the same through parallel.train_step on a PATCHED reference-shaped ft class (which has no `touched_views`) -------
def _patched_ft_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), NEURAY_EMU_THREADS='2')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from conftest import GOLDEN_DIR
    import test_integrate as ti
    from neuray_amd import integrate
    import ref_harness
    if ref_harness.reference_available():            # the real class where the tree exists, the stand-in elsewhere
        ref_harness.import_reference()
    else:
        sys.path.insert(0, ti.STUB_ROOT)
    mod = integrate.patch_reference()
    mod.to_cuda = lambda d: d
    gold = np.load(os.path.join(GOLDEN_DIR, 'case_scene.npz'))
    ft, n = ti.build_ft(mod, gold)
    assert not hasattr(ft, 'touched_views')
    ft._engine_test_lib = emu_lib()
    ft.train()
    opt = torch.optim.Adam(ft.parameters(), lr=1e-2)
    log = []

    def loss_fn(t):
        return ((t['pixel_colors_nr_fine'] - t['pixel_colors_gt']) ** 2).mean() + t['hit_prob_self_fine'].mean()
    for step in range(2):
        np.random.seed(100 * step + rank)
        torch.manual_seed(100 * step + rank)
        parallel.train_step(ft, {'index': 0}, loss_fn, opt)
        log.append([i for i, p in enumerate(ft.ray_feats) if p.grad is not None])
    state = torch.cat([p.detach().reshape(-1) for p in ft.parameters()])
    torch.save({'log': log, 'state': state, 'cls': mod.__file__}, os.path.join(out_dir, 'pft_rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_step_on_a_patched_reference_ft_renderer(tmp_path):
    """the reference's NeuralRayFtRenderer has a `ray_feats` ParameterList but no `touched_views`;
    parallel.train_step reads the touched set off the gradients and the ranks agree on the exchange length themselves.
    Replicas stay bit-identical and only the union's maps carry a gradient."""
    emu_lib()
    port = 35500 + os.getpid() % 2000
    mp.spawn(_patched_ft_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'pft_rank0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'pft_rank1.pt'))
    assert torch.equal(a['state'], b['state'])
    for ga, gb in zip(a['log'], b['log']):
        assert ga == gb and 0 < len(ga) <= 8


# ---- single-image latency path: encoder phase sharded by view + ONE all-gather of the maps --------------
def _enc_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), NEURAY_EMU_THREADS='2')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from neuray_amd import synthetic
    from neuray_amd.network.renderer import NeuralRayBaseRenderer
    from test_encoders import fill_by_name
    cfg = {'use_hierarchical_sampling': True, 'dist_decoder_cfg': {'use_vis': False}, 'depth_sample_num': 8,
           'fine_depth_sample_num': 8, 'agg_net_cfg': {'sample_num': 8}, 'fine_agg_net_cfg': {'sample_num': 8},
           'build_encoders': True, 'ray_batch_num': 16, 'hip_min_ray_batch': 0}
    torch.manual_seed(0)
    r = NeuralRayBaseRenderer(cfg).eval()
    fill_by_name(r)
    r._engine_test_lib = emu_lib()
    que, ref = synthetic.make_scene(48, 64, 3, seed=5)              # 3 views over 2 ranks: an uneven split (2 + 1)
    que['coords'] = (np.random.RandomState(6).rand(1, 23, 2) * np.array([63, 47])).astype(np.float32)
    tq = {k: torch.from_numpy(v) for k, v in que.items()}
    tr = {k: torch.from_numpy(v) for k, v in ref.items() if k != 'img_feats'}        # images + INITIAL ray_feats only
    calls = {'n': 0}
    enc = r.image_encoder.forward

    def counting(x):
        calls['n'] += x.shape[0]
        return enc(x)
    r.image_encoder.forward = counting
    with torch.no_grad():
        full = parallel.render_image_sharded(r, dict(tq), dict(tr))
    encoded_here = calls['n']
    with torch.no_grad():
        single = r.render(dict(tq), dict(tr), False)                # the replicated path: this process encodes all views
        # the maps the collective must have delivered: each rank's views encoded as ITS batch (2 + 1), concatenated
        f_img = torch.cat([enc(tr['imgs'][s:e]) for s, e in ((0, 2), (2, 3))], 0)
        f_ray = torch.cat([r.vis_encoder(tr['ray_feats'][s:e], f_img[s:e]) for s, e in ((0, 2), (2, 3))], 0)
        local = parallel.render_ray_shard(r, dict(tq), dict(tr, img_feats=f_img, ray_feats=f_ray), 0, 1)[0]
        maps = parallel.encode_views_sharded(r, dict(tr))
    torch.save({'sharded': full, 'single': single, 'encoded': encoded_here, 'from_same_maps': local,
                'maps_equal': bool(torch.equal(maps['img_feats'], f_img) and torch.equal(maps['ray_feats'], f_ray))},
               os.path.join(out_dir, 'enc_rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_view_sharded_encoders_equal_the_replicated_path(tmp_path):
    """every rank encodes only its views (2 + 1 of 3) and one all-gather hands everyone all maps: the gathered maps equal the
    per-shard encodings bit for bit, the sharded image equals a single-process render from those maps bit for bit, and it
    equals the replicated path (one process encoding all three views as ONE batch) to fp32 rounding - a convolution's
    summation order may depend on the batch size (oneDNN here, MIOpen on the GPU), nothing else differs"""
    emu_lib()
    port = 37500 + os.getpid() % 2000
    mp.spawn(_enc_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'enc_rank0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'enc_rank1.pt'))
    assert (a['encoded'], b['encoded']) == (2, 1)
    for res in (a, b):
        assert res['maps_equal']
        assert set(res['sharded']) == set(res['single']) == set(res['from_same_maps'])
        for k, v in res['single'].items():
            assert torch.equal(res['sharded'][k], res['from_same_maps'][k]), k
            tol = 1e-3 if k.endswith('_fine') else 2e-5           # (chained coarse -> fine: DESIGN.md 2.4)
            assert torch.allclose(res['sharded'][k].float(), v.float(), atol=tol), (k, float((res['sharded'][k].float() - v.float()).abs().max()))
    for k in a['sharded']:
        assert torch.equal(a['sharded'][k], b['sharded'][k]), k            # every rank holds the same full image


# ---- the gradient flags without a read-back: steady state, a local change, an unannounced remote change ------------------------------
def _flags_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.ModuleDict({'a': torch.nn.Linear(4, 4), 'b': torch.nn.Linear(4, 4), 'c': torch.nn.Linear(4, 4)})
    x = torch.full((2, 4), float(rank + 1))
    log = []

    def step(use_b, use_c):
        for p in net.parameters():
            p.grad = None
        y = net['a'](x)
        if use_b:
            y = y + net['b'](x)
        if use_c:
            y = y + net['c'](x)
        y.sum().backward()
        parallel.allreduce_gradients(net.parameters(), average=False)
        log.append([p.grad is None for p in net.parameters()] + [float(net['a'].weight.grad.sum())])

    step(True, False)          # first step: flags read at the call; c has no gradient anywhere -> stays None
    step(True, False)          # steady state: the previous union, own flags copied for the check
    assert parallel._FLAG_STATE and any('pending' in st for st in parallel._FLAG_STATE.values())
    step(True, False)          # ... whose check passes here
    step(True, True)           # the LOCAL pattern changes on both ranks: the call notices (host-side) and reads the flags again
    step(True, True)
    parallel.check_deferred_flags()
    # a rank changes its pattern while the other does not: the unchanged rank learns it one call late and says so
    err = None
    try:
        step(True, rank == 0)  # rank 1 drops c: its own pattern changed -> it re-reads (union still has c, from rank 0); rank 0 keeps the old union: fine
        step(True, rank == 0)
        step(rank == 0, rank == 0)      # rank 1 drops b too; rank 0's pattern is unchanged and b is still in the union via rank 0: consistent
        parallel.check_deferred_flags()
    except RuntimeError as e:           # noqa: BLE001
        err = str(e)
    np.save(os.path.join(out_dir, 'log%d.npy' % rank), np.array(log, dtype=np.float64))
    with open(os.path.join(out_dir, 'err%d.txt' % rank), 'w') as f:
        f.write(err or '')
    dist.barrier()
    dist.destroy_process_group()


def _flags_mismatch_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.ModuleDict({'a': torch.nn.Linear(4, 4), 'b': torch.nn.Linear(4, 4)})
    x = torch.ones(2, 4)

    def step(use_b):
        for p in net.parameters():
            p.grad = None
        y = net['a'](x) + (net['b'](x) if use_b else 0.0)
        y.sum().backward()
        parallel.allreduce_gradients(net.parameters(), average=False)

    step(rank == 0)            # only rank 0 reaches b: the union has it
    step(rank == 0)
    step(False)                # rank 0 drops b (re-reads: b is gone); rank 1's own pattern is unchanged -> it still assumes b this step ...
    msg = ''
    try:
        parallel.check_deferred_flags()        # ... and is told here (or by its next allreduce_gradients call), one call late
    except RuntimeError as e:
        msg = str(e)
    with open(os.path.join(out_dir, 'msg%d.txt' % rank), 'w') as f:
        f.write(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_flags_need_no_read_back_in_the_steady_state(tmp_path):
    """allreduce_gradients decides `grad = None` from the previous step's union of flags (no device -> host wait); a change of the
    rank's own pattern is seen at once, a change on another rank alone is reported by the next call"""
    port = 33500 + os.getpid() % 2000
    mp.spawn(_flags_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    logs = [np.load(os.path.join(str(tmp_path), 'log%d.npy' % r)) for r in range(2)]
    for r in range(2):
        assert open(os.path.join(str(tmp_path), 'err%d.txt' % r)).read() == ''
    # parameters: a.w a.b b.w b.b c.w c.b ; steps 0-2: c is None everywhere; steps 3-4: nothing is None; same gradients on both ranks
    for r in range(2):
        assert logs[r][0][:6].tolist() == [0, 0, 0, 0, 1, 1] and logs[r][2][:6].tolist() == [0, 0, 0, 0, 1, 1]
        assert logs[r][4][:6].tolist() == [0, 0, 0, 0, 0, 0]
        assert logs[r][7][:6].tolist() == [0, 0, 0, 0, 0, 0]          # b and c live on through rank 0
    assert np.array_equal(logs[0][:, 6], logs[1][:, 6])
    port += 7
    mp.spawn(_flags_mismatch_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(os.path.join(str(tmp_path), 'msg0.txt')).read() == ''
    assert 'changed' in open(os.path.join(str(tmp_path), 'msg1.txt')).read()
