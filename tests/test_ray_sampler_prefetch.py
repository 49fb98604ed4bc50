"""The fine-tuning step's host-side ray sampler (reference network/renderer.py:521-543 train_step,
utils/base_utils.py:585-603 sample_train_coords): `neuray_mt19937_shuffle` is np.random.shuffle - same permutation, same generator
state afterwards - and the speculative prefetch of the next step's draws never changes what a seeded run draws, whoever else touches
np.random in between."""
import os

import numpy as np
import pytest
import torch

from neuray_amd import _lib, synthetic, pipeline
from neuray_amd.network import renderer as R


needs_lib = pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason='libneuray_hip.so not built (python -m neuray_amd.build)')


@needs_lib
@pytest.mark.parametrize('n', [4096, 4097, 65536, 640000])
@pytest.mark.parametrize('dtype', [np.int32, np.int64])
def test_native_shuffle_is_numpys_on_the_global_generator(n, dtype):
    assert R._host_lib() is not None
    for seed, burn in ((0, 0), (7, 311), (123, 624), (5, 1247)):             # `burn` draws: every position inside the MT19937 block
        np.random.seed(seed)
        np.random.random_sample(burn)
        want = np.arange(n).astype(dtype)
        np.random.shuffle(want)
        after = np.random.randint(0, 1 << 30, 4)
        np.random.seed(seed)
        np.random.random_sample(burn)
        got = np.arange(n).astype(dtype)
        R.shuffle_like_numpy(got)
        assert np.array_equal(got, want), (seed, burn)
        assert np.array_equal(np.random.randint(0, 1 << 30, 4), after)


@needs_lib
def test_native_shuffle_on_a_private_generator_and_with_a_cached_gaussian():
    """a RandomState copy is advanced, the global generator is not touched; the legacy state's cached normal deviate survives"""
    np.random.seed(9)
    np.random.standard_normal(1)                                            # leaves has_gauss = 1 in the state
    start = np.random.get_state()
    assert start[3] == 1
    rs = np.random.RandomState()
    rs.set_state(start)
    a = np.arange(100000, dtype=np.int32)
    R.shuffle_like_numpy(a, rs)
    assert R._same_generator_state(np.random.get_state(), start)
    b = np.arange(100000, dtype=np.int32)
    np.random.shuffle(b)
    assert np.array_equal(a, b) and R._same_generator_state(np.random.get_state(), rs.get_state())
    assert rs.standard_normal(1)[0] == np.random.standard_normal(1)[0]


def test_small_and_odd_arrays_go_through_numpy():
    np.random.seed(1)
    a = np.arange(50)
    R.shuffle_like_numpy(a)
    np.random.seed(1)
    b = np.arange(50)
    np.random.shuffle(b)
    assert np.array_equal(a, b)
    c = np.arange(20000, dtype=np.int16)
    np.random.seed(1)
    R.shuffle_like_numpy(c)
    d = np.arange(20000, dtype=np.int16)
    np.random.seed(1)
    np.random.shuffle(d)
    assert np.array_equal(c, d)


def make_ft(prefetch):
    db = synthetic.MemoryDatabase(7, 96, 128, seed=0)
    scene = {'ref_imgs_info': pipeline.build_imgs_info(db, db.get_img_ids(), -1, True, False, True, True)}
    cfg = {'use_hierarchical_sampling': True, 'depth_sample_num': 8, 'fine_depth_sample_num': 8, 'agg_net_cfg': {'sample_num': 8},
           'fine_agg_net_cfg': {'sample_num': 8}, 'use_self_hit_prob': True, 'use_validation': False, 'train_ray_num': 64,
           'neighbor_view_num': 3, 'neighbor_pool_ratio': 2, 'ray_feats_res': [24, 32], 'hip_prefetch_ray_sampling': prefetch}
    torch.manual_seed(0)
    ft = R.NeuralRayFtRenderer(cfg, scene=scene).train()
    seen = []

    def fake_render(que, ref, is_train):                  # (the draws are what is under test: no kernels)
        seen.append((que['coords'].numpy().copy(), tuple(ft.touched_views)))
        return {}
    ft.render = fake_render
    return ft, seen


def run(prefetch, script):
    """`script`: per step, what an outsider does to np.random before the step"""
    ft, seen = make_ft(prefetch)
    np.random.seed(42)
    for act in script:
        if act == 'draw':
            np.random.random_sample(3)
        elif act == 'reseed':
            np.random.seed(77)
        elif act == 'cfg':
            ft.cfg['train_ray_num'] = 32
        elif act == 'same_state':
            np.random.set_state(np.random.get_state())
        ft.train_step()
    tail = np.random.random_sample(2)
    R._prefetch_take(ft)                                  # (collect the last speculation's thread)
    return seen, tail


@pytest.mark.parametrize('script', [
    [None] * 6,
    [None, None, 'draw', None, 'draw', 'draw'],
    [None, 'reseed', None, None, 'reseed', None],
    [None, None, 'cfg', None, None],
    [None, 'same_state', 'draw', 'reseed', 'cfg', None, None],
])
def test_prefetched_steps_draw_what_unprefetched_steps_draw(script):
    want, tail_want = run(False, script)
    got, tail_got = run(True, script)
    assert len(want) == len(got) == len(script)
    for (cw, tw), (cg, tg) in zip(want, got):
        assert np.array_equal(cw, cg) and tw == tg
    assert np.array_equal(tail_want, tail_got)            # the global generator ends where the reference's order leaves it


def test_prefetch_is_adopted_when_nobody_interferes():
    ft, seen = make_ft(True)
    np.random.seed(3)
    ft.train_step()
    slot = R._PREFETCH.get(ft)
    assert slot is not None
    slot['thread'].join()
    drawn = slot['drawn']
    ft.train_step()
    assert np.array_equal(seen[1][0], drawn[2])           # the second step used the speculative draws
    R._prefetch_take(ft)


def test_renderer_with_a_pending_prefetch_can_be_deep_copied_and_collected():
    import copy
    import gc
    ft, _ = make_ft(True)
    ft.train_step()
    del ft.render                                          # (the test's closure)
    twin = copy.deepcopy(ft)
    assert twin is not ft and R._PREFETCH.get(twin) is None
    del ft, twin
    gc.collect()


from hypothesis import HealthCheck, given, settings, strategies as st     # noqa: E402


@needs_lib
@settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2 ** 32 - 1), burn=st.integers(0, 1400), n=st.integers(4096, 90000), wide=st.booleans(), gauss=st.booleans())
def test_native_shuffle_property(seed, burn, n, wide, gauss):
    """any seed, any position inside the MT19937 block (incl. a refill in the middle of the shuffle), any length, both item sizes, with
    and without a cached normal deviate in the legacy state: the permutation and the generator afterwards are numpy's"""
    def prepare():
        np.random.seed(seed)
        np.random.random_sample(burn)
        if gauss:
            np.random.standard_normal(1)
    dtype = np.int64 if wide else np.int32
    prepare()
    want = np.arange(n).astype(dtype)
    np.random.shuffle(want)
    tail_want = (np.random.standard_normal(2), np.random.randint(0, 1 << 30, 3))
    prepare()
    got = np.arange(n).astype(dtype)
    R.shuffle_like_numpy(got)
    tail_got = (np.random.standard_normal(2), np.random.randint(0, 1 << 30, 3))
    assert np.array_equal(got, want)
    assert np.array_equal(tail_got[0], tail_want[0]) and np.array_equal(tail_got[1], tail_want[1])
