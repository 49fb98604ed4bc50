"""the host input pipeline (neuray_amd/pipeline.py) against the reference's utils/imgs_info.py,
utils/view_select.py and colour mapping (tests/golden/case_pipeline.npz), the device-resident view cache, and the
per-pose render loop."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from emu_util import emu_lib
from test_encoders import fill_by_name
from neuray_amd import pipeline, synthetic
from neuray_amd.network import render_ops as ro

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


@pytest.fixture(scope='module')
def gold():
    return np.load(os.path.join(GOLDEN_DIR, 'case_pipeline.npz'))


def same(info, gold, prefix):
    keys = [k[len(prefix):] for k in gold.files if k.startswith(prefix)]
    assert sorted(info) == sorted(keys)
    for k in keys:
        assert info[k].dtype == gold[prefix + k].dtype and np.array_equal(info[k], gold[prefix + k]), k


def test_build_imgs_info_equals_reference(gold):
    db = synthetic.MemoryDatabase(7, 37, 53, seed=31)
    same(pipeline.build_imgs_info(db, [4, 0, 6], 16, True, False, True, True), gold, 'aligned_')       # reflect-padded to 48 x 64
    same(pipeline.build_imgs_info(db, [4, 0, 6], -1, True, True, False), gold, 'nodepth_')
    ragged = synthetic.MemoryDatabase(3, 30, 41, seed=32, ragged=True)
    same(pipeline.build_imgs_info(ragged, [0, 1, 2], -1, False), gold, 'ragged_')
    r = pipeline.build_render_imgs_info(db.get_pose(2), db.get_K(2), (37, 53), (2.0, 6.0))
    assert r['shape'] == (37, 53)
    for k in ('poses', 'Ks', 'coords', 'depth_range'):
        assert np.array_equal(r[k], gold['render_' + k]), k
    qp = np.stack([db.get_pose(1), db.get_pose(5)])
    assert np.array_equal(pipeline.select_working_views_db(db, None, qp, 3, False), gold['working'])
    assert np.array_equal(pipeline.select_working_views_db(db, [6, 5, 4, 3, 2], qp, 2, True), gold['working_excl'])
    assert np.array_equal(pipeline.color_map_backward(gold['cmap_in']), gold['cmap_back'])
    assert np.array_equal(pipeline.color_map_backward(torch.from_numpy(gold['cmap_in'])).numpy(), gold['cmap_back'])


def test_device_view_cache_equals_host_build_and_uploads_once():
    db = synthetic.MemoryDatabase(6, 37, 53, seed=5)
    cache = pipeline.DeviceViewCache(db, 'cpu', pad_interval=16)
    a = cache.imgs_info([3, 1, 4])
    want = pipeline.build_imgs_info(db, [3, 1, 4], 16, True, False, True, True)
    assert sorted(a) == sorted(want)
    for k in want:
        assert torch.equal(a[k], torch.from_numpy(want[k])), k
    first = cache.uploaded_bytes
    assert first == 3 * (48 * 64 * 3 + 2 * 48 * 64 * 4 + 12 * 4 + 9 * 4 + 2 * 4)       # uint8 image + fp32 mask / depth + camera
    b = cache.imgs_info([1, 4, 5])
    assert cache.uploaded_bytes == first // 3 * 4 and torch.equal(b['imgs'][0], a['imgs'][1])      # only view 5 was new


@pytest.mark.parametrize('backend', BACKENDS)
def test_render_loop_equals_direct_renderer_calls(backend):
    from neuray_amd.network import renderer as R
    dev = 'cpu' if backend == 'emu' else 'cuda:0'
    ro._ENGINES.clear()
    ro._TEST_LIB = emu_lib() if backend == 'emu' else None
    try:
        gen = R.NeuralRayGenRenderer({'use_hierarchical_sampling': True, 'depth_sample_num': 8, 'fine_depth_sample_num': 8,
                                      'agg_net_cfg': {'sample_num': 8}, 'fine_agg_net_cfg': {'sample_num': 8}, 'ray_batch_num': 512,
                                      'init_net_type': 'depth', 'depth_loss_coords_num': 8}).eval()
        fill_by_name(gen)
        if backend == 'emu':
            gen._engine_test_lib = emu_lib()
        gen = gen.to(dev)
        db = synthetic.MemoryDatabase(6, 32, 32, seed=7)
        qposes = np.stack([db.get_pose(0), db.get_pose(3)])
        ref_ids = pipeline.select_working_views_db(db, None, qposes, 3, True)
        cache = pipeline.DeviceViewCache(db, dev, pad_interval=16)
        saved = {}
        imgs = pipeline.render_poses(gen, db, qposes, [db.get_K(0)] * 2, [(32, 32)] * 2, [(2.0, 6.0)] * 2, ref_ids, cache=cache,
                                     save_fn=lambda qi, im: saved.__setitem__(qi, im))
        assert imgs == [] and sorted(saved) == [0, 1] and saved[0].shape == (32, 32, 3) and saved[0].dtype == np.uint8
        # the same image through the reference-style host build + a plain renderer call
        ref = {k: torch.from_numpy(v).to(dev) for k, v in pipeline.build_imgs_info(db, list(ref_ids[1]), 16, True, False, True, True).items()}
        que = pipeline.build_render_imgs_info(qposes[1], db.get_K(0), (32, 32), (2.0, 6.0))
        que.pop('shape')
        with torch.no_grad():
            out = gen({'que_imgs_info': {k: torch.from_numpy(v).to(dev) for k, v in que.items()}, 'ref_imgs_info': ref, 'eval': True})
        want = pipeline.color_map_backward(out['pixel_colors_nr_fine'].reshape(32, 32, 3).cpu().numpy())
        diff = np.abs(saved[1].astype(np.int32) - want.astype(np.int32))
        if backend == 'emu':
            assert diff.max() == 0
        else:       # MIOpen convolutions need not repeat bit for bit, and the coarse -> fine chain amplifies that on a few rays
            assert np.mean(diff <= 1) >= 0.98
        assert len(cache._views) == len(set(ref_ids.reshape(-1).tolist()))
    finally:
        ro._TEST_LIB = None
        ro._ENGINES.clear()
