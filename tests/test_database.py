"""on-disk scene adapters (neuray_amd/database.py) for the two evaluation formats of BASELINE.json's
configs.  A fixture writes a small nerf_synthetic scene (transforms_*.json, RGBA PNGs, COLMAP depth maps) and a small
llff_colmap scene (COLMAP binary model, cached images, depth_range.npy, depth maps) to a temporary directory; the adapters
are compared accessor by accessor with the REFERENCE's own database classes reading the same files (where the reference
tree exists: its cv2 / skimage imports are the stubs of tests/golden/ref_harness.py, `imread` is pointed at PIL), and
render.py's eval loop (prepare_render_info -> select_working_views_db -> build_imgs_info -> renderer -> imsave) runs end to
end on the on-disk scene through the kernels on the emulator."""
import json
import os
import struct
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from emu_util import emu_lib
from neuray_amd import database as D
from neuray_amd import pipeline, synthetic

sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import ref_harness  # noqa: E402


def write_nerf_synthetic(root, scene='toy', size=32, n_train=5, n_test=2, seed=3):
    from PIL import Image
    rng = np.random.RandomState(seed)
    base = os.path.join(root, scene)
    flip = np.diag([1.0, -1.0, -1.0])
    for split, n in (('train', n_train), ('val', 1), ('test', n_test)):
        os.makedirs(os.path.join(base, split), exist_ok=True)
        frames = []
        for i in range(n):
            w2c = synthetic.look_at_pose(synthetic.sphere_pos(4.03, 360.0 * rng.rand(), 15.0 + 30.0 * rng.rand())).astype(np.float64)
            R, t = flip @ w2c[:, :3], flip @ w2c[:, 3:]              # OpenCV -> Blender camera axes
            c2w = np.eye(4)
            c2w[:3, :3], c2w[:3, 3:] = R.T, -R.T @ t
            frames.append({'file_path': './%s/r_%d' % (split, i), 'transform_matrix': c2w.tolist()})
            blk = 1 if size < 100 else 16                             # (large fixtures: blocky images, small PNGs)
            rgba = np.kron(rng.randint(0, 256, size=(size // blk, size // blk, 4)), np.ones((blk, blk, 1))).astype(np.uint8)
            rgba[:, :, 3] = np.kron(np.where(rng.rand(size // blk, size // blk) > 0.3, rng.randint(1, 256, size=(size // blk, size // blk)), 0),
                                    np.ones((blk, blk))).astype(np.uint8)
            Image.fromarray(rgba, 'RGBA').save(os.path.join(base, split, 'r_%d.png' % i))
            if split == 'train' and i != 1:                           # one training view without a depth map
                D.write_colmap_array(os.path.join(base, 'colmap_depth', '%s-r_%d.png.geometric.bin' % (split, i)),
                                     (2.0 + 4.0 * rng.rand(size, size)).astype(np.float32))
        with open(os.path.join(base, 'transforms_%s.json' % split), 'w') as f:
            json.dump({'camera_angle_x': 0.6911112070083618, 'frames': frames}, f)
    return 'nerf_synthetic/%s/black_%d' % (scene, size)



def write_llff(root, scene='toyfern', n=9, seed=5):
    from PIL import Image
    rng = np.random.RandomState(seed)
    base = os.path.join(root, scene)
    h, w = 378, 504
    os.makedirs(os.path.join(base, 'sparse'), exist_ok=True)
    os.makedirs(os.path.join(base, 'cache', 'low'), exist_ok=True)
    with open(os.path.join(base, 'sparse', 'cameras.bin'), 'wb') as f:          # one PINHOLE camera (model 1), full-size 4032 x 3024
        f.write(struct.pack('<Q', 1))
        f.write(struct.pack('<iiQQ', 1, 1, 4032, 3024))
        f.write(np.asarray([3300.5, 3310.25, 2010.0, 1500.5], '<f8').tobytes())
    with open(os.path.join(base, 'sparse', 'images.bin'), 'wb') as f:
        f.write(struct.pack('<Q', n))
        for i in range(n):
            q = rng.randn(4); q /= np.linalg.norm(q)
            f.write(struct.pack('<i', i + 1))
            f.write(np.asarray(q, '<f8').tobytes()); f.write(np.asarray(rng.randn(3), '<f8').tobytes())
            f.write(struct.pack('<i', 1))
            f.write(('image%03d.png' % i).encode() + b'\x00')
            npts = int(rng.randint(0, 4))
            f.write(struct.pack('<Q', npts))
            for _ in range(npts):
                f.write(struct.pack('<ddq', rng.rand(), rng.rand(), -1))
            small = rng.randint(0, 256, size=(h // 6, w // 6, 3)).astype(np.uint8)      # blocky image: compresses well
            Image.fromarray(np.kron(small, np.ones((6, 6, 1), np.uint8))).save(os.path.join(base, 'cache', 'low', 'image%03d.png' % i))
            D.write_colmap_array(os.path.join(base, 'colmap_depth', '%d.jpg.geometric.bin' % (i + 1)), (1.2 + 10 * rng.rand(h, w)).astype(np.float32))
    np.save(os.path.join(base, 'depth_range.npy'), np.stack([1.0 + rng.rand(n), 9.0 + rng.rand(n)], 1).astype(np.float32))
    return 'llff_colmap/%s/low' % scene


@pytest.fixture(scope='module')
def scenes(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('data'))
    # 'nerf': small, for the end-to-end loop;  'nerf800': the released data's size, which is the only one the reference reads
    # without its cv2 resize (dataset/database.py:312-314 hard-codes 800)
    return {'root': root, 'nerf': write_nerf_synthetic(os.path.join(root, 'nerf_synthetic')),
            'nerf800': write_nerf_synthetic(os.path.join(root, 'nerf_synthetic'), scene='toy800', size=800, n_train=3, n_test=1, seed=4),
            'llff': write_llff(os.path.join(root, 'llff_colmap'))}


@pytest.fixture(scope='module')
def ref_db(scenes):
    """the reference's dataset.database with its roots pointed at the fixture and imread at PIL"""
    if not ref_harness.reference_available():
        pytest.skip('reference tree not present')
    ref_harness.import_reference()
    import importlib
    mod = importlib.import_module('dataset.database')
    mod.NERF_SYN_ROOT = os.path.join(scenes['root'], 'nerf_synthetic')
    mod.LLFF_ROOT = os.path.join(scenes['root'], 'llff_colmap')
    mod.imread = D.imread
    return mod


def same(a, b):
    if a is None or b is None:
        return a is None and b is None
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b)


@pytest.mark.parametrize('which', ['nerf800', 'llff'])
def test_adapters_equal_the_reference_classes(scenes, ref_db, which):
    name = scenes[which]
    ours = D.parse_database_name(name, os.path.join(scenes['root'], name.split('/')[0]))
    theirs = ref_db.parse_database_name(name)
    assert list(ours.get_img_ids()) == list(theirs.get_img_ids())
    assert list(ours.get_img_ids(check_depth_exist=True)) == list(theirs.get_img_ids(check_depth_exist=True))
    for i in ours.get_img_ids():
        assert same(ours.get_image(i), theirs.get_image(i)), i
        assert same(ours.get_mask(i), theirs.get_mask(i)), i
        assert same(ours.get_K(i), theirs.get_K(i)), i
        assert np.allclose(ours.get_pose(i), theirs.get_pose(i), rtol=0, atol=0) and ours.get_pose(i).dtype == theirs.get_pose(i).dtype, i
        assert same(ours.get_depth_range(i), theirs.get_depth_range(i)), i
        assert same(ours.get_depth(i), theirs.get_depth(i)), i
    for split in ('val', 'val_all', 'test', 'test_all'):
        a, b = D.get_database_split(ours, split), ref_db.get_database_split(theirs, split)
        assert list(a[0]) == list(b[0]) and list(a[1]) == list(b[1]), split
    if which == 'nerf800':
        assert ours.get_bbox('train-r_0') == theirs.get_bbox('train-r_0')


def test_unsupported_inputs_raise(scenes):
    root = os.path.join(scenes['root'], 'nerf_synthetic')
    with pytest.raises(NotImplementedError, match='resize'):
        D.parse_database_name('nerf_synthetic/toy/black_16', root)          # would need the reference's cv2 resize
    with pytest.raises(NotImplementedError):
        D.parse_database_name('dtu_test/snowman/black_800')
    os.rename(os.path.join(scenes['root'], 'llff_colmap', 'toyfern', 'cache', 'low', 'image000.png'), os.path.join(scenes['root'], 'hidden.png'))
    try:
        with pytest.raises(NotImplementedError, match='cache'):
            D.parse_database_name(scenes['llff'], os.path.join(scenes['root'], 'llff_colmap'))
    finally:
        os.rename(os.path.join(scenes['root'], 'hidden.png'), os.path.join(scenes['root'], 'llff_colmap', 'toyfern', 'cache', 'low', 'image000.png'))


def test_colmap_array_roundtrip(tmp_path):
    a = np.random.RandomState(0).rand(7, 5).astype(np.float32)
    D.write_colmap_array(str(tmp_path / 'a.bin'), a)
    assert np.array_equal(D.read_colmap_array(str(tmp_path / 'a.bin')), a)


BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


@pytest.mark.parametrize('backend', BACKENDS)
def test_eval_render_loop_on_the_on_disk_scene(scenes, tmp_path, backend):
    """render.py:19-27,124-141,49-56 on files: held-out cameras of the scene, working views by camera distance, depth init
    net + encoders + HIP render path, uint8 write-back, image files out"""
    from test_encoders import fill_by_name
    from neuray_amd.network.renderer import NeuralRayGenRenderer
    db = D.parse_database_name(scenes['nerf'], os.path.join(scenes['root'], 'nerf_synthetic'))
    que_poses, que_Ks, que_shapes, que_ranges, ref_ids, render_ids = D.prepare_eval_render(db, use_depth=True)
    assert len(render_ids) == 2 and all(i.startswith('train') for i in ref_ids) and 'train-r_1' not in ref_ids
    ref_ids_list = pipeline.select_working_views_db(db, ref_ids, que_poses, 3)
    cfg = {'use_hierarchical_sampling': True, 'depth_sample_num': 8, 'fine_depth_sample_num': 8, 'agg_net_cfg': {'sample_num': 8},
           'fine_agg_net_cfg': {'sample_num': 8}, 'ray_batch_num': 1024, 'init_net_type': 'depth'}
    r = NeuralRayGenRenderer(cfg).eval()
    fill_by_name(r)
    if backend == 'emu':
        from neuray_amd.network import render_ops as ro
        r._engine_test_lib = ro._TEST_LIB = emu_lib()
        ro._ENGINES.clear()
    else:
        r = r.cuda()
    out_dir = str(tmp_path / 'render')
    try:
        pipeline.render_poses(r, db, que_poses, que_Ks, que_shapes, que_ranges, ref_ids_list, pad_interval=16,
                              save_fn=lambda qi, img: D.imsave('%s/%d-nr_fine.jpg' % (out_dir, qi), img))
    finally:
        if backend == 'emu':
            ro._TEST_LIB = None
            ro._ENGINES.clear()
    for qi in range(2):
        img = D.imread('%s/%d-nr_fine.jpg' % (out_dir, qi))
        assert img.shape == (32, 32, 3) and img.dtype == np.uint8 and img.std() > 0
