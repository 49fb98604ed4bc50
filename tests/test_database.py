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


def write_dtu_test(root, scene='toyscan', n=12, seed=7):
    """a DTU-test-shaped scene (dataset/database.py:138-249): image/%06d.png and mask/%03d.png at the native 1600 x 1200,
    cameras.npz with IDR-style world_mat_i = K [R|t] (normalised world) and scale_mat_i, depth_range.npy, 800-wide COLMAP depth"""
    from PIL import Image
    rng = np.random.RandomState(seed)
    base = os.path.join(root, scene)
    for d in ('image', 'mask', 'colmap_depth'):
        os.makedirs(os.path.join(base, d), exist_ok=True)
    cams = {}
    K = np.array([[2892.33, 0.0, 823.2], [0.0, 2883.18, 619.07], [0.0, 0.0, 1.0]])
    for i in range(n):
        w2c = synthetic.look_at_pose(synthetic.sphere_pos(2.6, 30.0 * i, 20.0 + 10.0 * rng.rand())).astype(np.float64)
        # IDR convention: world_mat_i = K [R | t] in the scan's METRIC world (mm), scale_mat_i maps the normalised world (object in the
        # unit sphere) to it; the database undoes scale_mat on the camera centre
        off, mm = np.array([-12.0, 8.0, 630.0]), 310.0
        R, c_n = w2c[:, :3], -w2c[:, :3].T @ w2c[:, 3]
        c_m = mm * c_n + off
        P = np.eye(4)
        P[:3] = K @ np.concatenate([R, -(R @ c_m)[:, None]], 1)
        scale = np.eye(4)
        scale[:3, :3] *= mm
        scale[:3, 3] = off
        cams['world_mat_%d' % i] = P
        cams['scale_mat_%d' % i] = scale
        blk = 40
        img = np.kron(rng.randint(0, 256, size=(1200 // blk, 1600 // blk, 3)), np.ones((blk, blk, 1))).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(base, 'image', '%06d.png' % i))
        m = np.kron((rng.rand(1200 // blk, 1600 // blk) > 0.25).astype(np.uint8) * 255, np.ones((blk, blk), np.uint8))
        Image.fromarray(np.stack([m, m, m], -1)).save(os.path.join(base, 'mask', '%03d.png' % i))
        D.write_colmap_array(os.path.join(base, 'colmap_depth', '%d.jpg.geometric.bin' % i), (400 + 500 * rng.rand(600, 800)).astype(np.float32))
    np.savez(os.path.join(base, 'cameras.npz'), **cams)
    np.save(os.path.join(base, 'depth_range.npy'), np.array([1.1, 4.3], np.float32))
    return scene


@pytest.fixture(scope='module')
def scenes(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('data'))
    # 'nerf': small, for the end-to-end loop;  'nerf800': the released data's size, which is the only one the reference reads
    # without its cv2 resize (dataset/database.py:312-314 hard-codes 800)
    return {'root': root, 'nerf': write_nerf_synthetic(os.path.join(root, 'nerf_synthetic')),
            'nerf800': write_nerf_synthetic(os.path.join(root, 'nerf_synthetic'), scene='toy800', size=800, n_train=3, n_test=1, seed=4),
            'llff': write_llff(os.path.join(root, 'llff_colmap')),
            'dtu': write_dtu_test(os.path.join(root, 'dtu_test'))}


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
    mod.imsave = D.imsave
    # the reference's cv2 calls (resize / GaussianBlur / decomposeProjectionMatrix) run on the numpy restatements (tests/cv2_shim.py)
    import cv2_shim
    bu = importlib.import_module('utils.base_utils')
    mod.cv2 = bu.cv2 = cv2_shim
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


@pytest.mark.parametrize('size', [800, 400, 1600])
def test_dtu_test_adapter_equals_the_reference_class(scenes, ref_db, size, monkeypatch, tmp_path):
    """BASELINE.json config 4 names dtu_test/<scan>/black_800: intrinsics / poses from decomposing world_mat_i, the scale_mat
    normalisation, the world flip, the 1600 -> 800 image / mask resize, masked images and depth maps, splits"""
    name = 'dtu_test/%s/black_%d' % (scenes['dtu'], size)
    os.symlink(scenes['root'], str(tmp_path / 'data'))                    # the reference hard-codes 'data/dtu_test' (database.py:142)
    monkeypatch.chdir(tmp_path)
    ours = D.parse_database_name(name, os.path.join(scenes['root'], 'dtu_test'))
    theirs = ref_db.parse_database_name(name)
    assert list(ours.get_img_ids()) == list(theirs.get_img_ids()) and len(ours.get_img_ids()) == 12
    assert (ours.h, ours.w) == (theirs.h, theirs.w) == (int(size / 1600 * 1200), size)
    for i in ours.get_img_ids():
        assert np.array_equal(ours.get_K(i), theirs.get_K(i)) and ours.get_K(i).dtype == theirs.get_K(i).dtype, i
        assert np.array_equal(ours.get_pose(i), theirs.get_pose(i)) and ours.get_pose(i).dtype == theirs.get_pose(i).dtype, i
        assert same(ours.get_depth_range(i), theirs.get_depth_range(i))
    for i in ours.get_img_ids()[:2]:
        assert same(ours.get_mask(i), theirs.get_mask(i)), i
        assert same(ours.get_image(i), theirs.get_image(i)), i
        assert same(ours.get_depth(i), theirs.get_depth(i)), i
        assert ours.get_image(i).shape == (ours.h, ours.w, 3)
        assert not ours.get_image(i)[~ours.get_mask(i)].any() and not ours.get_depth(i)[~ours.get_mask(i)].any()
    # the adapter's camera is the fixture's normalised-world camera up to the reference's world flip (y, z -> -y, -z): the
    # normalised origin projects to the principal point at the camera's distance
    K, Rt = ours.get_K('0'), ours.get_pose('0')
    x = K @ Rt[:, 3]
    assert abs(x[0] / x[2] - 823.2 * size / 1600) < 1e-2 and abs(x[1] / x[2] - 619.07 * size / 1600) < 1e-2 and abs(x[2] - 2.6) < 1e-4
    assert np.allclose(Rt[:, :3] @ Rt[:, :3].T, np.eye(3), atol=1e-5)
    for split in ('val', 'val_all', 'test', 'test_all'):
        a, b = D.get_database_split(ours, split), ref_db.get_database_split(theirs, split)
        assert list(a[0]) == list(b[0]) and list(a[1]) == list(b[1]) == ['3'], split


def test_nerf_synthetic_black_400_goes_through_the_resize_path(scenes, ref_db):
    """BASELINE.json config 1 names nerf_synthetic/lego/black_400: images blurred + bilinearly halved, masks / depth nearest"""
    name = 'nerf_synthetic/toy800/black_400'
    ours = D.parse_database_name(name, os.path.join(scenes['root'], 'nerf_synthetic'))
    theirs = ref_db.parse_database_name(name)
    for i in ours.get_img_ids()[:2]:
        assert ours.get_image(i).shape == (400, 400, 3) and ours.get_mask(i).shape == (400, 400)
        assert same(ours.get_image(i), theirs.get_image(i)) and same(ours.get_mask(i), theirs.get_mask(i)), i
        assert same(ours.get_K(i), theirs.get_K(i)) and same(ours.get_depth(i), theirs.get_depth(i)), i
    assert np.allclose(ours.get_K('train-r_0')[:2], 0.5 * D.parse_database_name(scenes['nerf800'], os.path.join(scenes['root'], 'nerf_synthetic')).get_K('train-r_0')[:2])


def test_llff_cache_is_built_like_the_reference_builds_it(scenes, ref_db, tmp_path):
    """database.py:84-97: a scene without cache/<res> gets it from images/: Gaussian pre-blur + INTER_AREA"""
    import shutil
    from PIL import Image
    src = os.path.join(scenes['root'], 'llff_colmap', 'toyfern')
    for tag in ('a', 'b'):
        dst = os.path.join(scenes['root'], 'llff_colmap', 'toyfern_' + tag)
        shutil.copytree(src, dst, ignore=shutil.ignore_patterns('cache'))
        os.makedirs(os.path.join(dst, 'images'), exist_ok=True)
        rng = np.random.RandomState(11)
        for i in range(9):            # (quarter-size captures keep the test light: 'low' is then an INTER_AREA factor of 2 instead of 8)
            small = rng.randint(0, 256, size=(756 // 12, 1008 // 12, 3)).astype(np.uint8)
            Image.fromarray(np.kron(small, np.ones((12, 12, 1), np.uint8))).save(os.path.join(dst, 'images', 'image%03d.png' % i))
    ours = D.parse_database_name('llff_colmap/toyfern_a/low', os.path.join(scenes['root'], 'llff_colmap'))
    theirs = ref_db.parse_database_name('llff_colmap/toyfern_b/low')
    for i in ours.get_img_ids()[:3]:
        assert ours.get_image(i).shape == (378, 504, 3)
        assert same(ours.get_image(i), theirs.get_image(i)), i


def test_unsupported_inputs_raise(scenes):
    with pytest.raises(NotImplementedError):
        D.parse_database_name('blended_mvs/building/black_800')
    with pytest.raises(NotImplementedError):
        D.parse_database_name('dtu_test/%s/white_800' % scenes['dtu'], os.path.join(scenes['root'], 'dtu_test'))
    os.rename(os.path.join(scenes['root'], 'llff_colmap', 'toyfern', 'cache', 'low', 'image000.png'), os.path.join(scenes['root'], 'hidden.png'))
    try:
        with pytest.raises(FileNotFoundError, match='images'):             # neither the cached image nor a source to make it from
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


@pytest.mark.parametrize('backend', BACKENDS)
def test_eval_render_loop_on_a_dtu_shaped_scene(scenes, tmp_path, backend):
    """render.py's eval loop on an on-disk DTU-test-shaped scene (BASELINE.json config 4's family) read
    through the dtu_test adapter at black_400 (1600 x 1200 PNGs -> 300 x 400 through the blur + bilinear resize, masks and
    depth maps nearest): held-out view '3', 3 working views, depth init net + encoders + HIP render path, image file out"""
    from test_encoders import fill_by_name
    from neuray_amd.network.renderer import NeuralRayGenRenderer
    db = D.parse_database_name('dtu_test/%s/black_400' % scenes['dtu'], os.path.join(scenes['root'], 'dtu_test'))
    que_poses, que_Ks, que_shapes, que_ranges, ref_ids, render_ids = D.prepare_eval_render(db, use_depth=True)
    assert render_ids == ['3'] and '3' not in ref_ids and len(ref_ids) == 11 and tuple(que_shapes[0]) == (300, 400)
    ref_ids_list = pipeline.select_working_views_db(db, ref_ids, que_poses, 3)
    cfg = {'use_hierarchical_sampling': True, 'depth_sample_num': 8, 'fine_depth_sample_num': 8, 'agg_net_cfg': {'sample_num': 8},
           'fine_agg_net_cfg': {'sample_num': 8}, 'ray_batch_num': 4096 if backend == 'emu' else 32768, 'init_net_type': 'depth'}
    r = NeuralRayGenRenderer(cfg).eval()
    fill_by_name(r)
    if backend == 'emu':
        from neuray_amd.network import render_ops as ro
        r._engine_test_lib = ro._TEST_LIB = emu_lib()
        ro._ENGINES.clear()
        que_shapes = que_shapes // 10                  # (the emulator renders ~5 rays/s: a 30 x 40 crop of the same cameras)
        que_Ks = que_Ks.copy()
        que_Ks[:, :2] /= 10
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
    img = D.imread('%s/0-nr_fine.jpg' % out_dir)
    assert img.shape == tuple(que_shapes[0]) + (3,) and img.dtype == np.uint8 and img.std() > 0
