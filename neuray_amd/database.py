"""On-disk scene adapters for the host pipeline: the three evaluation formats BASELINE.json's
configs read - `nerf_synthetic/<scene>/<background>_<size>` (configs 1, 2), `llff_colmap/<scene>/<high|low>` (config 3) and
`dtu_test/<scene>/black_<size>` (config 4) - exposing the accessors of the reference's `BaseDatabase` (dataset/database.py:25-58) so that `neuray_amd.pipeline`
(`build_imgs_info`, `DeviceViewCache`, `render_poses`) and the reference's own `render.py` loop can run on them, plus the
reference's `parse_database_name` (database.py:983-1003) / `get_database_split` (:1005-1046) for these two families and a
PNG / JPEG writer for the rendered images (render.py:49-56).  Decoding is PIL + numpy (the reference uses skimage / cv2,
which this image does not have); COLMAP's binary models and depth maps are read with a few lines of `struct`.

The reference's on-the-fly RESIZING paths (`black_400`: `resize_img` = cv2 Gaussian blur + bilinear resize, database.py:312-314;
nearest-neighbour masks / depth maps, :341,349; the DTU 1600 -> 800 / 400 images and masks, :203,244; the LLFF `cache/<res>`
images it writes on first use, :84-97) go through neuray_amd/imgproc.py, numpy restatements of the OpenCV routines - see its
header for what is exact (nearest, area, the projection-matrix decomposition) and what may differ from OpenCV's fixed-point
8-bit paths by one grey level (blur, bilinear); nothing there could be pinned against cv2 in this container.
"""
import json
import os
import struct

import numpy as np

from . import imgproc

NERF_SYN_ROOT = 'data/nerf_synthetic'            # asset.py / database.py:258
LLFF_ROOT = 'data/llff_colmap'                   # asset.py:50
DTU_TEST_ROOT = 'data/dtu_test'                  # database.py:142
nerf_syn_val_ids = ['val-r_39', 'val-r_2', 'val-r_94', 'val-r_62', 'val-r_23', 'val-r_36']      # asset.py:45


def imread(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


def imsave(path, img):
    """uint8 [h,w,3] (or [h,w]) -> PNG / JPEG by extension (render.py:49-56 uses skimage.io.imsave)"""
    from PIL import Image
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(np.asarray(img)).save(path)


def color_map_backward(rgb):
    """utils/base_utils.py:496-499"""
    return np.clip(rgb * 255, a_min=0, a_max=255).astype(np.uint8)


def read_colmap_array(path):
    """colmap/read_write_dense.py:40-53: '<width>&<height>&<channels>&' then float32, column-major"""
    with open(path, 'rb') as f:
        head = b''
        while head.count(b'&') < 3:
            ch = f.read(1)
            if not ch:
                raise ValueError('%s: truncated COLMAP array header' % path)
            head += ch
        w, h, c = (int(x) for x in head.split(b'&')[:3])
        data = np.fromfile(f, np.float32)
    return np.transpose(data.reshape((w, h, c), order='F'), (1, 0, 2)).squeeze()


def write_colmap_array(path, arr):
    """inverse of read_colmap_array (test fixtures)"""
    arr = np.asarray(arr, np.float32)
    a3 = arr[:, :, None] if arr.ndim == 2 else arr
    h, w, c = a3.shape
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, 'wb') as f:
        f.write(('%d&%d&%d&' % (w, h, c)).encode())
        np.transpose(a3, (1, 0, 2)).reshape(-1, order='F').astype(np.float32).tofile(f)


# ---- COLMAP sparse model (colmap/read_write_model.py: read_cameras_binary, read_images_binary) ------------------------
_CAMERA_PARAMS = {0: 3, 1: 4, 2: 4, 3: 5, 4: 8, 5: 8, 6: 12, 7: 5, 8: 4, 9: 5, 10: 12}     # model id -> number of parameters


def read_cameras_binary(path):
    """-> {camera_id: dict(model_id, width, height, params)}"""
    cams = {}
    with open(path, 'rb') as f:
        (n,) = struct.unpack('<Q', f.read(8))
        for _ in range(n):
            cid, model, w, h = struct.unpack('<iiQQ', f.read(24))
            params = np.frombuffer(f.read(8 * _CAMERA_PARAMS[model]), '<f8').copy()
            cams[cid] = {'model_id': model, 'width': int(w), 'height': int(h), 'params': params}
    return cams


def read_images_binary(path):
    """-> {image_id: dict(qvec, tvec, camera_id, name)} (the 2-D observations are skipped)"""
    imgs = {}
    with open(path, 'rb') as f:
        (n,) = struct.unpack('<Q', f.read(8))
        for _ in range(n):
            iid, = struct.unpack('<i', f.read(4))
            qvec = np.frombuffer(f.read(32), '<f8').copy()
            tvec = np.frombuffer(f.read(24), '<f8').copy()
            cid, = struct.unpack('<i', f.read(4))
            name = b''
            while True:
                ch = f.read(1)
                if ch in (b'\x00', b''):
                    break
                name += ch
            (npts,) = struct.unpack('<Q', f.read(8))
            f.seek(24 * npts, 1)
            imgs[iid] = {'qvec': qvec, 'tvec': tvec, 'camera_id': cid, 'name': name.decode()}
    return imgs


def qvec2rotmat(q):
    """colmap/read_write_model.py qvec2rotmat (w, x, y, z)"""
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


class BaseDatabase:
    def __init__(self, database_name):
        self.database_name = database_name

    def get_bbox(self, img_id):
        raise NotImplementedError


class NeRFSyntheticDatabase(BaseDatabase):
    """dataset/database.py:251-353.  database_name = 'nerf_synthetic/<scene>/<black|white>_<size>'; the PNGs are the released
    800 x 800 ones, any other <size> (black_400: BASELINE.json config 1) is produced by the reference's resize path."""

    def __init__(self, database_name, root=None):
        super().__init__(database_name)
        _, model_name, background_size = database_name.split('/')
        background, size = background_size.split('_')
        if background not in ('black', 'white'):
            raise NotImplementedError(background)
        self.model_name, self.img_size, self.background = model_name, int(size), background
        self.root_dir = os.path.join(root if root is not None else NERF_SYN_ROOT, model_name)
        ids, poses = [], []
        for split in ('train', 'val', 'test'):                 # database.py:260-266: ids in this order
            i, p, K = self.parse_info(split)
            ids += i
            poses += p
        self.img_ids, self.poses = ids, poses
        self.range_dict = {i: np.asarray((2.0, 6.0), np.float32) for i in ids}
        # database.py:269-270,312-314,341,349 resize relative to the constant 800, the size of the released PNGs; here relative
        # to the PNGs' own size (the same thing for the released data)
        self.native = imread('%s/%s.png' % (self.root_dir, self.img_id2img_path(ids[0]))).shape[0]
        self.ratio = self.img_size / self.native
        self.K = np.diag([self.ratio, self.ratio, 1.0]).astype(np.float32) @ K
        self.depth_img_ids = [i for i in ids if os.path.exists(self._depth_fn(i))]

    def parse_info(self, split='train'):
        """database.py:273-291: Blender camera-to-world -> OpenCV world-to-camera [R|t]"""
        with open('%s/transforms_%s.json' % (self.root_dir, split)) as f:
            info = json.load(f)
        focal = float(info['camera_angle_x'])
        img_ids, poses = [], []
        flip = np.diag(np.asarray([1, -1, -1]))
        for frame in info['frames']:
            img_ids.append('-'.join(frame['file_path'].split('/')[1:]))
            pose = np.asarray(frame['transform_matrix'], np.float32)
            R = pose[:3, :3].T
            t = -R @ pose[:3, 3:]
            poses.append(np.concatenate([flip @ R, flip @ t], 1))
        h, w, _ = imread('%s/%s.png' % (self.root_dir, self.img_id2img_path(img_ids[0]))).shape
        focal = .5 * w / np.tan(.5 * focal)
        return img_ids, poses, np.asarray([[focal, 0, w / 2], [0, focal, h / 2], [0, 0, 1]], np.float32)

    @staticmethod
    def img_id2img_path(img_id):
        return '/'.join(img_id.split('-'))

    def _depth_fn(self, img_id):
        return '%s/colmap_depth/%s.png.geometric.bin' % (self.root_dir, img_id)

    def get_image(self, img_id):
        img = imread('%s/%s.png' % (self.root_dir, self.img_id2img_path(img_id)))
        alpha = img[:, :, 3:].astype(np.float32) / 255.0
        rgb = img[:, :, :3].astype(np.float32) / 255.0
        rgb = rgb * alpha if self.background == 'black' else rgb * alpha + 1.0 - alpha
        img = color_map_backward(rgb)
        return imgproc.resize_img(img, self.ratio) if self.img_size != self.native else img

    def get_K(self, img_id):
        return self.K.astype(np.float32).copy()

    def get_pose(self, img_id):
        return self.poses[self.img_ids.index(img_id)].astype(np.float32).copy()

    def get_img_ids(self, check_depth_exist=False):
        return self.depth_img_ids if check_depth_exist else self.img_ids

    def get_bbox(self, img_id):
        alpha = imread('%s/%s.png' % (self.root_dir, self.img_id2img_path(img_id)))[:, :, 3]
        ys, xs = np.nonzero(alpha > 0)
        return [xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1]

    def get_depth(self, img_id):
        fn = self._depth_fn(img_id)
        if not os.path.exists(fn):
            return None
        depth = read_colmap_array(fn)
        return imgproc.resize(depth, (self.img_size, self.img_size), imgproc.INTER_NEAREST) if self.img_size != self.native else depth

    def get_mask(self, img_id):
        alpha = imread('%s/%s.png' % (self.root_dir, self.img_id2img_path(img_id)))[:, :, 3]
        if self.img_size != self.native:
            alpha = imgproc.resize(alpha, (self.img_size, self.img_size), imgproc.INTER_NEAREST)
        return alpha > 0

    def get_depth_range(self, img_id):
        return self.range_dict[img_id].copy()


class LLFFColmapDatabase(BaseDatabase):
    """dataset/database.py:61-136.  database_name = 'llff_colmap/<scene>/<high|low>'; reads <root>/<scene>/sparse/
    {cameras,images}.bin, cache/<res>/<image name>, depth_range.npy and colmap_depth/<id>.jpg.geometric.bin."""

    def __init__(self, database_name, root=None):
        super().__init__(database_name)
        _, self.model_name, self.res_type = database_name.split('/')
        if self.res_type not in ('high', 'low'):
            raise NotImplementedError(self.res_type)
        self.root_dir = os.path.join(root if root is not None else LLFF_ROOT, self.model_name)
        self.cameras_colmap = read_cameras_binary('%s/sparse/cameras.bin' % self.root_dir)
        self.images_colmap = read_images_binary('%s/sparse/images.bin' % self.root_dir)
        self.img_ids = [str(k + 1) for k in range(len(self.images_colmap))]
        self.image_dir = '%s/cache/%s' % (self.root_dir, self.res_type)
        self._cache_resolution()
        self.bounds = np.load('%s/depth_range.npy' % self.root_dir)

    def _cache_resolution(self):
        """database.py:84-97: images missing from cache/<res> are made from images/<name>: Gaussian pre-blur for the ratio
        w / 4032, then an INTER_AREA resize (an integer factor for the 4032 x 3024 captures: 4 or 8)"""
        os.makedirs(self.image_dir, exist_ok=True)
        h, w = self.get_resolution()
        for img_id in self.img_ids:
            fn = self.images_colmap[int(img_id)]['name']
            if os.path.exists('%s/%s' % (self.image_dir, fn)):
                continue
            src = '%s/images/%s' % (self.root_dir, fn)
            if not os.path.exists(src):
                raise FileNotFoundError("neuray_amd.database: neither %s/%s nor %s exists" % (self.image_dir, fn, src))
            img = imgproc.downsample_gaussian_blur(imread(src), w / 4032)
            imsave('%s/%s' % (self.image_dir, fn), imgproc.resize(img, (w, h), imgproc.INTER_AREA))

    def get_resolution(self):
        return (756, 1008) if self.res_type == 'high' else (756 // 2, 1008 // 2)

    def get_img_ids(self, check_depth_exist=False):
        return self.img_ids

    def get_image(self, img_id):
        return imread('%s/%s' % (self.image_dir, self.images_colmap[int(img_id)]['name']))

    def get_K(self, img_id):
        cam = self.cameras_colmap[self.images_colmap[int(img_id)]['camera_id']]
        h, w = self.get_resolution()
        fx, fy, cx, cy = cam['params'][:4]
        K = np.asarray([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
        return (np.diag([w / cam['width'], h / cam['height'], 1]) @ K).astype(np.float32)

    def get_pose(self, img_id):
        info = self.images_colmap[int(img_id)]
        return np.concatenate([qvec2rotmat(info['qvec']), info['tvec'][:, None]], 1)

    def get_depth(self, img_id):
        return read_colmap_array('%s/colmap_depth/%s.jpg.geometric.bin' % (self.root_dir, img_id))

    def get_mask(self, img_id):
        h, w = self.get_resolution()
        return np.ones([h, w], dtype=np.bool_)

    def get_depth_range(self, img_id):
        return self.bounds[int(img_id) - 1]


class DTUTestDatabase(BaseDatabase):
    """dataset/database.py:138-249.  database_name = 'dtu_test/<scene>/black_<size>' (size 1600 = native, 800 = BASELINE.json
    config 4, 400).  Reads <root>/<scene>/{image/%06d.png, mask/%03d.png, cameras.npz (IDR-style world_mat_i / scale_mat_i),
    depth_range.npy, colmap_depth/<i>.jpg.geometric.bin}.  Intrinsics and world-to-camera poses come from decomposing
    world_mat_i; the normalisation of scale_mat_i is undone on the camera centre and the world is flipped (y, z -> -y, -z)."""

    def __init__(self, database_name, root=None):
        super().__init__(database_name)
        _, self.model_name, background_size = database_name.split('/')
        self.background, size = background_size.split('_')
        if self.background != 'black':
            raise NotImplementedError(self.background)
        self.image_size = int(size)
        self.root_dir = os.path.join(root if root is not None else DTU_TEST_ROOT, self.model_name)
        self.ratio = self.image_size / 1600
        self.h, self.w = int(self.ratio * 1200), self.image_size
        n = len(sorted(f for f in os.listdir(os.path.join(self.root_dir, 'image')) if f.endswith(('.jpg', '.png'))))
        self.depth_range = np.load('%s/depth_range.npy' % self.root_dir)
        cams = np.load(os.path.join(self.root_dir, 'cameras.npz'))
        flip_world = np.diag([1.0, -1.0, -1.0, 1.0]).astype(np.float32)
        self.Ks, self.Rts, self.img_ids = [], [], []
        for i in range(n):
            K, R, ch = imgproc.decompose_projection_matrix(cams['world_mat_%d' % i][:3])
            cam2world = np.eye(4, dtype=np.float32)                  # (float32 from here on, as the reference's np.eye(4, float32))
            cam2world[:3, :3] = R.T
            cam2world[:3, 3] = (ch[:3] / ch[3])[:, 0]
            if 'scale_mat_%d' % i in cams.files:
                sm = cams['scale_mat_%d' % i]
                cam2world[:3, 3:] -= sm[:3, 3:]
                cam2world[:3, 3:] /= np.diagonal(sm[:3, :3])[..., None]
            cam2world = (flip_world @ cam2world)[:3]
            self.Rts.append(np.concatenate([cam2world[:, :3].T, -cam2world[:, :3].T @ cam2world[:, 3:]], 1))
            self.Ks.append(np.diag([self.ratio, self.ratio, 1]) @ K)
            self.img_ids.append(str(i))
        self._imgs, self._depths, self._masks = {}, {}, {}
        self.depth_img_ids = [i for i in self.img_ids if os.path.exists('%s/depth_maps/%s.jpg.geometric.bin' % (self.root_dir, i))]

    def get_image(self, img_id):
        if img_id not in self._imgs:
            img = imread(os.path.join(self.root_dir, 'image', '%06d.png' % int(img_id)))
            if self.w != 1600:
                img = imgproc.resize(imgproc.downsample_gaussian_blur(img, self.ratio), (self.w, self.h), imgproc.INTER_LINEAR)
            self._imgs[img_id] = img * self.get_mask(img_id).astype(np.uint8)[:, :, None]
        return self._imgs[img_id]

    def get_K(self, img_id):
        return self.Ks[int(img_id)].copy()

    def get_pose(self, img_id):
        return self.Rts[int(img_id)].copy()

    def get_img_ids(self, check_depth_exist=False):
        return self.img_ids                           # (database.py:216-217: the flag is ignored for this family)

    def get_depth(self, img_id):
        if img_id not in self._depths:
            fn = '%s/colmap_depth/%s.jpg.geometric.bin' % (self.root_dir, img_id)
            if not os.path.exists(fn):
                raise NotImplementedError("neuray_amd.database: no depth map %s" % fn)
            depth = np.ascontiguousarray(read_colmap_array(fn), dtype=np.float32)
            if self.w != 800:                         # (database.py:236: the COLMAP maps of this family are 800 wide)
                depth = imgproc.resize(depth, (self.w, self.h), imgproc.INTER_NEAREST)
            depth = depth.copy()
            depth[~self.get_mask(img_id)] = 0
            self._depths[img_id] = depth
        return self._depths[img_id]

    def get_mask(self, img_id):
        if img_id not in self._masks:
            mask = np.sum(imread(os.path.join(self.root_dir, 'mask', '%03d.png' % int(img_id))), -1) > 0
            if self.w != 1600:
                mask = imgproc.resize(mask.astype(np.uint8), (self.w, self.h), imgproc.INTER_NEAREST) > 0
            self._masks[img_id] = mask
        return self._masks[img_id]

    def get_depth_range(self, img_id):
        return self.depth_range.copy()


name2database = {'nerf_synthetic': NeRFSyntheticDatabase, 'llff_colmap': LLFFColmapDatabase, 'dtu_test': DTUTestDatabase}


def parse_database_name(database_name, root=None):
    """dataset/database.py:983-1003 (the two evaluation families built here)"""
    kind = database_name.split('/')[0]
    if kind not in name2database:
        raise NotImplementedError("neuray_amd.database: no adapter for %r (have %s)" % (kind, sorted(name2database)))
    return name2database[kind](database_name, root)


def get_database_split(database, split_type='val'):
    """dataset/database.py:1005-1046 -> (reference / training view ids, validation or test view ids)"""
    name = database.database_name
    parts = split_type.split('_')
    depth_valid = not (len(parts) > 1 and parts[1] == 'all')
    if not (split_type.startswith('val') or split_type.startswith('test')):
        raise NotImplementedError(split_type)
    if name.startswith('nerf_synthetic'):
        train_ids = [i for i in database.get_img_ids(check_depth_exist=depth_valid) if i.startswith('tr')]
        val_ids = list(nerf_syn_val_ids) if split_type.startswith('val') else [i for i in database.get_img_ids() if i.startswith('te')]
    elif name.startswith('llff'):
        val_ids = database.get_img_ids()[::8]
        train_ids = [i for i in database.get_img_ids(check_depth_exist=depth_valid) if i not in val_ids]
    elif name.startswith('dtu_test'):
        val_ids = database.get_img_ids()[3:-3:8]
        train_ids = [i for i in database.get_img_ids(check_depth_exist=depth_valid) if i not in val_ids]
    else:
        raise NotImplementedError(name)
    return train_ids, val_ids


def prepare_eval_render(database, use_depth=True):
    """render.py:19-27 (`pose_type == 'eval'`): the held-out views' cameras as render targets
    -> que_poses [n,3,4], que_Ks [n,3,3], que_shapes [n,2], que_depth_ranges [n,2], ref_ids, render_ids"""
    ref_ids, render_ids = get_database_split(database, 'test' if use_depth else 'test_all')
    que_Ks = np.asarray([database.get_K(i) for i in render_ids], np.float32)
    que_poses = np.asarray([database.get_pose(i) for i in render_ids], np.float32)
    que_shapes = np.asarray([database.get_image(i).shape[:2] for i in render_ids], np.int64)
    que_depth_ranges = np.asarray([database.get_depth_range(i) for i in render_ids], np.float32)
    return que_poses, que_Ks, que_shapes, que_depth_ranges, ref_ids, render_ids
