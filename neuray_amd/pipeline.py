"""Host input pipeline and output write-back around the render path: the reference's
`build_imgs_info` / `build_render_imgs_info` (utils/imgs_info.py:60-131), view selection (utils/view_select.py:18-27,80-87)
and the per-pose loop of render.py:124-141,49-56 - minus file formats: a `database` is any object with the reference's
accessor methods (`get_image` uint8 HxWx3, `get_mask`, `get_depth`, `get_pose`, `get_K`, `get_depth_range`,
`get_img_ids`; dataset/database.py:14-50).

What is different is where the bytes live.  The reference rebuilds `ref_imgs_info` on the host for every query pose and
uploads it (8 x 800 x 800 fp32 images = 61 MB over PCIe per rendered image, although neighbouring poses share most of
their reference views).  `DeviceViewCache` uploads every view ONCE, as uint8 (4x fewer PCIe bytes; the /255 of
`color_map_forward` runs on the GPU), keeps it resident in HBM (a 100-view 800 x 800 scene is 0.8 GB of 288), and builds a
pose's `ref_imgs_info` by stacking device tensors; rendered images are quantised to uint8 on the GPU
(`color_map_backward`, utils/base_utils.py:496-499) before the copy back (1.9 MB instead of 7.7 MB)."""
import numpy as np
import torch

from .network.renderer import nearest_view_table, pad_views


def color_map_forward(rgb):
    """utils/base_utils.py:492-493"""
    return rgb.astype(np.float32) / 255


def color_map_backward(rgb):
    """utils/base_utils.py:496-499 (numpy array or device tensor: truncating cast after the clip, as .astype(np.uint8))"""
    if torch.is_tensor(rgb):
        return torch.clamp(rgb * 255, 0, 255).to(torch.uint8)
    return np.clip(rgb * 255, a_min=0, a_max=255).astype(np.uint8)


def _pad_end(img, th, tw, mode):
    h, w = img.shape[:2]
    if (th, tw) == (h, w):
        return img
    pads = ((0, th - h), (0, tw - w)) + ((0, 0),) * (img.ndim - 2)
    return np.pad(img, pads, mode, constant_values=0) if mode == 'constant' else np.pad(img, pads, mode)


def build_imgs_info(database, ref_ids, pad_interval=-1, is_aligned=True, align_depth_range=False, has_depth=True,
                    replace_none_depth=False):
    """utils/imgs_info.py:77-122 -> dict of numpy arrays: imgs [n,3,h,w] in [0,1], masks [n,1,h,w], depth [n,1,h,w],
    poses [n,3,4], Ks [n,3,3], depth_range [n,2]"""
    imgs = [database.get_image(i) for i in ref_ids]
    masks = [database.get_mask(i) for i in ref_ids]
    depths = [database.get_depth(i) for i in ref_ids] if has_depth else None
    if not is_aligned:      # images of different sizes: pad at the end to the largest (image reflect, mask / depth zero)
        assert has_depth
        th, tw = np.max(np.asarray([im.shape[:2] for im in imgs]), 0)
        imgs = [_pad_end(im, th, tw, 'reflect') for im in imgs]
        masks = [_pad_end(m, th, tw, 'constant') for m in masks]
        depths = [_pad_end(d, th, tw, 'constant') for d in depths]
    elif has_depth and replace_none_depth:
        h, w = imgs[0].shape[:2]
        depths = [np.zeros([h, w], np.float32) if d is None else d for d in depths]
    info = {'imgs': color_map_forward(np.stack(imgs, 0)).transpose([0, 3, 1, 2]),
            'poses': np.asarray([database.get_pose(i) for i in ref_ids], np.float32),
            'Ks': np.asarray([database.get_K(i) for i in ref_ids], np.float32),
            'depth_range': np.asarray([database.get_depth_range(i) for i in ref_ids], np.float32),
            # (quirk kept: only the aligned path casts masks / depth to float32, imgs_info.py:96-98,101-108)
            'masks': np.asarray(masks, np.float32)[:, None] if is_aligned else np.stack(masks, 0)[:, None]}
    if align_depth_range:
        info['depth_range'][:, 0], info['depth_range'][:, 1] = info['depth_range'][:, 0].min(), info['depth_range'][:, 1].max()
    if has_depth:
        info['depth'] = np.asarray(depths, np.float32)[:, None] if is_aligned else np.stack(depths, 0)[:, None]
    if pad_interval != -1:
        info = pad_views(info, pad_interval)
    return info


def build_render_imgs_info(que_pose, que_K, que_shape, que_depth_range):
    """utils/imgs_info.py:124-131: the query side of one rendered image (all pixels, x fastest)"""
    h, w = int(que_shape[0]), int(que_shape[1])
    coords = np.stack(np.meshgrid(np.arange(w), np.arange(h)), -1).reshape(1, -1, 2).astype(np.float32)
    return {'poses': que_pose.astype(np.float32)[None], 'Ks': que_K.astype(np.float32)[None], 'coords': coords,
            'depth_range': np.asarray(que_depth_range, np.float32)[None], 'shape': (h, w)}


def select_working_views_db(database, ref_ids, que_poses, work_num, exclude_self=False):
    """utils/view_select.py:80-87 -> [qn, work_num] view ids nearest to each query camera"""
    ref_ids = np.asarray(database.get_img_ids() if ref_ids is None else ref_ids)
    order = nearest_view_table(np.asarray(que_poses), np.asarray([database.get_pose(i) for i in ref_ids]))
    return ref_ids[order[:, 1:work_num + 1] if exclude_self else order[:, :work_num]]


class DeviceViewCache:
    """Per-view device-resident copies of a database's views (see the module docstring)."""

    def __init__(self, database, device, pad_interval=-1, has_depth=True):
        self.database, self.device = database, torch.device(device)
        self.pad_interval, self.has_depth = pad_interval, has_depth
        self._views = {}
        self.uploaded_bytes = 0

    def _load(self, view_id):
        img = self.database.get_image(view_id)                             # uint8 [h,w,3]
        mask = np.asarray(self.database.get_mask(view_id), np.float32)
        depth = self.database.get_depth(view_id) if self.has_depth else None
        if self.has_depth and depth is None:
            depth = np.zeros(img.shape[:2], np.float32)
        if self.pad_interval != -1:
            h, w = img.shape[:2]
            ph, pw = (-h) % self.pad_interval, (-w) % self.pad_interval
            if ph or pw:
                img = np.pad(img, ((0, ph), (0, pw), (0, 0)), 'reflect')
                mask = np.pad(mask, ((0, ph), (0, pw)), 'reflect')
                depth = np.pad(depth, ((0, ph), (0, pw)), 'reflect') if depth is not None else None
        host = {'imgs': np.ascontiguousarray(img.transpose(2, 0, 1)), 'masks': mask[None], 'poses': np.asarray(self.database.get_pose(view_id), np.float32),
                'Ks': np.asarray(self.database.get_K(view_id), np.float32),
                'depth_range': np.asarray(self.database.get_depth_range(view_id), np.float32)}
        if depth is not None:
            host['depth'] = np.asarray(depth, np.float32)[None]
        dev = {}
        for k, v in host.items():
            t = torch.from_numpy(v).to(self.device, non_blocking=True)
            self.uploaded_bytes += v.nbytes
            dev[k] = t.float() / 255 if k == 'imgs' else t            # color_map_forward on the device
        return dev

    def view(self, view_id):
        if view_id not in self._views:
            self._views[view_id] = self._load(view_id)
        return self._views[view_id]

    def imgs_info(self, ref_ids):
        """build_imgs_info(database, ref_ids, pad_interval) as device tensors, without touching the host for cached views"""
        views = [self.view(i) for i in ref_ids]
        return {k: torch.stack([v[k] for v in views], 0) for k in views[0]}


def render_poses(renderer, database, que_poses, que_Ks, que_shapes, que_depth_ranges, ref_ids_list, pad_interval=16,
                 cache=None, key=None, save_fn=None):
    """render.py:124-141 for a generalisation renderer: one image per query pose from its working views.  Returns the list
    of uint8 [h,w,3] images (numpy), or hands each one to `save_fn(qi, image)` instead (file formats are the caller's
    business).  `cache`: a DeviceViewCache to keep across calls."""
    dev = next(renderer.parameters()).device
    cache = cache or DeviceViewCache(database, dev, pad_interval)
    images = []
    for qi in range(len(que_poses)):
        que = build_render_imgs_info(que_poses[qi], que_Ks[qi], que_shapes[qi], que_depth_ranges[qi])
        h, w = que.pop('shape')
        que = {k: torch.from_numpy(v).to(dev) for k, v in que.items()}
        with torch.no_grad():
            out = renderer({'que_imgs_info': que, 'ref_imgs_info': cache.imgs_info(ref_ids_list[qi]), 'eval': True})
        k = key or ('pixel_colors_nr_fine' if 'pixel_colors_nr_fine' in out else 'pixel_colors_nr')
        img = color_map_backward(out[k].reshape(h, w, 3)).cpu().numpy()
        if save_fn is not None:
            save_fn(qi, img)
        else:
            images.append(img)
    return images
