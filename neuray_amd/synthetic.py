"""Seeded synthetic 'lego-800'-style scenes and the PSNR metric (network/metrics.py:14-27).
Pure numpy input generation shared by bench.py, smoke() and the tests; no rendering arithmetic here."""
import numpy as np


def look_at_pose(cam_pos, target=(0, 0, 0), up=(0, 0, 1)):
    """OpenCV world->camera [R|t]: camera at cam_pos looking at target (z forward, y down)."""
    cam_pos = np.asarray(cam_pos, np.float64)
    z = np.asarray(target, np.float64) - cam_pos
    z /= np.linalg.norm(z)
    x = np.cross(z, np.asarray(up, np.float64))
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], 0)
    t = -R @ cam_pos
    return np.concatenate([R, t[:, None]], 1).astype(np.float32)


def sphere_pos(radius, azim_deg, elev_deg):
    a, e = np.deg2rad(azim_deg), np.deg2rad(elev_deg)
    return np.array([radius * np.cos(e) * np.cos(a), radius * np.cos(e) * np.sin(a), radius * np.sin(e)])


def smooth_field(rng, n, c, h, w, waves=6, max_cycles=5.0):
    """[n,c,h,w] band-limited random fields (sums of `waves` plane waves of at most `max_cycles` periods across the map),
    zero mean and about unit variance per channel: the smooth counterpart of the white-noise maps - neighbouring texels
    and neighbouring views' values are correlated, as encoder outputs of real images are."""
    ys = np.linspace(0.0, 1.0, h)[None, None, :, None]
    xs = np.linspace(0.0, 1.0, w)[None, None, None, :]
    out = np.zeros((n, c, h, w), np.float64)
    for _ in range(waves):
        fx = rng.uniform(-max_cycles, max_cycles, size=(n, c, 1, 1))
        fy = rng.uniform(-max_cycles, max_cycles, size=(n, c, 1, 1))
        ph = rng.uniform(0.0, 2.0 * np.pi, size=(n, c, 1, 1))
        amp = rng.uniform(0.5, 1.0, size=(n, c, 1, 1))
        out += amp * np.sin(2.0 * np.pi * (fx * xs + fy * ys) + ph)
    return out * np.sqrt(2.0 / (waves * 0.583))          # E[amp^2] = 0.583, E[sin^2] = 1/2


def make_scene(h=800, w=800, rfn=8, seed=0, depth_range=(2.0, 6.0), radius=4.03, feat_dim=32,
               que_imgs=False, fov_x=0.6911112070083618, smooth=False):
    """Seeded synthetic 'lego-like' scene: cameras on a sphere looking at the origin,
    random images and feature maps (the per-image encoders are bypassed).  `smooth`: band-limited images and maps
    (smooth_field) instead of white noise."""
    rng = np.random.RandomState(seed)
    f = 0.5 * w / np.tan(0.5 * fov_x)
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)
    offs = [(-5, 5), (5, -5), (-10, -8), (10, 8), (-15, 12), (15, -12), (-20, -3), (20, 3),
            (-25, 15), (25, -15), (-30, 6), (30, -6), (0, 20), (0, -20), (12, 18), (-12, -18)]
    assert rfn <= len(offs)
    que_pose = look_at_pose(sphere_pos(radius, 30.0, 25.0))
    ref_poses = np.stack([look_at_pose(sphere_pos(radius, 30.0 + a, 25.0 + e)) for a, e in offs[:rfn]])
    fh, fw = h // 4, w // 4
    if smooth:
        img = lambda n: np.clip(0.5 + 0.22 * smooth_field(rng, n, 3, h, w), 0.0, 1.0).astype(np.float32)      # noqa: E731
        feat = lambda n: smooth_field(rng, n, feat_dim, fh, fw).astype(np.float32)                            # noqa: E731
        ref = {'imgs': img(rfn), 'poses': ref_poses.astype(np.float32), 'Ks': np.repeat(K[None], rfn, 0),
               'depth_range': np.repeat(np.asarray(depth_range, np.float32)[None], rfn, 0), 'ray_feats': feat(rfn), 'img_feats': feat(rfn)}
        que = {'poses': que_pose[None], 'Ks': K[None].copy(), 'depth_range': np.asarray(depth_range, np.float32)[None]}
        if que_imgs:
            que['imgs'], que['ray_feats'] = img(1), feat(1)
        return que, ref
    ref = {
        'imgs': rng.rand(rfn, 3, h, w).astype(np.float32),
        'poses': ref_poses.astype(np.float32),
        'Ks': np.repeat(K[None], rfn, 0),
        'depth_range': np.repeat(np.asarray(depth_range, np.float32)[None], rfn, 0),
        'ray_feats': rng.randn(rfn, feat_dim, fh, fw).astype(np.float32),
        'img_feats': rng.randn(rfn, feat_dim, fh, fw).astype(np.float32),
    }
    que = {
        'poses': que_pose[None], 'Ks': K[None].copy(),
        'depth_range': np.asarray(depth_range, np.float32)[None],
    }
    if que_imgs:
        que['imgs'] = rng.rand(1, 3, h, w).astype(np.float32)
        que['ray_feats'] = rng.randn(1, feat_dim, fh, fw).astype(np.float32)
    return que, ref


def meshgrid_coords(h, w):
    """x-fastest pixel grid [1,h*w,2] as utils/imgs_info.py:122-131"""
    xs, ys = np.meshgrid(np.arange(w), np.arange(h))
    return np.stack([xs, ys], -1).reshape(1, -1, 2).astype(np.float32)


def psnr_uint8(a, b):
    """PSNR on uint8-quantised images, as network/metrics.py:14-27"""
    qa = np.clip(np.round(a * 255), 0, 255)
    qb = np.clip(np.round(b * 255), 0, 255)
    mse = np.mean((qa - qb) ** 2)
    return float('inf') if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse))


class MemoryDatabase:
    """An in-memory scene with the accessor methods of the reference's BaseDatabase (dataset/database.py:14-50):
    seeded random uint8 images, masks, depth maps, cameras on a sphere.  `ragged`: views of slightly different sizes."""

    def __init__(self, n, h, w, seed=0, ragged=False, radius=4.03, depth_range=(2.0, 6.0)):
        rng = np.random.RandomState(seed)
        self.ids = list(range(n))
        self.sizes = [(h - (i % 3) * 2, w - (i % 2) * 3) if ragged else (h, w) for i in range(n)]
        self.images = [rng.randint(0, 256, size=(hh, ww, 3)).astype(np.uint8) for hh, ww in self.sizes]
        self.masks = [rng.rand(hh, ww) > 0.4 for hh, ww in self.sizes]
        self.depths = [(depth_range[0] + (depth_range[1] - depth_range[0]) * rng.rand(hh, ww)).astype(np.float32) for hh, ww in self.sizes]
        self.poses = [look_at_pose(sphere_pos(radius, 360.0 * i / n + 5.0 * rng.rand(), 20.0 + 15.0 * rng.rand())).astype(np.float32) for i in range(n)]
        f = 0.5 * w / np.tan(0.5 * 0.6911112070083618)
        self.K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)
        self.depth_range = np.asarray(depth_range, np.float32)

    def get_img_ids(self):
        return list(self.ids)

    def get_image(self, i):
        return self.images[i]

    def get_mask(self, i):
        return self.masks[i]

    def get_depth(self, i):
        return self.depths[i]

    def get_pose(self, i):
        return self.poses[i]

    def get_K(self, i):
        return self.K.copy()

    def get_depth_range(self, i):
        return self.depth_range.copy()
