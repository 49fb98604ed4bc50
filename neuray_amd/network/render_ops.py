"""Mirror of network/render_ops.py: the reference's 12 free functions with their signatures, running on the HIP
kernels (stand-alone entry points of include/neuray_hip.h that share their device code with the fused render
path).  Tensors must live on the HIP device; there is no eager fallback.

Reference lines: coords2rays :4, depth2points :27, depth2dists :41, depth2inv_dists :46, interpolate_feature_map :54,
alpha_values2hit_prob :72, project_points_coords :82, project_points_directions :106, project_points_ref_views :117,
project_points_dict :132, sample_depth :146, sample_fine_depth :172.
"""
import torch

from ..engine import RenderEngine

__all__ = ['coords2rays', 'depth2points', 'depth2dists', 'depth2inv_dists', 'interpolate_feature_map', 'alpha_values2hit_prob',
           'project_points_coords', 'project_points_directions', 'project_points_ref_views', 'project_points_dict',
           'sample_depth', 'sample_fine_depth']

_ENGINES = {}
_TEST_LIB = None      # CPU test-suite hook: emulator build of the kernels


def engine_for(device):
    device = torch.device(device)
    key = (device.type, device.index)
    if key not in _ENGINES:
        _ENGINES[key] = RenderEngine(device, _test_lib=_TEST_LIB)
    return _ENGINES[key]


def check_deferred_inputs(device, wait=True):
    """Raise what the device-side input checks of the stand-alone ops found (engine.check_deferred): the neighbour-index range and the
    singular-projection check of warp_variance, which replaced host-side checks that cost a device -> host wait per call.  Call it where
    the program waits for the device anyway - after `loss.item()` of a training step, at the end of an epoch, after an inference call."""
    key = (torch.device(device).type, torch.device(device).index)
    if key in _ENGINES:
        _ENGINES[key].check_deferred(wait=wait)


def _query_const(eng, pose, K, depth_range=None):
    info = {'poses': pose[None], 'Ks': K[None],
            'depth_range': depth_range[None] if depth_range is not None else torch.ones(1, 2, device=pose.device)}
    return eng.prepare_query(info)


def coords2rays(coords, poses, Ks):
    eng = engine_for(coords.device)
    cs, ds = [], []
    for v in range(coords.shape[0]):
        c, d = eng.rays_points(_query_const(eng, poses[v], Ks[v]), coords[v])
        cs.append(c); ds.append(d)
    return torch.stack(cs, 0), torch.stack(ds, 0)


def depth2points(que_imgs_info, que_depth):
    eng = engine_for(que_depth.device)
    pts, dirs = [], []
    for q in range(que_depth.shape[0]):
        qc = _query_const(eng, que_imgs_info['poses'][q], que_imgs_info['Ks'][q])
        _, _, p, d = eng.rays_points(qc, que_imgs_info['coords'][q], que_depth[q])
        pts.append(p); dirs.append(d)
    return torch.stack(pts, 0), torch.stack(dirs, 0)


def depth2dists(depth):
    return engine_for(depth.device).depth_dists(depth, None)


def depth2inv_dists(depth, depth_range):
    eng = engine_for(depth.device)
    return torch.stack([eng.depth_dists(depth[q], depth_range[q]) for q in range(depth.shape[0])], 0)


class _InterpFn(torch.autograd.Function):
    """interpolate_feats with the gradient w.r.t. the feature map (the coordinates are constants on every call site of the
    render / depth-loss path, renderer.py:127-155,282-300)."""

    @staticmethod
    def forward(ctx, feats, coords, mask, h, w, align):
        eng = engine_for(feats.device)
        ctx.save_for_backward(coords, mask)
        ctx.meta = (eng, tuple(feats.shape), h, w, align)
        return eng.interpolate_feats(feats, coords, h, w, align_corners=align, mask=mask)

    @staticmethod
    def backward(ctx, d_out):
        coords, mask = ctx.saved_tensors
        eng, shape, h, w, align = ctx.meta
        return eng.interpolate_feats_backward(d_out.contiguous(), shape, coords, h, w, align_corners=align, mask=mask), \
            None, None, None, None, None


def interpolate_feature_map(ray_feats, coords, mask, h, w, border_type='border'):
    if border_type != 'border':
        raise NotImplementedError("neuray_amd: only padding_mode='border' is on the render path")
    fh, fw = ray_feats.shape[-2:]
    align = (fh == h and fw == w)
    if torch.is_grad_enabled() and ray_feats.requires_grad:
        return _InterpFn.apply(ray_feats, coords, mask.float(), h, w, align)
    return engine_for(ray_feats.device).interpolate_feats(ray_feats, coords, h, w, align_corners=align, mask=mask.float())


def alpha_values2hit_prob(alpha_values):
    return engine_for(alpha_values.device).alpha2hit_prob(alpha_values)


def _project(poses, Ks, pts, h, w):
    eng = engine_for(pts.device)
    rfn = poses.shape[0]
    dr = torch.ones(rfn, 2, device=pts.device)
    return eng.project_points(eng.setup_views(poses, Ks, dr), pts, rfn, h, w)


def project_points_coords(pts, Rt, K):
    _, pts2d, depth, mask = _project(Rt, K, pts, 0, 0)      # h = w = 0: no image-bounds test
    return pts2d, mask, depth.unsqueeze(-1)


def project_points_directions(poses, points):
    eye = torch.eye(3, device=points.device)[None].expand(poses.shape[0], -1, -1).contiguous()
    return _project(poses, eye, points, 0, 0)[0]


def project_points_ref_views(ref_imgs_info, que_points):
    h, w = ref_imgs_info['imgs'].shape[-2:]
    prj_dir, prj_pts, prj_depth, mask = _project(ref_imgs_info['poses'], ref_imgs_info['Ks'], que_points, h, w)
    return prj_dir, prj_pts, prj_depth.unsqueeze(-1), mask


def project_points_dict(ref_imgs_info, que_pts):
    qn, rn, dn, _ = que_pts.shape
    prj_dir, prj_pts, prj_depth, prj_mask = project_points_ref_views(ref_imgs_info, que_pts.reshape(qn * rn * dn, 3))
    rfn, _, h, w = ref_imgs_info['imgs'].shape
    d = {'dir': prj_dir, 'pts': prj_pts, 'depth': prj_depth, 'mask': prj_mask.float(),
         'ray_feats': interpolate_feature_map(ref_imgs_info['ray_feats'], prj_pts, prj_mask, h, w),
         'rgb': interpolate_feature_map(ref_imgs_info['imgs'], prj_pts, prj_mask, h, w)}
    return {k: v.reshape(rfn, qn, rn, dn, -1) for k, v in d.items()}


def sample_depth(depth_range, coords, sample_num, random_sample):
    eng = engine_for(coords.device)
    qn, rn, _ = coords.shape
    # random_sample: the jitter uniforms are drawn exactly as render_ops.py:161 draws them (same shape, dtype and device,
    # hence the same generator stream)
    u = torch.rand(qn, rn, sample_num - 2, dtype=torch.float32, device=coords.device) if random_sample else None
    depth = torch.stack([eng.sample_coarse_depth(depth_range[q], rn, sample_num, None if u is None else u[q]) for q in range(qn)], 0)
    dists = torch.cat([depth[..., 1:], torch.full_like(depth[..., :1], 1e6)], -1) - depth
    return depth, dists


def sample_fine_depth(depth, hit_prob, depth_range, sample_num, random_sample, inv_mode=True):
    eng = engine_for(depth.device)
    qn, rn, dn = depth.shape
    u = torch.rand([qn, rn, sample_num]) if random_sample else None      # CPU generator, as render_ops.py:205
    outs = []
    for q in range(qn):
        qc = _query_const(eng, torch.eye(3, 4, device=depth.device), torch.eye(3, device=depth.device), depth_range[0])
        outs.append(eng.sample_fine_depth(qc, depth[q].contiguous(), hit_prob[q].contiguous(), sample_num,
                                          u=None if u is None else u[q], sort=False, inv_mode=inv_mode))
    return torch.stack(outs, 0)
