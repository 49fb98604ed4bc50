"""CPU oracle: numpy fp32 restatement of the NeuRay per-ray render path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it; nothing under neuray_amd/ does.

Every function cites the reference file:line it restates (paths relative to the
reference tree, liuyuan-pal/NeuRay).  The restatement is written from the maths in
SURVEY.md Appendix A and is *pinned* against the reference itself: the golden
vectors under tests/golden/ were produced by running the reference's own modules
(tests/golden/make_golden.py, build container only) and tests/test_oracle_golden.py
checks this file against them.

Conventions: everything float32; weights are a flat dict {state_dict key: ndarray}
using the reference's state_dict names, accessed through a
prefix ('dist_decoder.', 'agg_net.', 'fine_dist_decoder.', 'fine_agg_net.').

Small fixed-size contractions (3x3, 3x4, 4x4 camera algebra) are written as
explicit left-to-right sums of separate multiplies and adds so that the HIP
kernels can follow the same rounding sequence (see DESIGN.md "Rounding contract").
"""
import numpy as np

F32 = np.float32


def f32(x):
    return np.asarray(x, dtype=np.float32)


# --------------------------------------------------------------------------------------
# small ordered linear algebra (rounding contract shared with the HIP kernels)
# --------------------------------------------------------------------------------------
def dot3(a0, a1, a2, b0, b1, b2):
    """(a0*b0 + a1*b1) + a2*b2 with one rounding per operation (no FMA)."""
    return (a0 * b0 + a1 * b1) + a2 * b2


def inv3x3(K):
    """3x3 inverse by the adjugate.  The reference uses torch.inverse
    (network/render_ops.py:20); callers that need bit-parity with it pass a
    precomputed inverse instead (the host mirror does exactly that)."""
    K = K.astype(np.float64)
    return np.linalg.inv(K).astype(np.float32)


# --------------------------------------------------------------------------------------
# a1  sample_depth                                   network/render_ops.py:146-170
# --------------------------------------------------------------------------------------
def sample_depth(depth_range, rn, dn):
    """depth_range [qn,2] -> que_depth [qn,rn,dn] (random_sample=False, the only mode
    the renderer uses, network/renderer.py:219)."""
    depth_range = f32(depth_range)
    assert dn > 2
    near, far = depth_range[:, 0], depth_range[:, 1]
    one = F32(1.0)
    diff = one / far - one / near                          # qn
    interval = diff / F32(dn - 1)
    val = np.arange(1, dn - 1, dtype=np.float32)[None, :]  # 1,dn-2
    ticks_mid = interval[:, None] * val                    # qn,dn-2
    ticks = np.concatenate([np.zeros_like(diff)[:, None], ticks_mid, diff[:, None]], 1)  # qn,dn
    que_depth = one / (one / near[:, None] + ticks)        # qn,dn
    qn = depth_range.shape[0]
    return np.broadcast_to(que_depth[:, None, :], (qn, rn, dn)).astype(np.float32).copy()


# --------------------------------------------------------------------------------------
# a2  coords2rays / depth2points                     network/render_ops.py:4-39
# --------------------------------------------------------------------------------------
def camera_center(pose):
    """-R^T t, network/render_ops.py:15-16 (trans = -rot @ t with rot = R^T)."""
    R, t = pose[:, :3], pose[:, 3]
    # rot = R^T ; (-rot) @ t : row i of rot is column i of R
    c = np.stack([dot3(-R[0, i], -R[1, i], -R[2, i], t[0], t[1], t[2]) for i in range(3)])
    return c.astype(np.float32)


def coords2rays(coords, poses, Ks_inv):
    """coords [qn,rn,2], poses [qn,3,4], Ks_inv [qn,3,3] -> centers, directions [qn,rn,3].
    directions are un-normalised: (R^T (K^-1 [x,y,1]) + c) - c, network/render_ops.py:21-23."""
    coords, poses, Ks_inv = f32(coords), f32(poses), f32(Ks_inv)
    qn, rn, _ = coords.shape
    centers = np.zeros([qn, rn, 3], np.float32)
    dirs = np.zeros([qn, rn, 3], np.float32)
    one = np.ones([rn], np.float32)
    for q in range(qn):
        c = camera_center(poses[q])
        x, y = coords[q, :, 0], coords[q, :, 1]
        Ki = Ks_inv[q]
        cam = [dot3(Ki[i, 0], Ki[i, 1], Ki[i, 2], x, y, one) for i in range(3)]
        R = poses[q, :, :3]
        for i in range(3):  # rot = R^T -> row i of rot = column i of R
            w = dot3(R[0, i], R[1, i], R[2, i], cam[0], cam[1], cam[2])
            dirs[q, :, i] = (w + c[i]) - c[i]
            centers[q, :, i] = c[i]
    return centers, dirs


def depth2points(coords, poses, Ks_inv, que_depth):
    """-> que_pts [qn,rn,dn,3], que_dir [qn,rn,dn,3]   network/render_ops.py:27-39"""
    centers, dirs = coords2rays(coords, poses, Ks_inv)
    que_pts = centers[:, :, None, :] + dirs[:, :, None, :] * f32(que_depth)[..., None]
    nrm = np.sqrt((dirs[..., 0] * dirs[..., 0] + dirs[..., 1] * dirs[..., 1]) + dirs[..., 2] * dirs[..., 2])
    que_dir = -dirs / nrm[..., None]
    dn = que_depth.shape[-1]
    que_dir = np.broadcast_to(que_dir[:, :, None, :], que_pts.shape).copy()
    return que_pts.astype(np.float32), que_dir.astype(np.float32)


# --------------------------------------------------------------------------------------
# a3  depth2dists / depth2inv_dists                  network/render_ops.py:41-52
# --------------------------------------------------------------------------------------
def depth2dists(depth):
    d = depth[..., 1:] - depth[..., :-1]
    return np.concatenate([d, np.full(depth.shape[:-1] + (1,), 1e6, np.float32)], -1)


def depth2inv_dists(depth, depth_range):
    depth, depth_range = f32(depth), f32(depth_range)
    near = (F32(-1.0) / depth_range[:, 0])[:, None, None]
    far = (F32(-1.0) / depth_range[:, 1])[:, None, None]
    depth_inv = F32(-1.0) / depth
    depth_inv = (depth_inv - near) / (far - near)
    return depth2dists(depth_inv)


# --------------------------------------------------------------------------------------
# a4-a6  projection                                  network/render_ops.py:82-130
# --------------------------------------------------------------------------------------
def compute_H(poses, Ks):
    """KRt = K @ Rt, network/render_ops.py:94 (rows of [K Rt; 0 0 0 1])."""
    poses, Ks = f32(poses), f32(Ks)
    rfn = poses.shape[0]
    H = np.zeros([rfn, 3, 4], np.float32)
    for i in range(3):
        for j in range(4):
            H[:, i, j] = dot3(Ks[:, i, 0], Ks[:, i, 1], Ks[:, i, 2], poses[:, 0, j], poses[:, 1, j], poses[:, 2, j])
    return H


def project_points_coords(pts, H):
    """pts [pn,3], H [rfn,3,4] -> pts2d [rfn,pn,2], valid [rfn,pn] bool, depth [rfn,pn,1]
    network/render_ops.py:82-104.  No z>0 test (quirk A.9.1)."""
    pts = f32(pts)
    x, y, z = pts[None, :, 0], pts[None, :, 1], pts[None, :, 2]
    cam = [((H[:, i, 0, None] * x + H[:, i, 1, None] * y) + H[:, i, 2, None] * z) + H[:, i, 3, None] for i in range(3)]
    depth = cam[2].copy()
    invalid = np.abs(depth) < F32(1e-4)
    depth[invalid] = F32(1e-3)
    pts2d = np.stack([cam[0] / depth, cam[1] / depth], -1)
    return pts2d.astype(np.float32), ~invalid, depth[..., None].astype(np.float32)


def project_points_directions(poses, points):
    """network/render_ops.py:106-115"""
    poses, points = f32(poses), f32(points)
    rfn = poses.shape[0]
    out = np.zeros([rfn, points.shape[0], 3], np.float32)
    for v in range(rfn):
        c = camera_center(poses[v])
        d = points - c[None, :]
        nrm = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
        out[v] = -d / np.maximum(nrm, F32(1e-5))[:, None]
    return out


def project_points_ref_views(ref_poses, ref_Ks, h, w, que_points):
    """network/render_ops.py:117-130 -> prj_dir, prj_pts, prj_depth, valid_mask"""
    H = compute_H(ref_poses, ref_Ks)
    prj_pts, valid, prj_depth = project_points_coords(que_points, H)
    px, py = prj_pts[..., 0], prj_pts[..., 1]
    invalid_img = (px < F32(-0.5)) | (px >= F32(w - 0.5)) | (py < F32(-0.5)) | (py >= F32(h - 0.5))
    valid_mask = valid & (~invalid_img)
    prj_dir = project_points_directions(ref_poses, que_points)
    return prj_dir, prj_pts, prj_depth, valid_mask


# --------------------------------------------------------------------------------------
# a7  bilinear gather                      network/ops.py:14-34, render_ops.py:54-70
# --------------------------------------------------------------------------------------
def texel_coords(p, size_full, size_map, align_corners):
    """pixel coordinate p (full-res units) -> clamped texel coordinate in the map.
    Mirrors interpolate_feats' normalisation (ops.py:28-29) followed by
    F.grid_sample's unnormalise + border clip."""
    p = f32(p)
    n = p / F32(size_full - 1) * F32(2.0) - F32(1.0)
    if align_corners:
        ix = ((n + F32(1.0)) / F32(2.0)) * F32(size_map - 1)
    else:
        ix = ((n + F32(1.0)) * F32(size_map) - F32(1.0)) / F32(2.0)
    return np.minimum(F32(size_map - 1), np.maximum(ix, F32(0.0)))


def interpolate_feats(feats, points, h=None, w=None, align_corners=False):
    """feats [b,f,fh,fw] NCHW, points [b,n,2] (x,y) -> [b,n,f]; bilinear, border padding.
    network/ops.py:14-34."""
    feats, points = f32(feats), f32(points)
    b, f, fh, fw = feats.shape
    if h is None and w is None:
        h, w = fh, fw
    ix = texel_coords(points[..., 0], w, fw, align_corners)
    iy = texel_coords(points[..., 1], h, fh, align_corners)
    x0f, y0f = np.floor(ix), np.floor(iy)
    x0, y0 = x0f.astype(np.int64), y0f.astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    wx1, wy1 = ix - x0f, iy - y0f                   # (ix - ix_nw), (iy - iy_nw)
    wx0, wy0 = (x0f + F32(1.0)) - ix, (y0f + F32(1.0)) - iy
    out = np.zeros([b, points.shape[1], f], np.float32)
    bi = np.arange(b)[:, None]
    for (xx, yy, ww) in ((x0, y0, wx0 * wy0), (x1, y0, wx1 * wy0), (x0, y1, wx0 * wy1), (x1, y1, wx1 * wy1)):
        inb = (xx >= 0) & (xx < fw) & (yy >= 0) & (yy < fh)
        xc, yc = np.clip(xx, 0, fw - 1), np.clip(yy, 0, fh - 1)
        vals = feats[bi, :, yc, xc]                 # b,n,f
        out += vals * (ww * inb.astype(np.float32))[..., None]
    return out


def interpolate_feature_map(feats, coords, mask, h, w):
    """network/render_ops.py:54-70: align_corners=True iff the map is full resolution."""
    fh, fw = feats.shape[-2:]
    out = interpolate_feats(feats, coords, h, w, align_corners=(fh == h and fw == w))
    return out * mask.astype(np.float32)[..., None]


def project_points_dict(ref, que_pts):
    """network/render_ops.py:132-144.  ref: dict with poses, Ks, imgs, ray_feats."""
    qn, rn, dn, _ = que_pts.shape
    rfn, _, h, w = ref['imgs'].shape
    prj_dir, prj_pts, prj_depth, prj_mask = project_points_ref_views(
        ref['poses'], ref['Ks'], h, w, que_pts.reshape(qn * rn * dn, 3))
    d = {
        'dir': prj_dir, 'pts': prj_pts, 'depth': prj_depth, 'mask': prj_mask.astype(np.float32)[..., None],
        'ray_feats': interpolate_feature_map(ref['ray_feats'], prj_pts, prj_mask, h, w),
        'rgb': interpolate_feature_map(ref['imgs'], prj_pts, prj_mask, h, w),
    }
    return {k: v.reshape(rfn, qn, rn, dn, -1) for k, v in d.items()}


# --------------------------------------------------------------------------------------
# activations (PyTorch CPU semantics)
# --------------------------------------------------------------------------------------
def elu(x):
    return np.where(x > 0, x, np.exp(np.minimum(x, F32(0))) - F32(1.0)).astype(np.float32)


def softplus(x):
    return np.where(x > F32(20.0), x, np.log1p(np.exp(np.minimum(x, F32(20.0))))).astype(np.float32)


def sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(np.float32)


def relu(x):
    return np.maximum(x, F32(0.0))


def linear(x, W, b=None):
    y = x @ W.T
    if b is not None:
        y = y + b
    return y.astype(np.float32)


# --------------------------------------------------------------------------------------
# a9/a10  MixtureLogisticsDistDecoder               network/dist_decoder.py
# --------------------------------------------------------------------------------------
def has_vis_decoder(weights, prefix):
    return (prefix + 'vis_decoder.0.weight') in weights


def _head(weights, prefix, name, x, final):
    p = prefix + name + '.'
    hdn = elu(linear(x, weights[p + '0.weight'], weights[p + '0.bias']))
    hdn = elu(linear(hdn, weights[p + '2.weight'], weights[p + '2.bias']))
    return final(linear(hdn, weights[p + '4.weight'], weights[p + '4.bias']))


def dist_decoder_forward(weights, prefix, feats, bias_val=0.05):
    """network/dist_decoder.py:99-107 -> mean [...,2], var [...,2], vis [...,1] or None, aw [...,1]"""
    mean = _head(weights, prefix, 'mean_decoder', feats, softplus)
    var = _head(weights, prefix, 'var_decoder', feats, softplus) + F32(bias_val)
    aw = _head(weights, prefix, 'aw_decoder', feats, sigmoid)
    vis = _head(weights, prefix, 'vis_decoder', feats, sigmoid) if has_vis_decoder(weights, prefix) else None
    return mean, var, vis, aw


def get_near_far_points(depth, interval, depth_range, is_ref):
    """network/dist_decoder.py:6-51 (fixed_interval=False)"""
    depth, interval, depth_range = f32(depth), f32(interval), f32(depth_range)
    sh = (-1,) + (1,) * (depth.ndim - 1)
    near_r = (F32(-1.0) / depth_range[:, 0]).reshape(sh)
    far_r = (F32(-1.0) / depth_range[:, 1]).reshape(sh)
    depth = np.maximum(depth, F32(1e-5))
    depth = F32(-1.0) / depth
    depth = (depth - near_r) / (far_r - near_r)
    half = interval / F32(2.0)
    if is_ref:
        ext = np.concatenate([half[..., 0:1], half], -1)
        near = depth - ext[..., :-1]
        far = depth + ext[..., 1:]
    else:
        first = depth[..., 0] - half[..., 0]
        last = depth[..., -1] + half[..., -1]
        mid = (depth[..., :-1] + depth[..., 1:]) / F32(2.0)
        ext = np.concatenate([first[..., None], mid, last[..., None]], -1)
        near, far = ext[..., :-1], ext[..., 1:]
    return near.astype(np.float32), far.astype(np.float32)


def compute_prob(depth, interval, mean, var, vis, aw, is_ref, depth_range, use_vis):
    """network/dist_decoder.py:109-140 -> alpha_value, visibility, hit_prob"""
    near, far = get_near_far_points(depth, interval, depth_range, is_ref)
    mix = np.concatenate([aw, F32(1.0) - aw], -1)
    near, far = near[..., None], far[..., None]
    d0 = (near - mean) * var
    d1 = (far - mean) * var
    cdf0 = F32(0.5) + F32(0.5) * np.tanh(d0)
    cdf1 = F32(0.5) + F32(0.5) * np.tanh(d1)
    if use_vis:
        cdf0, cdf1 = cdf0 * vis, cdf1 * vis
    visibility = F32(1.0) - cdf0
    hit = cdf1 - cdf0
    visibility = np.sum(visibility * mix, -1, dtype=np.float32)
    hit = np.sum(hit * mix, -1, dtype=np.float32)
    eps = F32(1e-5)
    alpha = np.log(hit / (visibility - hit + eps) + eps)
    return alpha.astype(np.float32), visibility.astype(np.float32), hit.astype(np.float32)


# --------------------------------------------------------------------------------------
# a13/a14  aggregation net          network/aggregate_net.py, network/ibrnet.py:239-369
# --------------------------------------------------------------------------------------
def posenc(d_hid, n_samples):
    """network/ibrnet.py:305-313"""
    def angle(position):
        return [position / np.power(10000, 2 * (j // 2) / d_hid) for j in range(d_hid)]
    t = np.array([angle(p) for p in range(n_samples)])
    t[:, 0::2] = np.sin(t[:, 0::2])
    t[:, 1::2] = np.cos(t[:, 1::2])
    return t.astype(np.float32)[None]


def seq(weights, prefix, x, acts):
    """nn.Sequential of Linear layers at even indices with the given activations."""
    for i, act in enumerate(acts):
        p = '%s%d.' % (prefix, 2 * i)
        x = linear(x, weights[p + 'weight'], weights.get(p + 'bias'))
        if act is not None:
            x = act(x)
    return x


def fused_mean_variance(x, weight):
    """network/ibrnet.py:112-116 (reduction over the view axis = 2)"""
    mean = np.sum(x * weight, 2, keepdims=True, dtype=np.float32)
    var = np.sum(weight * (x - mean) ** 2, 2, keepdims=True, dtype=np.float32)
    return mean, var


def layer_norm(x, g, b, eps=1e-6):
    mu = np.mean(x, -1, keepdims=True, dtype=np.float32)
    var = np.mean((x - mu) ** 2, -1, keepdims=True, dtype=np.float32)
    return ((x - mu) / np.sqrt(var + F32(eps)) * g + b).astype(np.float32)


def softmax(x, axis):
    m = np.max(x, axis, keepdims=True)
    e = np.exp(x - m)
    return (e / np.sum(e, axis, keepdims=True, dtype=np.float32)).astype(np.float32)


def ray_attention(weights, prefix, x, mask):
    """MultiHeadAttention(4,16,4,4), network/ibrnet.py:52-102 with the query-row mask
    of ScaledDotProductAttention (:7-27).  x [b,dn,16], mask [b,dn,1]"""
    b, dn, _ = x.shape
    nh, dk = 4, 4
    q = linear(x, weights[prefix + 'w_qs.weight']).reshape(b, dn, nh, dk).transpose(0, 2, 1, 3)
    k = linear(x, weights[prefix + 'w_ks.weight']).reshape(b, dn, nh, dk).transpose(0, 2, 1, 3)
    v = linear(x, weights[prefix + 'w_vs.weight']).reshape(b, dn, nh, dk).transpose(0, 2, 1, 3)
    attn = (q / F32(dk ** 0.5)) @ k.transpose(0, 1, 3, 2)            # b,nh,dn,dn
    attn = np.where(mask[:, None, :, :] == 0, F32(-1e9), attn)       # mask broadcasts over keys
    attn = softmax(attn, -1)
    o = (attn @ v).transpose(0, 2, 1, 3).reshape(b, dn, nh * dk)
    o = linear(o, weights[prefix + 'fc.weight']) + x
    return layer_norm(o, weights[prefix + 'layer_norm.weight'], weights[prefix + 'layer_norm.bias'])


def ibrnet_forward(weights, prefix, rgb_feat, neuray_feat, ray_diff, mask, return_aux=False):
    """IBRNetWithNeuRay.forward, network/ibrnet.py:315-369.
    rgb_feat [rn,dn,rfn,35], neuray_feat [rn,dn,rfn,32], ray_diff [rn,dn,rfn,4], mask [rn,dn,rfn,1]"""
    p = prefix
    rfn = rgb_feat.shape[2]
    dn = rgb_feat.shape[1]
    direction_feat = seq(weights, p + 'ray_dir_fc.', ray_diff, [elu, elu])
    rgb_in = rgb_feat[..., :3]
    rgb_feat = rgb_feat + direction_feat
    weight = mask / (np.sum(mask, 2, keepdims=True, dtype=np.float32) + F32(1e-8))
    weight0 = sigmoid(seq(weights, p + 'neuray_fc.', neuray_feat, [elu, None])) * weight
    mean0, var0 = fused_mean_variance(rgb_feat, weight0)
    mean1, var1 = fused_mean_variance(rgb_feat, weight)
    globalfeat = np.concatenate([mean0, var0, mean1, var1], -1)
    x = np.concatenate([np.broadcast_to(globalfeat, globalfeat.shape[:2] + (rfn, globalfeat.shape[-1])),
                        rgb_feat, neuray_feat], -1)
    x = seq(weights, p + 'base_fc.', x, [elu, elu])
    x_vis = seq(weights, p + 'vis_fc.', x * weight, [elu, elu])
    x_res, vis = x_vis[..., :-1], x_vis[..., -1:]
    vis = sigmoid(vis) * mask
    x = x + x_res
    vis = seq(weights, p + 'vis_fc2.', x * vis, [elu, sigmoid]) * mask
    weight = vis / (np.sum(vis, 2, keepdims=True, dtype=np.float32) + F32(1e-8))
    mean, var = fused_mean_variance(x, weight)
    globalfeat = np.concatenate([mean[:, :, 0], var[:, :, 0], np.mean(weight, 2, dtype=np.float32)], -1)
    globalfeat = seq(weights, p + 'geometry_fc.', globalfeat, [elu, elu])        # rn,dn,16
    geo_feat = globalfeat
    num_valid_obs = np.sum(mask, 2, dtype=np.float32)                          # rn,dn,1
    globalfeat = globalfeat + posenc(16, dn)
    globalfeat = ray_attention(weights, p + 'ray_attention.', globalfeat, (num_valid_obs > 1).astype(np.float32))
    sigma = seq(weights, p + 'out_geometry_fc.', globalfeat, [elu, relu])
    sigma_out = np.where(num_valid_obs < 1, F32(0.0), sigma)
    x = np.concatenate([x, vis, ray_diff], -1)
    x = seq(weights, p + 'rgb_fc.', x, [elu, elu, None])
    x = np.where(mask == 0, F32(-1e9), x)
    blend = softmax(x, 2)
    rgb_out = np.sum(rgb_in * blend, 2, dtype=np.float32)
    out = np.concatenate([rgb_out, sigma_out], -1).astype(np.float32)
    if return_aux:
        return out, {'geo_feat': geo_feat, 'num_valid': num_valid_obs[..., 0]}
    return out


def get_dir_diff(prj_dir, que_dir):
    """network/aggregate_net.py:8-14"""
    rfn, qn, rn, dn, _ = prj_dir.shape
    diff = prj_dir - que_dir[None]
    dot = np.sum(prj_dir * que_dir[None], -1, keepdims=True, dtype=np.float32)
    d = np.concatenate([diff, dot], -1)
    return d.reshape(rfn, qn * rn, dn, -1).transpose(1, 2, 0, 3)


def agg_net_forward(weights, prefix, prj, que_dir, return_aux=False):
    """DefaultAggregationNet.forward, network/aggregate_net.py:34-68 -> density [qn,rn,dn], colors [qn,rn,dn,3]"""
    hit = (prj['hit_prob'] - F32(0.5)) * F32(2.0)
    vis = (prj['vis'] - F32(0.5)) * F32(2.0)
    rfn, qn, rn, dn, _ = hit.shape
    emb = seq(weights, prefix + 'prob_embed.', np.concatenate([prj['ray_feats'], hit, vis], -1), [relu, None])
    dir_diff = get_dir_diff(prj['dir'], que_dir)
    perm = lambda t: t.reshape(rfn, qn * rn, dn, -1).transpose(1, 2, 0, 3)
    mask = perm(prj['mask'])
    img = perm(np.concatenate([prj['rgb'], prj['img_feats']], -1))
    emb = perm(emb)
    res = ibrnet_forward(weights, prefix + 'agg_impl.', img, emb, dir_diff, mask, return_aux)
    outs, aux = res if return_aux else (res, None)
    colors = outs[..., :3].reshape(qn, rn, dn, 3)
    density = outs[..., 3].reshape(qn, rn, dn)
    if return_aux:
        return density, colors, aux
    return density, colors


# --------------------------------------------------------------------------------------
# a15-a17  compositing and fine sampling
# --------------------------------------------------------------------------------------
def alpha_values2hit_prob(alpha_values):
    """network/render_ops.py:72-80"""
    a = f32(alpha_values)
    no_hit = np.concatenate([np.ones(a.shape[:-1] + (1,), np.float32), F32(1.0) - a + F32(1e-10)], -1)
    return (a * np.cumprod(no_hit, -1, dtype=np.float32)[..., :-1]).astype(np.float32)


# --------------------------------------------------------------------------------------
# a18 direct rendering (cfg['use_dr_prediction'])      network/renderer.py:85-125, network/sph_solver.py:1-59
# --------------------------------------------------------------------------------------
SPH_REGS = np.concatenate([np.zeros(1), np.ones(3) * 0.001, np.ones(5) * 0.005, np.ones(7) * 0.05]).astype(np.float32)   # sph_solver.py:9-11


def sph_basis(d):
    """sph_solver.py:14-31, degree 3: [..., 3] -> [..., 16]"""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    l0 = [np.ones_like(x)]
    l1 = [x, y, z]
    l2 = [x * y, y * z, -x ** 2 - y ** 2 + F32(2) * z ** 2, z * x, x ** 2 - y ** 2]
    l3 = [(F32(3) * x ** 2 - y ** 2) * y, x * y * z, y * (F32(4) * z ** 2 - x ** 2 - y ** 2),
          z * (F32(2) * z ** 2 - F32(3) * x ** 2 - F32(3) * y ** 2), x * (F32(4) * z ** 2 - x ** 2 - y ** 2), (x ** 2 - y ** 2) * z,
          (x ** 2 - F32(3) * y ** 2) * x]
    return np.stack(l0 + l1 + l2 + l3, -1).astype(d.dtype)


def sph_fit(directions, colors, weights, regs=SPH_REGS, eps=1e-4):
    """SphericalHarmonicsSolver.forward (sph_solver.py:33-50): directions [b,n,3], colors [b,n,3], weights [b,n] -> theta [b,16,3]"""
    dt = directions.dtype
    A = sph_basis(directions)
    insufficient = np.sum(weights, 1, keepdims=True, dtype=dt) < eps
    weights = weights + insufficient.astype(dt) * dt.type(eps)
    A_ = np.transpose(A * weights[..., None], (0, 2, 1))
    inv_mat = A_ @ A + np.diag(regs.astype(dt))[None]
    return (np.linalg.inv(inv_mat) @ (A_ @ colors)).astype(dt)


def direct_rendering(cfg, prj, que_dir, colors_nr, dtype=np.float32):
    """renderer.py:85-125 -> hit_prob_dr [qn,rn,dn], colors [qn,rn,dn,3], pixel_colors_dr [qn,rn,3].
    dtype=np.float64: the same formulas in double (a yardstick for the 16 x 16 inverse, which is ill conditioned)."""
    dt = np.dtype(dtype)
    ground = dt.type(cfg['alpha_value_ground_state'])
    alpha_v, vis_v, mask = prj['alpha'].astype(dt), prj['vis'].astype(dt), prj['mask']
    alpha = np.sum(vis_v * alpha_v, 0, dtype=dt) / (np.sum(vis_v, 0, dtype=dt) + dt.type(1e-5))
    invalid = (np.sum(mask.astype(np.int32)[..., 0], 0) == 0).astype(dt)[..., None]
    alpha = (alpha * (1 - invalid) + invalid * ground)[..., 0]                 # qn,rn,dn
    if cfg.get('use_nr_color_for_dr', False):
        colors = colors_nr.astype(dt)
    else:
        rfn, qn, rn, dn, _ = prj['rgb'].shape
        pn = qn * rn * dn
        hit = prj['hit_prob'].reshape(rfn, pn, 1).astype(dt)
        w = hit / (np.sum(hit, 0, keepdims=True, dtype=dt) + dt.type(1e-3))
        theta = sph_fit(np.transpose(prj['dir'].reshape(rfn, pn, 3).astype(dt), (1, 0, 2)),
                        np.transpose(prj['rgb'].reshape(rfn, pn, 3).astype(dt), (1, 0, 2)), np.transpose(w[..., 0], (1, 0)))
        colors = (sph_basis(que_dir.reshape(pn, 1, 3).astype(dt)) @ theta)[:, 0].reshape(qn, rn, dn, 3)
    alpha_values = 1 / (1 + np.exp(-alpha))                                   # decode_alpha_value (dist_decoder.py:142-144)
    no_hit = np.concatenate([np.ones(alpha_values.shape[:-1] + (1,), dt), 1 - alpha_values + dt.type(1e-10)], -1)
    hit_prob = alpha_values * np.cumprod(no_hit, -1, dtype=dt)[..., :-1]
    return hit_prob.astype(dt), colors.astype(dt), np.sum(hit_prob[..., None] * colors, 2, dtype=dt)


def sample_fine_depth(depth, hit_prob, depth_range, fdn, u=None, trace=False):
    """network/render_ops.py:172-229 (inv_mode=True).  u=None -> deterministic stratified u
    (random_sample=False); otherwise u [qn,rn,fdn] is the externally drawn uniform sample.
    trace: -> (fine depths, searchsorted bins [qn,rn,fdn], cdf [qn,rn,dn+1]) - what tests/test_fine_index.py pins the kernel's index path to."""
    depth, hit_prob, depth_range = f32(depth), f32(hit_prob), f32(depth_range)
    near, far = F32(-1.0) / depth_range[0, 0], F32(-1.0) / depth_range[0, 1]
    s = (F32(-1.0) / depth - near) / (far - near)
    center = (s[..., 1:] + s[..., :-1]) / F32(2.0)
    edges = np.concatenate([s[..., 0:1], center, s[..., -1:]], -1)               # dn+1
    hp = hit_prob + F32(1e-5)
    pdf = hp / np.sum(hp, -1, keepdims=True, dtype=np.float32)
    cdf = np.cumsum(pdf, -1, dtype=np.float32)
    cdf = np.concatenate([np.zeros_like(cdf[..., :1]), cdf], -1)                # dn+1
    if u is None:
        interval = F32(1.0 / fdn) if False else np.float32(1 / fdn)
        u = F32(0.5) * interval + np.arange(fdn, dtype=np.float32) * interval
        u = np.broadcast_to(u, cdf.shape[:-1] + (fdn,)).astype(np.float32)
    else:
        u = f32(u)
    flat_cdf = cdf.reshape(-1, cdf.shape[-1])
    flat_u = u.reshape(-1, fdn)
    inds = np.stack([np.searchsorted(c, uu, side='right') for c, uu in zip(flat_cdf, flat_u)]).reshape(u.shape)
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, cdf.shape[-1] - 1)
    cdf_b = np.take_along_axis(cdf, below, -1)
    cdf_a = np.take_along_axis(cdf, above, -1)
    bin_b = np.take_along_axis(edges, below, -1)
    bin_a = np.take_along_axis(edges, above, -1)
    denom = cdf_a - cdf_b
    denom = np.where(denom < F32(1e-5), F32(1.0), denom)
    t = (u - cdf_b) / denom
    fine = bin_b + t * (bin_a - bin_b)
    fine = fine * (far - near) + near
    fine = (F32(-1.0) / fine).astype(np.float32)
    return (fine, inds, cdf) if trace else fine


# --------------------------------------------------------------------------------------
# a11, a12, a15, a16, a18, a19   renderer orchestration      network/renderer.py:67-226
# --------------------------------------------------------------------------------------
DEFAULT_CFG = {
    'use_hierarchical_sampling': False, 'fine_depth_sample_num': 64, 'fine_depth_use_all': False,
    'depth_sample_num': 64, 'alpha_value_ground_state': -15, 'use_self_hit_prob': False,
    'use_ray_mask': True, 'ray_mask_view_num': 2, 'ray_mask_point_num': 8, 'render_depth': False,
    'use_dr_prediction': False, 'use_nr_color_for_dr': False,
    'coarse_use_vis': True,   # = dist_decoder_cfg['use_vis'] of the COARSE decoder (quirk A.9.2)
    'fine_use_vis': True,     # = fine_dist_decoder_cfg['use_vis']
}


def predict_proj_ray_prob(weights, cfg, prj, ref_depth_range, que_dists, is_fine):
    """network/renderer.py:67-83.  compute_prob always uses the coarse decoder's use_vis."""
    prefix = 'fine_dist_decoder.' if is_fine else 'dist_decoder.'
    mean, var, vis, aw = dist_decoder_forward(weights, prefix, prj['ray_feats'])
    use_vis = cfg['coarse_use_vis']
    alpha, visib, hit = compute_prob(prj['depth'][..., 0], que_dists[None], mean, var, vis, aw, True,
                                     ref_depth_range, use_vis)
    m = prj['mask']
    prj['alpha'] = alpha[..., None] * m + (F32(1.0) - m) * F32(cfg['alpha_value_ground_state'])
    prj['vis'] = visib[..., None] * m
    prj['hit_prob'] = hit[..., None] * m
    prj['_mean'], prj['_var'], prj['_aw'] = mean, var, aw
    return prj


def predict_self_hit_prob(weights, cfg, que, que_depth, que_dists, is_fine):
    """network/renderer.py:137-155"""
    _, _, h, w = que['imgs'].shape
    qn, rn, _ = que['coords'].shape
    feats = interpolate_feature_map(que['ray_feats'], que['coords'], np.ones([qn, rn], np.float32), h, w)
    prefix = 'fine_dist_decoder.' if is_fine else 'dist_decoder.'
    use_vis = cfg['fine_use_vis'] if is_fine else cfg['coarse_use_vis']
    mean, var, vis, aw = dist_decoder_forward(weights, prefix, feats)
    e = lambda t: None if t is None else t[:, :, None]
    _, _, hit = compute_prob(que_depth, que_dists, e(mean), e(var), e(vis), e(aw), False, que['depth_range'], use_vis)
    return hit


def render_by_depth(weights, cfg, que_depth, que, ref, is_train, is_fine, return_aux=False):
    """network/renderer.py:168-203"""
    Ks_inv = que['Ks_inv'] if 'Ks_inv' in que else np.stack([inv3x3(K) for K in que['Ks']])
    que_dists = depth2inv_dists(que_depth, que['depth_range'])
    que_pts, que_dir = depth2points(que['coords'], que['poses'], Ks_inv, que_depth)
    prj = project_points_dict(ref, que_pts)
    prj = predict_proj_ray_prob(weights, cfg, prj, ref['depth_range'], que_dists, is_fine)
    rfn, _, h, w = ref['imgs'].shape
    _, qn, rn, dn, _ = prj['pts'].shape
    prj['img_feats'] = interpolate_feature_map(ref['img_feats'], prj['pts'].reshape(rfn, qn * rn * dn, 2),
                                               prj['mask'].reshape(rfn, qn * rn * dn), h, w).reshape(rfn, qn, rn, dn, -1)
    prefix = 'fine_agg_net.' if is_fine else 'agg_net.'
    res = agg_net_forward(weights, prefix, prj, que_dir, return_aux)
    density, colors = res[0], res[1]
    alpha_values = F32(1.0) - np.exp(-relu(density))
    hit_prob = alpha_values2hit_prob(alpha_values)
    pixel = np.sum(hit_prob[..., None] * colors, 2, dtype=np.float32)
    out = {'pixel_colors_nr': pixel, 'hit_prob_nr': hit_prob}
    if cfg.get('use_dr_prediction', False):                                    # renderer.py:181-185
        out['hit_prob_dr'], colors_dr, out['pixel_colors_dr'] = direct_rendering(cfg, prj, que_dir, colors)
        if cfg.get('_dr_float64', False):     # test hook: the same branch evaluated in double on the same fp32 inputs
            out['hit_prob_dr64'], _, out['pixel_colors_dr64'] = direct_rendering(cfg, prj, que_dir, colors, np.float64)
    if is_train and cfg['use_self_hit_prob']:
        out['hit_prob_self'] = predict_self_hit_prob(weights, cfg, que, que_depth, que_dists, is_fine)
    if 'imgs' in que:
        out['pixel_colors_gt'] = interpolate_feats(que['imgs'], que['coords'], align_corners=True)
    if cfg['use_ray_mask']:
        nv = np.sum(prj['mask'].astype(np.int32), 0) > cfg['ray_mask_view_num']       # qn,rn,dn,1
        out['ray_mask'] = (np.sum(nv, 2) > cfg['ray_mask_point_num'])[..., 0]
    if cfg['render_depth']:
        out['render_depth'] = np.sum(hit_prob * que_depth, -1, dtype=np.float32)
    if return_aux:
        aux = dict(res[2])
        aux.update({'que_pts': que_pts, 'que_dir': que_dir, 'que_dists': que_dists, 'prj': prj,
                    'density': density, 'colors': colors})
        return out, aux
    return out


def render_impl(weights, cfg, que, ref, is_train=False, u=None, coarse_hit_prob=None):
    """network/renderer.py:205-226 (coarse + optional fine pass).
    u: optional externally drawn uniforms [qn,rn,fdn] for is_train (the reference draws
    torch.rand on the CPU, render_ops.py:205).
    coarse_hit_prob: test hook - place the fine samples from this hit_prob instead of the
    coarse pass's own (fine-sample placement is ill-conditioned on near-empty rays, so stage-wise
    parity tests feed both implementations the same coarse result)."""
    cfg = {**DEFAULT_CFG, **cfg}
    qn, rn, _ = que['coords'].shape
    que_depth = sample_depth(que['depth_range'], rn, cfg['depth_sample_num'])
    out = render_by_depth(weights, cfg, que_depth, que, ref, is_train, False)
    if cfg['use_hierarchical_sampling']:
        hp = out['hit_prob_nr'] if coarse_hit_prob is None else coarse_hit_prob
        fine_depth = sample_fine_depth(que_depth, hp, que['depth_range'],
                                       cfg['fine_depth_sample_num'], u if is_train else None)
        if cfg['fine_depth_use_all']:
            d = np.sort(np.concatenate([que_depth, fine_depth], -1), -1)
        else:
            d = np.sort(fine_depth, -1)
        fine = render_by_depth(weights, cfg, d, que, ref, is_train, True)
        for k, v in fine.items():
            out[k + '_fine'] = v
        out['_fine_depth'] = d
    out['_coarse_depth'] = que_depth
    return out


# --------------------------------------------------------------------------------------
# synthetic scene generator + PSNR: shared input generator, lives in the package (numpy only)
# --------------------------------------------------------------------------------------
from neuray_amd.synthetic import look_at_pose, sphere_pos, make_scene, meshgrid_coords, psnr_uint8  # noqa: E402,F401


# --------------------------------------------------------------------------------------
# depth init net front end              network/init_net.py:13-76
# --------------------------------------------------------------------------------------
def extract_depth_for_init(depth_range, depth):
    """network/init_net.py:63-76: metric depth [rfn,1,h,w] -> normalised inverse depth in [0,1] of the view's own range"""
    depth_range, depth = f32(depth_range), f32(depth)
    near_inv = (F32(-1.0) / depth_range[:, 0])[:, None, None, None]
    far_inv = (F32(-1.0) / depth_range[:, 1])[:, None, None, None]
    d = F32(-1.0) / np.maximum(depth, F32(1e-5))
    return np.clip((d - near_inv) / (far_inv - near_inv), F32(0.0), F32(1.0)).astype(np.float32)


def normalised_to_metric_depth(depth_range, depth_norm):
    """network/init_net.py:32-37: back to metric depth (clamped to the view's range by the round trip)"""
    depth_range = f32(depth_range)
    near_inv = (F32(-1.0) / depth_range[:, 0])[:, None, None, None]
    far_inv = (F32(-1.0) / depth_range[:, 1])[:, None, None, None]
    return (F32(-1.0) / (f32(depth_norm) * (far_inv - near_inv) + near_inv)).astype(np.float32)


def depth2pts3d(depth, Ks_inv, poses):
    """network/init_net.py:13-28: every pixel (x, y) of every view lifted with its depth -> [rfn, h*w, 3] world points"""
    depth, poses = f32(depth), f32(poses)
    rfn, _, h, w = depth.shape
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    out = np.zeros([rfn, h * w, 3], np.float32)
    for v in range(rfn):
        d = depth[v, 0].reshape(-1)
        sx, sy, sz = d * xx.reshape(-1), d * yy.reshape(-1), d
        Ki = Ks_inv[v]
        cam = [dot3(Ki[i, 0], Ki[i, 1], Ki[i, 2], sx, sy, sz) for i in range(3)]
        c = camera_center(poses[v])
        P = poses[v]
        for j in range(3):
            out[v, :, j] = dot3(P[0, j], P[1, j], P[2, j], cam[0], cam[1], cam[2]) + c[j]
    return out


def get_diff_feats(ref, depth_norm, Ks_inv=None):
    """network/init_net.py:30-61 (+ masked_mean_var, ops.py:36-41) -> [rfn,8,h,w] = [rgb_mean 3, rgb_var 3, dpt_mean, dpt_var]"""
    imgs, poses, Ks, dr = f32(ref['imgs']), f32(ref['poses']), f32(ref['Ks']), f32(ref['depth_range'])
    rfn, _, h, w = imgs.shape
    depth = normalised_to_metric_depth(dr, depth_norm)
    if Ks_inv is None:
        Ks_inv = np.stack([inv3x3(K) for K in Ks])
    pts = depth2pts3d(depth, Ks_inv, poses).reshape(-1, 3)
    _, pts2d, prj_depth, valid = project_points_ref_views(poses, Ks, h, w, pts)        # [rfn, rfn*h*w, .]
    dpt_int = interpolate_feats(depth, pts2d, align_corners=True)                      # rfn,rfn*h*w,1
    rgb_int = interpolate_feats(imgs, pts2d, align_corners=True)                       # rfn,rfn*h*w,3
    rgb_diff = np.abs(rgb_int - imgs.transpose(0, 2, 3, 1).reshape(1, rfn * h * w, 3))
    dpt_diff = np.abs(F32(-1.0) / np.maximum(dpt_int, F32(1e-5)) + F32(1.0) / np.maximum(prj_depth, F32(1e-5)))
    near_inv, far_inv = (F32(-1.0) / dr[:, 0])[:, None, None], (F32(-1.0) / dr[:, 1])[:, None, None]
    dpt_diff = np.minimum(dpt_diff / (far_inv - near_inv), F32(1.5))
    m = valid.astype(np.float32)[..., None]

    def mean_var(x):
        msum = np.maximum(m.sum(0, keepdims=True), F32(1e-4))
        mean = (x * m).sum(0, keepdims=True) / msum
        return mean, ((x - mean) ** 2 * m).sum(0, keepdims=True) / msum
    dm, dv = mean_var(dpt_diff)
    rm, rv = mean_var(rgb_diff)
    to_map = lambda t, c: t.reshape(rfn, h, w, c).transpose(0, 3, 1, 2)
    return np.concatenate([to_map(rm, 3), to_map(rv, 3), to_map(dm, 1), to_map(dv, 1)], 1).astype(np.float32)


# --------------------------------------------------------------------------------------
# plane-sweep warp + variance     network/mvsnet/modules.py:25-64, mvsnet.py:186-203
# --------------------------------------------------------------------------------------
def homo_warp(src_feat, src_proj, ref_proj_inv, depth_values):
    """src_feat [B,C,H,W], 4x4 projections [B,4,4], depth_values [B,D] -> warped [B,C,D,H,W]
    (bilinear, zero padding, align_corners=True; z clamped to >= 1e-4)"""
    src_feat = f32(src_feat)
    B, C, H, W = src_feat.shape
    D = depth_values.shape[1]
    out = np.zeros([B, C, D, H, W], np.float32)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    for b in range(B):
        M = (f32(src_proj[b]) @ f32(ref_proj_inv[b]))[:3]
        for d in range(D):
            dv = F32(depth_values[b, d])
            gx, gy, gz = xx * dv, yy * dv, np.full_like(xx, dv)
            X = dot3(M[0, 0], M[0, 1], M[0, 2], gx, gy, gz) + M[0, 3]
            Y = dot3(M[1, 0], M[1, 1], M[1, 2], gx, gy, gz) + M[1, 3]
            Z = dot3(M[2, 0], M[2, 1], M[2, 2], gx, gy, gz) + M[2, 3]
            Z = np.where(Z < F32(1e-4), F32(1e-4), Z)
            ix = ((X / Z) / F32((W - 1) / 2) - F32(1) + F32(1)) / F32(2) * F32(W - 1)
            iy = ((Y / Z) / F32((H - 1) / 2) - F32(1) + F32(1)) / F32(2) * F32(H - 1)
            x0f, y0f = np.floor(ix), np.floor(iy)
            wx1, wy1 = ix - x0f, iy - y0f
            wx0, wy0 = (x0f + F32(1)) - ix, (y0f + F32(1)) - iy
            acc = np.zeros([C, H, W], np.float32)
            for (xf, yf, ww) in ((x0f, y0f, wx0 * wy0), (x0f + 1, y0f, wx1 * wy0), (x0f, y0f + 1, wx0 * wy1), (x0f + 1, y0f + 1, wx1 * wy1)):
                ok = (xf >= 0) & (xf <= W - 1) & (yf >= 0) & (yf <= H - 1)
                xi = np.where(ok, xf, 0).astype(np.int64)
                yi = np.where(ok, yf, 0).astype(np.int64)
                acc += src_feat[b][:, yi, xi] * np.where(ok, ww, F32(0))[None]
            out[b, :, d] = acc
    return out


def variance_volume(ref_feats, src_feats, nn_ids, ref_prjs, src_prjs, depth_values):
    """mvsnet.py:186-203 without the U-Net: [rfn,32,dn,h,w]"""
    ref_feats, src_feats = f32(ref_feats), f32(src_feats)
    rfn, n_num = nn_ids.shape
    inv = np.stack([np.linalg.inv(f32(p).astype(np.float64)).astype(np.float32) for p in ref_prjs])
    dn = depth_values.shape[1]
    s = np.repeat(ref_feats[:, :, None], dn, 2)
    sq = s * s
    for j in range(n_num):
        wv = homo_warp(src_feats[nn_ids[:, j]], f32(src_prjs)[nn_ids[:, j]], inv, depth_values)
        s = s + wv
        sq = sq + wv * wv
    V = F32(n_num + 1)
    return (sq / V - (s / V) ** 2).astype(np.float32)
