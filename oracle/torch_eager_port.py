"""Eager-PyTorch port of the reference's per-ray op sequence (TEST / BASELINE INFRASTRUCTURE, not product code).

The reference itself (liuyuan-pal/NeuRay) is not present on the GPU box, so the "stock PyTorch-ROCm" baseline of
BASELINE.md (B2, the denominator of the north star's ">= 10x") is measured with this port: the same op granularity
as network/renderer.py:168-226 and its callees - F.grid_sample on NCHW maps, one Linear per layer, the materialised
207-wide concat (ibrnet.py:342), cumprod / cumsum / searchsorted / sort - written against the numpy oracle
(oracle/neuray_oracle.py) and checked against the same reference-generated golden vectors
(tests/test_oracle_golden.py::test_torch_eager_port_matches_reference).  Only bench.py's baseline leg and tests/
import it.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _lin(x, w, p):
    return F.linear(x, w[p + 'weight'], w.get(p + 'bias'))


def _mlp(w, prefix, x, acts):
    for i, a in enumerate(acts):
        x = _lin(x, w, '%s%d.' % (prefix, 2 * i))
        if a is not None:
            x = a(x)
    return x


def _grid_gather(maps, pts, h, w):
    """interpolate_feature_map: bilinear, border padding, align_corners iff full resolution (render_ops.py:54-70)"""
    fh, fw = maps.shape[-2:]
    gx = pts[..., 0] / (w - 1) * 2 - 1
    gy = pts[..., 1] / (h - 1) * 2 - 1
    grid = torch.stack([gx, gy], -1).unsqueeze(1)
    out = F.grid_sample(maps, grid, mode='bilinear', padding_mode='border', align_corners=(fh == h and fw == w))
    return out.squeeze(2).permute(0, 2, 1)


def _posenc(dn, device, dtype=torch.float32):
    pos = np.arange(dn, dtype=np.float64)[:, None]
    j = np.arange(16)[None, :]
    t = pos / np.power(10000, 2 * (j // 2) / 16)
    t[:, 0::2] = np.sin(t[:, 0::2])
    t[:, 1::2] = np.cos(t[:, 1::2])
    return torch.from_numpy(t.astype(np.float32)).to(device=device, dtype=dtype)[None]     # (the reference's table is built in fp32: ibrnet.py:305-313)


def render_pass(w, cfg, depth, que, ref, is_fine):
    """one render_by_depth (renderer.py:168-203), eval mode.  depth [1,rn,dn]"""
    dp = 'fine_dist_decoder.' if is_fine else 'dist_decoder.'
    ap = ('fine_agg_net.' if is_fine else 'agg_net.')
    ip = ap + 'agg_impl.'
    dev = depth.device
    rn, dn = depth.shape[1:]
    rfn, _, h, wd = ref['imgs'].shape
    # ---- geometry
    near_q, far_q = -1 / que['depth_range'][:, 0], -1 / que['depth_range'][:, 1]
    s = (-1 / depth - near_q[:, None, None]) / (far_q - near_q)[:, None, None]
    dists = torch.cat([s[..., 1:] - s[..., :-1], torch.full_like(s[..., :1], 1e6)], -1)
    R, t = que['poses'][:, :, :3], que['poses'][:, :, 3:]
    rot = R.transpose(1, 2)
    centre = (-rot @ t).squeeze(-1)                                             # 1,3
    homog = torch.cat([que['coords'], torch.ones_like(que['coords'][..., :1])], -1)
    cam = (torch.inverse(que['Ks']).unsqueeze(1) @ homog.unsqueeze(-1))
    dirs = (rot.unsqueeze(1) @ cam).squeeze(-1) + centre[:, None] - centre[:, None]
    pts = centre[:, None, None] + dirs[:, :, None] * depth[..., None]           # 1,rn,dn,3
    que_dir = (-dirs / dirs.norm(dim=-1, keepdim=True))[:, :, None].expand_as(pts)
    P = pts.reshape(-1, 3)
    H = ref['Ks'] @ ref['poses']
    hp = torch.cat([P, torch.ones_like(P[:, :1])], 1)
    pc = (H[:, None] @ hp[None, :, :, None])[..., 0]                            # rfn,pn,3
    z = pc[..., 2:].clone()
    bad = z.abs() < 1e-4
    z[bad] = 1e-3
    uv = pc[..., :2] / z
    mask = (~bad[..., 0]) & ~((uv[..., 0] < -0.5) | (uv[..., 0] >= wd - 0.5) | (uv[..., 1] < -0.5) | (uv[..., 1] >= h - 0.5))
    cv = (-ref['poses'][:, :, :3].transpose(1, 2) @ ref['poses'][:, :, 3:]).transpose(1, 2)
    dvec = P[None] - cv
    prj_dir = -dvec / dvec.norm(dim=2, keepdim=True).clamp_min(1e-5)
    mf = mask.to(depth.dtype).unsqueeze(-1)
    f_ray = _grid_gather(ref['ray_feats'], uv, h, wd) * mf
    rgb = _grid_gather(ref['imgs'], uv, h, wd) * mf
    f_img = _grid_gather(ref['img_feats'], uv, h, wd) * mf
    shp = (rfn, 1, rn, dn, -1)
    f_ray, rgb, f_img, prj_dir, mf, z = [x.reshape(shp) for x in (f_ray, rgb, f_img, prj_dir, mf, z)]
    # ---- dist decoder + probabilities
    sp = F.softplus
    mean = _mlp(w, dp + 'mean_decoder.', f_ray, [F.elu, F.elu, sp])
    var = _mlp(w, dp + 'var_decoder.', f_ray, [F.elu, F.elu, sp]) + 0.05
    aw = _mlp(w, dp + 'aw_decoder.', f_ray, [F.elu, F.elu, torch.sigmoid])
    use_vis = cfg['coarse_use_vis']
    vis_d = _mlp(w, dp + 'vis_decoder.', f_ray, [F.elu, F.elu, torch.sigmoid]) if (dp + 'vis_decoder.0.weight') in w else None
    near_r = (-1 / ref['depth_range'][:, 0])[:, None, None, None]
    far_r = (-1 / ref['depth_range'][:, 1])[:, None, None, None]
    tt = (-1 / z[..., 0].clamp_min(1e-5) - near_r) / (far_r - near_r)
    half = dists.unsqueeze(0) / 2
    ext = torch.cat([half[..., :1], half], -1)
    lo, hi = (tt - ext[..., :-1])[..., None], (tt + ext[..., 1:])[..., None]
    c0 = 0.5 + 0.5 * torch.tanh((lo - mean) * var)
    c1 = 0.5 + 0.5 * torch.tanh((hi - mean) * var)
    if use_vis:
        c0, c1 = c0 * vis_d, c1 * vis_d
    mix = torch.cat([aw, 1 - aw], -1)
    vis = ((1 - c0) * mix).sum(-1, keepdim=True) * mf
    hit = ((c1 - c0) * mix).sum(-1, keepdim=True) * mf
    # ---- aggregation (aggregate_net.py:34-68, ibrnet.py:315-369)
    emb = _mlp(w, ap + 'prob_embed.', torch.cat([f_ray, (hit - 0.5) * 2, (vis - 0.5) * 2], -1), [F.relu, None])
    perm = lambda x: x.reshape(rfn, rn, dn, -1).permute(1, 2, 0, 3)
    ddiff = torch.cat([prj_dir - que_dir[None], (prj_dir * que_dir[None]).sum(-1, keepdim=True)], -1)
    ddiff, m, emb = perm(ddiff), perm(mf), perm(emb)
    feat = perm(torch.cat([rgb, f_img], -1))
    rgb_in = feat[..., :3]
    feat = feat + _mlp(w, ip + 'ray_dir_fc.', ddiff, [F.elu, F.elu])
    wgt = m / (m.sum(2, keepdim=True) + 1e-8)
    wgt0 = torch.sigmoid(_mlp(w, ip + 'neuray_fc.', emb, [F.elu, None])) * wgt

    def mv(x, ww):
        mu = (x * ww).sum(2, keepdim=True)
        return mu, (ww * (x - mu) ** 2).sum(2, keepdim=True)

    m0, v0 = mv(feat, wgt0)
    m1, v1 = mv(feat, wgt)
    glob = torch.cat([m0, v0, m1, v1], -1)
    x = torch.cat([glob.expand(-1, -1, rfn, -1), feat, emb], -1)
    x = _mlp(w, ip + 'base_fc.', x, [F.elu, F.elu])
    xv = _mlp(w, ip + 'vis_fc.', x * wgt, [F.elu, F.elu])
    xres, v = xv[..., :-1], xv[..., -1:]
    v = torch.sigmoid(v) * m
    x = x + xres
    v = _mlp(w, ip + 'vis_fc2.', x * v, [F.elu, torch.sigmoid]) * m
    wgt = v / (v.sum(2, keepdim=True) + 1e-8)
    mu, va = mv(x, wgt)
    g = _mlp(w, ip + 'geometry_fc.', torch.cat([mu.squeeze(2), va.squeeze(2), wgt.mean(2)], -1), [F.elu, F.elu])
    nvalid = m.sum(2)
    g = g + _posenc(dn, dev, g.dtype)
    q = _lin(g, w, ip + 'ray_attention.w_qs.').view(rn, dn, 4, 4).transpose(1, 2)
    k = _lin(g, w, ip + 'ray_attention.w_ks.').view(rn, dn, 4, 4).transpose(1, 2)
    vv = _lin(g, w, ip + 'ray_attention.w_vs.').view(rn, dn, 4, 4).transpose(1, 2)
    att = (q / 2) @ k.transpose(2, 3)
    att = att.masked_fill(((nvalid > 1).float().unsqueeze(1)) == 0, -1e9)
    o = (F.softmax(att, -1) @ vv).transpose(1, 2).reshape(rn, dn, 16)
    o = F.layer_norm(_lin(o, w, ip + 'ray_attention.fc.') + g, (16,), w[ip + 'ray_attention.layer_norm.weight'],
                     w[ip + 'ray_attention.layer_norm.bias'], 1e-6)
    sigma = _mlp(w, ip + 'out_geometry_fc.', o, [F.elu, F.relu]).masked_fill(nvalid < 1, 0.)
    logit = _mlp(w, ip + 'rgb_fc.', torch.cat([x, v, ddiff], -1), [F.elu, F.elu, None]).masked_fill(m == 0, -1e9)
    colors = (rgb_in * F.softmax(logit, 2)).sum(2)                                  # rn,dn,3
    # ---- compositing
    alpha = 1 - torch.exp(-torch.relu(sigma[..., 0]))
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1), -1)[:, :-1]
    hp_ = alpha * T
    out = {'pixel_colors_nr': (hp_.unsqueeze(-1) * colors).sum(1)[None], 'hit_prob_nr': hp_[None]}
    cnt = (mf.reshape(rfn, rn, dn).sum(0) > cfg.get('ray_mask_view_num', 2)).sum(1)
    out['ray_mask'] = (cnt > cfg.get('ray_mask_point_num', 8))[None]
    return out


def self_hit_prob(w, cfg, depth, que, is_fine):
    """predict_self_hit_prob (renderer.py:137-155): the query view's own distribution along its rays"""
    dp = 'fine_dist_decoder.' if is_fine else 'dist_decoder.'
    _, _, h, wd = que['imgs'].shape
    f = _grid_gather(que['ray_feats'], que['coords'], h, wd)                       # 1,rn,32
    sp = F.softplus
    mean = _mlp(w, dp + 'mean_decoder.', f, [F.elu, F.elu, sp]).unsqueeze(2)
    var = (_mlp(w, dp + 'var_decoder.', f, [F.elu, F.elu, sp]) + 0.05).unsqueeze(2)
    aw = _mlp(w, dp + 'aw_decoder.', f, [F.elu, F.elu, torch.sigmoid]).unsqueeze(2)
    use_vis = cfg['fine_use_vis'] if is_fine else cfg['coarse_use_vis']
    near_q, far_q = -1 / que['depth_range'][:, 0], -1 / que['depth_range'][:, 1]
    s = (-1 / depth - near_q[:, None, None]) / (far_q - near_q)[:, None, None]
    dists = torch.cat([s[..., 1:] - s[..., :-1], torch.full_like(s[..., :1], 1e6)], -1)
    t = (-1 / depth.clamp_min(1e-5) - near_q[:, None, None]) / (far_q - near_q)[:, None, None]
    half = dists / 2
    ext = torch.cat([(t[..., 0] - half[..., 0])[..., None], (t[..., :-1] + t[..., 1:]) / 2, (t[..., -1] + half[..., -1])[..., None]], -1)
    lo, hi = ext[..., :-1, None], ext[..., 1:, None]
    c0 = 0.5 + 0.5 * torch.tanh((lo - mean) * var)
    c1 = 0.5 + 0.5 * torch.tanh((hi - mean) * var)
    if use_vis:
        vis_d = _mlp(w, dp + 'vis_decoder.', f, [F.elu, F.elu, torch.sigmoid]).unsqueeze(2)
        c0, c1 = c0 * vis_d, c1 * vis_d
    mix = torch.cat([aw, 1 - aw], -1)
    return ((c1 - c0) * mix).sum(-1)


def sample_fine(depth, hit, depth_range, fdn, u=None):
    """render_ops.py:172-229; u = externally drawn uniforms (training) or None (stratified)"""
    near, far = -1 / depth_range[0, 0], -1 / depth_range[0, 1]
    s = (-1 / depth - near) / (far - near)
    edges = torch.cat([s[..., :1], (s[..., 1:] + s[..., :-1]) / 2, s[..., -1:]], -1)
    pdf = hit + 1e-5
    pdf = pdf / pdf.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    if u is None:
        interval = 1 / fdn
        u = (0.5 * interval + torch.arange(fdn, device=depth.device, dtype=depth.dtype) * interval).expand(list(cdf.shape[:-1]) + [fdn])
    u = u.to(depth.device).contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    lo, hi = (idx - 1).clamp_min(0), idx.clamp_max(cdf.shape[-1] - 1)
    cl, ch = torch.gather(cdf, -1, lo), torch.gather(cdf, -1, hi)
    el, eh = torch.gather(edges, -1, lo), torch.gather(edges, -1, hi)
    den = ch - cl
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    fine = el + (u - cl) / den * (eh - el)
    return -1 / (fine * (far - near) + near)


def render_impl(w, cfg, que, ref, is_train=False, u=None):
    """coarse + fine (renderer.py:217-226); tensors on any device.  Differentiable: autograd through this function is
    the gradient oracle for the backward kernels (checked against the reference's own autograd,
    tests/test_oracle_golden.py::test_torch_eager_port_gradients_match_reference)."""
    rn = que['coords'].shape[1]
    dn = cfg['depth_sample_num']
    near, far = que['depth_range'][:, 0], que['depth_range'][:, 1]
    ticks = torch.arange(dn, device=near.device, dtype=near.dtype)[None] * ((1 / far - 1 / near) / (dn - 1))[:, None]
    ticks[:, -1] = (1 / far - 1 / near)
    depth = (1 / (1 / near[:, None] + ticks))[:, None].expand(-1, rn, -1).contiguous()
    out = render_pass(w, cfg, depth, que, ref, False)
    self_hp = is_train and cfg.get('use_self_hit_prob', False)
    if self_hp:
        out['hit_prob_self'] = self_hit_prob(w, cfg, depth, que, False)
    if cfg.get('use_hierarchical_sampling', False):
        fd = torch.sort(sample_fine(depth, out['hit_prob_nr'].detach(), que['depth_range'], cfg['fine_depth_sample_num'],
                                    u if is_train else None), -1)[0]
        if cfg.get('fine_depth_use_all', False):
            fd = torch.sort(torch.cat([depth, fd], -1), -1)[0]
        for k, v in render_pass(w, cfg, fd, que, ref, True).items():
            out[k + '_fine'] = v
        if self_hp:
            out['hit_prob_self_fine'] = self_hit_prob(w, cfg, fd, que, True)
        out['_fine_depth'] = fd.detach()
    out['_coarse_depth'] = depth.detach()
    return out


def get_diff_feats(info, depth_in):
    """The tensor formulation of network/init_net.py:13-61 (depth2pts3d, project_points_ref_views, two grid_samples on
    [rfn, rfn*h*w, .] tensors, masked_mean_var) - the eager baseline beside the fused neuray_diff_feats kernel."""
    imgs, dr, Ks, poses = info['imgs'], info['depth_range'], info['Ks'], info['poses']
    rfn, _, h, w = imgs.shape
    dev = imgs.device
    near, far = dr[:, 0][:, None, None], dr[:, 1][:, None, None]
    near_inv, far_inv = -1 / near[..., None], -1 / far[..., None]
    depth = -1 / (depth_in * (far_inv - near_inv) + near_inv)
    ys, xs = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing='ij')
    coords = torch.stack([xs, ys, torch.ones_like(xs)], -1).float()[None, :, :, None, :]          # 1,h,w,1,3
    pts = (depth.permute(0, 2, 3, 1).unsqueeze(-1) * coords).reshape(rfn, h * w, 3).permute(0, 2, 1)
    pts = torch.inverse(Ks) @ pts
    R = poses[:, :, :3].permute(0, 2, 1)
    pts = (R @ pts + (-R @ poses[:, :, 3:])).permute(0, 2, 1).reshape(-1, 3)                      # rfn*h*w,3
    hom = torch.cat([pts, torch.ones_like(pts[:, :1])], 1)
    cam = hom[None] @ (Ks @ poses).permute(0, 2, 1)                                               # rfn,N,3
    z = cam[..., 2:]
    bad = z.abs() < 1e-4
    z = torch.where(bad, torch.full_like(z, 1e-3), z)
    uv = cam[..., :2] / z
    valid = (~bad[..., 0]) & (uv[..., 0] >= -0.5) & (uv[..., 0] < w - 0.5) & (uv[..., 1] >= -0.5) & (uv[..., 1] < h - 0.5)
    d_int = _grid_gather(depth, uv, h, w)
    c_int = _grid_gather(imgs, uv, h, w)
    rgb_diff = (c_int - imgs.permute(0, 2, 3, 1).reshape(1, rfn * h * w, 3)).abs()
    dpt_diff = (-1 / d_int.clamp(min=1e-5) + 1 / z.clamp(min=1e-5)).abs() / ((-1 / far) - (-1 / near))
    dpt_diff = dpt_diff.clamp(max=1.5)
    m = valid.float().unsqueeze(-1)

    def mean_var(x):
        msum = m.sum(0, keepdim=True).clamp_min(1e-4)
        mean = (x * m).sum(0, keepdim=True) / msum
        return mean, ((x - mean) ** 2 * m).sum(0, keepdim=True) / msum
    dm, dv = mean_var(dpt_diff)
    rm, rv = mean_var(rgb_diff)
    to_map = lambda t, c: t.reshape(rfn, h, w, c).permute(0, 3, 1, 2)
    return torch.cat([to_map(rm, 3), to_map(rv, 3), to_map(dm, 1), to_map(dv, 1)], 1)


def variance_volume(ref_feats, src_feats, nn_ids, ref_prjs, src_prjs, depth_values):
    """The tensor formulation of network/mvsnet/mvsnet.py:186-203 + modules.py:25-64 (one [B,32,D,h,w] grid_sample per
    source view, two running sums) - the eager baseline beside the fused neuray_warp_variance kernel."""
    B, C, H, W = ref_feats.shape
    D = depth_values.shape[1]
    dev = ref_feats.device
    inv = torch.inverse(ref_prjs)
    s = ref_feats.unsqueeze(2).repeat(1, 1, D, 1, 1)
    sq = s ** 2
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing='ij')
    grid = torch.stack([xs, ys, torch.ones_like(xs)], 0).reshape(1, 3, H * W).expand(B, -1, -1)
    n_num = nn_ids.shape[1]
    for j in range(n_num):
        tr = src_prjs[nn_ids[:, j]] @ inv
        R, T = tr[:, :3, :3], tr[:, :3, 3:]
        g = (grid.unsqueeze(2) * depth_values.view(B, 1, D, 1)).reshape(B, 3, D * H * W)
        p = R @ g + T
        z = p[:, 2:].clamp(min=1e-4)
        xy = p[:, :2] / z
        gx = xy[:, 0] / ((W - 1) / 2) - 1
        gy = xy[:, 1] / ((H - 1) / 2) - 1
        samp = torch.stack([gx, gy], -1).view(B, D, H * W, 2)
        wv = F.grid_sample(src_feats[nn_ids[:, j]], samp, mode='bilinear', padding_mode='zeros', align_corners=True).view(B, C, D, H, W)
        s = s + wv
        sq = sq + wv ** 2
    V = n_num + 1
    return sq / V - (s / V) ** 2
