"""Multi-GPU partitioning of the render path.

Rays are independent and the renderer is bitwise batching-invariant, so the path shards with NO data-path
collective: every rank holds the weights and the reference maps and renders a disjoint set of images, or a disjoint
contiguous ray range of one image.  The only communication is an optional all-gather of the rendered tiles when one
image is split (RCCL over xGMI on the GPU box: torch.distributed backend "nccl"; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced [start, end) of n items for `rank` of `world` (first n % world ranks get one extra)."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def images_for_rank(num_images, rank, world):
    """Image-level sharding (zero communication): rank r renders images r, r + world, ..."""
    return list(range(rank, num_images, world))


def render_ray_shard(renderer, que_imgs_info, ref_imgs_info, rank, world, is_train=False):
    """Render this rank's contiguous ray range of one query image.  Returns (outputs, (start, end))."""
    coords = que_imgs_info['coords']
    start, end = shard_range(coords.shape[1], rank, world)
    q = dict(que_imgs_info)
    q['coords'] = coords[:, start:end]
    return renderer.render(q, ref_imgs_info, is_train), (start, end)


def gather_tiles(local, n_total, rank, world, group=None):
    """All-gather per-ray tensors [1, n_local, ...] of the ranks' contiguous shards into [1, n_total, ...] with ONE
    collective: every output key is flattened to columns of one fp32 buffer [pad, C] (bools as 0/1 - exact), the buffer
    is all-gathered once, and the columns are cut back (launch latency dominates the 7.7 MB of an
    800 x 800 image, so one fused all-gather per image).  Keys are walked in sorted order, which every rank agrees on."""
    if n_total < world:
        raise ValueError("neuray_amd.parallel: %d rays cannot be split over %d ranks (every rank needs at least one ray)" % (n_total, world))
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    pad, keys = max(sizes), sorted(local)
    cols, spec = [], []
    for k in keys:
        v = local[k]
        assert v.shape[0] == 1 and v.shape[1] == sizes[rank], (k, tuple(v.shape))
        if v.dtype not in (torch.float32, torch.bool):       # the fp32 round trip is exact for these only
            raise TypeError("neuray_amd.parallel.gather_tiles: output %r is %s; only float32 and bool tiles are gathered" % (k, v.dtype))
        flat = v[0].reshape(v.shape[1], -1).to(torch.float32)
        spec.append((k, v.dtype, tuple(v.shape[2:]), flat.shape[1]))
        cols.append(flat)
    packed = torch.cat(cols, 1)
    buf = torch.zeros(pad, packed.shape[1], dtype=torch.float32, device=packed.device)
    buf[:packed.shape[0]] = packed
    parts = torch.empty(world, pad, packed.shape[1], dtype=torch.float32, device=packed.device)
    dist.all_gather_into_tensor(parts.view(world * pad, -1), buf, group=group)
    full = torch.cat([parts[r, :s] for r, s in enumerate(sizes)], 0)           # [n_total, C]
    out, c0 = {}, 0
    for k, dtype, tail, c in spec:
        t = full[:, c0:c0 + c].reshape((1, n_total) + tail)
        out[k] = t > 0.5 if dtype == torch.bool else t.to(dtype)
        c0 += c
    return out


def encode_views_sharded(renderer, ref_imgs_info, group=None):
    """Single-image latency path of SURVEY.md 8(e): the per-image encoder phase (renderer.py:229-235: `image_encoder` on the
    reference images, `vis_encoder` on the initial ray_feats) sharded BY VIEW - rank r encodes the contiguous views
    shard_range(rfn, r, world) - followed by ONE all-gather of a packed buffer [views, (C_img + C_ray) * fh * fw] (17.9 MB
    per view at 800 x 800: far below a millisecond over xGMI, so one fused collective).  Every rank ends up with the full
    `img_feats` / `ray_feats` in `ref_imgs_info`, exactly what the replicated `renderer.render()` computes (the encoders are
    per-sample networks: InstanceNorm, no batch statistics).  No-op when the maps are already there."""
    if 'img_feats' in ref_imgs_info:
        return ref_imgs_info
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    imgs, ray0 = ref_imgs_info['imgs'], ref_imgs_info['ray_feats']
    rfn = imgs.shape[0]
    spans = [shard_range(rfn, r, world) for r in range(world)]
    lo, hi = spans[rank]
    pad = max(e - s for s, e in spans)
    with torch.no_grad():
        if hi > lo:
            f_img = renderer.image_encoder(imgs[lo:hi])
            f_ray = renderer.vis_encoder(ray0[lo:hi], f_img)
            mine = torch.cat([f_img.reshape(hi - lo, -1), f_ray.reshape(hi - lo, -1)], 1).float()
            shapes = torch.tensor([f_img.shape[1], f_ray.shape[1], f_img.shape[2], f_img.shape[3]], dtype=torch.int64)
        else:
            mine, shapes = None, torch.zeros(4, dtype=torch.int64)
    # ranks without a view learn the map shape (rfn < world only): host tensors over the gloo side group - no wait for the device queue
    dist.all_reduce(shapes, op=dist.ReduceOp.MAX, group=_meta_group(group))
    c_img, c_ray, fh, fw = (int(v) for v in shapes.tolist())
    width = (c_img + c_ray) * fh * fw
    buf = torch.zeros(pad, width, dtype=torch.float32, device=imgs.device)
    if mine is not None:
        buf[:hi - lo] = mine
    parts = torch.empty(world * pad, width, dtype=torch.float32, device=imgs.device)
    dist.all_gather_into_tensor(parts, buf, group=group)
    full = torch.cat([parts[r * pad:r * pad + (e - s)] for r, (s, e) in enumerate(spans)], 0)      # [rfn, width]
    ref_imgs_info['img_feats'] = full[:, :c_img * fh * fw].reshape(rfn, c_img, fh, fw).contiguous()
    ref_imgs_info['ray_feats'] = full[:, c_img * fh * fw:].reshape(rfn, c_ray, fh, fw).contiguous()
    return ref_imgs_info


def render_image_sharded(renderer, que_imgs_info, ref_imgs_info, group=None, shard_encoders=True):
    """One image split over all ranks of the default (or given) process group; every rank gets the full image.
    shard_encoders: when `ref_imgs_info` still holds images + initial ray_feats (a renderer with encoders), encode the views
    sharded by view and all-gather the maps (encode_views_sharded) instead of every rank encoding all of them."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = que_imgs_info['coords'].shape[1]
    if n < world:
        raise ValueError("neuray_amd.parallel: %d rays cannot be split over %d ranks" % (n, world))
    if shard_encoders and 'img_feats' not in ref_imgs_info and hasattr(renderer, 'image_encoder'):
        ref_imgs_info = encode_views_sharded(renderer, dict(ref_imgs_info), group)
    local, _ = render_ray_shard(renderer, que_imgs_info, ref_imgs_info, rank, world)
    return gather_tiles(local, n, rank, world, group)


# ---- gradient flags without a device read-back -------------------------------------------------------------------------------------
# Which parameters received a gradient on SOME rank decides `grad = None` (torch.optim skips the parameter, as the reference's
# single-process training does) against `grad = the reduced sum`.  That union rides in the all-reduced buffer, i.e. on the device:
# reading it back (`.tolist()`) made every rank wait for its own backward AND the collective before it could queue the optimiser
# step.  Which parameters a step reaches is a property of the graph, not of the data, so the union of step t is the union of step
# t - 1: a step uses the previous step's union (a host-side list) and copies its own flags to pinned memory without blocking; the
# NEXT call checks that copy (long complete by then) against what was assumed and raises if a rank's pattern had changed
# unannounced.  The first step, and any step whose LOCAL pattern differs from the previous one, reads the flags the blocking way.
_FLAG_STATE = {}
DEFERRED_FLAGS = True           # False: always read the flags back at the call (the round-5 behaviour)


class _Pending:
    def __init__(self, flags_dev, assumed):
        self.assumed = assumed
        if flags_dev.is_cuda:
            self.host = torch.empty(flags_dev.shape, dtype=flags_dev.dtype, pin_memory=True)
            self.host.copy_(flags_dev, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(flags_dev.device))
        else:
            self.host, self.event = flags_dev.clone(), None

    def union(self):
        if self.event is not None:
            self.event.synchronize()
        return [v > 0 for v in self.host.tolist()]


def check_deferred_flags():
    """Drain the pending flag copies (end of training / before a checkpoint): raises if the last step assumed the wrong union."""
    for key, st in list(_FLAG_STATE.items()):
        pend = st.pop('pending', None)
        if pend is not None and pend.union() != pend.assumed:
            _FLAG_STATE.pop(key, None)
            raise RuntimeError("neuray_amd.parallel.allreduce_gradients: the set of parameters that received a gradient on some rank changed "
                               "without this rank's own set changing; the previous optimiser step treated %d parameter(s) by the older set"
                               % sum(a != b for a, b in zip(pend.union(), pend.assumed)))


def allreduce_gradients(parameters, average=True, group=None, skip=()):
    """Data-parallel training step: sum (or average) the gradients of `parameters` over the ranks with
    ONE flattened all-reduce (the shared nets are ~2.2 M parameters = 8.7 MB: a single bucket; on the MI355X node
    this is RCCL over xGMI via backend 'nccl').  A parameter that received a gradient on SOME rank ends up with the same
    gradient on every rank; a parameter that received none anywhere keeps `grad = None` everywhere (one flag per
    parameter rides in the same buffer), so torch.optim skips it on every replica exactly as the reference's
    single-process training does.  No device -> host wait in the steady state (see _FLAG_STATE above).  `skip`: parameters to
    leave alone - in fine-tuning mode the per-view `NeuralRayFtRenderer.ray_feats` (512 MB on lego-800), which go through
    allreduce_scene_feature_gradients instead."""
    skip_ids = {id(p) for p in skip}
    params = [p for p in parameters if p.requires_grad and id(p) not in skip_ids]
    if not params:
        return
    dev = params[0].device
    local = [p.grad is not None for p in params]
    key = (tuple(id(p) for p in params), id(group))
    st = _FLAG_STATE.get(key)
    if st is not None and 'pending' in st:               # what the previous step assumed, against what its flags turned out to be
        pend = st.pop('pending')
        actual = pend.union()
        if actual != pend.assumed:
            _FLAG_STATE.pop(key, None)
            raise RuntimeError("neuray_amd.parallel.allreduce_gradients: the set of parameters that received a gradient on some rank changed "
                               "without this rank's own set changing (%d parameter(s)); the previous optimiser step used the older set"
                               % sum(a != b for a, b in zip(actual, pend.assumed)))
    flags = torch.tensor([1.0 if f else 0.0 for f in local], dtype=torch.float32, device=dev)
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params] + [flags])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if DEFERRED_FLAGS and st is not None and st['local'] == local:
        union = st['union']
        st['pending'] = _Pending(flat[-len(params):], union)
    else:
        union = [v > 0 for v in flat[-len(params):].tolist()]
        _FLAG_STATE[key] = {'local': local, 'union': union}
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p, any_grad in zip(params, union):
        n = p.numel()
        if any_grad:
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
        off += n


def shared_parameters(model):
    """every parameter of a renderer except the per-view `ray_feats` maps of NeuralRayFtRenderer"""
    own = {id(p) for p in getattr(model, 'ray_feats', [])}
    return [p for p in model.parameters() if id(p) not in own]


def train_step(model, data, loss_fn, optimizer, group=None):
    """One data-parallel training step around the reference's `train_network(train_data)` call (train/trainer.py:123;
    the reference itself refuses multi-GPU training, trainer.py:65-70): every rank runs forward + backward on its own
    sample through the HIP kernels, the shared networks' gradients are summed with one all-reduce, a fine-tuning
    renderer's per-view maps with the sparse exchange, then every replica takes the same optimiser step.
    -> (outputs, loss)"""
    optimizer.zero_grad(set_to_none=True)
    outputs = model(data)
    loss = loss_fn(outputs)
    loss.backward()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        ray_feats = getattr(model, 'ray_feats', None)
        allreduce_gradients(model.parameters(), group=group, skip=list(ray_feats) if ray_feats is not None else ())
        if ray_feats is not None:
            # the views this rank's step gave a gradient to.  The mirror class records them (`touched_views`); the
            # reference's own NeuralRayFtRenderer (renderer.py:437, what integrate.patch_reference() runs) has only the
            # ParameterList, so there the set is read off the gradients themselves
            touched = getattr(model, 'touched_views', None)
            if touched is None:
                touched = [i for i, p in enumerate(ray_feats) if p.grad is not None]
            allreduce_scene_feature_gradients(ray_feats, touched, group=group)
    optimizer.step()
    return outputs, loss


_META_GROUPS = {}


def _meta_group(group=None):
    """A gloo process group over the same ranks as `group` for HOST-side metadata (created on first use; every rank reaches this
    point in the same step).  With a gloo default group it is the group itself."""
    if dist.get_backend(group) == 'gloo':
        return group
    key = id(group)
    if key not in _META_GROUPS:
        ranks = None if group is None else dist.get_process_group_ranks(group)
        _META_GROUPS[key] = dist.new_group(ranks=ranks, backend='gloo')
    return _META_GROUPS[key]


def allreduce_scene_feature_gradients(ray_feats, touched, max_touched=None, average=True, group=None):
    """Fine-tuning mode (SURVEY.md 8(e) caveat): `ray_feats` is the per-view nn.ParameterList of NeuralRayFtRenderer
    (100 views x 5 MB on lego-800) and a step gives gradients to the <= neighbor_view_num + 1 views this rank rendered
    from (`touched`: their indices, e.g. renderer.touched_views).  Instead of reducing all 512 MB (or marking unused
    parameters), the ranks all-gather their touched ids (fixed length `max_touched`, padded with -1) and all-reduce only
    the union's maps in one flattened buffer (<= world * 9 x 5 MB).  Afterwards every rank holds the same gradient for
    every view of the union and `grad is None` for all others, so per-parameter Adam state advances identically on all
    ranks (torch.optim skips parameters without a gradient, as the reference's single-GPU training does).  `max_touched=None`: the ranks agree on
    the padded length first (one scalar MAX all-reduce), so a caller need not know how many views a step can touch.
    -> sorted list of the union's view indices"""
    world = dist.get_world_size(group)
    touched = sorted(set(int(i) for i in touched))
    # The ids are host-side facts (which views this rank's sampler picked): they are exchanged as HOST tensors over a gloo group, so no
    # rank waits for its device queue - the backward kernels keep running under the exchange (round 5 sent them through the device and
    # read them back with .tolist() / .item(): two to three waits per step).
    meta = _meta_group(group)
    if max_touched is None:
        n = torch.tensor([len(touched)], dtype=torch.int64)
        dist.all_reduce(n, op=dist.ReduceOp.MAX, group=meta)
        max_touched = max(1, int(n[0]))
    if len(touched) > max_touched:
        raise ValueError("neuray_amd.parallel: this rank touched %d views, more than max_touched = %d" % (len(touched), max_touched))
    ids = torch.full((max_touched,), -1, dtype=torch.int64)
    ids[:len(touched)] = torch.as_tensor(touched, dtype=torch.int64)
    gathered = [torch.empty_like(ids) for _ in range(world)]
    dist.all_gather(gathered, ids, group=meta)
    union = sorted(set(int(i) for g in gathered for i in g.tolist() if i >= 0))
    if not union:
        return union
    flat = torch.cat([(ray_feats[i].grad if ray_feats[i].grad is not None else torch.zeros_like(ray_feats[i])).reshape(-1).float()
                      for i in union])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= world
    off = 0
    for i in union:
        p = ray_feats[i]
        g = flat[off:off + p.numel()].view_as(p).to(p.dtype)
        p.grad = g.clone() if p.grad is None else p.grad.copy_(g)
        off += p.numel()
    return union
