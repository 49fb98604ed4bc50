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
    """All-gather per-ray tensors [1, n_local, ...] of the ranks' contiguous shards into [1, n_total, ...]."""
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    pad = max(sizes)
    out = {}
    for k, v in local.items():
        t = v.to(torch.float32) if v.dtype == torch.bool else v
        buf = torch.zeros((v.shape[0], pad) + tuple(v.shape[2:]), dtype=t.dtype, device=t.device)
        buf[:, :v.shape[1]] = t
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
        full = torch.cat([p[:, :s] for p, s in zip(parts, sizes)], 1)
        out[k] = full.bool() if v.dtype == torch.bool else full
    return out


def render_image_sharded(renderer, que_imgs_info, ref_imgs_info, group=None):
    """One image split over all ranks of the default (or given) process group; every rank gets the full image."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = que_imgs_info['coords'].shape[1]
    local, _ = render_ray_shard(renderer, que_imgs_info, ref_imgs_info, rank, world)
    return gather_tiles(local, n, rank, world, group)


def allreduce_gradients(parameters, average=True, group=None):
    """Data-parallel training step: sum (or average) the gradients of `parameters` over the ranks with
    ONE flattened all-reduce (the shared nets are ~2.2 M parameters = 8.7 MB: a single bucket; on the MI355X node
    this is RCCL over xGMI via backend 'nccl').  Parameters without a gradient on this rank contribute zeros, so every
    rank ends up with the same gradient for every parameter that received one anywhere."""
    import torch.distributed as dist
    params = [p for p in parameters if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n


def allreduce_scene_feature_gradients(ray_feats, touched, max_touched, average=True, group=None):
    """Fine-tuning mode (SURVEY.md 8(e) caveat): `ray_feats` is the per-view nn.ParameterList of NeuralRayFtRenderer
    (100 views x 5 MB on lego-800) and a step gives gradients to the <= neighbor_view_num + 1 views this rank rendered
    from (`touched`: their indices, e.g. renderer.touched_views).  Instead of reducing all 512 MB (or marking unused
    parameters), the ranks all-gather their touched ids (fixed length `max_touched`, padded with -1) and all-reduce only
    the union's maps in one flattened buffer (<= world * 9 x 5 MB).  Afterwards every rank holds the same gradient for
    every view of the union and `grad is None` for all others, so per-parameter Adam state advances identically on all
    ranks (torch.optim skips parameters without a gradient, as the reference's single-GPU training does).
    -> sorted list of the union's view indices"""
    world = dist.get_world_size(group)
    dev = ray_feats[0].device
    ids = torch.full((max_touched,), -1, dtype=torch.int64, device=dev)
    assert len(touched) <= max_touched
    ids[:len(touched)] = torch.as_tensor(sorted(touched), dtype=torch.int64)
    gathered = [torch.empty_like(ids) for _ in range(world)]
    dist.all_gather(gathered, ids, group=group)
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
