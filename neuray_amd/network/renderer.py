"""Mirror of network/renderer.py's NeuralRayBaseRenderer call surface on top of the HIP render path.

Same cfg keys (base_cfg, network/renderer.py:25-52), same state_dict names for the hot-path modules, same
`render_impl(que_imgs_info, ref_imgs_info, is_train) -> dict` contract (output keys/shapes: SURVEY.md 8(b)).
What differs is underneath: instead of ~700 small PyTorch kernels per ray batch, each pass is three HIP
launches (point kernel, ray kernel, fine-sampling kernel) through include/neuray_hip.h.

Under autograd (training) each pass is a torch.autograd.Function whose backward runs the backward kernels
(network/autograd.py).  Anything the HIP path does not implement raises - it never silently switches to an eager
implementation.

The per-ray methods themselves (`render_by_depth`, `fine_render_impl`, `render_impl`, `predict_self_hit_prob`) live in
network/hip_path.py so that neuray_amd/integrate.py can graft the very same code onto the reference's own classes.
"""
import ctypes
import threading
import weakref

import numpy as np
import torch
import torch.nn as nn

from .. import _lib
from .hip_path import HipRenderPath
from .aggregate_net import name2agg_net
from .dist_decoder import name2dist_decoder
from .encoders import ImageEncoder, name2vis_encoder


class NeuralRayBaseRenderer(HipRenderPath, nn.Module):
    base_cfg = {   # network/renderer.py:25-52
        'vis_encoder_type': 'default', 'vis_encoder_cfg': {},
        'dist_decoder_type': 'mixture_logistics', 'dist_decoder_cfg': {},
        'agg_net_type': 'default', 'agg_net_cfg': {},
        'use_hierarchical_sampling': False, 'fine_agg_net_cfg': {}, 'fine_dist_decoder_cfg': {},
        'fine_depth_sample_num': 64, 'fine_depth_use_all': False,
        'ray_batch_num': 2048, 'depth_sample_num': 64, 'alpha_value_ground_state': -15,
        'use_dr_prediction': False, 'use_nr_color_for_dr': False, 'use_self_hit_prob': False,
        'use_ray_mask': True, 'ray_mask_view_num': 2, 'ray_mask_point_num': 8, 'render_depth': False,
        # not a reference key: also build image_encoder / vis_encoder (network/encoders.py), which makes the state_dict
        # equal to the reference base renderer's and lets render() start from images + initial ray_feats
        'build_encoders': False,
        # not a reference key: 'fp32' = the product library; 'bf16' = the separately built bf16-operand variant of the
        # kernels (libneuray_hip_bf16.so; inference only, never the default)
        'hip_variant': 'fp32',
        # not a reference key: arithmetic of the MLP contractions of the inference point kernel (fp32 library): 'f32' = the fp32 MFMA;
        # 'x3' = every operand split exactly into three bf16 parts on the K = 32 bf16 MFMA, six products per K = 32 with fp32
        # accumulation - each product within 2^-23 of exact (NeurayPointsArgs.arith).  Training always 'f32'.
        'hip_arith': 'f32',
        # not a reference key: the inference packs carry prob_embed.2 folded into its consumers neuray_fc.0 / base_fc.0 (one 32 x 32
        # layer less per (point, view); the same function up to fp32 rounding - neuray_pack_pass_weights_folded)
        'hip_fold_prob_embed': True,
        # not a reference key: inference render() merges ray batches up to this many rays per launch (0 = exactly cfg['ray_batch_num'])
        'hip_min_ray_batch': 65536,
    }

    def __init__(self, cfg):
        super().__init__()
        self.cfg = {**self.base_cfg, **cfg}
        self.dist_decoder = name2dist_decoder[self.cfg['dist_decoder_type']](self.cfg['dist_decoder_cfg'])
        self.agg_net = name2agg_net[self.cfg['agg_net_type']](self.cfg['agg_net_cfg'])
        if self.cfg['use_hierarchical_sampling']:
            self.fine_dist_decoder = name2dist_decoder[self.cfg['dist_decoder_type']](self.cfg['fine_dist_decoder_cfg'])
            self.fine_agg_net = name2agg_net[self.cfg['agg_net_type']](self.cfg['fine_agg_net_cfg'])
        if self.cfg['build_encoders']:       # renderer.py:56-58
            self.vis_encoder = name2vis_encoder[self.cfg['vis_encoder_type']](self.cfg['vis_encoder_cfg'])
            self.image_encoder = ImageEncoder()
            # the reference registers the (unused unless use_dr_prediction) spherical-harmonics regulariser as a buffer
            # (renderer.py:65, sph_solver.py:10-12): kept so that its checkpoints load strictly
            self.sph_fitter = nn.Module()
            self.sph_fitter.register_buffer('regs', torch.tensor([0.0] + [0.001] * 3 + [0.005] * 5 + [0.05] * 7, dtype=torch.float32))
        self._engine = None
        self._engine_test_lib = None     # CPU test-suite hook (emulator build of the kernels)
        self._packed = {}

    def encode_views(self, imgs, ray_feats):
        """renderer.py:229-235: image_encoder, then vis_encoder on its output -> (img_feats, refined ray_feats).
        (Capturing the two stacks' forward + backward as HIP graphs - torch.cuda.make_graphed_callables - was measured and is not
        done: the replayed step is slower than launching the ~900 kernels one by one)"""
        feats = self.image_encoder(imgs)
        return feats, self.vis_encoder(ray_feats, feats)

    def _early_query(self, que_imgs_info):
        """The query constants need K^-1 from the host's LAPACK (engine.prepare_query: the bit-exact geometry contract is pinned to the
        CPU result): a 36-byte device -> host copy, i.e. a wait for everything queued so far.  Taken BEFORE the init net and the encoders
        are queued it waits for next to nothing; taken where the per-ray path first needs it, it drained 25 ms of queued convolutions per
        training step and left the GPU idle behind them (profiles/r04_n_gen_host_profile.txt: 8.6 ms per step).  Cached in the dict."""
        coords = que_imgs_info.get('coords')
        if coords is not None and coords.is_cuda and coords.shape[0] == 1 and 'Ks' in que_imgs_info:
            self._query(self.engine(coords.device), que_imgs_info)

    def render(self, que_imgs_info, ref_imgs_info, is_train):
        """network/renderer.py:228-254.  ref_imgs_info either carries 'img_feats' and the encoded 'ray_feats' already, or
        (cfg['build_encoders']) 'imgs' and the initial 'ray_feats', which go through image_encoder / vis_encoder first."""
        self._early_query(que_imgs_info)
        if 'img_feats' not in ref_imgs_info:
            # renderer.py:229-235: encode the reference images, refine the initial ray_feats with them
            if not self.cfg.get('build_encoders', hasattr(self, 'image_encoder')):      # (the reference's cfg has no such key: its classes always build them)
                raise NotImplementedError("neuray_amd: render() needs ref_imgs_info['img_feats'] (and the encoded 'ray_feats'), "
                                          "or a renderer built with cfg['build_encoders'] = True")
            self_hit = is_train and self.cfg['use_self_hit_prob']
            if self_hit and que_imgs_info['imgs'].shape[1:] == ref_imgs_info['imgs'].shape[1:]:
                # the query view of the self-hit loss rides along as one more sample of the same batch: the stacks are per-sample
                # (convolutions + InstanceNorm), so this is the reference's two encoder passes (renderer.py:229-235) in half the
                # launches, forward and backward - a training step is host-bound on exactly those
                n = ref_imgs_info['imgs'].shape[0]
                joint = ref_imgs_info.get('_neuray_joint')
                if joint is None or que_imgs_info.get('_neuray_joint') is not joint:      # (slice_imgs_info hands the parts over as views of one tensor)
                    joint = (torch.cat([ref_imgs_info['imgs'], que_imgs_info['imgs']], 0),
                             torch.cat([ref_imgs_info['ray_feats'], que_imgs_info['ray_feats']], 0))
                # consumed: the private entry must not outlive this call - a host class that only pops the reference's keys (the reference's
                # own train_step, renderer.py:521-543, under patch_ft_host) would otherwise return it inside render_outputs['que_imgs_info']
                # and keep ~115 MB alive per step
                ref_imgs_info.pop('_neuray_joint', None)
                que_imgs_info.pop('_neuray_joint', None)
                feats, rays = self.encode_views(*joint)
                # (split, not three slices: one concatenation per tensor in the backward instead of a zero-fill + copy per slice)
                ref_imgs_info['img_feats'] = feats.split([n, feats.shape[0] - n])[0]
                ref_imgs_info['ray_feats'], que_imgs_info['ray_feats'] = rays.split([n, rays.shape[0] - n])
            else:
                ref_imgs_info['img_feats'], ref_imgs_info['ray_feats'] = self.encode_views(ref_imgs_info['imgs'], ref_imgs_info['ray_feats'])
                if self_hit:
                    que_imgs_info['ray_feats'] = self.encode_views(que_imgs_info['imgs'], que_imgs_info['ray_feats'])[1]
        if 'ray_feats' not in ref_imgs_info:
            raise NotImplementedError("neuray_amd: render() needs ref_imgs_info['ray_feats']")
        ray_batch_num = self.cfg['ray_batch_num']
        if not is_train and not torch.is_grad_enabled():
            # `ray_batch_num` bounds the reference's activation memory (1.7 GB per 4096 rays); here a batch costs 80 B per
            # sample point, results do not depend on the batching bit for bit, and a 4096-ray launch leaves a third of the
            # machine idle in its tail (2.56 vs 2.79 M rays/s): inference uses at least cfg['hip_min_ray_batch'] rays per launch
            # Direct rendering (use_dr_prediction) adds the per-(point, view) record: rn * dn * rfn * 16 floats per pass (1.1 GB at 32768
            # rays x 64 samples x 8 views), so there the merged batch is bounded to ~1 GiB of record and never below the configured size.
            merged = int(self.cfg.get('hip_min_ray_batch', 65536))
            if self.cfg.get('use_dr_prediction', False):
                dn_max = max(self.cfg['depth_sample_num'], self.cfg['fine_depth_sample_num'] if self.cfg['use_hierarchical_sampling'] else 0)
                merged = min(merged, max(1, (1 << 30) // (dn_max * ref_imgs_info['imgs'].shape[0] * 16 * 4)))
            ray_batch_num = max(ray_batch_num, merged)
        coords = que_imgs_info['coords']
        ray_num = coords.shape[1]
        render_info_all = {}
        for ray_id in range(0, ray_num, ray_batch_num):
            que_imgs_info['coords'] = coords[:, ray_id:ray_id + ray_batch_num]
            render_info = self.render_impl(que_imgs_info, ref_imgs_info, is_train)
            for k, v in render_info.items():
                if is_train or (not k.startswith('hit_prob')):
                    render_info_all.setdefault(k, []).append(v)
        que_imgs_info['coords'] = coords
        return {k: torch.cat(v, 1) for k, v in render_info_all.items()}


# ---- generalisation renderer ----------------------------------------------------------------------------
from .init_net import name2init_net      # noqa: E402  'depth'  and 'cost_volume' (f-3)


class NeuralRayGenRenderer(NeuralRayBaseRenderer):
    """network/renderer.py:256-326: init_net -> encoders -> render, plus the mean-depth readout of the depth loss."""
    default_cfg = {'init_net_type': 'depth', 'init_net_cfg': {}, 'use_depth_loss': False, 'depth_loss_coords_num': 8192}

    def __init__(self, cfg, init_net=None):
        super().__init__({**self.default_cfg, **cfg, 'build_encoders': True})
        if init_net is None and self.cfg['init_net_type'] in name2init_net:
            init_net = name2init_net[self.cfg['init_net_type']](self.cfg['init_net_cfg'])
        if init_net is not None:
            self.init_net = init_net
        else:
            self.init_net = None      # plain attribute: no parameters, not in the state_dict

    def render_call(self, que_imgs_info, ref_imgs_info, is_train, src_imgs_info=None):
        self._early_query(que_imgs_info)
        if self.init_net is not None:
            ref_imgs_info['ray_feats'] = self.init_net(ref_imgs_info, src_imgs_info, is_train)
        elif 'ray_feats' not in ref_imgs_info:
            raise NotImplementedError(
                "neuray_amd: no init_net registered for init_net_type=%r (pass one as init_net=) and no initial "
                "ref_imgs_info['ray_feats'] was handed over" % self.cfg['init_net_type'])
        return self.render(que_imgs_info, ref_imgs_info, is_train)

    def gen_depth_loss_coords(self, h, w, device):
        """renderer.py:272-278 (quirk kept: the pairs are (row, col) although the gather reads them as (x, y))."""
        num = self.cfg['depth_loss_coords_num']
        slot = self.__dict__.pop('_depth_coords_slot', None)
        if slot is not None:
            slot['thread'].join()
        if slot is not None and slot.get('n') == h * w and 'perm' in slot and torch.equal(torch.get_rng_state(), slot['state0']):
            pick = slot['perm'][:num]                 # drawn ahead of time by _prefetch_depth_coords, at this draw's place in the stream
            torch.set_rng_state(slot['state1'])       # (the global generator moves on as the in-line draw would have moved it)
        else:
            pick = torch.randperm(h * w)[:num]
        pairs = torch.stack([pick // w, pick % w], -1)
        # (a copy from pageable memory blocks the host until everything queued so far has run - 5 ms of a training step behind the per-ray
        # kernels; pinned staging + an asynchronous copy does not)
        return pairs.pin_memory().to(device, non_blocking=True) if torch.device(device).type == 'cuda' else pairs.to(device)

    def _prefetch_depth_coords(self, n):
        """torch.randperm(h * w) over the pixels of a reference view costs 5 ms of host time per generalisation step (a quarter of a
        million draws; the step is host-bound) and releases the interpreter lock: it runs on a worker thread, started by
        render_impl right after the step's only other draw from the CPU generator (the fine-sampling uniforms, render_ops.py:205) - the
        reference's order, so a seeded run consumes the generator identically - while the main thread queues the per-ray kernels."""
        # The worker draws from a PRIVATE generator started at the global generator's present state; the draw is adopted at the join
        # (gen_depth_loss_coords) only if the global generator is still exactly there - then its state is advanced to where the private one
        # ended, which is what the in-line torch.randperm would have left - and discarded for the in-line draw otherwise (a custom
        # init net, a hook or another thread drew in between).  The main thread's own draws never race with the worker.
        slot = {'n': n, 'state0': torch.get_rng_state()}

        def work():
            g = torch.Generator()
            g.set_state(slot['state0'])
            slot['perm'] = torch.randperm(n, generator=g)
            slot['state1'] = g.get_state()
        slot['thread'] = threading.Thread(target=work, name='neuray-depth-coords', daemon=True)
        self.__dict__['_depth_coords_slot'] = slot
        slot['thread'].start()

    def predict_mean_for_depth_loss(self, ref_imgs_info):
        """renderer.py:280-316: decoded mixture means of every reference view at random pixels of that view."""
        from . import render_ops
        ray_feats = ref_imgs_info['ray_feats']
        rfn, _, h, w = ref_imgs_info['imgs'].shape
        coords = self.gen_depth_loss_coords(h, w, ray_feats.device)[None].repeat(rfn, 1, 1)
        ones = torch.ones(coords.shape[:2], dtype=torch.float32, device=ray_feats.device)
        feats = render_ops.interpolate_feature_map(ray_feats, coords.float(), ones, h, w)      # rfn,pn,f
        mean = self.dist_decoder.predict_mean(feats)
        outputs = {'depth_mean': mean[..., 0], 'depth_coords': coords, 'depth_mean_2': mean[..., 1]}
        if self.cfg['use_hierarchical_sampling']:
            fine = self.fine_dist_decoder.predict_mean(feats)
            outputs['depth_mean_fine'], outputs['depth_mean_fine_2'] = fine[..., 0], fine[..., 1]
        return outputs

    def forward(self, data):
        ref_imgs_info, que_imgs_info = data['ref_imgs_info'].copy(), data['que_imgs_info'].copy()
        is_train = 'eval' not in data
        src_imgs_info = data['src_imgs_info'].copy() if 'src_imgs_info' in data else None
        depth_readout = (self.cfg['use_depth_loss'] and 'true_depth' in ref_imgs_info) or not is_train
        coords = que_imgs_info['coords']
        if (depth_readout and is_train and self.cfg.get('hip_prefetch_depth_coords', True) and coords.shape[0] == 1
                and coords.shape[1] <= self.cfg['ray_batch_num']):          # (one render_impl call: one place in the stream to start at)
            _, _, h, w = ref_imgs_info['imgs'].shape
            self.__dict__['_neuray_after_fine_draw'] = lambda: self._prefetch_depth_coords(h * w)
        try:
            outputs = self.render_call(que_imgs_info, ref_imgs_info, is_train, src_imgs_info)
        except BaseException:
            slot = self.__dict__.pop('_depth_coords_slot', None)       # a permutation drawn for a step that failed is not kept for the next one
            if slot is not None:
                slot['thread'].join()
            raise
        finally:
            self.__dict__.pop('_neuray_after_fine_draw', None)
        if depth_readout:
            outputs.update(self.predict_mean_for_depth_loss(ref_imgs_info))
        else:
            slot = self.__dict__.pop('_depth_coords_slot', None)
            if slot is not None:
                slot['thread'].join()
        return outputs


# ---- per-scene fine-tuning renderer --------------------------------------------------------------------
def camera_centres(poses):
    """[n,3,4] world->camera poses -> [n,3] centres  -R^T t"""
    poses = np.asarray(poses, np.float64)
    return -np.einsum('nji,nj->ni', poses[:, :, :3], poses[:, :, 3])


def nearest_view_table(que_poses, ref_poses):
    """utils/view_select.py:7-16: for every query camera the reference cameras ordered by centre distance -> [qn,rfn] int"""
    d = np.linalg.norm(camera_centres(ref_poses)[None] - camera_centres(que_poses)[:, None], 2, 2)
    return np.argsort(d, 1)


def pixel_lists(fg_mask):
    """row-major flat indices of the foreground and of the background pixels of a mask (the order np.nonzero lists them in) and the
    image width: what sample_train_coords needs of a view, 2.5 MB per 800 x 800 view, constant over a fine-tuning run"""
    fg_mask = np.asarray(fg_mask, bool)
    return np.flatnonzero(fg_mask).astype(np.int32), np.flatnonzero(~fg_mask).astype(np.int32), fg_mask.shape[1]


_HOST_LIB = []


def _host_lib():
    """libneuray_hip.so for its host-only entry points (they need no GPU), or None where it has not been built"""
    if not _HOST_LIB:
        try:
            _HOST_LIB.append(_lib.load())
        except (_lib.NeurayLibError, OSError, AttributeError):
            _HOST_LIB.append(None)
    return _HOST_LIB[0]


def shuffle_like_numpy(arr, rs=np.random):
    """rs.shuffle(arr) for a large 1-D array - the same permutation, the same generator state afterwards - through
    neuray_mt19937_shuffle: numpy's Fisher-Yates on its MT19937 state, outside the interpreter lock (numpy holds it for the 4 ms of a
    640 000-entry shuffle, which is what keeps the next step's draws from overlapping this step's launches).  `rs`: the np.random
    module (the global generator) or a np.random.RandomState."""
    lib = _host_lib()
    if lib is None or arr.ndim != 1 or arr.itemsize not in (4, 8) or not arr.flags.c_contiguous or arr.shape[0] < 4096:
        rs.shuffle(arr)
        return
    st = rs.get_state()
    if not _is_mt19937_state(st):
        rs.shuffle(arr)           # another bit generator behind the legacy interface (np.random.set_bit_generator): numpy's own shuffle
        return
    key = np.array(st[1], dtype=np.uint32)
    pos = ctypes.c_int(int(st[2]))
    _lib.check(lib, lib.neuray_mt19937_shuffle(key.ctypes.data, ctypes.byref(pos), arr.ctypes.data, arr.shape[0], arr.itemsize))
    rs.set_state((st[0], key, pos.value, st[3], st[4]))


def sample_train_coords(fg_mask, ray_num, foreground_ratio, lists=None, rs=np.random):
    """utils/base_utils.py:585-603: `ray_num` pixel coordinates (x, y) of one image, at least
    int(ray_num * foreground_ratio) of them drawn from the foreground mask (when it has that many), the rest from the
    remaining pixels.  The two np.random.shuffle draws over the full pixel lists are the reference's (same lengths, same order,
    so the same generator stream and the same pixels); what the reference does around them - two np.nonzero passes, gathering
    and concatenating the whole shuffled lists to keep 512 rows - is replaced by cached lists (`lists` = pixel_lists(fg_mask))
    and by indexing only the rows that are returned (19 -> 8 ms per step at 800 x 800).  `rs`: the generator the draws come from
    (the global one, or the private copy of it a prefetching NeuralRayFtRenderer works on)."""
    want_fg = int(ray_num * foreground_ratio)
    fg, bg, w = lists if lists is not None else pixel_lists(fg_mask)
    order = np.arange(fg.shape[0], dtype=np.int32)                         # (the permutation does not depend on the item type)
    shuffle_like_numpy(order, rs)
    flat = fg[order[:want_fg]]
    if want_fg < ray_num:
        nbg = bg.shape[0]
        order2 = np.arange(nbg + max(fg.shape[0] - want_fg, 0), dtype=np.int32)     # the pool: background, then the unpicked foreground
        shuffle_like_numpy(order2, rs)
        idx = order2[:ray_num - want_fg]
        rest = np.empty(idx.shape[0], np.int32)
        from_bg = idx < nbg
        rest[from_bg] = bg[idx[from_bg]]
        rest[~from_bg] = fg[order[want_fg + idx[~from_bg] - nbg]]
        flat = np.concatenate([flat, rest], 0)
    return np.stack([flat % w, flat // w], 1).astype(np.float32)


def _same_generator_state(a, b):
    return _is_mt19937_state(a) and _is_mt19937_state(b) and a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and np.array_equal(a[1], b[1])


# speculative draws of the NEXT training step, per renderer (kept outside the module so that it stays picklable / deep-copyable)
_PREFETCH = weakref.WeakKeyDictionary()


def _train_coords(ft, val_idx, rs):
    """[1, train_ray_num, 2] training rays of reference view `val_idx` (sample_train_coords on its cached pixel lists)"""
    cached = ft.__dict__.setdefault('_pixel_lists', {})
    lists = cached.get(int(val_idx))
    if lists is None:
        lists = cached[int(val_idx)] = pixel_lists(ft.ref_imgs_info['masks'][val_idx, 0].cpu().numpy() > 0)
    return sample_train_coords(None, ft.cfg['train_ray_num'], ft.cfg['foreground_ratio'], lists, rs).reshape(1, -1, 2)


def _sampling_cfg(ft):
    return tuple(ft.cfg[k] for k in ('include_self_prob', 'neighbor_view_num', 'neighbor_pool_ratio', 'train_ray_num', 'foreground_ratio'))


def _draw_train_step(ft, rs):
    """renderer.py:521-529 + base_utils.py:585-603: every np.random draw of one training step, in the reference's order, from `rs`
    -> (query view, its reference views, ray coordinates)"""
    que_i = rs.randint(0, len(ft.ref_ids))
    ref_idx = ft.ref_dist_idx[que_i]
    if rs.random_sample() > ft.cfg['include_self_prob']:
        ref_idx = ref_idx[1:]
    ref_idx = ref_idx[:ft.cfg['neighbor_view_num'] * ft.cfg['neighbor_pool_ratio']].copy()
    rs.shuffle(ref_idx)
    ref_idx = ref_idx[:ft.cfg['neighbor_view_num']]
    return que_i, ref_idx, _train_coords(ft, que_i, rs)


def _is_mt19937_state(st):
    return isinstance(st, tuple) and len(st) == 5 and st[0] == 'MT19937' and len(st[1]) == 624


def _prefetch_start(ft):
    start, cfg = np.random.get_state(), _sampling_cfg(ft)
    if not _is_mt19937_state(start):       # another bit generator behind np.random (np.random.set_bit_generator): no speculation, the
        return                             # step draws from the global generator as the reference does
    slot = {'start': start, 'cfg': cfg}

    def work():
        try:
            rs = np.random.RandomState()
            rs.set_state(start)
            slot['drawn'] = _draw_train_step(ft_ref(), rs)
            slot['end'] = rs.get_state()
        except Exception:                                    # (a vanished renderer, a changed scene: the next step draws afresh)
            slot.pop('drawn', None)

    ft_ref = weakref.ref(ft)
    slot['thread'] = threading.Thread(target=work, name='neuray-ray-sampler', daemon=True)
    _PREFETCH[ft] = slot
    slot['thread'].start()


def _prefetch_take(ft):
    slot = _PREFETCH.pop(ft, None)
    if slot is None:
        return None
    slot['thread'].join()
    if 'drawn' not in slot or 'end' not in slot or slot['cfg'] != _sampling_cfg(ft):
        return None
    if not _same_generator_state(np.random.get_state(), slot['start']):
        return None
    np.random.set_state(slot['end'])
    return slot['drawn']


def pad_views(info, interval):
    """utils/imgs_info.py:60-75: reflect-pad imgs / depth / masks (/ true_depth) at the bottom and right to multiples of
    `interval` (the encoders halve the resolution four times)."""
    h, w = info['imgs'].shape[-2:]
    ph, pw = (-h) % interval, (-w) % interval
    if ph or pw:
        for k in ('imgs', 'depth', 'masks', 'true_depth'):
            if k in info:
                info[k] = np.pad(np.asarray(info[k]), ((0, 0), (0, 0), (0, ph), (0, pw)), 'reflect')
    return info


def _as_torch(info):
    return {k: torch.from_numpy(v) if isinstance(v, np.ndarray) else v for k, v in info.items()}


def _take(info, idx):
    """the entries of an imgs_info at the view indices `idx` (a handful of Python ints).  No index tensor: uploading one from
    pageable memory is a synchronous copy, i.e. a wait for everything queued on the device - one view is a slice (no kernel), several
    are stacked (one kernel per entry)"""
    idx = [int(i) for i in np.asarray(idx).reshape(-1)]
    if len(idx) == 1:
        return {k: v[idx[0]:idx[0] + 1] for k, v in info.items()}
    return {k: torch.stack([v[i] for i in idx], 0) for k, v in info.items()}


def _upload(array, device):
    """numpy -> device tensor without stalling the launch queue (pinned staging + asynchronous copy on a GPU)"""
    t = torch.from_numpy(array)
    if device.type == 'cuda':
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)


class NeuralRayFtRenderer(NeuralRayBaseRenderer):
    """network/renderer.py:328-545: the per-scene renderer.  Every reference view owns a learnable visibility feature
    map (`ray_feats`, an nn.ParameterList of [1,32,h/4,w/4]); a step renders rays of one view from its neighbours.

    The reference constructor reads the scene through its dataset layer (`parse_database_name`, `build_imgs_info`:
    file I/O, SURVEY.md 8(f) f-4, not built) and initialises `ray_feats` by running a generalisation checkpoint's
    init_net (f-2/f-3).  Here both enter as data:
      scene = {'ref_imgs_info': imgs_info of all reference views (imgs, masks, depth, poses, Ks, depth_range; numpy or
               torch, host memory), 'val_imgs_info': the same for the validation views (optional), 'database': anything}
      init_ray_feats = per-view initial maps (list of [1,dim,fh,fw]) or None for the reference's "init from scratch"
    and `load_gen_state_dict` copies the shared networks from a generalisation state_dict (renderer.py:466-475)."""
    default_cfg = {
        'database_name': None, 'database_split': 'val_all', 'ref_pad_interval': 16, 'use_consistent_depth_range': True,
        'gen_cfg': None, 'use_validation': True, 'validate_initialization': True, 'init_view_num': 8, 'init_src_view_num': 3,
        'include_self_prob': 0.01, 'neighbor_view_num': 8, 'neighbor_pool_ratio': 2, 'train_ray_num': 512,
        'foreground_ratio': 0.5, 'ray_feats_res': [200, 200], 'ray_feats_dim': 32,
    }

    def __init__(self, cfg, scene=None, init_ray_feats=None):
        super().__init__({**self.default_cfg, **cfg, 'build_encoders': True})
        if scene is None:
            raise NotImplementedError("neuray_amd: NeuralRayFtRenderer needs scene={'ref_imgs_info': ..., ['val_imgs_info': ...]} - "
                                      "the dataset layer behind cfg['database_name'] is host I/O outside the render path")
        self.database = scene.get('database')
        ref = pad_views({k: np.asarray(v) if not torch.is_tensor(v) else v.numpy() for k, v in scene['ref_imgs_info'].items()},
                        self.cfg['ref_pad_interval'])
        if self.cfg['use_consistent_depth_range']:
            ref['depth_range'] = np.array(ref['depth_range'], np.float32)
            ref['depth_range'][:, 0], ref['depth_range'][:, 1] = ref['depth_range'].min(), ref['depth_range'].max()
        self.ref_imgs_info = _as_torch(ref)
        self.ref_ids = np.arange(ref['imgs'].shape[0])
        self.ref_dist_idx = nearest_view_table(ref['poses'], ref['poses'])          # rfn,rfn (column 0: the view itself)
        self.val_imgs_info = None
        if self.cfg['use_validation'] and 'val_imgs_info' in scene:
            val = {k: np.asarray(v) if not torch.is_tensor(v) else v.numpy() for k, v in scene['val_imgs_info'].items()}
            self.val_imgs_info = _as_torch(val)
            self.val_dist_idx = nearest_view_table(val['poses'], ref['poses'])
            self.val_num = val['imgs'].shape[0]
        self.ray_feats = nn.ParameterList()
        n = len(self.ref_ids)
        if init_ray_feats is None:          # renderer.py:476-482
            fh, fw = self.cfg['ray_feats_res']
            init_ray_feats = [torch.randn(1, self.cfg['ray_feats_dim'], fh, fw) for _ in range(n)]
        assert len(init_ray_feats) == n
        for t in init_ray_feats:
            self.ray_feats.append(nn.Parameter(torch.as_tensor(t).detach().clone().float()))
        self.touched_views = []             # reference-view indices whose ray_feats the last train_step used
        self._scene_dev, self._enc_cache, self._pixel_lists = {}, {}, {}

    def load_gen_state_dict(self, state_dict):
        """renderer.py:466-475: take the shared networks of a generalisation model (everything but its init_net)."""
        own = self.state_dict()
        picked = {k: v for k, v in state_dict.items() if k in own and not k.startswith('ray_feats.')}
        missing = [k for k in own if not k.startswith('ray_feats.') and k not in picked]
        if missing:
            raise KeyError("generalisation state_dict lacks %s" % missing[:4])
        self.load_state_dict(picked, strict=False)

    # eval: hand render() the cached per-view encoder outputs (`img_feats` + encoded `ray_feats`).  False where render() encodes
    # unconditionally - the reference's own class with these host-path methods grafted on (integrate.patch_ft_host)
    cache_encoded_views = True

    def _device(self):
        return self.ray_feats[0].device

    def _resident(self, which):
        """The scene's views stay resident in HBM (a 100-view 800x800 scene is 0.8 GB of 288): moved once per device
        instead of the reference's per-step `to_cuda(imgs_info_slice(...))` (renderer.py:487)."""
        dev = self._device()
        scene_dev = self.__dict__.setdefault('_scene_dev', {})     # (state is created on first use: these methods are also grafted onto the reference's class)
        hit = scene_dev.get(which)
        if hit is None or hit[0] != dev:
            src = self.ref_imgs_info if which == 'ref' else self.val_imgs_info
            src = {k: v for k, v in src.items() if torch.is_tensor(v)}
            # K^-1 of every view, evaluated where engine.prepare_query would evaluate it (host LAPACK, the same call on the same bytes),
            # once: handed over as `Ks_inv`, a step needs no device -> host round trip for its query view
            from ..engine import host_inverse
            src['Ks_inv'] = torch.cat([host_inverse(src['Ks'][i:i + 1]) for i in range(src['Ks'].shape[0])], 0)
            scene_dev[which] = (dev, {k: v.to(dev) for k, v in src.items()})
        return scene_dev[which][1]

    def _encoded(self, ref_idx):
        """eval only: per-view encoder outputs, reused while the encoders and that view's ray_feats are unchanged
        (neighbouring poses share most of their reference views) -> (img_feats, ray_feats)"""
        enc_stamp = tuple(p._version for p in self.image_encoder.parameters()) + tuple(p._version for p in self.vis_encoder.parameters())
        imgs = self._resident('ref')['imgs']
        out, cache = [], self.__dict__.setdefault('_enc_cache', {})
        for i in (int(i) for i in ref_idx):
            stamp = (enc_stamp, self.ray_feats[i]._version, self._device())
            hit = cache.get(i)
            if hit is None or hit[0] != stamp:
                with torch.no_grad():
                    f = self.image_encoder(imgs[i:i + 1])
                    hit = (stamp, f, self.vis_encoder(self.ray_feats[i], f))
                cache[i] = hit
            out.append(hit)
        return torch.cat([h[1] for h in out], 0), torch.cat([h[2] for h in out], 0)

    def _ref_views(self, ref_idx, is_train):
        ref_imgs_info = _take(self._resident('ref'), ref_idx)
        if is_train or not self.cache_encoded_views:
            ref_imgs_info['ray_feats'] = torch.cat([self.ray_feats[int(i)] for i in ref_idx], 0)
        else:
            ref_imgs_info['img_feats'], ref_imgs_info['ray_feats'] = self._encoded(ref_idx)
        return ref_imgs_info

    def slice_imgs_info(self, ref_idx, val_idx, is_train, coords=None):
        """renderer.py:484-507.  coords: the training rays of `val_idx` when they have been drawn already (train_step's prefetch)."""
        if is_train and self.cfg['use_self_hit_prob']:
            # the query view rides along with the reference views through the encoders (render()): gather the nine views and their nine
            # ray_feats maps ONCE and hand out the two parts as views - one 69 MB stack and one 46 MB concatenation per step instead of
            # two of each (the second pair was render()'s torch.cat of the parts)
            n = len(ref_idx)
            joint = _take(self._resident('ref'), list(ref_idx) + [val_idx])
            maps = torch.cat([self.ray_feats[int(i)] for i in list(ref_idx) + [val_idx]], 0)
            ref_imgs_info, que = {k: v[:n] for k, v in joint.items()}, {k: v[n:] for k, v in joint.items()}
            ref_imgs_info['ray_feats'] = maps[:n]
            ref_imgs_info['_neuray_joint'] = que['_neuray_joint'] = (joint['imgs'], maps)
        else:
            ref_imgs_info = self._ref_views(ref_idx, is_train)
            que = _take(self._resident('ref' if is_train else 'val'), [val_idx])
        if is_train:
            if coords is None:
                coords = _train_coords(self, val_idx, np.random)
        else:
            hn, wn = que['imgs'].shape[-2:]
            coords = np.stack(np.meshgrid(np.arange(wn), np.arange(hn)), -1).reshape(1, -1, 2).astype(np.float32)
        que['coords'] = _upload(coords, self._device())
        if is_train and self.cfg['use_self_hit_prob']:
            que['ray_feats'] = maps[n:]
        return ref_imgs_info, que

    def validate_step(self, val_idx):
        """renderer.py:509-519"""
        ref_idx = self.val_dist_idx[val_idx][:self.cfg['neighbor_view_num']]
        ref_imgs_info, que_imgs_info = self.slice_imgs_info(ref_idx, val_idx, False)
        with torch.no_grad():
            outputs = self.render(que_imgs_info, ref_imgs_info, False)
        for k in ('ray_feats', 'img_feats', '_neuray_views'):
            ref_imgs_info.pop(k, None)
        outputs.update({'ref_imgs_info': ref_imgs_info, 'que_imgs_info': que_imgs_info})
        return outputs

    def train_step(self):
        """renderer.py:521-543: a random view as the query, 8 of its 16 nearest other views (itself with 1 % probability).

        The step's np.random draws (view, neighbours, the ray sampler's two shuffles over the 640 000 pixels of the query image: 8 ms
        of the host's 22 ms per step) are the reference's, in its order.  With cfg['hip_prefetch_ray_sampling'] (default
        on) the NEXT step's draws are made speculatively on a worker thread, on a private copy of the global generator, while this
        step's kernels are being queued; the next call adopts them - and moves the global generator to where those draws left the copy -
        only if the global generator is still exactly where the speculation started (nobody else drew from np.random or re-seeded it in
        between) and the sampling cfg is unchanged; otherwise it draws afresh.  Either way the stream and the rays are the reference's."""
        drawn = _prefetch_take(self)
        if drawn is None:
            drawn = _draw_train_step(self, np.random)
        que_i, ref_idx, coords = drawn
        if self.cfg.get('hip_prefetch_ray_sampling', True):
            _prefetch_start(self)
        ref_imgs_info, que_imgs_info = self.slice_imgs_info(ref_idx, que_i, True, coords=coords)
        self.touched_views = sorted(set(int(i) for i in ref_idx) | ({int(que_i)} if self.cfg['use_self_hit_prob'] else set()))
        outputs = self.render(que_imgs_info.copy(), ref_imgs_info.copy(), True)
        for k in ('ray_feats', 'img_feats', '_neuray_qconst', '_neuray_qconst_entry', '_neuray_joint'):
            que_imgs_info.pop(k, None)
        outputs['que_imgs_info'] = que_imgs_info
        return outputs

    def render_pose(self, render_imgs_info):
        """renderer.py:545-555: render an arbitrary pose from its nearest reference views (the nearest one is skipped,
        as in the reference's `select_working_views(..., exclude_self=True)`)."""
        order = nearest_view_table(render_imgs_info['poses'].cpu().numpy(), self.ref_imgs_info['poses'].numpy())[0]
        ref_idx = order[1:self.cfg['neighbor_view_num'] + 1]
        with torch.no_grad():
            return self.render(render_imgs_info, self._ref_views(ref_idx, False), False)

    def forward(self, data):
        if 'eval' not in data:
            return self.train_step()
        return self.validate_step(data['index'])


name2network = {'neuray_base': NeuralRayBaseRenderer, 'neuray_gen': NeuralRayGenRenderer, 'neuray_ft': NeuralRayFtRenderer}
