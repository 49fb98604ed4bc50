// C ABI of libneuray_hip.so (declared in include/neuray_hip.h).  Host-side launch logic only; the device
// code lives in nr_kernels.h / nr_device.h.  Built with hipcc --offload-arch=gfx950 (product) or, for the
// CPU test emulator, g++ -DNEURAY_EMU (tests/emu/build_emu.py).
#include "nr_kernels.h"
#include "nr_kernels_bwd.h"
#include "nr_kernels_dr.h"
#include "nr_kernels_norm.h"
#include "nr_kernels_conv3d.h"
#include "nr_kernels_conv2d.h"
// the plain bf16-operand build is inference only; the fp32 build and the split build (hi + lo bf16 operands: fp32-grade products)
// carry the training path
#if defined(NR_BF16_QUADS) && !defined(NR_BF16_SPLIT)
#define NR_INFERENCE_ONLY 1
#else
#include "nr_kernels_bwd2.h"
#endif
#include "nr_pack.h"
#include "../../include/neuray_hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>

namespace {

thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
    return 0;
}

int grid_for(long long work_items, int per_block, int max_blocks) {
    long long b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

static_assert(NEURAY_POINT_REC == nr::kPointRec, "abi");
static_assert(NEURAY_VIEW_CONST == nr::kViewConst, "abi");
static_assert(NEURAY_QUERY_CONST == nr::kQueryConst, "abi");
static_assert(NEURAY_PASS_TENSORS == nr::T_COUNT, "abi");
static_assert(NEURAY_DBG_FIELDS == nr::kDbgFields, "abi");
static_assert(NEURAY_RAY_ATT_SAVE == nr::kRayAttSave, "abi");
static_assert(NEURAY_MAX_SAMPLES == nr::kMaxSamples, "abi");

template <int NT, int VPW, bool HAS_VIS, int OWN, int MINW, bool SAVE = false, int AR = nr::AR_F32>
int launch_points_own(const nr::PointParams& p, void* stream) {
    const int npts = p.rn * p.dn;
    const int nwaves = (p.rfn + VPW - 1) / VPW;
    const size_t smem = nr::point_smem_bytes<NT>(nwaves, AR);
    if (smem > 160 * 1024) return fail("neuray_render_points: %zu bytes of LDS needed (rfn=%d)", smem, p.rfn);
    // persistent-style grid: enough workgroups to fill the 768 resident slots (3 per CU) twenty times over, grid-stride beyond.  Measured on
    // the 800 x 800 workload (131 072 tiles per coarse launch), same box: 768 workgroups 2.30 M rays/s, 1536 2.35, 3072 / 3840 2.38,
    // 4096 2.42, 6144 2.37, 8192 ... 32768 2.44 - finer-grained balancing wins over fewer prologues, and counts that divide the tiles
    // evenly beat those that do not.  Round 4 (tiles of 16 neighbouring rays): 4096 5.59 ms per launch, 8192 5.49, 16384 5.47, 32768 5.55.
#ifndef NR_POINT_GRID
#define NR_POINT_GRID (256 * 64)
#endif
    static const int grid_env = [] { const char* e = getenv("NEURAY_POINT_GRID"); return e ? atoi(e) : 0; }();     // A/B hook (tools/ab_forward.py)
    int grid = grid_for(npts, 16 * NT, grid_env > 0 ? grid_env : NR_POINT_GRID);
#ifndef NR_POINT_MIN_TILES
#define NR_POINT_MIN_TILES 2       // a workgroup's prologue (constants, first weight phase) wants at least this many tiles behind it
#endif
    if (!SAVE && npts / 16 >= 4096 && grid > npts / (16 * NR_POINT_MIN_TILES)) grid = npts / (16 * NR_POINT_MIN_TILES);
    grid = (grid + 7) / 8 * 8;                         // the XCD-aware tile map needs a multiple of 8
    const int threads = 64 * nwaves;
    // __launch_bounds__(1024) caps the kernel at 128 VGPRs so that 4 waves share a SIMD (DESIGN.md "occupancy")
    auto k = nr::points_kernel<NT, VPW, HAS_VIS, OWN, 1024 / VPW, MINW, SAVE, false, AR>;
    if constexpr (!SAVE) {
        if (p.dbg) k = nr::points_kernel<NT, VPW, HAS_VIS, OWN, 1024 / VPW, MINW, false, true, AR>;     // the per-view record: its own instantiation
    } else if (p.dbg) {
        return fail("neuray_render_points: dbg_dev and saved_dev together are not built (run the pass twice)");
    }
#ifndef NEURAY_EMU
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
#endif
    NR_LAUNCH(k, dim3(grid), dim3(threads), smem, stream, p);
    return check_launch("neuray_render_points");
}

template <int NT, int VPW, bool HAS_VIS, int MINW, int AR = nr::AR_F32>
int launch_points(const nr::PointParams& p, void* stream) {
    const int nwaves = (p.rfn + VPW - 1) / VPW;
    if (nwaves >= 4) return launch_points_own<NT, VPW, HAS_VIS, 1, MINW, false, AR>(p, stream);
    if (nwaves >= 2) return launch_points_own<NT, VPW, HAS_VIS, 2, MINW, false, AR>(p, stream);
    return launch_points_own<NT, VPW, HAS_VIS, 4, MINW, false, AR>(p, stream);
}

// training forward: the same kernels with the cross-view quantities written out for the resident backward (rfn <= 8; two views per
// wave, one for a single view)
template <bool HAS_VIS>
int launch_points_save(const nr::PointParams& p, void* stream) {
#ifdef NR_INFERENCE_ONLY
    return fail("neuray_render_points: saved_dev is a training feature; the bf16-operand variant is inference only");
#else
    if (p.rfn > 8) return fail("neuray_render_points: saved_dev needs rfn <= 8 (rfn=%d)", p.rfn);
    if (p.rfn == 1) return launch_points_own<1, 1, HAS_VIS, 4, 4, true>(p, stream);
    const int nwaves = (p.rfn + 1) / 2;
    if (nwaves >= 4) return launch_points_own<1, 2, HAS_VIS, 1, 3, true>(p, stream);
    if (nwaves >= 2) return launch_points_own<1, 2, HAS_VIS, 2, 3, true>(p, stream);
    return launch_points_own<1, 2, HAS_VIS, 4, 3, true>(p, stream);
#endif
}

// Built decompositions (measured on MI355X, DESIGN.md "point kernel tuning"):
//   views_per_wave = 2, 168 VGPRs, 3 waves per SIMD  - default (2.35 M rays/s on the lego-800 workload)
//   views_per_wave = 1, 128 VGPRs, 4 waves per SIMD  - single reference view, and the A/B reference
#ifndef NR_POINT_MINW
#define NR_POINT_MINW 3        // workgroups per CU the 2-views-per-wave kernel is compiled for (A/B: -DNR_POINT_MINW=2 = 256 VGPRs, no spills)
#endif
#ifndef NR_POINT_MINW_X3
#define NR_POINT_MINW_X3 2     // the same for the AR_X3 instantiation: 207 VGPRs, no spills; compiled for 3 (168 VGPRs, 33 spilled) it is 8.5 % slower (profiles/r06_c_*)
#endif
template <bool HAS_VIS>
int launch_points_cfg(const nr::PointParams& p, int vpw, int arith, void* stream) {
    if (arith == NEURAY_ARITH_X3) {
#ifdef NR_BF16_QUADS
        return fail("neuray_render_points: arith = NEURAY_ARITH_X3 lives in the fp32 library (this is a bf16-operand variant build)");
#else
        if (vpw == 2) return launch_points<1, 2, HAS_VIS, NR_POINT_MINW_X3, nr::AR_X3>(p, stream);
        return fail("neuray_render_points: arith = NEURAY_ARITH_X3 is built for views_per_wave = 2 (rfn >= 2)");
#endif
    }
    if (vpw == 2) return launch_points<1, 2, HAS_VIS, NR_POINT_MINW>(p, stream);
    if (vpw == 1) return launch_points<1, 1, HAS_VIS, 4>(p, stream);
    return fail("neuray_render_points: views_per_wave=%d is not built (1 or 2)", vpw);
}

}  // namespace

namespace {
template <int NT, int MTW, int WC = 1>
void launch_conv2d_x3(const nr::Conv2dX3Params& p, int bands, void* stream) {
    const int lw = p.tw + 2, hp = p.h + 2 * p.pad;
    const long long q = (long long)p.n * hp * lw;
    const int per = nr::kC2Waves * NT * 16;
    const size_t smem = (size_t)nr::conv2d_x3_smem_bytes(NT, lw);
    auto k = nr::conv2d_x3_kernel<NT, MTW, WC>;
#ifndef NEURAY_EMU
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
#endif
    NR_LAUNCH(k, dim3((unsigned)((q + per - 1) / per), (unsigned)(p.cout / 16 / MTW / WC), (unsigned)bands), dim3(64 * nr::kC2Waves * WC), smem, stream, p);
}
int g_conv2d_nt = 0, g_conv2d_mtw = 0, g_conv2d_tw = 0, g_conv2d_wc = 0;     // NEURAY_CONV2D_NT / _MTW / _TW / _WC: tile-shape overrides of the A/B tools
}  // namespace

extern "C" {

int neuray_abi_version(void) { return NEURAY_ABI_VERSION; }
const char* neuray_last_error(void) { return g_err; }
int neuray_is_device_build(void) {
#ifdef NEURAY_EMU
    return 0;
#else
    return 1;
#endif
}

size_t neuray_packed_pass_floats(void) { return (size_t)nr::kPackedPassFloats; }

int neuray_pack_pass_weights(const float* const* tensors_host, float* packed_host) {
    if (!tensors_host || !packed_host) return fail("neuray_pack_pass_weights: null argument");
    const int rc = nr::pack_pass_weights(tensors_host, packed_host);
    if (rc) return fail("neuray_pack_pass_weights: tensor %d of the pass is missing", rc - 1);
    return 0;
}

int neuray_pack_pass_weights_folded(const float* const* tensors_host, float* packed_host) {
    if (!tensors_host || !packed_host) return fail("neuray_pack_pass_weights_folded: null argument");
    const int rc = nr::pack_pass_weights(tensors_host, packed_host, true);
    if (rc) return fail("neuray_pack_pass_weights_folded: tensor %d of the pass is missing", rc - 1);
    return 0;
}

size_t neuray_packed_points_floats_x3(void) { return (size_t)nr::kPackedPointFloatsX3; }

int neuray_pack_pass_weights_x3(const float* const* tensors_host, float* packed_host) {
    if (!tensors_host || !packed_host) return fail("neuray_pack_pass_weights_x3: null argument");
    const int rc = nr::pack_pass_weights_x3(tensors_host, packed_host);
    if (rc) return fail("neuray_pack_pass_weights_x3: tensor %d of the pass is missing", rc - 1);
    return 0;
}

int neuray_mt19937_shuffle(unsigned int* key624_host, int* pos_host, void* data_host, long long n, int itemsize) {
    if (nr::mt19937_shuffle(key624_host, pos_host, data_host, n, itemsize))
        return fail("neuray_mt19937_shuffle: needs key[624], 0 <= pos <= 624, a data pointer and 4- or 8-byte items");
    return 0;
}

int neuray_operand_precision(void) {
#if defined(NR_BF16_SPLIT)
    return 48;         // hi + lo bf16 operands, three bf16 MFMAs per fp32 quad (libneuray_hip_bf16x3.so)
#elif defined(NR_BF16_QUADS)
    return 16;
#else
    return 32;
#endif
}

int neuray_pack_pass_index_map(int has_vis_head, int* index_host, float* scale_host) {
    if (!index_host || !scale_host) return fail("neuray_pack_pass_index_map: null argument");
#ifdef NR_INFERENCE_ONLY
    return fail("neuray_pack_pass_index_map: the bf16-operand build packs on the host only (inference variant, no training path)");
#endif
    if (nr::pack_pass_index_map(has_vis_head != 0, index_host, scale_host)) return fail("neuray_pack_pass_index_map: internal error");
    return 0;
}

// float ranges [begin, end) of the quad fragments inside the packed pass buffer (transposed = 0) or the transposed pack (= 1): what a
// device-side packer of the split library converts from four fp32 weights to (hi pair, hi pair, lo pair, lo pair) after the gather
int neuray_packed_quad_ranges(int transposed, int* ranges_host, int max_pairs) {
    if (!ranges_host || max_pairs < 1) return fail("neuray_packed_quad_ranges: null argument");
    int n = 0;
    const int first = transposed ? (int)nr::L_FWD_COUNT : 0, last = transposed ? (int)nr::L_COUNT : (int)nr::L_FWD_COUNT;
    for (int l = first; l < last; ++l) {
        if (nr::quads_floats(l) == 0) continue;
        if (n >= max_pairs) return fail("neuray_packed_quad_ranges: more than %d ranges", max_pairs);
        ranges_host[2 * n] = nr::quads_offset(l);
        ranges_host[2 * n + 1] = nr::quads_offset(l) + nr::quads_floats(l);
        ++n;
    }
    return -n;          // (negative count on success: 0 is reserved for "no error" elsewhere, positive for failure)
}

int neuray_setup_views(const float* poses, const float* Ks, const float* depth_range, int n, float* out, void* stream) {
    if (n < 1 || n > NEURAY_MAX_VIEWS) return fail("neuray_setup_views: n=%d outside [1,%d]", n, NEURAY_MAX_VIEWS);
    NR_LAUNCH(nr::view_setup_kernel, dim3(1), dim3(64), 0, stream, poses, Ks, depth_range, n, out);
    return check_launch("neuray_setup_views");
}

int neuray_setup_query(const float* pose, const float* Kinv, const float* depth_range, float* out, void* stream) {
    NR_LAUNCH(nr::query_setup_kernel, dim3(1), dim3(64), 0, stream, pose, Kinv, depth_range, out);
    return check_launch("neuray_setup_query");
}

int neuray_relayout_nhwc(const float* src, float* dst, int n, int c, int h, int w, int c_pad, void* stream) {
    if (c_pad < c || n < 1) return fail("neuray_relayout_nhwc: bad shape n=%d c=%d c_pad=%d", n, c, c_pad);
    const int grid = grid_for((long long)n * h * w, 256, 256 * 16);
    NR_LAUNCH(nr::relayout_kernel, dim3(grid), dim3(256), 0, stream, src, dst, n, c, h, w, c_pad);
    return check_launch("neuray_relayout_nhwc");
}

int neuray_sample_coarse_depth(const float* depth_range, int rn, int dn, float* depth, void* stream) {
    return neuray_sample_coarse_depth_jittered(depth_range, nullptr, rn, dn, depth, stream);
}

int neuray_sample_coarse_depth_jittered(const float* depth_range, const float* uniforms, int rn, int dn, float* depth, void* stream) {
    if (dn <= 2) return fail("neuray_sample_coarse_depth: dn=%d must be > 2 (render_ops.py:157)", dn);
    const int grid = grid_for((long long)rn * dn, 256, 256 * 8);
    NR_LAUNCH(nr::coarse_depth_kernel, dim3(grid), dim3(256), 0, stream, depth_range, uniforms, rn, dn, depth);
    return check_launch("neuray_sample_coarse_depth");
}

int neuray_render_points(const NeurayPointsArgs* a, void* stream) {
    if (!a) return fail("neuray_render_points: null args");
    if (a->rfn < 1 || a->rfn > NEURAY_MAX_VIEWS) return fail("neuray_render_points: rfn=%d outside [1,%d]", a->rfn, NEURAY_MAX_VIEWS);
    if (a->dn <= 2 || a->rn < 1) return fail("neuray_render_points: bad rn=%d dn=%d", a->rn, a->dn);
    if (a->use_vis && !a->has_vis_head) return fail("neuray_render_points: use_vis set but the decoder has no vis head");
    if ((long long)a->rn * a->dn > 0x7fffffffLL / 2) return fail("neuray_render_points: rn*dn too large for one call");
    nr::PointParams p;
    p.que_const = a->query_const_dev; p.view_const = a->view_const_dev; p.coords = a->coords_dev; p.depth = a->depth_dev;
    p.ray_feats = a->ray_feats_nhwc_dev; p.img_feats = a->img_feats_nhwc_dev; p.rgba = a->rgba_dev;
    p.weights = a->packed_weights_dev; p.point_out = a->point_out_dev; p.dbg = a->dbg_dev; p.saved = a->saved_dev;
    p.rfn = a->rfn; p.rn = a->rn; p.dn = a->dn; p.h = a->h; p.w = a->w; p.fh = a->fh; p.fw = a->fw;
    p.use_vis = a->use_vis; p.var_bias = a->var_bias; p.folded = a->folded; p.slot_stats = a->slot_stats_dev;
    if (a->folded && a->saved_dev) return fail("neuray_render_points: saved_dev (training forward) takes the unfolded pack");
    if (a->arith != NEURAY_ARITH_F32 && a->arith != NEURAY_ARITH_X3) return fail("neuray_render_points: arith=%d is not built (0 or 1)", a->arith);
    if (a->arith == NEURAY_ARITH_X3 && a->saved_dev) return fail("neuray_render_points: saved_dev (training forward) runs on NEURAY_ARITH_F32");
    // work decomposition: reference views processed per wave (0 = default)
    int vpw = a->views_per_wave ? a->views_per_wave : (a->rfn >= 2 ? 2 : 1);
    // the vis head is only evaluated when compute_prob consumes it (a fine decoder's vis head is ignored on the
    // reference-view path when the coarse decoder has use_vis = False: quirk A.9.2)
    const bool vis = a->has_vis_head && a->use_vis;
    if (a->saved_dev) return vis ? launch_points_save<true>(p, stream) : launch_points_save<false>(p, stream);
    return vis ? launch_points_cfg<true>(p, vpw, a->arith, stream) : launch_points_cfg<false>(p, vpw, a->arith, stream);
}

int neuray_render_rays(const NeurayRaysArgs* a, void* stream) {
    if (!a) return fail("neuray_render_rays: null args");
    if (a->dn < 1 || a->dn > 4 * NEURAY_MAX_SAMPLES) return fail("neuray_render_rays: dn=%d unsupported", a->dn);
    nr::RayParams p;
    p.point_rec = a->point_rec_dev; p.depth = a->depth_dev; p.pos_enc = a->pos_enc_dev; p.weights = a->packed_weights_dev;
    p.hit_prob = a->hit_prob_dev; p.pixel = a->pixel_dev; p.render_depth = a->render_depth_dev; p.ray_mask = a->ray_mask_dev;
    p.density = a->density_dev; p.att_save = a->att_save_dev; p.rn = a->rn; p.dn = a->dn;
    p.mask_view_num = a->ray_mask_view_num; p.mask_point_num = a->ray_mask_point_num;
    // two rays per wave when a ray's samples fill at most half of one (the 32-sample fine pass of the headline configuration)
    const int rpw = a->dn <= 32 ? 2 : 1;
    const size_t smem = nr::ray_smem_bytes(a->dn, rpw);
    if (smem > 160 * 1024) return fail("neuray_render_rays: dn=%d needs %zu bytes of LDS", a->dn, smem);
    const int grid = grid_for(a->rn, nr::kRayWaves * rpw, 256 * 16);
    auto launch = [&](auto k) {
#ifndef NEURAY_EMU
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
#endif
        NR_LAUNCH(k, dim3(grid), dim3(64 * nr::kRayWaves), smem, stream, p);
    };
    if (a->att_save_dev) {                                 // training forward: the attention's softmax statistics are kept for the backward
        if (rpw == 2) launch(nr::rays_kernel<true, 2>); else launch(nr::rays_kernel<true, 1>);
    } else {
        if (rpw == 2) launch(nr::rays_kernel<false, 2>); else launch(nr::rays_kernel<false, 1>);
    }
    return check_launch("neuray_render_rays");
}

int neuray_sample_fine_depth(const float* query_const, const float* depth, const float* hit_prob, const float* u,
                             int rn, int dn, int fdn, int use_all, float* out, void* stream) {
    return neuray_sample_fine_depth_traced(query_const, depth, hit_prob, u, rn, dn, fdn, use_all, out, nullptr, nullptr, stream);
}

int neuray_sample_fine_depth_traced(const float* query_const, const float* depth, const float* hit_prob, const float* u,
                                    int rn, int dn, int fdn, int use_all, float* out, int* idx_out, float* cdf_out, void* stream) {
    if (dn < 2 || dn > NEURAY_MAX_SAMPLES || fdn < 1 || fdn > NEURAY_MAX_SAMPLES)
        return fail("neuray_sample_fine_depth: dn=%d fdn=%d outside [2,%d]", dn, fdn, NEURAY_MAX_SAMPLES);
    nr::FineParams p;
    p.que_const = query_const; p.depth = depth; p.hit_prob = hit_prob; p.u = u; p.out = out; p.idx_out = idx_out; p.cdf_out = cdf_out;
    p.rn = rn; p.dn = dn; p.fdn = fdn; p.use_all = use_all & 1; p.no_sort = (use_all >> 1) & 1; p.linear = (use_all >> 2) & 1;
    const int grid = grid_for(rn, nr::kRayWaves, 256 * 16);
    NR_LAUNCH(nr::fine_kernel, dim3(grid), dim3(64 * nr::kRayWaves), 0, stream, p);
    return check_launch("neuray_sample_fine_depth");
}

int neuray_interpolate_feats(const float* feats, const float* points, const float* mask, int b, int n, int c, int fh, int fw,
                             int h_full, int w_full, int align_corners, float* out, void* stream) {
    if (b < 1 || n < 1 || c < 1) return fail("neuray_interpolate_feats: bad shape b=%d n=%d c=%d", b, n, c);
    const int grid = grid_for((long long)b * n * c, 256, 256 * 8);
    NR_LAUNCH(nr::interpolate_kernel, dim3(grid), dim3(256), 0, stream, feats, points, mask, b, n, c, fh, fw, h_full, w_full,
              align_corners, out);
    return check_launch("neuray_interpolate_feats");
}

int neuray_diff_feats(const float* view_const, const float* lift_const, const float* rgbd, int rfn, int h, int w, float* out, void* stream) {
    if (!view_const || !lift_const || !rgbd || !out) return fail("neuray_diff_feats: null pointer");
    if (rfn < 1 || rfn > NEURAY_MAX_VIEWS || h < 2 || w < 2) return fail("neuray_diff_feats: bad shape rfn=%d h=%d w=%d", rfn, h, w);
    nr::DiffFeatsParams p;
    p.view_const = view_const; p.lift_const = lift_const; p.rgbd = rgbd; p.out = out; p.rfn = rfn; p.h = h; p.w = w;
    const int grid = grid_for((long long)rfn * h * w, 256, 256 * 32);
    NR_LAUNCH(nr::diff_feats_kernel, dim3(grid), dim3(256), 0, stream, p);
    return check_launch("neuray_diff_feats");
}

int neuray_warp_variance(const float* ref_feats, const float* src_feats, const int* nn_ids, const float* transforms, const float* depth_vals,
                         int rfn, int sn, int n_num, int dn, int fh, int fw, float* out, void* stream) {
    return neuray_warp_variance_layout(ref_feats, src_feats, nn_ids, transforms, depth_vals, rfn, sn, n_num, dn, fh, fw, 0, out, stream);
}

int neuray_conv3d_c32_c8(const float* x_ndhwc, const float* wpack, const float* bias, float slope, int n, int d, int h, int w, float* out, void* stream) {
    if (!x_ndhwc || !wpack || !bias || !out) return fail("neuray_conv3d_c32_c8: null pointer");
    if (n < 1 || d < 1 || h < 1 || w < 1 || (long long)d * h * w * 128 >= 0x7fffff00LL)
        return fail("neuray_conv3d_c32_c8: bad shape n=%d d=%d h=%d w=%d (one image's volume must stay below 2^31 bytes)", n, d, h, w);
    nr::Conv0Params p;
    p.x = x_ndhwc; p.wpack = wpack; p.bias = bias; p.out = out; p.n = n; p.d = d; p.h = h; p.w = w; p.slope = slope;
    const long long strips = (long long)n * d * ((h + 1) / 2) * ((w + 15) / 16);       // a wave takes two output rows of a 16-voxel strip
    const int grid = grid_for(strips, nr::kConv0Waves, 256 * 8);
    const size_t smem = sizeof(float) * nr::kConv0PackFloats;                            // 72 KB: two workgroups per CU
    auto k = nr::costreg_conv0_kernel;
#ifndef NEURAY_EMU
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
#endif
    NR_LAUNCH(k, dim3(grid), dim3(64 * nr::kConv0Waves), smem, stream, p);
    return check_launch("neuray_conv3d_c32_c8");
}

long long neuray_conv3x3_x3_pack_bytes(int cin, int cout) {
    if (cin < 32 || cout < 32 || cin % 32 || cout % 32) return -1;
    return (long long)9 * cin * cout * 6;
}

int neuray_conv3x3_x3_pack(const float* w, int cout, int cin, void* wpack, void* wpack_t, void* stream) {
    if (!w || (!wpack && !wpack_t)) return fail("neuray_conv3x3_x3_pack: null pointer");
    if (cin < 32 || cout < 32 || cin % 32 || cout % 32)
        return fail("neuray_conv3x3_x3_pack: (C_in, C_out) = (%d, %d): both must be multiples of 32", cin, cout);
    nr::Conv2dPackParams p;
    p.w = w; p.wpack = (unsigned*)wpack; p.wpack_t = (unsigned*)wpack_t; p.cout = cout; p.cin = cin;
    const int total = 9 * (cin / 32) * (cout / 16) * 256;
    NR_LAUNCH(nr::conv2d_x3_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, p);
    return check_launch("neuray_conv3x3_x3_pack");
}

int neuray_conv3x3_x3(const float* x, const void* wpack, const float* bias, int n, int cin, int cout, int h, int w, int pad, float* out, void* stream) {
    if (!x || !wpack || !out) return fail("neuray_conv3x3_x3: null pointer");
    if (neuray_conv3x3_x3_pack_bytes(cin, cout) < 0)
        return fail("neuray_conv3x3_x3: (C_in, C_out) = (%d, %d): both must be multiples of 32", cin, cout);
    if (n < 1 || pad < 0 || pad > 2 || h + 2 * pad < 3 || w + 2 * pad < 3) return fail("neuray_conv3x3_x3: bad shape n=%d h=%d w=%d pad=%d", n, h, w, pad);
    const int oh = h + 2 * pad - 2, ow = w + 2 * pad - 2;
    if ((long long)n * cin * h * w * 4 >= 0x7fffff00LL || (long long)n * cout * oh * ow * 4 >= 0x7fffff00LL)
        return fail("neuray_conv3x3_x3: the input and the output must each stay below 2^31 bytes");
    static bool env_read = false;
    if (!env_read) {
        env_read = true;
        if (const char* e = getenv("NEURAY_CONV2D_NT")) g_conv2d_nt = atoi(e);
        if (const char* e = getenv("NEURAY_CONV2D_MTW")) g_conv2d_mtw = atoi(e);
        if (const char* e = getenv("NEURAY_CONV2D_TW")) g_conv2d_tw = atoi(e);
        if (const char* e = getenv("NEURAY_CONV2D_WC")) g_conv2d_wc = atoi(e);
    }
    // bands of at most 62 output columns (52-wide rows on 50 / 100 / 200-pixel maps: 2 dropped positions per row, a halo of 106 positions)
    const int tw_max = g_conv2d_tw > 0 ? g_conv2d_tw : 62;
    const int bands = (ow + tw_max - 1) / tw_max;
    nr::Conv2dX3Params p;
    p.x = x; p.wpack = (const unsigned*)wpack; p.bias = bias; p.out = out; p.n = n; p.cin = cin; p.cout = cout; p.h = h; p.w = w; p.pad = pad;
    p.tw = (ow + bands - 1) / bands;
    const int mt = cout / 16;
    const long long q = (long long)n * (h + 2 * pad) * (p.tw + 2);
    if (q + 1024 >= 0x7fffffffLL) return fail("neuray_conv3x3_x3: n * (h + 2 pad) * (band width + 2) must stay below 2^31");
    // 64 positions x 32 output channels per wave: two workgroups per CU (70 KB of LDS each), the fastest shape on every encoder layer
    // (profiles/r06_zz5_conv2d_probes.log); the larger tiles stay built for the A/B tool
    int nt = 4, mtw = 2;
    if (g_conv2d_nt == 4 || g_conv2d_nt == 8) nt = g_conv2d_nt;
    if ((g_conv2d_mtw == 2 || g_conv2d_mtw == 4) && mt % g_conv2d_mtw == 0) mtw = g_conv2d_mtw;
    if ((size_t)nr::conv2d_x3_smem_bytes(nt, p.tw + 2) > 160 * 1024 || nr::conv2d_x3_positions(nt, p.tw + 2) > 64 * nr::conv2d_x3_max_passes(nt))
        return fail("neuray_conv3x3_x3: band of %d columns does not fit the LDS", p.tw);
    if (nt == 8 && mtw == 4) launch_conv2d_x3<8, 4>(p, bands, stream);
    else if (nt == 8) launch_conv2d_x3<8, 2>(p, bands, stream);
    else if (mtw == 4) launch_conv2d_x3<4, 4>(p, bands, stream);
    // two channel groups per workgroup where there are at least eight tiles of output channels: -7 % forward / -10 % data gradient at 128
    // channels, +-3 % at 64 (profiles/r06_zzf_conv2d_wc2.log)
    else if (mt >= 8 && (mt / 2) % 2 == 0 && nr::conv2d_x3_positions(4, p.tw + 2) % 128 == 0 && g_conv2d_wc != 1) launch_conv2d_x3<4, 2, 2>(p, bands, stream);
    else launch_conv2d_x3<4, 2>(p, bands, stream);
    return check_launch("neuray_conv3x3_x3");
}

namespace {
int conv2d_wrw_ksplit(int n, int cin, int cout, int hp, int wp) {
    const int pairs = (cin / 32) * (cout / 32);
    const long long nkb = (long long)n * (((long long)(hp - 2) * wp + 31) / 32);
    long long ks = 512 / pairs;                     // two workgroups per CU over the whole launch ...
    if (ks > nkb / 8) ks = nkb / 8;                 // ... but at least two K blocks per wave
    return ks < 1 ? 1 : (int)ks;
}
}  // namespace

long long neuray_conv3x3_x3_wrw_workspace_floats(int n, int cin, int cout, int hp, int wp) {
    if (cin < 32 || cout < 32 || cin % 32 || cout % 32 || n < 1 || hp < 3 || wp < 10 || (wp & 1)) return -1;
    return (long long)conv2d_wrw_ksplit(n, cin, cout, hp, wp) * (cin / 32) * (cout / 32) * nr::kWrwAcc * 64;
}

int neuray_conv3x3_x3_wrw(const float* dy, const float* xp, int n, int cin, int cout, int hp, int wp, float* workspace, float* dw, void* stream) {
    if (!dy || !xp || !workspace || !dw) return fail("neuray_conv3x3_x3_wrw: null pointer");
    if (neuray_conv3x3_x3_wrw_workspace_floats(n, cin, cout, hp, wp) < 0)
        return fail("neuray_conv3x3_x3_wrw: n=%d (C_in, C_out) = (%d, %d) padded input %d x %d: the channel counts must be multiples of 32 and the padded width even and >= 10",
                    n, cin, cout, hp, wp);
    if ((long long)n * cin * hp * wp * 4 >= 0x7fffff00LL || (long long)n * cout * hp * wp * 4 >= 0x7fffff00LL)
        return fail("neuray_conv3x3_x3_wrw: the input and the output gradient must each stay below 2^31 bytes");
    nr::Conv2dWrwParams p;
    p.dy = dy; p.xp = xp; p.ws = workspace; p.dw = dw; p.n = n; p.cin = cin; p.cout = cout; p.hp = hp; p.wp = wp;
    p.ksplit = conv2d_wrw_ksplit(n, cin, cout, hp, wp);
    const int pairs = (cin / 32) * (cout / 32);
    const size_t smem = sizeof(float) * 2 * nr::kWrwAcc * 64;
    auto k = nr::conv2d_x3_wrw_kernel;
#ifndef NEURAY_EMU
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
#endif
    NR_LAUNCH(k, dim3((unsigned)p.ksplit, (unsigned)pairs), dim3(256), smem, stream, p);
    NR_LAUNCH(nr::conv2d_x3_wrw_reduce_kernel, dim3((unsigned)(pairs * nr::kWrwAcc)), dim3(64 * nr::kWrwRedWaves), 0, stream, p);
    return check_launch("neuray_conv3x3_x3_wrw");
}

int neuray_convtranspose3d_bn_leaky(const float* x, const float* wpack, const float* bias, float slope, const float* skip, int n, int cin, int cout,
                                    int d, int h, int w, float* out, void* stream) {
    if (!x || !wpack || !bias || !out) return fail("neuray_convtranspose3d_bn_leaky: null pointer");
    const long long groups = (long long)(((long long)h * w + 255) / 256) * d * n;      // one thread per input voxel = 2 x 2 x 2 output block
    if (n < 1 || d < 1 || h < 1 || w < 1 || groups > 0x7fffffffLL)
        return fail("neuray_convtranspose3d_bn_leaky: bad shape n=%d d=%d h=%d w=%d", n, d, h, w);
    nr::Up11Params p;
    p.x = x; p.wpack = wpack; p.bias = bias; p.skip = skip; p.out = out; p.n = n; p.d = d; p.h = h; p.w = w; p.slope = slope;
    if (cin == 16 && cout == 8) NR_LAUNCH((nr::costreg_up11_kernel<16, 8>), dim3((unsigned)groups, 8 / nr::kUp11Co), dim3(256), 0, stream, p);
    else if (cin == 32 && cout == 16) NR_LAUNCH((nr::costreg_up11_kernel<32, 16>), dim3((unsigned)groups, 16 / nr::kUp11Co), dim3(256), 0, stream, p);
    else return fail("neuray_convtranspose3d_bn_leaky: (C_in, C_out) = (%d, %d) is not built (16 -> 8, 32 -> 16)", cin, cout);
    return check_launch("neuray_convtranspose3d_bn_leaky");
}

int neuray_convtranspose3d_c16_c8(const float* x, const float* wpack, const float* bias, float slope, const float* skip, int n, int d, int h, int w,
                                  float* out, void* stream) {
    return neuray_convtranspose3d_bn_leaky(x, wpack, bias, slope, skip, n, 16, 8, d, h, w, out, stream);
}

int neuray_conv3d_c8_c1(const float* x, const float* w27, float bias, int n, int d, int h, int w, float* out, void* stream) {
    if (!x || !w27 || !out) return fail("neuray_conv3d_c8_c1: null pointer");
    if (n < 1 || d < 1 || h < 1 || w < 1) return fail("neuray_conv3d_c8_c1: bad shape n=%d d=%d h=%d w=%d", n, d, h, w);
    nr::ProbParams p;
    p.x = x; p.w = w27; p.out = out; p.n = n; p.d = d; p.h = h; p.w_ = w; p.bias = bias;
    const long long grid = (long long)((h * w + 255) / 256) * ((d + nr::kProbSeg - 1) / nr::kProbSeg) * n;
    if (grid > 0x7fffffffLL) return fail("neuray_conv3d_c8_c1: volume too large");
    NR_LAUNCH(nr::costreg_prob_kernel, dim3((unsigned)grid), dim3(256), 0, stream, p);
    return check_launch("neuray_conv3d_c8_c1");
}

int neuray_conv3d_bn_leaky(const float* x, const float* wpack, const float* bias, float slope, int n, int cin, int cout, int stride, int d,
                           int h, int w, float* out, void* stream) {
    if (!x || !wpack || !bias || !out) return fail("neuray_conv3d_bn_leaky: null pointer");
    if (n < 1 || d < 1 || h < 1 || w < 1) return fail("neuray_conv3d_bn_leaky: bad shape n=%d d=%d h=%d w=%d", n, d, h, w);
    if ((long long)cin * d * h * w * 4 >= 0x7fffff00LL) return fail("neuray_conv3d_bn_leaky: one image's input volume must stay below 2^31 bytes");
    if (stride != 1 && stride != 2) return fail("neuray_conv3d_bn_leaky: stride %d", stride);
    nr::Conv3dParams p;
    p.x = x; p.wpack = wpack; p.bias = bias; p.out = out; p.n = n; p.d = d; p.h = h; p.w = w; p.slope = slope; p.cin = cin; p.cout = cout;
    const int od = (d - 1) / stride + 1, oh = (h - 1) / stride + 1, ow = (w - 1) / stride + 1;
    const long long tasks = (long long)n * od * ((oh + nr::kC3Rows - 1) / nr::kC3Rows) * ((ow + 15) / 16);
    const dim3 grid(grid_for(tasks, nr::kC3Waves, 256 * 16)), block(64 * nr::kC3Waves);
    // the kernel's channel counts: C_in rounded up to a multiple of 4, C_out to a multiple of 16 (wpack / bias are padded with zeros to them)
    const int ci = (cin + 3) / 4 * 4, co = (cout + 15) / 16 * 16;
    if (ci == 16 && co == 16 && stride == 1) NR_LAUNCH((nr::conv3d_kernel<16, 16, 1>), grid, block, 0, stream, p);
    else if (ci == 32 && co == 32 && stride == 1) NR_LAUNCH((nr::conv3d_kernel<32, 32, 1>), grid, block, 0, stream, p);
    else if (ci == 8 && co == 16 && stride == 2) NR_LAUNCH((nr::conv3d_kernel<8, 16, 2>), grid, block, 0, stream, p);
    else if (ci == 16 && co == 32 && stride == 2) NR_LAUNCH((nr::conv3d_kernel<16, 32, 2>), grid, block, 0, stream, p);
    else if (ci == 4 && co == 16 && stride == 1) NR_LAUNCH((nr::conv3d_kernel<4, 16, 1>), grid, block, 0, stream, p);
    else if (ci == 8 && co == 16 && stride == 1) NR_LAUNCH((nr::conv3d_kernel<8, 16, 1>), grid, block, 0, stream, p);
    else return fail("neuray_conv3d_bn_leaky: (C_in, C_out, stride) = (%d, %d, %d) is not built (padded to multiples of 4 / 16: 4 | 8 | 16 -> 16 and 32 -> 32 at "
                     "stride 1, 8 -> 16 and 16 -> 32 at stride 2)", cin, cout, stride);
    return check_launch("neuray_conv3d_bn_leaky");
}

int neuray_scale_shift_leaky(float* x, const float* scale, const float* shift, int n, int c, long long inner, float slope, void* stream) {
    if (!x || !scale || !shift) return fail("neuray_scale_shift_leaky: null pointer");
    if (n < 1 || c < 1 || inner < 1 || (long long)n * c > 0x7fffffffLL) return fail("neuray_scale_shift_leaky: bad shape n=%d c=%d inner=%lld", n, c, inner);
    nr::ScaleShiftLeakyParams p;
    p.x = x; p.scale = scale; p.shift = shift; p.inner = inner; p.n = n; p.c = c; p.slope = slope;
    const long long planes = (long long)n * c;
    if (planes > 65535) return fail("neuray_scale_shift_leaky: %lld planes exceed the grid's y range", planes);
    long long chunks = (4096 + planes - 1) / planes, most = (inner / 4 + 255) / 256;     // ~4096 workgroups, each thread >= one 16-byte piece
    if (chunks > most) chunks = most;
    if (chunks < 1) chunks = 1;
    NR_LAUNCH(nr::scale_shift_leaky_kernel, dim3((unsigned)chunks, (unsigned)planes), dim3(256), 0, stream, p);
    return check_launch("neuray_scale_shift_leaky");
}

int neuray_warp_variance_layout(const float* ref_feats, const float* src_feats, const int* nn_ids, const float* transforms, const float* depth_vals,
                                int rfn, int sn, int n_num, int dn, int fh, int fw, int channels_last, float* out, void* stream) {
    if (!ref_feats || !src_feats || !nn_ids || !transforms || !depth_vals || !out) return fail("neuray_warp_variance: null pointer");
    if (rfn < 1 || sn < 1 || n_num < 1 || dn < 1 || fh < 2 || fw < 2)
        return fail("neuray_warp_variance: bad shape rfn=%d sn=%d n_num=%d dn=%d fh=%d fw=%d", rfn, sn, n_num, dn, fh, fw);
    nr::WarpVarParams p;
    p.ref_feats = ref_feats; p.src_feats = src_feats; p.nn_ids = nn_ids; p.transforms = transforms; p.depth_vals = depth_vals; p.out = out;
    p.rfn = rfn; p.n_num = n_num; p.dn = dn; p.fh = fh; p.fw = fw; p.channels_last = channels_last;
    if (channels_last && (long long)rfn * dn <= 65535) {        // eight lanes per voxel, a tap = one 128-byte line per instruction
        NR_LAUNCH(nr::warp_variance_cl_kernel, dim3((fh * fw + 31) / 32, rfn * dn), dim3(256), 0, stream, p);
        return check_launch("neuray_warp_variance (channels last)");
    }
    const int grid = grid_for((long long)rfn * dn * fh * fw, 256, 256 * 64);
    NR_LAUNCH(nr::warp_variance_kernel, dim3(grid), dim3(256), 0, stream, p);
    return check_launch("neuray_warp_variance");
}

// ---- fused InstanceNorm + activation (+ residual) + reflection pad of the per-image encoders (nr_kernels_norm.h) ----------------
namespace {
int norm_chunks(int planes, int hw) {          // workgroups per plane: ~4096 workgroups in all (16 per CU), each thread >= 4 elements
    int want = (4096 + planes - 1) / planes, most = (hw + 1023) / 1024;
    if (want > most) want = most;
    return want < 1 ? 1 : want;
}
}  // namespace

int neuray_inorm_forward(const float* x, const float* gamma, const float* beta, const float* res, long long res_stride_n,
                         long long res_stride_c, long long res_stride_h, int n, int c, int h, int w, int pad, int act, float eps,
                         float* raw_zeroed, float* stats, float* out_padded, long long out_stride_n, void* stream) {
    if (!x || !gamma || !beta || !raw_zeroed || !stats || !out_padded) return fail("neuray_inorm_forward: null pointer");
    if (n < 1 || c < 1 || h < 1 || w < 1 || pad < 0 || pad >= h || pad >= w || act < 0 || act > 2 || (long long)(h + 2 * pad) * (w + 2 * pad) >= (1 << 23))
        return fail("neuray_inorm_forward: bad arguments n=%d c=%d h=%d w=%d pad=%d act=%d", n, c, h, w, pad, act);
    const long long img = (long long)c * (h + 2 * pad) * (w + 2 * pad);
    if (out_stride_n != 0 && out_stride_n < img) return fail("neuray_inorm_forward: out_stride_n %lld < %lld", out_stride_n, img);
    const int planes = n * c, hw = h * w;
    NR_LAUNCH(nr::inorm_stats_kernel, dim3(norm_chunks(planes, hw), planes), dim3(256), 0, stream, x, hw, raw_zeroed);
    nr::NormApplyParams p;
    p.x = x; p.raw = raw_zeroed; p.gamma = gamma; p.beta = beta; p.res = res; p.out = out_padded; p.stats = stats;
    p.rs_n = res_stride_n; p.rs_c = res_stride_c; p.rs_h = res_stride_h; p.out_stride_n = out_stride_n ? out_stride_n : img;
    p.n = n; p.c = c; p.h = h; p.w = w; p.pad = pad; p.act = act; p.eps = eps;
    NR_LAUNCH(nr::inorm_apply_kernel, dim3(norm_chunks(planes, (h + 2 * pad) * (w + 2 * pad)), planes), dim3(256), 0, stream, p);
    return check_launch("neuray_inorm_forward");
}

int neuray_inorm_backward(const float* x, const float* out_padded, long long out_stride_n, const float* d_out_padded, long long d_out_stride_n,
                          const float* stats, const float* gamma, int n, int c, int h, int w, int pad, int act, float* raw_zeroed, float* dx,
                          float* d_res, float* d_gamma, float* d_beta, void* stream) {
    if (!x || !out_padded || !d_out_padded || !stats || !gamma || !raw_zeroed || !dx) return fail("neuray_inorm_backward: null pointer");
    if (n < 1 || c < 1 || h < 1 || w < 1 || pad < 0 || pad >= h || pad >= w || act < 0 || act > 2 || (long long)(h + 2 * pad) * (w + 2 * pad) >= (1 << 23))
        return fail("neuray_inorm_backward: bad arguments n=%d c=%d h=%d w=%d pad=%d act=%d", n, c, h, w, pad, act);
    const long long img = (long long)c * (h + 2 * pad) * (w + 2 * pad);
    if ((out_stride_n != 0 && out_stride_n < img) || (d_out_stride_n != 0 && d_out_stride_n < img))
        return fail("neuray_inorm_backward: image strides %lld / %lld < %lld", out_stride_n, d_out_stride_n, img);
    nr::NormBwdParams p;
    p.x = x; p.out = out_padded; p.d_out = d_out_padded; p.stats = stats; p.gamma = gamma; p.raw = raw_zeroed; p.dx = dx; p.d_res = d_res;
    p.d_gamma = d_gamma; p.d_beta = d_beta;
    p.out_stride_n = out_stride_n ? out_stride_n : img; p.d_out_stride_n = d_out_stride_n ? d_out_stride_n : img;
    p.n = n; p.c = c; p.h = h; p.w = w; p.pad = pad; p.act = act;
    const int planes = n * c, hw = h * w;
    NR_LAUNCH(nr::inorm_backward_reduce_kernel, dim3(norm_chunks(planes, hw), planes), dim3(256), 0, stream, p);
    NR_LAUNCH(nr::inorm_backward_apply_kernel, dim3(norm_chunks(planes, hw), planes), dim3(256), 0, stream, p);
    return check_launch("neuray_inorm_backward");
}

int neuray_upsample2x_pad_forward(const float* x, int planes, int h, int w, int pad, float scale_y, float scale_x, float* out_padded, void* stream) {
    if (!x || !out_padded) return fail("neuray_upsample2x_pad_forward: null pointer");
    if (planes < 1 || h < 2 || w < 2 || pad < 0 || pad > 1 || (long long)(2 * h + 2 * pad) * (2 * w + 2 * pad) >= (1 << 23))
        return fail("neuray_upsample2x_pad_forward: bad arguments planes=%d h=%d w=%d pad=%d", planes, h, w, pad);
    nr::UpsampleParams p;
    p.x = x; p.out = out_padded; p.planes = planes; p.h = h; p.w = w; p.pad = pad; p.sy = scale_y; p.sx = scale_x;
    NR_LAUNCH(nr::upsample2x_pad_kernel, dim3(norm_chunks(planes, (2 * h + 2 * pad) * (2 * w + 2 * pad)), planes), dim3(256), 0, stream, p);
    return check_launch("neuray_upsample2x_pad_forward");
}

int neuray_upsample2x_pad_backward(const float* d_out_padded, int planes, int h, int w, int pad, const int* cnt_y, const int* idx_y,
                                   const float* wgt_y, const int* cnt_x, const int* idx_x, const float* wgt_x, float* dx, void* stream) {
    if (!d_out_padded || !cnt_y || !idx_y || !wgt_y || !cnt_x || !idx_x || !wgt_x || !dx) return fail("neuray_upsample2x_pad_backward: null pointer");
    if (planes < 1 || h < 2 || w < 2 || w > 2047 || pad < 0 || pad > 1 || (long long)h * w >= (1 << 23))
        return fail("neuray_upsample2x_pad_backward: bad arguments planes=%d h=%d w=%d (<= 2047) pad=%d", planes, h, w, pad);
    nr::UpsampleBwdParams p;
    p.d_out = d_out_padded; p.dx = dx; p.cnt_y = cnt_y; p.idx_y = idx_y; p.wgt_y = wgt_y; p.cnt_x = cnt_x; p.idx_x = idx_x; p.wgt_x = wgt_x;
    p.planes = planes; p.h = h; p.w = w; p.hp = 2 * h + 2 * pad; p.wp = 2 * w + 2 * pad;
    NR_LAUNCH(nr::upsample2x_pad_backward_kernel, dim3((h + nr::kUpRows - 1) / nr::kUpRows, planes), dim3(256),
              sizeof(float) * nr::kUpRows * (size_t)p.wp, stream, p);
    return check_launch("neuray_upsample2x_pad_backward");
}

int neuray_rays_points(const float* query_const, const float* coords, const float* depth, int rn, int dn, float* centers,
                       float* dirs, float* pts, float* que_dir, void* stream) {
    if (rn < 1 || (pts && (dn < 1 || !depth || !que_dir))) return fail("neuray_rays_points: bad arguments rn=%d dn=%d", rn, dn);
    const int grid = grid_for((long long)rn * (pts ? dn : 1), 256, 256 * 8);
    NR_LAUNCH(nr::rays_points_kernel, dim3(grid), dim3(256), 0, stream, query_const, coords, depth, rn, dn, centers, dirs, pts, que_dir);
    return check_launch("neuray_rays_points");
}

int neuray_depth_dists(const float* depth, const float* que_depth_range, int inverse, int rows, int dn, float* out, void* stream) {
    if (rows < 1 || dn < 1 || (inverse && !que_depth_range)) return fail("neuray_depth_dists: bad arguments");
    const int grid = grid_for((long long)rows * dn, 256, 256 * 8);
    NR_LAUNCH(nr::dists_kernel, dim3(grid), dim3(256), 0, stream, depth, que_depth_range, inverse, rows, dn, out);
    return check_launch("neuray_depth_dists");
}

int neuray_project_points(const float* view_const, const float* pts, int rfn, int pn, int h, int w, float* dir, float* pts2d,
                          float* depth, unsigned char* mask, void* stream) {
    if (rfn < 1 || pn < 1) return fail("neuray_project_points: bad shape rfn=%d pn=%d", rfn, pn);
    const int grid = grid_for((long long)rfn * pn, 256, 256 * 8);
    NR_LAUNCH(nr::project_kernel, dim3(grid), dim3(256), 0, stream, view_const, pts, rfn, pn, h, w, dir, pts2d, depth, mask);
    return check_launch("neuray_project_points");
}

int neuray_alpha2hit_prob(const float* alpha, int rows, int dn, float* out, void* stream) {
    if (rows < 1 || dn < 1) return fail("neuray_alpha2hit_prob: bad shape");
    const int grid = grid_for(rows, 64, 256 * 8);
    NR_LAUNCH(nr::hit_prob_kernel, dim3(grid), dim3(64), 0, stream, alpha, rows, dn, out);
    return check_launch("neuray_alpha2hit_prob");
}

int neuray_direct_render_points(const float* query_const, const float* view_const, const float* coords, const float* depth,
                                const float* rgba, const float* view_rec, const float* regs, int rfn, int rn, int dn, int h, int w,
                                float ground, float* alpha, float* color, void* stream) {
    if (rfn < 1 || rfn > NEURAY_MAX_VIEWS || rn < 1 || dn < 1) return fail("neuray_direct_render_points: bad shape rfn=%d rn=%d dn=%d", rfn, rn, dn);
    if (!view_rec || !alpha) return fail("neuray_direct_render_points: view_rec / alpha missing");
    const int grid = grid_for((long long)rn * dn, 128, 256 * 16);
    NR_LAUNCH(nr::dr_points_kernel, dim3(grid), dim3(128), 0, stream, query_const, view_const, coords, depth, rgba, view_rec, regs, rfn, rn,
              dn, h, w, ground, alpha, color);
    return check_launch("neuray_direct_render_points");
}

int neuray_direct_render_rays(const float* alpha, const float* colors, int color_stride, int color_first, int rn, int dn,
                              float* hit_prob, float* pixel, void* stream) {
    if (rn < 1 || dn < 1 || color_stride < 3 || color_first < 0) return fail("neuray_direct_render_rays: bad shape");
    const int grid = grid_for(rn, 64, 256 * 8);
    NR_LAUNCH(nr::dr_rays_kernel, dim3(grid), dim3(64), 0, stream, alpha, colors, color_stride, color_first, rn, dn, hit_prob, pixel);
    return check_launch("neuray_direct_render_rays");
}

int neuray_dist_decoder_rows(const float* feats, const float* packed_weights, int n, int has_vis_head, float var_bias,
                             float* mean, float* var, float* aw, float* vis, void* stream) {
    if (n < 1) return fail("neuray_dist_decoder_rows: n=%d", n);
    if (has_vis_head && !vis) return fail("neuray_dist_decoder_rows: vis output missing");
    const int grid = grid_for(n, 32 * 4, 256 * 8);
    if (has_vis_head) NR_LAUNCH(nr::decoder_rows_kernel<true>, dim3(grid), dim3(256), 0, stream, feats, packed_weights, n, var_bias, mean, var, aw, vis);
    else NR_LAUNCH(nr::decoder_rows_kernel<false>, dim3(grid), dim3(256), 0, stream, feats, packed_weights, n, var_bias, mean, var, aw, vis);
    return check_launch("neuray_dist_decoder_rows");
}

int neuray_self_hit_prob(const float* query_const, const float* depth, const float* mean, const float* var, const float* aw,
                         const float* vis, int rn, int dn, float* out, void* stream) {
    if (rn < 1 || dn < 2) return fail("neuray_self_hit_prob: bad shape rn=%d dn=%d", rn, dn);
    const int grid = grid_for((long long)rn * dn, 256, 256 * 8);
    NR_LAUNCH(nr::self_hit_prob_kernel, dim3(grid), dim3(256), 0, stream, query_const, depth, mean, var, aw, vis, rn, dn, out);
    return check_launch("neuray_self_hit_prob");
}

int neuray_mfma_selftest(const float* A, const float* B, float* D, void* stream) {
    NR_LAUNCH(nr::mfma_selftest_kernel, dim3(1), dim3(64), 0, stream, A, B, D);
    return check_launch("neuray_mfma_selftest");
}

int neuray_x3_selftest(const float* A, const float* B, float* D, float* parts, void* stream) {
    if (!A || !B || !D) return fail("neuray_x3_selftest: null pointer");
    NR_LAUNCH(nr::x3_selftest_kernel, dim3(1), dim3(64), 0, stream, A, B, D, parts);
    return check_launch("neuray_x3_selftest");
}

int neuray_points_resident_workgroups(int arith, int rfn) {
#ifdef NEURAY_EMU
    (void)arith; (void)rfn;
    return 0;
#else
    if (rfn < 7 || rfn > 8) return -1;                 // (the headline shape: four waves of two views)
    int n = 0;
    hipError_t e;
    if (arith == NEURAY_ARITH_X3) {
#ifdef NR_BF16_QUADS
        return -1;
#else
        auto k = nr::points_kernel<1, 2, false, 1, 512, NR_POINT_MINW_X3, false, false, nr::AR_X3>;
        const size_t smem = nr::point_smem_bytes<1>(4, nr::AR_X3);
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, smem);
#endif
    } else {
        auto k = nr::points_kernel<1, 2, false, 1, 512, NR_POINT_MINW, false, false, nr::AR_F32>;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, nr::point_smem_bytes<1>(4, nr::AR_F32));
    }
    return e == hipSuccess ? n : -1;
#endif
}

static_assert(NEURAY_PACKED_RAY_FLOATS == nr::kPackedRayFloats && NEURAY_RW_WQ == nr::RW_WQ && NEURAY_RW_WK == nr::RW_WK &&
              NEURAY_RW_WV == nr::RW_WV && NEURAY_RW_FC == nr::RW_FC && NEURAY_RW_LNW == nr::RW_LNW && NEURAY_RW_LNB == nr::RW_LNB &&
              NEURAY_RW_OG0W == nr::RW_OG0W && NEURAY_RW_OG0B == nr::RW_OG0B && NEURAY_RW_OG2W == nr::RW_OG2W &&
              NEURAY_RW_OG2B == nr::RW_OG2B, "abi");

int neuray_render_rays_backward(const NeurayRaysBwdArgs* a, void* stream) {
    if (!a || !a->point_rec_dev || !a->depth_dev || !a->pos_enc_dev || !a->packed_weights_dev || !a->d_pixel_dev ||
        !a->d_point_rec_dev || !a->d_ray_weights_dev)
        return fail("neuray_render_rays_backward: null argument");
    if (a->rn < 1) return fail("neuray_render_rays_backward: rn=%d", a->rn);
    if (a->dn < 3 || a->dn > NEURAY_MAX_SAMPLES) return fail("neuray_render_rays_backward: dn=%d outside [3,%d]", a->dn, NEURAY_MAX_SAMPLES);
    nr::RayBwdParams p;
    p.point_rec = a->point_rec_dev; p.depth = a->depth_dev; p.pos_enc = a->pos_enc_dev; p.weights = a->packed_weights_dev;
    p.d_pixel = a->d_pixel_dev; p.d_hit_prob = a->d_hit_prob_dev; p.d_depth = a->d_render_depth_dev;
    p.d_point_rec = a->d_point_rec_dev; p.d_weights = a->d_ray_weights_dev; p.att_saved = a->att_saved_dev; p.rn = a->rn; p.dn = a->dn;
    const size_t smem = nr::ray_bwd_smem_bytes(a->dn);
    const int waves = nr::ray_bwd_waves(a->dn);            // rays per workgroup
    const int grid = grid_for(a->rn, waves, 256 * 4);
    if (a->dn <= 64) {
        auto k = nr::rays_backward_kernel<1>;
#ifndef NEURAY_EMU
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
#endif
        NR_LAUNCH(k, dim3(grid), dim3(64 * waves), smem, stream, p);
    } else {                                               // two samples per lane
        auto k = nr::rays_backward_kernel<2>;
#ifndef NEURAY_EMU
        if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
#endif
        NR_LAUNCH(k, dim3(grid), dim3(64 * waves), smem, stream, p);
    }
    return check_launch("neuray_render_rays_backward");
}

size_t neuray_flat_pass_floats(void) { return (size_t)nr::kFlatPassFloats; }
size_t neuray_packed_t_floats(void) { return (size_t)nr::kPackedTFloats; }
size_t neuray_points_saved_floats(int npts) { return npts < 1 ? 0 : (size_t)((npts + 15) / 16) * nr::kSavedTileFloats; }
int neuray_pack_pass_t_index_map(int has_vis_head, int* index) {
    if (!index) return fail("neuray_pack_pass_t_index_map: null argument");
#ifdef NR_INFERENCE_ONLY
    return fail("neuray_pack_pass_t_index_map: the bf16-operand library is inference only");
#else
    const int rc = nr::pack_pass_t_index_map(has_vis_head != 0, index);
    return rc ? fail("neuray_pack_pass_t_index_map: tensor %d missing", rc - 1) : 0;
#endif
}
size_t neuray_flat_tensor_offset(int t) { return (t < 0 || t > nr::T_COUNT) ? (size_t)0 : (size_t)nr::tensor_offset(t); }

// The point backward is nr_kernels_bwd2.h: 8 waves x 1 view at two waves per SIMD, run as its two halves - tail, then front - with a
// hand-over buffer in between (NeurayPointsBwdArgs.handover_dev).  Round 3's second decomposition (4 waves x 2 views per wave) was retired
// in round 4; round 6 removed the first-version kernel (rfn 9..16: a view count no shipped configuration trains with) and the one-launch
// form of the resident kernel (299 spilled VGPRs, 0.92 against 0.63 ms per pass).
size_t neuray_points_backward_handover_floats(int npoints) {
#ifdef NR_INFERENCE_ONLY
    (void)npoints;
    return 0;
#else
    return npoints < 1 ? 0 : nr::point_bwd2_handover_floats(npoints);
#endif
}

int neuray_render_points_backward(const NeurayPointsBwdArgs* a, void* stream) {
    if (!a || !a->query_const_dev || !a->view_const_dev || !a->coords_dev || !a->depth_dev || !a->ray_feats_nhwc_dev ||
        !a->img_feats_nhwc_dev || !a->rgba_dev || !a->flat_weights_dev || !a->d_point_rec_dev || !a->d_flat_weights_dev ||
        !a->d_ray_feats_nhwc_dev || !a->d_img_feats_nhwc_dev)
        return fail("neuray_render_points_backward: null argument");
    if (a->rfn < 1 || a->rfn > NEURAY_MAX_VIEWS) return fail("neuray_render_points_backward: rfn=%d outside [1,%d]", a->rfn, NEURAY_MAX_VIEWS);
    if (a->rn < 1 || a->dn < 3 || a->dn > NEURAY_MAX_SAMPLES) return fail("neuray_render_points_backward: rn=%d dn=%d", a->rn, a->dn);
#ifdef NR_INFERENCE_ONLY
    return fail("neuray_render_points_backward: the bf16-operand library is inference only");
#else
    if (a->rfn > nr::kB2Waves)
        return fail("neuray_render_points_backward: rfn=%d - the backward covers at most %d reference views (the forward kernels take %d)",
                    a->rfn, nr::kB2Waves, NEURAY_MAX_VIEWS);
    if (!a->packed_weights_dev || !a->packed_t_weights_dev)
        return fail("neuray_render_points_backward: packed_weights_dev and packed_t_weights_dev are needed (neuray_pack_pass_weights / neuray_pack_pass_t_index_map)");
    if (!a->saved_dev) return fail("neuray_render_points_backward: saved_dev is NULL (run neuray_render_points with saved_dev on the same inputs first)");
    if (!a->handover_dev) return fail("neuray_render_points_backward: handover_dev is NULL (neuray_points_backward_handover_floats(rn * dn) floats of scratch)");
    nr::PointBwd2Params q;
    q.que_const = a->query_const_dev; q.view_const = a->view_const_dev; q.coords = a->coords_dev; q.depth = a->depth_dev;
    q.ray_feats = a->ray_feats_nhwc_dev; q.img_feats = a->img_feats_nhwc_dev; q.rgba = a->rgba_dev;
    q.weights = a->packed_weights_dev; q.weights_t = a->packed_t_weights_dev;
    q.d_point_rec = a->d_point_rec_dev; q.d_flat = a->d_flat_weights_dev; q.d_ray_feats = a->d_ray_feats_nhwc_dev;
    q.d_img_feats = a->d_img_feats_nhwc_dev; q.saved = a->saved_dev;
    q.rfn = a->rfn; q.rn = a->rn; q.dn = a->dn; q.h = a->h; q.w = a->w; q.fh = a->fh; q.fw = a->fw;
    q.use_vis = a->use_vis; q.var_bias = a->var_bias;
    q.handover = a->handover_dev;
    const int grid2 = grid_for((long long)a->rn * a->dn, 16, 256);            // persistent: one workgroup per CU
    const size_t smem = nr::point_bwd2_smem_bytes();
    auto launch = [&](auto k) {
#ifndef NEURAY_EMU
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
#endif
        NR_LAUNCH(k, dim3(grid2), dim3(64 * nr::kB2Waves), smem, stream, q);
    };
    // (a vis head that compute_prob does not consume - the fine decoder's when the coarse decoder has use_vis = False, quirk A.9.2 -
    // has an identically zero gradient on this path: the kernel without the head is the same computation)
    const bool vis = a->has_vis_head && a->use_vis;
    // two launches: tail, then front (nr_kernels_bwd2.h B2Part)
    if (vis) { launch(nr::points_backward2_kernel<true, nr::B2_TAIL>); launch(nr::points_backward2_kernel<true, nr::B2_FRONT>); }
    else { launch(nr::points_backward2_kernel<false, nr::B2_TAIL>); launch(nr::points_backward2_kernel<false, nr::B2_FRONT>); }
    return check_launch("neuray_render_points_backward");
#endif
}

int neuray_self_hit_prob_backward(const float* qc, const float* depth, const float* feats, const float* packed, const float* packed_t,
                                           int has_vis_head, int use_vis, float var_bias, const float* d_hit, int rn, int dn,
                                           float* d_feats, float* d_flat, void* stream) {
#ifdef NR_INFERENCE_ONLY
    return fail("neuray_self_hit_prob_backward: the bf16-operand variant is inference only");
#else
    if (!qc || !depth || !feats || !packed || !packed_t || !d_hit || !d_feats || !d_flat)
        return fail("neuray_self_hit_prob_backward: null argument");
    if (rn < 1 || dn < 3 || dn > NEURAY_MAX_SAMPLES) return fail("neuray_self_hit_prob_backward: rn=%d dn=%d", rn, dn);
    nr::SelfHitBwd2Params p;
    p.que_const = qc; p.depth = depth; p.feats = feats; p.weights = packed; p.weights_t = packed_t; p.d_hit = d_hit;
    p.d_feats = d_feats; p.d_flat = d_flat; p.rn = rn; p.dn = dn; p.use_vis = use_vis; p.var_bias = var_bias;
    const dim3 grid(grid_for(rn, 16, 512));
    if (has_vis_head && use_vis) NR_LAUNCH(nr::self_hit_backward2_kernel<true>, grid, dim3(64), 0, stream, p);   // (an unused vis head: zero gradient)
    else NR_LAUNCH(nr::self_hit_backward2_kernel<false>, grid, dim3(64), 0, stream, p);
    return check_launch("neuray_self_hit_prob_backward");
#endif
}

int neuray_dist_decoder_rows_backward(const float* feats, const float* packed, const float* packed_t, int n, int has_vis_head,
                                               float var_bias, const float* d_mean, const float* d_var, const float* d_aw, const float* d_vis,
                                               float* d_feats, float* d_flat, void* stream) {
#ifdef NR_INFERENCE_ONLY
    return fail("neuray_dist_decoder_rows_backward: the bf16-operand variant is inference only");
#else
    if (!feats || !packed || !packed_t || !d_feats || !d_flat) return fail("neuray_dist_decoder_rows_backward: null argument");
    if (n < 1) return fail("neuray_dist_decoder_rows_backward: n=%d", n);
    nr::RowsBwd2Params p;
    p.feats = feats; p.weights = packed; p.weights_t = packed_t; p.d_mean = d_mean; p.d_var = d_var; p.d_aw = d_aw; p.d_vis = d_vis;
    p.d_feats = d_feats; p.d_flat = d_flat; p.n = n; p.var_bias = var_bias;
    // (a persistent grid: every workgroup ends with one atomicAdd per weight of the heads it ran - 2048 of them cost more in the flush than in the rows)
    const dim3 grid(grid_for(n, 16, 512));
    if (has_vis_head && d_vis) NR_LAUNCH(nr::decoder_rows_backward2_kernel<true>, grid, dim3(64), 0, stream, p);      // (no gradient into the vis head: not run)
    else NR_LAUNCH(nr::decoder_rows_backward2_kernel<false>, grid, dim3(64), 0, stream, p);
    return check_launch("neuray_dist_decoder_rows_backward");
#endif
}

int neuray_interpolate_feats_backward(const float* d_out, const float* points, const float* mask, int b, int n, int c, int fh,
                                      int fw, int h_full, int w_full, int align_corners, float* d_feats, void* stream) {
    if (!d_out || !points || !d_feats) return fail("neuray_interpolate_feats_backward: null argument");
    if (b < 1 || n < 1 || c < 1 || fh < 1 || fw < 1) return fail("neuray_interpolate_feats_backward: bad shape");
    NR_LAUNCH(nr::interpolate_backward_kernel, dim3(grid_for((long long)b * n * c, 256, 4096)), dim3(256), 0, stream, d_out, points,
              mask, b, n, c, fh, fw, h_full, w_full, align_corners, d_feats);
    return check_launch("neuray_interpolate_feats_backward");
}

int neuray_interpolate_feats_backward_staged(const float* d_out, const float* points, const float* mask, int b, int n, int c, int fh,
                                             int fw, int h_full, int w_full, int align_corners, float* tmp_nhwc_zeroed, float* d_feats, void* stream) {
    if (!d_out || !points || !d_feats || !tmp_nhwc_zeroed) return fail("neuray_interpolate_feats_backward_staged: null argument");
    if (b < 1 || n < 1 || c < 1 || fh < 1 || fw < 1 || (long long)b * ((c + 31) / 32) > 65535) return fail("neuray_interpolate_feats_backward_staged: bad shape");
    NR_LAUNCH(nr::interpolate_backward_nhwc_kernel, dim3(grid_for((long long)b * n * c, 256, 4096)), dim3(256), 0, stream, d_out, points,
              mask, b, n, c, fh, fw, h_full, w_full, align_corners, tmp_nhwc_zeroed);
    if (int rc = check_launch("neuray_interpolate_feats_backward_staged")) return rc;
    NR_LAUNCH(nr::nhwc_add_to_nchw_kernel, dim3((fh * fw + 31) / 32, ((c + 31) / 32) * b), dim3(256), 0, stream, tmp_nhwc_zeroed, fh * fw, c, d_feats);
    return check_launch("neuray_interpolate_feats_backward_staged");
}

int neuray_group_sum_selftest(const float* x, float* y, void* stream) {
    NR_LAUNCH(nr::group_sum_selftest_kernel, dim3(1), dim3(64), 0, stream, x, y);
    return check_launch("neuray_group_sum_selftest");
}

#ifdef NR_B2_PROFILE
// profile build only (tools/profile_bwd2.py): read (and optionally clear) the per-mark cycle sums of points_backward2_kernel
int neuray_debug_b2_profile(unsigned long long* out, int clear) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(nr::nr_b2_prof), sizeof(unsigned long long) * nr::kB2Marks * 8) != hipSuccess) return 1;
    if (clear) {
        static unsigned long long zeros[nr::kB2Marks * 8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(nr::nr_b2_prof), zeros, sizeof(zeros)) != hipSuccess) return 1;
    }
    return 0;
}
#endif

}  // extern "C"
