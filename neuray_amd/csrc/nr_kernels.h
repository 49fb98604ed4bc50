// Kernels of the NeuRay per-ray render path (gfx950).  See DESIGN.md for the decomposition:
//   view_setup   : per-view / query camera constants                     (render_ops.py:15-16,94)
//   relayout     : NCHW -> channels-last maps (one 128 B line per bilinear tap)
//   coarse_depth : a1
//   points (P)   : a2-a14 minus attention, one wave per reference view, MFMA fp32 MLP stack
//   rays (R)     : attention over the samples of a ray, sigma head, compositing (a14-a16)
//   fine (F)     : inverse-CDF resampling + sort (a17)
#pragma once
#include "nr_device.h"

namespace nr {

// -------------------------------------------------------------------------------------------------
// view / query constants
// -------------------------------------------------------------------------------------------------
__global__ void view_setup_kernel(const float* __restrict__ poses, const float* __restrict__ Ks,
                                  const float* __restrict__ depth_range, int n, float* __restrict__ out) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const float* P = poses + v * 12;
    const float* K = Ks + v * 9;
    float* o = out + v * kViewConst;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j)
            o[i * 4 + j] = dot3(K[i * 3 + 0], K[i * 3 + 1], K[i * 3 + 2], P[j], P[4 + j], P[8 + j]);
    for (int i = 0; i < 3; ++i) o[12 + i] = dot3(-P[i], -P[4 + i], -P[8 + i], P[3], P[7], P[11]);
    o[15] = rn_div(-1.0f, depth_range[2 * v]);
    o[16] = rn_div(-1.0f, depth_range[2 * v + 1]);
    o[17] = rn_div(1.0f, rn_sub(o[16], o[15]));   // 1 / (far' - near'): feature-path normalisation
    o[18] = 0.0f; o[19] = 0.0f;
}

__global__ void query_setup_kernel(const float* __restrict__ pose, const float* __restrict__ Kinv,
                                   const float* __restrict__ depth_range, float* __restrict__ o) {
    if (blockIdx.x * blockDim.x + threadIdx.x != 0) return;
    for (int i = 0; i < 9; ++i) o[i] = Kinv[i];
    for (int i = 0; i < 12; ++i) o[9 + i] = pose[i];
    for (int i = 0; i < 3; ++i) o[21 + i] = dot3(-pose[i], -pose[4 + i], -pose[8 + i], pose[3], pose[7], pose[11]);
    o[24] = rn_div(-1.0f, depth_range[0]);
    o[25] = rn_div(-1.0f, depth_range[1]);
    o[26] = depth_range[0];
    o[27] = rn_div(1.0f, rn_sub(o[25], o[24]));   // 1 / (far' - near')
}

// -------------------------------------------------------------------------------------------------
// NCHW -> NHWC (c_pad >= c channels, padding zero-filled).  One thread per (n, y, x).
// -------------------------------------------------------------------------------------------------
__global__ void relayout_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int c, int h, int w, int c_pad) {
    const long long total = (long long)n * h * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long img = i / ((long long)h * w);
        const long long pix = i - img * h * w;
        const float* s = src + img * c * h * w + pix;
        float* d = dst + i * c_pad;
        for (int ch = 0; ch < c_pad; ++ch) d[ch] = ch < c ? s[(long long)ch * h * w] : 0.0f;
    }
}

// -------------------------------------------------------------------------------------------------
// a1: coarse depths [rn][dn]
// -------------------------------------------------------------------------------------------------
// rnd: null, or the uniforms of sample_depth(random_sample=True) [rn][dn-2] (render_ops.py:160-161): interior tick i
// becomes i + (r - 0.5) * 0.999
__global__ void coarse_depth_kernel(const float* __restrict__ depth_range, const float* __restrict__ rnd, int rn, int dn,
                                    float* __restrict__ out) {
    const float near = depth_range[0], far = depth_range[1];
    const long long total = (long long)rn * dn;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % dn);
        if (rnd && k > 0 && k < dn - 1) {
            const float inv_near = rn_div(1.0f, near);
            const float step = rn_div(rn_sub(rn_div(1.0f, far), inv_near), (float)(dn - 1));
            const float val = rn_add((float)k, rn_mul(rn_sub(rnd[(i / dn) * (dn - 2) + (k - 1)], 0.5f), 0.999f));
            out[i] = rn_div(1.0f, rn_add(inv_near, rn_mul(step, val)));
        } else {
            out[i] = coarse_depth(near, far, k, dn);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// point kernel
// -------------------------------------------------------------------------------------------------
struct PointParams {
    const float* que_const;    // [kQueryConst]
    const float* view_const;   // [rfn][kViewConst]
    const float* coords;       // [rn][2] pixel (x, y)
    const float* depth;        // [rn][dn]
    const float* ray_feats;    // [rfn][fh][fw][32]
    const float* img_feats;    // [rfn][fh][fw][32]
    const float* rgba;         // [rfn][h][w][4]
    const float* weights;      // packed pass weights
    float* point_out;          // [rn*dn][kPointRec]
    float* dbg;                // optional [rn*dn][rfn][16]
    float* saved;              // training (SAVE kernels): [ceil(rn*dn / 16)][kSavedTileFloats] cross-view quantities for the backward
    int rfn, rn, dn, h, w, fh, fw;
    int use_vis;               // the COARSE decoder's use_vis governs compute_prob in both passes (renderer.py:75)
    float var_bias;            // AddBias value of var_decoder (dist_decoder.py:81)
    int folded;                // the packed weights carry prob_embed.2 folded into neuray_fc.0 / base_fc.0 (inference packs)
    unsigned long long* slot_stats;   // optional [2]: += (view slots whose layers ran, view slots) of the launch (slot skipping)
};

constexpr int kDbgFields = 16;

// What the training forward leaves for points_backward2_kernel, per tile of 16 consecutive sample points ([rows][64 lanes],
// lane = 16 g + point): rows 0..15 base_fc.0's per-point part (output tile mo, register r -> row 4 mo + r; the LDS exchange
// layout of both kernels), rows 16..59 the four cross-view statistics mean0 | var0 | mean1 | var1 (11 rows each: 8 image
// channels 8 g + k, 3 rgb), row 60 sum of the view masks, row 61 sum of weight0, rows 62..77 geometry_fc.0's (scaled-ELU) hidden
// layer (tile mo, register r -> row 62 + 4 mo + r), 78..85 / 86..93 visibility-weighted mean / variance (channel 16 (k / 4) + 4 g +
// k % 4 in row k: the D layout), 94..97 geometry_fc's output, 98 max z, 99 sum exp(z - max z), 100 sum vis'', then
// [8 views][8][16 points]: mu0, mu1, s0, s1, aw, nu of the dist decoder.  The backward reads them instead of redoing the decoder
// forward, every cross-view all-reduce and the per-point layers' forward.
constexpr int kSavedBgRow = 0, kSavedStatRow = 16, kSavedMsumRow = 60, kSavedSw0Row = 61, kSavedGeoRow = 62, kSavedGmeanRow = 78,
              kSavedGvarRow = 86, kSavedGRow = 94, kSavedZmaxRow = 98, kSavedSezRow = 99, kSavedSvisRow = 100, kSavedDist = 101 * 64;
constexpr int kSavedTileFloats = kSavedDist + 8 * 8 * 16;

__device__ __forceinline__ float sel4(int g, float a, float b, float c, float d) {
    return g == 0 ? a : (g == 1 ? b : (g == 2 ? c : d));
}

template <int NT>
constexpr int point_rmax() { return 12 * NT; }

// AR_X3: the 16-row exchange buffer lives inside the all-reduce scratch (points_kernel xrow), which makes room for the larger weight stage
template <int NT>
inline size_t point_smem_bytes(int nwaves, int ar = AR_F32) {
    return sizeof(float) * (64 * ((size_t)(nwaves + 1) * point_rmax<NT>() + (ar == AR_X3 ? 0 : 16 * NT)) + weight_lds_floats(ar));
}

// owner waves accumulate one cross-view statistic (8 image channels in gathered order + 3 rgb channels per tile)
// into their tile(s) of base_fc.0's per-point part: statistic STAT uses quads [2*STAT, 2*STAT+2) and single STAT
template <int NT, int OWN, int STAT, class WS>
__device__ __forceinline__ void bg_accumulate(WS W, int lane, int g, int wave, int nw,
                                              const float (&st)[NT * 11], v4f (&accg)[OWN][NT]) {
    float xq[NT][8], x1[NT][1];
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t) {
        NR_PRAGMA_UNROLL
        for (int s = 0; s < 8; ++s) xq[t][s] = st[t * 11 + s];
        x1[t][0] = sel4(g, st[t * 11 + 8], st[t * 11 + 9], st[t * 11 + 10], 0.0f);
    }
    NR_PRAGMA_UNROLL
    for (int j = 0; j < OWN; ++j) {
        const int mo = wave + j * nw;
        if (mo < 4) layer_tile_slice<L_BG, NT, 2 * STAT, 2, STAT, 1>(W, lane, mo, xq, x1, accg[j]);      // (fp32 operands in both arithmetics)
    }
}

// AR_X3 (207 of the 256 registers of its two-workgroups-per-CU budget): the owner wave keeps its tile(s) of base_fc.0's per-point part - 8 quad
// fragments, 4 singles, the bias - in registers for the whole launch (the tile a wave owns never changes), instead of fetching a
// K-slice from L2 behind each of the four statistics all-reduces, where the wave - and at the next barrier its three siblings - waited
// for it.
template <int OWN> struct BgResident { float4 q[OWN][8]; float s1[OWN][4]; float4 b[OWN]; };
template <int OWN, class WS>
__device__ __forceinline__ void bg_resident_load(WS W, int lane, int wave, int nw, BgResident<OWN>& r) {
    constexpr int AR = ws_ar<WS>::value;
    NR_PRAGMA_UNROLL
    for (int j = 0; j < OWN; ++j) {
        const int mo = wave + j * nw, m = mo < 4 ? mo : 0;
        NR_PRAGMA_UNROLL
        for (int kq = 0; kq < 8; ++kq) r.q[j][kq] = wld4(W, lane * 16 + m * (8 * 1024), (quads_offset(L_BG, AR) + kq * 256) * 4);
        NR_PRAGMA_UNROLL
        for (int k1 = 0; k1 < 4; ++k1) r.s1[j][k1] = wld1(W, lane * 4 + m * (4 * 256), (single_offset(L_BG, AR) + k1 * 64) * 4);
        r.b[j] = wld4(W, (lane >> 4) * 16 + m * 64, bias_offset(L_BG, AR) * 4);
    }
}
template <int NT, int OWN, int STAT>
__device__ __forceinline__ void bg_accumulate(const BgResident<OWN>& r, int g, int wave, int nw,
                                              const float (&st)[NT * 11], v4f (&accg)[OWN][NT]) {
    float xq[NT][8], x1[NT][1];
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t) {
        NR_PRAGMA_UNROLL
        for (int s = 0; s < 8; ++s) xq[t][s] = st[t * 11 + s];
        x1[t][0] = sel4(g, st[t * 11 + 8], st[t * 11 + 9], st[t * 11 + 10], 0.0f);
    }
    NR_PRAGMA_UNROLL
    for (int j = 0; j < OWN; ++j) {
        const int mo = wave + j * nw;
        if (mo < 4) {                    // (the same K order as layer_tile_slice: the slice's two quads, then its single)
            mfma_quad<NT>(r.q[j][2 * STAT], 0, xq, accg[j]);
            mfma_quad<NT>(r.q[j][2 * STAT + 1], 1, xq, accg[j]);
            NR_PRAGMA_UNROLL
            for (int t = 0; t < NT; ++t) accg[j][t] = nr_mfma16(r.s1[j][STAT], x1[t][0], accg[j][t]);
        }
    }
}

// sum (or max; RED_MAX0: row 0 max, the other rows sum) over the NA ACTIVE view slots of the wave, then over the waves of the
// workgroup.  The IDLE = slots - NA skipped slots of a wave (points_kernel "slot skipping") contribute what a fully masked view
// contributes in the reference: exact zeros to every sum and the masked_fill value -1e9 (ibrnet.py:365) to the maximum row.
template <int NA, int IDLE, int R, int RMAX, int OP>
__device__ __forceinline__ void view_allreduce(const float (&v)[NA > 0 ? NA : 1][R], float (&out)[R], float* red, int wave,
                                               int nw, int lane) {
    NR_PRAGMA_UNROLL
    for (int r = 0; r < R; ++r) {
        const bool is_max = OP == RED_MAX || (OP == RED_MAX0 && r == 0);
        float a = NA > 0 ? v[0][r] : (is_max ? -1e9f : 0.0f);
        NR_PRAGMA_UNROLL
        for (int s = 1; s < NA; ++s) a = red_combine<OP, R>(a, v[s][r], r);
        if (IDLE > 0 && NA > 0 && is_max) a = fmaxf(a, -1e9f);
        out[r] = a;
    }
    block_allreduce<R, RMAX, OP, R>(out, red, wave, nw, lane);
}

template <int N> struct SlotCount { static constexpr int value = N; };

#ifndef NR_POINT_SKIP
#define NR_POINT_SKIP 1        // 0: every view slot runs its layers, masked or not (A/B timing)
#endif
#ifndef NR_POINT_TILE_ORDER
#define NR_POINT_TILE_ORDER 0
#endif
#ifndef NR_POINT_TRANSPOSED
#define NR_POINT_TRANSPOSED 1  // 0: inference tiles are 16 consecutive samples of one ray as well (A/B timing)
#endif

// Point kernel.  One workgroup = ceil(rfn / VPW) waves x one tile of 16 sample points; wave w processes the reference
// views [w*VPW, w*VPW + VPW) of the tile ("slots").  The slots of a wave share every weight fragment and give the MFMA
// pipe independent accumulator chains.
//   OWN = number of 16-feature output tiles of the per-point layers (base_fc.0 global part, geometry_fc.0) a wave
//         owns: ceil(4 / nwaves).
//   DBG = the per-(point, view) record p.dbg is written (tests, direct rendering).  A template flag, not a run-time test of the
//         pointer: the product instantiation carries none of its 6 predicated store blocks per tile nor their lane masks
//         (loop-invariant SGPR pairs that hipcc spilled to VGPR lanes and read back with v_readlane + s_nop inside the loop).
// Slot skipping (inference instantiations with two views per wave): a slot whose 16 (point, view) columns are ALL outside the
//   view (mask = 0: render_ops.py:100-104,127-128) contributes exactly 0 to every cross-view sum of the reference - weight,
//   weight0, vis and vis'' are products with the mask (ibrnet.py:333-349), its colour logit is masked_fill'ed to -1e9 (:365) and
//   the gathered features are multiplied by the mask (render_ops.py:140-143) - so none of its per-view layers is evaluated: the
//   tile body is instantiated for 2, 1 and 0 active slots (NA) and picked per wave and tile by a wave-uniform branch (a lone
//   active slot is moved to position 0 first; the two-term slot sums are commutative, so the result does not depend on it).
//   Every variant executes the same barriers, weight-stage copies and per-point (owner wave) work.
//   p.folded: the packed weights carry prob_embed.2 folded into its two consumers (nr_pack.cpp pack_pass_weights fold = true):
//   the layer is skipped and prob_embed.0's ReLU output takes its place.
//   AR = arithmetic of the MFMA layers (nr_layout.h): AR_F32, or AR_X3 = three-way split bf16 operands on v_mfma_f32_16x16x32_bf16
//   (p.weights is then the split pack of neuray_pack_pass_weights_x3: inference, always folded).
template <int NT, int VPW, bool HAS_VIS, int OWN, int MAXT, int MINW, bool SAVE = false, bool DBG = false, int AR = AR_F32>
__global__ void __launch_bounds__(MAXT, MINW) points_kernel(PointParams p) {
    static_assert(NT == 1, "one 16-point tile per wave iteration");
    static_assert(AR == AR_F32 || !SAVE, "the training forward runs on the fp32 MFMA");
    NR_DYNAMIC_SMEM(float, smem);
    constexpr int RMAX = point_rmax<NT>();
    constexpr int NS = VPW;
    constexpr bool SKIP = (NR_POINT_SKIP != 0) && !SAVE && !DBG && VPW == 2;
    const int lane = threadIdx.x & 63;
    const int wave = NR_UNIFORM((int)(threadIdx.x >> 6));
    const int nw = (p.rfn + VPW - 1) / VPW;          // waves per workgroup
    const int g = lane >> 4, c = lane & 15;
    float* red = smem;
    float* xch = smem + (size_t)(nw + 1) * RMAX * 64;
    float* wl = xch + (AR == AR_X3 ? 0 : 16 * NT * 64);      // staged weights of the current phase
    // the 16 rows through which the owner waves hand base_fc.0's per-point part / geometry_fc.0's hidden layer to the others: row
    // 4 mo + r of tile mo.  AR_X3: inside the all-reduce scratch, in the region of the wave that owns the tile (mo = wave + j nw -> its
    // rows 4 j + r; with one wave the 16 rows run on into the result region).  A wave's region is read by others only between the two
    // barriers of an all-reduce, the rows are written after one and consumed before the next phase barrier: no extra synchronisation.
    auto xrow = [&](int mo, int r) -> float* {
        if constexpr (AR == AR_X3) return red + ((size_t)(mo % nw) * RMAX + (mo / nw) * 4 + r) * 64;
        else return xch + (mo * 4 + r) * 64;
    };
    auto make_w = [&]() {
        if constexpr (AR == AR_X3) return GlbW3{nr_make_wbuf(p.weights, sizeof(float) * kPackedPointFloatsX3)};
        else return nr_make_wbuf(p.weights, sizeof(float) * kPackedPassFloats);
    };
    const auto W = make_w();
    const float* __restrict__ qc = p.que_const;
    const float qnearp = qc[24], qfarp = qc[25], qinv = qc[27];
    const float w_m1 = (float)(p.w - 1), h_m1 = (float)(p.h - 1);
    const float inv_w_m1 = 1.0f / w_m1, inv_h_m1 = 1.0f / h_m1, inv_rfn = 1.0f / (float)p.rfn;
    const size_t fmap = (size_t)p.fh * p.fw * 32, imap = (size_t)p.h * p.w * 4;
    const nr_mbuf rf_map = nr_make_mbuf(p.ray_feats, sizeof(float) * fmap * p.rfn);
    const nr_mbuf if_map = nr_make_mbuf(p.img_feats, sizeof(float) * fmap * p.rfn);
    const nr_mbuf rgb_map = nr_make_mbuf(p.rgba, sizeof(float) * imap * p.rfn);
    const int goff = 32 * g;
    const int npts = p.rn * p.dn;
    const int dn = p.dn;
    const bool use_vis = p.use_vis != 0;
    const bool folded = AR == AR_X3 || (!SAVE && p.folded != 0);
    const bool dbg_lane = DBG && (g == 0);
    // XCD-aware tile map: workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only).  Giving every
    // XCD a contiguous run of tiles keeps the texels that neighbouring samples / rays share inside one private L2
    // instead of fetching them into all eight (the grid is a multiple of 8).
    const int bid = (int)(blockIdx.x % 8) * (int)(gridDim.x / 8) + (int)(blockIdx.x / 8);
    int seq0 = 0;                                       // phases entered so far (selects the stage region)
#ifndef NR_X3_BG_RESIDENT
#define NR_X3_BG_RESIDENT 1
#endif
#ifndef NR_X3_GF_PREFETCH
#define NR_X3_GF_PREFETCH 1
#endif
    constexpr bool BG_RES = AR == AR_X3 && OWN == 1 && (NR_X3_BG_RESIDENT != 0);    // base_fc.0's per-point fragments live in registers (BgResident; 7 or 8 views: one tile per wave, 40 registers)
    BgResident<OWN> bgres;
    if constexpr (BG_RES) bg_resident_load(W, lane, wave, nw, bgres);
    int n_active = 0, n_slots = 0;                      // slot-skipping statistics of this wave (p.slot_stats)
    // Tile = 16 sample points.  Training (SAVE) and the per-view record: 16 consecutive samples of one ray (the saved buffer's and the
    // backward's tiling).  Inference (TR): the SAME sample index of 16 consecutive rays - neighbouring pixels at one depth project to
    // neighbouring texels and leave a view's image together, so more (tile, view) slots are fully masked and skipped (bench scene,
    // coarse pass: 14.1 % instead of 9.2 %) and the 16 columns of a slot share their texel lines.  A point's result does not depend on
    // its tile mates either way.
    constexpr bool TR = (NR_POINT_TRANSPOSED != 0) && !SAVE && !DBG;
    const int nloop = TR ? ((p.rn + 15) / 16) * dn * 16 : npts;
#ifndef NR_POINT_CHUNK
#define NR_POINT_CHUNK 1           // consecutive tiles a workgroup walks before it strides on by the grid (A/B: 2, 4, 8)
#endif
    constexpr int CH = NR_POINT_CHUNK;
    const int G = (int)gridDim.x;
    auto tile_base = [&](int it) { return (((it / CH) * G + bid) * CH + (it % CH)) * 16; };
    if (tile_base(0) < nloop) stage_issue<PH_DIST_M>(wl, W, wave, nw, lane);
    for (int it = 0, base; (base = tile_base(it)) < nloop; ++it, seq0 += phase_count(HAS_VIS)) {
        const bool more = tile_base(it + 1) < nloop;   // another tile follows: its first phase is prefetched
        // lane index for the weight loads that go to global memory (L_BG, L_GF1, L_GF2): opaque and re-made per tile,
        // otherwise hipcc treats these loop-invariant loads as hoistable, keeps ~50 fragment registers alive across the
        // whole tile loop and spills them (seen as "spills outside, reloads inside the loop" in -Rpass-missed=regalloc)
        const int glane = lane + nr_opaque_zero();
#ifndef NR_NO_OPAQUE_WAVE
        const int wave_t = wave + nr_opaque_szero();     // (see nr_opaque_szero)
#else
        const int wave_t = wave;
#endif
        const int gg = glane >> 4;
        // ---------------- geometry (a2-a6) ----------------------------------------------------
        int pidx; bool pvalid;
        float mask[NS], dlt[NS][4], tref[NS], pu[NS], pv[NS], lo, hi;
        int soff_f[NS], soff_c[NS];                    // byte offsets of the slot's view inside the feature / colour maps
        {
            int pi, ray, smp;
            if constexpr (TR) {
#if NR_POINT_TILE_ORDER == 1      // sample-major: consecutive tiles are neighbouring ray blocks at one sample index
                const int tix = base >> 4, nrb = (p.rn + 15) / 16;
                smp = tix / nrb;
                const int rb = tix - smp * nrb;
#else                             // ray-block-major: consecutive tiles walk along the rays of one block
                const int tix = base >> 4, rb = tix / dn;          // wave-uniform
                smp = tix - rb * dn;
#endif
                ray = rb * 16 + c;
                pvalid = ray < p.rn;
                ray = pvalid ? ray : p.rn - 1;
                pi = ray * dn + smp;
            } else {
                pi = base + c;
                pvalid = pi < npts;
                pi = pi < npts ? pi : npts - 1;
                ray = pi / dn;
                smp = pi - ray * dn;
            }
            pidx = pi;
            const Ray r = make_ray<false>(qc, p.coords[2 * ray], p.coords[2 * ray + 1]);
            const float* drow = p.depth + (size_t)ray * dn;
            const float d = drow[smp];
            // half intervals in normalised inverse depth (render_ops.py:46-52, dist_decoder.py:34-38); feature path:
            // hardware reciprocals (they feed only the logistic CDFs)
            const float s_c = norm_inv_depth_fast(d, qnearp, qfarp, qinv);
            const float s_n = norm_inv_depth_fast(drow[smp + 1 < dn ? smp + 1 : smp], qnearp, qfarp, qinv);
            const float s_p = norm_inv_depth_fast(drow[smp > 0 ? smp - 1 : 0], qnearp, qfarp, qinv);
            const float half_c = (smp == dn - 1) ? 500000.0f : (s_n - s_c) * 0.5f;
            const float half_p = (s_c - s_p) * 0.5f;
            hi = half_c;
            lo = (smp == 0) ? half_c : half_p;
            const float px = rn_add(r.cx, rn_mul(r.dx, d));
            const float py = rn_add(r.cy, rn_mul(r.dy, d));
            const float pz = rn_add(r.cz, rn_mul(r.dz, d));
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                const int vraw = wave_t * VPW + s;
                const bool vok = vraw < p.rfn;                // padding view when rfn % VPW != 0: masked out
                const int view = vok ? vraw : p.rfn - 1;
                const float* __restrict__ vc = p.view_const + view * kViewConst;
                Proj pr = project_point<false>(vc, px, py, pz, (float)p.w, (float)p.h);   // u, v, z, mask stay exact
                if (!vok) pr.mask = 0.0f;
                mask[s] = pr.mask;
                dlt[s][0] = pr.dirx - r.qx; dlt[s][1] = pr.diry - r.qy; dlt[s][2] = pr.dirz - r.qz;
                dlt[s][3] = dot3(pr.dirx, pr.diry, pr.dirz, r.qx, r.qy, r.qz);
                tref[s] = norm_inv_depth_fast(fmaxf(pr.z, 1e-5f), vc[15], vc[16], vc[17]);
                pu[s] = pr.u; pv[s] = pr.v;
                if (dbg_lane && pvalid && vok) {
                    float* d_ = p.dbg + ((size_t)pi * p.rfn + view) * kDbgFields;
                    d_[0] = pr.mask; d_[1] = pr.u; d_[2] = pr.v; d_[3] = pr.z;
                }
                soff_f[s] = view * (int)(fmap * sizeof(float)); soff_c[s] = view * (int)(imap * sizeof(float));
            }
        }

        // ---------------- the tile body for NA active slots (slots [0, NA) are the active ones) --------------------------
        auto tile = [&](auto na_tag) NR_LAMBDA_INLINE {
            constexpr int NA = decltype(na_tag)::value, NA1 = NA > 0 ? NA : 1, IDLE = NS - NA;
            // gathers (a7): the 20 tap loads of a slot are all issued before anything is blended, so they share one memory
            // round trip (left alone hipcc serialises load -> wait -> blend per map: 3 dependent round trips per slot).
            // Measured: +5% whole-job; holding two slots' taps (160 registers) or staggering slot s+1's loads into slot s's
            // blends spills and is slower than this.
            float fray[NA1][8], fimg[NA1][8], rgb[NA1][3];
#ifdef NR_PROBE_PRE
            float fm1[NA1][8], fv1[NA1][8], fa1[NA1][8];
#endif
            if constexpr (NA > 0) {
                Taps tfs[NA], tcs[NA];
                NR_PRAGMA_UNROLL
                for (int s = 0; s < NA; ++s) {
                    tfs[s] = make_taps_fast(pu[s], pv[s], w_m1, h_m1, inv_w_m1, inv_h_m1, p.fw, p.fh, p.fw == p.w && p.fh == p.h);
                    tcs[s] = make_taps_fast(pu[s], pv[s], w_m1, h_m1, inv_w_m1, inv_h_m1, p.w, p.h, true);
#ifdef NR_PROBE_NO_GATHER      // timing probe (wrong results): every tap of every lane reads texel 0 - what the texture path costs
                    tfs[s].o00 = tfs[s].o10 = tfs[s].o01 = tfs[s].o11 = 0;
                    tcs[s].o00 = tcs[s].o10 = tcs[s].o01 = tcs[s].o11 = 0;
#endif
                }
                NR_PRAGMA_UNROLL
                for (int s = 0; s < NA; ++s) {
                    float4 qf[8], qi[8], qc4[4];
                    issue8(rf_map, goff, soff_f[s], tfs[s], qf);
                    issue8(if_map, goff, soff_f[s], tfs[s], qi);
                    issue_rgb(rgb_map, soff_c[s], tcs[s], qc4);
                    NR_PIN();
                    blend8(qf, tfs[s], mask[s], fray[s]);
                    blend8(qi, tfs[s], mask[s], fimg[s]);
                    blend_rgb(qc4, tcs[s], mask[s], rgb[s]);
                    // the blends happen HERE: left alone the img / rgb blends are sunk to their first use (ray_dir_fc, two phases
                    // later) and the 48 raw tap registers of the slot are carried - spilled - through the whole dist decoder
                    NR_PRAGMA_UNROLL
                    for (int k = 0; k < 8; ++k) { NR_KEEP(fray[s][k]); NR_KEEP(fimg[s][k]); }
                    NR_PRAGMA_UNROLL
                    for (int j = 0; j < 3; ++j) NR_KEEP(rgb[s][j]);
                    NR_PIN();
#ifdef NR_PROBE_PRE            // timing probe (wrong results): the gathers of a map that carries the first layers of the dist heads per texel
                    {
                        float4 q1[8], q2[8], q3[8];
                        issue8(rf_map, goff, soff_f[s] + 4096, tfs[s], q1);
                        issue8(rf_map, goff, soff_f[s] + 8192, tfs[s], q2);
                        issue8(rf_map, goff, soff_f[s] + 12288, tfs[s], q3);
                        NR_PIN();
                        blend8(q1, tfs[s], mask[s], fm1[s]);
                        blend8(q2, tfs[s], mask[s], fv1[s]);
                        blend8(q3, tfs[s], mask[s], fa1[s]);
                        NR_PRAGMA_UNROLL
                        for (int k = 0; k < 8; ++k) { NR_KEEP(fm1[s][k]); NR_KEEP(fv1[s][k]); NR_KEEP(fa1[s][k]); }
                        NR_PIN();
                    }
#endif
                }
            }
            float none[NA1][1];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NA1; ++s) none[s][0] = 0.0f;
            NoLayer last;      // "no next layer": the following layer belongs to the next phase
            // the layers' quad operands in the form the arithmetic takes (AR_X3: split once here, used by the three or four dist heads
            // and prob_embed.0)
            decltype(auto) o_fray = operand<AR>(fray);
            decltype(auto) o_none = operand<AR>(none);

            // ---------------- dist decoder (a9) + probabilities (a10, a11) ----------------------------
            float hit[NA1], vis[NA1];
            {
                float h1[NA1][8], h2[NA1][8], fm[NA1][2], fv[NA1][2], fa[NA1][1];
                float mu0[NA1], mu1[NA1], s0[NA1], s1[NA1], aw[NA1], nu[NA1];
                LayerPreT<L_DV2, AR> p_dv2; VecPre<L_DFIN_V> p_fv;
                const auto W1 = phase_enter<PH_DIST_M, HAS_VIS>(wl, W, seq0, true, wave_t, nw, lane);
                if constexpr (NA > 0) {
                    LayerPreT<L_DM1, AR> p_dm1; LayerPreT<L_DM2, AR> p_dm2; LayerPreT<L_DV1, AR> p_dv1; VecPre<L_DFIN_M> p_fm;
#ifdef NR_PROBE_PRE
                    layer_prefetch<L_DM2>(W1, lane, p_dm2);
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s)
                        NR_PRAGMA_UNROLL
                        for (int k = 0; k < 8; ++k) h1[s][k] = elu_s(fm1[s][k]);
#else
                    layer_prefetch<L_DM1>(W1, lane, p_dm1);
                    layer_fwd<L_DM1, NA, ACT_ELU>(W1, lane, p_dm1, o_fray, none, h1, p_dm2);
#endif
                    layer_fwd<L_DM2, NA, ACT_ELU>(W1, lane, p_dm2, operand<AR>(h1), none, h2, p_fm);
                    layer_prefetch<L_DV1>(W1, lane, p_dv1);
                    layer_vec<L_DFIN_M, NA>(p_fm, h2, fm);
#ifdef NR_PROBE_PRE
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s)
                        NR_PRAGMA_UNROLL
                        for (int k = 0; k < 8; ++k) h1[s][k] = elu_s(fv1[s][k]);
#else
                    layer_fwd<L_DV1, NA, ACT_ELU>(W1, lane, p_dv1, o_fray, none, h1, last);
#endif
                }
                const auto W2 = phase_enter<PH_DIST_VA, HAS_VIS>(wl, W, seq0, true, wave_t, nw, lane);
                if constexpr (NA > 0) {
                    LayerPreT<L_DA1, AR> p_da1; LayerPreT<L_DA2, AR> p_da2; VecPre<L_DFIN_A> p_fa;
                    layer_prefetch<L_DV2>(W2, lane, p_dv2);
                    layer_fwd<L_DV2, NA, ACT_ELU>(W2, lane, p_dv2, operand<AR>(h1), none, h2, p_fv);
                    layer_prefetch<L_DA1>(W2, lane, p_da1);
                    layer_vec<L_DFIN_V, NA>(p_fv, h2, fv);
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) {
                        mu0[s] = softplus(fm[s][0]); mu1[s] = softplus(fm[s][1]);
                        s0[s] = softplus(fv[s][0]) + p.var_bias; s1[s] = softplus(fv[s][1]) + p.var_bias;
                    }
#ifdef NR_PROBE_PRE
                    layer_prefetch<L_DA2>(W2, lane, p_da2);
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s)
                        NR_PRAGMA_UNROLL
                        for (int k = 0; k < 8; ++k) h1[s][k] = elu_s(fa1[s][k]);
#else
                    layer_fwd<L_DA1, NA, ACT_ELU>(W2, lane, p_da1, o_fray, none, h1, p_da2);
#endif
                    layer_fwd<L_DA2, NA, ACT_ELU>(W2, lane, p_da2, operand<AR>(h1), none, h2, p_fa);
                    layer_vec<L_DFIN_A, NA>(p_fa, h2, fa);
                }
                if constexpr (HAS_VIS) {
                    const auto W2s = phase_enter<PH_DIST_S, HAS_VIS>(wl, W, seq0, true, wave_t, nw, lane);
                    if constexpr (NA > 0) {
                        LayerPreT<L_DS1, AR> p_ds1; LayerPreT<L_DS2, AR> p_ds2; VecPre<L_DFIN_S> p_fs;
                        float fs[NA][1];
                        layer_prefetch<L_DS1>(W2s, lane, p_ds1);
                        layer_fwd<L_DS1, NA, ACT_ELU>(W2s, lane, p_ds1, o_fray, none, h1, p_ds2);
                        layer_fwd<L_DS2, NA, ACT_ELU>(W2s, lane, p_ds2, operand<AR>(h1), none, h2, p_fs);
                        layer_vec<L_DFIN_S, NA>(p_fs, h2, fs);
                        NR_PRAGMA_UNROLL
                        for (int s = 0; s < NA; ++s) { aw[s] = sigmoidf(fa[s][0]); nu[s] = sigmoidf(fs[s][0]); }
                    }
                } else {
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) { aw[s] = sigmoidf(fa[s][0]); nu[s] = 1.0f; }
                }
                NR_PRAGMA_UNROLL
                for (int s = 0; s < NA; ++s) {
                    float v_, h_;
                    logistic_prob(tref[s], lo, hi, mu0[s], mu1[s], s0[s], s1[s], aw[s], nu[s], use_vis && HAS_VIS, v_, h_);
                    vis[s] = v_ * mask[s]; hit[s] = h_ * mask[s];
                    const int vraw = wave_t * VPW + s;
                    if constexpr (SAVE) {
                        if (g == 0 && vraw < p.rfn && vraw < 8) {
                            float* d_ = p.saved + (size_t)(base / 16) * kSavedTileFloats + kSavedDist + vraw * 128 + c;
                            d_[0] = mu0[s]; d_[16] = mu1[s]; d_[32] = s0[s]; d_[48] = s1[s]; d_[64] = aw[s]; d_[80] = nu[s];
                        }
                    }
                    if (dbg_lane && pvalid && vraw < p.rfn) {
                        float* d_ = p.dbg + ((size_t)pidx * p.rfn + vraw) * kDbgFields;
                        d_[4] = hit[s]; d_[5] = vis[s]; d_[6] = mu0[s]; d_[7] = mu1[s]; d_[8] = s0[s]; d_[9] = s1[s];
                        d_[10] = aw[s]; d_[11] = nu[s];
                    }
                }
            }

            // ---------------- prob_embed (a13)                         aggregate_net.py:43 -----------------
            const auto W3 = phase_enter<PH_EMBED, HAS_VIS>(wl, W, seq0, true, wave_t, nw, lane);
            float e[NA1][8], gi[NA1][8], gr[NA1][3], sn[NA1];
            Opnd3<NA1, 8> e3;                                          // AR_X3: e split once (neuray_fc.0, and half of base_fc.0's operand)
            LayerPreT<L_NF1, AR> p_nf1; VecPre<L_NF2> p_nf2;
            if constexpr (NA > 0) {
                LayerPreT<L_PE1, AR> p_pe1; LayerPreT<L_RD1, AR> p_rd1; LayerPreT<L_RD2, AR> p_rd2; VecPre<L_RD2> p_rd2v;
                layer_prefetch<L_PE1>(W3, lane, p_pe1);
                {
                    float x1[NA][1];
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s)
                        x1[s][0] = sel4(g, (hit[s] - 0.5f) * 2.0f, (vis[s] - 0.5f) * 2.0f, 0.0f, 0.0f);
                    if constexpr (AR == AR_X3) {                     // (always the folded pack)
                        layer_fwd<L_PE1, NA, ACT_RELU>(W3, lane, p_pe1, o_fray, x1, e, p_rd1);
                    } else if (folded) {
                        // prob_embed.2 lives inside neuray_fc.0 / base_fc.0 of this pack: their input is the ReLU output itself
#ifdef NR_PROBE_PRE
                        layer_prefetch<L_RD1>(W3, lane, p_rd1);
                        NR_PRAGMA_UNROLL
                        for (int s = 0; s < NA; ++s)
                            NR_PRAGMA_UNROLL
                            for (int k = 0; k < 8; ++k) e[s][k] = fmaxf(fray[s][k] + x1[s][0], 0.0f);
#else
                        layer_fwd<L_PE1, NA, ACT_RELU>(W3, lane, p_pe1, fray, x1, e, p_rd1);
#endif
                    } else {
                        float h[NA][8];
                        LayerPre<L_PE2> p_pe2;
                        layer_fwd<L_PE1, NA, ACT_RELU>(W3, lane, p_pe1, fray, x1, h, p_pe2);
                        layer_fwd<L_PE2, NA, ACT_NONE>(W3, lane, p_pe2, h, none, e, p_rd1);
                    }
                }
                // ---------------- ray_dir_fc, rgb_feat + direction_feat     ibrnet.py:324-327 -----------------
                {
                    float x1[NA][1], h[NA][4], df[NA][8], dc[NA][3];
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) x1[s][0] = sel4(g, dlt[s][0], dlt[s][1], dlt[s][2], dlt[s][3]);
                    layer_fwd<L_RD1, NA, ACT_ELU>(W3, lane, p_rd1, o_none, x1, h, p_rd2);
                    layer_prefetch<L_RD2>(W3, lane, p_rd2v);
                    layer_fwd<L_RD2, NA, ACT_ELU>(W3, lane, p_rd2, operand<AR>(h), none, df, last);
                    layer_vec<L_RD2, NA>(p_rd2v, h, dc);          // the three rgb rows of ray_dir_fc.2
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) {
                        NR_PRAGMA_UNROLL
                        for (int k = 0; k < 8; ++k) gi[s][k] = fimg[s][k] + df[s][k];
                        NR_PRAGMA_UNROLL
                        for (int j = 0; j < 3; ++j) gr[s][j] = rgb[s][j] + elu(dc[s][j]);
                    }
                }
            }
            // ---------------- neuray_fc -> sigmoid                      ibrnet.py:337 -------------------------
            // (this phase also holds base_fc.0's per-view rows 0..31, used after the statistics below)
            const auto W4 = phase_enter<PH_NF_BV0, HAS_VIS>(wl, W, seq0, true, wave_t, nw, lane);
            if constexpr (NA > 0) {
                float h[NA][4], o[NA][1];
                layer_prefetch<L_NF1>(W4, lane, p_nf1);
                if constexpr (AR == AR_X3) { e3 = split_operand(e); layer_fwd<L_NF1, NA, ACT_ELU>(W4, lane, p_nf1, e3, none, h, p_nf2); }
                else layer_fwd<L_NF1, NA, ACT_ELU>(W4, lane, p_nf1, e, none, h, p_nf2);
                layer_vec<L_NF2, NA>(p_nf2, h, o);
                NR_PRAGMA_UNROLL
                for (int s = 0; s < NA; ++s) sn[s] = sigmoidf(o[s][0]);
            }
            // ---------------- cross-view weighted mean / variance       ibrnet.py:334-340 ---------------------
            // Each statistic is all-reduced over the views and immediately consumed by the owner waves as a K-slice of
            // base_fc.0's per-point part (columns 0..139), so the four 35-vectors never coexist.
            float msum, wv[NA1];
            v4f accv[NA1][4];
            {
                v4f accg[OWN][1];
                NR_PRAGMA_UNROLL
                for (int j = 0; j < OWN; ++j) {
                    const int mo = wave_t + j * nw;
                    float4 b;
                    if constexpr (BG_RES) b = bgres.b[j];
                    else b = wld4(W, gg * 16 + (mo < 4 ? mo : 0) * 64, bias_offset(L_BG, AR) * 4);      // (run-time tile index in the lane offset: see layer_tile_slice)
                    accg[j][0][0] = b.x; accg[j][0][1] = b.y; accg[j][0][2] = b.z; accg[j][0][3] = b.w;
                }
                float part[NA1][11], st[11], sv[11], wk[NA1];
                {   // k = 0, first all-reduce: sum(mask) rides along with the un-normalised weighted sum
                    //   weight = mask / (sum(mask) + 1e-8), weight0 = sigmoid(neuray_fc) * weight, mean0 = sum(x * weight0)
                    //   is evaluated as sum(x * sigmoid * mask) / (sum(mask) + 1e-8)
                    float p12[NA1][12], o12[12];
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) {
                        const float w_ = sn[s] * mask[s];
                        NR_PRAGMA_UNROLL
                        for (int q = 0; q < 8; ++q) p12[s][q] = gi[s][q] * w_;
                        NR_PRAGMA_UNROLL
                        for (int j = 0; j < 3; ++j) p12[s][8 + j] = gr[s][j] * w_;
                        p12[s][11] = mask[s];
                    }
                    view_allreduce<NA, IDLE, 12, RMAX, RED_SUM>(p12, o12, red, wave_t, nw, lane);
                    msum = o12[11];
                    const float inv = nr_fast_rcp(msum + 1e-8f);
                    NR_PRAGMA_UNROLL
                    for (int q = 0; q < 11; ++q) st[q] = o12[q] * inv;
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) wv[s] = mask[s] * nr_fast_rcp(msum + 1e-8f);
                    if constexpr (SAVE) {
                        if (wave_t == 0) p.saved[(size_t)(base / 16) * kSavedTileFloats + kSavedMsumRow * 64 + lane] = msum;
                    }
                }
                // k = 0: weight0 = sigmoid(neuray_fc) * weight  -> mean0, var0 ;  k = 1: weight -> mean1, var1
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 2; ++k) {
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) wk[s] = k == 0 ? sn[s] * wv[s] : wv[s];
                    if (k == 1) {
                        NR_PRAGMA_UNROLL
                        for (int s = 0; s < NA; ++s) {
                            NR_PRAGMA_UNROLL
                            for (int q = 0; q < 8; ++q) part[s][q] = gi[s][q] * wk[s];
                            NR_PRAGMA_UNROLL
                            for (int j = 0; j < 3; ++j) part[s][8 + j] = gr[s][j] * wk[s];
                        }
                        view_allreduce<NA, IDLE, 11, RMAX, RED_SUM>(part, st, red, wave_t, nw, lane);
                    }
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) {
                        NR_PRAGMA_UNROLL
                        for (int q = 0; q < 8; ++q) { const float d_ = gi[s][q] - st[q]; part[s][q] = wk[s] * (d_ * d_); }
                        NR_PRAGMA_UNROLL
                        for (int j = 0; j < 3; ++j) { const float d_ = gr[s][j] - st[8 + j]; part[s][8 + j] = wk[s] * (d_ * d_); }
                    }
                    if constexpr (BG_RES) {
                        if (k == 0) bg_accumulate<NT, OWN, 0>(bgres, g, wave_t, nw, st, accg);
                        else bg_accumulate<NT, OWN, 2>(bgres, g, wave_t, nw, st, accg);
                    } else {
                        if (k == 0) bg_accumulate<NT, OWN, 0>(W, glane, g, wave_t, nw, st, accg);
                        else bg_accumulate<NT, OWN, 2>(W, glane, g, wave_t, nw, st, accg);
                    }
                    view_allreduce<NA, IDLE, 11, RMAX, RED_SUM>(part, sv, red, wave_t, nw, lane);
                    if constexpr (BG_RES) {
                        if (k == 0) bg_accumulate<NT, OWN, 1>(bgres, g, wave_t, nw, sv, accg);
                        else bg_accumulate<NT, OWN, 3>(bgres, g, wave_t, nw, sv, accg);
                    } else {
                        if (k == 0) bg_accumulate<NT, OWN, 1>(W, glane, g, wave_t, nw, sv, accg);
                        else bg_accumulate<NT, OWN, 3>(W, glane, g, wave_t, nw, sv, accg);
                    }
                    if constexpr (SAVE) {                          // every wave_t holds the statistics: wave_t 2k % nw writes the mean, (2k + 1) % nw the variance
                        float* d_ = p.saved + (size_t)(base / 16) * kSavedTileFloats + (kSavedStatRow + 22 * k) * 64 + lane;
                        if (wave_t == (2 * k) % nw) {
                            NR_PRAGMA_UNROLL
                            for (int q = 0; q < 11; ++q) d_[q * 64] = st[q];
                        }
                        if (wave_t == (2 * k + 1) % nw) {
                            NR_PRAGMA_UNROLL
                            for (int q = 0; q < 11; ++q) d_[(11 + q) * 64] = sv[q];
                        }
                    }
                }
                NR_PRAGMA_UNROLL
                for (int j = 0; j < OWN; ++j) {
                    const int mo = wave_t + j * nw;
                    if (mo < 4) {
                        NR_PRAGMA_UNROLL
                        for (int r = 0; r < 4; ++r) xrow(mo, r)[lane] = accg[j][0][r];
                        if constexpr (SAVE) {
                            NR_PRAGMA_UNROLL
                            for (int r = 0; r < 4; ++r)
                                p.saved[(size_t)(base / 16) * kSavedTileFloats + (kSavedBgRow + mo * 4 + r) * 64 + lane] = accg[j][0][r];
                        }
                    }
                }
                NR_BLOCK_SYNC();
                NR_PRAGMA_UNROLL
                for (int s = 0; s < NA; ++s)
                    NR_PRAGMA_UNROLL
                    for (int mo = 0; mo < (AR == AR_X3 ? 2 : 4); ++mo)    // (AR_X3: tiles 2, 3 are read when base_fc.0's second half starts)
                        NR_PRAGMA_UNROLL
                        for (int r = 0; r < 4; ++r) accv[s][mo][r] = xrow(mo, r)[lane];
            }
            // ---------------- base_fc per-view part, vis_fc, vis_fc2, rgb_fc   ibrnet.py:342-349,363-365 ------
            float x[NA1][8], vis2[NA1], z[NA1];
            constexpr bool GF_PRE = BG_RES && (NR_X3_GF_PREFETCH != 0);
            float4 gfq[4]; float gfs = 0.0f;
            typename ws_lds<decltype(make_w())>::type W5;
            LayerPreT<L_VF1, AR> p_vf1;
            {
                float xq[NA1][16], x1[NA1][1], h64[NA1][16];
                Opnd3<NA1, 16> xq3;                                    // AR_X3: xq split once for base_fc.0's two halves
                LayerPreT<L_BV0, AR> p_bv0; LayerPreT<L_BV1, AR> p_bv1; LayerPreT<L_B2, AR> p_b2;
                v4f acch[NA1][2];
                if constexpr (NA > 0) {
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) {
                        NR_PRAGMA_UNROLL
                        for (int k = 0; k < 8; ++k) { xq[s][k] = gi[s][k]; xq[s][8 + k] = e[s][k]; }
                        x1[s][0] = sel4(g, gr[s][0], gr[s][1], gr[s][2], 0.0f);
                    }
                    layer_prefetch<L_BV0>(W4, lane, p_bv0);
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) { acch[s][0] = accv[s][0]; acch[s][1] = accv[s][1]; }
                    if constexpr (AR == AR_X3) {
                        const Opnd3<NA1, 8> gi3 = split_operand(gi);
                        NR_PRAGMA_UNROLL
                        for (int pt = 0; pt < 3; ++pt)
                            NR_PRAGMA_UNROLL
                            for (int s = 0; s < NA; ++s)
                                NR_PRAGMA_UNROLL
                                for (int i = 0; i < 4; ++i) { xq3.p[pt][s][i] = gi3.p[pt][s][i]; xq3.p[pt][s][4 + i] = e3.p[pt][s][i]; }
                        layer_acc<L_BV0, NA>(W4, lane, p_bv0, xq3, x1, acch, last);
                    } else layer_acc<L_BV0, NA>(W4, lane, p_bv0, xq, x1, acch, last);
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s)
                        NR_PRAGMA_UNROLL
                        for (int mo = 0; mo < 2; ++mo)
                            NR_PRAGMA_UNROLL
                            for (int r = 0; r < 4; ++r) h64[s][4 * mo + r] = elu_s(acch[s][mo][r]);   // kOutScaled[L_BV0]
                }
                const auto W4b = phase_enter<PH_BV1, HAS_VIS>(wl, W, seq0, true, wave_t, nw, lane);
                if constexpr (NA > 0) {
                    layer_prefetch<L_BV1>(W4b, lane, p_bv1);
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) {
                        if constexpr (AR == AR_X3) {
                            NR_PRAGMA_UNROLL
                            for (int mo = 0; mo < 2; ++mo)
                                NR_PRAGMA_UNROLL
                                for (int r = 0; r < 4; ++r) acch[s][mo][r] = xrow(2 + mo, r)[lane];
                        } else { acch[s][0] = accv[s][2]; acch[s][1] = accv[s][3]; }
                    }
                    if constexpr (AR == AR_X3) layer_acc<L_BV1, NA>(W4b, lane, p_bv1, xq3, x1, acch, last);
                    else layer_acc<L_BV1, NA>(W4b, lane, p_bv1, xq, x1, acch, last);
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s)
                        NR_PRAGMA_UNROLL
                        for (int mo = 0; mo < 2; ++mo)
                            NR_PRAGMA_UNROLL
                            for (int r = 0; r < 4; ++r) h64[s][8 + 4 * mo + r] = elu_s(acch[s][mo][r]);   // kOutScaled[L_BV1]
                }
                W5 = phase_enter<PH_B2_VF1, HAS_VIS>(wl, W, seq0, true, wave_t, nw, lane);
                if constexpr (NA > 0) {
                    layer_prefetch<L_B2>(W5, lane, p_b2);
                    layer_fwd<L_B2, NA, ACT_ELU>(W5, lane, p_b2, operand<AR>(h64), none, x, p_vf1);
                }
            }
            {
                float xin[NA1][8], h[NA1][8];
                if constexpr (NA > 0) {
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s)
                        NR_PRAGMA_UNROLL
                        for (int k = 0; k < 8; ++k) xin[s][k] = x[s][k] * wv[s];
                    layer_fwd<L_VF1, NA, ACT_ELU>(W5, lane, p_vf1, operand<AR>(xin), none, h, last);
                }
                W5 = phase_enter<PH_TAIL, HAS_VIS>(wl, W, seq0, more, wave_t, nw, lane);
                if constexpr (GF_PRE) {
                    // geometry_fc.0's fragments of the tile this wave owns are fetched HERE, a phase ahead of their use behind the last two
                    // all-reduces, where the wave (and at the next barrier its siblings) used to wait for them (AR_X3: registers to spare)
                    const int m = wave_t < 4 ? wave_t : 0;
                    NR_PRAGMA_UNROLL
                    for (int kq = 0; kq < 4; ++kq) gfq[kq] = wld4(W, glane * 16 + m * (4 * 1024), (quads_offset(L_GF1, AR) + kq * 256) * 4);
                    gfs = wld1(W, glane * 4 + m * 256, single_offset(L_GF1, AR) * 4);
                }
                if constexpr (NA > 0) {
                    float y[NA][8], yv[NA][1], o[NA][1];
                    LayerPreT<L_VF2, AR> p_vf2; LayerPreT<L_V21, AR> p_v21; LayerPreT<L_RF1, AR> p_rf1; LayerPreT<L_RF2, AR> p_rf2;
                    VecPre<L_VF2> p_vf2v; VecPre<L_V22> p_v22; VecPre<L_RF3> p_rf3;
                    layer_prefetch<L_VF2>(W5, lane, p_vf2);
                    layer_prefetch<L_VF2>(W5, lane, p_vf2v);
                    layer_fwd<L_VF2, NA, ACT_ELU>(W5, lane, p_vf2, operand<AR>(h), none, y, p_v21);
                    layer_vec<L_VF2, NA>(p_vf2v, h, yv);          // row 32: the visibility logit
                    float visp[NA];
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) {
                        visp[s] = sigmoidf(elu(yv[s][0])) * mask[s];      // sigmoid on an ELU output: quirk A.9.4
                        NR_PRAGMA_UNROLL
                        for (int k = 0; k < 8; ++k) { x[s][k] = x[s][k] + y[s][k]; xin[s][k] = x[s][k] * visp[s]; }
                    }
                    Opnd3<NA, 8> x3;                                   // AR_X3: x (after the vis_fc update) split once, for vis_fc2.0 and rgb_fc.0
                    if constexpr (AR == AR_X3) {
                        x3 = split_operand(x);
                        layer_fwd_scaled<L_V21, NA, ACT_ELU>(W5, lane, p_v21, x3, visp, h, p_v22);       // W (x vis') = vis' (W x)
                    } else layer_fwd<L_V21, NA, ACT_ELU>(W5, lane, p_v21, xin, none, h, p_v22);
                    layer_prefetch<L_RF1>(W5, lane, p_rf1);
                    layer_vec<L_V22, NA>(p_v22, h, o);
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) vis2[s] = sigmoidf(o[s][0]) * mask[s];
                    float x1[NA][2], h16[NA][4], h8[NA][4];
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) {
                        x1[s][0] = sel4(g, vis2[s], dlt[s][0], dlt[s][1], dlt[s][2]);
                        x1[s][1] = sel4(g, dlt[s][3], 0.0f, 0.0f, 0.0f);
                    }
                    if constexpr (AR == AR_X3) layer_fwd<L_RF1, NA, ACT_ELU>(W5, lane, p_rf1, x3, x1, h16, p_rf2);
                    else layer_fwd<L_RF1, NA, ACT_ELU>(W5, lane, p_rf1, x, x1, h16, p_rf2);
                    layer_fwd<L_RF2, NA, ACT_ELU>(W5, lane, p_rf2, operand<AR>(h16), none, h8, p_rf3);
                    layer_vec<L_RF3, NA>(p_rf3, h8, o);
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NA; ++s) {
                        z[s] = mask[s] > 0.0f ? o[s][0] : -1e9f;
                        const int vraw = wave_t * VPW + s;
                        if (dbg_lane && pvalid && vraw < p.rfn) {
                            float* d_ = p.dbg + ((size_t)pidx * p.rfn + vraw) * kDbgFields;
                            d_[12] = sn[s]; d_[13] = visp[s]; d_[14] = vis2[s]; d_[15] = z[s];
                        }
                    }
                }
            }
            // ---------------- cross-view: blending softmax, visibility-weighted mean/var  ibrnet.py:350-367 ---
            // Two all-reduces: {max z, sum vis''} and {sum wh x (8), sum e (1), sum e rgb (3)} with e = exp(z - max z); the
            // softmax blend sum(rgb * e / sum e) is evaluated as sum(rgb * e) / sum(e), the mean weight sum(wh) / rfn from
            // sum(vis'') directly.
            constexpr int ZR = SAVE ? 3 : 2;                       // (training: sum of weight0 = sn * weight rides along for the backward)
            float zv[ZR], big[12], wh[NA1], meanw;
            {
                float z2[NA1][ZR];
                NR_PRAGMA_UNROLL
                for (int s = 0; s < NA; ++s) {
                    z2[s][0] = z[s]; z2[s][1] = vis2[s];
                    if constexpr (SAVE) z2[s][2] = sn[s] * wv[s];
                }
                view_allreduce<NA, IDLE, ZR, RMAX, RED_MAX0>(z2, zv, red, wave_t, nw, lane);
                float b12[NA1][12];
                NR_PRAGMA_UNROLL
                for (int s = 0; s < NA; ++s) {
                    const float ev = nr_fast_exp(z[s] - zv[0]);
                    wh[s] = vis2[s] * nr_fast_rcp(zv[1] + 1e-8f);
                    NR_PRAGMA_UNROLL
                    for (int k = 0; k < 8; ++k) b12[s][k] = x[s][k] * wh[s];
                    b12[s][8] = ev;
                    NR_PRAGMA_UNROLL
                    for (int j = 0; j < 3; ++j) b12[s][9 + j] = rgb[s][j] * ev;
                }
                if constexpr (IDLE > 0) {
                    // the softmax terms of the skipped slots: exp(-1e9 - max z) each (0, or 1 where every view of the point is masked: the
                    // uniform softmax over zero colours of the reference); their colour / feature products are zeros
                    const float ev_idle = nr_fast_exp(-1e9f - zv[0]);
                    if constexpr (NA == 0) {
                        NR_PRAGMA_UNROLL
                        for (int q = 0; q < 12; ++q) b12[0][q] = 0.0f;
                        float acc_ = ev_idle;
                        NR_PRAGMA_UNROLL
                        for (int i = 1; i < IDLE; ++i) acc_ += ev_idle;
                        b12[0][8] = acc_;
                        view_allreduce<1, 0, 12, RMAX, RED_SUM>(b12, big, red, wave_t, nw, lane);
                    } else {
                        NR_PRAGMA_UNROLL
                        for (int i = 0; i < IDLE; ++i) b12[NA - 1][8] += ev_idle;
                        view_allreduce<NA, 0, 12, RMAX, RED_SUM>(b12, big, red, wave_t, nw, lane);
                    }
                } else {
                    view_allreduce<NA, 0, 12, RMAX, RED_SUM>(b12, big, red, wave_t, nw, lane);
                }
                {
                    const float ie = nr_fast_rcp(big[8]);
                    big[9] *= ie; big[10] *= ie; big[11] *= ie;
                    meanw = zv[1] * nr_fast_rcp(zv[1] + 1e-8f);
                }
                if constexpr (SAVE) {
                    float* d_ = p.saved + (size_t)(base / 16) * kSavedTileFloats + lane;
                    if (wave_t == 0) {
                        NR_PRAGMA_UNROLL
                        for (int k = 0; k < 8; ++k) d_[(kSavedGmeanRow + k) * 64] = big[k];
                    }
                    if (wave_t == 2 % nw) {
                        d_[kSavedZmaxRow * 64] = zv[0]; d_[kSavedSezRow * 64] = big[8];
                        d_[kSavedSvisRow * 64] = zv[1]; d_[kSavedSw0Row * 64] = zv[ZR - 1];
                    }
                }
            }
            // geometry_fc.0 (a14): owner waves stream the mean part, then the variance part   ibrnet.py:353-354
            v4f accf[OWN][1];
            NR_PRAGMA_UNROLL
            for (int j = 0; j < OWN; ++j) {
                const int mo = wave_t + j * nw;
                const float4 b = wld4(W, gg * 16 + (mo < 4 ? mo : 0) * 64, bias_offset(L_GF1, AR) * 4);
                accf[j][0][0] = b.x; accf[j][0][1] = b.y; accf[j][0][2] = b.z; accf[j][0][3] = b.w;
            }
            float var[8];
            {
                float xq[1][8], x1[1][1], v8[NA1][8];
                NR_PRAGMA_UNROLL
                for (int s = 0; s < NA; ++s)
                    NR_PRAGMA_UNROLL
                    for (int k = 0; k < 8; ++k) { const float d_ = x[s][k] - big[k]; v8[s][k] = wh[s] * (d_ * d_); }
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) xq[0][k] = big[k];
                x1[0][0] = sel4(g, meanw * inv_rfn, 0.0f, 0.0f, 0.0f);
                NR_PRAGMA_UNROLL
                for (int j = 0; j < OWN; ++j) {
                    const int mo = wave_t + j * nw;
                    if constexpr (GF_PRE) {
                        if (mo < 4) {                 // (layer_tile_slice's order: the slice's two quads, then its single)
                            mfma_quad<NT>(gfq[0], 0, xq, accf[j]);
                            mfma_quad<NT>(gfq[1], 1, xq, accf[j]);
                            accf[j][0] = nr_mfma16(gfs, x1[0][0], accf[j][0]);
                        }
                    } else if (mo < 4) layer_tile_slice<L_GF1, NT, 0, 2, 0, 1>(W, glane, mo, xq, x1, accf[j]);
                }
                view_allreduce<NA, IDLE, 8, RMAX, RED_SUM>(v8, var, red, wave_t, nw, lane);
                if constexpr (SAVE) {
                    if (wave_t == 1 % nw) {
                        NR_PRAGMA_UNROLL
                        for (int k = 0; k < 8; ++k) p.saved[(size_t)(base / 16) * kSavedTileFloats + (kSavedGvarRow + k) * 64 + lane] = var[k];
                    }
                }
            }
            {
                float xq[1][8], nonet[1][1];
                nonet[0][0] = 0.0f;
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) xq[0][k] = var[k];
                NR_PRAGMA_UNROLL
                for (int j = 0; j < OWN; ++j) {
                    const int mo = wave_t + j * nw;
                    if (mo < 4) {
                        if constexpr (GF_PRE) { mfma_quad<NT>(gfq[2], 0, xq, accf[j]); mfma_quad<NT>(gfq[3], 1, xq, accf[j]); }
                        else layer_tile_slice<L_GF1, NT, 2, 2, 0, 0>(W, glane, mo, xq, nonet, accf[j]);
                        NR_PRAGMA_UNROLL
                        for (int r = 0; r < 4; ++r) xrow(mo, r)[lane] = elu_s(accf[j][0][r]);   // kOutScaled[L_GF1]
                        if constexpr (SAVE) {
                            NR_PRAGMA_UNROLL
                            for (int r = 0; r < 4; ++r)
                                p.saved[(size_t)(base / 16) * kSavedTileFloats + (kSavedGeoRow + mo * 4 + r) * 64 + lane] = elu_s(accf[j][0][r]);
                        }
                    }
                }
            }
            NR_BLOCK_SYNC();
            if (wave_t == 0) {
                float h[1][16], G[1][4], nonet[1][1];
                nonet[0][0] = 0.0f;
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) h[0][k] = xrow(k >> 2, k & 3)[lane];
                layer_fwd<L_GF2, NT, ACT_ELU>(W, glane, h, nonet, G);
                if (pvalid)
                    *reinterpret_cast<float4*>(p.point_out + (size_t)pidx * kPointRec + 4 * g) = make_float4(G[0][0], G[0][1], G[0][2], G[0][3]);
                if constexpr (SAVE) {
                    NR_PRAGMA_UNROLL
                    for (int r = 0; r < 4; ++r) p.saved[(size_t)(base / 16) * kSavedTileFloats + (kSavedGRow + r) * 64 + lane] = G[0][r];
                }
            }
            if (wave_t == (nw > 1 ? 1 : 0) && g == 0) {
                if (pvalid)
                    *reinterpret_cast<float4*>(p.point_out + (size_t)pidx * kPointRec + 16) = make_float4(big[9], big[10], big[11], msum);
            }
        };

        if constexpr (SKIP) {
            const bool a0 = __ballot(mask[0] != 0.0f) != 0ull, a1 = __ballot(mask[1] != 0.0f) != 0ull;
#ifdef NR_FORCE_NA          // timing probe (wrong results): every wave_t takes the NA-slot body
#ifdef NR_FORCE_NA_CONST
            const int na = NR_FORCE_NA;
#else
            const int na = NR_FORCE_NA + (p.rn < 0 ? (a0 ? 1 : 0) + (a1 ? 1 : 0) : 0);
#endif
#else
            const int na = (a0 ? 1 : 0) + (a1 ? 1 : 0);
#endif
            n_active += na; n_slots += NS;
            if (na == 2) {
                tile(SlotCount<2>{});
            } else if (na == 1) {
                if (!a0) {          // the lone active slot becomes slot 0
                    mask[0] = mask[1]; tref[0] = tref[1]; pu[0] = pu[1]; pv[0] = pv[1]; soff_f[0] = soff_f[1]; soff_c[0] = soff_c[1];
                    NR_PRAGMA_UNROLL
                    for (int k = 0; k < 4; ++k) dlt[0][k] = dlt[1][k];
                }
                tile(SlotCount<1>{});
            } else {
                tile(SlotCount<0>{});
            }
        } else {
            tile(SlotCount<NS>{});
        }
    }
    if constexpr (SKIP) {
        if (p.slot_stats) {                 // (statistics launches only: one pair of atomics per workgroup, not per wave)
            NR_BLOCK_SYNC();
            if (lane == 0) { red[2 * wave] = (float)n_active; red[2 * wave + 1] = (float)n_slots; }
            NR_BLOCK_SYNC();
            if (threadIdx.x == 0) {
                float a = 0.0f, b = 0.0f;
                for (int w = 0; w < nw; ++w) { a += red[2 * w]; b += red[2 * w + 1]; }
                atomicAdd(p.slot_stats, (unsigned long long)a);
                atomicAdd(p.slot_stats + 1, (unsigned long long)b);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// ray kernel: one wave per ray.  + positional encoding, 4-head self attention over the dn samples,
// LayerNorm, sigma head, alpha compositing, ray mask, depth.       ibrnet.py:52-102,356-360;
// renderer.py:163-165,195-202; render_ops.py:72-80
// -------------------------------------------------------------------------------------------------
struct RayParams {
    const float* point_rec;   // [rn][dn][kPointRec]
    const float* depth;       // [rn][dn]
    const float* pos_enc;     // [dn][16]
    const float* weights;     // packed pass weights (ray part at kPackedPointFloats)
    float* hit_prob;          // [rn][dn]
    float* pixel;             // [rn][3]
    float* render_depth;      // [rn] or null
    unsigned char* ray_mask;  // [rn] or null
    float* density;           // [rn][dn] or null (debug / tests)
    float* att_save;          // training (SAVE kernel): [rn][dn][kRayAttSave] softmax shift (4), 1 / denominator (4), attention output (16)
    int rn, dn, mask_view_num, mask_point_num;
};
constexpr int kRayAttSave = 24;

constexpr int kRayWaves = 4;
// LDS: attention / sigma-head weights (shared by the workgroup) + per wave K, V, transmittance factors, alpha
inline size_t ray_smem_bytes(int dn, int rays_per_wave = 1) { return sizeof(float) * (kPackedRayFloats + 12 + kRayWaves * rays_per_wave * ((size_t)dn * 34)); }

__device__ __forceinline__ float wave_sum(float v) {
    NR_PRAGMA_UNROLL
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
    NR_PRAGMA_UNROLL
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// y[16] = M[16x16] x[16] with M row-major in LDS (every lane reads the same address: broadcast, conflict-free)
__device__ __forceinline__ void matvec16(const float* __restrict__ M, const float (&x)[16], float (&y)[16]) {
    NR_PRAGMA_UNROLL
    for (int o = 0; o < 16; ++o) {
        float s = 0.0f;
        NR_PRAGMA_UNROLL
        for (int k4 = 0; k4 < 4; ++k4) {
            const float4 w = ld4(M + o * 16 + 4 * k4);
            s = fmaf(w.x, x[4 * k4], s); s = fmaf(w.y, x[4 * k4 + 1], s);
            s = fmaf(w.z, x[4 * k4 + 2], s); s = fmaf(w.w, x[4 * k4 + 3], s);
        }
        y[o] = s;
        if ((o & 1) == 1) NR_PIN();     // bound the scheduler's load clustering (it would hold all 64 rows in VGPRs)
    }
}

// sum / max over the 64 / RPW lanes of a ray's segment of the wave
template <int RPW> __device__ __forceinline__ float seg_sum(float v) {
    NR_PRAGMA_UNROLL
    for (int m = 32 / RPW; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
template <int RPW> __device__ __forceinline__ float seg_max(float v) {
    NR_PRAGMA_UNROLL
    for (int m = 32 / RPW; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// RPW = rays per wave: 1 (any dn; lane = sample, 64 samples per chunk) or 2 (dn <= 32: the two halves of the wave take one ray each - a
// 32-sample fine pass otherwise idles half of every wave).  The arithmetic per ray is the same in both.
template <bool SAVE, int RPW = 1>
__global__ void __launch_bounds__(256, 4) rays_kernel(RayParams p) {
    NR_DYNAMIC_SMEM(float, smem);
    constexpr int SEG = 64 / RPW;
    const int lane = threadIdx.x & 63;
    const int wave = NR_UNIFORM((int)(threadIdx.x >> 6));
    const int seg = RPW == 2 ? (lane >> 5) : 0, li = RPW == 2 ? (lane & 31) : lane;      // the ray of this lane inside the wave, its lane index there
    const int dn = p.dn;
    float* RW = smem + nr_opaque_zero();               // [kPackedRayFloats] (+ pad to a 16-byte multiple)
    float* ks = smem + kPackedRayFloats + 12 + (size_t)(wave * RPW + seg) * (dn * 34);
    float* vs = ks + dn * 16;
    float* tr = vs + dn * 16;        // [dn] transmittance factors
    float* al = tr + dn;             // [dn] alpha
    for (int i = threadIdx.x; i < kPackedRayFloats; i += blockDim.x) RW[i] = p.weights[kPackedPointFloats + i];
    __syncthreads();
    const int nch = RPW == 2 ? 1 : (dn + 63) >> 6;
    const int nray_iter = (p.rn + kRayWaves * RPW - 1) / (kRayWaves * RPW);

    for (int it = blockIdx.x; it < nray_iter; it += gridDim.x) {
        int ray = (it * kRayWaves + wave) * RPW + seg;
        const bool rvalid = ray < p.rn;
        ray = rvalid ? ray : p.rn - 1;
        const float* rec = p.point_rec + (size_t)ray * dn * kPointRec;
        // ---- phase 1: K, V of every sample -> LDS
        // (compiler barrier: keeps hipcc from hoisting the uniform LDS weight reads out of the ray loop into SGPRs)
        asm volatile("" ::: "memory");
        float kn2[4] = {0.0f, 0.0f, 0.0f, 0.0f};      // per head: largest squared key norm of the ray (this lane's samples)
        for (int ch = 0; ch < nch; ++ch) {
            const int i = ch * 64 + li;
            if (i < dn) {
                float G[16], y[16];
                NR_PRAGMA_UNROLL
                for (int k4 = 0; k4 < 4; ++k4) {
                    const float4 a = ld4(rec + (size_t)i * kPointRec + 4 * k4), b = ld4(p.pos_enc + i * 16 + 4 * k4);
                    G[4 * k4] = a.x + b.x; G[4 * k4 + 1] = a.y + b.y; G[4 * k4 + 2] = a.z + b.z; G[4 * k4 + 3] = a.w + b.w;
                }
                matvec16(RW + RW_WK, G, y);
                NR_PRAGMA_UNROLL
                for (int k4 = 0; k4 < 4; ++k4) {
                    *reinterpret_cast<float4*>(ks + i * 16 + 4 * k4) = make_float4(y[4 * k4], y[4 * k4 + 1], y[4 * k4 + 2], y[4 * k4 + 3]);
                    kn2[k4] = fmaxf(kn2[k4], fmaf(y[4 * k4 + 3], y[4 * k4 + 3], fmaf(y[4 * k4 + 2], y[4 * k4 + 2],
                                    fmaf(y[4 * k4 + 1], y[4 * k4 + 1], y[4 * k4] * y[4 * k4]))));
                }
                matvec16(RW + RW_WV, G, y);
                NR_PRAGMA_UNROLL
                for (int k4 = 0; k4 < 4; ++k4)
                    *reinterpret_cast<float4*>(vs + i * 16 + 4 * k4) = make_float4(y[4 * k4], y[4 * k4 + 1], y[4 * k4 + 2], y[4 * k4 + 3]);
            }
        }
        NR_PRAGMA_UNROLL
        for (int hh = 0; hh < 4; ++hh) kn2[hh] = seg_max<RPW>(kn2[hh]);
        NR_WAVE_SYNC();         // (K, V, alpha and the transmittance factors are this wave's own LDS)
        // ---- phase 2: attention row, LayerNorm, sigma, alpha
        for (int ch = 0; ch < nch; ++ch) {
            const int iraw = ch * 64 + li;
            const bool act = iraw < dn;               // lanes past the last sample redo sample dn-1 and store nothing, so
            const int i = act ? iraw : dn - 1;        // that the wave stays converged for the ballot below
            asm volatile("" ::: "memory");
            {
                float G[16], q[16], o[16];
                NR_PRAGMA_UNROLL
                for (int k4 = 0; k4 < 4; ++k4) {
                    const float4 a = ld4(rec + (size_t)i * kPointRec + 4 * k4), b = ld4(p.pos_enc + i * 16 + 4 * k4);
                    G[4 * k4] = a.x + b.x; G[4 * k4 + 1] = a.y + b.y; G[4 * k4 + 2] = a.z + b.z; G[4 * k4 + 3] = a.w + b.w;
                }
                const float nvalid = rec[(size_t)i * kPointRec + 19];
                matvec16(RW + RW_WQ, G, q);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) q[k] = q[k] / 2.0f;      // temperature = sqrt(d_k) = 2 (exact)
                // query-row mask (quirk A.9.3): every score of the row becomes -1e9, i.e. the softmax is uniform - exactly
                // what a zero query gives, so the row is handled by zeroing q instead of a select per (query, key)
                if (!(nvalid > 1.0f)) {
                    NR_PRAGMA_UNROLL
                    for (int k = 0; k < 16; ++k) q[k] = 0.0f;
                }
                // softmax shift: any c >= max_j s_ij gives the same softmax; c = |q_i| * max_j |k_j| (Cauchy-Schwarz)
                // needs no pass over the keys.  exp(s - c) cannot overflow; it could underflow for every key only if
                // c - max_j s_ij > 87, so rows with c > 40 (never seen with trained or kaiming weights, exercised by
                // tests/test_edge_cases.py) take the exact two-pass form.
                float cb[4], a_sh[4], a_rd[4];                 // (a_sh / a_rd: the shift and 1 / denominator actually used, for the backward)
                bool fast = true;
                NR_PRAGMA_UNROLL
                for (int hh = 0; hh < 4; ++hh) {
                    const float qn2 = fmaf(q[hh * 4 + 3], q[hh * 4 + 3], fmaf(q[hh * 4 + 2], q[hh * 4 + 2],
                                           fmaf(q[hh * 4 + 1], q[hh * 4 + 1], q[hh * 4] * q[hh * 4])));
                    cb[hh] = sqrtf(qn2 * kn2[hh]) * 1.0000005f;      // (rounding slack: the bound must not fall below the max)
                    fast = fast && (cb[hh] <= 40.0f);
                }
                if (__ballot(!fast) == 0ull) {
                    NR_PRAGMA_UNROLL
                    for (int hh = 0; hh < 4; ++hh) {
                        float den = 0.0f, a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                        for (int j = 0; j < dn; ++j) {
                            const float4 kj = ld4(ks + j * 16 + hh * 4);
#ifdef NR_RAY_PROBE_NOSCORE     // timing probe (wrong results): what the QK^T products cost at most
                            const float s = q[hh * 4] * kj.x;
#else
                            const float s = fmaf(q[hh * 4 + 3], kj.w, fmaf(q[hh * 4 + 2], kj.z, fmaf(q[hh * 4 + 1], kj.y, q[hh * 4] * kj.x)));
#endif
                            const float e_ = nr_fast_exp(s - cb[hh]);
                            const float4 vj = ld4(vs + j * 16 + hh * 4);
                            den += e_; a0 = fmaf(e_, vj.x, a0); a1 = fmaf(e_, vj.y, a1); a2 = fmaf(e_, vj.z, a2); a3 = fmaf(e_, vj.w, a3);
                        }
                        o[hh * 4] = a0 / den; o[hh * 4 + 1] = a1 / den; o[hh * 4 + 2] = a2 / den; o[hh * 4 + 3] = a3 / den;
                        if constexpr (SAVE) { a_sh[hh] = cb[hh]; a_rd[hh] = 1.0f / den; }
                    }
                } else {
                    NR_PRAGMA_UNROLL
                    for (int hh = 0; hh < 4; ++hh) {
                        float mx = -INFINITY;
                        for (int j = 0; j < dn; ++j) {
                            const float4 kj = ld4(ks + j * 16 + hh * 4);
                            const float s = fmaf(q[hh * 4 + 3], kj.w, fmaf(q[hh * 4 + 2], kj.z, fmaf(q[hh * 4 + 1], kj.y, q[hh * 4] * kj.x)));
                            mx = fmaxf(mx, s);
                        }
                        float den = 0.0f, a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                        for (int j = 0; j < dn; ++j) {
                            const float4 kj = ld4(ks + j * 16 + hh * 4);
                            const float s = fmaf(q[hh * 4 + 3], kj.w, fmaf(q[hh * 4 + 2], kj.z, fmaf(q[hh * 4 + 1], kj.y, q[hh * 4] * kj.x)));
                            const float e_ = nr_fast_exp(s - mx);
                            const float4 vj = ld4(vs + j * 16 + hh * 4);
                            den += e_; a0 = fmaf(e_, vj.x, a0); a1 = fmaf(e_, vj.y, a1); a2 = fmaf(e_, vj.z, a2); a3 = fmaf(e_, vj.w, a3);
                        }
                        o[hh * 4] = a0 / den; o[hh * 4 + 1] = a1 / den; o[hh * 4 + 2] = a2 / den; o[hh * 4 + 3] = a3 / den;
                        if constexpr (SAVE) { a_sh[hh] = mx; a_rd[hh] = 1.0f / den; }
                    }
                }
                if constexpr (SAVE) {
                    if (act && rvalid) {
                        float* d_ = p.att_save + ((size_t)ray * dn + i) * kRayAttSave;
                        *reinterpret_cast<float4*>(d_) = make_float4(a_sh[0], a_sh[1], a_sh[2], a_sh[3]);
                        *reinterpret_cast<float4*>(d_ + 4) = make_float4(a_rd[0], a_rd[1], a_rd[2], a_rd[3]);
                        NR_PRAGMA_UNROLL
                        for (int k4 = 0; k4 < 4; ++k4)
                            *reinterpret_cast<float4*>(d_ + 8 + 4 * k4) = make_float4(o[4 * k4], o[4 * k4 + 1], o[4 * k4 + 2], o[4 * k4 + 3]);
                    }
                }
                float y[16], mean = 0.0f;
                matvec16(RW + RW_FC, o, y);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) { y[k] += G[k]; mean += y[k]; }
                mean /= 16.0f;
                float var = 0.0f;
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) { const float d_ = y[k] - mean; var = fmaf(d_, d_, var); }
                var /= 16.0f;
                const float rstd = 1.0f / sqrtf(var + 1e-6f);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) y[k] = fmaf((y[k] - mean) * rstd, RW[RW_LNW + k], RW[RW_LNB + k]);
                matvec16(RW + RW_OG0W, y, o);
                float sg = RW[RW_OG2B];
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) sg = fmaf(RW[RW_OG2W + k], elu(o[k] + RW[RW_OG0B + k]), sg);
                sg = fmaxf(sg, 0.0f);
                if (nvalid < 1.0f) sg = 0.0f;
                if (p.density && rvalid && act) p.density[(size_t)ray * dn + i] = sg;
                const float alpha = 1.0f - expf(-fmaxf(sg, 0.0f));
                if (act) { al[i] = alpha; tr[i] = (1.0f - alpha) + 1e-10f; }
            }
        }
        NR_WAVE_SYNC();         // (K, V, alpha and the transmittance factors are this wave's own LDS)
        // ---- phase 3: compositing.  Transmittance = exclusive prefix product over the samples: a wavefront-shuffle scan (north_star;
        // round 3).  -DNR_SEQ_COMPOSITE: the round-1/2 form, every lane multiplying its own prefix in torch.cumprod's order.
        float cr = 0.0f, cg = 0.0f, cb = 0.0f, cd = 0.0f;
        int cnt = 0;
        for (int ch = 0; ch < nch; ++ch) {
            const int i = ch * 64 + li;
            const bool ok = i < dn;
            float T = 1.0f;
#ifndef NR_SEQ_COMPOSITE
            // exclusive prefix product over the lanes of the ray's segment in log2 shuffle steps (a second chunk of a > 64-sample ray
            // carries the first chunk's product).  The product associates as a tree, so hit_prob can differ from torch.cumprod's order
            // in the last bit or two (relative 1e-7; the gates are 1e-4); it stays deterministic and independent of batching.  Measured
            // against the sequential form: ray kernel 0.302 vs 0.312 ms per launch, identical parity figures (profiles/r03_p_scan_ab.log).
            {
                float x = ok ? tr[i] : 1.0f, carry = 1.0f;
                for (int cc = 0; cc < ch; ++cc) { float q = 1.0f; for (int j = cc * 64; j < cc * 64 + 64 && j < dn; ++j) q *= tr[j]; carry = q * carry; }
                NR_PRAGMA_UNROLL
                for (int o = 1; o < SEG; o <<= 1) { const float y = __shfl_up(x, o); x = li >= o ? x * y : x; }
                const float ex = __shfl_up(x, 1);
                T = (li == 0 ? 1.0f : ex) * carry;
            }
#else
            for (int j = 0; j < dn; ++j) { const float tj = tr[j]; T = (j < i) ? T * tj : T; }
#endif
            const float hp = ok ? al[ok ? i : 0] * T : 0.0f;
            const float* rc = rec + (size_t)(ok ? i : 0) * kPointRec;
            if (ok && rvalid) p.hit_prob[(size_t)ray * dn + i] = hp;
            const float4 c4 = ld4(rc + 16);
            cr = fmaf(hp, c4.x, cr); cg = fmaf(hp, c4.y, cg); cb = fmaf(hp, c4.z, cb);
            cd = fmaf(hp, p.depth[(size_t)ray * dn + (ok ? i : 0)], cd);
            const unsigned long long b = __ballot(ok && c4.w > (float)p.mask_view_num);
            cnt += RPW == 2 ? __builtin_popcountll((b >> (32 * seg)) & 0xffffffffull) : __builtin_popcountll(b);
        }
        cr = seg_sum<RPW>(cr); cg = seg_sum<RPW>(cg); cb = seg_sum<RPW>(cb); cd = seg_sum<RPW>(cd);
        if (li == 0 && rvalid) {
            p.pixel[(size_t)ray * 3] = cr; p.pixel[(size_t)ray * 3 + 1] = cg; p.pixel[(size_t)ray * 3 + 2] = cb;
            if (p.render_depth) p.render_depth[ray] = cd;
            if (p.ray_mask) p.ray_mask[ray] = cnt > p.mask_point_num ? 1 : 0;
        }
        NR_WAVE_SYNC();         // (K, V, alpha and the transmittance factors are this wave's own LDS)
    }
}

// -------------------------------------------------------------------------------------------------
// fine kernel: inverse-CDF resampling in normalised inverse depth, then ascending sort.
//   render_ops.py:172-229, renderer.py:210-213.  One wave per ray.
// -------------------------------------------------------------------------------------------------
struct FineParams {
    const float* que_const;
    const float* depth;      // [rn][dn]
    const float* hit_prob;   // [rn][dn]
    const float* u;          // [rn][fdn] or null -> stratified (k + 0.5)/fdn
    float* out;              // [rn][nout], nout = fdn (+ dn when use_all)
    int* idx_out;            // optional [rn][fdn]: the searchsorted(right=True) bin of every sample, in the order of u (before the sort)
    float* cdf_out;          // optional [rn][dn + 1]: the cdf the bins were looked up in
    int rn, dn, fdn, use_all, no_sort;
    int linear;              // sample_fine_depth(inv_mode=False): interpolate the metric depths themselves
};

constexpr int kMaxSamples = 128;   // dn, fdn <= 128
constexpr int kMaxSort = 256;

__global__ void __launch_bounds__(256) fine_kernel(FineParams p) {
    __shared__ float s_s[kRayWaves][kMaxSamples];
    __shared__ float s_pdf[kRayWaves][kMaxSamples];
    __shared__ float s_cdf[kRayWaves][kMaxSamples + 1];
    __shared__ float s_edge[kRayWaves][kMaxSamples + 1];
    __shared__ float s_sort[kRayWaves][kMaxSort];
    const int lane = threadIdx.x & 63;
    const int wave = NR_UNIFORM((int)(threadIdx.x >> 6));
    const int dn = p.dn, fdn = p.fdn;
    const int nout = fdn + (p.use_all ? dn : 0);
    int npad = 1;
    while (npad < nout) npad <<= 1;
    const float nearp = p.que_const[24], farp = p.que_const[25];
    float* ss = s_s[wave]; float* pdf = s_pdf[wave]; float* cdf = s_cdf[wave]; float* edge = s_edge[wave]; float* srt = s_sort[wave];
    // (every LDS array below is this wave's own: the ordering points are wave-level, not workgroup barriers - 22 s_barriers per ray less)
    const int nray_iter = (p.rn + kRayWaves - 1) / kRayWaves;
    for (int it = blockIdx.x; it < nray_iter; it += gridDim.x) {
        int ray = it * kRayWaves + wave;
        const bool rvalid = ray < p.rn;
        ray = rvalid ? ray : p.rn - 1;
        const float* drow = p.depth + (size_t)ray * dn;
        const float* hrow = p.hit_prob + (size_t)ray * dn;
        for (int i = lane; i < dn; i += 64) { ss[i] = p.linear ? drow[i] : norm_inv_depth(drow[i], nearp, farp); pdf[i] = hrow[i] + 1e-5f; }
        NR_WAVE_SYNC();
        // sum(hit_prob + 1e-5) (render_ops.py:194) in the order of the oracle's np.sum - numpy's pairwise sum, which for n <= 128 is one
        // block: eight interleaved partial sums r_j = a[j] + a[8 + j] + ..., the tree ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)),
        // then the n % 8 tail added one by one - so that the cdf, and with it every searchsorted bin, is the oracle's bit for bit on
        // identical inputs (tests/test_fine_index.py; a wave butterfly differed in the total's last bit on ~40 % of the rays, which
        // moved a bin only where u sat within an ulp of a cdf entry - shown there as well).  Every lane computes the same value.
        // Scope of the claim: the ORACLE's order (numpy), not torch.sum's, and only while numpy sums one block, i.e. dn <= 128 - which
        // is every dn this entry point accepts (neuray_sample_fine_depth rejects dn > NEURAY_MAX_SAMPLES):
        static_assert(kMaxSamples <= 128, "above 128 entries np.sum splits recursively: implement the split or drop the bit-exact claim");
        float tot = 0.0f;
        if (dn < 8) {
            for (int i = 0; i < dn; ++i) tot += pdf[i];
        } else {
            const int j8 = lane & 7, n8 = dn - (dn & 7);
            float r = pdf[j8];
            for (int i = 8; i < n8; i += 8) r += pdf[i + j8];
            r = r + __shfl_xor(r, 1);
            r = r + __shfl_xor(r, 2);
            r = r + __shfl_xor(r, 4);
            for (int i = n8; i < dn; ++i) r += pdf[i];
            tot = r;
        }
        NR_WAVE_SYNC();
        for (int i = lane; i < dn; i += 64) pdf[i] = pdf[i] / tot;
        NR_WAVE_SYNC();
        for (int i = lane; i <= dn; i += 64) {
            // sequential prefix sum (bit-exact cumsum order)
            float cdfv = 0.0f;
            for (int j = 0; j < dn; ++j) { const float pj = pdf[j]; cdfv = (j < i) ? cdfv + pj : cdfv; }
            cdf[i] = cdfv;
            if (p.cdf_out && rvalid) p.cdf_out[(size_t)ray * (dn + 1) + i] = cdfv;
            edge[i] = (i == 0) ? ss[0] : (i == dn ? ss[dn - 1] : rn_div(rn_add(ss[i], ss[i - 1]), 2.0f));
        }
        NR_WAVE_SYNC();
        const float interval = (float)(1.0 / (double)fdn);
        for (int k = lane; k < npad; k += 64) {
            float val = INFINITY;
            if (k < fdn) {
                const float uu = p.u ? p.u[(size_t)ray * fdn + k] : rn_add(rn_mul(0.5f, interval), rn_mul((float)k, interval));
                int idx = 0;
                for (int m = 0; m <= dn; ++m) idx += (cdf[m] <= uu) ? 1 : 0;     // searchsorted(right=True)
                if (p.idx_out && rvalid) p.idx_out[(size_t)ray * fdn + k] = idx;
                const int below = idx - 1 > 0 ? idx - 1 : 0;
                const int above = idx < dn ? idx : dn;
                float denom = rn_sub(cdf[above], cdf[below]);
                if (denom < 1e-5f) denom = 1.0f;
                const float tt = rn_div(rn_sub(uu, cdf[below]), denom);
                float sf = rn_add(edge[below], rn_mul(tt, rn_sub(edge[above], edge[below])));
                if (p.linear) {
                    val = sf;
                } else {
                    sf = rn_add(rn_mul(sf, rn_sub(farp, nearp)), nearp);
                    val = rn_div(-1.0f, sf);
                }
            } else if (k < nout) {
                val = drow[k - fdn];
            }
            srt[k] = val;
        }
        NR_WAVE_SYNC();
        // bitonic sort (ascending) of npad values in LDS
        for (int kk = 2; kk <= npad && !p.no_sort; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int i = lane; i < npad; i += 64) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const float a = srt[i], b = srt[ixj];
                        const bool up = (i & kk) == 0;
                        if ((a > b) == up) { srt[i] = b; srt[ixj] = a; }
                    }
                }
                NR_WAVE_SYNC();
            }
        if (rvalid)
            for (int k = lane; k < nout; k += 64) p.out[(size_t)ray * nout + k] = srt[k];
        NR_WAVE_SYNC();
    }
}

// -------------------------------------------------------------------------------------------------
// plane-sweep variance volume of the cost-volume init net
//   network/mvsnet/mvsnet.py:186-203 (construct_cost_volume_with_src) with homo_warp, mvsnet/modules.py:25-64, fused:
// for reference view r, depth plane d and feature pixel (x, y):  p = M_rj (x d, y d, d, 1),  M_rj = (src_proj_j ref_proj_r^-1)[:3]
// (computed by the caller as the reference computes it), z clamped to >= 1e-4, the 32-channel feature of source view j
// read at (p.x/z, p.y/z) (bilinear, ZERO padding, align_corners=True), and over the n_num sources plus the reference's
// own feature   var_c = sum_sq_c / V - (sum_c / V)^2,  V = n_num + 1.
// The reference materialises a [B,32,D,h,w] volume per source view plus the two running sums (the memory peak of the
// whole pipeline); here a thread owns one (r, d, y, x), keeps the 2 x 32 sums in registers and writes the variance once.
// feats are NHWC (8 x 16-byte loads per tap); out is NCDHW for the 3-D convolutions that follow.
// -------------------------------------------------------------------------------------------------
struct WarpVarParams {
    const float* ref_feats;    // [rfn][fh][fw][32]
    const float* src_feats;    // [sn][fh][fw][32]
    const int* nn_ids;         // [rfn][n_num] rows of src_feats
    const float* transforms;   // [rfn][n_num][12]: 3x4, row-major
    const float* depth_vals;   // [rfn][dn]
    float* out;                // [rfn][32][dn][fh][fw], or channels_last: [rfn][dn][fh][fw][32]
    int rfn, n_num, dn, fh, fw;
    int channels_last;         // the layout costreg_conv0_kernel reads (csrc/nr_kernels_conv3d.h): a voxel's 32 channels are one 128-byte line
};

__global__ void __launch_bounds__(256) warp_variance_kernel(WarpVarParams p) {
    const long long hw = (long long)p.fh * p.fw, per_view = hw * p.dn, total = per_view * p.rfn;
    const float wm1 = (float)(p.fw - 1), hm1 = (float)(p.fh - 1);
    const float half_w = rn_div(wm1, 2.0f), half_h = rn_div(hm1, 2.0f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / per_view);
        const long long rem = i - (long long)r * per_view;
        const int d = (int)(rem / hw), pix = (int)(rem - (long long)d * hw);
        const int y = pix / p.fw, x = pix - y * p.fw;
        const float dv = p.depth_vals[r * p.dn + d];
        const float gx = rn_mul((float)x, dv), gy = rn_mul((float)y, dv), gz = dv;
        float sum[32], sq[32];
        {
            const float4* own = reinterpret_cast<const float4*>(p.ref_feats + ((long long)r * hw + pix) * 32);
            NR_PRAGMA_UNROLL
            for (int q = 0; q < 8; ++q) {
                const float4 v = own[q];
                sum[4 * q] = v.x; sum[4 * q + 1] = v.y; sum[4 * q + 2] = v.z; sum[4 * q + 3] = v.w;
                sq[4 * q] = v.x * v.x; sq[4 * q + 1] = v.y * v.y; sq[4 * q + 2] = v.z * v.z; sq[4 * q + 3] = v.w * v.w;
            }
        }
        for (int j = 0; j < p.n_num; ++j) {
            const float* M = p.transforms + ((long long)r * p.n_num + j) * 12;
            const float X = rn_add(dot3(M[0], M[1], M[2], gx, gy, gz), M[3]);
            const float Y = rn_add(dot3(M[4], M[5], M[6], gx, gy, gz), M[7]);
            float Z = rn_add(dot3(M[8], M[9], M[10], gx, gy, gz), M[11]);
            if (Z < 1e-4f) Z = 1e-4f;
            // grid_sample(align_corners=True) un-normalisation of  g = s / ((size-1)/2) - 1
            const float ix = rn_mul(rn_div(rn_add(rn_sub(rn_div(rn_div(X, Z), half_w), 1.0f), 1.0f), 2.0f), wm1);
            const float iy = rn_mul(rn_div(rn_add(rn_sub(rn_div(rn_div(Y, Z), half_h), 1.0f), 1.0f), 2.0f), hm1);
            const float x0f = floorf(ix), y0f = floorf(iy);
            const float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = (x0f + 1.0f) - ix, wy0 = (y0f + 1.0f) - iy;
            // (NaN / huge coordinates fail every bounds test below: zero contribution, as in grid_sample)
            const bool okx0 = x0f >= 0.0f && x0f <= wm1, okx1 = x0f + 1.0f >= 0.0f && x0f + 1.0f <= wm1;
            const bool oky0 = y0f >= 0.0f && y0f <= hm1, oky1 = y0f + 1.0f >= 0.0f && y0f + 1.0f <= hm1;
            const int x0 = okx0 ? (int)x0f : 0, x1 = okx1 ? (int)x0f + 1 : 0, y0 = oky0 ? (int)y0f : 0, y1 = oky1 ? (int)y0f + 1 : 0;
            const float w00 = (okx0 && oky0) ? wx0 * wy0 : 0.0f, w10 = (okx1 && oky0) ? wx1 * wy0 : 0.0f;
            const float w01 = (okx0 && oky1) ? wx0 * wy1 : 0.0f, w11 = (okx1 && oky1) ? wx1 * wy1 : 0.0f;
            const float4* m = reinterpret_cast<const float4*>(p.src_feats + (long long)p.nn_ids[r * p.n_num + j] * hw * 32);
            const float4* t00 = m + ((long long)y0 * p.fw + x0) * 8;
            const float4* t10 = m + ((long long)y0 * p.fw + x1) * 8;
            const float4* t01 = m + ((long long)y1 * p.fw + x0) * 8;
            const float4* t11 = m + ((long long)y1 * p.fw + x1) * 8;
            NR_PRAGMA_UNROLL
            for (int q = 0; q < 8; ++q) {
                const float4 a = t00[q], b = t10[q], c = t01[q], e = t11[q];
                const float v0 = a.x * w00 + b.x * w10 + c.x * w01 + e.x * w11;
                const float v1 = a.y * w00 + b.y * w10 + c.y * w01 + e.y * w11;
                const float v2 = a.z * w00 + b.z * w10 + c.z * w01 + e.z * w11;
                const float v3 = a.w * w00 + b.w * w10 + c.w * w01 + e.w * w11;
                sum[4 * q] += v0; sum[4 * q + 1] += v1; sum[4 * q + 2] += v2; sum[4 * q + 3] += v3;
                sq[4 * q] += v0 * v0; sq[4 * q + 1] += v1 * v1; sq[4 * q + 2] += v2 * v2; sq[4 * q + 3] += v3 * v3;
            }
        }
        const float V = (float)(p.n_num + 1);
        if (p.channels_last) {
            float4* o4 = reinterpret_cast<float4*>(p.out + i * 32);
            NR_PRAGMA_UNROLL
            for (int q = 0; q < 8; ++q) {
                float v[4];
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 4; ++k) {
                    const float mean = rn_div(sum[4 * q + k], V);
                    v[k] = rn_sub(rn_div(sq[4 * q + k], V), rn_mul(mean, mean));
                }
                o4[q] = make_float4(v[0], v[1], v[2], v[3]);
            }
            continue;
        }
        float* o = p.out + (long long)r * 32 * per_view + (long long)d * hw + pix;
        NR_PRAGMA_UNROLL
        for (int c = 0; c < 32; ++c) {
            const float mean = rn_div(sum[c], V);
            o[(long long)c * per_view] = rn_sub(rn_div(sq[c], V), rn_mul(mean, mean));
        }
    }
}

// -------------------------------------------------------------------------------------------------
// cross-view consistency features of the depth init net   network/init_net.py:13-61
// Every pixel of every view t is lifted with its depth (depth2pts3d, :13-28), projected into every view s (a4-a6 of the
// render path), the (rgb, depth) map of s is read there (bilinear, border, align_corners=True) and
//   rgb_diff = |rgb_s(uv) - rgb_t(pixel)|,   dpt_diff = min(|-1/max(d_s(uv),1e-5) + 1/max(z,1e-5)| / (far'_s - near'_s), 1.5)
// are reduced over s to a masked mean and variance (ops.py:36-41, the mask sum clamped to 1e-4):
//   out[t][y][x] = [rgb_mean 3, rgb_var 3, dpt_mean, dpt_var]   (NHWC, i.e. the channels-last storage of [rfn,8,h,w]).
// rgbd: [rfn][h][w][4] = rgb + metric depth: one 16-byte load per tap.  The reference materialises
// [rfn, rfn*h*w, {2,1,1,3,3}] tensors for this (41 M projections at 8 x 800 x 800); here it is one pass, one thread per
// pixel, the per-view values of a pixel kept in registers for the two-pass variance.
// lift_const: per view the query constants of neuray_setup_query (K^-1, pose, centre).
// -------------------------------------------------------------------------------------------------
struct DiffFeatsParams {
    const float* view_const;   // [rfn][kViewConst]
    const float* lift_const;   // [rfn][kQueryConst]
    const float* rgbd;         // [rfn][h][w][4]
    float* out;                // [rfn][h][w][8]
    int rfn, h, w;
};

__global__ void __launch_bounds__(256) diff_feats_kernel(DiffFeatsParams p) {
    const long long hw = (long long)p.h * p.w, total = hw * p.rfn;
    const float wf = (float)p.w, hf = (float)p.h;
    const float4* maps = reinterpret_cast<const float4*>(p.rgbd);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i / hw);
        const int pix = (int)(i - t * hw);
        const int y = pix / p.w, x = pix - y * p.w;
        const float* lc = p.lift_const + t * kQueryConst;
        const float4 own = maps[i];
        // depth * [x, y, 1] -> K^-1 -> R^T (.) + t',  t' = -R^T t (the camera centre)
        const float sx = rn_mul(own.w, (float)x), sy = rn_mul(own.w, (float)y), sz = own.w;
        const float c0 = dot3(lc[0], lc[1], lc[2], sx, sy, sz);
        const float c1 = dot3(lc[3], lc[4], lc[5], sx, sy, sz);
        const float c2 = dot3(lc[6], lc[7], lc[8], sx, sy, sz);
        const float* P = lc + 9;
        const float X = rn_add(dot3(P[0], P[4], P[8], c0, c1, c2), lc[21]);
        const float Y = rn_add(dot3(P[1], P[5], P[9], c0, c1, c2), lc[22]);
        const float Z = rn_add(dot3(P[2], P[6], P[10], c0, c1, c2), lc[23]);
        float val[kMaxViews][4], msk[kMaxViews];
        NR_PRAGMA_UNROLL
        for (int s = 0; s < kMaxViews; ++s) {
            val[s][0] = 0.0f; val[s][1] = 0.0f; val[s][2] = 0.0f; val[s][3] = 0.0f; msk[s] = 0.0f;
            if (s < p.rfn) {
                const float* vc = p.view_const + s * kViewConst;
                const Proj pr = project_point<false>(vc, X, Y, Z, wf, hf);
                const Taps tp = taps_from(texel_coord(pr.u, wf, wf, true), texel_coord(pr.v, hf, hf, true), p.w, p.h);
                const float4* m = maps + (long long)s * hw;
                const float4 a = m[tp.o00], b = m[tp.o10], c = m[tp.o01], d = m[tp.o11];
                const float gr = a.x * tp.w00 + b.x * tp.w10 + c.x * tp.w01 + d.x * tp.w11;
                const float gg = a.y * tp.w00 + b.y * tp.w10 + c.y * tp.w01 + d.y * tp.w11;
                const float gb = a.z * tp.w00 + b.z * tp.w10 + c.z * tp.w01 + d.z * tp.w11;
                const float gd = a.w * tp.w00 + b.w * tp.w10 + c.w * tp.w01 + d.w * tp.w11;
                val[s][0] = fabsf(gr - own.x); val[s][1] = fabsf(gg - own.y); val[s][2] = fabsf(gb - own.z);
                const float dd = fabsf(rn_add(rn_div(-1.0f, fmaxf(gd, 1e-5f)), rn_div(1.0f, fmaxf(pr.z, 1e-5f))));
                val[s][3] = fminf(rn_div(dd, rn_sub(vc[16], vc[15])), 1.5f);
                msk[s] = pr.mask;
            }
        }
        float msum = 0.0f, mean[4] = {0.0f, 0.0f, 0.0f, 0.0f}, var[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        NR_PRAGMA_UNROLL
        for (int s = 0; s < kMaxViews; ++s) {
            msum += msk[s];
            NR_PRAGMA_UNROLL
            for (int c = 0; c < 4; ++c) mean[c] += val[s][c] * msk[s];
        }
        msum = fmaxf(msum, 1e-4f);
        NR_PRAGMA_UNROLL
        for (int c = 0; c < 4; ++c) mean[c] = rn_div(mean[c], msum);
        NR_PRAGMA_UNROLL
        for (int s = 0; s < kMaxViews; ++s) {
            NR_PRAGMA_UNROLL
            for (int c = 0; c < 4; ++c) { const float e = val[s][c] - mean[c]; var[c] += e * e * msk[s]; }
        }
        NR_PRAGMA_UNROLL
        for (int c = 0; c < 4; ++c) var[c] = rn_div(var[c], msum);
        float4* o = reinterpret_cast<float4*>(p.out) + 2 * i;
        o[0] = make_float4(mean[0], mean[1], mean[2], var[0]);
        o[1] = make_float4(var[1], var[2], mean[3], var[3]);
    }
}

// -------------------------------------------------------------------------------------------------
// a7 standalone: interpolate_feats / interpolate_feature_map on NCHW maps (network/ops.py:14-34,
// render_ops.py:54-70).  One thread per (batch, point); used for pixel_colors_gt and the auxiliary
// (non per-sample) gathers.  points [b][n][2] pixel (x,y) in units of the (w_full, h_full) image.
// -------------------------------------------------------------------------------------------------
__global__ void interpolate_kernel(const float* __restrict__ feats, const float* __restrict__ points,
                                   const float* __restrict__ mask, int b, int n, int c, int fh, int fw,
                                   int h_full, int w_full, int align, float* __restrict__ out) {
    // one thread per (point, channel): the training batches are a few hundred points, one thread per point left the GPU idle
    const long long total = (long long)b * n * c;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long i = e / c;
        const int ch = (int)(e - i * c);
        const int bi = (int)(i / n);
        const float u = points[2 * i], v = points[2 * i + 1];
        const float ix = texel_coord(u, (float)w_full, (float)fw, align != 0);
        const float iy = texel_coord(v, (float)h_full, (float)fh, align != 0);
        const float x0f = floorf(ix), y0f = floorf(iy);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const int x1 = x0 + 1 < fw ? x0 + 1 : fw - 1, yb = y0 + 1 < fh ? y0 + 1 : fh - 1;
        const float wx1 = rn_sub(ix, x0f), wy1 = rn_sub(iy, y0f);
        const float wx0 = rn_sub(rn_add(x0f, 1.0f), ix), wy0 = rn_sub(rn_add(y0f, 1.0f), iy);
        Taps t;
        t.o00 = y0 * fw + x0; t.o10 = y0 * fw + x1; t.o01 = yb * fw + x0; t.o11 = yb * fw + x1;
        t.w00 = rn_mul(wx0, wy0); t.w10 = rn_mul(wx1, wy0); t.w01 = rn_mul(wx0, wy1); t.w11 = rn_mul(wx1, wy1);
        if (x0 + 1 > fw - 1) { t.w10 = 0.0f; t.w11 = 0.0f; }
        if (y0 + 1 > fh - 1) { t.w01 = 0.0f; t.w11 = 0.0f; }
        const float m = mask ? mask[i] : 1.0f;
        const float* pl = feats + ((size_t)bi * c + ch) * fh * fw;
        out[e] = blend4(pl[t.o00], pl[t.o10], pl[t.o01], pl[t.o11], t) * m;
    }
}

// -------------------------------------------------------------------------------------------------
// Stand-alone ops of the network.render_ops surface (same device functions as the fused point kernel), used by the
// host mirror neuray_amd/network/render_ops.py and by the function-level parity tests.
// -------------------------------------------------------------------------------------------------
// a2: coords2rays / depth2points (render_ops.py:4-39).  centers/dirs [rn][3]; pts, que_dir [rn][dn][3] (may be null)
__global__ void rays_points_kernel(const float* __restrict__ qc, const float* __restrict__ coords, const float* __restrict__ depth,
                                   int rn, int dn, float* __restrict__ centers, float* __restrict__ dirs,
                                   float* __restrict__ pts, float* __restrict__ que_dir) {
    const long long total = (long long)rn * (pts ? dn : 1);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ray = pts ? (int)(i / dn) : (int)i;
        const Ray r = make_ray(qc, coords[2 * ray], coords[2 * ray + 1]);
        if (!pts || i % dn == 0) {
            if (centers) { centers[3 * ray] = r.cx; centers[3 * ray + 1] = r.cy; centers[3 * ray + 2] = r.cz; }
            if (dirs) { dirs[3 * ray] = r.dx; dirs[3 * ray + 1] = r.dy; dirs[3 * ray + 2] = r.dz; }
        }
        if (pts) {
            const float d = depth[i];
            pts[3 * i] = rn_add(r.cx, rn_mul(r.dx, d)); pts[3 * i + 1] = rn_add(r.cy, rn_mul(r.dy, d));
            pts[3 * i + 2] = rn_add(r.cz, rn_mul(r.dz, d));
            que_dir[3 * i] = r.qx; que_dir[3 * i + 1] = r.qy; que_dir[3 * i + 2] = r.qz;
        }
    }
}

// a3: depth2dists (inv = 0) / depth2inv_dists (inv = 1, range = [near, far] of the query) (render_ops.py:41-52)
__global__ void dists_kernel(const float* __restrict__ depth, const float* __restrict__ range, int inv, int rows, int dn,
                             float* __restrict__ out) {
    const long long total = (long long)rows * dn;
    float nearp = 0.0f, farp = 1.0f;
    if (inv) { nearp = rn_div(-1.0f, range[0]); farp = rn_div(-1.0f, range[1]); }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(i % dn);
        if (s == dn - 1) { out[i] = 1e6f; continue; }
        const float a = inv ? norm_inv_depth(depth[i], nearp, farp) : depth[i];
        const float b = inv ? norm_inv_depth(depth[i + 1], nearp, farp) : depth[i + 1];
        out[i] = rn_sub(b, a);
    }
}

// a4-a6: project_points_ref_views (render_ops.py:82-130).  pts [pn][3] -> dir [rfn][pn][3], pts2d [rfn][pn][2],
// depth [rfn][pn], mask [rfn][pn] (0/1 bytes)
__global__ void project_kernel(const float* __restrict__ view_const, const float* __restrict__ pts, int rfn, int pn, int h, int w,
                               float* __restrict__ dir, float* __restrict__ pts2d, float* __restrict__ depth,
                               unsigned char* __restrict__ mask) {
    const long long total = (long long)rfn * pn;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i / pn);
        const long long pi = i - (long long)v * pn;
        const Proj pr = project_point(view_const + v * kViewConst, pts[3 * pi], pts[3 * pi + 1], pts[3 * pi + 2], (float)w, (float)h);
        dir[3 * i] = pr.dirx; dir[3 * i + 1] = pr.diry; dir[3 * i + 2] = pr.dirz;
        pts2d[2 * i] = pr.u; pts2d[2 * i + 1] = pr.v; depth[i] = pr.z; mask[i] = pr.mask > 0.0f ? 1 : 0;
    }
}

// a15: alpha_values2hit_prob (render_ops.py:72-80): hit = alpha * exclusive cumprod(1 - alpha + 1e-10), sequential order
__global__ void hit_prob_kernel(const float* __restrict__ alpha, int rows, int dn, float* __restrict__ out) {
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
        float T = 1.0f;
        for (int i = 0; i < dn; ++i) {
            const float a = alpha[r * dn + i];
            out[r * dn + i] = a * T;
            T = T * ((1.0f - a) + 1e-10f);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// a9 stand-alone: MixtureLogisticsDistDecoder.forward on arbitrary rows (dist_decoder.py:99-107).  feats [n][32]
// row major -> mean [n][2], var [n][2] (bias included), aw [n], vis [n] (only with a vis head).  One wave handles
// 2 tiles of 16 rows; same MFMA layers and packed weights as the point kernel (weights straight from global).
// -------------------------------------------------------------------------------------------------
template <bool HAS_VIS>
__global__ void __launch_bounds__(256) decoder_rows_kernel(const float* __restrict__ feats, const float* __restrict__ weights, int n,
                                                           float var_bias, float* __restrict__ mean, float* __restrict__ var,
                                                           float* __restrict__ aw, float* __restrict__ vis) {
    constexpr int NS = 2;
    const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int wave_global = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (int)((gridDim.x * blockDim.x) >> 6);
    const nr_wbuf W = nr_make_wbuf(weights, sizeof(float) * kPackedPassFloats);
    for (int base = wave_global * 16 * NS; base < n; base += nwaves * 16 * NS) {
        float f[NS][8], none[NS][1];
        int row[NS]; bool ok[NS];
        NR_PRAGMA_UNROLL
        for (int s = 0; s < NS; ++s) {
            int r = base + 16 * s + c;
            ok[s] = r < n; r = ok[s] ? r : n - 1; row[s] = r; none[s][0] = 0.0f;
            const float4 a = ld4(feats + (size_t)r * 32 + 8 * g), b = ld4(feats + (size_t)r * 32 + 8 * g + 4);
            f[s][0] = a.x; f[s][1] = a.y; f[s][2] = a.z; f[s][3] = a.w; f[s][4] = b.x; f[s][5] = b.y; f[s][6] = b.z; f[s][7] = b.w;
        }
        float h1[NS][8], h2[NS][8], fm[NS][2], fv[NS][2], fa[NS][1];
        layer_fwd<L_DM1, NS, ACT_ELU>(W, lane, f, none, h1);
        layer_fwd<L_DM2, NS, ACT_ELU>(W, lane, h1, none, h2);
        layer_vec<L_DFIN_M, NS>(W, lane, h2, fm);
        layer_fwd<L_DV1, NS, ACT_ELU>(W, lane, f, none, h1);
        layer_fwd<L_DV2, NS, ACT_ELU>(W, lane, h1, none, h2);
        layer_vec<L_DFIN_V, NS>(W, lane, h2, fv);
        NR_PRAGMA_UNROLL
        for (int s = 0; s < NS; ++s)
            if (ok[s] && g == 0) {
                mean[2 * row[s]] = softplus(fm[s][0]); mean[2 * row[s] + 1] = softplus(fm[s][1]);
                var[2 * row[s]] = softplus(fv[s][0]) + var_bias; var[2 * row[s] + 1] = softplus(fv[s][1]) + var_bias;
            }
        layer_fwd<L_DA1, NS, ACT_ELU>(W, lane, f, none, h1);
        layer_fwd<L_DA2, NS, ACT_ELU>(W, lane, h1, none, h2);
        layer_vec<L_DFIN_A, NS>(W, lane, h2, fa);
        float fs[NS][1];
        if constexpr (HAS_VIS) {
            layer_fwd<L_DS1, NS, ACT_ELU>(W, lane, f, none, h1);
            layer_fwd<L_DS2, NS, ACT_ELU>(W, lane, h1, none, h2);
            layer_vec<L_DFIN_S, NS>(W, lane, h2, fs);
        }
        NR_PRAGMA_UNROLL
        for (int s = 0; s < NS; ++s)
            if (ok[s] && g == 0) {
                aw[row[s]] = sigmoidf(fa[s][0]);
                if constexpr (HAS_VIS) vis[row[s]] = sigmoidf(fs[s][0]);
            }
    }
}

// a19: compute_prob(is_ref=False) of a ray's own decoded distribution (dist_decoder.py:39-46,109-140; renderer.py:137-155)
//   mean, var [rn][2], aw, vis [rn] (vis may be null), depth [rn][dn] -> hit_prob [rn][dn]
__global__ void self_hit_prob_kernel(const float* __restrict__ qc, const float* __restrict__ depth, const float* __restrict__ mean,
                                     const float* __restrict__ var, const float* __restrict__ aw, const float* __restrict__ vis,
                                     int rn, int dn, float* __restrict__ out) {
    const float nearp = qc[24], farp = qc[25];
    const long long total = (long long)rn * dn;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ray = (int)(i / dn), smp = (int)(i - (long long)ray * dn);
        const float* drow = depth + (size_t)ray * dn;
        // normalised inverse depth of the query samples: the interval fed in is que_dists = depth2inv_dists(depth)
        // (un-clamped), the positions are clamped at 1e-5 (dist_decoder.py:24-28)
        const float t_c = norm_inv_depth(fmaxf(drow[smp], 1e-5f), nearp, farp);
        float lo, hi;
        if (smp == 0) {
            const float half0 = rn_div(rn_sub(norm_inv_depth(drow[1], nearp, farp), norm_inv_depth(drow[0], nearp, farp)), 2.0f);
            lo = rn_sub(t_c, half0);
        } else {
            lo = rn_div(rn_add(norm_inv_depth(fmaxf(drow[smp - 1], 1e-5f), nearp, farp), t_c), 2.0f);
        }
        if (smp == dn - 1) hi = rn_add(t_c, 500000.0f);
        else hi = rn_div(rn_add(t_c, norm_inv_depth(fmaxf(drow[smp + 1], 1e-5f), nearp, farp)), 2.0f);
        float v_, h_;
        const float nu = vis ? vis[ray] : 1.0f;
        // logistic_prob takes (t, lo_half, hi_half) with near = t - lo_half, far = t + hi_half
        logistic_prob(0.0f, -lo, hi, mean[2 * ray], mean[2 * ray + 1], var[2 * ray], var[2 * ray + 1], aw[ray], nu, vis != nullptr, v_, h_);
        out[i] = h_;
    }
}

// -------------------------------------------------------------------------------------------------
// MFMA layout self-test: D = A * B with A, B both asymmetric; host checks against a plain triple loop.
// -------------------------------------------------------------------------------------------------
__global__ void mfma_selftest_kernel(const float* __restrict__ A /*16x4*/, const float* __restrict__ B /*4x16*/,
                                     float* __restrict__ D /*16x16*/) {
    const int lane = threadIdx.x & 63;
    v4f acc; acc[0] = 0.0f; acc[1] = 0.0f; acc[2] = 0.0f; acc[3] = 0.0f;
    acc = nr_mfma16(A[(lane & 15) * 4 + (lane >> 4)], B[(lane >> 4) * 16 + (lane & 15)], acc);
    for (int r = 0; r < 4; ++r) D[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
}

// AR_X3 (nr_layout.h): D[m][n] = sum_k A[m][k] B[k][n], K = 32, through the point kernel's own operand path - nr_split3 on BOTH
// operands, six v_mfma_f32_16x16x32_bf16 products in mfma_unit3's order.  parts (optional, [3][16][32]): the three bf16 parts of A as
// fp32 values, for a host-side evaluation of the split's own error.
__global__ void x3_selftest_kernel(const float* __restrict__ A /*16x32*/, const float* __restrict__ B /*32x16*/, float* __restrict__ D /*16x16*/,
                                   float* __restrict__ parts) {
    const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
    float av[1][8], bv[1][8];
    for (int i = 0; i < 8; ++i) { av[0][i] = A[c * 32 + 8 * g + i]; bv[0][i] = B[(8 * g + i) * 16 + c]; }
    const Opnd3<1, 8> a3 = split_operand(av), b3 = split_operand(bv);
    Frag3 f;
    for (int pt = 0; pt < 3; ++pt)
        for (int i = 0; i < 4; ++i) f.p[pt][i] = a3.p[pt][0][i];
    v4f acc[1];
    acc[0][0] = 0.0f; acc[0][1] = 0.0f; acc[0][2] = 0.0f; acc[0][3] = 0.0f;
    mfma_unit3<L_DM1, 1>(f, 0, b3, acc);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + c] = acc[0][r];
    if (parts)
        for (int pt = 0; pt < 3; ++pt)
            for (int i = 0; i < 8; ++i) {
                const unsigned w = a3.p[pt][0][i / 2], u = (i & 1) ? (w & 0xffff0000u) : (w << 16);
                parts[(pt * 16 + c) * 32 + 8 * g + i] = __int_as_float((int)u);
            }
}

__global__ void group_sum_selftest_kernel(const float* __restrict__ x, float* __restrict__ y) {
    y[threadIdx.x & 63] = nr_group_sum(x[threadIdx.x & 63]);
}

}  // namespace nr
