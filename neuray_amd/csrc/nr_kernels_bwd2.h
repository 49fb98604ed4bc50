// points_backward2_kernel: the register / LDS resident backward of points_kernel (training).
//
// Gradient of: projection -> gathers -> dist decoder -> probabilities -> prob_embed / ray_dir_fc / neuray_fc -> cross-view
// statistics -> base_fc -> vis_fc -> vis_fc2 -> rgb_fc -> softmax blend + visibility-weighted statistics -> geometry_fc
// (autograd of dist_decoder.py:53-140, renderer.py:67-83,127-135, aggregate_net.py:34-68, ibrnet.py:315-354,361-367) with
// respect to every weight of the pass and to the ray_feats / img_feats maps.
//
// Decomposition (mirrors the forward kernel instead of the first version's one-wave arena in global memory):
//   * workgroup = 8 waves x one tile of 16 sample points; wave w owns reference view w (rfn <= 8).  Activations and their
//     gradients of a (point, view) column live in REGISTERS in the forward kernel's D layout, so the recomputed forward
//     (layer_fwd on the packed weights) and the input gradients dX = W^T dY (the same layer code on the transposed pack,
//     nr_layout.h LT_*) chain from layer to layer without leaving the register file.  Nothing goes to global memory but
//     the final gradients.
//   * weight gradients dW = sum over columns dY X^T contract over the columns, which the D layout keeps in lanes: each
//     stage writes its (dY, X) rows to an LDS staging area [feature][128 columns] (the transposition), one barrier, and
//     the [O x 128] x [128 x K] products run as 16 x 16 tiles ("jobs") on the fp32 MFMA, dealt round-robin to the 8
//     waves; every wave keeps the accumulators of its jobs in registers across ALL tiles of the launch (persistent grid)
//     and adds them to global memory once, at the end: one atomicAdd per weight and workgroup.
//   * what needs the OTHER views in the forward direction is not recomputed: the training forward (points_kernel<.., SAVE>)
//     leaves per tile base_fc.0's per-point part, the four weighted statistics, the softmax / visibility sums, geometry_fc's
//     hidden layer and output and every view's dist decoder outputs (nr_kernels.h kSaved*); they arrive here by LDS-DMA /
//     plain loads.  The per-view layers are recomputed.  The backward direction's own cross-view sums (blend, d h64) are the
//     forward kernel's deterministic LDS all-reduce; the transposed per-point layers run on owner waves.
// Formulas: the first version (nr_kernels_bwd.h, kept as the rfn > 8 fallback and as an on-device cross-check).
#pragma once
#include <type_traits>
#include "nr_kernels_bwd.h"

namespace nr {

struct PointBwd2Params {
    const float* que_const;
    const float* view_const;
    const float* coords;       // [rn][2]
    const float* depth;        // [rn][dn]
    const float* ray_feats;    // [rfn][fh][fw][32]
    const float* img_feats;    // [rfn][fh][fw][32]
    const float* rgba;         // [rfn][h][w][4]
    const float* weights;      // packed pass weights (forward layers)
    const float* weights_t;    // packed transposed layers (kPackedTFloats)
    const float* d_point_rec;  // [rn*dn][kPointRec]: [0..15] d geometry feature, [16..18] d colour
    const float* saved;        // [ceil(rn*dn / 16)][kSavedTileFloats]: what the training forward (points_kernel<SAVE>) left (nr_kernels.h)
    float* handover;           // [ceil(rn*dn / 16)][8 waves][kB2Handover][64 lanes]: tail kernel -> front kernel (the two-kernel form), else null
    float* d_flat;             // [kFlatPassFloats], accumulated
    float* d_ray_feats;        // [rfn][fh][fw][32], accumulated
    float* d_img_feats;        // [rfn][fh][fw][32], accumulated
    int rfn, rn, dn, h, w, fh, fw;
    int use_vis;
    float var_bias;
};

constexpr int kB2Waves = 8;
constexpr int kB2Stride = 132;           // floats per staged row: 128 columns (8 views x 16 points) + 4 (bank spread, keeps 16-byte rows)
constexpr int kB2StageRows = 176;
constexpr int kB2PStride = 20;           // per-point staging: 16 columns + 4
constexpr int kB2Rmax = 12;              // rows of the widest all-reduce
// The kernel runs as TWO launches (round 4; its one-launch form - 299 spilled VGPRs, 0.92 ms - was removed in round 6): B2_TAIL = geometry_fc -> blend -> rgb_fc -> vis_fc2 -> vis_fc -> base_fc -> the
// cross-view statistics, leaving per (point, view) column the gradients of what the front of the network produced (d gi: 8 per lane,
// d e: 8, d gr: 3, d sn: 1 = kB2Handover floats per lane); B2_FRONT = neuray_fc / ray_dir_fc / prob_embed -> probabilities -> dist decoder
// heads -> map gradients from those.  Each half holds only its own weight-gradient accumulators (109 / 57 of the 166 jobs) and its own
// part of the chain state: 80 + 0 spilled VGPRs instead of 290, at the price of the geometry, the gathers and part of the recomputed
// forward twice and 84 MB of hand-over per pass: 0.63 ms instead of 0.92.
enum B2Part { B2_TAIL = 0, B2_FRONT = 1 };
constexpr int kB2Handover = 20;
inline size_t point_bwd2_handover_floats(int npts) { return (size_t)((npts + 15) / 16) * kB2Waves * kB2Handover * 64; }
// LDS (floats): all-reduce scratch | xch (base_fc.0 per-point part: 64 features x 16 points, kept from the forward to its
// backward) | stash (the four cross-view statistics, lane layout: 44 values per lane) | hx (hand-off of per-point
// gradients from their owner waves to every wave: <= 44 values per lane) | staging (also: per-point staging, geometry
// exchange, scatter slabs)
constexpr int kB2Red = (kB2Waves + 1) * kB2Rmax * 64 + (kB2Waves + 1) * 64;      // all-reduce scratch + the scratch of the one max-reduce      // three rotating sum banks + the scratch of the one max-reduce
constexpr int kB2Xch = 16 * 64;
constexpr int kB2Stash = 44 * 64;
constexpr int kB2Hx = 44 * 64;
constexpr int kB2Stage = kB2StageRows * kB2Stride;
inline size_t point_bwd2_smem_bytes() { return sizeof(float) * (size_t)(kB2Red + kB2Xch + kB2Stash + kB2Hx + kB2Stage); }

// ---- profile build (-DNR_B2_PROFILE, tools/profile_bwd2.py): cycles between consecutive marks, summed over the tiles of
// workgroup 0, per wave.  Not compiled into the product library.
#ifdef NR_B2_PROFILE
constexpr int kB2Marks = 64;
__device__ unsigned long long nr_b2_prof[kB2Marks * 8];
#define B2_MARK(i) do { if (blockIdx.x == 0 && lane == 0) { const unsigned long long t_ = clock64(); nr_b2_prof[(i) * 8 + wave] += t_ - tprev_; tprev_ = t_; } } while (0)
#define B2_MARK_ARGS , unsigned long long& tprev_
#define B2_MARK_PASS , tprev_
#else
#define B2_MARK(i) do {} while (0)
#define B2_MARK_ARGS
#define B2_MARK_PASS
#endif

// ---- weight-gradient jobs -----------------------------------------------------------------------------------------
// One entry per weight tensor (or column range of one): dW[O x K] = sum_columns dY X^T, as OT x KT tiles of 16 x 16.
struct DwSpec { int tw, tb, ldw, col0, O, K; bool xscaled; bool per_point; };
enum DwId {
    DW_RF4, DW_RF2, DW_RF0, DW_V22, DW_V20, DW_VF2, DW_VF0, DW_B2, DW_BV, DW_NF2, DW_NF0, DW_RD2, DW_RD0, DW_PE2, DW_PE0,
    DW_M4, DW_M2, DW_M0, DW_V4, DW_V2, DW_V0, DW_A4, DW_A2, DW_A0,
    DW_GF2, DW_GF0, DW_BG,
    DW_S4, DW_S2, DW_S0,                      // vis head (decoder with a vis head only): last, so that the numbering is a prefix
    DW_COUNT
};
constexpr DwSpec kDw[DW_COUNT] = {
    {T_RF4_W, T_RF4_B, 8, 0, 1, 8, true, false},   {T_RF2_W, T_RF2_B, 16, 0, 8, 16, true, false},  {T_RF0_W, T_RF0_B, 37, 0, 16, 37, false, false},
    {T_V22_W, T_V22_B, 32, 0, 1, 32, true, false}, {T_V20_W, T_V20_B, 32, 0, 32, 32, false, false},
    {T_VF2_W, T_VF2_B, 32, 0, 33, 32, true, false}, {T_VF0_W, T_VF0_B, 32, 0, 32, 32, false, false},
    {T_BASE2_W, T_BASE2_B, 64, 0, 32, 64, true, false}, {T_BASE0_W, T_BASE0_B, 207, 140, 64, 67, false, false},
    {T_NF2_W, T_NF2_B, 8, 0, 1, 8, true, false},   {T_NF0_W, T_NF0_B, 32, 0, 8, 32, false, false},
    {T_RD2_W, T_RD2_B, 16, 0, 35, 16, true, false}, {T_RD0_W, T_RD0_B, 4, 0, 16, 4, false, false},
    {T_PE2_W, T_PE2_B, 32, 0, 32, 32, false, false}, {T_PE0_W, T_PE0_B, 34, 0, 32, 34, false, false},
    {T_MEAN4_W, T_MEAN4_B, 32, 0, 2, 32, true, false}, {T_MEAN2_W, T_MEAN2_B, 32, 0, 32, 32, true, false}, {T_MEAN0_W, T_MEAN0_B, 32, 0, 32, 32, false, false},
    {T_VAR4_W, T_VAR4_B, 32, 0, 2, 32, true, false},  {T_VAR2_W, T_VAR2_B, 32, 0, 32, 32, true, false},  {T_VAR0_W, T_VAR0_B, 32, 0, 32, 32, false, false},
    {T_AW4_W, T_AW4_B, 32, 0, 1, 32, true, false},   {T_AW2_W, T_AW2_B, 32, 0, 32, 32, true, false},   {T_AW0_W, T_AW0_B, 32, 0, 32, 32, false, false},
    {T_GF2_W, T_GF2_B, 64, 0, 16, 64, true, true},   {T_GF0_W, T_GF0_B, 65, 0, 64, 65, false, true},   {T_BASE0_W, -1, 207, 0, 64, 140, false, true},
    {T_VIS4_W, T_VIS4_B, 32, 0, 1, 32, true, false}, {T_VIS2_W, T_VIS2_B, 32, 0, 32, 32, true, false}, {T_VIS0_W, T_VIS0_B, 32, 0, 32, 32, false, false},
};
constexpr int dw_ot(int id) { return (kDw[id].O + 15) / 16; }
constexpr int dw_kt(int id) { return (kDw[id].K + 15) / 16; }
constexpr int dw_job0(int id) {          // running job index: jobs are dealt to wave (job % 8), accumulator slot (job / 8)
    int j = 0;
    for (int i = 0; i < id; ++i) j += dw_ot(i) * dw_kt(i);
    return j;
}
constexpr int dw_bias0(int id) {         // running index of the bias accumulators (one per 16-row block of dY)
    int j = 0;
    for (int i = 0; i < id; ++i) j += (kDw[i].tb >= 0 ? dw_ot(i) : 0);
    return j;
}
constexpr int kDwJobs = dw_job0(DW_COUNT), kDwAcc = (kDwJobs + kB2Waves - 1) / kB2Waves;
constexpr int kDwBias = dw_bias0(DW_COUNT), kDwBiasAcc = (kDwBias + kB2Waves - 1) / kB2Waves;
// accumulators per wave when NWV waves share the jobs
constexpr int dw_acc_n(int nwv) { return (kDwJobs + nwv - 1) / nwv; }
constexpr int dw_bias_n(int nwv) { return (kDwBias + nwv - 1) / nwv; }

// accumulate the jobs of weight tensor ID that this wave owns.  S: staging area, rowA / rowB: first staged row of dY / X,
// NCQ: column groups of 16 (8 = all views, 1 = per-point operands with stride kB2PStride)
template <int ID, int NWV = kB2Waves>
__device__ __forceinline__ void dw_jobs(v4f (&acc)[dw_acc_n(NWV)], float (&bacc)[dw_bias_n(NWV)], const float* S, int rowA, int rowB, int wave, int lane) {
    constexpr int OT = dw_ot(ID), KT = dw_kt(ID), J0 = dw_job0(ID), B0 = dw_bias0(ID);
    constexpr bool PP = kDw[ID].per_point;
    constexpr int STR = PP ? kB2PStride : kB2Stride, NCQ = PP ? 1 : 8;
    const int m = lane & 15, kk = lane >> 4;
    NR_PRAGMA_UNROLL
    for (int a = 0; a < OT; ++a) {
        NR_PRAGMA_UNROLL
        for (int b = 0; b < KT; ++b) {
            const int job = J0 + a * KT + b;
            if (job % NWV == wave) {
                const float* pa = S + (rowA + 16 * a + m) * STR + 4 * kk;
                const float* pb = S + (rowB + 16 * b + m) * STR + 4 * kk;
                // (software pipeline: the operands of column group j + 1 are read while group j is on the matrix pipe)
                v4f d = acc[job / NWV];
                float4 av = ld4(pa), bv = ld4(pb);
                NR_PRAGMA_UNROLL
                for (int j = 0; j < NCQ; ++j) {
                    float4 an = av, bn = bv;
                    if (j + 1 < NCQ) { an = ld4(pa + 16 * (j + 1)); bn = ld4(pb + 16 * (j + 1)); }
                    NR_PIN();
                    d = nr_mfma16(av.x, bv.x, d); d = nr_mfma16(av.y, bv.y, d);
                    d = nr_mfma16(av.z, bv.z, d); d = nr_mfma16(av.w, bv.w, d);
                    av = an; bv = bn;
                }
                acc[job / NWV] = d;
            }
        }
        if (kDw[ID].tb >= 0) {          // bias: row sums of dY, by the owner of this 16-row block's first job
            const int bj = B0 + a;
            if (bj % NWV == wave) {
                const float* pa = S + (rowA + 16 * a + m) * STR + 4 * kk;
                float s = bacc[bj / NWV];
                NR_PRAGMA_UNROLL
                for (int j = 0; j < NCQ; ++j) { const float4 av = ld4(pa + 16 * j); s += (av.x + av.y) + (av.z + av.w); }
                bacc[bj / NWV] = s;
            }
        }
    }
}

// end of the launch: this wave's accumulators of tensor ID -> global gradient buffer (natural layout)
template <int ID, int NWV = kB2Waves>
__device__ __forceinline__ void dw_flush(const v4f (&acc)[dw_acc_n(NWV)], const float (&bacc)[dw_bias_n(NWV)], float* d_flat, int wave, int lane) {
    constexpr int OT = dw_ot(ID), KT = dw_kt(ID), J0 = dw_job0(ID), B0 = dw_bias0(ID);
    constexpr DwSpec sp = kDw[ID];
    const float xs = sp.xscaled ? (float)(1.0 / kLog2e) : 1.0f;       // X was staged as a scaled-ELU activation L * h
    const int m = lane & 15, kk = lane >> 4;
    NR_PRAGMA_UNROLL
    for (int a = 0; a < OT; ++a) {
        NR_PRAGMA_UNROLL
        for (int b = 0; b < KT; ++b) {
            const int job = J0 + a * KT + b;
            if (job % NWV == wave) {
                const v4f d = acc[job / NWV];
                NR_PRAGMA_UNROLL
                for (int r = 0; r < 4; ++r) {
                    const int o = 16 * a + 4 * kk + r, k = 16 * b + m;
                    if (o < sp.O && k < sp.K) atomicAdd(d_flat + tensor_offset(sp.tw) + o * sp.ldw + sp.col0 + k, d[r] * xs);
                }
            }
        }
        if (sp.tb >= 0) {
            const int bj = B0 + a;
            if (bj % NWV == wave) {
                const float s = nr_group_sum(bacc[bj / NWV]);          // the four column subsets kk
                const int o = 16 * a + m;
                if (kk == 0 && o < sp.O) atomicAdd(d_flat + tensor_offset(sp.tb) + o, s);
            }
        }
    }
}

// ---- staging writes (the transposition): this wave's 16 columns are [col0, col0 + 16) ---------------------------------
// D-layout registers x[4t + r] = feature 16t + 4g + r of point c -> rows row0 + feature
template <int NREG>
__device__ __forceinline__ void st_nat(float* S, int stride, int row0, const float (&x)[NREG], int col, int g) {
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NREG / 4; ++t)
        NR_PRAGMA_UNROLL
        for (int r = 0; r < 4; ++r) S[(row0 + 16 * t + 4 * g + r) * stride + col] = x[4 * t + r];
}
// gathered-order registers x[k] = channel 8g + k
__device__ __forceinline__ void st_gat(float* S, int stride, int row0, const float (&x)[8], int col, int g) {
    NR_PRAGMA_UNROLL
    for (int k = 0; k < 8; ++k) S[(row0 + 8 * g + k) * stride + col] = x[k];
}
// a value replicated over the four lane groups -> one row
__device__ __forceinline__ void st_one(float* S, int stride, int row, float v, int col, int g) {
    if (g == 0) S[row * stride + col] = v;
}

// derivative of the ELU from its output: scaled form (h' = L * ELU(y)) and plain form
__device__ __forceinline__ float delu_s(float hs) { return hs > 0.0f ? 1.0f : fmaf(hs, (float)(1.0 / kLog2e), 1.0f); }
__device__ __forceinline__ float delu(float y) { return y > 0.0f ? 1.0f : y + 1.0f; }

// dX[4t + r] = F * sum_j w_j[t].r * dy[j]: input gradient of a forward vector-row layer L from its own weights
// (F = L where the packed vector weights carry 1 / L, nr_pack.cpp pack_vec)
template <int L, int NREG>
__device__ __forceinline__ void vec_bwd(const VecPre<L>& p, const float (&dy)[kVec[L].n], float (&dx)[NREG]) {
    constexpr int N = kVec[L].n, TI = kVec[L].tiles;
    static_assert(NREG >= 4 * TI, "vec_bwd: output too small");
    const float f = kInScaled[L] ? (float)kLog2e : 1.0f;
    NR_PRAGMA_UNROLL
    for (int t = 0; t < TI; ++t) {
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        NR_PRAGMA_UNROLL
        for (int j = 0; j < N; ++j) {
            const float4 w = p.w[j * TI + t];
            a.x = fmaf(w.x, dy[j], a.x); a.y = fmaf(w.y, dy[j], a.y); a.z = fmaf(w.z, dy[j], a.z); a.w = fmaf(w.w, dy[j], a.w);
        }
        dx[4 * t] = a.x * f; dx[4 * t + 1] = a.y * f; dx[4 * t + 2] = a.z * f; dx[4 * t + 3] = a.w * f;
    }
}

// bwd_prob (nr_kernels_bwd.h) on the hardware transcendentals (tanh_ of nr_device.h: ~1e-7 absolute, as the forward kernel)
__device__ __forceinline__ void b2_prob_bwd(float nearv, float farv, float mu0, float mu1, float sd0, float sd1, float aw, float nuu,
                                            bool use_vis, float dv, float dh, float& dmu0, float& dmu1, float& dsd0, float& dsd1,
                                            float& daw, float& dnu) {
    const float t00 = tanh_((nearv - mu0) * sd0), t01 = tanh_((nearv - mu1) * sd1);
    const float t10 = tanh_((farv - mu0) * sd0), t11 = tanh_((farv - mu1) * sd1);
    const float g00 = 0.5f + 0.5f * t00, g01 = 0.5f + 0.5f * t01, g10 = 0.5f + 0.5f * t10, g11 = 0.5f + 0.5f * t11;
    const float c00 = g00 * nuu, c01 = g01 * nuu, c10 = g10 * nuu, c11 = g11 * nuu;
    const float mix0 = aw, mix1 = 1.0f - aw;
    const float dmix0 = dv * (1.0f - c00) + dh * (c10 - c00), dmix1 = dv * (1.0f - c01) + dh * (c11 - c01);
    const float dc00 = -mix0 * (dv + dh), dc01 = -mix1 * (dv + dh), dc10 = mix0 * dh, dc11 = mix1 * dh;
    if (use_vis) dnu += dc00 * g00 + dc01 * g01 + dc10 * g10 + dc11 * g11;
    const float da00 = dc00 * nuu * 0.5f * (1.0f - t00 * t00), da01 = dc01 * nuu * 0.5f * (1.0f - t01 * t01);
    const float da10 = dc10 * nuu * 0.5f * (1.0f - t10 * t10), da11 = dc11 * nuu * 0.5f * (1.0f - t11 * t11);
    dmu0 += -sd0 * (da00 + da10); dmu1 += -sd1 * (da01 + da11);
    dsd0 += (nearv - mu0) * da00 + (farv - mu0) * da10;
    dsd1 += (nearv - mu1) * da01 + (farv - mu1) * da11;
    daw += dmix0 - dmix1;
}

// One all-reduce (sum) over the 8 view waves of R per-lane values: the forward kernel's deterministic reduce-scatter +
// all-gather through LDS (two barriers).  (Tried: every wave adds into one bank with LDS float atomics and a single
// barrier - ds_add_f32 from eight waves onto the same 64 words made the kernel 0.4 ms slower, 1.09 -> 1.49 ms.)
struct B2Red { float* base; };
template <int R>
__device__ __forceinline__ void b2_allsum(float (&v)[R], B2Red& rd, int wave, int lane) {
    block_allreduce<R, kB2Rmax, RED_SUM>(v, rd.base, wave, kB2Waves, lane);
}

// the three (four) 32 -> 32 -> 32 -> out heads of the dist decoder: forward on f_ray, outputs only
template <bool HAS_VIS>
__device__ __forceinline__ void b2_dist_fwd(nr_wbuf W, int lane, const float (&fray)[1][8], float var_bias,
                                            float& mu0, float& mu1, float& s0, float& s1, float& aw, float& nu) {
    float none[1][1] = {{0.0f}}, h1[1][8], h2[1][8], o2[1][2], o1[1][1];
    layer_fwd<L_DM1, 1, ACT_ELU>(W, lane, fray, none, h1); layer_fwd<L_DM2, 1, ACT_ELU>(W, lane, h1, none, h2);
    layer_vec<L_DFIN_M, 1>(W, lane, h2, o2);
    mu0 = softplus(o2[0][0]); mu1 = softplus(o2[0][1]);
    layer_fwd<L_DV1, 1, ACT_ELU>(W, lane, fray, none, h1); layer_fwd<L_DV2, 1, ACT_ELU>(W, lane, h1, none, h2);
    layer_vec<L_DFIN_V, 1>(W, lane, h2, o2);
    s0 = softplus(o2[0][0]) + var_bias; s1 = softplus(o2[0][1]) + var_bias;
    layer_fwd<L_DA1, 1, ACT_ELU>(W, lane, fray, none, h1); layer_fwd<L_DA2, 1, ACT_ELU>(W, lane, h1, none, h2);
    layer_vec<L_DFIN_A, 1>(W, lane, h2, o1);
    aw = sigmoidf(o1[0][0]);
    nu = 1.0f;
    if constexpr (HAS_VIS) {
        layer_fwd<L_DS1, 1, ACT_ELU>(W, lane, fray, none, h1); layer_fwd<L_DS2, 1, ACT_ELU>(W, lane, h1, none, h2);
        layer_vec<L_DFIN_S, 1>(W, lane, h2, o1);
        nu = sigmoidf(o1[0][0]);
    }
}

// backward of one dist head: recompute h1, h2, stage (d out, h2, d h2, h1, d h1, f_ray), weight-gradient jobs, dFR +=
//   L1, L2, LF: forward layers; T1, T2: transposed layers; D4, D2, D0: weight-gradient ids; NOUT outputs with gradients dout
template <int L1, int L2, int LF, int T1, int T2, int D4, int D2, int D0, int NOUT>
__device__ __forceinline__ void b2_dist_head_bwd(nr_wbuf W, nr_wbuf WT, int wlane, int lane, int wave, int col, int g, const float (&fray)[1][8],
                                                 const float (&dout)[NOUT], float (&dfr)[8], float* S,
                                                 v4f (&acc)[kDwAcc], float (&bacc)[kDwBiasAcc] B2_MARK_ARGS) {
    float none[1][1] = {{0.0f}}, h1[1][8], h2[1][8], dh2[1][8], dh1[1][8], dx[1][8];
    B2_MARK(34);
    layer_fwd<L1, 1, ACT_ELU>(W, wlane, fray, none, h1);
    layer_fwd<L2, 1, ACT_ELU>(W, wlane, h1, none, h2);
    B2_MARK(35);
    VecPre<LF> pf;
    layer_prefetch<LF>(W, wlane, pf);
    float dy[kVec[LF].n];
    NR_PRAGMA_UNROLL
    for (int j = 0; j < NOUT; ++j) dy[j] = dout[j];
    vec_bwd<LF>(pf, dy, dh2[0]);
    NR_PRAGMA_UNROLL
    for (int k = 0; k < 8; ++k) dh2[0][k] *= delu_s(h2[0][k]);
    layer_fwd<T2, 1, ACT_NONE>(WT, wlane, dh2, none, dh1);
    NR_PRAGMA_UNROLL
    for (int k = 0; k < 8; ++k) dh1[0][k] *= delu_s(h1[0][k]);
    layer_fwd<T1, 1, ACT_NONE>(WT, wlane, dh1, none, dx);
    NR_PRAGMA_UNROLL
    for (int k = 0; k < 8; ++k) dfr[k] += dx[0][k];
    // rows: 0 d out (16), 16 h2 (32), 48 d h2 (32), 80 h1 (32), 112 d h1 (32), 144 f_ray (32)
    B2_MARK(36);
    __syncthreads();
    B2_MARK(37);
    NR_PRAGMA_UNROLL
    for (int j = 0; j < NOUT; ++j) st_one(S, kB2Stride, j, dout[j], col, g);
    st_nat<8>(S, kB2Stride, 16, h2[0], col, g); st_nat<8>(S, kB2Stride, 48, dh2[0], col, g);
    st_nat<8>(S, kB2Stride, 80, h1[0], col, g); st_nat<8>(S, kB2Stride, 112, dh1[0], col, g);
    st_gat(S, kB2Stride, 144, fray[0], col, g);
    B2_MARK(38);
    __syncthreads();
    B2_MARK(39);
    dw_jobs<D4>(acc, bacc, S, 0, 16, wave, lane);
    dw_jobs<D2>(acc, bacc, S, 48, 80, wave, lane);
    dw_jobs<D0>(acc, bacc, S, 112, 144, wave, lane);
    B2_MARK(40);
}

template <int... IDS>
__device__ __forceinline__ void dw_flush_all(const v4f (&acc)[kDwAcc], const float (&bacc)[kDwBiasAcc], float* d_flat, int wave, int lane) {
    (dw_flush<IDS>(acc, bacc, d_flat, wave, lane), ...);
}

template <bool HAS_VIS, int PART>
__global__ void __launch_bounds__(512, 2) points_backward2_kernel(PointBwd2Params p) {
    constexpr bool DO_TAIL = PART == B2_TAIL, DO_FRONT = PART == B2_FRONT;
    NR_DYNAMIC_SMEM(float, smem);
    const int lane = threadIdx.x & 63;
    const int wave = NR_UNIFORM((int)(threadIdx.x >> 6));
    const int g = lane >> 4, c = lane & 15;
    float* redm = smem + (kB2Waves + 1) * kB2Rmax * 64;     // scratch of the max-reduce (block_allreduce with RMAX = 1)
    B2Red red{smem};
    float* xch = smem + kB2Red;
    float* stash = xch + kB2Xch;
    float* hx = stash + kB2Stash;
    float* S = hx + kB2Hx;
    float* xg = S + (kB2StageRows - 64) * kB2Stride;        // geometry hidden exchange: the tail of the staging area
    const nr_wbuf W = nr_make_wbuf(p.weights, sizeof(float) * kPackedPassFloats);
    const nr_wbuf WT = nr_make_wbuf(p.weights_t, sizeof(float) * kPackedTFloats);
    const float* __restrict__ qc = p.que_const;
    const float qnearp = qc[24], qfarp = qc[25], qinv = qc[27];
    const float w_m1 = (float)(p.w - 1), h_m1 = (float)(p.h - 1);
    const float inv_w_m1 = 1.0f / w_m1, inv_h_m1 = 1.0f / h_m1, inv_rfn = 1.0f / (float)p.rfn;
    const size_t fmap = (size_t)p.fh * p.fw * 32, imap = (size_t)p.h * p.w * 4;
    const nr_mbuf rf_map = nr_make_mbuf(p.ray_feats, sizeof(float) * fmap * p.rfn);
    const nr_mbuf if_map = nr_make_mbuf(p.img_feats, sizeof(float) * fmap * p.rfn);
    const nr_mbuf rgb_map = nr_make_mbuf(p.rgba, sizeof(float) * imap * p.rfn);
    const int goff = 32 * g;
    const int npts = p.rn * p.dn, dn = p.dn;
    const bool use_vis = HAS_VIS && p.use_vis != 0;
    const bool vok = wave < p.rfn;                         // padding waves (rfn < 8): masked out everywhere
    const int view = vok ? wave : p.rfn - 1;
    const int col = wave * 16 + c;                         // this lane's column of the staging area
    const float* __restrict__ vc = p.view_const + view * kViewConst;
    const int soff_f = view * (int)(fmap * sizeof(float)), soff_c = view * (int)(imap * sizeof(float));

    v4f acc[kDwAcc];
    float bacc[kDwBiasAcc];
    NR_PRAGMA_UNROLL
    for (int i = 0; i < kDwAcc; ++i) { acc[i][0] = 0.0f; acc[i][1] = 0.0f; acc[i][2] = 0.0f; acc[i][3] = 0.0f; }
    NR_PRAGMA_UNROLL
    for (int i = 0; i < kDwBiasAcc; ++i) bacc[i] = 0.0f;
    float none[1][1] = {{0.0f}};

    for (int base = (int)blockIdx.x * 16; base < npts; base += (int)gridDim.x * 16) {
        const int glane = lane + nr_opaque_zero();
        const int gg = glane >> 4;
#ifdef NR_B2_PROFILE
        unsigned long long tprev_ = clock64();
#endif
        // base_fc.0's per-point part and the four statistics of this tile, as the forward kernel left them: global -> xch | stash by
        // LDS-DMA (15 KB, two 1 KB pieces per wave); visible after the barriers of the first all-reduce below
        const float* svt = p.saved + (size_t)(base / 16) * kSavedTileFloats;
        float* ho = p.handover ? p.handover + ((size_t)(base / 16) * kB2Waves + wave) * (kB2Handover * 64) + lane : nullptr;
        if constexpr (DO_TAIL) {
            const nr_wbuf SV = nr_make_wbuf(svt, sizeof(float) * kSavedTileFloats);
            for (int i = wave; i < (kB2Xch + kB2Stash) / 256; i += kB2Waves) nr_dma16(SV, xch + i * 256, lane, lane * 16, i * 1024);
            // rows kSavedGeoRow .. kSavedSvisRow (39 rows; 40 copied) -> the geometry exchange area xg: hidden layer of geometry_fc (16
            // rows), weighted mean (8) / variance (8), geometry_fc's output (4), max z, sum exp, sum vis''
            for (int i = wave; i < 10; i += kB2Waves) nr_dma16(SV, xg + i * 256, lane, lane * 16, (kSavedGeoRow * 64 + i * 256) * 4);
        }
        // ================= geometry + gathers (as points_kernel) =================
        int pi = base + c;
        const bool pvalid = pi < npts;
        pi = pvalid ? pi : npts - 1;
        const int ray = pi / dn, smp = pi - ray * dn;
        const Ray r = make_ray<false>(qc, p.coords[2 * ray], p.coords[2 * ray + 1]);
        const float* drow = p.depth + (size_t)ray * dn;
        const float d = drow[smp];
        const float s_c = norm_inv_depth_fast(d, qnearp, qfarp, qinv);
        const float s_n = norm_inv_depth_fast(drow[smp + 1 < dn ? smp + 1 : smp], qnearp, qfarp, qinv);
        const float s_p = norm_inv_depth_fast(drow[smp > 0 ? smp - 1 : 0], qnearp, qfarp, qinv);
        const float half_c = (smp == dn - 1) ? 500000.0f : (s_n - s_c) * 0.5f;
        const float hi = half_c, lo = (smp == 0) ? half_c : (s_c - s_p) * 0.5f;
        const float px = rn_add(r.cx, rn_mul(r.dx, d)), py = rn_add(r.cy, rn_mul(r.dy, d)), pz = rn_add(r.cz, rn_mul(r.dz, d));
        Proj pr = project_point<false>(vc, px, py, pz, (float)p.w, (float)p.h);
        const float mask = vok ? pr.mask : 0.0f;
        float dlt[4];
        dlt[0] = pr.dirx - r.qx; dlt[1] = pr.diry - r.qy; dlt[2] = pr.dirz - r.qz;
        dlt[3] = dot3(pr.dirx, pr.diry, pr.dirz, r.qx, r.qy, r.qz);
        const float tref = norm_inv_depth_fast(fmaxf(pr.z, 1e-5f), vc[15], vc[16], vc[17]);
        const Taps tfs = make_taps_fast(pr.u, pr.v, w_m1, h_m1, inv_w_m1, inv_h_m1, p.fw, p.fh, p.fw == p.w && p.fh == p.h);
        float fray[1][8], fimg[8], rgb[3];
        {
            const Taps tcs = make_taps_fast(pr.u, pr.v, w_m1, h_m1, inv_w_m1, inv_h_m1, p.w, p.h, true);
            float4 qf[8], qi[8], qcl[4];
            issue8(rf_map, goff, soff_f, tfs, qf);
            issue8(if_map, goff, soff_f, tfs, qi);
            issue_rgb(rgb_map, soff_c, tcs, qcl);
            NR_PIN();
            blend8(qf, tfs, mask, fray[0]);
            blend8(qi, tfs, mask, fimg);
            blend_rgb(qcl, tcs, mask, rgb);
        }
        B2_MARK(0);
        // ================= forward (recomputed; checkpoints stay in registers) =================
        // the dist decoder's outputs come from the forward (its layers are recomputed head by head in the backward part only)
        const float* sd = svt + kSavedDist + view * 128 + c;
        const float mu0 = sd[0], mu1 = sd[16], s0 = sd[32], s1 = sd[48], aw = sd[64], nu = sd[80];
        B2_MARK(1);
        const float nuu = use_vis ? nu : 1.0f;
        float vis, hit;
        {
            float v_, h_;
            logistic_prob(tref, lo, hi, mu0, mu1, s0, s1, aw, nu, use_vis, v_, h_);
            vis = v_ * mask; hit = h_ * mask;
        }
        // prob_embed -> e
        float e[1][8];
        {
            float x1[1][1], h[1][8];
            x1[0][0] = sel4(g, (hit - 0.5f) * 2.0f, (vis - 0.5f) * 2.0f, 0.0f, 0.0f);
            layer_fwd<L_PE1, 1, ACT_RELU>(W, glane, fray, x1, h);
            layer_fwd<L_PE2, 1, ACT_NONE>(W, glane, h, none, e);
        }
        // ray_dir_fc -> gi (img part, gathered order), gr (rgb part)
        float gi[8], gr[3];
        {
            float x1[1][1], h[1][4], df[1][8], dc[1][3];
            x1[0][0] = sel4(g, dlt[0], dlt[1], dlt[2], dlt[3]);
            layer_fwd<L_RD1, 1, ACT_ELU>(W, glane, none, x1, h);
            layer_fwd<L_RD2, 1, ACT_ELU>(W, glane, h, none, df);
            layer_vec<L_RD2, 1>(W, glane, h, dc);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) gi[k] = fimg[k] + df[0][k];
            NR_PRAGMA_UNROLL
            for (int j = 0; j < 3; ++j) gr[j] = rgb[j] + elu(dc[0][j]);
        }
        // neuray_fc -> sn
        float sn;
        {
            float h[1][4], o[1][1];
            layer_fwd<L_NF1, 1, ACT_ELU>(W, glane, e, none, h);
            layer_vec<L_NF2, 1>(W, glane, h, o);
            sn = sigmoidf(o[0][0]);
        }
        B2_MARK(2);
        // cross-view weights (ibrnet.py:334-340): weight = mask / (sum mask + 1e-8), weight0 = sn * weight.  The statistics themselves
        // (mean_k = sum_v w_k x, var_k = sum_v w_k (x - mean_k)^2) and base_fc.0's per-point part are the forward's (stash, xch).
        float wv, w0, sa0, sa1;
        {
            const float msum = svt[kSavedMsumRow * 64 + lane];
            wv = mask / (msum + 1e-8f);
            w0 = sn * wv;
            sa0 = svt[kSavedSw0Row * 64 + lane];
            sa1 = msum / (msum + 1e-8f);
            __syncthreads();                                   // every wave's DMA pieces have landed (the fence waits for them)
        }
        B2_MARK(3);
        B2_MARK(4);
        // base_fc -> x;  h64 is recomputed in the backward from xch
        float x[1][8];
        auto base_hidden = [&](float (&h64)[1][16]) {
            float xq[1][16], x1[1][1];
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) { xq[0][k] = gi[k]; xq[0][8 + k] = e[0][k]; }
            x1[0][0] = sel4(g, gr[0], gr[1], gr[2], 0.0f);
            v4f a0[1][2], a1[1][2];
            NR_PRAGMA_UNROLL
            for (int mo = 0; mo < 2; ++mo)
                NR_PRAGMA_UNROLL
                for (int r_ = 0; r_ < 4; ++r_) { a0[0][mo][r_] = xch[(mo * 4 + r_) * 64 + lane]; a1[0][mo][r_] = xch[((2 + mo) * 4 + r_) * 64 + lane]; }
            LayerPre<L_BV0> p0; LayerPre<L_BV1> p1; NoLayer last;
            layer_prefetch<L_BV0>(W, glane, p0);
            layer_acc<L_BV0, 1>(W, glane, p0, xq, x1, a0, last);
            layer_prefetch<L_BV1>(W, glane, p1);
            layer_acc<L_BV1, 1>(W, glane, p1, xq, x1, a1, last);
            NR_PRAGMA_UNROLL
            for (int mo = 0; mo < 2; ++mo)
                NR_PRAGMA_UNROLL
                for (int r_ = 0; r_ < 4; ++r_) { h64[0][4 * mo + r_] = elu_s(a0[0][mo][r_]); h64[0][8 + 4 * mo + r_] = elu_s(a1[0][mo][r_]); }
        };
        {
            float h64[1][16];
            base_hidden(h64);
            layer_fwd<L_B2, 1, ACT_ELU>(W, glane, h64, none, x);
        }
        // vis_fc -> x2, visp;  vis_fc2 -> vis2;  rgb_fc -> z
        float x2[1][8], visp, yv32, vis2, v2sig, z;
        {
            float xin[1][8], h[1][8], y[1][8], yv[1][1], o[1][1];
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) xin[0][k] = x[0][k] * wv;
            layer_fwd<L_VF1, 1, ACT_ELU>(W, glane, xin, none, h);
            layer_fwd<L_VF2, 1, ACT_ELU>(W, glane, h, none, y);
            layer_vec<L_VF2, 1>(W, glane, h, yv);
            yv32 = elu(yv[0][0]);
            visp = sigmoidf(yv32) * mask;
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) { x2[0][k] = x[0][k] + y[0][k]; xin[0][k] = x2[0][k] * visp; }
            layer_fwd<L_V21, 1, ACT_ELU>(W, glane, xin, none, h);
            layer_vec<L_V22, 1>(W, glane, h, o);
            v2sig = sigmoidf(o[0][0]);
            vis2 = v2sig * mask;
            float x1[1][2], h16[1][4], h8[1][4];
            x1[0][0] = sel4(g, vis2, dlt[0], dlt[1], dlt[2]);
            x1[0][1] = sel4(g, dlt[3], 0.0f, 0.0f, 0.0f);
            layer_fwd<L_RF1, 1, ACT_ELU>(W, glane, x2, x1, h16);
            layer_fwd<L_RF2, 1, ACT_ELU>(W, glane, h16, none, h8);
            layer_vec<L_RF3, 1>(W, glane, h8, o);
            z = mask > 0.0f ? o[0][0] : -1e9f;
        }
        B2_MARK(5);
        // softmax blend weights (ibrnet.py:350-354,366-367) from the forward's max z / sum exp / sum vis''; the visibility-weighted mean
        // and variance stay in xg (rows 16..31) until the blend backward
        float beta, svis, wh, swh;
        const float* gm_l = xg + 16 * 64 + lane;              // weighted mean row k: gm_l[k * 64], variance: gm_l[(8 + k) * 64]
        {
            const float ez = vok ? nr_fast_exp(z - xg[36 * 64 + lane]) : 0.0f;     // (padding waves take no part in the softmax)
            beta = ez / xg[37 * 64 + lane];
            svis = xg[38 * 64 + lane];
            wh = vis2 / (svis + 1e-8f);
            swh = svis / (svis + 1e-8f);
        }
        const float meanw = swh * inv_rfn;

        B2_MARK(6);
        // ================= backward =================
        const float* up = p.d_point_rec + (size_t)pi * kPointRec;
        const float gsc = pvalid ? 1.0f : 0.0f;
        float dgi[8], dgr[3], de[1][8], dsn;
        if constexpr (DO_TAIL) {
        // ---- geometry_fc (per point; ibrnet.py:353-354).  Hidden layer h and output G are the forward's (xg rows 0..15, 32..35).
        // Waves 0..3 each redo the small transposed geometry_fc.2 and take output tile `wave` of geometry_fc.0^T.
        float dgm[8], dgv[8], dmeanw;                          // d mean, d var (natural D layout), d mean weight
        {
            // per-point staging (stride kB2PStride): rows 0 d Gpre (16), 16 h64 (64), 80 d h64 (64), 144 input (65 -> 80)
            float* SP = S;
            B2_MARK(7);
            if (wave < 4) {
                float h[1][16], dG[1][4], dh[1][16];
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) h[0][k] = xg[k * 64 + lane];
                const float4 u4 = ld4(up + 4 * g);
                dG[0][0] = u4.x * gsc * delu(xg[32 * 64 + lane]); dG[0][1] = u4.y * gsc * delu(xg[33 * 64 + lane]);
                dG[0][2] = u4.z * gsc * delu(xg[34 * 64 + lane]); dG[0][3] = u4.w * gsc * delu(xg[35 * 64 + lane]);
                B2_MARK(9);
                layer_fwd<LT_GF2, 1, ACT_NONE>(WT, glane, dG, none, dh);
                B2_MARK(11);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) dh[0][k] *= delu_s(h[0][k]);
                v4f a4[1];
                a4[0][0] = 0.0f; a4[0][1] = 0.0f; a4[0][2] = 0.0f; a4[0][3] = 0.0f;
                layer_tile<LT_GF1, 1>(WT, glane, wave, dh, none, a4);
                // hand d mean / d var / d mean weight to every wave (hx row 4 tile + r: d mean rows 0..7, d var rows 8..15)
                NR_PRAGMA_UNROLL
                for (int r_ = 0; r_ < 4; ++r_) hx[(wave * 4 + r_) * 64 + lane] = a4[0][r_];
                if (wave == 0) {
                    float dmw[1][1];
                    layer_vec<LT_GF1, 1>(WT, glane, dh, dmw);
                    hx[16 * 64 + lane] = dmw[0][0];
                    st_nat<4>(SP, kB2PStride, 0, dG[0], c, g);
                    st_nat<16>(SP, kB2PStride, 16, h[0], c, g);
                }
                B2_MARK(12);
                if (wave == 1) st_nat<16>(SP, kB2PStride, 80, dh[0], c, g);
                if (wave == 2) {
                    float gmv[16];
                    NR_PRAGMA_UNROLL
                    for (int k = 0; k < 16; ++k) gmv[k] = gm_l[k * 64];
                    st_nat<16>(SP, kB2PStride, 144, gmv, c, g);           // rows 144..175 mean, 176..207 variance
                    st_one(SP, kB2PStride, 208, meanw, c, g);
                }
                B2_MARK(13);
            }
            __syncthreads();
            B2_MARK(14);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) { dgm[k] = hx[k * 64 + lane]; dgv[k] = hx[(8 + k) * 64 + lane]; }
            dmeanw = hx[16 * 64 + lane];
            dw_jobs<DW_GF2>(acc, bacc, SP, 0, 16, wave, lane);
            dw_jobs<DW_GF0>(acc, bacc, SP, 80, 144, wave, lane);
            B2_MARK(15);
        }
        // ---- visibility-weighted mean / variance + softmax blend -> d x2, d vis2, d z
        float dx2[1][8], dvis2, dz;
        {
            float part = 0.0f;
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) {
                const float xv = x2[0][k], mean = gm_l[k * 64];
                const float dmt = dgm[k] - 2.0f * dgv[k] * mean * (1.0f - swh);
                dx2[0][k] = wh * (dmt + 2.0f * (xv - mean) * dgv[k]);
                part += dmt * xv + dgv[k] * (xv - mean) * (xv - mean);
            }
            const float dwh = dmeanw * inv_rfn + nr_group_sum(part);            // the 32 features sit in 8 registers x 4 lane groups
            const float dbeta = (up[16] * rgb[0] + up[17] * rgb[1] + up[18] * rgb[2]) * gsc;
            float s2[2] = {dwh * wh, beta * dbeta};
            b2_allsum<2>(s2, red, wave, lane);
            dvis2 = (dwh - s2[0]) / (svis + 1e-8f);
            dz = beta * (dbeta - s2[1]);
            if (!(mask > 0.0f)) dz = 0.0f;
        }
        B2_MARK(16);
        // ---- rgb_fc backward (ibrnet.py:363-365)
        {
            float x1[1][2], h16[1][4], h8[1][4], d8[1][4], d16[1][4], dxa[1][8], o1[1][1];
            x1[0][0] = sel4(g, vis2, dlt[0], dlt[1], dlt[2]);
            x1[0][1] = sel4(g, dlt[3], 0.0f, 0.0f, 0.0f);
            layer_fwd<L_RF1, 1, ACT_ELU>(W, glane, x2, x1, h16);
            layer_fwd<L_RF2, 1, ACT_ELU>(W, glane, h16, none, h8);
            VecPre<L_RF3> p3;
            layer_prefetch<L_RF3>(W, glane, p3);
            float dy1[1] = {dz};
            vec_bwd<L_RF3>(p3, dy1, d8[0]);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 4; ++k) d8[0][k] *= delu_s(h8[0][k]);
            layer_fwd<LT_RF2, 1, ACT_NONE>(WT, glane, d8, none, d16);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 4; ++k) d16[0][k] *= delu_s(h16[0][k]);
            layer_fwd<LT_RF1, 1, ACT_NONE>(WT, glane, d16, none, dxa);
            layer_vec<LT_RF1, 1>(WT, glane, d16, o1);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) dx2[0][k] += dxa[0][k];
            dvis2 += o1[0][0];
            // rows: 0 dz (16), 16 h8 (16), 32 d8 (16), 48 h16 (16), 64 d16 (16), 80 [x2 32, vis2, dl 4] (48)
            B2_MARK(17);
            __syncthreads();
            B2_MARK(18);
            st_one(S, kB2Stride, 0, dz, col, g);
            st_nat<4>(S, kB2Stride, 16, h8[0], col, g); st_nat<4>(S, kB2Stride, 32, d8[0], col, g);
            st_nat<4>(S, kB2Stride, 48, h16[0], col, g); st_nat<4>(S, kB2Stride, 64, d16[0], col, g);
            st_nat<8>(S, kB2Stride, 80, x2[0], col, g);
            st_one(S, kB2Stride, 112, vis2, col, g);
            NR_PRAGMA_UNROLL
            for (int j = 0; j < 4; ++j) st_one(S, kB2Stride, 113 + j, dlt[j], col, g);
            B2_MARK(19);
            __syncthreads();
            B2_MARK(20);
            dw_jobs<DW_RF4>(acc, bacc, S, 0, 16, wave, lane);
            dw_jobs<DW_RF2>(acc, bacc, S, 32, 48, wave, lane);
            dw_jobs<DW_RF0>(acc, bacc, S, 64, 80, wave, lane);
            B2_MARK(21);
        }
        // ---- vis_fc2 backward (ibrnet.py:347-348): vis2 = sigmoid(a) * mask
        float dvisp;
        {
            float xin[1][8], h[1][8], dh[1][8], dxin[1][8];
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) xin[0][k] = x2[0][k] * visp;
            layer_fwd<L_V21, 1, ACT_ELU>(W, glane, xin, none, h);
            const float da = dvis2 * mask * v2sig * (1.0f - v2sig);
            VecPre<L_V22> pv;
            layer_prefetch<L_V22>(W, glane, pv);
            float dy1[1] = {da};
            vec_bwd<L_V22>(pv, dy1, dh[0]);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) dh[0][k] *= delu_s(h[0][k]);
            layer_fwd<LT_V21, 1, ACT_NONE>(WT, glane, dh, none, dxin);
            float dv = 0.0f;
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) { dv = fmaf(dxin[0][k], x2[0][k], dv); dx2[0][k] = fmaf(dxin[0][k], visp, dx2[0][k]); }
            dvisp = nr_group_sum(dv);
            // rows: 0 da (16), 16 h (32), 48 dh (32), 80 xin (32)
            __syncthreads();
            st_one(S, kB2Stride, 0, da, col, g);
            st_nat<8>(S, kB2Stride, 16, h[0], col, g); st_nat<8>(S, kB2Stride, 48, dh[0], col, g); st_nat<8>(S, kB2Stride, 80, xin[0], col, g);
            __syncthreads();
            dw_jobs<DW_V22>(acc, bacc, S, 0, 16, wave, lane);
            dw_jobs<DW_V20>(acc, bacc, S, 48, 80, wave, lane);
        }
        B2_MARK(22);
        // ---- vis_fc backward (ibrnet.py:343-346): x2 = x + y, visp = sigmoid(ELU(y32)) * mask; dx2 becomes d x
        float dx[1][8];
        {
            float xin[1][8], h[1][8], y[1][8], dy[1][8], dh[1][8], dxin[1][8], x1[1][1];
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) xin[0][k] = x[0][k] * wv;
            layer_fwd<L_VF1, 1, ACT_ELU>(W, glane, xin, none, h);
            layer_fwd<L_VF2, 1, ACT_ELU>(W, glane, h, none, y);
            const float sg = sigmoidf(yv32);
            const float dy32 = dvisp * mask * sg * (1.0f - sg) * delu(yv32);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) dy[0][k] = dx2[0][k] * delu(y[0][k]);
            x1[0][0] = sel4(g, dy32, 0.0f, 0.0f, 0.0f);
            layer_fwd<LT_VF2, 1, ACT_NONE>(WT, glane, dy, x1, dh);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) dh[0][k] *= delu_s(h[0][k]);
            layer_fwd<LT_VF1, 1, ACT_NONE>(WT, glane, dh, none, dxin);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) dx[0][k] = fmaf(dxin[0][k], wv, dx2[0][k]);
            // rows: 0 dy (33 -> 48), 48 h (32), 80 dh (32), 112 xin (32)
            __syncthreads();
            st_nat<8>(S, kB2Stride, 0, dy[0], col, g); st_one(S, kB2Stride, 32, dy32, col, g);
            st_nat<8>(S, kB2Stride, 48, h[0], col, g); st_nat<8>(S, kB2Stride, 80, dh[0], col, g); st_nat<8>(S, kB2Stride, 112, xin[0], col, g);
            __syncthreads();
            dw_jobs<DW_VF2>(acc, bacc, S, 0, 48, wave, lane);
            dw_jobs<DW_VF0>(acc, bacc, S, 80, 112, wave, lane);
        }
        B2_MARK(23);
        // ---- base_fc backward (ibrnet.py:342) -> d gi, d gr, d e, d (statistics)
        {
            float h64[1][16], dxp[1][8], dh64[1][16], dcat[1][16], drgb[1][3];
            base_hidden(h64);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) dxp[0][k] = dx[0][k] * delu(x[0][k]);
            layer_fwd<LT_B2, 1, ACT_NONE>(WT, glane, dxp, none, dh64);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) dh64[0][k] *= delu_s(h64[0][k]);
            layer_fwd<LT_BV, 1, ACT_NONE>(WT, glane, dh64, none, dcat);
            layer_vec<LT_BV, 1>(WT, glane, dh64, drgb);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) { dgi[k] = dcat[0][k]; de[0][k] = dcat[0][8 + k]; }
            NR_PRAGMA_UNROLL
            for (int j = 0; j < 3; ++j) dgr[j] = drgb[0][j];
            // round 1 rows: 0 dxp (32), 32 h64 (64)
            B2_MARK(24);
            __syncthreads();
            B2_MARK(25);
            st_nat<8>(S, kB2Stride, 0, dxp[0], col, g); st_nat<16>(S, kB2Stride, 32, h64[0], col, g);
            __syncthreads();
            dw_jobs<DW_B2>(acc, bacc, S, 0, 32, wave, lane);
            B2_MARK(26);
            // round 2 rows: 0 dh64 (64), 64 [rgb 3 | img 32 | e 32] (67 -> 80): the natural column order 140..206 of base_fc.0
            __syncthreads();
            st_nat<16>(S, kB2Stride, 0, dh64[0], col, g);
            NR_PRAGMA_UNROLL
            for (int j = 0; j < 3; ++j) st_one(S, kB2Stride, 64 + j, gr[j], col, g);
            st_gat(S, kB2Stride, 67, gi, col, g);
            st_nat<8>(S, kB2Stride, 99, e[0], col, g);
            __syncthreads();
            dw_jobs<DW_BV>(acc, bacc, S, 0, 64, wave, lane);
            B2_MARK(27);
            // per-point part: sum over the views of d h64, then d statistics = W_gl^T (sum d h64) by owner waves
            float sd16[1][16];
            {
                float part[kB2Rmax];
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 12; ++k) part[k] = dh64[0][k];
                b2_allsum<kB2Rmax>(part, red, wave, lane);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 12; ++k) sd16[0][k] = part[k];
                float p4[4] = {dh64[0][12], dh64[0][13], dh64[0][14], dh64[0][15]};
                b2_allsum<4>(p4, red, wave, lane);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 4; ++k) sd16[0][12 + k] = p4[k];
                B2_MARK(28);
            }
            // d statistic tile `wave` of LT_BG (8 tiles: statistic wave / 2, half wave % 2 of its 32 gathered channels):
            //   register r of lane group g of tile 2j + mo = channel 8g + 4mo + r of statistic j
            {
                v4f a8[1];
                a8[0][0] = 0.0f; a8[0][1] = 0.0f; a8[0][2] = 0.0f; a8[0][3] = 0.0f;
                layer_tile<LT_BG, 1>(WT, glane, wave, sd16, none, a8);
                // hand-off area hx: [44 rows][64 lanes] in the lane layout of the statistics: row 11 j + k (k < 8: channel
                // 8g + k), row 11 j + 8 + i (rgb i)
                const int j = wave >> 1, mo = wave & 1;
                __syncthreads();                               // (every wave is done with the geometry hand-off and the DW_BV jobs)
                NR_PRAGMA_UNROLL
                for (int r_ = 0; r_ < 4; ++r_) hx[(11 * j + 4 * mo + r_) * 64 + lane] = a8[0][r_];
                if (wave < 4) {
                    float o3[1][3];
                    if (wave == 0) layer_vec<LT_BG_R0, 1>(WT, glane, sd16, o3);
                    else if (wave == 1) layer_vec<LT_BG_R1, 1>(WT, glane, sd16, o3);
                    else if (wave == 2) layer_vec<LT_BG_R2, 1>(WT, glane, sd16, o3);
                    else layer_vec<LT_BG_R3, 1>(WT, glane, sd16, o3);
                    NR_PRAGMA_UNROLL
                    for (int i = 0; i < 3; ++i) hx[(11 * wave + 8 + i) * 64 + lane] = o3[0][i];
                }
                // per-point staging for dW of the statistics columns: rows 0 sum d h64 (64), 64 statistics in the natural column
                // order [mean0 35 | var0 35 | mean1 35 | var1 35]: statistic j -> rows 64 + 35 j + {rgb 0..2, 3 + channel}
                float* SP = S;
                if (wave == 5) st_nat<16>(SP, kB2PStride, 0, sd16[0], c, g);
                if (wave >= 4) {                               // statistic jj = wave - 4
                    const int jj = wave - 4;
                    NR_PRAGMA_UNROLL
                    for (int k = 0; k < 8; ++k) SP[(64 + 35 * jj + 3 + 8 * g + k) * kB2PStride + c] = stash[(11 * jj + k) * 64 + lane];
                    if (g == 0) {
                        NR_PRAGMA_UNROLL
                        for (int i = 0; i < 3; ++i) SP[(64 + 35 * jj + i) * kB2PStride + c] = stash[(11 * jj + 8 + i) * 64 + lane];
                    }
                }
                __syncthreads();
                B2_MARK(29);
                dw_jobs<DW_BG>(acc, bacc, SP, 0, 64, wave, lane);
                B2_MARK(30);
            }
        }
        // ---- cross-view statistics backward: d statistics (per point, in hx rows 0..43) -> d gi / d gr +=, d sn
        {
            float dw0 = 0.0f, dw0r = 0.0f;
            NR_PRAGMA_UNROLL
            for (int q = 0; q < 11; ++q) {
                const float xv = q < 8 ? gi[q] : gr[q - 8];
                const float m0 = stash[q * 64 + lane], m1 = stash[(22 + q) * 64 + lane];
                const float dm0 = hx[q * 64 + lane], dv0 = hx[(11 + q) * 64 + lane];
                const float dm1 = hx[(22 + q) * 64 + lane], dv1 = hx[(33 + q) * 64 + lane];
                const float dmt0 = dm0 - 2.0f * dv0 * m0 * (1.0f - sa0);
                const float dmt1 = dm1 - 2.0f * dv1 * m1 * (1.0f - sa1);
                const float add = w0 * (dmt0 + 2.0f * (xv - m0) * dv0) + wv * (dmt1 + 2.0f * (xv - m1) * dv1);
                const float t0 = dmt0 * xv + dv0 * (xv - m0) * (xv - m0);
                if (q < 8) { dgi[q] += add; dw0 += t0; } else { dgr[q - 8] += add; dw0r += t0; }
            }
            dsn = (nr_group_sum(dw0) + dw0r) * wv;              // img channels: 8 registers x 4 lane groups; rgb replicated
        }
        } else {                                               // front kernel: what the tail kernel left for this lane
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) { dgi[k] = ho[k * 64]; de[0][k] = ho[(8 + k) * 64]; }
            dgr[0] = ho[16 * 64]; dgr[1] = ho[17 * 64]; dgr[2] = ho[18 * 64]; dsn = ho[19 * 64];
        }
        if constexpr (DO_TAIL) {
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) { ho[k * 64] = dgi[k]; ho[(8 + k) * 64] = de[0][k]; }
            ho[16 * 64] = dgr[0]; ho[17 * 64] = dgr[1]; ho[18 * 64] = dgr[2]; ho[19 * 64] = dsn;
            __syncthreads();                                   // (the next tile's DMA overwrites xch / stash / xg)
        }
        if constexpr (DO_FRONT) {
        B2_MARK(31);
        // ---- neuray_fc backward -> d e +=            (rows: 0 do (16), 16 h8 (16), 32 dh8 (16), 48 e (32))
        // ---- ray_dir_fc backward (weights only)      (rows: 80 dy35 (48), 128 h16 (16), 144 dh16 (16), 160 dl (16))
        {
            float h8[1][4], dh8[1][4], dea[1][8];
            layer_fwd<L_NF1, 1, ACT_ELU>(W, glane, e, none, h8);
            const float d_o = dsn * sn * (1.0f - sn);
            VecPre<L_NF2> pn;
            layer_prefetch<L_NF2>(W, glane, pn);
            float dy1[1] = {d_o};
            vec_bwd<L_NF2>(pn, dy1, dh8[0]);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 4; ++k) dh8[0][k] *= delu_s(h8[0][k]);
            layer_fwd<LT_NF1, 1, ACT_NONE>(WT, glane, dh8, none, dea);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) de[0][k] += dea[0][k];
            float x1[1][1], h16[1][4], df[1][8], dc[1][3], dyi[1][8], dyr[3], dh16[1][4], xr[1][1];
            x1[0][0] = sel4(g, dlt[0], dlt[1], dlt[2], dlt[3]);
            layer_fwd<L_RD1, 1, ACT_ELU>(W, glane, none, x1, h16);
            layer_fwd<L_RD2, 1, ACT_ELU>(W, glane, h16, none, df);
            layer_vec<L_RD2, 1>(W, glane, h16, dc);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) dyi[0][k] = dgi[k] * delu(df[0][k]);
            NR_PRAGMA_UNROLL
            for (int j = 0; j < 3; ++j) dyr[j] = dgr[j] * delu(elu(dc[0][j]));
            xr[0][0] = sel4(g, dyr[0], dyr[1], dyr[2], 0.0f);
            layer_fwd<LT_RD2, 1, ACT_NONE>(WT, glane, dyi, xr, dh16);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 4; ++k) dh16[0][k] *= delu_s(h16[0][k]);
            __syncthreads();
            st_one(S, kB2Stride, 0, d_o, col, g);
            st_nat<4>(S, kB2Stride, 16, h8[0], col, g); st_nat<4>(S, kB2Stride, 32, dh8[0], col, g); st_nat<8>(S, kB2Stride, 48, e[0], col, g);
            NR_PRAGMA_UNROLL
            for (int j = 0; j < 3; ++j) st_one(S, kB2Stride, 80 + j, dyr[j], col, g);
            st_gat(S, kB2Stride, 83, dyi[0], col, g);
            st_nat<4>(S, kB2Stride, 128, h16[0], col, g); st_nat<4>(S, kB2Stride, 144, dh16[0], col, g);
            NR_PRAGMA_UNROLL
            for (int j = 0; j < 4; ++j) st_one(S, kB2Stride, 160 + j, dlt[j], col, g);
            __syncthreads();
            dw_jobs<DW_NF2>(acc, bacc, S, 0, 16, wave, lane);
            dw_jobs<DW_NF0>(acc, bacc, S, 32, 48, wave, lane);
            dw_jobs<DW_RD2>(acc, bacc, S, 80, 128, wave, lane);
            dw_jobs<DW_RD0>(acc, bacc, S, 144, 160, wave, lane);
        }
        B2_MARK(32);
        // ---- prob_embed backward -> d f_ray, d hit, d vis     (rows: 0 de (32), 32 h (32), 64 dh (32), 96 [f_ray 32, hit', vis'] (48))
        float dfr[8], dhit, dvis;
        {
            float x1[1][1], h[1][8], dh[1][8], dxa[1][8], o2[1][2];
            const float hp = (hit - 0.5f) * 2.0f, vp_ = (vis - 0.5f) * 2.0f;
            x1[0][0] = sel4(g, hp, vp_, 0.0f, 0.0f);
            layer_fwd<L_PE1, 1, ACT_RELU>(W, glane, fray, x1, h);
            layer_fwd<LT_PE2, 1, ACT_NONE>(WT, glane, de, none, dh);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) dh[0][k] = h[0][k] > 0.0f ? dh[0][k] : 0.0f;
            layer_fwd<LT_PE1, 1, ACT_NONE>(WT, glane, dh, none, dxa);
            layer_vec<LT_PE1, 1>(WT, glane, dh, o2);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) dfr[k] = dxa[0][k];
            dhit = 2.0f * o2[0][0]; dvis = 2.0f * o2[0][1];
            __syncthreads();
            st_nat<8>(S, kB2Stride, 0, de[0], col, g); st_nat<8>(S, kB2Stride, 32, h[0], col, g); st_nat<8>(S, kB2Stride, 64, dh[0], col, g);
            st_gat(S, kB2Stride, 96, fray[0], col, g);
            st_one(S, kB2Stride, 128, hp, col, g); st_one(S, kB2Stride, 129, vp_, col, g);
            __syncthreads();
            dw_jobs<DW_PE2>(acc, bacc, S, 0, 32, wave, lane);
            dw_jobs<DW_PE0>(acc, bacc, S, 64, 96, wave, lane);
        }
        B2_MARK(33);
        // ---- probabilities backward (dist_decoder.py:109-140) and the dist decoder heads
        {
            float dmu0 = 0.0f, dmu1 = 0.0f, dsd0 = 0.0f, dsd1 = 0.0f, daw = 0.0f, dnu = 0.0f;
            b2_prob_bwd(tref - lo, tref + hi, mu0, mu1, s0, s1, aw, nuu, use_vis, dvis * mask, dhit * mask, dmu0, dmu1, dsd0, dsd1, daw, dnu);
            // through the output non-linearities: softplus' = 1 - exp(-softplus), sigmoid' = s (1 - s)
            const float dm[2] = {dmu0 * (1.0f - nr_fast_exp(-mu0)), dmu1 * (1.0f - nr_fast_exp(-mu1))};
            const float dv[2] = {dsd0 * (1.0f - nr_fast_exp(-(s0 - p.var_bias))), dsd1 * (1.0f - nr_fast_exp(-(s1 - p.var_bias)))};
            const float da[1] = {daw * aw * (1.0f - aw)};
            b2_dist_head_bwd<L_DM1, L_DM2, L_DFIN_M, LT_DM1, LT_DM2, DW_M4, DW_M2, DW_M0, 2>(W, WT, glane, lane, wave, col, g, fray, dm, dfr, S, acc, bacc B2_MARK_PASS);
            b2_dist_head_bwd<L_DV1, L_DV2, L_DFIN_V, LT_DV1, LT_DV2, DW_V4, DW_V2, DW_V0, 2>(W, WT, glane, lane, wave, col, g, fray, dv, dfr, S, acc, bacc B2_MARK_PASS);
            b2_dist_head_bwd<L_DA1, L_DA2, L_DFIN_A, LT_DA1, LT_DA2, DW_A4, DW_A2, DW_A0, 1>(W, WT, glane, lane, wave, col, g, fray, da, dfr, S, acc, bacc B2_MARK_PASS);
            if constexpr (HAS_VIS) {
                const float ds[1] = {dnu * nu * (1.0f - nu)};
                b2_dist_head_bwd<L_DS1, L_DS2, L_DFIN_S, LT_DS1, LT_DS2, DW_S4, DW_S2, DW_S0, 1>(W, WT, glane, lane, wave, col, g, fray, ds, dfr, S, acc, bacc B2_MARK_PASS);
            }
        }
        B2_MARK(41);
        // ---- gathers backward: f_ray = mask * bilinear(ray_feats), f_img = mask * bilinear(img_feats) (render_ops.py:54-70).
        // The wave's 16 columns go through a private LDS slab [16 points][64 channels: 32 ray | 32 img] so that 32 lanes add the
        // 32 contiguous channels of one texel with one instruction; the two halves of the wave take two taps at a time.
        {
            __syncthreads();                                   // the staging area is free (last weight-gradient jobs are done)
            float* slab = S + wave * (16 * 72);
            const float sc = (vok && pvalid) ? mask : 0.0f;
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) { slab[c * 72 + 8 * g + k] = dfr[k]; slab[c * 72 + 32 + 8 * g + k] = dgi[k]; }
            if (g == 0) {
                slab[c * 72 + 64] = __int_as_float(tfs.o00); slab[c * 72 + 65] = __int_as_float(tfs.o10);
                slab[c * 72 + 66] = __int_as_float(tfs.o01); slab[c * 72 + 67] = __int_as_float(tfs.o11);
                slab[c * 72 + 68] = tfs.w00 * sc; slab[c * 72 + 69] = tfs.w10 * sc; slab[c * 72 + 70] = tfs.w01 * sc; slab[c * 72 + 71] = tfs.w11 * sc;
            }
            __syncthreads();
            const int ch = lane & 31, half = lane >> 5;
            const size_t voff = (size_t)view * fmap;
            for (int l = 0; l < 16; ++l) {
                const float g_r = slab[l * 72 + ch], g_i = slab[l * 72 + 32 + ch];
                NR_PRAGMA_UNROLL
                for (int tp = 0; tp < 2; ++tp) {
                    const int tap = 2 * tp + half;
                    const float wt = slab[l * 72 + 68 + tap];
                    if (wt != 0.0f) {
                        const size_t o = voff + (size_t)__float_as_int(slab[l * 72 + 64 + tap]) * 32 + ch;
                        atomicAdd(p.d_ray_feats + o, wt * g_r);
                        atomicAdd(p.d_img_feats + o, wt * g_i);
                    }
                }
            }
            __syncthreads();
            B2_MARK(42);
        }
    
        }
    }
    // ================= end of the launch: accumulated weight gradients -> global =================
    if constexpr (DO_TAIL)
        dw_flush_all<DW_RF4, DW_RF2, DW_RF0, DW_V22, DW_V20, DW_VF2, DW_VF0, DW_B2, DW_BV, DW_GF2, DW_GF0, DW_BG>(acc, bacc, p.d_flat, wave, lane);
    if constexpr (DO_FRONT) {
        dw_flush_all<DW_NF2, DW_NF0, DW_RD2, DW_RD0, DW_PE2, DW_PE0, DW_M4, DW_M2, DW_M0, DW_V4, DW_V2, DW_V0, DW_A4, DW_A2, DW_A0>(acc, bacc, p.d_flat, wave, lane);
        if constexpr (HAS_VIS) dw_flush_all<DW_S4, DW_S2, DW_S0>(acc, bacc, p.d_flat, wave, lane);
    }
}

// =====================================================================================================================
// self_hit_backward2_kernel: backward of the a19 path (renderer.py:137-155; dist_decoder.py:99-107,109-140,146-151) on the
// resident scheme: hit_prob_self [rn][dn] as a function of the gathered query-view feature [rn][32] and the dist decoder.
// One wave per tile of 16 rays (lane = (ray c, lane group g)); the decoder runs in registers on the packed / transposed
// packs exactly as in points_backward2_kernel; the probability backward of a ray's dn samples is split over its four lane
// groups; the weight gradients contract the 16 rays of the tile (4 MFMA per 16 x 16 tile, operands staged in LDS) into
// register accumulators that go to global memory once per wave.  (The first version, self_hit_backward_kernel: lane = ray,
// activations in a global arena, 8 waves for 512 rays - 0.26 ms per pass, 11 % of a training step.)
// =====================================================================================================================
struct SelfHitBwd2Params {
    const float* que_const;
    const float* depth;        // [rn][dn]
    const float* feats;        // [rn][32] gathered query-view ray feature
    const float* weights;      // packed pass weights
    const float* weights_t;    // packed transposed layers
    const float* d_hit;        // [rn][dn]
    float* d_feats;            // [rn][32]
    float* d_flat;             // accumulated (dist decoder tensors only)
    int rn, dn, use_vis;
    float var_bias;
};

constexpr int kShIds[12] = {DW_M4, DW_M2, DW_M0, DW_V4, DW_V2, DW_V0, DW_A4, DW_A2, DW_A0, DW_S4, DW_S2, DW_S0};
constexpr int sh_job0(int id) {          // first accumulator of weight tensor id (DW_COUNT: the total)
    int j = 0;
    for (int i = 0; i < 12 && kShIds[i] != id; ++i) j += dw_ot(kShIds[i]) * dw_kt(kShIds[i]);
    return j;
}
constexpr int sh_bias0(int id) {
    int j = 0;
    for (int i = 0; i < 12 && kShIds[i] != id; ++i) j += dw_ot(kShIds[i]);
    return j;
}
constexpr int kShAcc = sh_job0(DW_COUNT), kShBias = sh_bias0(DW_COUNT);
constexpr int kShStageRows = 176;

// all 16 x 16 tiles of weight tensor ID: acc += dY[rows rowA..] X[rows rowB..]^T over the tile's 16 columns
template <int ID>
__device__ __forceinline__ void sh_tiles(v4f (&acc)[kShAcc], float (&bacc)[kShBias], const float* S, int rowA, int rowB, int lane) {
    constexpr int OT = dw_ot(ID), KT = dw_kt(ID), J0 = sh_job0(ID), B0 = sh_bias0(ID);
    const int m = lane & 15, kk = lane >> 4;
    NR_PRAGMA_UNROLL
    for (int a = 0; a < OT; ++a) {
        const float4 av = ld4(S + (rowA + 16 * a + m) * kB2PStride + 4 * kk);
        NR_PRAGMA_UNROLL
        for (int b = 0; b < KT; ++b) {
            const float4 bv = ld4(S + (rowB + 16 * b + m) * kB2PStride + 4 * kk);
            v4f d = acc[J0 + a * KT + b];
            d = nr_mfma16(av.x, bv.x, d); d = nr_mfma16(av.y, bv.y, d);
            d = nr_mfma16(av.z, bv.z, d); d = nr_mfma16(av.w, bv.w, d);
            acc[J0 + a * KT + b] = d;
        }
        bacc[B0 + a] += (av.x + av.y) + (av.z + av.w);
    }
}
template <int ID>
__device__ __forceinline__ void sh_flush(const v4f (&acc)[kShAcc], const float (&bacc)[kShBias], float* d_flat, int lane) {
    constexpr int OT = dw_ot(ID), KT = dw_kt(ID), J0 = sh_job0(ID), B0 = sh_bias0(ID);
    constexpr DwSpec sp = kDw[ID];
    const float xs = sp.xscaled ? (float)(1.0 / kLog2e) : 1.0f;
    const int m = lane & 15, kk = lane >> 4;
    NR_PRAGMA_UNROLL
    for (int a = 0; a < OT; ++a) {
        NR_PRAGMA_UNROLL
        for (int b = 0; b < KT; ++b) {
            const v4f d = acc[J0 + a * KT + b];
            NR_PRAGMA_UNROLL
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * a + 4 * kk + r, k = 16 * b + m;
                if (o < sp.O && k < sp.K) atomicAdd(d_flat + tensor_offset(sp.tw) + o * sp.ldw + sp.col0 + k, d[r] * xs);
            }
        }
        const float sb = nr_group_sum(bacc[B0 + a]);
        const int o = 16 * a + m;
        if (kk == 0 && o < sp.O) atomicAdd(d_flat + tensor_offset(sp.tb) + o, sb);
    }
}

template <int L1, int L2, int LF, int T1, int T2, int D4, int D2, int D0, int NOUT>
__device__ __forceinline__ void sh_head_bwd(nr_wbuf W, nr_wbuf WT, int wlane, int lane, const float (&fray)[1][8], const float (&dout)[NOUT],
                                            float (&dfr)[8], float* S, v4f (&acc)[kShAcc], float (&bacc)[kShBias]) {
    const int g = lane >> 4, c = lane & 15;
    float none[1][1] = {{0.0f}}, h1[1][8], h2[1][8], dh2[1][8], dh1[1][8], dx[1][8];
    layer_fwd<L1, 1, ACT_ELU>(W, wlane, fray, none, h1);
    layer_fwd<L2, 1, ACT_ELU>(W, wlane, h1, none, h2);
    VecPre<LF> pf;
    layer_prefetch<LF>(W, wlane, pf);
    float dy[kVec[LF].n];
    NR_PRAGMA_UNROLL
    for (int j = 0; j < NOUT; ++j) dy[j] = dout[j];
    vec_bwd<LF>(pf, dy, dh2[0]);
    NR_PRAGMA_UNROLL
    for (int k = 0; k < 8; ++k) dh2[0][k] *= delu_s(h2[0][k]);
    layer_fwd<T2, 1, ACT_NONE>(WT, wlane, dh2, none, dh1);
    NR_PRAGMA_UNROLL
    for (int k = 0; k < 8; ++k) dh1[0][k] *= delu_s(h1[0][k]);
    layer_fwd<T1, 1, ACT_NONE>(WT, wlane, dh1, none, dx);
    NR_PRAGMA_UNROLL
    for (int k = 0; k < 8; ++k) dfr[k] += dx[0][k];
    // rows: 0 d out (16), 16 h2 (32), 48 d h2 (32), 80 h1 (32), 112 d h1 (32), 144 f_ray (32)
    __syncthreads();
    NR_PRAGMA_UNROLL
    for (int j = 0; j < NOUT; ++j) st_one(S, kB2PStride, j, dout[j], c, g);
    st_nat<8>(S, kB2PStride, 16, h2[0], c, g); st_nat<8>(S, kB2PStride, 48, dh2[0], c, g);
    st_nat<8>(S, kB2PStride, 80, h1[0], c, g); st_nat<8>(S, kB2PStride, 112, dh1[0], c, g);
    st_gat(S, kB2PStride, 144, fray[0], c, g);
    __syncthreads();
    sh_tiles<D4>(acc, bacc, S, 0, 16, lane);
    sh_tiles<D2>(acc, bacc, S, 48, 80, lane);
    sh_tiles<D0>(acc, bacc, S, 112, 144, lane);
}

template <bool HAS_VIS>
__global__ void __launch_bounds__(64) self_hit_backward2_kernel(SelfHitBwd2Params p) {
    __shared__ __attribute__((aligned(16))) float S[kShStageRows * kB2PStride];
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
    const nr_wbuf W = nr_make_wbuf(p.weights, sizeof(float) * kPackedPassFloats);
    const nr_wbuf WT = nr_make_wbuf(p.weights_t, sizeof(float) * kPackedTFloats);
    const float nearp = p.que_const[24], farp = p.que_const[25];
    const bool use_vis = HAS_VIS && p.use_vis != 0;
    const int dn = p.dn;
    v4f acc[kShAcc];
    float bacc[kShBias];
    NR_PRAGMA_UNROLL
    for (int i = 0; i < kShAcc; ++i) { acc[i][0] = 0.0f; acc[i][1] = 0.0f; acc[i][2] = 0.0f; acc[i][3] = 0.0f; }
    NR_PRAGMA_UNROLL
    for (int i = 0; i < kShBias; ++i) bacc[i] = 0.0f;
    const int ntiles = (p.rn + 15) / 16;
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        const int glane = lane + nr_opaque_zero();
        int ray = tile * 16 + c;
        const bool valid = ray < p.rn;
        ray = valid ? ray : p.rn - 1;
        float fray[1][8];
        {
            const float4 f0 = ld4(p.feats + (size_t)ray * 32 + 8 * g), f1 = ld4(p.feats + (size_t)ray * 32 + 8 * g + 4);
            fray[0][0] = f0.x; fray[0][1] = f0.y; fray[0][2] = f0.z; fray[0][3] = f0.w;
            fray[0][4] = f1.x; fray[0][5] = f1.y; fray[0][6] = f1.z; fray[0][7] = f1.w;
        }
        float mu0, mu1, s0, s1, aw, nu;
        b2_dist_fwd<HAS_VIS>(W, glane, fray, p.var_bias, mu0, mu1, s0, s1, aw, nu);
        const float nuu = use_vis ? nu : 1.0f;
        // probability backward (is_ref = False interval rule, dist_decoder.py:109-127): lane group g takes samples g, g + 4, ...
        float dmu0 = 0.0f, dmu1 = 0.0f, dsd0 = 0.0f, dsd1 = 0.0f, daw = 0.0f, dnu = 0.0f;
        const float* drow = p.depth + (size_t)ray * dn;
        for (int smp = g; smp < dn; smp += 4) {
            const float t_c = norm_inv_depth(fmaxf(drow[smp], 1e-5f), nearp, farp);
            float lo, hi;
            if (smp == 0) lo = t_c - (norm_inv_depth(drow[1], nearp, farp) - norm_inv_depth(drow[0], nearp, farp)) / 2.0f;
            else lo = (norm_inv_depth(fmaxf(drow[smp - 1], 1e-5f), nearp, farp) + t_c) / 2.0f;
            if (smp == dn - 1) hi = t_c + 500000.0f;
            else hi = (t_c + norm_inv_depth(fmaxf(drow[smp + 1], 1e-5f), nearp, farp)) / 2.0f;
            const float dh = valid ? p.d_hit[(size_t)ray * dn + smp] : 0.0f;
            b2_prob_bwd(lo, hi, mu0, mu1, s0, s1, aw, nuu, use_vis, 0.0f, dh, dmu0, dmu1, dsd0, dsd1, daw, dnu);
        }
        dmu0 = nr_group_sum(dmu0); dmu1 = nr_group_sum(dmu1); dsd0 = nr_group_sum(dsd0); dsd1 = nr_group_sum(dsd1);
        daw = nr_group_sum(daw); dnu = nr_group_sum(dnu);
        // through the output non-linearities: softplus' = 1 - exp(-softplus), sigmoid' = s (1 - s)
        const float dm[2] = {dmu0 * (1.0f - nr_fast_exp(-mu0)), dmu1 * (1.0f - nr_fast_exp(-mu1))};
        const float dv[2] = {dsd0 * (1.0f - nr_fast_exp(-(s0 - p.var_bias))), dsd1 * (1.0f - nr_fast_exp(-(s1 - p.var_bias)))};
        const float da[1] = {daw * aw * (1.0f - aw)};
        float dfr[8];
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 8; ++k) dfr[k] = 0.0f;
        sh_head_bwd<L_DM1, L_DM2, L_DFIN_M, LT_DM1, LT_DM2, DW_M4, DW_M2, DW_M0, 2>(W, WT, glane, lane, fray, dm, dfr, S, acc, bacc);
        sh_head_bwd<L_DV1, L_DV2, L_DFIN_V, LT_DV1, LT_DV2, DW_V4, DW_V2, DW_V0, 2>(W, WT, glane, lane, fray, dv, dfr, S, acc, bacc);
        sh_head_bwd<L_DA1, L_DA2, L_DFIN_A, LT_DA1, LT_DA2, DW_A4, DW_A2, DW_A0, 1>(W, WT, glane, lane, fray, da, dfr, S, acc, bacc);
        if constexpr (HAS_VIS) {
            const float ds[1] = {dnu * nu * (1.0f - nu)};
            sh_head_bwd<L_DS1, L_DS2, L_DFIN_S, LT_DS1, LT_DS2, DW_S4, DW_S2, DW_S0, 1>(W, WT, glane, lane, fray, ds, dfr, S, acc, bacc);
        }
        if (valid) {
            float* o = p.d_feats + (size_t)ray * 32 + 8 * g;
            *reinterpret_cast<float4*>(o) = make_float4(dfr[0], dfr[1], dfr[2], dfr[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(dfr[4], dfr[5], dfr[6], dfr[7]);
        }
    }
    sh_flush<DW_M4>(acc, bacc, p.d_flat, lane); sh_flush<DW_M2>(acc, bacc, p.d_flat, lane); sh_flush<DW_M0>(acc, bacc, p.d_flat, lane);
    sh_flush<DW_V4>(acc, bacc, p.d_flat, lane); sh_flush<DW_V2>(acc, bacc, p.d_flat, lane); sh_flush<DW_V0>(acc, bacc, p.d_flat, lane);
    sh_flush<DW_A4>(acc, bacc, p.d_flat, lane); sh_flush<DW_A2>(acc, bacc, p.d_flat, lane); sh_flush<DW_A0>(acc, bacc, p.d_flat, lane);
    if constexpr (HAS_VIS) {
        sh_flush<DW_S4>(acc, bacc, p.d_flat, lane); sh_flush<DW_S2>(acc, bacc, p.d_flat, lane); sh_flush<DW_S0>(acc, bacc, p.d_flat, lane);
    }
}

// =====================================================================================================================
// decoder_rows_backward2_kernel: backward of the dist decoder on stand-alone rows (network/dist_decoder.py:99-107 forward /
// predict_mean as the generalisation renderer's depth loss calls them, renderer.py:280-316) on the resident scheme of
// self_hit_backward2_kernel: one wave per tile of 16 rows, the heads in registers on the packed / transposed packs, weight gradients
// contracted over the tile on the MFMA into register accumulators.  d_mean / d_var / d_aw / d_vis are the gradients of the decoder's
// OUTPUTS (after softplus / sigmoid); a head whose gradient pointer is null is skipped (predict_mean: the mean head only).
// (The first version, decoder_rows_backward_kernel: lane = row, activations in a global arena - 0.34 ms per 65 536 rows, 0.67 ms of a
// generalisation step for two calls.)
// =====================================================================================================================
struct RowsBwd2Params {
    const float* feats;        // [n][32]
    const float* weights;      // packed pass weights
    const float* weights_t;    // packed transposed layers
    const float* d_mean; const float* d_var; const float* d_aw; const float* d_vis;      // [n][2], [n][2], [n], [n] or null
    float* d_feats;            // [n][32]
    float* d_flat;             // accumulated (dist decoder tensors only)
    int n;
    float var_bias;
};

template <bool HAS_VIS>
__global__ void __launch_bounds__(64) decoder_rows_backward2_kernel(RowsBwd2Params p) {
    __shared__ __attribute__((aligned(16))) float S[kShStageRows * kB2PStride];
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, c = lane & 15;
    const nr_wbuf W = nr_make_wbuf(p.weights, sizeof(float) * kPackedPassFloats);
    const nr_wbuf WT = nr_make_wbuf(p.weights_t, sizeof(float) * kPackedTFloats);
    v4f acc[kShAcc];
    float bacc[kShBias];
    NR_PRAGMA_UNROLL
    for (int i = 0; i < kShAcc; ++i) { acc[i][0] = 0.0f; acc[i][1] = 0.0f; acc[i][2] = 0.0f; acc[i][3] = 0.0f; }
    NR_PRAGMA_UNROLL
    for (int i = 0; i < kShBias; ++i) bacc[i] = 0.0f;
    const int ntiles = (p.n + 15) / 16;
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        const int glane = lane + nr_opaque_zero();
        int row = tile * 16 + c;
        const bool valid = row < p.n;
        row = valid ? row : p.n - 1;
        const float gsc = valid ? 1.0f : 0.0f;
        float fray[1][8];
        {
            const float4 f0 = ld4(p.feats + (size_t)row * 32 + 8 * g), f1 = ld4(p.feats + (size_t)row * 32 + 8 * g + 4);
            fray[0][0] = f0.x; fray[0][1] = f0.y; fray[0][2] = f0.z; fray[0][3] = f0.w;
            fray[0][4] = f1.x; fray[0][5] = f1.y; fray[0][6] = f1.z; fray[0][7] = f1.w;
        }
        float mu0, mu1, s0, s1, aw, nu;
        b2_dist_fwd<HAS_VIS>(W, glane, fray, p.var_bias, mu0, mu1, s0, s1, aw, nu);
        float dfr[8];
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 8; ++k) dfr[k] = 0.0f;
        // through the output non-linearities: softplus' = 1 - exp(-softplus), sigmoid' = s (1 - s)
        if (p.d_mean) {                                        // (uniform)
            const float dm[2] = {gsc * p.d_mean[2 * row] * (1.0f - nr_fast_exp(-mu0)), gsc * p.d_mean[2 * row + 1] * (1.0f - nr_fast_exp(-mu1))};
            sh_head_bwd<L_DM1, L_DM2, L_DFIN_M, LT_DM1, LT_DM2, DW_M4, DW_M2, DW_M0, 2>(W, WT, glane, lane, fray, dm, dfr, S, acc, bacc);
        }
        if (p.d_var) {
            const float dv[2] = {gsc * p.d_var[2 * row] * (1.0f - nr_fast_exp(-(s0 - p.var_bias))),
                                 gsc * p.d_var[2 * row + 1] * (1.0f - nr_fast_exp(-(s1 - p.var_bias)))};
            sh_head_bwd<L_DV1, L_DV2, L_DFIN_V, LT_DV1, LT_DV2, DW_V4, DW_V2, DW_V0, 2>(W, WT, glane, lane, fray, dv, dfr, S, acc, bacc);
        }
        if (p.d_aw) {
            const float da[1] = {gsc * p.d_aw[row] * aw * (1.0f - aw)};
            sh_head_bwd<L_DA1, L_DA2, L_DFIN_A, LT_DA1, LT_DA2, DW_A4, DW_A2, DW_A0, 1>(W, WT, glane, lane, fray, da, dfr, S, acc, bacc);
        }
        if constexpr (HAS_VIS) {
            if (p.d_vis) {
                const float ds[1] = {gsc * p.d_vis[row] * nu * (1.0f - nu)};
                sh_head_bwd<L_DS1, L_DS2, L_DFIN_S, LT_DS1, LT_DS2, DW_S4, DW_S2, DW_S0, 1>(W, WT, glane, lane, fray, ds, dfr, S, acc, bacc);
            }
        }
        if (valid) {
            float* o = p.d_feats + (size_t)row * 32 + 8 * g;
            *reinterpret_cast<float4*>(o) = make_float4(dfr[0], dfr[1], dfr[2], dfr[3]);
            *reinterpret_cast<float4*>(o + 4) = make_float4(dfr[4], dfr[5], dfr[6], dfr[7]);
        }
    }
    // (only the heads that received a gradient: the flush is one atomicAdd per weight and workgroup)
    if (p.d_mean) { sh_flush<DW_M4>(acc, bacc, p.d_flat, lane); sh_flush<DW_M2>(acc, bacc, p.d_flat, lane); sh_flush<DW_M0>(acc, bacc, p.d_flat, lane); }
    if (p.d_var) { sh_flush<DW_V4>(acc, bacc, p.d_flat, lane); sh_flush<DW_V2>(acc, bacc, p.d_flat, lane); sh_flush<DW_V0>(acc, bacc, p.d_flat, lane); }
    if (p.d_aw) { sh_flush<DW_A4>(acc, bacc, p.d_flat, lane); sh_flush<DW_A2>(acc, bacc, p.d_flat, lane); sh_flush<DW_A0>(acc, bacc, p.d_flat, lane); }
    if constexpr (HAS_VIS) {
        if (p.d_vis) { sh_flush<DW_S4>(acc, bacc, p.d_flat, lane); sh_flush<DW_S2>(acc, bacc, p.d_flat, lane); sh_flush<DW_S0>(acc, bacc, p.d_flat, lane); }
    }
}

}  // namespace nr
