// points_backward3_kernel: points_backward2_kernel re-shaped to 4 waves x 2 reference views per wave at ONE wave per SIMD
// (VERDICT r2 next #2).  Same mathematics, same saved quantities, same packed / transposed weight buffers, same staging
// area and job table as nr_kernels_bwd2.h - what changes is who does what:
//
//   * workgroup = 4 waves x one tile of 16 sample points; wave w owns reference views 2w and 2w + 1 as two SLOTS, exactly
//     the forward kernel's `views_per_wave = 2`: the slots share every weight fragment (8 MFMAs per fragment load instead of
//     4: half the L1 / L2 fragment traffic per tile) and give the matrix pipe two independent accumulator chains, which is
//     what covers the fragment round trips now that no second wave shares the SIMD.
//   * 256 threads per workgroup = one wave per SIMD: a wave may use all 512 registers of its SIMD lane slice (256
//     architectural + 256 accumulation registers).  The persistent weight-gradient accumulators (42 tiles + 13 bias sums per
//     wave) are MFMA accumulators and can live in AGPRs; what still does not fit the architectural file spills to AGPRs
//     (v_accvgpr_write / read, no memory) instead of scratch.  The 8-wave kernel ran 2 waves per SIMD with 256 registers
//     each, 0 AGPRs, and spilled 270-315 registers to scratch memory (0.45 GB per dispatch, every reload a ~1 k-cycle
//     exposed round trip).
//   * barriers are among 4 waves; the per-point owner sections (geometry_fc^T tiles, base_fc.0's statistics columns)
//     are spread over all four waves.
#pragma once
#include "nr_kernels_bwd2.h"

namespace nr {

constexpr int kB3Waves = 4, kB3Slots = 2;
constexpr int kB3Red = (kB3Waves + 1) * kB2Rmax * 64;
inline size_t point_bwd3_smem_bytes() { return sizeof(float) * (size_t)(kB3Red + kB2Xch + kB2Stash + kB2Hx + kB2Stage); }
constexpr int kB3Acc = dw_acc_n(kB3Waves), kB3BiasAcc = dw_bias_n(kB3Waves);

template <int R>
__device__ __forceinline__ void b3_allsum(float (&v)[R], float* red, int wave, int lane) {
    block_allreduce<R, kB2Rmax, RED_SUM>(v, red, wave, kB3Waves, lane);
}

// backward of one dist head for the wave's two slots (b2_dist_head_bwd with NS = 2)
template <int L1, int L2, int LF, int T1, int T2, int D4, int D2, int D0, int NOUT>
__device__ __forceinline__ void b3_dist_head_bwd(nr_wbuf W, nr_wbuf WT, int wlane, int lane, int wave, const int (&col)[kB3Slots], int g,
                                                 const float (&fray)[kB3Slots][8], const float (&dout)[kB3Slots][NOUT],
                                                 float (&dfr)[kB3Slots][8], float* S, v4f (&acc)[kB3Acc], float (&bacc)[kB3BiasAcc]) {
    constexpr int NS = kB3Slots;
    float none[NS][1], h1[NS][8], h2[NS][8], dh2[NS][8], dh1[NS][8], dx[NS][8];
    NR_PRAGMA_UNROLL
    for (int s = 0; s < NS; ++s) none[s][0] = 0.0f;
    layer_fwd<L1, NS, ACT_ELU>(W, wlane, fray, none, h1);
    layer_fwd<L2, NS, ACT_ELU>(W, wlane, h1, none, h2);
    VecPre<LF> pf;
    layer_prefetch<LF>(W, wlane, pf);
    NR_PRAGMA_UNROLL
    for (int s = 0; s < NS; ++s) {
        static_assert(kVec[LF].n == NOUT, "head output width");
        vec_bwd<LF>(pf, dout[s], dh2[s]);
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 8; ++k) dh2[s][k] *= delu_s(h2[s][k]);
    }
    layer_fwd<T2, NS, ACT_NONE>(WT, wlane, dh2, none, dh1);
    NR_PRAGMA_UNROLL
    for (int s = 0; s < NS; ++s)
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 8; ++k) dh1[s][k] *= delu_s(h1[s][k]);
    layer_fwd<T1, NS, ACT_NONE>(WT, wlane, dh1, none, dx);
    NR_PRAGMA_UNROLL
    for (int s = 0; s < NS; ++s)
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 8; ++k) dfr[s][k] += dx[s][k];
    // rows: 0 d out (16), 16 h2 (32), 48 d h2 (32), 80 h1 (32), 112 d h1 (32), 144 f_ray (32)
    __syncthreads();
    NR_PRAGMA_UNROLL
    for (int s = 0; s < NS; ++s) {
        NR_PRAGMA_UNROLL
        for (int j = 0; j < NOUT; ++j) st_one(S, kB2Stride, j, dout[s][j], col[s], g);
        st_nat<8>(S, kB2Stride, 16, h2[s], col[s], g); st_nat<8>(S, kB2Stride, 48, dh2[s], col[s], g);
        st_nat<8>(S, kB2Stride, 80, h1[s], col[s], g); st_nat<8>(S, kB2Stride, 112, dh1[s], col[s], g);
        st_gat(S, kB2Stride, 144, fray[s], col[s], g);
    }
    __syncthreads();
    dw_jobs<D4, kB3Waves>(acc, bacc, S, 0, 16, wave, lane);
    dw_jobs<D2, kB3Waves>(acc, bacc, S, 48, 80, wave, lane);
    dw_jobs<D0, kB3Waves>(acc, bacc, S, 112, 144, wave, lane);
}

template <int... IDS>
__device__ __forceinline__ void dw3_flush_all(const v4f (&acc)[kB3Acc], const float (&bacc)[kB3BiasAcc], float* d_flat, int wave, int lane) {
    (dw_flush<IDS, kB3Waves>(acc, bacc, d_flat, wave, lane), ...);
}

template <bool HAS_VIS>
__global__ void __launch_bounds__(256, 1) points_backward3_kernel(PointBwd2Params p) {
    NR_DYNAMIC_SMEM(float, smem);
    constexpr int NS = kB3Slots;
    const int lane = threadIdx.x & 63;
    const int wave = NR_UNIFORM((int)(threadIdx.x >> 6));
    const int g = lane >> 4, c = lane & 15;
    float* red = smem;
    float* xch = smem + kB3Red;
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
    bool vok[NS];
    int view[NS], col[NS];
    NR_PRAGMA_UNROLL
    for (int s = 0; s < NS; ++s) {
        const int vraw = wave * NS + s;
        vok[s] = vraw < p.rfn;                               // padding slots (rfn < 8): masked out everywhere
        view[s] = vok[s] ? vraw : p.rfn - 1;
        col[s] = vraw * 16 + c;                              // this lane's column of the staging area
    }

    v4f acc[kB3Acc];
    float bacc[kB3BiasAcc];
    NR_PRAGMA_UNROLL
    for (int i = 0; i < kB3Acc; ++i) { acc[i][0] = 0.0f; acc[i][1] = 0.0f; acc[i][2] = 0.0f; acc[i][3] = 0.0f; }
    NR_PRAGMA_UNROLL
    for (int i = 0; i < kB3BiasAcc; ++i) bacc[i] = 0.0f;
    float none[NS][1], none1[1][1] = {{0.0f}};
    NR_PRAGMA_UNROLL
    for (int s = 0; s < NS; ++s) none[s][0] = 0.0f;

    for (int base = (int)blockIdx.x * 16; base < npts; base += (int)gridDim.x * 16) {
        const int glane = lane + nr_opaque_zero();
        // the forward's saved tile data: base_fc.0's per-point part + the four statistics -> xch | stash, geometry rows -> xg (LDS-DMA)
        const float* svt = p.saved + (size_t)(base / 16) * kSavedTileFloats;
        {
            const nr_wbuf SV = nr_make_wbuf(svt, sizeof(float) * kSavedTileFloats);
            for (int i = wave; i < (kB2Xch + kB2Stash) / 256; i += kB3Waves) nr_dma16(SV, xch + i * 256, lane, lane * 16, i * 1024);
            for (int i = wave; i < 10; i += kB3Waves) nr_dma16(SV, xg + i * 256, lane, lane * 16, (kSavedGeoRow * 64 + i * 256) * 4);
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
        float mask[NS], dlt[NS][4], tref[NS], fray[NS][8], fimg[NS][8], rgb[NS][3];
        Taps tfs[NS];
        NR_PRAGMA_UNROLL
        for (int s = 0; s < NS; ++s) {
            const float* __restrict__ vc = p.view_const + view[s] * kViewConst;
            const Proj pr = project_point<false>(vc, px, py, pz, (float)p.w, (float)p.h);
            mask[s] = vok[s] ? pr.mask : 0.0f;
            dlt[s][0] = pr.dirx - r.qx; dlt[s][1] = pr.diry - r.qy; dlt[s][2] = pr.dirz - r.qz;
            dlt[s][3] = dot3(pr.dirx, pr.diry, pr.dirz, r.qx, r.qy, r.qz);
            tref[s] = norm_inv_depth_fast(fmaxf(pr.z, 1e-5f), vc[15], vc[16], vc[17]);
            tfs[s] = make_taps_fast(pr.u, pr.v, w_m1, h_m1, inv_w_m1, inv_h_m1, p.fw, p.fh, p.fw == p.w && p.fh == p.h);
            const Taps tcs = make_taps_fast(pr.u, pr.v, w_m1, h_m1, inv_w_m1, inv_h_m1, p.w, p.h, true);
            const int soff_f = view[s] * (int)(fmap * sizeof(float)), soff_c = view[s] * (int)(imap * sizeof(float));
            float4 qf[8], qi[8], qcl[4];
            issue8(rf_map, goff, soff_f, tfs[s], qf);
            issue8(if_map, goff, soff_f, tfs[s], qi);
            issue_rgb(rgb_map, soff_c, tcs, qcl);
            NR_PIN();
            blend8(qf, tfs[s], mask[s], fray[s]);
            blend8(qi, tfs[s], mask[s], fimg[s]);
            blend_rgb(qcl, tcs, mask[s], rgb[s]);
        }
        // ================= forward (recomputed; checkpoints stay in registers) =================
        float mu0[NS], mu1[NS], s0[NS], s1[NS], aw[NS], nu[NS], nuu[NS], vis[NS], hit[NS];
        NR_PRAGMA_UNROLL
        for (int s = 0; s < NS; ++s) {
            const float* sd = svt + kSavedDist + view[s] * 128 + c;
            mu0[s] = sd[0]; mu1[s] = sd[16]; s0[s] = sd[32]; s1[s] = sd[48]; aw[s] = sd[64]; nu[s] = sd[80];
            nuu[s] = use_vis ? nu[s] : 1.0f;
            float v_, h_;
            logistic_prob(tref[s], lo, hi, mu0[s], mu1[s], s0[s], s1[s], aw[s], nu[s], use_vis, v_, h_);
            vis[s] = v_ * mask[s]; hit[s] = h_ * mask[s];
        }
        // prob_embed -> e
        float e[NS][8];
        {
            float x1[NS][1], h[NS][8];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) x1[s][0] = sel4(g, (hit[s] - 0.5f) * 2.0f, (vis[s] - 0.5f) * 2.0f, 0.0f, 0.0f);
            layer_fwd<L_PE1, NS, ACT_RELU>(W, glane, fray, x1, h);
            layer_fwd<L_PE2, NS, ACT_NONE>(W, glane, h, none, e);
        }
        // ray_dir_fc -> gi (img part, gathered order), gr (rgb part)
        float gi[NS][8], gr[NS][3];
        {
            float x1[NS][1], h[NS][4], df[NS][8], dc[NS][3];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) x1[s][0] = sel4(g, dlt[s][0], dlt[s][1], dlt[s][2], dlt[s][3]);
            layer_fwd<L_RD1, NS, ACT_ELU>(W, glane, none, x1, h);
            layer_fwd<L_RD2, NS, ACT_ELU>(W, glane, h, none, df);
            layer_vec<L_RD2, NS>(W, glane, h, dc);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) gi[s][k] = fimg[s][k] + df[s][k];
                NR_PRAGMA_UNROLL
                for (int j = 0; j < 3; ++j) gr[s][j] = rgb[s][j] + elu(dc[s][j]);
            }
        }
        // neuray_fc -> sn
        float sn[NS];
        {
            float h[NS][4], o[NS][1];
            layer_fwd<L_NF1, NS, ACT_ELU>(W, glane, e, none, h);
            layer_vec<L_NF2, NS>(W, glane, h, o);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) sn[s] = sigmoidf(o[s][0]);
        }
        // cross-view weights (ibrnet.py:334-340); the statistics and base_fc.0's per-point part are the forward's (stash, xch)
        float wv[NS], w0[NS], sa0, sa1;
        {
            const float msum = svt[kSavedMsumRow * 64 + lane];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) { wv[s] = mask[s] / (msum + 1e-8f); w0[s] = sn[s] * wv[s]; }
            sa0 = svt[kSavedSw0Row * 64 + lane];
            sa1 = msum / (msum + 1e-8f);
            __syncthreads();                                   // every wave's DMA pieces have landed (the fence waits for them)
        }
        // base_fc -> x;  h64 is recomputed in the backward from xch
        float x[NS][8];
        auto base_hidden = [&](float (&h64)[NS][16]) {
            float xq[NS][16], x1[NS][1];
            v4f a0[NS][2], a1[NS][2];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) { xq[s][k] = gi[s][k]; xq[s][8 + k] = e[s][k]; }
                x1[s][0] = sel4(g, gr[s][0], gr[s][1], gr[s][2], 0.0f);
            }
            NR_PRAGMA_UNROLL
            for (int mo = 0; mo < 2; ++mo)
                NR_PRAGMA_UNROLL
                for (int r_ = 0; r_ < 4; ++r_) {
                    const float b0 = xch[(mo * 4 + r_) * 64 + lane], b1 = xch[((2 + mo) * 4 + r_) * 64 + lane];
                    NR_PRAGMA_UNROLL
                    for (int s = 0; s < NS; ++s) { a0[s][mo][r_] = b0; a1[s][mo][r_] = b1; }
                }
            LayerPre<L_BV0> p0; LayerPre<L_BV1> p1; NoLayer last;
            layer_prefetch<L_BV0>(W, glane, p0);
            layer_acc<L_BV0, NS>(W, glane, p0, xq, x1, a0, last);
            layer_prefetch<L_BV1>(W, glane, p1);
            layer_acc<L_BV1, NS>(W, glane, p1, xq, x1, a1, last);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s)
                NR_PRAGMA_UNROLL
                for (int mo = 0; mo < 2; ++mo)
                    NR_PRAGMA_UNROLL
                    for (int r_ = 0; r_ < 4; ++r_) { h64[s][4 * mo + r_] = elu_s(a0[s][mo][r_]); h64[s][8 + 4 * mo + r_] = elu_s(a1[s][mo][r_]); }
        };
        {
            float h64[NS][16];
            base_hidden(h64);
            layer_fwd<L_B2, NS, ACT_ELU>(W, glane, h64, none, x);
        }
        // vis_fc -> x2, visp;  vis_fc2 -> vis2;  rgb_fc -> z
        float x2[NS][8], visp[NS], yv32[NS], vis2[NS], v2sig[NS], z[NS];
        {
            float xin[NS][8], h[NS][8], y[NS][8], yv[NS][1], o[NS][1];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s)
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) xin[s][k] = x[s][k] * wv[s];
            layer_fwd<L_VF1, NS, ACT_ELU>(W, glane, xin, none, h);
            layer_fwd<L_VF2, NS, ACT_ELU>(W, glane, h, none, y);
            layer_vec<L_VF2, NS>(W, glane, h, yv);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                yv32[s] = elu(yv[s][0]);
                visp[s] = sigmoidf(yv32[s]) * mask[s];
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) { x2[s][k] = x[s][k] + y[s][k]; xin[s][k] = x2[s][k] * visp[s]; }
            }
            layer_fwd<L_V21, NS, ACT_ELU>(W, glane, xin, none, h);
            layer_vec<L_V22, NS>(W, glane, h, o);
            float x1[NS][2], h16[NS][4], h8[NS][4];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                v2sig[s] = sigmoidf(o[s][0]);
                vis2[s] = v2sig[s] * mask[s];
                x1[s][0] = sel4(g, vis2[s], dlt[s][0], dlt[s][1], dlt[s][2]);
                x1[s][1] = sel4(g, dlt[s][3], 0.0f, 0.0f, 0.0f);
            }
            layer_fwd<L_RF1, NS, ACT_ELU>(W, glane, x2, x1, h16);
            layer_fwd<L_RF2, NS, ACT_ELU>(W, glane, h16, none, h8);
            layer_vec<L_RF3, NS>(W, glane, h8, o);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) z[s] = mask[s] > 0.0f ? o[s][0] : -1e9f;
        }
        // softmax blend weights (ibrnet.py:350-354,366-367) from the forward's max z / sum exp / sum vis''
        float beta[NS], wh[NS], svis, swh;
        const float* gm_l = xg + 16 * 64 + lane;              // weighted mean row k: gm_l[k * 64], variance: gm_l[(8 + k) * 64]
        {
            svis = xg[38 * 64 + lane];
            swh = svis / (svis + 1e-8f);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                const float ez = vok[s] ? nr_fast_exp(z[s] - xg[36 * 64 + lane]) : 0.0f;     // (padding slots take no part in the softmax)
                beta[s] = ez / xg[37 * 64 + lane];
                wh[s] = vis2[s] / (svis + 1e-8f);
            }
        }
        const float meanw = swh * inv_rfn;

        // ================= backward =================
        const float* up = p.d_point_rec + (size_t)pi * kPointRec;
        const float gsc = pvalid ? 1.0f : 0.0f;
        // ---- geometry_fc (per point; ibrnet.py:353-354): every wave redoes the small transposed geometry_fc.2 and takes output
        // tile `wave` of geometry_fc.0^T
        float dgm[8], dgv[8], dmeanw;
        {
            // per-point staging (stride kB2PStride): rows 0 d Gpre (16), 16 h64 (64), 80 d h64 (64), 144 input (65 -> 80)
            float* SP = S;
            {
                float h[1][16], dG[1][4], dh[1][16];
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) h[0][k] = xg[k * 64 + lane];
                const float4 u4 = ld4(up + 4 * g);
                dG[0][0] = u4.x * gsc * delu(xg[32 * 64 + lane]); dG[0][1] = u4.y * gsc * delu(xg[33 * 64 + lane]);
                dG[0][2] = u4.z * gsc * delu(xg[34 * 64 + lane]); dG[0][3] = u4.w * gsc * delu(xg[35 * 64 + lane]);
                layer_fwd<LT_GF2, 1, ACT_NONE>(WT, glane, dG, none1, dh);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) dh[0][k] *= delu_s(h[0][k]);
                v4f a4[1];
                a4[0][0] = 0.0f; a4[0][1] = 0.0f; a4[0][2] = 0.0f; a4[0][3] = 0.0f;
                layer_tile<LT_GF1, 1>(WT, glane, wave, dh, none1, a4);
                // hand d mean / d var / d mean weight to every wave (hx row 4 tile + r: d mean rows 0..7, d var rows 8..15)
                NR_PRAGMA_UNROLL
                for (int r_ = 0; r_ < 4; ++r_) hx[(wave * 4 + r_) * 64 + lane] = a4[0][r_];
                if (wave == 0) {
                    float dmw[1][1];
                    layer_vec<LT_GF1, 1>(WT, glane, dh, dmw);
                    hx[16 * 64 + lane] = dmw[0][0];
                    st_nat<4>(SP, kB2PStride, 0, dG[0], c, g);
                }
                if (wave == 3) st_nat<16>(SP, kB2PStride, 16, h[0], c, g);
                if (wave == 1) st_nat<16>(SP, kB2PStride, 80, dh[0], c, g);
                if (wave == 2) {
                    float gmv[16];
                    NR_PRAGMA_UNROLL
                    for (int k = 0; k < 16; ++k) gmv[k] = gm_l[k * 64];
                    st_nat<16>(SP, kB2PStride, 144, gmv, c, g);           // rows 144..175 mean, 176..207 variance
                    st_one(SP, kB2PStride, 208, meanw, c, g);
                }
            }
            __syncthreads();
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 8; ++k) { dgm[k] = hx[k * 64 + lane]; dgv[k] = hx[(8 + k) * 64 + lane]; }
            dmeanw = hx[16 * 64 + lane];
            dw_jobs<DW_GF2, kB3Waves>(acc, bacc, SP, 0, 16, wave, lane);
            dw_jobs<DW_GF0, kB3Waves>(acc, bacc, SP, 80, 144, wave, lane);
        }
        // ---- visibility-weighted mean / variance + softmax blend -> d x2, d vis2, d z
        float dx2[NS][8], dvis2[NS], dz[NS];
        {
            float dwh[NS], dbeta[NS], s2[2] = {0.0f, 0.0f};
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                float part = 0.0f;
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) {
                    const float xv = x2[s][k], mean = gm_l[k * 64];
                    const float dmt = dgm[k] - 2.0f * dgv[k] * mean * (1.0f - swh);
                    dx2[s][k] = wh[s] * (dmt + 2.0f * (xv - mean) * dgv[k]);
                    part += dmt * xv + dgv[k] * (xv - mean) * (xv - mean);
                }
                dwh[s] = dmeanw * inv_rfn + nr_group_sum(part);            // the 32 features sit in 8 registers x 4 lane groups
                dbeta[s] = (up[16] * rgb[s][0] + up[17] * rgb[s][1] + up[18] * rgb[s][2]) * gsc;
                s2[0] += dwh[s] * wh[s]; s2[1] += beta[s] * dbeta[s];
            }
            b3_allsum<2>(s2, red, wave, lane);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                dvis2[s] = (dwh[s] - s2[0]) / (svis + 1e-8f);
                dz[s] = beta[s] * (dbeta[s] - s2[1]);
                if (!(mask[s] > 0.0f)) dz[s] = 0.0f;
            }
        }
        // ---- rgb_fc backward (ibrnet.py:363-365)
        {
            float x1[NS][2], h16[NS][4], h8[NS][4], d8[NS][4], d16[NS][4], dxa[NS][8], o1[NS][1];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                x1[s][0] = sel4(g, vis2[s], dlt[s][0], dlt[s][1], dlt[s][2]);
                x1[s][1] = sel4(g, dlt[s][3], 0.0f, 0.0f, 0.0f);
            }
            layer_fwd<L_RF1, NS, ACT_ELU>(W, glane, x2, x1, h16);
            layer_fwd<L_RF2, NS, ACT_ELU>(W, glane, h16, none, h8);
            VecPre<L_RF3> p3;
            layer_prefetch<L_RF3>(W, glane, p3);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                float dy1[1] = {dz[s]};
                vec_bwd<L_RF3>(p3, dy1, d8[s]);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 4; ++k) d8[s][k] *= delu_s(h8[s][k]);
            }
            layer_fwd<LT_RF2, NS, ACT_NONE>(WT, glane, d8, none, d16);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s)
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 4; ++k) d16[s][k] *= delu_s(h16[s][k]);
            layer_fwd<LT_RF1, NS, ACT_NONE>(WT, glane, d16, none, dxa);
            layer_vec<LT_RF1, NS>(WT, glane, d16, o1);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) dx2[s][k] += dxa[s][k];
                dvis2[s] += o1[s][0];
            }
            // rows: 0 dz (16), 16 h8 (16), 32 d8 (16), 48 h16 (16), 64 d16 (16), 80 [x2 32, vis2, dl 4] (48)
            __syncthreads();
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                st_one(S, kB2Stride, 0, dz[s], col[s], g);
                st_nat<4>(S, kB2Stride, 16, h8[s], col[s], g); st_nat<4>(S, kB2Stride, 32, d8[s], col[s], g);
                st_nat<4>(S, kB2Stride, 48, h16[s], col[s], g); st_nat<4>(S, kB2Stride, 64, d16[s], col[s], g);
                st_nat<8>(S, kB2Stride, 80, x2[s], col[s], g);
                st_one(S, kB2Stride, 112, vis2[s], col[s], g);
                NR_PRAGMA_UNROLL
                for (int j = 0; j < 4; ++j) st_one(S, kB2Stride, 113 + j, dlt[s][j], col[s], g);
            }
            __syncthreads();
            dw_jobs<DW_RF4, kB3Waves>(acc, bacc, S, 0, 16, wave, lane);
            dw_jobs<DW_RF2, kB3Waves>(acc, bacc, S, 32, 48, wave, lane);
            dw_jobs<DW_RF0, kB3Waves>(acc, bacc, S, 64, 80, wave, lane);
        }
        // ---- vis_fc2 backward (ibrnet.py:347-348): vis2 = sigmoid(a) * mask
        float dvisp[NS];
        {
            float xin[NS][8], h[NS][8], dh[NS][8], dxin[NS][8], da[NS];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s)
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) xin[s][k] = x2[s][k] * visp[s];
            layer_fwd<L_V21, NS, ACT_ELU>(W, glane, xin, none, h);
            VecPre<L_V22> pv;
            layer_prefetch<L_V22>(W, glane, pv);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                da[s] = dvis2[s] * mask[s] * v2sig[s] * (1.0f - v2sig[s]);
                float dy1[1] = {da[s]};
                vec_bwd<L_V22>(pv, dy1, dh[s]);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) dh[s][k] *= delu_s(h[s][k]);
            }
            layer_fwd<LT_V21, NS, ACT_NONE>(WT, glane, dh, none, dxin);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                float dv = 0.0f;
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) { dv = fmaf(dxin[s][k], x2[s][k], dv); dx2[s][k] = fmaf(dxin[s][k], visp[s], dx2[s][k]); }
                dvisp[s] = nr_group_sum(dv);
            }
            // rows: 0 da (16), 16 h (32), 48 dh (32), 80 xin (32)
            __syncthreads();
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                st_one(S, kB2Stride, 0, da[s], col[s], g);
                st_nat<8>(S, kB2Stride, 16, h[s], col[s], g); st_nat<8>(S, kB2Stride, 48, dh[s], col[s], g); st_nat<8>(S, kB2Stride, 80, xin[s], col[s], g);
            }
            __syncthreads();
            dw_jobs<DW_V22, kB3Waves>(acc, bacc, S, 0, 16, wave, lane);
            dw_jobs<DW_V20, kB3Waves>(acc, bacc, S, 48, 80, wave, lane);
        }
        // ---- vis_fc backward (ibrnet.py:343-346): x2 = x + y, visp = sigmoid(ELU(y32)) * mask; dx2 becomes d x
        float dx[NS][8];
        {
            float xin[NS][8], h[NS][8], y[NS][8], dy[NS][8], dh[NS][8], dxin[NS][8], x1[NS][1], dy32[NS];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s)
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) xin[s][k] = x[s][k] * wv[s];
            layer_fwd<L_VF1, NS, ACT_ELU>(W, glane, xin, none, h);
            layer_fwd<L_VF2, NS, ACT_ELU>(W, glane, h, none, y);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                const float sg = sigmoidf(yv32[s]);
                dy32[s] = dvisp[s] * mask[s] * sg * (1.0f - sg) * delu(yv32[s]);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) dy[s][k] = dx2[s][k] * delu(y[s][k]);
                x1[s][0] = sel4(g, dy32[s], 0.0f, 0.0f, 0.0f);
            }
            layer_fwd<LT_VF2, NS, ACT_NONE>(WT, glane, dy, x1, dh);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s)
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) dh[s][k] *= delu_s(h[s][k]);
            layer_fwd<LT_VF1, NS, ACT_NONE>(WT, glane, dh, none, dxin);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s)
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) dx[s][k] = fmaf(dxin[s][k], wv[s], dx2[s][k]);
            // rows: 0 dy (33 -> 48), 48 h (32), 80 dh (32), 112 xin (32)
            __syncthreads();
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                st_nat<8>(S, kB2Stride, 0, dy[s], col[s], g); st_one(S, kB2Stride, 32, dy32[s], col[s], g);
                st_nat<8>(S, kB2Stride, 48, h[s], col[s], g); st_nat<8>(S, kB2Stride, 80, dh[s], col[s], g); st_nat<8>(S, kB2Stride, 112, xin[s], col[s], g);
            }
            __syncthreads();
            dw_jobs<DW_VF2, kB3Waves>(acc, bacc, S, 0, 48, wave, lane);
            dw_jobs<DW_VF0, kB3Waves>(acc, bacc, S, 80, 112, wave, lane);
        }
        // ---- base_fc backward (ibrnet.py:342) -> d gi, d gr, d e, d (statistics)
        float dgi[NS][8], dgr[NS][3], de[NS][8];
        {
            float h64[NS][16], dxp[NS][8], dh64[NS][16], dcat[NS][16], drgb[NS][3];
            base_hidden(h64);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s)
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) dxp[s][k] = dx[s][k] * delu(x[s][k]);
            layer_fwd<LT_B2, NS, ACT_NONE>(WT, glane, dxp, none, dh64);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s)
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) dh64[s][k] *= delu_s(h64[s][k]);
            layer_fwd<LT_BV, NS, ACT_NONE>(WT, glane, dh64, none, dcat);
            layer_vec<LT_BV, NS>(WT, glane, dh64, drgb);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) { dgi[s][k] = dcat[s][k]; de[s][k] = dcat[s][8 + k]; }
                NR_PRAGMA_UNROLL
                for (int j = 0; j < 3; ++j) dgr[s][j] = drgb[s][j];
            }
            // round 1 rows: 0 dxp (32), 32 h64 (64)
            __syncthreads();
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) { st_nat<8>(S, kB2Stride, 0, dxp[s], col[s], g); st_nat<16>(S, kB2Stride, 32, h64[s], col[s], g); }
            __syncthreads();
            dw_jobs<DW_B2, kB3Waves>(acc, bacc, S, 0, 32, wave, lane);
            // round 2 rows: 0 dh64 (64), 64 [rgb 3 | img 32 | e 32] (67 -> 80): the natural column order 140..206 of base_fc.0
            __syncthreads();
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                st_nat<16>(S, kB2Stride, 0, dh64[s], col[s], g);
                NR_PRAGMA_UNROLL
                for (int j = 0; j < 3; ++j) st_one(S, kB2Stride, 64 + j, gr[s][j], col[s], g);
                st_gat(S, kB2Stride, 67, gi[s], col[s], g);
                st_nat<8>(S, kB2Stride, 99, e[s], col[s], g);
            }
            __syncthreads();
            dw_jobs<DW_BV, kB3Waves>(acc, bacc, S, 0, 64, wave, lane);
            // per-point part: sum over the views of d h64 (the wave's two slots first), then d statistics = W_gl^T (sum d h64)
            float sd16[1][16];
            {
                float part[kB2Rmax];
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 12; ++k) part[k] = dh64[0][k] + dh64[1][k];
                b3_allsum<kB2Rmax>(part, red, wave, lane);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 12; ++k) sd16[0][k] = part[k];
                float p4[4] = {dh64[0][12] + dh64[1][12], dh64[0][13] + dh64[1][13], dh64[0][14] + dh64[1][14], dh64[0][15] + dh64[1][15]};
                b3_allsum<4>(p4, red, wave, lane);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 4; ++k) sd16[0][12 + k] = p4[k];
            }
            // d statistic `wave` of LT_BG (tiles 2 wave, 2 wave + 1: the two halves of its 32 gathered channels):
            //   register r of lane group g of tile 2j + mo = channel 8g + 4mo + r of statistic j
            {
                v4f a8[2][1];
                NR_PRAGMA_UNROLL
                for (int mo = 0; mo < 2; ++mo) {
                    a8[mo][0][0] = 0.0f; a8[mo][0][1] = 0.0f; a8[mo][0][2] = 0.0f; a8[mo][0][3] = 0.0f;
                    layer_tile<LT_BG, 1>(WT, glane, 2 * wave + mo, sd16, none1, a8[mo]);
                }
                float o3[1][3];
                if (wave == 0) layer_vec<LT_BG_R0, 1>(WT, glane, sd16, o3);
                else if (wave == 1) layer_vec<LT_BG_R1, 1>(WT, glane, sd16, o3);
                else if (wave == 2) layer_vec<LT_BG_R2, 1>(WT, glane, sd16, o3);
                else layer_vec<LT_BG_R3, 1>(WT, glane, sd16, o3);
                // hand-off area hx: [44 rows][64 lanes] in the lane layout of the statistics: row 11 j + k (k < 8: channel
                // 8g + k), row 11 j + 8 + i (rgb i)
                __syncthreads();                               // (every wave is done with the geometry hand-off and the DW_BV jobs)
                NR_PRAGMA_UNROLL
                for (int mo = 0; mo < 2; ++mo)
                    NR_PRAGMA_UNROLL
                    for (int r_ = 0; r_ < 4; ++r_) hx[(11 * wave + 4 * mo + r_) * 64 + lane] = a8[mo][0][r_];
                NR_PRAGMA_UNROLL
                for (int i = 0; i < 3; ++i) hx[(11 * wave + 8 + i) * 64 + lane] = o3[0][i];
                // per-point staging for dW of the statistics columns: rows 0 sum d h64 (64), 64 statistics in the natural column
                // order [mean0 35 | var0 35 | mean1 35 | var1 35]: statistic j -> rows 64 + 35 j + {rgb 0..2, 3 + channel}
                float* SP = S;
                if (wave == 0) st_nat<16>(SP, kB2PStride, 0, sd16[0], c, g);
                {
                    const int jj = wave;
                    NR_PRAGMA_UNROLL
                    for (int k = 0; k < 8; ++k) SP[(64 + 35 * jj + 3 + 8 * g + k) * kB2PStride + c] = stash[(11 * jj + k) * 64 + lane];
                    if (g == 0) {
                        NR_PRAGMA_UNROLL
                        for (int i = 0; i < 3; ++i) SP[(64 + 35 * jj + i) * kB2PStride + c] = stash[(11 * jj + 8 + i) * 64 + lane];
                    }
                }
                __syncthreads();
                dw_jobs<DW_BG, kB3Waves>(acc, bacc, SP, 0, 64, wave, lane);
            }
        }
        // ---- cross-view statistics backward: d statistics (per point, in hx rows 0..43) -> d gi / d gr +=, d sn
        float dsn[NS];
        NR_PRAGMA_UNROLL
        for (int s = 0; s < NS; ++s) {
            float dw0 = 0.0f, dw0r = 0.0f;
            NR_PRAGMA_UNROLL
            for (int q = 0; q < 11; ++q) {
                const float xv = q < 8 ? gi[s][q] : gr[s][q - 8];
                const float m0 = stash[q * 64 + lane], m1 = stash[(22 + q) * 64 + lane];
                const float dm0 = hx[q * 64 + lane], dv0 = hx[(11 + q) * 64 + lane];
                const float dm1 = hx[(22 + q) * 64 + lane], dv1 = hx[(33 + q) * 64 + lane];
                const float dmt0 = dm0 - 2.0f * dv0 * m0 * (1.0f - sa0);
                const float dmt1 = dm1 - 2.0f * dv1 * m1 * (1.0f - sa1);
                const float add = w0[s] * (dmt0 + 2.0f * (xv - m0) * dv0) + wv[s] * (dmt1 + 2.0f * (xv - m1) * dv1);
                const float t0 = dmt0 * xv + dv0 * (xv - m0) * (xv - m0);
                if (q < 8) { dgi[s][q] += add; dw0 += t0; } else { dgr[s][q - 8] += add; dw0r += t0; }
            }
            dsn[s] = (nr_group_sum(dw0) + dw0r) * wv[s];          // img channels: 8 registers x 4 lane groups; rgb replicated
        }
        // ---- neuray_fc backward -> d e +=            (rows: 0 do (16), 16 h8 (16), 32 dh8 (16), 48 e (32))
        // ---- ray_dir_fc backward (weights only)      (rows: 80 dy35 (48), 128 h16 (16), 144 dh16 (16), 160 dl (16))
        {
            float h8[NS][4], dh8[NS][4], dea[NS][8], d_o[NS];
            layer_fwd<L_NF1, NS, ACT_ELU>(W, glane, e, none, h8);
            VecPre<L_NF2> pn;
            layer_prefetch<L_NF2>(W, glane, pn);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                d_o[s] = dsn[s] * sn[s] * (1.0f - sn[s]);
                float dy1[1] = {d_o[s]};
                vec_bwd<L_NF2>(pn, dy1, dh8[s]);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 4; ++k) dh8[s][k] *= delu_s(h8[s][k]);
            }
            layer_fwd<LT_NF1, NS, ACT_NONE>(WT, glane, dh8, none, dea);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s)
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) de[s][k] += dea[s][k];
            float x1[NS][1], h16[NS][4], df[NS][8], dc[NS][3], dyi[NS][8], dyr[NS][3], dh16[NS][4], xr[NS][1];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) x1[s][0] = sel4(g, dlt[s][0], dlt[s][1], dlt[s][2], dlt[s][3]);
            layer_fwd<L_RD1, NS, ACT_ELU>(W, glane, none, x1, h16);
            layer_fwd<L_RD2, NS, ACT_ELU>(W, glane, h16, none, df);
            layer_vec<L_RD2, NS>(W, glane, h16, dc);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) dyi[s][k] = dgi[s][k] * delu(df[s][k]);
                NR_PRAGMA_UNROLL
                for (int j = 0; j < 3; ++j) dyr[s][j] = dgr[s][j] * delu(elu(dc[s][j]));
                xr[s][0] = sel4(g, dyr[s][0], dyr[s][1], dyr[s][2], 0.0f);
            }
            layer_fwd<LT_RD2, NS, ACT_NONE>(WT, glane, dyi, xr, dh16);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s)
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 4; ++k) dh16[s][k] *= delu_s(h16[s][k]);
            __syncthreads();
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                st_one(S, kB2Stride, 0, d_o[s], col[s], g);
                st_nat<4>(S, kB2Stride, 16, h8[s], col[s], g); st_nat<4>(S, kB2Stride, 32, dh8[s], col[s], g); st_nat<8>(S, kB2Stride, 48, e[s], col[s], g);
                NR_PRAGMA_UNROLL
                for (int j = 0; j < 3; ++j) st_one(S, kB2Stride, 80 + j, dyr[s][j], col[s], g);
                st_gat(S, kB2Stride, 83, dyi[s], col[s], g);
                st_nat<4>(S, kB2Stride, 128, h16[s], col[s], g); st_nat<4>(S, kB2Stride, 144, dh16[s], col[s], g);
                NR_PRAGMA_UNROLL
                for (int j = 0; j < 4; ++j) st_one(S, kB2Stride, 160 + j, dlt[s][j], col[s], g);
            }
            __syncthreads();
            dw_jobs<DW_NF2, kB3Waves>(acc, bacc, S, 0, 16, wave, lane);
            dw_jobs<DW_NF0, kB3Waves>(acc, bacc, S, 32, 48, wave, lane);
            dw_jobs<DW_RD2, kB3Waves>(acc, bacc, S, 80, 128, wave, lane);
            dw_jobs<DW_RD0, kB3Waves>(acc, bacc, S, 144, 160, wave, lane);
        }
        // ---- prob_embed backward -> d f_ray, d hit, d vis     (rows: 0 de (32), 32 h (32), 64 dh (32), 96 [f_ray 32, hit', vis'] (48))
        float dfr[NS][8], dhit[NS], dvis[NS];
        {
            float x1[NS][1], h[NS][8], dh[NS][8], dxa[NS][8], o2[NS][2], hp[NS], vp_[NS];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                hp[s] = (hit[s] - 0.5f) * 2.0f; vp_[s] = (vis[s] - 0.5f) * 2.0f;
                x1[s][0] = sel4(g, hp[s], vp_[s], 0.0f, 0.0f);
            }
            layer_fwd<L_PE1, NS, ACT_RELU>(W, glane, fray, x1, h);
            layer_fwd<LT_PE2, NS, ACT_NONE>(WT, glane, de, none, dh);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s)
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) dh[s][k] = h[s][k] > 0.0f ? dh[s][k] : 0.0f;
            layer_fwd<LT_PE1, NS, ACT_NONE>(WT, glane, dh, none, dxa);
            layer_vec<LT_PE1, NS>(WT, glane, dh, o2);
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) dfr[s][k] = dxa[s][k];
                dhit[s] = 2.0f * o2[s][0]; dvis[s] = 2.0f * o2[s][1];
            }
            __syncthreads();
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                st_nat<8>(S, kB2Stride, 0, de[s], col[s], g); st_nat<8>(S, kB2Stride, 32, h[s], col[s], g); st_nat<8>(S, kB2Stride, 64, dh[s], col[s], g);
                st_gat(S, kB2Stride, 96, fray[s], col[s], g);
                st_one(S, kB2Stride, 128, hp[s], col[s], g); st_one(S, kB2Stride, 129, vp_[s], col[s], g);
            }
            __syncthreads();
            dw_jobs<DW_PE2, kB3Waves>(acc, bacc, S, 0, 32, wave, lane);
            dw_jobs<DW_PE0, kB3Waves>(acc, bacc, S, 64, 96, wave, lane);
        }
        // ---- probabilities backward (dist_decoder.py:109-140) and the dist decoder heads
        {
            float dm[NS][2], dv[NS][2], da[NS][1], ds[NS][1];
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                float dmu0 = 0.0f, dmu1 = 0.0f, dsd0 = 0.0f, dsd1 = 0.0f, daw = 0.0f, dnu = 0.0f;
                b2_prob_bwd(tref[s] - lo, tref[s] + hi, mu0[s], mu1[s], s0[s], s1[s], aw[s], nuu[s], use_vis, dvis[s] * mask[s], dhit[s] * mask[s],
                            dmu0, dmu1, dsd0, dsd1, daw, dnu);
                // through the output non-linearities: softplus' = 1 - exp(-softplus), sigmoid' = s (1 - s)
                dm[s][0] = dmu0 * (1.0f - nr_fast_exp(-mu0[s])); dm[s][1] = dmu1 * (1.0f - nr_fast_exp(-mu1[s]));
                dv[s][0] = dsd0 * (1.0f - nr_fast_exp(-(s0[s] - p.var_bias))); dv[s][1] = dsd1 * (1.0f - nr_fast_exp(-(s1[s] - p.var_bias)));
                da[s][0] = daw * aw[s] * (1.0f - aw[s]);
                ds[s][0] = dnu * nu[s] * (1.0f - nu[s]);
            }
            b3_dist_head_bwd<L_DM1, L_DM2, L_DFIN_M, LT_DM1, LT_DM2, DW_M4, DW_M2, DW_M0, 2>(W, WT, glane, lane, wave, col, g, fray, dm, dfr, S, acc, bacc);
            b3_dist_head_bwd<L_DV1, L_DV2, L_DFIN_V, LT_DV1, LT_DV2, DW_V4, DW_V2, DW_V0, 2>(W, WT, glane, lane, wave, col, g, fray, dv, dfr, S, acc, bacc);
            b3_dist_head_bwd<L_DA1, L_DA2, L_DFIN_A, LT_DA1, LT_DA2, DW_A4, DW_A2, DW_A0, 1>(W, WT, glane, lane, wave, col, g, fray, da, dfr, S, acc, bacc);
            if constexpr (HAS_VIS)
                b3_dist_head_bwd<L_DS1, L_DS2, L_DFIN_S, LT_DS1, LT_DS2, DW_S4, DW_S2, DW_S0, 1>(W, WT, glane, lane, wave, col, g, fray, ds, dfr, S, acc, bacc);
        }
        // ---- gathers backward: f_ray = mask * bilinear(ray_feats), f_img = mask * bilinear(img_feats) (render_ops.py:54-70).
        // Each slot's 16 columns go through their own LDS slab [16 points][64 channels: 32 ray | 32 img] so that 32 lanes add the
        // 32 contiguous channels of one texel with one instruction; the two halves of the wave take two taps at a time.
        {
            __syncthreads();                                   // the staging area is free (last weight-gradient jobs are done)
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                float* slab = S + (wave * NS + s) * (16 * 72);
                const float sc = (vok[s] && pvalid) ? mask[s] : 0.0f;
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 8; ++k) { slab[c * 72 + 8 * g + k] = dfr[s][k]; slab[c * 72 + 32 + 8 * g + k] = dgi[s][k]; }
                if (g == 0) {
                    slab[c * 72 + 64] = __int_as_float(tfs[s].o00); slab[c * 72 + 65] = __int_as_float(tfs[s].o10);
                    slab[c * 72 + 66] = __int_as_float(tfs[s].o01); slab[c * 72 + 67] = __int_as_float(tfs[s].o11);
                    slab[c * 72 + 68] = tfs[s].w00 * sc; slab[c * 72 + 69] = tfs[s].w10 * sc; slab[c * 72 + 70] = tfs[s].w01 * sc; slab[c * 72 + 71] = tfs[s].w11 * sc;
                }
            }
            __syncthreads();
            const int ch = lane & 31, half = lane >> 5;
            NR_PRAGMA_UNROLL
            for (int s = 0; s < NS; ++s) {
                const float* slab = S + (wave * NS + s) * (16 * 72);
                const size_t voff = (size_t)view[s] * fmap;
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
            }
            __syncthreads();
        }
    }
    // ================= end of the launch: accumulated weight gradients -> global =================
    dw3_flush_all<DW_RF4, DW_RF2, DW_RF0, DW_V22, DW_V20, DW_VF2, DW_VF0, DW_B2, DW_BV, DW_NF2, DW_NF0, DW_RD2, DW_RD0, DW_PE2, DW_PE0,
                  DW_M4, DW_M2, DW_M0, DW_V4, DW_V2, DW_V0, DW_A4, DW_A2, DW_A0, DW_GF2, DW_GF0, DW_BG>(acc, bacc, p.d_flat, wave, lane);
    if constexpr (HAS_VIS) dw3_flush_all<DW_S4, DW_S2, DW_S0>(acc, bacc, p.d_flat, wave, lane);
}

}  // namespace nr
