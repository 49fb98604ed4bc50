// Backward kernels of the per-ray path (first one: the ray kernel).  Same conventions as nr_kernels.h.
//
// rays_backward_kernel: gradient of rays_kernel (nr_kernels.h) - positional encoding, 4-head self attention over the dn
// samples of a ray, LayerNorm, sigma head, alpha compositing - with respect to the per-point records (geometry feature
// and blended colour) and the attention / sigma-head weights.
//   reference forward: network/ibrnet.py:52-102,356-360; renderer.py:157-166; render_ops.py:72-80 (autograd there).
// One wave per ray, lane = sample, the forward is recomputed (nothing is saved by the forward kernel).
//   hit_i = a_i T_i,  T_i = prod_{j<i} t_j,  t_j = 1 - a_j + 1e-10:
//       da_i = dhit_i T_i - (sum_{k>i} dhit_k hit_k) / t_i
//   attention, per head:  P = softmax_j(q~_i . k_j),  o_i = sum_j P_ij v_j,  D_i = do_i . o_i
//       dS_ij = P_ij (do_i . v_j - D_i),  dq~_i = sum_j dS_ij k_j,  dk_j = sum_i dS_ij q~_i,  dv_j = sum_i P_ij do_i
// Weight gradients: every per-lane contribution is summed over the wave (shuffles), then over the rays of the
// workgroup in LDS, then added to global memory with one atomicAdd per weight and workgroup.
#pragma once
#include "nr_kernels.h"

namespace nr {

struct RayBwdParams {
    const float* point_rec;   // [rn][dn][kPointRec]
    const float* depth;       // [rn][dn]
    const float* pos_enc;     // [dn][16]
    const float* weights;     // packed pass weights (ray part at kPackedPointFloats)
    const float* d_pixel;     // [rn][3]
    const float* d_hit_prob;  // [rn][dn] or null
    const float* d_depth;     // [rn] (gradient of render_depth) or null
    float* d_point_rec;       // [rn][dn][kPointRec]: [0..15] d geometry feature, [16..18] d colour, [19] 0
    float* d_weights;         // [kPackedRayFloats], accumulated (+=)
    const float* att_saved;   // null, or what rays_kernel<SAVE> left: [rn][dn][kRayAttSave] softmax shift, 1 / denominator, attention output
    int rn, dn;
};

constexpr int kRayBwdPerSample = 16 * 4 + 12 + 3;     // K, V, q~, do | shift, den, D | t, alpha, u
constexpr int kRayBwdTranspose = 2 * 64 * 17;         // per wave: two [64][17] buffers of wave_outer_add
// rays per workgroup: 4 (one sample per lane, dn <= 64) or 2 (two samples per lane, dn <= 128: the per-sample LDS state doubles)
inline int ray_bwd_waves(int dn) { return dn <= 64 ? kRayWaves : 2; }
inline size_t ray_bwd_smem_bytes(int dn) {
    return sizeof(float) * (2 * (kPackedRayFloats + 12) + ray_bwd_waves(dn) * ((size_t)dn * kRayBwdPerSample + kRayBwdTranspose));
}

// acc[o * 16 + k] += sum over the wave of a[o] * b[k]   (acc in LDS, shared by the waves of the workgroup).
// A [16 x 64 samples] x [64 samples x 16] contraction: the per-lane vectors are transposed through LDS (tA, tB: [64][17]
// per wave) and multiplied on the fp32 MFMA (16 K-steps of 4 samples) instead of 256 wave-wide shuffle reductions.
__device__ __forceinline__ void wave_outer_add(float* acc, const float (&a)[16], const float (&b)[16], bool act, int lane,
                                               float* tA, float* tB) {
    __syncthreads();
    NR_PRAGMA_UNROLL
    for (int k = 0; k < 16; ++k) { tA[lane * 17 + k] = act ? a[k] : 0.0f; tB[lane * 17 + k] = act ? b[k] : 0.0f; }
    __syncthreads();
    const int m = lane & 15, kk = lane >> 4;
    v4f d; d[0] = 0.0f; d[1] = 0.0f; d[2] = 0.0f; d[3] = 0.0f;
    NR_PRAGMA_UNROLL
    for (int s = 0; s < 16; ++s) d = nr_mfma16(tA[(4 * s + kk) * 17 + m], tB[(4 * s + kk) * 17 + m], d);
    NR_PRAGMA_UNROLL
    for (int r = 0; r < 4; ++r) atomicAdd(acc + (4 * kk + r) * 16 + m, d[r]);      // D: row o = 4 kk + r, column k = m
}
__device__ __forceinline__ void wave_vec_add(float* acc, const float (&a)[16], bool act, int lane) {
    for (int o = 0; o < 16; ++o) {
        const float s = wave_sum(act ? a[o] : 0.0f);
        if (lane == 0) atomicAdd(acc + o, s);
    }
}
// y[k] = sum_o M[o][k] x[o]   (transposed product with the row-major LDS matrix M)
__device__ __forceinline__ void matvec16_t(const float* __restrict__ M, const float (&x)[16], float (&y)[16]) {
    NR_PRAGMA_UNROLL
    for (int k = 0; k < 16; ++k) y[k] = 0.0f;
    NR_PRAGMA_UNROLL
    for (int o = 0; o < 16; ++o)
        NR_PRAGMA_UNROLL
        for (int k4 = 0; k4 < 4; ++k4) {
            const float4 w = ld4(M + o * 16 + 4 * k4);
            y[4 * k4] = fmaf(w.x, x[o], y[4 * k4]); y[4 * k4 + 1] = fmaf(w.y, x[o], y[4 * k4 + 1]);
            y[4 * k4 + 2] = fmaf(w.z, x[o], y[4 * k4 + 2]); y[4 * k4 + 3] = fmaf(w.w, x[o], y[4 * k4 + 3]);
        }
}

// per-sample state that lives from the forward to the backward stages of a ray (one instance per sample a lane owns)
struct RayBwdSample {
    float G[16], kk[16], vv[16], q[16], o[16], mx[4], den[4], yh[16], z[16], pre1[16], h1[16];
    float c[3], nvalid, rstd, alpha, ti, em, hit, T, dhit;
    bool inr, act, qmask, sig_on;
    int i;
};

// NCH samples per lane: NCH = 1 for dn <= 64 (4 rays per workgroup), NCH = 2 for dn <= 128 (2 rays per workgroup)
template <int NCH>
__global__ void __launch_bounds__(256) rays_backward_kernel(RayBwdParams p) {
    NR_DYNAMIC_SMEM(float, smem);
    const int lane = threadIdx.x & 63;
    const int wave = NR_UNIFORM((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    const int dn = p.dn;
    float* RW = smem + nr_opaque_zero();
    float* WA = smem + kPackedRayFloats + 12;          // weight-gradient accumulators of the workgroup
    float* base = smem + 2 * (kPackedRayFloats + 12) + (size_t)wave * (dn * kRayBwdPerSample + kRayBwdTranspose);
    float* ks = base; float* vs = ks + dn * 16; float* qs = vs + dn * 16; float* dos = qs + dn * 16;
    float* st = dos + dn * 16;                         // [dn][12]: softmax shift (4), 1 / denominator (4), D (4)
    float* tr = st + dn * 12; float* al = tr + dn; float* us = al + dn;
    float* tA = us + dn; float* tB = tA + 64 * 17;
    for (int i = threadIdx.x; i < kPackedRayFloats; i += blockDim.x) { RW[i] = p.weights[kPackedPointFloats + i]; WA[i] = 0.0f; }
    __syncthreads();
    const int nray_iter = (p.rn + nwaves - 1) / nwaves;

    for (int it = blockIdx.x; it < nray_iter; it += gridDim.x) {
        int ray = it * nwaves + wave;
        const bool rvalid = ray < p.rn;
        ray = rvalid ? ray : p.rn - 1;
        RayBwdSample sm[NCH];
        asm volatile("" ::: "memory");
        // ---- forward, part 1: G, K, V
        NR_PRAGMA_UNROLL
        for (int ch = 0; ch < NCH; ++ch) {
            RayBwdSample& s = sm[ch];
            const int iraw = ch * 64 + lane;
            s.inr = iraw < dn;
            s.i = s.inr ? iraw : dn - 1;               // lanes past the last sample redo sample dn-1 and contribute nothing
            s.act = s.inr && rvalid;
            const float* rec = p.point_rec + ((size_t)ray * dn + s.i) * kPointRec;
            NR_PRAGMA_UNROLL
            for (int k4 = 0; k4 < 4; ++k4) {
                const float4 a = ld4(rec + 4 * k4), b = ld4(p.pos_enc + s.i * 16 + 4 * k4);
                s.G[4 * k4] = a.x + b.x; s.G[4 * k4 + 1] = a.y + b.y; s.G[4 * k4 + 2] = a.z + b.z; s.G[4 * k4 + 3] = a.w + b.w;
            }
            const float4 c4 = ld4(rec + 16);           // colour (3), number of valid views
            s.c[0] = c4.x; s.c[1] = c4.y; s.c[2] = c4.z; s.nvalid = c4.w;
            matvec16(RW + RW_WK, s.G, s.kk);
            matvec16(RW + RW_WV, s.G, s.vv);
            if (s.inr) {
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) { ks[s.i * 16 + k] = s.kk[k]; vs[s.i * 16 + k] = s.vv[k]; }
            }
        }
        __syncthreads();
        // ---- forward, part 2: attention, LayerNorm, sigma head
        NR_PRAGMA_UNROLL
        for (int ch = 0; ch < NCH; ++ch) {
            RayBwdSample& s = sm[ch];
            matvec16(RW + RW_WQ, s.G, s.q);
            s.qmask = !(s.nvalid > 1.0f);              // quirk A.9.3: the row's scores are all -1e9 <=> q~ = 0
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) s.q[k] = s.qmask ? 0.0f : s.q[k] / 2.0f;
            if (p.att_saved) {                         // the forward's softmax shift, 1 / denominator and attention output
                const float* sv = p.att_saved + ((size_t)ray * dn + s.i) * kRayAttSave;
                const float4 sh = ld4(sv), rd = ld4(sv + 4);
                s.mx[0] = sh.x; s.mx[1] = sh.y; s.mx[2] = sh.z; s.mx[3] = sh.w;
                s.den[0] = rd.x; s.den[1] = rd.y; s.den[2] = rd.z; s.den[3] = rd.w;
                NR_PRAGMA_UNROLL
                for (int k4 = 0; k4 < 4; ++k4) {
                    const float4 o4 = ld4(sv + 8 + 4 * k4);
                    s.o[4 * k4] = o4.x; s.o[4 * k4 + 1] = o4.y; s.o[4 * k4 + 2] = o4.z; s.o[4 * k4 + 3] = o4.w;
                }
            } else {
            NR_PRAGMA_UNROLL
            for (int hh = 0; hh < 4; ++hh) {
                float m_ = -INFINITY;
                for (int j = 0; j < dn; ++j) {
                    const float4 kj = ld4(ks + j * 16 + hh * 4);
                    m_ = fmaxf(m_, fmaf(s.q[hh * 4 + 3], kj.w, fmaf(s.q[hh * 4 + 2], kj.z, fmaf(s.q[hh * 4 + 1], kj.y, s.q[hh * 4] * kj.x))));
                }
                float d_ = 0.0f, a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                for (int j = 0; j < dn; ++j) {
                    const float4 kj = ld4(ks + j * 16 + hh * 4);
                    const float sc = fmaf(s.q[hh * 4 + 3], kj.w, fmaf(s.q[hh * 4 + 2], kj.z, fmaf(s.q[hh * 4 + 1], kj.y, s.q[hh * 4] * kj.x)));
                    const float e_ = nr_fast_exp(sc - m_);
                    const float4 vj = ld4(vs + j * 16 + hh * 4);
                    d_ += e_; a0 = fmaf(e_, vj.x, a0); a1 = fmaf(e_, vj.y, a1); a2 = fmaf(e_, vj.z, a2); a3 = fmaf(e_, vj.w, a3);
                }
                const float rd_ = 1.0f / d_;        // (the key loops below multiply by it: one division per head instead of one per key)
                s.mx[hh] = m_; s.den[hh] = rd_;
                s.o[hh * 4] = a0 * rd_; s.o[hh * 4 + 1] = a1 * rd_; s.o[hh * 4 + 2] = a2 * rd_; s.o[hh * 4 + 3] = a3 * rd_;
            }
            }
            float y[16], mean = 0.0f, var = 0.0f;
            matvec16(RW + RW_FC, s.o, y);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) { y[k] += s.G[k]; mean += y[k]; }
            mean /= 16.0f;
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) { const float d_ = y[k] - mean; var = fmaf(d_, d_, var); }
            var /= 16.0f;
            s.rstd = 1.0f / sqrtf(var + 1e-6f);
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) { s.yh[k] = (y[k] - mean) * s.rstd; s.z[k] = fmaf(s.yh[k], RW[RW_LNW + k], RW[RW_LNB + k]); }
            matvec16(RW + RW_OG0W, s.z, s.pre1);
            float spre = RW[RW_OG2B];
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) {
                s.pre1[k] += RW[RW_OG0B + k];
                s.h1[k] = s.pre1[k] > 0.0f ? s.pre1[k] : expf(s.pre1[k]) - 1.0f;
                spre = fmaf(RW[RW_OG2W + k], s.h1[k], spre);
            }
            s.sig_on = (spre > 0.0f) && !(s.nvalid < 1.0f);
            const float sg = s.sig_on ? spre : 0.0f;
            s.em = expf(-sg);                          // 1 - alpha
            s.alpha = 1.0f - s.em;
            s.ti = (1.0f - s.alpha) + 1e-10f;
            if (s.inr) { tr[s.i] = s.ti; al[s.i] = s.alpha; }
        }
        __syncthreads();
        // ---- compositing forward + its backward
        const float gp0 = p.d_pixel[(size_t)ray * 3], gp1 = p.d_pixel[(size_t)ray * 3 + 1], gp2 = p.d_pixel[(size_t)ray * 3 + 2];
        NR_PRAGMA_UNROLL
        for (int ch = 0; ch < NCH; ++ch) {
            RayBwdSample& s = sm[ch];
            float T = 1.0f;
            for (int j = 0; j < dn; ++j) { const float tj = tr[j]; T = (j < s.i) ? T * tj : T; }
            s.T = T;
            s.hit = s.alpha * T;
            float dhit = gp0 * s.c[0] + gp1 * s.c[1] + gp2 * s.c[2];
            if (p.d_depth) dhit = fmaf(p.d_depth[ray], p.depth[(size_t)ray * dn + s.i], dhit);
            if (p.d_hit_prob) dhit += p.d_hit_prob[(size_t)ray * dn + s.i];
            s.dhit = dhit;
            if (s.inr) us[s.i] = dhit * s.hit;
        }
        __syncthreads();
        float dq[NCH][16];
        float dyk[NCH][16];
        NR_PRAGMA_UNROLL
        for (int ch = 0; ch < NCH; ++ch) {
            RayBwdSample& s = sm[ch];
            float S = 0.0f;
            for (int j = 0; j < dn; ++j) { const float uj = us[j]; S = (j > s.i) ? S + uj : S; }
            const float dalpha = s.dhit * s.T - S / s.ti;
            const float dsg = s.sig_on ? dalpha * s.em : 0.0f;
            // ---- sigma head backward
            float dpre1[16], dz[16], dyh[16], dO[16];
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) dpre1[k] = dsg * RW[RW_OG2W + k] * (s.pre1[k] > 0.0f ? 1.0f : s.h1[k] + 1.0f);
            {
                float t16[16];
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) t16[k] = dsg * s.h1[k];
                wave_vec_add(WA + RW_OG2W, t16, s.act, lane);
                const float sb = wave_sum(s.act ? dsg : 0.0f);
                if (lane == 0) atomicAdd(WA + RW_OG2B, sb);
                wave_vec_add(WA + RW_OG0B, dpre1, s.act, lane);
                wave_outer_add(WA + RW_OG0W, dpre1, s.z, s.act, lane, tA, tB);
            }
            matvec16_t(RW + RW_OG0W, dpre1, dz);
            // ---- LayerNorm backward
            float m1 = 0.0f, m2 = 0.0f;
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) { dyh[k] = dz[k] * RW[RW_LNW + k]; m1 += dyh[k]; m2 = fmaf(dyh[k], s.yh[k], m2); }
            m1 /= 16.0f; m2 /= 16.0f;
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) dyk[ch][k] = s.rstd * (dyh[k] - m1 - s.yh[k] * m2);
            {
                float t16[16];
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) t16[k] = dz[k] * s.yh[k];
                wave_vec_add(WA + RW_LNW, t16, s.act, lane);
                wave_vec_add(WA + RW_LNB, dz, s.act, lane);
                wave_outer_add(WA + RW_FC, dyk[ch], s.o, s.act, lane, tA, tB);
            }
            matvec16_t(RW + RW_FC, dyk[ch], dO);
            // ---- attention backward, query side (this lane = query i)
            float Dh[4];
            NR_PRAGMA_UNROLL
            for (int hh = 0; hh < 4; ++hh) {
                Dh[hh] = dO[hh * 4] * s.o[hh * 4] + dO[hh * 4 + 1] * s.o[hh * 4 + 1] + dO[hh * 4 + 2] * s.o[hh * 4 + 2] + dO[hh * 4 + 3] * s.o[hh * 4 + 3];
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
                for (int j = 0; j < dn; ++j) {
                    const float4 kj = ld4(ks + j * 16 + hh * 4);
                    const float4 vj = ld4(vs + j * 16 + hh * 4);
                    const float sc = fmaf(s.q[hh * 4 + 3], kj.w, fmaf(s.q[hh * 4 + 2], kj.z, fmaf(s.q[hh * 4 + 1], kj.y, s.q[hh * 4] * kj.x)));
                    const float P = nr_fast_exp(sc - s.mx[hh]) * s.den[hh];
                    const float dP = dO[hh * 4] * vj.x + dO[hh * 4 + 1] * vj.y + dO[hh * 4 + 2] * vj.z + dO[hh * 4 + 3] * vj.w;
                    const float dS = P * (dP - Dh[hh]);
                    a0 = fmaf(dS, kj.x, a0); a1 = fmaf(dS, kj.y, a1); a2 = fmaf(dS, kj.z, a2); a3 = fmaf(dS, kj.w, a3);
                }
                // q~ = q / 2 (and q~ = 0, without gradient, on masked rows)
                dq[ch][hh * 4] = s.qmask ? 0.0f : a0 * 0.5f; dq[ch][hh * 4 + 1] = s.qmask ? 0.0f : a1 * 0.5f;
                dq[ch][hh * 4 + 2] = s.qmask ? 0.0f : a2 * 0.5f; dq[ch][hh * 4 + 3] = s.qmask ? 0.0f : a3 * 0.5f;
            }
            if (s.inr) {
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 16; ++k) { qs[s.i * 16 + k] = s.q[k]; dos[s.i * 16 + k] = s.act ? dO[k] : 0.0f; }
                NR_PRAGMA_UNROLL
                for (int hh = 0; hh < 4; ++hh) { st[s.i * 12 + hh] = s.mx[hh]; st[s.i * 12 + 4 + hh] = s.den[hh]; st[s.i * 12 + 8 + hh] = Dh[hh]; }
            }
        }
        __syncthreads();
        // ---- attention backward, key side (this lane = key i): dk_i, dv_i
        NR_PRAGMA_UNROLL
        for (int ch = 0; ch < NCH; ++ch) {
            RayBwdSample& s = sm[ch];
            float dk[16], dv[16];
            NR_PRAGMA_UNROLL
            for (int hh = 0; hh < 4; ++hh) {
                float k0 = 0.0f, k1 = 0.0f, k2 = 0.0f, k3 = 0.0f, v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
                for (int j = 0; j < dn; ++j) {               // j = query
                    const float4 qj = ld4(qs + j * 16 + hh * 4);
                    const float4 dj = ld4(dos + j * 16 + hh * 4);
                    const float sc = fmaf(qj.w, s.kk[hh * 4 + 3], fmaf(qj.z, s.kk[hh * 4 + 2], fmaf(qj.y, s.kk[hh * 4 + 1], qj.x * s.kk[hh * 4])));
                    const float P = nr_fast_exp(sc - st[j * 12 + hh]) * st[j * 12 + 4 + hh];
                    const float dP = dj.x * s.vv[hh * 4] + dj.y * s.vv[hh * 4 + 1] + dj.z * s.vv[hh * 4 + 2] + dj.w * s.vv[hh * 4 + 3];
                    const float dS = P * (dP - st[j * 12 + 8 + hh]);
                    // rows whose ray is invalid carry do = 0 and D = 0: dS = 0, no contribution
                    k0 = fmaf(dS, qj.x, k0); k1 = fmaf(dS, qj.y, k1); k2 = fmaf(dS, qj.z, k2); k3 = fmaf(dS, qj.w, k3);
                    v0 = fmaf(P, dj.x, v0); v1 = fmaf(P, dj.y, v1); v2 = fmaf(P, dj.z, v2); v3 = fmaf(P, dj.w, v3);
                }
                dk[hh * 4] = k0; dk[hh * 4 + 1] = k1; dk[hh * 4 + 2] = k2; dk[hh * 4 + 3] = k3;
                dv[hh * 4] = v0; dv[hh * 4 + 1] = v1; dv[hh * 4 + 2] = v2; dv[hh * 4 + 3] = v3;
            }
            wave_outer_add(WA + RW_WQ, dq[ch], s.G, s.act, lane, tA, tB);
            wave_outer_add(WA + RW_WK, dk, s.G, s.act, lane, tA, tB);
            wave_outer_add(WA + RW_WV, dv, s.G, s.act, lane, tA, tB);
            float gq[16], gk[16], gv[16];
            matvec16_t(RW + RW_WQ, dq[ch], gq);
            matvec16_t(RW + RW_WK, dk, gk);
            matvec16_t(RW + RW_WV, dv, gv);
            if (s.act) {
                float* out = p.d_point_rec + ((size_t)ray * dn + s.i) * kPointRec;
                NR_PRAGMA_UNROLL
                for (int k4 = 0; k4 < 4; ++k4)
                    *reinterpret_cast<float4*>(out + 4 * k4) =
                        make_float4(dyk[ch][4 * k4] + gq[4 * k4] + gk[4 * k4] + gv[4 * k4], dyk[ch][4 * k4 + 1] + gq[4 * k4 + 1] + gk[4 * k4 + 1] + gv[4 * k4 + 1],
                                    dyk[ch][4 * k4 + 2] + gq[4 * k4 + 2] + gk[4 * k4 + 2] + gv[4 * k4 + 2], dyk[ch][4 * k4 + 3] + gq[4 * k4 + 3] + gk[4 * k4 + 3] + gv[4 * k4 + 3]);
                *reinterpret_cast<float4*>(out + 16) = make_float4(s.hit * gp0, s.hit * gp1, s.hit * gp2, 0.0f);
            }
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < kPackedRayFloats; k += blockDim.x) atomicAdd(p.d_weights + k, WA[k]);
}


// -------------------------------------------------------------------------------------------------
// points_backward_kernel: gradient of points_kernel (projection -> gathers -> dist decoder -> probabilities ->
// prob_embed / ray_dir_fc / neuray_fc -> cross-view statistics -> base_fc -> vis_fc -> vis_fc2 -> rgb_fc -> softmax
// blend + visibility-weighted statistics -> geometry_fc) with respect to every weight of the pass and to the ray_feats
// and img_feats maps.  Autograd of dist_decoder.py:53-140, renderer.py:67-83,127-135, aggregate_net.py:34-68,
// ibrnet.py:315-354,361-367 in the reference.
//
// First, correctness-oriented version: one wave per workgroup, lane = (point, view) with the views of a point in VP
// consecutive lanes (VP = power of two >= rfn), so every cross-view reduction is a shuffle.  Activations and
// gradients live as rows of a per-workgroup arena in global memory (row r, lane l at arena[r*64 + l]); dense layers are
// runtime loops over natural-layout weights (nr_layout.h "flat natural layout"); a weight gradient
// dW[o][k] = sum_lanes dY[o][lane] X[k][lane] is computed with one (o,k) per lane and added with one atomicAdd per
// weight and tile.  The forward is recomputed stage by stage.  (Training batches are ~0.5 M (point, view) pairs per
// step: this costs a few tens of ms; the MFMA version is future work.)
// -------------------------------------------------------------------------------------------------
struct PointBwdParams {
    const float* que_const;
    const float* view_const;
    const float* coords;       // [rn][2]
    const float* depth;        // [rn][dn]
    const float* ray_feats;    // [rfn][fh][fw][32]
    const float* img_feats;    // [rfn][fh][fw][32]
    const float* rgba;         // [rfn][h][w][4]
    const float* flat;         // [kFlatPassFloats] natural-layout weights
    const float* d_point_rec;  // [rn*dn][kPointRec]: [0..15] d geometry feature, [16..18] d colour
    float* d_flat;             // [kFlatPassFloats], accumulated
    float* d_ray_feats;        // [rfn][fh][fw][32], accumulated
    float* d_img_feats;        // [rfn][fh][fw][32], accumulated
    float* workspace;          // [gridDim.x][kBwdRows][64]
    int rfn, rn, dn, h, w, fh, fw;
    int vp;                    // lanes per point: power of two >= rfn
    int has_vis_head, use_vis;
    float var_bias;
};

// arena rows
constexpr int BR_FR = 0, BR_FI = 32, BR_RGB = 64, BR_DL = 67;
constexpr int BR_GL = 71, BR_GP = 211, BR_E = 246;                   // base_fc.0 input = [GL(140) GP(35) E(32)] contiguous
constexpr int BR_X = 278, BR_X2 = 310;
constexpr int BR_DGL = 342, BR_DGP = 482, BR_DE = 517;               // gradient of the base_fc.0 input, same order
constexpr int BR_DFR = 549, BR_DX = 581;
constexpr int BR_S0 = 613, BR_S1 = 677, BR_S2 = 741, BR_S3 = 805;    // 64-row scratch areas
constexpr int BR_SC = 869;                                           // 32 rows of per-lane scalars
constexpr int kBwdRows = 901;

enum BwdAct { BA_NONE, BA_ELU, BA_RELU };
__device__ __forceinline__ float bwd_act(float x, int a) {
    if (a == BA_ELU) return x > 0.0f ? x : expf(x) - 1.0f;
    if (a == BA_RELU) return fmaxf(x, 0.0f);
    return x;
}
// derivative of the activation, from its OUTPUT y
__device__ __forceinline__ float bwd_dact(float y, int a) {
    if (a == BA_ELU) return y > 0.0f ? 1.0f : y + 1.0f;
    if (a == BA_RELU) return y > 0.0f ? 1.0f : 0.0f;
    return 1.0f;
}
// C[r][n] = act(init + sum_q A(r, q) Bm[q][n]) over the wave's 64 columns n, with A(r, q) = W[r*rs + q*qs] taken from
// the natural-layout weights: fp32 MFMA 16x16x4 per (16 rows, 16 columns) tile.  init = C (accumulate), the bias, or 0.
// Operand mapping chosen so that every arena access is a 16-byte load / store per lane:
//   column tiles: tile t of lane column m is arena column 4m + t  (one float4 of row q feeds the four tiles);
//   contraction: a batch covers 32 values of q, lane group kk contributes q = b0 + 8 kk + u in MFMA step u.
// D: lane (m, kk), register r of tile t -> C[r0 + 4 kk + r][4 m + t].
// Reads and writes other lanes' columns: barriers on entry and exit.
// (not inlined: one copy with run-time loops; inlined + unrolled at ~90 call sites the kernel grew to 70 k instructions
// with 3,300 spilled registers.  The pointer arguments are cast to the global address space - a non-inlined function
// cannot see where they point and would use FLAT accesses - and the sizes are made wave-uniform, they arrive in VGPRs.
// Tried and measured slower, 15.1 vs 11.1 ms per training step: all row blocks of C in one pass over Bm, 64 accumulator
// registers, so that Bm is read once per call - the re-reads it saves hit in L2 anyway.)
__device__ __noinline__ void bwd_mm(const float* __restrict__ W_, int rs, int qs, int R, int Q, const float* __restrict__ Bm_, float* __restrict__ C_,
                                       const float* __restrict__ bias_, int act, bool accumulate, int lane) {
    __syncthreads();
    NR_GLOBAL_PTR(const float) W = NR_TO_GLOBAL(const float, W_);
    NR_GLOBAL_PTR(const float) Bm = NR_TO_GLOBAL(const float, Bm_);
    NR_GLOBAL_PTR(const float) bias = NR_TO_GLOBAL(const float, bias_);
    NR_GLOBAL_PTR(float) C = NR_TO_GLOBAL(float, C_);
    R = NR_UNIFORM(R); Q = NR_UNIFORM(Q); rs = NR_UNIFORM(rs); qs = NR_UNIFORM(qs); act = NR_UNIFORM(act);
    const int m = lane & 15, kk = lane >> 4;
    for (int r0 = 0; r0 < R; r0 += 16) {
        const bool aok = r0 + m < R;
        v4f acc[4];
        NR_PRAGMA_UNROLL
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + 4 * kk + r;
            float4 c4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (row < R) {
                if (accumulate) c4 = nr_gld4(C + row * 64 + 4 * m);
                else if (bias_) { const float bv = bias[row]; c4 = make_float4(bv, bv, bv, bv); }
            }
            acc[0][r] = c4.x; acc[1][r] = c4.y; acc[2][r] = c4.z; acc[3][r] = c4.w;
        }
        for (int b0 = 0; b0 < Q; b0 += 32) {               // 8 A + 8 wide B loads in flight, then 32 MFMAs
            float a[8];
            float4 b[8];
            NR_PRAGMA_UNROLL
            for (int u = 0; u < 8; ++u) {
                const int q = b0 + 8 * kk + u;
                a[u] = (aok && q < Q) ? W[(r0 + m) * rs + q * qs] : 0.0f;
                b[u] = q < Q ? nr_gld4(Bm + q * 64 + 4 * m) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            NR_PRAGMA_UNROLL
            for (int u = 0; u < 8; ++u) {
                acc[0] = nr_mfma16(a[u], b[u].x, acc[0]); acc[1] = nr_mfma16(a[u], b[u].y, acc[1]);
                acc[2] = nr_mfma16(a[u], b[u].z, acc[2]); acc[3] = nr_mfma16(a[u], b[u].w, acc[3]);
            }
        }
        NR_PRAGMA_UNROLL
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + 4 * kk + r;
            if (row < R)
                nr_gst4(C + row * 64 + 4 * m, make_float4(bwd_act(acc[0][r], act), bwd_act(acc[1][r], act), bwd_act(acc[2][r], act), bwd_act(acc[3][r], act)));
        }
    }
    __syncthreads();
}
// Y[o] = act(b[o] + sum_k W[o*ldw + k] X[k])
__device__ __forceinline__ void bwd_dense(const float* __restrict__ W, int ldw, const float* __restrict__ b, int O, int K,
                                          const float* X, float* Y, int act, int lane) {
    bwd_mm(W, ldw, 1, O, K, X, Y, b, act, false, lane);
}
// dY[o] *= act'(Y[o])
__device__ __noinline__ void bwd_through_act(float* __restrict__ dY_, const float* __restrict__ Y_, int O, int act, int lane) {
    NR_GLOBAL_PTR(float) dY = NR_TO_GLOBAL(float, dY_);
    NR_GLOBAL_PTR(const float) Y = NR_TO_GLOBAL(const float, Y_);
    int o = 0;
    for (; o + 8 <= O; o += 8) {        // eight rows per batch: the loads are issued together
        float d[8], y[8];
        NR_PRAGMA_UNROLL
        for (int j = 0; j < 8; ++j) { d[j] = dY[(o + j) * 64 + lane]; y[j] = Y[(o + j) * 64 + lane]; }
        NR_PRAGMA_UNROLL
        for (int j = 0; j < 8; ++j) dY[(o + j) * 64 + lane] = d[j] * bwd_dact(y[j], act);
    }
    for (; o < O; ++o) dY[o * 64 + lane] *= bwd_dact(Y[o * 64 + lane], act);
}
// dX[k] (+)= sum_o W[o*ldw + k] dY[o]
__device__ __forceinline__ void bwd_dense_dx(const float* __restrict__ W, int ldw, int O, int K, const float* dY, float* dX,
                                             bool accumulate, int lane) {
    bwd_mm(W, 1, ldw, K, O, dY, dX, nullptr, BA_NONE, accumulate, lane);
}
// dW[o*ldw + k] += sum_lanes dY[o][lane] X[k][lane],  db[o] += sum_lanes dY[o][lane]: a [O x 64] x [64 x K] contraction
// over the wave's 64 (point, view) columns -> fp32 MFMA 16x16x4 per 16 x 16 tile of dW.  The contraction index (the
// column l) is dealt as l = 16 kk + s to lane group kk in MFMA step s, so each lane reads 16 consecutive floats of its
// row (four 16-byte loads):  A[m][kk] = dY[o0 + m][16 kk + s],  B[kk][n] = X[k0 + n][16 kk + s];
// D: lane (c, g) holds dW[o0 + 4 g + r][k0 + c].
__device__ __noinline__ void bwd_dense_dw(float* dW_, int ldw, float* db_, int O, int K, const float* dY_, const float* X_, int lane) {
    __syncthreads();
    NR_GLOBAL_PTR(float) dW = NR_TO_GLOBAL(float, dW_);
    NR_GLOBAL_PTR(float) db = NR_TO_GLOBAL(float, db_);
    NR_GLOBAL_PTR(const float) dY = NR_TO_GLOBAL(const float, dY_);
    NR_GLOBAL_PTR(const float) X = NR_TO_GLOBAL(const float, X_);
    O = NR_UNIFORM(O); K = NR_UNIFORM(K); ldw = NR_UNIFORM(ldw);
    const int m = lane & 15, kk = lane >> 4;
    for (int o0 = 0; o0 < O; o0 += 16) {
        float4 a[4];
        const bool aok = o0 + m < O;
        NR_PRAGMA_UNROLL
        for (int j = 0; j < 4; ++j)
            a[j] = aok ? nr_gld4(dY + (o0 + m) * 64 + 16 * kk + 4 * j) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int k0 = 0; k0 < K; k0 += 32) {               // two 16-column tiles of dW per pass
            float4 b[2][4];
            NR_PRAGMA_UNROLL
            for (int t = 0; t < 2; ++t) {
                const bool bok = k0 + 16 * t + m < K;
                NR_PRAGMA_UNROLL
                for (int j = 0; j < 4; ++j)
                    b[t][j] = bok ? nr_gld4(X + (k0 + 16 * t + m) * 64 + 16 * kk + 4 * j)
                                  : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            NR_PRAGMA_UNROLL
            for (int t = 0; t < 2; ++t) {
                v4f acc; acc[0] = 0.0f; acc[1] = 0.0f; acc[2] = 0.0f; acc[3] = 0.0f;
                NR_PRAGMA_UNROLL
                for (int j = 0; j < 4; ++j) {
                    acc = nr_mfma16(a[j].x, b[t][j].x, acc); acc = nr_mfma16(a[j].y, b[t][j].y, acc);
                    acc = nr_mfma16(a[j].z, b[t][j].z, acc); acc = nr_mfma16(a[j].w, b[t][j].w, acc);
                }
                NR_PRAGMA_UNROLL
                for (int r = 0; r < 4; ++r) {
                    const int o = o0 + 4 * kk + r, k = k0 + 16 * t + m;
                    if (o < O && k < K) atomicAdd(NR_FROM_GLOBAL(float, dW + o * ldw + k), acc[r]);
                }
            }
        }
    }
    if (db_)
        for (int o = lane; o < O; o += 64) {
            float sacc = 0.0f;
            for (int l = 0; l < 16; ++l) { const float4 a4 = nr_gld4(dY + o * 64 + 4 * l); sacc += (a4.x + a4.y) + (a4.z + a4.w); }
            atomicAdd(NR_FROM_GLOBAL(float, db + o), sacc);
        }
    __syncthreads();
}
// sum / max over the VP lanes of a point (every lane of the group receives the result)
__device__ __forceinline__ float vp_sum(float v, int vp) {
    for (int m = 1; m < vp; m <<= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float vp_max(float v, int vp) {
    for (int m = 1; m < vp; m <<= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float bwd_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float bwd_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// backward of the output non-linearities and the three (four) 32->32->32->out MLPs of the dist decoder
// (dist_decoder.py:64-97): gradients of mu (softplus), s (softplus + bias), aw / nu (sigmoid) -> weight gradients and
// DFR += d f_ray.  FR: the 32 input rows; S0..S3: 64-row scratch areas.
__device__ __noinline__ void bwd_dist_heads(const float* flat, float* d_flat, bool has_vis, float var_bias,
                                               const float* FR, float* S0, float* S1, float* S2, float* S3, float* DFR,
                                               float mu0, float mu1, float sd0, float sd1, float aw, float nu,
                                               float dmu0, float dmu1, float dsd0, float dsd1, float daw, float dnu, int lane) {
    for (int head = 0; head < (has_vis ? 4 : 3); ++head) {
        const int t0 = head == 0 ? T_MEAN0_W : (head == 1 ? T_VAR0_W : (head == 2 ? T_AW0_W : T_VIS0_W));
        const int nout = head < 2 ? 2 : 1;
        const float* f = flat; float* g = d_flat;
        bwd_dense(f + tensor_offset(t0), 32, f + tensor_offset(t0 + 1), 32, 32, FR, S0, BA_ELU, lane);
        bwd_dense(f + tensor_offset(t0 + 2), 32, f + tensor_offset(t0 + 3), 32, 32, S0, S1, BA_ELU, lane);
        float d0, d1 = 0.0f;
        if (head == 0) { d0 = dmu0 * (1.0f - expf(-mu0)); d1 = dmu1 * (1.0f - expf(-mu1)); }      // softplus' = sigmoid = 1 - exp(-softplus)
        else if (head == 1) { d0 = dsd0 * (1.0f - expf(-(sd0 - var_bias))); d1 = dsd1 * (1.0f - expf(-(sd1 - var_bias))); }
        else if (head == 2) d0 = daw * aw * (1.0f - aw);
        else d0 = dnu * nu * (1.0f - nu);
        S2[lane] = d0; S2[64 + lane] = d1;
        bwd_dense_dw(g + tensor_offset(t0 + 4), 32, g + tensor_offset(t0 + 5), nout, 32, S2, S1, lane);
        float* DH = S2 + 8 * 64;      // rows 8..39: d of the 32-wide hiddens
        bwd_dense_dx(f + tensor_offset(t0 + 4), 32, nout, 32, S2, DH, false, lane);
        bwd_through_act(DH, S1, 32, BA_ELU, lane);
        bwd_dense_dw(g + tensor_offset(t0 + 2), 32, g + tensor_offset(t0 + 3), 32, 32, DH, S0, lane);
        bwd_dense_dx(f + tensor_offset(t0 + 2), 32, 32, 32, DH, S3, false, lane);
        bwd_through_act(S3, S0, 32, BA_ELU, lane);
        bwd_dense_dw(g + tensor_offset(t0), 32, g + tensor_offset(t0 + 1), 32, 32, S3, FR, lane);
        bwd_dense_dx(f + tensor_offset(t0), 32, 32, 32, S3, DFR, true, lane);
    }
}

// forward of the dist decoder heads on the 32 rows FR (outputs only)
__device__ __noinline__ void bwd_dist_heads_fwd(const float* flat, bool has_vis, float var_bias, const float* FR, float* S0,
                                                   float* S1, float* S2, float& mu0, float& mu1, float& sd0, float& sd1,
                                                   float& aw, float& nu, int lane) {
    const float* f = flat;
    bwd_dense(f + tensor_offset(T_MEAN0_W), 32, f + tensor_offset(T_MEAN0_B), 32, 32, FR, S0, BA_ELU, lane);
    bwd_dense(f + tensor_offset(T_MEAN2_W), 32, f + tensor_offset(T_MEAN2_B), 32, 32, S0, S1, BA_ELU, lane);
    bwd_dense(f + tensor_offset(T_MEAN4_W), 32, f + tensor_offset(T_MEAN4_B), 2, 32, S1, S2, BA_NONE, lane);
    mu0 = bwd_softplus(S2[lane]); mu1 = bwd_softplus(S2[64 + lane]);
    bwd_dense(f + tensor_offset(T_VAR0_W), 32, f + tensor_offset(T_VAR0_B), 32, 32, FR, S0, BA_ELU, lane);
    bwd_dense(f + tensor_offset(T_VAR2_W), 32, f + tensor_offset(T_VAR2_B), 32, 32, S0, S1, BA_ELU, lane);
    bwd_dense(f + tensor_offset(T_VAR4_W), 32, f + tensor_offset(T_VAR4_B), 2, 32, S1, S2, BA_NONE, lane);
    sd0 = bwd_softplus(S2[lane]) + var_bias; sd1 = bwd_softplus(S2[64 + lane]) + var_bias;
    bwd_dense(f + tensor_offset(T_AW0_W), 32, f + tensor_offset(T_AW0_B), 32, 32, FR, S0, BA_ELU, lane);
    bwd_dense(f + tensor_offset(T_AW2_W), 32, f + tensor_offset(T_AW2_B), 32, 32, S0, S1, BA_ELU, lane);
    bwd_dense(f + tensor_offset(T_AW4_W), 32, f + tensor_offset(T_AW4_B), 1, 32, S1, S2, BA_NONE, lane);
    aw = bwd_sigmoid(S2[lane]);
    nu = 1.0f;
    if (has_vis) {
        bwd_dense(f + tensor_offset(T_VIS0_W), 32, f + tensor_offset(T_VIS0_B), 32, 32, FR, S0, BA_ELU, lane);
        bwd_dense(f + tensor_offset(T_VIS2_W), 32, f + tensor_offset(T_VIS2_B), 32, 32, S0, S1, BA_ELU, lane);
        bwd_dense(f + tensor_offset(T_VIS4_W), 32, f + tensor_offset(T_VIS4_B), 1, 32, S1, S2, BA_NONE, lane);
        nu = bwd_sigmoid(S2[lane]);
    }
}

// gradient of (visibility, hit) of one interval [near, far] (dist_decoder.py:109-140) w.r.t. the mixture parameters,
// accumulated into dmu*, dsd*, daw, dnu.  nuu = nu if the decoder's use_vis else 1.
__device__ __forceinline__ void bwd_prob(float nearv, float farv, float mu0, float mu1, float sd0, float sd1, float aw, float nuu,
                                         bool use_vis, float dv, float dh, float& dmu0, float& dmu1, float& dsd0, float& dsd1,
                                         float& daw, float& dnu) {
    const float t00 = tanhf((nearv - mu0) * sd0), t01 = tanhf((nearv - mu1) * sd1);
    const float t10 = tanhf((farv - mu0) * sd0), t11 = tanhf((farv - mu1) * sd1);
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

#define FW(T) (p.flat + tensor_offset(T))
#define DW(T) (p.d_flat + tensor_offset(T))

// waves-per-EU 4: the kernel body fits 128 VGPRs (with spills) and the AMDGPU attributor hands the same bound to the
// non-inlined helpers ONLY if every kernel that calls them asks for it - the two small kernels below carry (64, 4) for
// that reason (with plain (64) the helpers took up to 213 VGPRs and this kernel ran at 2 waves per SIMD).  Asking for 2
// lets the allocator take all 512 registers and run 1 wave per SIMD (28 ms instead of 23 ms per training step).
__global__ void __launch_bounds__(64, 4) points_backward_kernel(PointBwdParams p) {
    const int lane = threadIdx.x & 63;
    float* A = p.workspace + (size_t)blockIdx.x * kBwdRows * 64;
    // per-lane scalars that live from the forward to the backward stages are kept in arena rows, not registers: the
    // out-of-line dense helpers are called ~90 times and every live register would be saved around each call
    float* SC = A + BR_SC * 64 + lane;
#define SCALAR(name, idx) float& name = SC[(idx) * 64]
    const int vp = p.vp, ppw = 64 / vp;
    const int pl = lane / vp, v = lane % vp;
    const int npts = p.rn * p.dn, dn = p.dn;
    const float* __restrict__ qc = p.que_const;
    const float qnearp = qc[24], qfarp = qc[25], qinv = qc[27];
    const size_t fmap = (size_t)p.fh * p.fw * 32, imap = (size_t)p.h * p.w * 4;
    const bool has_vis = p.has_vis_head != 0, use_vis = has_vis && (p.use_vis != 0);
    // (distinct row ranges of the arena: __restrict__ lets the per-row loops overlap their loads and stores)
    float* __restrict__ FR = A + BR_FR * 64; float* __restrict__ FI = A + BR_FI * 64; float* __restrict__ RGB = A + BR_RGB * 64;
    float* __restrict__ DL = A + BR_DL * 64;
    float* __restrict__ GL = A + BR_GL * 64; float* __restrict__ GP = A + BR_GP * 64; float* __restrict__ E = A + BR_E * 64;
    float* __restrict__ X = A + BR_X * 64; float* __restrict__ X2 = A + BR_X2 * 64;
    float* __restrict__ DGL = A + BR_DGL * 64; float* __restrict__ DGP = A + BR_DGP * 64; float* __restrict__ DE = A + BR_DE * 64;
    float* __restrict__ DFR = A + BR_DFR * 64; float* __restrict__ DX = A + BR_DX * 64;
    float* __restrict__ S0 = A + BR_S0 * 64; float* __restrict__ S1 = A + BR_S1 * 64; float* __restrict__ S2 = A + BR_S2 * 64;
    float* __restrict__ S3 = A + BR_S3 * 64;

    for (int base = blockIdx.x * ppw; base < npts; base += gridDim.x * ppw) {
        __syncthreads();
        // ================= geometry, gathers (as points_kernel) =================
        int pi = base + pl;
        const bool pvalid = pi < npts;
        pi = pvalid ? pi : npts - 1;
        const bool vok = v < p.rfn;
        const int view = vok ? v : p.rfn - 1;
        const int ray = pi / dn, smp = pi - ray * dn;
        const Ray r = make_ray<false>(qc, p.coords[2 * ray], p.coords[2 * ray + 1]);
        const float* drow = p.depth + (size_t)ray * dn;
        const float d = drow[smp];
        const float s_c = norm_inv_depth_fast(d, qnearp, qfarp, qinv);
        const float s_n = norm_inv_depth_fast(drow[smp + 1 < dn ? smp + 1 : smp], qnearp, qfarp, qinv);
        const float s_p = norm_inv_depth_fast(drow[smp > 0 ? smp - 1 : 0], qnearp, qfarp, qinv);
        const float half_c = (smp == dn - 1) ? 500000.0f : (s_n - s_c) * 0.5f;
        const float half_p = (s_c - s_p) * 0.5f;
        SCALAR(hi, 0); SCALAR(lo, 1);
        hi = half_c; lo = (smp == 0) ? half_c : half_p;
        const float px = rn_add(r.cx, rn_mul(r.dx, d)), py = rn_add(r.cy, rn_mul(r.dy, d)), pz = rn_add(r.cz, rn_mul(r.dz, d));
        const float* __restrict__ vc = p.view_const + view * kViewConst;
        Proj pr = project_point<false>(vc, px, py, pz, (float)p.w, (float)p.h);
        SCALAR(m, 2); SCALAR(tref, 3); SCALAR(pu, 4); SCALAR(pv, 5);
        m = vok ? pr.mask : 0.0f;
        tref = norm_inv_depth_fast(fmaxf(pr.z, 1e-5f), vc[15], vc[16], vc[17]);
        pu = pr.u; pv = pr.v;
        {
            const Taps tf = make_taps(pr.u, pr.v, p.w, p.h, p.fw, p.fh);
            const Taps tc = make_taps(pr.u, pr.v, p.w, p.h, p.w, p.h);
            const float* rf = p.ray_feats + (size_t)view * fmap;
            const float* im = p.img_feats + (size_t)view * fmap;
            const float* cm = p.rgba + (size_t)view * imap;
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) {
                FR[c * 64 + lane] = m * (tf.w00 * rf[(size_t)tf.o00 * 32 + c] + tf.w10 * rf[(size_t)tf.o10 * 32 + c] +
                                         tf.w01 * rf[(size_t)tf.o01 * 32 + c] + tf.w11 * rf[(size_t)tf.o11 * 32 + c]);
                FI[c * 64 + lane] = m * (tf.w00 * im[(size_t)tf.o00 * 32 + c] + tf.w10 * im[(size_t)tf.o10 * 32 + c] +
                                         tf.w01 * im[(size_t)tf.o01 * 32 + c] + tf.w11 * im[(size_t)tf.o11 * 32 + c]);
            }
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 3; ++c)
                RGB[c * 64 + lane] = m * (tc.w00 * cm[(size_t)tc.o00 * 4 + c] + tc.w10 * cm[(size_t)tc.o10 * 4 + c] +
                                          tc.w01 * cm[(size_t)tc.o01 * 4 + c] + tc.w11 * cm[(size_t)tc.o11 * 4 + c]);
            DL[0 * 64 + lane] = pr.dirx - r.qx; DL[1 * 64 + lane] = pr.diry - r.qy; DL[2 * 64 + lane] = pr.dirz - r.qz;
            DL[3 * 64 + lane] = dot3(pr.dirx, pr.diry, pr.dirz, r.qx, r.qy, r.qz);
        }
        // ================= forward =================
        // ---- dist decoder heads (dist_decoder.py:64-97): only the outputs are kept
        SCALAR(mu0, 6); SCALAR(mu1, 7); SCALAR(sd0, 8); SCALAR(sd1, 9); SCALAR(aw, 10); SCALAR(nu, 11);
        nu = 1.0f;
        {
            bwd_dense(FW(T_MEAN0_W), 32, FW(T_MEAN0_B), 32, 32, FR, S0, BA_ELU, lane);
            bwd_dense(FW(T_MEAN2_W), 32, FW(T_MEAN2_B), 32, 32, S0, S1, BA_ELU, lane);
            bwd_dense(FW(T_MEAN4_W), 32, FW(T_MEAN4_B), 2, 32, S1, S2, BA_NONE, lane);
            mu0 = bwd_softplus(S2[lane]); mu1 = bwd_softplus(S2[64 + lane]);
            bwd_dense(FW(T_VAR0_W), 32, FW(T_VAR0_B), 32, 32, FR, S0, BA_ELU, lane);
            bwd_dense(FW(T_VAR2_W), 32, FW(T_VAR2_B), 32, 32, S0, S1, BA_ELU, lane);
            bwd_dense(FW(T_VAR4_W), 32, FW(T_VAR4_B), 2, 32, S1, S2, BA_NONE, lane);
            sd0 = bwd_softplus(S2[lane]) + p.var_bias; sd1 = bwd_softplus(S2[64 + lane]) + p.var_bias;
            bwd_dense(FW(T_AW0_W), 32, FW(T_AW0_B), 32, 32, FR, S0, BA_ELU, lane);
            bwd_dense(FW(T_AW2_W), 32, FW(T_AW2_B), 32, 32, S0, S1, BA_ELU, lane);
            bwd_dense(FW(T_AW4_W), 32, FW(T_AW4_B), 1, 32, S1, S2, BA_NONE, lane);
            aw = bwd_sigmoid(S2[lane]);
            if (has_vis) {
                bwd_dense(FW(T_VIS0_W), 32, FW(T_VIS0_B), 32, 32, FR, S0, BA_ELU, lane);
                bwd_dense(FW(T_VIS2_W), 32, FW(T_VIS2_B), 32, 32, S0, S1, BA_ELU, lane);
                bwd_dense(FW(T_VIS4_W), 32, FW(T_VIS4_B), 1, 32, S1, S2, BA_NONE, lane);
                nu = bwd_sigmoid(S2[lane]);
            }
        }
        // ---- probabilities (dist_decoder.py:109-140, renderer.py:79-82)
        SCALAR(nuu, 12);
        nuu = use_vis ? nu : 1.0f;
        const float a00 = (tref - lo - mu0) * sd0, a01 = (tref - lo - mu1) * sd1;
        const float a10 = (tref + hi - mu0) * sd0, a11 = (tref + hi - mu1) * sd1;
        const float t00 = tanhf(a00), t01 = tanhf(a01), t10 = tanhf(a10), t11 = tanhf(a11);
        const float g00 = 0.5f + 0.5f * t00, g01 = 0.5f + 0.5f * t01, g10 = 0.5f + 0.5f * t10, g11 = 0.5f + 0.5f * t11;
        const float c00 = g00 * nuu, c01 = g01 * nuu, c10 = g10 * nuu, c11 = g11 * nuu;
        const float mix0 = aw, mix1 = 1.0f - aw;
        const float vis_raw = (1.0f - c00) * mix0 + (1.0f - c01) * mix1;
        const float hit_raw = (c10 - c00) * mix0 + (c11 - c01) * mix1;
        SCALAR(vis, 13); SCALAR(hit, 14);
        vis = vis_raw * m; hit = hit_raw * m;
        // ---- prob_embed (aggregate_net.py:43): input [f_ray, 2 hit - 1, 2 vis - 1]; hidden kept in S0 for the backward? no:
        // recomputed there.  E is kept.
        {
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) S3[c * 64 + lane] = FR[c * 64 + lane];
            S3[32 * 64 + lane] = (hit - 0.5f) * 2.0f; S3[33 * 64 + lane] = (vis - 0.5f) * 2.0f;
            bwd_dense(FW(T_PE0_W), 34, FW(T_PE0_B), 32, 34, S3, S0, BA_RELU, lane);
            bwd_dense(FW(T_PE2_W), 32, FW(T_PE2_B), 32, 32, S0, E, BA_NONE, lane);
        }
        // ---- ray_dir_fc, rgb_feat + direction feature (ibrnet.py:324-327)
        {
            bwd_dense(FW(T_RD0_W), 4, FW(T_RD0_B), 16, 4, DL, S0, BA_ELU, lane);
            bwd_dense(FW(T_RD2_W), 16, FW(T_RD2_B), 35, 16, S0, S1, BA_ELU, lane);
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 3; ++c) GP[c * 64 + lane] = RGB[c * 64 + lane] + S1[c * 64 + lane];
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) GP[(3 + c) * 64 + lane] = FI[c * 64 + lane] + S1[(3 + c) * 64 + lane];
        }
        // ---- neuray_fc -> sigmoid (ibrnet.py:337)
        SCALAR(sn, 15);
        {
            bwd_dense(FW(T_NF0_W), 32, FW(T_NF0_B), 8, 32, E, S0, BA_ELU, lane);
            bwd_dense(FW(T_NF2_W), 8, FW(T_NF2_B), 1, 8, S0, S1, BA_NONE, lane);
            sn = bwd_sigmoid(S1[lane]);
        }
        // ---- cross-view statistics (ibrnet.py:334-340)
        SCALAR(wv, 16); SCALAR(w0, 17); SCALAR(sa0, 18); SCALAR(sa1, 19);
        {
            const float msum = vp_sum(m, vp);
            wv = m / (msum + 1e-8f);
            w0 = sn * wv;
            sa0 = vp_sum(w0, vp); sa1 = vp_sum(wv, vp);
        }
        NR_PRAGMA_UNROLL4
        for (int f = 0; f < 35; ++f) {
            const float x = GP[f * 64 + lane];
            const float mean0 = vp_sum(w0 * x, vp), mean1 = vp_sum(wv * x, vp);
            const float var0 = vp_sum(w0 * (x - mean0) * (x - mean0), vp), var1 = vp_sum(wv * (x - mean1) * (x - mean1), vp);
            GL[f * 64 + lane] = mean0; GL[(35 + f) * 64 + lane] = var0; GL[(70 + f) * 64 + lane] = mean1; GL[(105 + f) * 64 + lane] = var1;
        }
        // ---- base_fc (ibrnet.py:342): hidden in S0 (64), X kept
        bwd_dense(FW(T_BASE0_W), 207, FW(T_BASE0_B), 64, 207, GL, S0, BA_ELU, lane);
        bwd_dense(FW(T_BASE2_W), 64, FW(T_BASE2_B), 32, 64, S0, X, BA_ELU, lane);
        // ---- vis_fc (ibrnet.py:343-346)
        SCALAR(visp, 20); SCALAR(vy32, 21);          // vis' = sigmoid(ELU(.)) * mask; vy32 = the ELU output it is taken from
        {
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) S3[c * 64 + lane] = X[c * 64 + lane] * wv;
            bwd_dense(FW(T_VF0_W), 32, FW(T_VF0_B), 32, 32, S3, S0, BA_ELU, lane);
            bwd_dense(FW(T_VF2_W), 32, FW(T_VF2_B), 33, 32, S0, S1, BA_ELU, lane);
            vy32 = S1[32 * 64 + lane];
            visp = bwd_sigmoid(vy32) * m;
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) X2[c * 64 + lane] = X[c * 64 + lane] + S1[c * 64 + lane];
        }
        // ---- vis_fc2 (ibrnet.py:347-348)
        SCALAR(vis2, 22); SCALAR(v2sig, 23);
        {
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) S3[c * 64 + lane] = X2[c * 64 + lane] * visp;
            bwd_dense(FW(T_V20_W), 32, FW(T_V20_B), 32, 32, S3, S0, BA_ELU, lane);
            bwd_dense(FW(T_V22_W), 32, FW(T_V22_B), 1, 32, S0, S1, BA_NONE, lane);
            v2sig = bwd_sigmoid(S1[lane]);
            vis2 = v2sig * m;
        }
        // ---- rgb_fc (ibrnet.py:363-365): input [x, vis, ray_diff]
        SCALAR(z, 24);
        {
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) S3[c * 64 + lane] = X2[c * 64 + lane];
            S3[32 * 64 + lane] = vis2;
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 4; ++c) S3[(33 + c) * 64 + lane] = DL[c * 64 + lane];
            bwd_dense(FW(T_RF0_W), 37, FW(T_RF0_B), 16, 37, S3, S0, BA_ELU, lane);
            bwd_dense(FW(T_RF2_W), 16, FW(T_RF2_B), 8, 16, S0, S1, BA_ELU, lane);
            bwd_dense(FW(T_RF4_W), 8, FW(T_RF4_B), 1, 8, S1, S2, BA_NONE, lane);
            z = m > 0.0f ? S2[lane] : -1e9f;
        }
        // ---- softmax blend weights, visibility-weighted statistics (ibrnet.py:350-354,366-367)
        SCALAR(beta, 25); SCALAR(svis, 26); SCALAR(wh, 27); SCALAR(swh, 28);
        {
            const float zmax = vp_max(z, vp);
            const float ez = expf(z - zmax);
            beta = ez / vp_sum(ez, vp);
            svis = vp_sum(vis2, vp);
            wh = vis2 / (svis + 1e-8f);
            swh = vp_sum(wh, vp);
        }
        // geometry_fc input [mean(32) var(32) mean weight] in S3 rows 0..64 (identical in the VP lanes of a point)
        NR_PRAGMA_UNROLL4
        for (int f = 0; f < 32; ++f) {
            const float x = X2[f * 64 + lane];
            const float mean = vp_sum(wh * x, vp);
            const float var = vp_sum(wh * (x - mean) * (x - mean), vp);
            S3[f * 64 + lane] = mean; S3[(32 + f) * 64 + lane] = var;
        }
        // wgt.mean(2): mean over the rfn views of the normalised weights (ibrnet.py:354)
        // (S3 row 64 lives in the next scratch area's first row: S3 has 64 rows -> use DX row 0 temporarily? no: keep in a register
        //  and write it to a dedicated row of S2 when geometry_fc runs)
        SCALAR(meanw, 29);
        meanw = swh / (float)p.rfn;

        // ================= backward =================
        const float* up = p.d_point_rec + (size_t)pi * kPointRec;
        const float gsc = pvalid ? 1.0f : 0.0f;
        const float own = (v == 0 && pvalid) ? 1.0f : 0.0f;       // per-point layers: one lane of the point carries the gradient
        // ---- geometry_fc (ibrnet.py:353-354): input rows: S3[0..63] + meanw; hidden S0 (64); output 16
        float dmean_w;
        {
            // forward recompute with the 65-wide input assembled in DGL (free at this point): rows 0..64
            NR_PRAGMA_UNROLL4
            for (int f = 0; f < 64; ++f) DGL[f * 64 + lane] = S3[f * 64 + lane];
            DGL[64 * 64 + lane] = meanw;
            bwd_dense(FW(T_GF0_W), 65, FW(T_GF0_B), 64, 65, DGL, S0, BA_ELU, lane);
            bwd_dense(FW(T_GF2_W), 64, FW(T_GF2_B), 16, 64, S0, S1, BA_ELU, lane);
            NR_PRAGMA_UNROLL4
            for (int o = 0; o < 16; ++o) S2[o * 64 + lane] = up[o] * own * bwd_dact(S1[o * 64 + lane], BA_ELU);
            bwd_dense_dw(DW(T_GF2_W), 64, DW(T_GF2_B), 16, 64, S2, S0, lane);
            bwd_dense_dx(FW(T_GF2_W), 64, 16, 64, S2, S1, false, lane);          // S1 <- d hidden (64)
            bwd_through_act(S1, S0, 64, BA_ELU, lane);
            bwd_dense_dw(DW(T_GF0_W), 65, DW(T_GF0_B), 64, 65, S1, DGL, lane);
            float* DIN = DGL + 70 * 64;                                          // 65 free rows behind the input copy
            bwd_dense_dx(FW(T_GF0_W), 65, 64, 65, S1, DIN, false, lane);         // d [mean var meanw] (own lane only)
            // broadcast the per-point gradient to the VP lanes of the point; S2 <- d mean (0..31), d var (32..63)
            NR_PRAGMA_UNROLL4
            for (int f = 0; f < 64; ++f) S2[f * 64 + lane] = vp_sum(DIN[f * 64 + lane], vp);
            dmean_w = vp_sum(DIN[64 * 64 + lane], vp);
        }
        // ---- visibility-weighted mean / variance + softmax blend: -> dX2 (DX), dvis2, dz
        float dvis2, dz;
        {
            float dwh = dmean_w / (float)p.rfn;
            NR_PRAGMA_UNROLL4
            for (int f = 0; f < 32; ++f) {
                const float x = X2[f * 64 + lane], mean = S3[f * 64 + lane];
                const float dmean = S2[f * 64 + lane], dvar = S2[(32 + f) * 64 + lane];
                const float dmt = dmean - 2.0f * dvar * mean * (1.0f - swh);
                DX[f * 64 + lane] = wh * (dmt + 2.0f * (x - mean) * dvar);
                dwh += dmt * x + dvar * (x - mean) * (x - mean);
            }
            const float sdw = vp_sum(dwh * wh, vp);
            dvis2 = (dwh - sdw) / (svis + 1e-8f);
            // colour = sum_v beta_v rgb_in_v
            const float dbeta = (up[16] * RGB[lane] + up[17] * RGB[64 + lane] + up[18] * RGB[128 + lane]) * gsc;
            const float sbb = vp_sum(beta * dbeta, vp);
            dz = beta * (dbeta - sbb);
            if (!(m > 0.0f)) dz = 0.0f;
        }
        // ---- rgb_fc backward
        {
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) S3[c * 64 + lane] = X2[c * 64 + lane];
            S3[32 * 64 + lane] = vis2;
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 4; ++c) S3[(33 + c) * 64 + lane] = DL[c * 64 + lane];
            bwd_dense(FW(T_RF0_W), 37, FW(T_RF0_B), 16, 37, S3, S0, BA_ELU, lane);
            bwd_dense(FW(T_RF2_W), 16, FW(T_RF2_B), 8, 16, S0, S1, BA_ELU, lane);
            S2[lane] = dz;
            bwd_dense_dw(DW(T_RF4_W), 8, DW(T_RF4_B), 1, 8, S2, S1, lane);
            float* D8 = S2 + 8 * 64;          // d of the 8-wide hidden
            bwd_dense_dx(FW(T_RF4_W), 8, 1, 8, S2, D8, false, lane);
            bwd_through_act(D8, S1, 8, BA_ELU, lane);
            bwd_dense_dw(DW(T_RF2_W), 16, DW(T_RF2_B), 8, 16, D8, S0, lane);
            float* D16 = S2 + 16 * 64;
            bwd_dense_dx(FW(T_RF2_W), 16, 8, 16, D8, D16, false, lane);
            bwd_through_act(D16, S0, 16, BA_ELU, lane);
            bwd_dense_dw(DW(T_RF0_W), 37, DW(T_RF0_B), 16, 37, D16, S3, lane);
            bwd_dense_dx(FW(T_RF0_W), 37, 16, 37, D16, S1, false, lane);         // S1 rows 0..36 <- d [x2, vis2, ray_diff]
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) DX[c * 64 + lane] += S1[c * 64 + lane];
            dvis2 += S1[32 * 64 + lane];
        }
        // ---- vis_fc2 backward: vis2 = sigmoid(a) * m
        float dvisp;
        {
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) S3[c * 64 + lane] = X2[c * 64 + lane] * visp;
            bwd_dense(FW(T_V20_W), 32, FW(T_V20_B), 32, 32, S3, S0, BA_ELU, lane);
            S2[lane] = dvis2 * m * v2sig * (1.0f - v2sig);
            bwd_dense_dw(DW(T_V22_W), 32, DW(T_V22_B), 1, 32, S2, S0, lane);
            bwd_dense_dx(FW(T_V22_W), 32, 1, 32, S2, S1, false, lane);           // S1 <- d hidden (32)
            bwd_through_act(S1, S0, 32, BA_ELU, lane);
            bwd_dense_dw(DW(T_V20_W), 32, DW(T_V20_B), 32, 32, S1, S3, lane);
            bwd_dense_dx(FW(T_V20_W), 32, 32, 32, S1, S2, false, lane);          // S2 <- d (x2 * vis')
            dvisp = 0.0f;
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) {
                dvisp = fmaf(S2[c * 64 + lane], X2[c * 64 + lane], dvisp);
                DX[c * 64 + lane] = fmaf(S2[c * 64 + lane], visp, DX[c * 64 + lane]);
            }
        }
        // ---- vis_fc backward: x2 = x + r, vis' = sigmoid(ELU(.)) * m; DX holds d x2 and becomes d x
        {
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) S3[c * 64 + lane] = X[c * 64 + lane] * wv;
            bwd_dense(FW(T_VF0_W), 32, FW(T_VF0_B), 32, 32, S3, S0, BA_ELU, lane);
            bwd_dense(FW(T_VF2_W), 32, FW(T_VF2_B), 33, 32, S0, S1, BA_ELU, lane);
            const float sg_ = bwd_sigmoid(vy32);
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) S2[c * 64 + lane] = DX[c * 64 + lane];
            S2[32 * 64 + lane] = dvisp * m * sg_ * (1.0f - sg_);
            bwd_through_act(S2, S1, 33, BA_ELU, lane);
            bwd_dense_dw(DW(T_VF2_W), 32, DW(T_VF2_B), 33, 32, S2, S0, lane);
            bwd_dense_dx(FW(T_VF2_W), 32, 33, 32, S2, S1, false, lane);          // S1 <- d hidden (32)
            bwd_through_act(S1, S0, 32, BA_ELU, lane);
            bwd_dense_dw(DW(T_VF0_W), 32, DW(T_VF0_B), 32, 32, S1, S3, lane);
            bwd_dense_dx(FW(T_VF0_W), 32, 32, 32, S1, S2, false, lane);          // S2 <- d (x * w)
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) DX[c * 64 + lane] = fmaf(S2[c * 64 + lane], wv, DX[c * 64 + lane]);
        }
        // ---- base_fc backward -> d [GL GP E] (DGL DGP DE contiguous)
        {
            bwd_dense(FW(T_BASE0_W), 207, FW(T_BASE0_B), 64, 207, GL, S0, BA_ELU, lane);
            bwd_through_act(DX, X, 32, BA_ELU, lane);
            bwd_dense_dw(DW(T_BASE2_W), 64, DW(T_BASE2_B), 32, 64, DX, S0, lane);
            bwd_dense_dx(FW(T_BASE2_W), 64, 32, 64, DX, S1, false, lane);        // S1 <- d hidden (64)
            bwd_through_act(S1, S0, 64, BA_ELU, lane);
            bwd_dense_dw(DW(T_BASE0_W), 207, DW(T_BASE0_B), 64, 207, S1, GL, lane);
            bwd_dense_dx(FW(T_BASE0_W), 207, 64, 207, S1, DGL, false, lane);
        }
        // ---- cross-view statistics backward: d GL (summed over the views) -> d GP, d sigmoid(neuray_fc)
        float dsn;
        {
            float dw0 = 0.0f;
            NR_PRAGMA_UNROLL4
            for (int f = 0; f < 35; ++f) {
                const float x = GP[f * 64 + lane];
                const float mean0 = GL[f * 64 + lane], mean1 = GL[(70 + f) * 64 + lane];
                const float dmean0 = vp_sum(DGL[f * 64 + lane], vp), dvar0 = vp_sum(DGL[(35 + f) * 64 + lane], vp);
                const float dmean1 = vp_sum(DGL[(70 + f) * 64 + lane], vp), dvar1 = vp_sum(DGL[(105 + f) * 64 + lane], vp);
                const float dmt0 = dmean0 - 2.0f * dvar0 * mean0 * (1.0f - sa0);
                const float dmt1 = dmean1 - 2.0f * dvar1 * mean1 * (1.0f - sa1);
                DGP[f * 64 + lane] += w0 * (dmt0 + 2.0f * (x - mean0) * dvar0) + wv * (dmt1 + 2.0f * (x - mean1) * dvar1);
                dw0 += dmt0 * x + dvar0 * (x - mean0) * (x - mean0);
            }
            dsn = dw0 * wv;            // weight0 = sigmoid(.) * weight; the masks carry no gradient
        }
        // ---- neuray_fc backward -> d E
        {
            bwd_dense(FW(T_NF0_W), 32, FW(T_NF0_B), 8, 32, E, S0, BA_ELU, lane);
            S2[lane] = dsn * sn * (1.0f - sn);
            bwd_dense_dw(DW(T_NF2_W), 8, DW(T_NF2_B), 1, 8, S2, S0, lane);
            bwd_dense_dx(FW(T_NF2_W), 8, 1, 8, S2, S1, false, lane);
            bwd_through_act(S1, S0, 8, BA_ELU, lane);
            bwd_dense_dw(DW(T_NF0_W), 32, DW(T_NF0_B), 8, 32, S1, E, lane);
            bwd_dense_dx(FW(T_NF0_W), 32, 8, 32, S1, DE, true, lane);
        }
        // ---- ray_dir_fc backward (weights only: the direction difference carries no gradient); d img_feats = d GP[3..34]
        {
            bwd_dense(FW(T_RD0_W), 4, FW(T_RD0_B), 16, 4, DL, S0, BA_ELU, lane);
            bwd_dense(FW(T_RD2_W), 16, FW(T_RD2_B), 35, 16, S0, S1, BA_ELU, lane);
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 35; ++c) S2[c * 64 + lane] = DGP[c * 64 + lane] * bwd_dact(S1[c * 64 + lane], BA_ELU);
            bwd_dense_dw(DW(T_RD2_W), 16, DW(T_RD2_B), 35, 16, S2, S0, lane);
            bwd_dense_dx(FW(T_RD2_W), 16, 35, 16, S2, S1, false, lane);
            bwd_through_act(S1, S0, 16, BA_ELU, lane);
            bwd_dense_dw(DW(T_RD0_W), 4, DW(T_RD0_B), 16, 4, S1, DL, lane);
        }
        // ---- prob_embed backward -> d f_ray (DFR), d hit, d vis
        float dhit, dvis;
        {
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) S3[c * 64 + lane] = FR[c * 64 + lane];
            S3[32 * 64 + lane] = (hit - 0.5f) * 2.0f; S3[33 * 64 + lane] = (vis - 0.5f) * 2.0f;
            bwd_dense(FW(T_PE0_W), 34, FW(T_PE0_B), 32, 34, S3, S0, BA_RELU, lane);
            bwd_dense_dw(DW(T_PE2_W), 32, DW(T_PE2_B), 32, 32, DE, S0, lane);
            bwd_dense_dx(FW(T_PE2_W), 32, 32, 32, DE, S1, false, lane);
            bwd_through_act(S1, S0, 32, BA_RELU, lane);
            bwd_dense_dw(DW(T_PE0_W), 34, DW(T_PE0_B), 32, 34, S1, S3, lane);
            bwd_dense_dx(FW(T_PE0_W), 34, 32, 34, S1, S2, false, lane);          // S2 rows 0..33
            NR_PRAGMA_UNROLL4
            for (int c = 0; c < 32; ++c) DFR[c * 64 + lane] = S2[c * 64 + lane];
            dhit = 2.0f * S2[32 * 64 + lane]; dvis = 2.0f * S2[33 * 64 + lane];
        }
        // ---- probabilities backward (dist_decoder.py:109-140)
        float dmu0 = 0.0f, dmu1 = 0.0f, dsd0 = 0.0f, dsd1 = 0.0f, daw = 0.0f, dnu = 0.0f;
        bwd_prob(tref - lo, tref + hi, mu0, mu1, sd0, sd1, aw, nuu, use_vis, dvis * m, dhit * m, dmu0, dmu1, dsd0, dsd1, daw, dnu);
        // ---- dist decoder heads backward -> DFR +=
        bwd_dist_heads(p.flat, p.d_flat, has_vis, p.var_bias, FR, S0, S1, S2, S3, DFR, mu0, mu1, sd0, sd1, aw, nu,
                       dmu0, dmu1, dsd0, dsd1, daw, dnu, lane);
        // ---- gathers backward: f_ray = mask * bilinear(ray_feats), f_img = mask * bilinear(img_feats) (render_ops.py:54-70)
        // Coalesced scatter: the 32 channels of a texel are contiguous (NHWC), so 32 lanes add one texel's channels
        // with one instruction; the two halves of the wave take two taps at a time.  Each pair's tap table (byte-free
        // float offsets + weights, 0 for masked / padded pairs) goes through 9 arena rows and is read back wave-uniformly.
        {
            float* TT = S0;                                   // rows 0..3: texel offsets, 4..7: weights, 8: view map offset
            const Taps tf = make_taps(pu, pv, p.w, p.h, p.fw, p.fh);
            const float sc = (vok && pvalid) ? m : 0.0f;
            TT[0 * 64 + lane] = __int_as_float(tf.o00); TT[1 * 64 + lane] = __int_as_float(tf.o10);
            TT[2 * 64 + lane] = __int_as_float(tf.o01); TT[3 * 64 + lane] = __int_as_float(tf.o11);
            TT[4 * 64 + lane] = tf.w00 * sc; TT[5 * 64 + lane] = tf.w10 * sc; TT[6 * 64 + lane] = tf.w01 * sc; TT[7 * 64 + lane] = tf.w11 * sc;
            TT[8 * 64 + lane] = __int_as_float(view);
            __syncthreads();
            const int c = lane & 31, half = lane >> 5;
            for (int l = 0; l < 64; ++l) {
                const float g_r = DFR[c * 64 + l], g_i = DGP[(3 + c) * 64 + l];
                const size_t voff = (size_t)__float_as_int(TT[8 * 64 + l]) * fmap;
                NR_PRAGMA_UNROLL
                for (int tp = 0; tp < 2; ++tp) {
                    const int tap = 2 * tp + half;
                    const float wt = TT[(4 + tap) * 64 + l];
                    if (wt != 0.0f) {
                        const size_t o = voff + (size_t)__float_as_int(TT[tap * 64 + l]) * 32 + c;
                        atomicAdd(p.d_ray_feats + o, wt * g_r);
                        atomicAdd(p.d_img_feats + o, wt * g_i);
                    }
                }
            }
        }
    }
}
#undef SCALAR
#undef FW
#undef DW

// -------------------------------------------------------------------------------------------------
// a19 backward: hit_prob_self = compute_prob(is_ref=False) of the decoded query-ray distributions
// (renderer.py:137-155, dist_decoder.py:39-46,99-140).  lane = ray; feats [rn][32] are the query view's ray_feats
// gathered at the ray's pixel (neuray_interpolate_feats); -> d_feats [rn][32] and the dist decoder weight gradients.
// -------------------------------------------------------------------------------------------------
struct SelfHitBwdParams {
    const float* que_const;
    const float* depth;       // [rn][dn]
    const float* feats;       // [rn][32]
    const float* flat;
    const float* d_hit;       // [rn][dn]
    float* d_feats;           // [rn][32]
    float* d_flat;            // accumulated
    float* workspace;         // [gridDim.x][kSelfBwdRows][64]
    int rn, dn, has_vis_head, use_vis;
    float var_bias;
};
constexpr int kSelfBwdRows = 32 + 4 * 64 + 32;

__global__ void __launch_bounds__(64, 4) self_hit_backward_kernel(SelfHitBwdParams p) {
    const int lane = threadIdx.x & 63;
    float* A = p.workspace + (size_t)blockIdx.x * kSelfBwdRows * 64;
    float* FR = A; float* S0 = A + 32 * 64; float* S1 = S0 + 64 * 64; float* S2 = S1 + 64 * 64; float* S3 = S2 + 64 * 64;
    float* DFR = S3 + 64 * 64;
    const float nearp = p.que_const[24], farp = p.que_const[25];
    const bool has_vis = p.has_vis_head != 0, use_vis = has_vis && (p.use_vis != 0);
    const int dn = p.dn;
    for (int base = blockIdx.x * 64; base < p.rn; base += gridDim.x * 64) {
        __syncthreads();
        const bool valid = base + lane < p.rn;
        const int ray = valid ? base + lane : p.rn - 1;
        for (int c = 0; c < 32; ++c) { FR[c * 64 + lane] = p.feats[(size_t)ray * 32 + c]; DFR[c * 64 + lane] = 0.0f; }
        float mu0, mu1, sd0, sd1, aw, nu;
        bwd_dist_heads_fwd(p.flat, has_vis, p.var_bias, FR, S0, S1, S2, mu0, mu1, sd0, sd1, aw, nu, lane);
        const float nuu = use_vis ? nu : 1.0f;
        float dmu0 = 0.0f, dmu1 = 0.0f, dsd0 = 0.0f, dsd1 = 0.0f, daw = 0.0f, dnu = 0.0f;
        const float* drow = p.depth + (size_t)ray * dn;
        for (int smp = 0; smp < dn; ++smp) {
            const float t_c = norm_inv_depth(fmaxf(drow[smp], 1e-5f), nearp, farp);
            float lo, hi;
            if (smp == 0) lo = t_c - (norm_inv_depth(drow[1], nearp, farp) - norm_inv_depth(drow[0], nearp, farp)) / 2.0f;
            else lo = (norm_inv_depth(fmaxf(drow[smp - 1], 1e-5f), nearp, farp) + t_c) / 2.0f;
            if (smp == dn - 1) hi = t_c + 500000.0f;
            else hi = (t_c + norm_inv_depth(fmaxf(drow[smp + 1], 1e-5f), nearp, farp)) / 2.0f;
            const float dh = valid ? p.d_hit[(size_t)ray * dn + smp] : 0.0f;
            bwd_prob(lo, hi, mu0, mu1, sd0, sd1, aw, nuu, use_vis, 0.0f, dh, dmu0, dmu1, dsd0, dsd1, daw, dnu);
        }
        bwd_dist_heads(p.flat, p.d_flat, has_vis, p.var_bias, FR, S0, S1, S2, S3, DFR, mu0, mu1, sd0, sd1, aw, nu,
                       dmu0, dmu1, dsd0, dsd1, daw, dnu, lane);
        if (valid)
            for (int c = 0; c < 32; ++c) p.d_feats[(size_t)ray * 32 + c] = DFR[c * 64 + lane];
    }
}


// backward of decoder_rows_kernel (dist decoder on arbitrary rows, dist_decoder.py:99-107,146-151): lane = row.
// d_mean [n][2], d_var [n][2], d_aw [n], d_vis [n] (any may be null) are the gradients w.r.t. the decoder OUTPUTS.
struct RowsBwdParams {
    const float* feats;       // [n][32]
    const float* flat;
    const float* d_mean; const float* d_var; const float* d_aw; const float* d_vis;
    float* d_feats;           // [n][32]
    float* d_flat;            // accumulated
    float* workspace;         // [gridDim.x][kSelfBwdRows][64]
    int n, has_vis_head;
    float var_bias;
};

__global__ void __launch_bounds__(64, 4) decoder_rows_backward_kernel(RowsBwdParams p) {
    const int lane = threadIdx.x & 63;
    float* A = p.workspace + (size_t)blockIdx.x * kSelfBwdRows * 64;
    float* FR = A; float* S0 = A + 32 * 64; float* S1 = S0 + 64 * 64; float* S2 = S1 + 64 * 64; float* S3 = S2 + 64 * 64;
    float* DFR = S3 + 64 * 64;
    const bool has_vis = p.has_vis_head != 0;
    for (int base = blockIdx.x * 64; base < p.n; base += gridDim.x * 64) {
        __syncthreads();
        const bool valid = base + lane < p.n;
        const int row = valid ? base + lane : p.n - 1;
        for (int c = 0; c < 32; ++c) { FR[c * 64 + lane] = p.feats[(size_t)row * 32 + c]; DFR[c * 64 + lane] = 0.0f; }
        float mu0, mu1, sd0, sd1, aw, nu;
        bwd_dist_heads_fwd(p.flat, has_vis, p.var_bias, FR, S0, S1, S2, mu0, mu1, sd0, sd1, aw, nu, lane);
        const float g = valid ? 1.0f : 0.0f;
        const float dmu0 = p.d_mean ? g * p.d_mean[2 * row] : 0.0f, dmu1 = p.d_mean ? g * p.d_mean[2 * row + 1] : 0.0f;
        const float dsd0 = p.d_var ? g * p.d_var[2 * row] : 0.0f, dsd1 = p.d_var ? g * p.d_var[2 * row + 1] : 0.0f;
        const float daw = p.d_aw ? g * p.d_aw[row] : 0.0f, dnu = (p.d_vis && has_vis) ? g * p.d_vis[row] : 0.0f;
        bwd_dist_heads(p.flat, p.d_flat, has_vis, p.var_bias, FR, S0, S1, S2, S3, DFR, mu0, mu1, sd0, sd1, aw, nu,
                       dmu0, dmu1, dsd0, dsd1, daw, dnu, lane);
        if (valid)
            for (int c = 0; c < 32; ++c) p.d_feats[(size_t)row * 32 + c] = DFR[c * 64 + lane];
    }
}

// backward of interpolate_kernel (bilinear, border padding; network/ops.py:14-34): d feats[b][c][fh][fw] += w_tap * d out
__global__ void interpolate_backward_kernel(const float* __restrict__ d_out, const float* __restrict__ points, const float* __restrict__ mask,
                                            int b, int n, int c, int fh, int fw, int h_full, int w_full, int align,
                                            float* __restrict__ d_feats) {
    const long long total = (long long)b * n * c;          // one thread per (point, channel)
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long i = e / c;
        const int ch = (int)(e - i * c);
        const int bi = (int)(i / n);
        const float mk = mask ? mask[i] : 1.0f;
        if (mk == 0.0f) continue;
        const float ix = texel_coord(points[2 * i], (float)w_full, (float)fw, align != 0);
        const float iy = texel_coord(points[2 * i + 1], (float)h_full, (float)fh, align != 0);
        const Taps t = taps_from(ix, iy, fw, fh);
        const int offs[4] = {t.o00, t.o10, t.o01, t.o11};
        const float wts[4] = {t.w00, t.w10, t.w01, t.w11};
        float* dst = d_feats + ((size_t)bi * c + ch) * fh * fw;
        const float g = d_out[e] * mk;
        for (int a = 0; a < 4; ++a)
            if (wts[a] != 0.0f) atomicAdd(dst + offs[a], wts[a] * g);
    }
}

// The same scatter with a point's channels CONTIGUOUS: d tmp[b][fh][fw][c] += w_tap * d out - consecutive lanes (channels of one point) add
// to consecutive floats of one texel, i.e. one cache line per tap and wave instead of one line per lane as in the channel-major map above
// (8.4 M scattered atomics for the generalisation step's 8 x 8192 depth-loss pixels: 0.49 ms) - followed by nhwc_add_to_nchw_kernel.
__global__ void interpolate_backward_nhwc_kernel(const float* __restrict__ d_out, const float* __restrict__ points, const float* __restrict__ mask,
                                                 int b, int n, int c, int fh, int fw, int h_full, int w_full, int align, float* __restrict__ tmp) {
    const long long total = (long long)b * n * c;          // one thread per (point, channel)
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long i = e / c;
        const int ch = (int)(e - i * c);
        const int bi = (int)(i / n);
        const float mk = mask ? mask[i] : 1.0f;
        if (mk == 0.0f) continue;
        const float ix = texel_coord(points[2 * i], (float)w_full, (float)fw, align != 0);
        const float iy = texel_coord(points[2 * i + 1], (float)h_full, (float)fh, align != 0);
        const Taps t = taps_from(ix, iy, fw, fh);
        const int offs[4] = {t.o00, t.o10, t.o01, t.o11};
        const float wts[4] = {t.w00, t.w10, t.w01, t.w11};
        float* dst = tmp + (size_t)bi * fh * fw * c + ch;
        const float g = d_out[e] * mk;
        for (int a = 0; a < 4; ++a)
            if (wts[a] != 0.0f) atomicAdd(dst + (size_t)offs[a] * c, wts[a] * g);
    }
}

// dst[b][c][hw] += tmp[b][hw][c], 32 x 32 tiles through LDS (both sides coalesced).  grid = (ceil(hw / 32), ceil(c / 32) * b), 256 threads
__global__ void __launch_bounds__(256) nhwc_add_to_nchw_kernel(const float* __restrict__ tmp, int hw, int c, float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int cb = (c + 31) / 32;
    const int p0 = blockIdx.x * 32, c0 = (int)(blockIdx.y % cb) * 32, bi = (int)(blockIdx.y / cb);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int pix = p0 + r, ch = c0 + tx;
        tile[r][tx] = (pix < hw && ch < c) ? tmp[((size_t)bi * hw + pix) * c + ch] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ch = c0 + r, pix = p0 + tx;
        if (ch < c && pix < hw) dst[((size_t)bi * c + ch) * hw + pix] += tile[tx][r];
    }
}

}  // namespace nr
