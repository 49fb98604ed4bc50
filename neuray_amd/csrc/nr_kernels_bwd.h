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


// (The first-version backward kernels of rounds 1-5 - points_backward_kernel, self_hit_backward_kernel, decoder_rows_backward_kernel: one
// wave per workgroup, activations as rows of a global-memory arena, 1 323 spilled VGPRs - lived here.  They were alive only for
// rfn 9..16 under autograd, a view count no shipped configuration trains with (dataset/train_dataset.py:73-74, renderer.py:350);
// round 6 removed them: the resident kernels of nr_kernels_bwd2.h are the backward, and more than 8 views under autograd raise.)

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
