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
    int rn, dn;
};

constexpr int kRayBwdPerSample = 16 * 4 + 12 + 3;     // K, V, q~, do | shift, den, D | t, alpha, u
inline size_t ray_bwd_smem_bytes(int dn) {
    return sizeof(float) * (2 * (kPackedRayFloats + 12) + kRayWaves * ((size_t)dn * kRayBwdPerSample));
}

// acc[o * 16 + k] += sum over the wave of a[o] * b[k]   (acc in LDS, shared by the waves of the workgroup)
__device__ __forceinline__ void wave_outer_add(float* acc, const float (&a)[16], const float (&b)[16], bool act, int lane) {
    for (int o = 0; o < 16; ++o)
        for (int k = 0; k < 16; ++k) {
            const float s = wave_sum(act ? a[o] * b[k] : 0.0f);
            if (lane == 0) atomicAdd(acc + o * 16 + k, s);
        }
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

__global__ void __launch_bounds__(256) rays_backward_kernel(RayBwdParams p) {
    NR_DYNAMIC_SMEM(float, smem);
    const int lane = threadIdx.x & 63;
    const int wave = NR_UNIFORM((int)(threadIdx.x >> 6));
    const int dn = p.dn;                               // <= 64: one sample per lane
    float* RW = smem + nr_opaque_zero();
    float* WA = smem + kPackedRayFloats + 12;          // weight-gradient accumulators of the workgroup
    float* base = smem + 2 * (kPackedRayFloats + 12) + (size_t)wave * (dn * kRayBwdPerSample);
    float* ks = base; float* vs = ks + dn * 16; float* qs = vs + dn * 16; float* dos = qs + dn * 16;
    float* st = dos + dn * 16;                         // [dn][12]: softmax shift (4), denominator (4), D (4)
    float* tr = st + dn * 12; float* al = tr + dn; float* us = al + dn;
    for (int i = threadIdx.x; i < kPackedRayFloats; i += blockDim.x) { RW[i] = p.weights[kPackedPointFloats + i]; WA[i] = 0.0f; }
    __syncthreads();
    const int nray_iter = (p.rn + kRayWaves - 1) / kRayWaves;
    const int iraw = lane;
    const bool inr = iraw < dn;
    const int i = inr ? iraw : dn - 1;                 // lanes past the last sample redo sample dn-1 and contribute nothing

    for (int it = blockIdx.x; it < nray_iter; it += gridDim.x) {
        int ray = it * kRayWaves + wave;
        const bool rvalid = ray < p.rn;
        ray = rvalid ? ray : p.rn - 1;
        const bool act = inr && rvalid;
        const float* rec = p.point_rec + ((size_t)ray * dn + i) * kPointRec;
        asm volatile("" ::: "memory");
        // ---- forward, part 1: G, K, V
        float G[16], kk[16], vv[16];
        NR_PRAGMA_UNROLL
        for (int k4 = 0; k4 < 4; ++k4) {
            const float4 a = ld4(rec + 4 * k4), b = ld4(p.pos_enc + i * 16 + 4 * k4);
            G[4 * k4] = a.x + b.x; G[4 * k4 + 1] = a.y + b.y; G[4 * k4 + 2] = a.z + b.z; G[4 * k4 + 3] = a.w + b.w;
        }
        const float4 c4 = ld4(rec + 16);               // colour (3), number of valid views
        const float nvalid = c4.w;
        matvec16(RW + RW_WK, G, kk);
        matvec16(RW + RW_WV, G, vv);
        if (inr) {
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) { ks[i * 16 + k] = kk[k]; vs[i * 16 + k] = vv[k]; }
        }
        __syncthreads();
        // ---- forward, part 2: attention, LayerNorm, sigma head
        float q[16], o[16], mx[4], den[4];
        matvec16(RW + RW_WQ, G, q);
        const bool qmask = !(nvalid > 1.0f);           // quirk A.9.3: the row's scores are all -1e9 <=> q~ = 0
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 16; ++k) q[k] = qmask ? 0.0f : q[k] / 2.0f;
        NR_PRAGMA_UNROLL
        for (int hh = 0; hh < 4; ++hh) {
            float m_ = -INFINITY;
            for (int j = 0; j < dn; ++j) {
                const float4 kj = ld4(ks + j * 16 + hh * 4);
                m_ = fmaxf(m_, fmaf(q[hh * 4 + 3], kj.w, fmaf(q[hh * 4 + 2], kj.z, fmaf(q[hh * 4 + 1], kj.y, q[hh * 4] * kj.x))));
            }
            float d_ = 0.0f, a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            for (int j = 0; j < dn; ++j) {
                const float4 kj = ld4(ks + j * 16 + hh * 4);
                const float s = fmaf(q[hh * 4 + 3], kj.w, fmaf(q[hh * 4 + 2], kj.z, fmaf(q[hh * 4 + 1], kj.y, q[hh * 4] * kj.x)));
                const float e_ = expf(s - m_);
                const float4 vj = ld4(vs + j * 16 + hh * 4);
                d_ += e_; a0 = fmaf(e_, vj.x, a0); a1 = fmaf(e_, vj.y, a1); a2 = fmaf(e_, vj.z, a2); a3 = fmaf(e_, vj.w, a3);
            }
            mx[hh] = m_; den[hh] = d_;
            o[hh * 4] = a0 / d_; o[hh * 4 + 1] = a1 / d_; o[hh * 4 + 2] = a2 / d_; o[hh * 4 + 3] = a3 / d_;
        }
        float y[16], yh[16], z[16], pre1[16], h1[16], mean = 0.0f, var = 0.0f;
        matvec16(RW + RW_FC, o, y);
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 16; ++k) { y[k] += G[k]; mean += y[k]; }
        mean /= 16.0f;
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 16; ++k) { const float d_ = y[k] - mean; var = fmaf(d_, d_, var); }
        var /= 16.0f;
        const float rstd = 1.0f / sqrtf(var + 1e-6f);
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 16; ++k) { yh[k] = (y[k] - mean) * rstd; z[k] = fmaf(yh[k], RW[RW_LNW + k], RW[RW_LNB + k]); }
        matvec16(RW + RW_OG0W, z, pre1);
        float spre = RW[RW_OG2B];
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 16; ++k) {
            pre1[k] += RW[RW_OG0B + k];
            h1[k] = pre1[k] > 0.0f ? pre1[k] : expf(pre1[k]) - 1.0f;
            spre = fmaf(RW[RW_OG2W + k], h1[k], spre);
        }
        const bool sig_on = (spre > 0.0f) && !(nvalid < 1.0f);
        const float sg = sig_on ? spre : 0.0f;
        const float em = expf(-sg);                    // 1 - alpha
        const float alpha = 1.0f - em;
        const float ti = (1.0f - alpha) + 1e-10f;
        if (inr) { tr[i] = ti; al[i] = alpha; }
        __syncthreads();
        // ---- compositing forward + its backward
        float T = 1.0f;
        for (int j = 0; j < dn; ++j) { const float tj = tr[j]; T = (j < i) ? T * tj : T; }
        const float hit = alpha * T;
        const float gp0 = p.d_pixel[(size_t)ray * 3], gp1 = p.d_pixel[(size_t)ray * 3 + 1], gp2 = p.d_pixel[(size_t)ray * 3 + 2];
        float dhit = gp0 * c4.x + gp1 * c4.y + gp2 * c4.z;
        if (p.d_depth) dhit = fmaf(p.d_depth[ray], p.depth[(size_t)ray * dn + i], dhit);
        if (p.d_hit_prob) dhit += p.d_hit_prob[(size_t)ray * dn + i];
        if (inr) us[i] = dhit * hit;
        __syncthreads();
        float S = 0.0f;
        for (int j = 0; j < dn; ++j) { const float uj = us[j]; S = (j > i) ? S + uj : S; }
        const float dalpha = dhit * T - S / ti;
        const float dsg = sig_on ? dalpha * em : 0.0f;
        // ---- sigma head backward
        float dpre1[16], dz[16], dyh[16], dy[16], dO[16];
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 16; ++k) dpre1[k] = dsg * RW[RW_OG2W + k] * (pre1[k] > 0.0f ? 1.0f : h1[k] + 1.0f);
        {
            float t16[16];
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) t16[k] = dsg * h1[k];
            wave_vec_add(WA + RW_OG2W, t16, act, lane);
            const float sb = wave_sum(act ? dsg : 0.0f);
            if (lane == 0) atomicAdd(WA + RW_OG2B, sb);
            wave_vec_add(WA + RW_OG0B, dpre1, act, lane);
            wave_outer_add(WA + RW_OG0W, dpre1, z, act, lane);
        }
        matvec16_t(RW + RW_OG0W, dpre1, dz);
        // ---- LayerNorm backward
        float m1 = 0.0f, m2 = 0.0f;
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 16; ++k) { dyh[k] = dz[k] * RW[RW_LNW + k]; m1 += dyh[k]; m2 = fmaf(dyh[k], yh[k], m2); }
        m1 /= 16.0f; m2 /= 16.0f;
        NR_PRAGMA_UNROLL
        for (int k = 0; k < 16; ++k) dy[k] = rstd * (dyh[k] - m1 - yh[k] * m2);
        {
            float t16[16];
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) t16[k] = dz[k] * yh[k];
            wave_vec_add(WA + RW_LNW, t16, act, lane);
            wave_vec_add(WA + RW_LNB, dz, act, lane);
            wave_outer_add(WA + RW_FC, dy, o, act, lane);
        }
        matvec16_t(RW + RW_FC, dy, dO);
        // ---- attention backward, query side (this lane = query i)
        float dq[16], Dh[4];
        NR_PRAGMA_UNROLL
        for (int hh = 0; hh < 4; ++hh) {
            Dh[hh] = dO[hh * 4] * o[hh * 4] + dO[hh * 4 + 1] * o[hh * 4 + 1] + dO[hh * 4 + 2] * o[hh * 4 + 2] + dO[hh * 4 + 3] * o[hh * 4 + 3];
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            for (int j = 0; j < dn; ++j) {
                const float4 kj = ld4(ks + j * 16 + hh * 4);
                const float4 vj = ld4(vs + j * 16 + hh * 4);
                const float s = fmaf(q[hh * 4 + 3], kj.w, fmaf(q[hh * 4 + 2], kj.z, fmaf(q[hh * 4 + 1], kj.y, q[hh * 4] * kj.x)));
                const float P = expf(s - mx[hh]) / den[hh];
                const float dP = dO[hh * 4] * vj.x + dO[hh * 4 + 1] * vj.y + dO[hh * 4 + 2] * vj.z + dO[hh * 4 + 3] * vj.w;
                const float dS = P * (dP - Dh[hh]);
                a0 = fmaf(dS, kj.x, a0); a1 = fmaf(dS, kj.y, a1); a2 = fmaf(dS, kj.z, a2); a3 = fmaf(dS, kj.w, a3);
            }
            // q~ = q / 2 (and q~ = 0, without gradient, on masked rows)
            dq[hh * 4] = qmask ? 0.0f : a0 * 0.5f; dq[hh * 4 + 1] = qmask ? 0.0f : a1 * 0.5f;
            dq[hh * 4 + 2] = qmask ? 0.0f : a2 * 0.5f; dq[hh * 4 + 3] = qmask ? 0.0f : a3 * 0.5f;
        }
        if (inr) {
            NR_PRAGMA_UNROLL
            for (int k = 0; k < 16; ++k) { qs[i * 16 + k] = q[k]; dos[i * 16 + k] = act ? dO[k] : 0.0f; }
            NR_PRAGMA_UNROLL
            for (int hh = 0; hh < 4; ++hh) { st[i * 12 + hh] = mx[hh]; st[i * 12 + 4 + hh] = den[hh]; st[i * 12 + 8 + hh] = Dh[hh]; }
        }
        __syncthreads();
        // ---- attention backward, key side (this lane = key i): dk_i, dv_i
        float dk[16], dv[16];
        NR_PRAGMA_UNROLL
        for (int hh = 0; hh < 4; ++hh) {
            float k0 = 0.0f, k1 = 0.0f, k2 = 0.0f, k3 = 0.0f, v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
            for (int j = 0; j < dn; ++j) {               // j = query
                const float4 qj = ld4(qs + j * 16 + hh * 4);
                const float4 dj = ld4(dos + j * 16 + hh * 4);
                const float s = fmaf(qj.w, kk[hh * 4 + 3], fmaf(qj.z, kk[hh * 4 + 2], fmaf(qj.y, kk[hh * 4 + 1], qj.x * kk[hh * 4])));
                const float P = expf(s - st[j * 12 + hh]) / st[j * 12 + 4 + hh];
                const float dP = dj.x * vv[hh * 4] + dj.y * vv[hh * 4 + 1] + dj.z * vv[hh * 4 + 2] + dj.w * vv[hh * 4 + 3];
                const float dS = P * (dP - st[j * 12 + 8 + hh]);
                // rows whose ray is invalid carry do = 0 and D = 0: dS = 0, no contribution
                k0 = fmaf(dS, qj.x, k0); k1 = fmaf(dS, qj.y, k1); k2 = fmaf(dS, qj.z, k2); k3 = fmaf(dS, qj.w, k3);
                v0 = fmaf(P, dj.x, v0); v1 = fmaf(P, dj.y, v1); v2 = fmaf(P, dj.z, v2); v3 = fmaf(P, dj.w, v3);
            }
            dk[hh * 4] = k0; dk[hh * 4 + 1] = k1; dk[hh * 4 + 2] = k2; dk[hh * 4 + 3] = k3;
            dv[hh * 4] = v0; dv[hh * 4 + 1] = v1; dv[hh * 4 + 2] = v2; dv[hh * 4 + 3] = v3;
        }
        wave_outer_add(WA + RW_WQ, dq, G, act, lane);
        wave_outer_add(WA + RW_WK, dk, G, act, lane);
        wave_outer_add(WA + RW_WV, dv, G, act, lane);
        float gq[16], gk[16], gv[16];
        matvec16_t(RW + RW_WQ, dq, gq);
        matvec16_t(RW + RW_WK, dk, gk);
        matvec16_t(RW + RW_WV, dv, gv);
        if (act) {
            float* out = p.d_point_rec + ((size_t)ray * dn + i) * kPointRec;
            NR_PRAGMA_UNROLL
            for (int k4 = 0; k4 < 4; ++k4)
                *reinterpret_cast<float4*>(out + 4 * k4) =
                    make_float4(dy[4 * k4] + gq[4 * k4] + gk[4 * k4] + gv[4 * k4], dy[4 * k4 + 1] + gq[4 * k4 + 1] + gk[4 * k4 + 1] + gv[4 * k4 + 1],
                                dy[4 * k4 + 2] + gq[4 * k4 + 2] + gk[4 * k4 + 2] + gv[4 * k4 + 2], dy[4 * k4 + 3] + gq[4 * k4 + 3] + gk[4 * k4 + 3] + gv[4 * k4 + 3]);
            *reinterpret_cast<float4*>(out + 16) = make_float4(hit * gp0, hit * gp1, hit * gp2, 0.0f);
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < kPackedRayFloats; k += blockDim.x) atomicAdd(p.d_weights + k, WA[k]);
}

}  // namespace nr
