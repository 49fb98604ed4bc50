// Direct rendering (cfg['use_dr_prediction'], network/renderer.py:85-125 + network/sph_solver.py:1-59): the second,
// network-free estimate of a ray's colour that the reference can emit next to the aggregation network's -
//   alpha_dr(point)  = sum_v vis_v alpha_v / (sum_v vis_v + 1e-5)           (alpha_v: the dist decoder's logit, -15 where masked)
//   colors_dr(point) = SH_16(que_dir) . theta,   theta = (A^T W A + diag(regs))^-1 A^T W C   (weighted degree-3 spherical
//                      harmonics fit of the views' colours C over their viewing directions, W = hit_v / (sum hit + 1e-3))
//   hit_prob_dr      = alpha_values2hit_prob(sigmoid(alpha_dr)),  pixel_colors_dr = sum_i hit_i colors_i.
// Off in every shipped config, so this is built for exactness, not for the roofline: one thread per sample point, the
// 16 x 16 normal matrix packed symmetric in registers (136 + 48 right-hand sides), an unrolled LDL^T elimination (the
// matrix is symmetric positive definite: sum w > 0 on the constant column, regs > 0 elsewhere) where the reference calls
// torch.inverse.  The per-(point, view) hit / vis come from the point kernel's per-view record (NeurayPointsArgs.dbg_dev,
// fields 4 / 5), geometry and colours are recomputed with the exact (reference-order) device functions.
#pragma once
#include "nr_device.h"

namespace nr {

// real spherical-harmonics polynomials of sph_solver.py:14-31 up to degree 3, in the reference's operation order
__device__ __forceinline__ void sh16(float x, float y, float z, float (&a)[16]) {
    const float xx = rn_mul(x, x), yy = rn_mul(y, y), zz = rn_mul(z, z);
    a[0] = 1.0f;
    a[1] = x; a[2] = y; a[3] = z;
    a[4] = rn_mul(x, y);
    a[5] = rn_mul(y, z);
    a[6] = rn_add(rn_sub(-xx, yy), rn_mul(2.0f, zz));
    a[7] = rn_mul(z, x);
    a[8] = rn_sub(xx, yy);
    a[9] = rn_mul(rn_sub(rn_mul(3.0f, xx), yy), y);
    a[10] = rn_mul(rn_mul(x, y), z);
    a[11] = rn_mul(y, rn_sub(rn_sub(rn_mul(4.0f, zz), xx), yy));
    a[12] = rn_mul(z, rn_sub(rn_sub(rn_mul(2.0f, zz), rn_mul(3.0f, xx)), rn_mul(3.0f, yy)));
    a[13] = rn_mul(x, rn_sub(rn_sub(rn_mul(4.0f, zz), xx), yy));
    a[14] = rn_mul(rn_sub(xx, yy), z);
    a[15] = rn_mul(rn_sub(xx, rn_mul(3.0f, yy)), x);
}

__device__ __forceinline__ constexpr int sym(int i, int j) { return i <= j ? i * 16 - i * (i - 1) / 2 + (j - i) : j * 16 - j * (j - 1) / 2 + (i - j); }

template <int K>
__device__ __forceinline__ void dr_eliminate(float (&M)[136], float (&R)[16][3]) {
    if constexpr (K < 16) {
        const float inv = 1.0f / M[sym(K, K)];
        NR_PRAGMA_UNROLL
        for (int i = K + 1; i < 16; ++i) {
            const float f = M[sym(K, i)] * inv;
            NR_PRAGMA_UNROLL
            for (int j = 0; j < 16; ++j)
                if (j >= i) M[sym(i, j)] = fmaf(-f, M[sym(K, j)], M[sym(i, j)]);
            NR_PRAGMA_UNROLL
            for (int c = 0; c < 3; ++c) R[i][c] = fmaf(-f, R[K][c], R[i][c]);
        }
        dr_eliminate<K + 1>(M, R);
    }
}
template <int K>
__device__ __forceinline__ void dr_back_substitute(const float (&M)[136], float (&R)[16][3]) {          // on the upper triangle left in M
    if constexpr (K >= 0) {
        const float inv = 1.0f / M[sym(K, K)];
        NR_PRAGMA_UNROLL
        for (int c = 0; c < 3; ++c) {
            float s = R[K][c];
            NR_PRAGMA_UNROLL
            for (int j = K + 1; j < 16; ++j) s = fmaf(-M[sym(K, j)], R[j][c], s);
            R[K][c] = s * inv;
        }
        dr_back_substitute<K - 1>(M, R);
    }
}

// per (point): alpha logit + SH colour.  view_rec [npts][rfn][kDbgFields] (the point kernel's per-view record),
// regs [16] (SphericalHarmonicsSolver.regs), alpha_out [npts], color_out [npts][3] (null: use_nr_color_for_dr).
#ifndef NR_DR_MINW
#define NR_DR_MINW 1
#endif
__global__ void __launch_bounds__(128, NR_DR_MINW) dr_points_kernel(const float* __restrict__ qc, const float* __restrict__ view_const,
                                                        const float* __restrict__ coords, const float* __restrict__ depth,
                                                        const float* __restrict__ rgba, const float* __restrict__ view_rec,
                                                        const float* __restrict__ regs, int rfn, int rn, int dn, int h, int w,
                                                        float ground, float* __restrict__ alpha_out, float* __restrict__ color_out) {
    const long long npts = (long long)rn * dn;
    const size_t imap = (size_t)h * w * 4;
    for (long long pi = (long long)blockIdx.x * blockDim.x + threadIdx.x; pi < npts; pi += (long long)gridDim.x * blockDim.x) {
        const int ray = (int)(pi / dn);
        const Ray r = make_ray<true>(qc, coords[2 * ray], coords[2 * ray + 1]);
        const float d = depth[pi];
        const float px = rn_add(r.cx, rn_mul(r.dx, d)), py = rn_add(r.cy, rn_mul(r.dy, d)), pz = rn_add(r.cz, rn_mul(r.dz, d));
        // ---- alpha (renderer.py:85-95) and the fit's weights (:103) --------------------------------------------------
        float s_va = 0.0f, s_v = 0.0f, s_hit = 0.0f;
        int n_valid = 0;
        for (int v = 0; v < rfn; ++v) {
            const float* rec = view_rec + ((size_t)pi * rfn + v) * kDbgFields;
            const float m = rec[0], hit = rec[4], vis = rec[5];     // hit, vis already masked (renderer.py:81-82)
            // compute_prob's logit (dist_decoder.py:137-138) on the un-masked values; where the mask is 0 it is replaced anyway
            const float logit = logf(rn_add(rn_div(hit, rn_add(rn_sub(vis, hit), 1e-5f)), 1e-5f));
            const float alpha = rn_add(rn_mul(logit, m), rn_mul(rn_sub(1.0f, m), ground));
            s_va = rn_add(s_va, rn_mul(vis, m > 0.0f ? alpha : ground));
            s_v = rn_add(s_v, vis);
            s_hit = rn_add(s_hit, hit);
            n_valid += m > 0.0f ? 1 : 0;
        }
        const float a_dr = rn_div(s_va, rn_add(s_v, 1e-5f));
        alpha_out[pi] = n_valid == 0 ? ground : a_dr;
        if (!color_out) continue;
        // ---- weighted SH least squares (sph_solver.py:33-50) ------------------------------------------------------------
        const float inv_hit = rn_add(s_hit, 1e-3f);
        float s_w = 0.0f;
        for (int v = 0; v < rfn; ++v) s_w = rn_add(s_w, rn_div(view_rec[((size_t)pi * rfn + v) * kDbgFields + 4], inv_hit));
        const float w_eps = s_w < 1e-4f ? 1e-4f : 0.0f;               // "insufficient" rays get a uniform floor
        float M[136], R[16][3];
        NR_PRAGMA_UNROLL
        for (int i = 0; i < 136; ++i) M[i] = 0.0f;
        NR_PRAGMA_UNROLL
        for (int i = 0; i < 16; ++i) { R[i][0] = 0.0f; R[i][1] = 0.0f; R[i][2] = 0.0f; }
        for (int v = 0; v < rfn; ++v) {
            const float* vc = view_const + v * kViewConst;
            const Proj pr = project_point<true>(vc, px, py, pz, (float)w, (float)h);
            const float wv = rn_add(rn_div(view_rec[((size_t)pi * rfn + v) * kDbgFields + 4], inv_hit), w_eps);
            float rgb[3];
            {
                const Taps t = make_taps(pr.u, pr.v, w, h, w, h);
                const float* mp = rgba + (size_t)v * imap;
                const float4 c00 = ld4(mp + (size_t)t.o00 * 4), c10 = ld4(mp + (size_t)t.o10 * 4);
                const float4 c01 = ld4(mp + (size_t)t.o01 * 4), c11 = ld4(mp + (size_t)t.o11 * 4);
                rgb[0] = blend4(c00.x, c10.x, c01.x, c11.x, t) * pr.mask;
                rgb[1] = blend4(c00.y, c10.y, c01.y, c11.y, t) * pr.mask;
                rgb[2] = blend4(c00.z, c10.z, c01.z, c11.z, t) * pr.mask;
            }
            float a[16], aw[16];
            sh16(pr.dirx, pr.diry, pr.dirz, a);
            NR_PRAGMA_UNROLL
            for (int i = 0; i < 16; ++i) aw[i] = rn_mul(a[i], wv);
            NR_PRAGMA_UNROLL
            for (int i = 0; i < 16; ++i) {
                NR_PRAGMA_UNROLL
                for (int j = 0; j < 16; ++j)          // (constant trip count + a predicate that folds: a bound that depends on i keeps the
                    if (j >= i) M[sym(i, j)] = fmaf(aw[i], a[j], M[sym(i, j)]);      //  inner loop from being unrolled before the outer one is)
                NR_PRAGMA_UNROLL
                for (int c = 0; c < 3; ++c) R[i][c] = fmaf(aw[i], rgb[c], R[i][c]);
            }
        }
        NR_PRAGMA_UNROLL
        for (int i = 0; i < 16; ++i) M[sym(i, i)] += regs[i];
        // LDL^T elimination of the SPD system (no pivoting needed), three right-hand sides.  One function instantiation per pivot: with the
        // pivot a run-time loop variable hipcc stopped unrolling part-way, indexed M / R dynamically and put both arrays into scratch
        // memory (752 B per lane, every access a memory round trip at one wave per SIMD: 2.99 ms per launch of 1.6 M points, round 4)
        dr_eliminate<0>(M, R);
        dr_back_substitute<15>(M, R);
        float q[16];
        sh16(r.qx, r.qy, r.qz, q);
        NR_PRAGMA_UNROLL
        for (int c = 0; c < 3; ++c) {
            float s = 0.0f;
            NR_PRAGMA_UNROLL
            for (int i = 0; i < 16; ++i) s = fmaf(q[i], R[i][c], s);
            color_out[pi * 3 + c] = s;
        }
    }
}

// per ray: decode_alpha_value (sigmoid, dist_decoder.py:142-144), alpha_values2hit_prob (render_ops.py:72-80, sequential
// transmittance product as the ray kernel's), pixel colour.  colors [rn*dn][stride] starting at `first` (the SH colours:
// stride 3, first 0; use_nr_color_for_dr: the point records' blended colour, stride kPointRec, first 16).
__global__ void dr_rays_kernel(const float* __restrict__ alpha, const float* __restrict__ colors, int stride, int first, int rn, int dn,
                               float* __restrict__ hit_out, float* __restrict__ pixel_out) {
    for (int ray = blockIdx.x * blockDim.x + threadIdx.x; ray < rn; ray += gridDim.x * blockDim.x) {
        float T = 1.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
        for (int i = 0; i < dn; ++i) {
            const size_t pi = (size_t)ray * dn + i;
            const float a = 1.0f / (1.0f + expf(-alpha[pi]));
            const float hit = a * T;
            T = T * ((1.0f - a) + 1e-10f);
            hit_out[pi] = hit;
            const float* c = colors + pi * stride + first;
            c0 = rn_add(c0, rn_mul(hit, c[0])); c1 = rn_add(c1, rn_mul(hit, c[1])); c2 = rn_add(c2, rn_mul(hit, c[2]));
        }
        pixel_out[ray * 3 + 0] = c0; pixel_out[ray * 3 + 1] = c1; pixel_out[ray * 3 + 2] = c2;
    }
}

}  // namespace nr
