// Device functions of the NeuRay per-ray path for gfx950.  Each block cites the reference lines it
// replaces (paths relative to the reference tree) and follows the rounding contract of DESIGN.md:
// camera algebra uses explicit rn_mul/rn_add sequences (no FMA contraction) in the same order as
// the oracle, so validity masks and texel indices are bit-identical between the two.
#pragma once
#include "nr_platform.h"
#include "nr_layout.h"

namespace nr {

// ---------------------------------------------------------------------------------------------
// ordered small algebra
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
    return rn_add(rn_add(rn_mul(a0, b0), rn_mul(a1, b1)), rn_mul(a2, b2));
}

// ---------------------------------------------------------------------------------------------
// activations (PyTorch semantics: ELU alpha=1 via exp(x)-1, Softplus beta=1 threshold=20)
// ---------------------------------------------------------------------------------------------
// Built on the hardware exp2/log2/rcp (about 1 ulp each): absolute error ~1e-7 per activation, two orders of
// magnitude inside the fp32 parity tolerance (tests/test_render_parity.py), at 3-4 VALU instructions per ELU instead
// of ~18 for the libm-exact expf (the precise forms made the point kernel VALU-bound: 10.8 VALU per MFMA).
// ELU(x) = median(x, exp(x) - 1, 0): for x > 0 the order is 0 < x <= exp(x)-1, for x < 0 it is x < exp(x)-1 < 0, so one
// v_med3_f32 replaces the compare + select (4 instead of 5 VALU per activation; +inf from exp overflow is harmless)
__device__ __forceinline__ float elu(float x) { return nr_med3(x, nr_fast_exp(x) - 1.0f, 0.0f); }
// scaled form (nr_layout.h kOutScaled): argument and result carry the factor L = log2(e).  The same ordering argument
// holds: y' > 0: 0 < y' <= L(2^y' - 1); y' < 0: y' <= L(2^y' - 1) < 0.
__device__ __forceinline__ float elu_s(float x) {
    return nr_med3(x, fmaf(nr_fast_exp2(x), (float)kLog2e, -(float)kLog2e), 0.0f);
}
__device__ __forceinline__ float softplus(float x) { return x > 20.0f ? x : nr_fast_log(1.0f + nr_fast_exp(x)); }
__device__ __forceinline__ float sigmoidf(float x) { return nr_fast_rcp(1.0f + nr_fast_exp(-x)); }
__device__ __forceinline__ float tanh_(float x) {   // 1 - 2/(exp(2x)+1); saturates cleanly to +-1
    return 1.0f - 2.0f * nr_fast_rcp(nr_fast_exp(2.0f * x) + 1.0f);
}

// ---------------------------------------------------------------------------------------------
// a1  coarse depth sample i of dn, uniform in inverse depth       network/render_ops.py:146-170
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float coarse_depth(float near, float far, int i, int dn) {
    const float inv_near = rn_div(1.0f, near);
    const float diff = rn_sub(rn_div(1.0f, far), inv_near);
    float tick;
    if (i == 0) tick = 0.0f;
    else if (i == dn - 1) tick = diff;
    else tick = rn_mul(rn_div(diff, (float)(dn - 1)), (float)i);
    return rn_div(1.0f, rn_add(inv_near, tick));
}

// normalised inverse depth s = (-1/d - near') / (far' - near'),  near' = -1/near, far' = -1/far
//   network/render_ops.py:46-52, dist_decoder.py:16-22
__device__ __forceinline__ float norm_inv_depth(float d, float nearp, float farp) {
    return rn_div(rn_sub(rn_div(-1.0f, d), nearp), rn_sub(farp, nearp));
}

// Feature-path divisions (they decide no mask and no index).  NR_FEATURE_RCP_REFINE = 1 (default): the hardware reciprocal
// (v_rcp_f32, 1 ulp) gets one Newton step and a quotient p / d computed as p * RN(1/d) gets one residual correction
// q' = q + r (p - d q) - two FMAs each, which makes both agree with the correctly rounded division of the reference's op
// sequence in all but rare halfway cases.  It matters for the texel coordinates: 1 ulp of u / (W - 1) is 1.2e-5 texels on a
// 200-wide map, i.e. ~2e-5 on a gathered feature of a white-noise map, which the MLP stack carries to the pixel
// (worst coarse error 1.6e-4 against the 2e-4 gate).  = 0: the round-2 forms, for A/B timing.
#ifndef NR_FEATURE_RCP_REFINE
#define NR_FEATURE_RCP_REFINE 1
#endif
#ifndef NR_FEATURE_RCP_NEWTON          // the Newton step of 1 / d alone (A/B: the quotient correction is what the texel coordinates need)
#define NR_FEATURE_RCP_NEWTON NR_FEATURE_RCP_REFINE
#endif
__device__ __forceinline__ float nr_rcp_refined(float d) {
    const float r = nr_fast_rcp(d);
#if NR_FEATURE_RCP_NEWTON
    return fmaf(fmaf(-d, r, 1.0f), r, r);
#else
    return r;
#endif
}
// p / d given r = RN(1 / d)
__device__ __forceinline__ float nr_div_refined(float p, float d, float r) {
    const float q = p * r;
#if NR_FEATURE_RCP_REFINE
    return fmaf(fmaf(-d, q, p), r, q);
#else
    return q;
#endif
}
// feature-path normalised inverse depth: (-1/d - near') / (far' - near') with inv_range = RN(1 / (far' - near'))
__device__ __forceinline__ float norm_inv_depth_fast(float d, float nearp, float farp, float inv_range) {
    return nr_div_refined(-nr_rcp_refined(d) - nearp, farp - nearp, inv_range);
}

// ---------------------------------------------------------------------------------------------
// a2  query ray                                                   network/render_ops.py:4-39
//   qc = query constants: [0..8] K^-1, [9..20] pose, [21..23] centre
// ---------------------------------------------------------------------------------------------
struct Ray { float cx, cy, cz, dx, dy, dz, qx, qy, qz; };   // centre, un-normalised dir, que_dir = -dir/|dir|

template <bool EXACT = true>
__device__ __forceinline__ Ray make_ray(const float* __restrict__ qc, float x, float y) {
    Ray r;
    const float cam0 = dot3(qc[0], qc[1], qc[2], x, y, 1.0f);
    const float cam1 = dot3(qc[3], qc[4], qc[5], x, y, 1.0f);
    const float cam2 = dot3(qc[6], qc[7], qc[8], x, y, 1.0f);
    const float* P = qc + 9;   // pose row-major 3x4; rot = R^T -> row i of rot is column i of R
    r.cx = qc[21]; r.cy = qc[22]; r.cz = qc[23];
    const float w0 = dot3(P[0], P[4], P[8], cam0, cam1, cam2);
    const float w1 = dot3(P[1], P[5], P[9], cam0, cam1, cam2);
    const float w2 = dot3(P[2], P[6], P[10], cam0, cam1, cam2);
    r.dx = rn_sub(rn_add(w0, r.cx), r.cx);
    r.dy = rn_sub(rn_add(w1, r.cy), r.cy);
    r.dz = rn_sub(rn_add(w2, r.cz), r.cz);
    const float nrm = rn_sqrt(rn_add(rn_add(rn_mul(r.dx, r.dx), rn_mul(r.dy, r.dy)), rn_mul(r.dz, r.dz)));
    if (EXACT) { r.qx = rn_div(-r.dx, nrm); r.qy = rn_div(-r.dy, nrm); r.qz = rn_div(-r.dz, nrm); }
    else { const float inv = nr_rcp_refined(nrm); r.qx = -r.dx * inv; r.qy = -r.dy * inv; r.qz = -r.dz * inv; }   // direction feature only
    return r;
}

// ---------------------------------------------------------------------------------------------
// a4-a6  projection into one reference view                        network/render_ops.py:82-130
//   vc = view constants: [0..11] H = K[R|t], [12..14] centre
// ---------------------------------------------------------------------------------------------
struct Proj { float u, v, z, mask, dirx, diry, dirz; };

template <bool EXACT = true>
__device__ __forceinline__ Proj project_point(const float* __restrict__ vc, float px, float py, float pz, float w_img, float h_img) {
    Proj o;
    const float c0 = rn_add(rn_add(rn_add(rn_mul(vc[0], px), rn_mul(vc[1], py)), rn_mul(vc[2], pz)), vc[3]);
    const float c1 = rn_add(rn_add(rn_add(rn_mul(vc[4], px), rn_mul(vc[5], py)), rn_mul(vc[6], pz)), vc[7]);
    float z = rn_add(rn_add(rn_add(rn_mul(vc[8], px), rn_mul(vc[9], py)), rn_mul(vc[10], pz)), vc[11]);
    const bool bad = fabsf(z) < 1e-4f;          // no z>0 test: quirk A.9.1
    if (bad) z = 1e-3f;
    o.u = rn_div(c0, z); o.v = rn_div(c1, z); o.z = z;
    // w_img <= 0: no image-bounds test (stand-alone project_points_coords, render_ops.py:82-104)
    const bool outside = (w_img > 0.0f) && ((o.u < -0.5f) | (o.u >= w_img - 0.5f) | (o.v < -0.5f) | (o.v >= h_img - 0.5f));
    o.mask = (!bad && !outside) ? 1.0f : 0.0f;
    const float dx = rn_sub(px, vc[12]), dy = rn_sub(py, vc[13]), dz = rn_sub(pz, vc[14]);
    const float nrm = rn_sqrt(rn_add(rn_add(rn_mul(dx, dx), rn_mul(dy, dy)), rn_mul(dz, dz)));
    const float den = fmaxf(nrm, 1e-5f);
    if (EXACT) { o.dirx = rn_div(-dx, den); o.diry = rn_div(-dy, den); o.dirz = rn_div(-dz, den); }
    else { const float inv = nr_rcp_refined(den); o.dirx = -dx * inv; o.diry = -dy * inv; o.dirz = -dz * inv; }   // feature only
    return o;
}

// ---------------------------------------------------------------------------------------------
// a7  bilinear taps (grid_sample, padding_mode='border')        network/ops.py:14-34
//   pixel coordinate p (full-res units) -> clamped texel coordinate; align_corners=True for maps at
//   full resolution (rgb), False otherwise (render_ops.py:64-68).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float texel_coord(float p, float size_full, float size_map, bool align) {
    const float n = rn_sub(rn_mul(rn_div(p, rn_sub(size_full, 1.0f)), 2.0f), 1.0f);
    float ix;
    if (align) ix = rn_mul(rn_div(rn_add(n, 1.0f), 2.0f), rn_sub(size_map, 1.0f));
    else ix = rn_div(rn_sub(rn_mul(rn_add(n, 1.0f), size_map), 1.0f), 2.0f);
    return fminf(rn_sub(size_map, 1.0f), fmaxf(ix, 0.0f));
}

struct Taps { int o00, o10, o01, o11; float w00, w10, w01, w11; };   // texel offsets (y*W + x) and weights

// feature-path texel coordinate: p * 1/(size_full-1) instead of the correctly rounded division (<= 1 ulp apart; the
// bilinear result is continuous in the coordinate, validity masks never depend on it)
__device__ __forceinline__ float texel_coord_fast(float p, float full_m1, float inv_full_m1, float size_map, bool align) {
    const float n = nr_div_refined(p, full_m1, inv_full_m1) * 2.0f - 1.0f;
    const float ix = align ? (n + 1.0f) * 0.5f * (size_map - 1.0f) : ((n + 1.0f) * size_map - 1.0f) * 0.5f;
    return fminf(size_map - 1.0f, fmaxf(ix, 0.0f));
}

__device__ __forceinline__ Taps taps_from(float ix, float iy, int mw, int mh) {
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = x0 + 1 < mw ? x0 + 1 : mw - 1;
    const int yb = y0 + 1 < mh ? y0 + 1 : mh - 1;
    const float wx1 = ix - x0f, wy1 = iy - y0f;
    const float wx0 = (x0f + 1.0f) - ix, wy0 = (y0f + 1.0f) - iy;
    Taps t;
    t.o00 = y0 * mw + x0; t.o10 = y0 * mw + x1; t.o01 = yb * mw + x0; t.o11 = yb * mw + x1;
    t.w00 = wx0 * wy0; t.w10 = wx1 * wy0; t.w01 = wx0 * wy1; t.w11 = wx1 * wy1;
    if (x0 + 1 > mw - 1) { t.w10 = 0.0f; t.w11 = 0.0f; }
    if (y0 + 1 > mh - 1) { t.w01 = 0.0f; t.w11 = 0.0f; }
    return t;
}

__device__ __forceinline__ Taps make_taps_fast(float u, float v, float w_m1, float h_m1, float inv_w_m1, float inv_h_m1, int mw, int mh,
                                               bool align) {
    return taps_from(texel_coord_fast(u, w_m1, inv_w_m1, (float)mw, align), texel_coord_fast(v, h_m1, inv_h_m1, (float)mh, align), mw, mh);
}

__device__ __forceinline__ Taps make_taps(float u, float v, int w_full, int h_full, int mw, int mh) {
    const bool align = (mw == w_full) && (mh == h_full);
    const float ix = texel_coord(u, (float)w_full, (float)mw, align);
    const float iy = texel_coord(v, (float)h_full, (float)mh, align);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = x0 + 1 < mw ? x0 + 1 : mw - 1;   // weight is exactly 0 when clamped
    const int yb = y0 + 1 < mh ? y0 + 1 : mh - 1;
    const float wx1 = rn_sub(ix, x0f), wy1 = rn_sub(iy, y0f);
    const float wx0 = rn_sub(rn_add(x0f, 1.0f), ix), wy0 = rn_sub(rn_add(y0f, 1.0f), iy);
    Taps t;
    t.o00 = y0 * mw + x0; t.o10 = y0 * mw + x1; t.o01 = yb * mw + x0; t.o11 = yb * mw + x1;
    t.w00 = rn_mul(wx0, wy0); t.w10 = rn_mul(wx1, wy0); t.w01 = rn_mul(wx0, wy1); t.w11 = rn_mul(wx1, wy1);
    if (x0 + 1 > mw - 1) { t.w10 = 0.0f; t.w11 = 0.0f; }
    if (y0 + 1 > mh - 1) { t.w01 = 0.0f; t.w11 = 0.0f; }
    return t;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// weight-fragment / map loads
__device__ __forceinline__ float4 wld4(nr_wbuf W, int voff, int soff) { return nr_buf_ld4(W, voff, soff); }
__device__ __forceinline__ float wld1(nr_wbuf W, int voff, int soff) { return nr_buf_ld1(W, voff, soff); }
// a staged phase of the packed weights in LDS: same byte offsets as the global buffer, rebased to the phase start
struct LdsW { const float* base; int begin_bytes; };
__device__ __forceinline__ float4 wld4(LdsW w, int voff, int soff) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(w.base) + (soff - w.begin_bytes) + voff);
}
__device__ __forceinline__ float2 wld2(nr_wbuf W, int voff, int soff) { return nr_buf_ld2(W, voff, soff); }
__device__ __forceinline__ float2 wld2(LdsW w, int voff, int soff) {
    return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(w.base) + (soff - w.begin_bytes) + voff);
}
__device__ __forceinline__ float wld1(LdsW w, int voff, int soff) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(w.base) + (soff - w.begin_bytes) + voff);
}
// ---- AR_X3 weight sources (nr_layout.h): the split-operand pack in global memory / a staged phase of it in LDS.  Distinct types, so
// that every layer function picks the layout (offsets, fragment format) from the source it is handed: ws_ar<WS>::value.
struct GlbW3 { nr_buf b; };
struct LdsW3 { const float* base; int begin_bytes; };
template <class WS> struct ws_ar { static constexpr int value = AR_F32; };
template <> struct ws_ar<GlbW3> { static constexpr int value = AR_X3; };
template <> struct ws_ar<LdsW3> { static constexpr int value = AR_X3; };
template <class WS> struct ws_lds { typedef LdsW type; };
template <> struct ws_lds<GlbW3> { typedef LdsW3 type; };
__device__ __forceinline__ nr_buf ws_raw(nr_wbuf W) { return W; }
__device__ __forceinline__ nr_buf ws_raw(GlbW3 W) { return W.b; }
__device__ __forceinline__ float4 wld4(GlbW3 W, int voff, int soff) { return nr_buf_ld4(W.b, voff, soff); }
__device__ __forceinline__ float wld1(GlbW3 W, int voff, int soff) { return nr_buf_ld1(W.b, voff, soff); }
__device__ __forceinline__ nr_v4u wld4u(GlbW3 W, int voff, int soff) { return nr_buf_ld4u(W.b, voff, soff); }
__device__ __forceinline__ nr_v2u wld2u(GlbW3 W, int voff, int soff) { return nr_buf_ld2u(W.b, voff, soff); }
__device__ __forceinline__ float4 wld4(LdsW3 w, int voff, int soff) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(w.base) + (soff - w.begin_bytes) + voff);
}
__device__ __forceinline__ float wld1(LdsW3 w, int voff, int soff) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(w.base) + (soff - w.begin_bytes) + voff);
}
__device__ __forceinline__ nr_v4u wld4u(LdsW3 w, int voff, int soff) {
    return *reinterpret_cast<const nr_v4u*>(reinterpret_cast<const char*>(w.base) + (soff - w.begin_bytes) + voff);
}
__device__ __forceinline__ nr_v2u wld2u(LdsW3 w, int voff, int soff) {
    return *reinterpret_cast<const nr_v2u*>(reinterpret_cast<const char*>(w.base) + (soff - w.begin_bytes) + voff);
}
// one quad fragment (nr_layout.h): 16 bytes per lane; in the bf16-operand build only the first 8 carry data
template <class WS> __device__ __forceinline__ float4 wldq(WS W, int voff, int soff) {
#if defined(NR_BF16_QUADS) && !defined(NR_BF16_SPLIT)
    const float2 h = wld2(W, voff, soff);
    return make_float4(h.x, h.y, 0.0f, 0.0f);
#else
    return wld4(W, voff, soff);
#endif
}
__device__ __forceinline__ float4 mld4(nr_mbuf M, int voff, int soff) { return nr_buf_ld4(M, voff, soff); }

__device__ __forceinline__ float blend4(float a, float b, float c, float d, const Taps& t) {
    return fmaf(d, t.w11, fmaf(c, t.w01, fmaf(b, t.w10, a * t.w00)));
}

// channels-last gather of 8 consecutive channels (this lane group's slice, byte offset goff = 32*g) of a
// 32-channel map: two 16-byte loads per tap, a texel's 128-byte line is covered by the 4 lane groups
__device__ __forceinline__ void gather8(nr_mbuf map, int goff, int soff, const Taps& t, float mask, float (&out)[8]) {
    const float4 a0 = mld4(map, t.o00 * 128 + goff, soff), a1 = mld4(map, t.o00 * 128 + goff, soff + 16);
    const float4 b0 = mld4(map, t.o10 * 128 + goff, soff), b1 = mld4(map, t.o10 * 128 + goff, soff + 16);
    const float4 c0 = mld4(map, t.o01 * 128 + goff, soff), c1 = mld4(map, t.o01 * 128 + goff, soff + 16);
    const float4 d0 = mld4(map, t.o11 * 128 + goff, soff), d1 = mld4(map, t.o11 * 128 + goff, soff + 16);
    out[0] = blend4(a0.x, b0.x, c0.x, d0.x, t) * mask; out[1] = blend4(a0.y, b0.y, c0.y, d0.y, t) * mask;
    out[2] = blend4(a0.z, b0.z, c0.z, d0.z, t) * mask; out[3] = blend4(a0.w, b0.w, c0.w, d0.w, t) * mask;
    out[4] = blend4(a1.x, b1.x, c1.x, d1.x, t) * mask; out[5] = blend4(a1.y, b1.y, c1.y, d1.y, t) * mask;
    out[6] = blend4(a1.z, b1.z, c1.z, d1.z, t) * mask; out[7] = blend4(a1.w, b1.w, c1.w, d1.w, t) * mask;
}

// split form: tap loads are issued ahead of their blends so that many are in flight together (hipcc otherwise
// serialises load -> wait -> blend per map and slot to save registers: ~6 dependent round trips per tile)
__device__ __forceinline__ void issue8(nr_mbuf m, int goff, int soff, const Taps& t, float4 (&q)[8]) {
    q[0] = mld4(m, t.o00 * 128 + goff, soff); q[1] = mld4(m, t.o00 * 128 + goff, soff + 16);
    q[2] = mld4(m, t.o10 * 128 + goff, soff); q[3] = mld4(m, t.o10 * 128 + goff, soff + 16);
    q[4] = mld4(m, t.o01 * 128 + goff, soff); q[5] = mld4(m, t.o01 * 128 + goff, soff + 16);
    q[6] = mld4(m, t.o11 * 128 + goff, soff); q[7] = mld4(m, t.o11 * 128 + goff, soff + 16);
}

__device__ __forceinline__ void issue_rgb(nr_mbuf m, int soff, const Taps& t, float4 (&q)[4]) {
    q[0] = mld4(m, t.o00 * 16, soff); q[1] = mld4(m, t.o10 * 16, soff);
    q[2] = mld4(m, t.o01 * 16, soff); q[3] = mld4(m, t.o11 * 16, soff);
}

// bilinear blend of channel PAIRS (v_pk_mul_f32 / v_pk_fma_f32: two channels per instruction); the validity mask is folded
// into the four weights (mask is 0 or 1, so w * mask is exact and the chain equals blend4(...) * mask)
struct TapW2 { nr_v2 w00, w10, w01, w11; };
__device__ __forceinline__ TapW2 tap_weights2(const Taps& t, float mask) {
    const float a = t.w00 * mask, b = t.w10 * mask, c = t.w01 * mask, d = t.w11 * mask;
    TapW2 w; w.w00 = nr_v2_make(a, a); w.w10 = nr_v2_make(b, b); w.w01 = nr_v2_make(c, c); w.w11 = nr_v2_make(d, d);
    return w;
}
__device__ __forceinline__ nr_v2 blend4_2(nr_v2 a, nr_v2 b, nr_v2 c, nr_v2 d, const TapW2& w) {
    return nr_v2_fma(d, w.w11, nr_v2_fma(c, w.w01, nr_v2_fma(b, w.w10, nr_v2_mul(a, w.w00))));
}
__device__ __forceinline__ void blend8(const float4 (&q)[8], const Taps& t, float mask, float (&out)[8]) {
    const TapW2 w = tap_weights2(t, mask);
    NR_PRAGMA_UNROLL
    for (int h = 0; h < 2; ++h) {            // q[h], q[2+h], q[4+h], q[6+h]: the four taps of channels 4h .. 4h+3
        const nr_v2 lo = blend4_2(nr_v2_make(q[h].x, q[h].y), nr_v2_make(q[2 + h].x, q[2 + h].y), nr_v2_make(q[4 + h].x, q[4 + h].y),
                                  nr_v2_make(q[6 + h].x, q[6 + h].y), w);
        const nr_v2 hi = blend4_2(nr_v2_make(q[h].z, q[h].w), nr_v2_make(q[2 + h].z, q[2 + h].w), nr_v2_make(q[4 + h].z, q[4 + h].w),
                                  nr_v2_make(q[6 + h].z, q[6 + h].w), w);
        out[4 * h] = lo.x; out[4 * h + 1] = lo.y; out[4 * h + 2] = hi.x; out[4 * h + 3] = hi.y;
    }
}
__device__ __forceinline__ void blend_rgb(const float4 (&c)[4], const Taps& t, float mask, float (&rgb)[3]) {
    const TapW2 w = tap_weights2(t, mask);
    const nr_v2 rg = blend4_2(nr_v2_make(c[0].x, c[0].y), nr_v2_make(c[1].x, c[1].y), nr_v2_make(c[2].x, c[2].y), nr_v2_make(c[3].x, c[3].y), w);
    const nr_v2 bb = blend4_2(nr_v2_make(c[0].z, c[0].w), nr_v2_make(c[1].z, c[1].w), nr_v2_make(c[2].z, c[2].w), nr_v2_make(c[3].z, c[3].w), w);
    rgb[0] = rg.x; rgb[1] = rg.y; rgb[2] = bb.x;
}

__device__ __forceinline__ void gather_rgb(nr_mbuf map, int soff, const Taps& t, float mask, float (&out)[3]) {
    const float4 a = mld4(map, t.o00 * 16, soff), b = mld4(map, t.o10 * 16, soff);
    const float4 c = mld4(map, t.o01 * 16, soff), d = mld4(map, t.o11 * 16, soff);
    out[0] = blend4(a.x, b.x, c.x, d.x, t) * mask;
    out[1] = blend4(a.y, b.y, c.y, d.y, t) * mask;
    out[2] = blend4(a.z, b.z, c.z, d.z, t) * mask;
}

// ---------------------------------------------------------------------------------------------
// a10  mixture-of-logistics probabilities of one projected sample      network/dist_decoder.py:109-140
//   t: normalised inverse depth on the reference ray, lo/hi: half intervals (dist_decoder.py:34-38)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void logistic_prob(float t, float lo, float hi, float mu0, float mu1, float s0, float s1,
                                              float aw, float nu, bool use_vis, float& visibility, float& hit) {
    const float near = t - lo, far = t + hi;
    float c00 = fmaf(0.5f, tanh_((near - mu0) * s0), 0.5f), c01 = fmaf(0.5f, tanh_((near - mu1) * s1), 0.5f);
    float c10 = fmaf(0.5f, tanh_((far - mu0) * s0), 0.5f), c11 = fmaf(0.5f, tanh_((far - mu1) * s1), 0.5f);
    if (use_vis) { c00 *= nu; c01 *= nu; c10 *= nu; c11 *= nu; }
    const float m0 = aw, m1 = 1.0f - aw;
    visibility = fmaf(1.0f - c01, m1, (1.0f - c00) * m0);
    hit = fmaf(c11 - c01, m1, (c10 - c00) * m0);
}

// ---------------------------------------------------------------------------------------------
// MFMA MLP layers (layout: nr_layout.h).  NT = point tiles (16 points each) processed per wave.
//   xq[t][4*kq + j] : B operands of the quad K-steps,  x1[t][k1] : B operands of the single K-steps
//   acc[t][mo]      : accumulators (D layout), caller decides the initial value
// ---------------------------------------------------------------------------------------------
template <int L, int NT, class WS>
__device__ __forceinline__ void layer_bias(WS W, int lane, v4f (&acc)[NT][kShape[L].mt_out]) {
    constexpr int MT = kShape[L].mt_out;
    NR_PRAGMA_UNROLL
    for (int mo = 0; mo < MT; ++mo) {
        const float4 b = wld4(W, (lane >> 4) * 16, (bias_offset(L, ws_ar<WS>::value) + mo * 16) * 4);
        NR_PRAGMA_UNROLL
        for (int t = 0; t < NT; ++t) { acc[t][mo][0] = b.x; acc[t][mo][1] = b.y; acc[t][mo][2] = b.z; acc[t][mo][3] = b.w; }
    }
}

#ifndef NR_PREFETCH
#define NR_PREFETCH 2
#endif

// One quad of A fragments feeds 4 K-steps x NT slots of MFMAs.
template <int NT, int KQX>
__device__ __forceinline__ void mfma_quad(const float4& a, int kq, const float (&xq)[NT][KQX], v4f (&acc)[NT]) {
#ifdef NR_BF16_SPLIT       // the split library: (a.x, a.y) = hi, (a.z, a.w) = lo bf16 halves of the quad's four weights, three MFMAs per slot
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t)
        acc[t] = nr_mfma16_bf16q3(a.x, a.y, a.z, a.w, xq[t][4 * kq + 0], xq[t][4 * kq + 1], xq[t][4 * kq + 2], xq[t][4 * kq + 3], acc[t]);
    return;
#elif defined(NR_BF16_QUADS)      // the bf16-operand library: (a.x, a.y) hold the quad's four weights as bf16, one MFMA per slot
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t)
        acc[t] = nr_mfma16_bf16q(a.x, a.y, xq[t][4 * kq + 0], xq[t][4 * kq + 1], xq[t][4 * kq + 2], xq[t][4 * kq + 3], acc[t]);
    return;
#endif
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t) acc[t] = nr_mfma16(a.x, xq[t][4 * kq + 0], acc[t]);
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t) acc[t] = nr_mfma16(a.y, xq[t][4 * kq + 1], acc[t]);
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t) acc[t] = nr_mfma16(a.z, xq[t][4 * kq + 2], acc[t]);
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t) acc[t] = nr_mfma16(a.w, xq[t][4 * kq + 3], acc[t]);
}

#if defined(NR_BF16_QUADS) && !defined(NR_BF16_SPLIT) && !defined(NEURAY_EMU)      // (the emulator keeps the per-quad form: the same sums in another order)
// The plain bf16-operand library on gfx950's K = 32 MFMA: two consecutive quads (a lane's 4 + 4 K-values) per v_mfma_f32_16x16x32_bf16 instead
// of one v_mfma_f32_16x16x16_bf16 each - half the MFMAs.  Measured (round 6): 2.84 -> 2.82 ms per launch - that library is VALU-bound, the
// matrix pipe was never its limit.  The two-way split library (NR_BF16_SPLIT) stays on K = 16: paired, its kernel keeps 288 B of scratch per
// lane at the 168-register budget and runs 4.89 instead of 3.73 ms; its K = 32 successor is AR_X3 of the fp32 library (nr_layout.h).
// a0 / a1: the fragments of quads kq and kq + 1.
// B operands of a layer's quad K-steps as bf16 pairs, converted (and, in the split library, split) ONCE per layer call
template <int NT, int KQX> struct BfOpnd {
    unsigned h[NT][(KQX + 1) / 2];
#ifdef NR_BF16_SPLIT
    unsigned l[NT][(KQX + 1) / 2];
#endif
};
template <int NT, int KQX>
__device__ __forceinline__ BfOpnd<NT, KQX> bf_operand(const float (&xq)[NT][KQX]) {
    BfOpnd<NT, KQX> o;
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t)
        NR_PRAGMA_UNROLL
        for (int i = 0; i < KQX / 2; ++i) {
#ifdef NR_BF16_SPLIT
            nr_split_bf16(xq[t][2 * i], xq[t][2 * i + 1], o.h[t][i], o.l[t][i]);
#else
            o.h[t][i] = nr_pk_bf16(xq[t][2 * i], xq[t][2 * i + 1]);
#endif
        }
    return o;
}
template <int NT, int KQX>
__device__ __forceinline__ void mfma_quad_pair(const float4& a0, const float4& a1, int kq, const BfOpnd<NT, KQX>& x, v4f (&acc)[NT]) {
    nr_v4u ah;
    ah[0] = __builtin_bit_cast(unsigned, a0.x); ah[1] = __builtin_bit_cast(unsigned, a0.y);
    ah[2] = __builtin_bit_cast(unsigned, a1.x); ah[3] = __builtin_bit_cast(unsigned, a1.y);
#ifdef NR_BF16_SPLIT
    nr_v4u al;
    al[0] = __builtin_bit_cast(unsigned, a0.z); al[1] = __builtin_bit_cast(unsigned, a0.w);
    al[2] = __builtin_bit_cast(unsigned, a1.z); al[3] = __builtin_bit_cast(unsigned, a1.w);
#endif
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t) {
        nr_v4u bh;
        bh[0] = x.h[t][2 * kq]; bh[1] = x.h[t][2 * kq + 1]; bh[2] = x.h[t][2 * kq + 2]; bh[3] = x.h[t][2 * kq + 3];
#ifdef NR_BF16_SPLIT
        nr_v4u bl;
        bl[0] = x.l[t][2 * kq]; bl[1] = x.l[t][2 * kq + 1]; bl[2] = x.l[t][2 * kq + 2]; bl[3] = x.l[t][2 * kq + 3];
        acc[t] = nr_mfma16x32_bf16(al, bh, acc[t]);          // (small terms first, as nr_mfma16_bf16q3)
        acc[t] = nr_mfma16x32_bf16(ah, bl, acc[t]);
        acc[t] = nr_mfma16x32_bf16(ah, bh, acc[t]);
#else
        acc[t] = nr_mfma16x32_bf16(ah, bh, acc[t]);
#endif
    }
}
#endif

// accumulate a K-slice (quads [KQ0, KQ0+KQN), singles [K10, K10+K1N)) of output tile `mo` (may be a runtime value)
// of layer L; the operand arrays hold only the slice.  The fragment stream is software-pipelined: the load of
// fragment i+1 is issued before the MFMAs of fragment i (hipcc otherwise emits load -> wait -> MFMAs per fragment and
// exposes the full LDS / L2 latency in front of every 4*NT MFMAs).
template <int L, int NT, int KQ0, int KQN, int K10, int K1N, class WS, int KQX, int K1X>
__device__ __forceinline__ void layer_tile_slice(WS W, int lane, int mo,
                                                 const float (&xq)[NT][KQX], const float (&x1)[NT][K1X], v4f (&acc)[NT]) {
    constexpr int KQ = kShape[L].kq, K1 = kShape[L].k1, AR = ws_ar<WS>::value;
    static_assert(!ar_splits(L, AR), "fp32 operands for a layer the source's layout stores split");
    static_assert(KQ0 + KQN <= KQ && K10 + K1N <= K1, "slice outside the layer");
    static_assert(KQX >= (KQN > 0 ? 4 * KQN : 1) && K1X >= (K1N > 0 ? K1N : 1), "operand arrays too small");
    // the run-time tile index goes into the per-lane byte offset, so that every load's scalar offset is a compile-time constant: as
    // `soffset = constant + mo * stride` each of the ~20 loads of a slice had its own loop-invariant SGPR, which hipcc hoisted out of
    // the tile loop and spilled to VGPR lanes (v_writelane / v_readlane + s_nop inside the loop)
    const int v1 = lane * 4 + mo * (K1 * 256), vq = lane * 16 + mo * (KQ * 1024);
    float s1[K1N > 0 ? K1N : 1];
    NR_PRAGMA_UNROLL
    for (int k1 = 0; k1 < K1N; ++k1) s1[k1] = wld1(W, v1, (single_offset(L, AR) + (K10 + k1) * 64) * 4);
#if defined(NR_BF16_QUADS) && !defined(NR_BF16_SPLIT) && !defined(NEURAY_EMU)
    if constexpr (KQN > 0 && KQN % 2 == 0) {
        const BfOpnd<NT, KQX> xb = bf_operand(xq);
        float4 c0 = wldq(W, vq, (quads_offset(L, AR) + KQ0 * 256) * 4), c1 = wldq(W, vq, (quads_offset(L, AR) + (KQ0 + 1) * 256) * 4);
        NR_PRAGMA_UNROLL
        for (int kq = 0; kq < KQN; kq += 2) {
            float4 n0 = c0, n1 = c1;
            if (kq + 2 < KQN) {
                n0 = wldq(W, vq, (quads_offset(L, AR) + (KQ0 + kq + 2) * 256) * 4);
                n1 = wldq(W, vq, (quads_offset(L, AR) + (KQ0 + kq + 3) * 256) * 4);
            }
            NR_PIN();
            mfma_quad_pair<NT>(c0, c1, kq, xb, acc);
            c0 = n0; c1 = n1;
        }
    } else
#endif
    if (KQN > 0) {
        float4 cur = wldq(W, vq, (quads_offset(L, AR) + KQ0 * 256) * 4);
        NR_PRAGMA_UNROLL
        for (int kq = 0; kq < KQN; ++kq) {
            float4 nxt = cur;
            if (kq + 1 < KQN) nxt = wldq(W, vq, (quads_offset(L, AR) + (KQ0 + kq + 1) * 256) * 4);
            NR_PIN();
            mfma_quad<NT>(cur, kq, xq, acc);
            cur = nxt;
        }
    }
    NR_PRAGMA_UNROLL
    for (int k1 = 0; k1 < K1N; ++k1)
        NR_PRAGMA_UNROLL
        for (int t = 0; t < NT; ++t) acc[t] = nr_mfma16(s1[k1], x1[t][k1], acc[t]);
}

// accumulate one whole output tile `mo` of layer L
template <int L, int NT, class WS, int KQX, int K1X>
__device__ __forceinline__ void layer_tile(WS W, int lane, int mo,
                                           const float (&xq)[NT][KQX], const float (&x1)[NT][K1X], v4f (&acc)[NT]) {
    layer_tile_slice<L, NT, 0, kShape[L].kq, 0, kShape[L].k1>(W, lane, mo, xq, x1, acc);
}

// ---- whole layers ------------------------------------------------------------------------------------------
// The first fragments of a layer (NR_PREFETCH + 1 quads, the singles, the bias) are loaded into a LayerPre by
// layer_prefetch; a layer running in the middle of a phase issues the NEXT layer's layer_prefetch two quads before
// its own MFMA stream ends, so the LDS (or L2) latency of the next layer's first operands hides behind MFMAs and the
// activation code instead of sitting in front of every layer (hipcc emits ds_read x5 -> s_waitcnt -> first MFMA).
template <int L> struct LayerPre {
    static constexpr int MT = kShape[L].mt_out, KQ = kShape[L].kq, K1 = kShape[L].k1, NQ = MT * KQ;
    static constexpr int NF = NQ < NR_PREFETCH + 1 ? NQ : NR_PREFETCH + 1;
    float4 q[NF > 0 ? NF : 1];
    float4 b[MT];
    float s1[MT * K1 > 0 ? MT * K1 : 1];
};
struct NoLayer {};

template <int L, class WS>
__device__ __forceinline__ void layer_prefetch(WS W, int lane, LayerPre<L>& p) {
    constexpr int AR = ws_ar<WS>::value;
    static_assert(!ar_splits(L, AR), "fp32 operands for a layer the source's layout stores split");
    NR_PRAGMA_UNROLL
    for (int i = 0; i < LayerPre<L>::NF; ++i) p.q[i] = wldq(W, lane * 16, (quads_offset(L, AR) + i * 256) * 4);
    NR_PRAGMA_UNROLL
    for (int mo = 0; mo < LayerPre<L>::MT; ++mo) p.b[mo] = wld4(W, (lane >> 4) * 16, (bias_offset(L, AR) + mo * 16) * 4);
    NR_PRAGMA_UNROLL
    for (int i = 0; i < LayerPre<L>::MT * LayerPre<L>::K1; ++i) p.s1[i] = wld1(W, lane * 4, (single_offset(L, AR) + i * 64) * 4);
}
template <class WS> __device__ __forceinline__ void layer_prefetch(WS, int, NoLayer&) {}

// ---- vector rows (nr_layout.h kVec): dot products on the VALU for the narrow output rows -------------------
template <int L> struct VecPre {
    static constexpr int N = kVec[L].n, TI = kVec[L].tiles;
    float4 w[N * TI > 0 ? N * TI : 1];
    float4 b;
};
template <int L, class WS>
__device__ __forceinline__ void layer_prefetch(WS W, int lane, VecPre<L>& p) {
    constexpr int AR = ws_ar<WS>::value;
    NR_PRAGMA_UNROLL
    for (int i = 0; i < VecPre<L>::N * VecPre<L>::TI; ++i) p.w[i] = wld4(W, (lane >> 4) * 16, (vec_offset(L, AR) + i * 16) * 4);
    p.b = wld4(W, (lane >> 4) * 16, vec_bias_offset(L, AR) * 4);
}
// out[t][j] = b_j + sum_f w_j[f] x[t][f], identical in the four lane groups; x in the D layout (4 registers per tile)
template <int L, int NT, int KX>
__device__ __forceinline__ void layer_vec(const VecPre<L>& p, const float (&x)[NT][KX], float (&out)[NT][kVec[L].n]) {
    constexpr int N = kVec[L].n, TI = kVec[L].tiles;
    static_assert(KX >= 4 * TI, "operand array too small");
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t)
        NR_PRAGMA_UNROLL
        for (int j = 0; j < N; ++j) {
            // two products per instruction (v_pk_mul / v_pk_fma over register pairs), the halves added at the end
            nr_v2 a2 = nr_v2_mul(nr_v2_make(p.w[j * TI].x, p.w[j * TI].y), nr_v2_make(x[t][0], x[t][1]));
            a2 = nr_v2_fma(nr_v2_make(p.w[j * TI].z, p.w[j * TI].w), nr_v2_make(x[t][2], x[t][3]), a2);
            NR_PRAGMA_UNROLL
            for (int ti = 1; ti < TI; ++ti) {
                a2 = nr_v2_fma(nr_v2_make(p.w[j * TI + ti].x, p.w[j * TI + ti].y), nr_v2_make(x[t][4 * ti], x[t][4 * ti + 1]), a2);
                a2 = nr_v2_fma(nr_v2_make(p.w[j * TI + ti].z, p.w[j * TI + ti].w), nr_v2_make(x[t][4 * ti + 2], x[t][4 * ti + 3]), a2);
            }
            const float a = a2.x + a2.y;
            const float bj = j == 0 ? p.b.x : (j == 1 ? p.b.y : (j == 2 ? p.b.z : p.b.w));
            out[t][j] = nr_group_sum(a) + bj;
        }
}
template <int L, int NT, class WS, int KX>
__device__ __forceinline__ void layer_vec(WS W, int lane, const float (&x)[NT][KX], float (&out)[NT][kVec[L].n]) {
    VecPre<L> p;
    layer_prefetch<L>(W, lane, p);
    layer_vec<L, NT>(p, x, out);
}

// all output tiles of layer L; the fragment stream (tile-major) is pipelined across tile boundaries too
template <int L, int NT, class WS, int KQX, int K1X, class PN>
__device__ __forceinline__ void layer_acc(WS W, int lane, const LayerPre<L>& pre, const float (&xq)[NT][KQX],
                                          const float (&x1)[NT][K1X], v4f (&acc)[NT][kShape[L].mt_out], PN& next) {
    constexpr int MT = kShape[L].mt_out, KQ = kShape[L].kq, K1 = kShape[L].k1;
    static_assert(KQX >= (KQ > 0 ? 4 * KQ : 1) && K1X >= (K1 > 0 ? K1 : 1), "operand arrays too small");
#if defined(NR_BF16_QUADS) && !defined(NR_BF16_SPLIT) && !defined(NEURAY_EMU)
    if constexpr (KQ > 0 && KQ % 2 == 0) {
        // quad pairs on the K = 32 MFMA (mfma_quad_pair): a ring of four fragments, two consumed and two fetched per step
        constexpr int NQ = MT * KQ, NF = LayerPre<L>::NF, R = NQ < 4 ? NQ : 4;
        const BfOpnd<NT, KQX> xb = bf_operand(xq);
        float4 ring[R];
        NR_PRAGMA_UNROLL
        for (int i = 0; i < R; ++i) ring[i] = i < NF ? pre.q[i < NF ? i : 0] : wldq(W, lane * 16, (quads_offset(L, ws_ar<WS>::value) + i * 256) * 4);
        NR_PRAGMA_UNROLL
        for (int i = 0; i < NQ; i += 2) {
            const float4 c0 = ring[i % R], c1 = ring[(i + 1) % R];
            if (i == (NQ >= 4 ? NQ - 4 : 0)) layer_prefetch(W, lane, next);
            NR_PIN();
            const int mo = i / KQ, kq = i % KQ;
            v4f a[NT];
            NR_PRAGMA_UNROLL
            for (int t = 0; t < NT; ++t) a[t] = acc[t][mo];
            mfma_quad_pair<NT>(c0, c1, kq, xb, a);
            NR_PRAGMA_UNROLL
            for (int t = 0; t < NT; ++t) acc[t][mo] = a[t];
            if (i + R < NQ) {
                ring[i % R] = wldq(W, lane * 16, (quads_offset(L, ws_ar<WS>::value) + (i + R) * 256) * 4);
                ring[(i + 1) % R] = wldq(W, lane * 16, (quads_offset(L, ws_ar<WS>::value) + (i + R + 1) * 256) * 4);
            }
        }
    } else
#endif
    if constexpr (KQ > 0) {
        // fragment stream with NR_PREFETCH quads in flight ahead of the one being consumed
        constexpr int NQ = MT * KQ, PF = NR_PREFETCH < NQ ? NR_PREFETCH : NQ - 1;
        float4 ring[PF + 1];
        NR_PRAGMA_UNROLL
        for (int i = 0; i <= PF; ++i) ring[i] = pre.q[i];
        NR_PRAGMA_UNROLL
        for (int i = 0; i < NQ; ++i) {
            const float4 cur = ring[i % (PF + 1)];
            if (i == (NQ >= 2 ? NQ - 2 : 0)) layer_prefetch(W, lane, next);
            NR_PIN();
            const int mo = i / KQ, kq = i % KQ;
            v4f a[NT];
            NR_PRAGMA_UNROLL
            for (int t = 0; t < NT; ++t) a[t] = acc[t][mo];
            mfma_quad<NT>(cur, kq, xq, a);
            NR_PRAGMA_UNROLL
            for (int t = 0; t < NT; ++t) acc[t][mo] = a[t];
            if (i + PF + 1 < NQ) ring[i % (PF + 1)] = wldq(W, lane * 16, (quads_offset(L, ws_ar<WS>::value) + (i + PF + 1) * 256) * 4);
        }
    } else {
        layer_prefetch(W, lane, next);
        NR_PIN();
    }
    NR_PRAGMA_UNROLL
    for (int mo = 0; mo < MT; ++mo)
        NR_PRAGMA_UNROLL
        for (int k1 = 0; k1 < K1; ++k1)
            NR_PRAGMA_UNROLL
            for (int t = 0; t < NT; ++t) acc[t][mo] = nr_mfma16(pre.s1[mo * K1 + k1], x1[t][k1], acc[t][mo]);
}

// y = act(W x + b) with D-layout output registers y[t][4*mo + r]
enum Act { ACT_NONE, ACT_ELU, ACT_RELU };
template <int A, int L> __device__ __forceinline__ float apply_act(float x) {
    if (A == ACT_ELU) return kOutScaled[L] ? elu_s(x) : elu(x);
    if (A == ACT_RELU) return fmaxf(x, 0.0f);
    return x;
}

// two activations at a time: the multiply-add around the exponentials is one packed instruction for the pair (bitwise the
// scalar result); exp2 and med3 have no packed form
template <int A, int L> __device__ __forceinline__ void apply_act2(float x0, float x1, float& y0, float& y1) {
    if (A == ACT_ELU) {
        if (kOutScaled[L]) {       // L(2^x - 1)
            const nr_v2 f = nr_v2_fma(nr_v2_make(nr_fast_exp2(x0), nr_fast_exp2(x1)), nr_v2_make((float)kLog2e, (float)kLog2e),
                                      nr_v2_make(-(float)kLog2e, -(float)kLog2e));
            y0 = nr_med3(x0, f.x, 0.0f); y1 = nr_med3(x1, f.y, 0.0f);
            return;
        }
    }
    y0 = apply_act<A, L>(x0); y1 = apply_act<A, L>(x1);
}

// layer L from its prefetched head `pre`; `next` (a LayerPre of the following layer of the same phase, or NoLayer)
// is filled on the way
template <int L, int NT, int A, class WS, int KQX, int K1X, class PN>
__device__ __forceinline__ void layer_fwd(WS W, int lane, const LayerPre<L>& pre, const float (&xq)[NT][KQX],
                                          const float (&x1)[NT][K1X], float (&y)[NT][kShape[L].mt_out * 4], PN& next) {
    constexpr int MT = kShape[L].mt_out;
    v4f acc[NT][MT];
    NR_PRAGMA_UNROLL
    for (int mo = 0; mo < MT; ++mo)
        NR_PRAGMA_UNROLL
        for (int t = 0; t < NT; ++t) {
            acc[t][mo][0] = pre.b[mo].x; acc[t][mo][1] = pre.b[mo].y; acc[t][mo][2] = pre.b[mo].z; acc[t][mo][3] = pre.b[mo].w;
        }
    layer_acc<L, NT>(W, lane, pre, xq, x1, acc, next);
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t)
        NR_PRAGMA_UNROLL
        for (int mo = 0; mo < MT; ++mo)
            NR_PRAGMA_UNROLL
            for (int r = 0; r < 4; r += 2) {
                apply_act2<A, L>(acc[t][mo][r], acc[t][mo][r + 1], y[t][4 * mo + r], y[t][4 * mo + r + 1]);
            }
}

// self-contained form: loads its own head (first layer of a phase, stand-alone kernels)
template <int L, int NT, int A, class WS, int KQX, int K1X>
__device__ __forceinline__ void layer_fwd(WS W, int lane, const float (&xq)[NT][KQX],
                                          const float (&x1)[NT][K1X], float (&y)[NT][kShape[L].mt_out * 4]) {
    LayerPre<L> pre;
    NoLayer none;
    layer_prefetch<L>(W, lane, pre);
    layer_fwd<L, NT, A>(W, lane, pre, xq, x1, y, none);
}

// =============================================================================================
// AR_X3 layers (nr_layout.h): the same layer functions on split operands.  Overloads of the fp32 forms above, selected by the
// types they are handed - a GlbW3 / LdsW3 weight source, a LayerPre3 head, an Opnd3 operand - so a kernel body written once
// (points_kernel) instantiates either arithmetic.  Single K-steps, biases, vector rows and activations are the fp32 forms.
// =============================================================================================
#ifndef NR_PREFETCH3
#define NR_PREFETCH3 1          // fragment units (3 x 16 bytes per lane) in flight ahead of the one being consumed
#endif

// B operands of a layer's quad K-steps, split: p[part][t][i] = the bf16 pair of registers (2 i, 2 i + 1) of slot t
template <int NT, int KQX> struct Opnd3 { unsigned p[3][NT][(KQX + 1) / 2]; };
template <int NT, int KQX>
__device__ __forceinline__ Opnd3<NT, KQX> split_operand(const float (&x)[NT][KQX]) {
    Opnd3<NT, KQX> o;
    if constexpr (KQX >= 2) {
        static_assert(KQX % 2 == 0, "register pairs");
        NR_PRAGMA_UNROLL
        for (int t = 0; t < NT; ++t)
            NR_PRAGMA_UNROLL
            for (int i = 0; i < KQX / 2; ++i) nr_split3(x[t][2 * i], x[t][2 * i + 1], o.p[0][t][i], o.p[1][t][i], o.p[2][t][i]);
    } else {
        NR_PRAGMA_UNROLL
        for (int t = 0; t < NT; ++t) { o.p[0][t][0] = 0u; o.p[1][t][0] = 0u; o.p[2][t][0] = 0u; }
    }
    return o;
}
// what a layer function of arithmetic AR takes as its quad operand: the registers themselves (fp32) or their split
template <int AR, int NT, int KQX>
__device__ __forceinline__ decltype(auto) operand(const float (&x)[NT][KQX]) {
    if constexpr (AR == AR_X3) return split_operand(x);
    else return (x);
}

// one fragment unit of an output tile: the three parts of a quad PAIR (K = 32), or - layers with a single quad - of that quad (K = 16,
// words 0..1 of each part)
struct Frag3 { nr_v4u p[3]; };
template <int L> struct Units3 {
    static_assert(ar_splits(L, AR_X3), "a per-point layer keeps fp32 operands (nr_layout.h)");
    static constexpr int KQ = kShape[L].kq;
    static_assert(KQ <= 1 || KQ % 2 == 0, "AR_X3: a layer's quads come in pairs, or it has a single one");
    static constexpr bool HALF = KQ == 1;
    static constexpr int UPM = HALF ? 1 : KQ / 2;                   // units per output tile
    static constexpr int NU = kShape[L].mt_out * UPM;
    static constexpr int TILE = tile_quads_floats(L, AR_X3);        // floats per output tile
    static constexpr int UNIT = HALF ? 384 : 768, PART = HALF ? 128 : 256, LANE = HALF ? 8 : 16;
};
// unit u of tile mo_c (+ a run-time tile index inside mo_bytes)
template <int L, class WS>
__device__ __forceinline__ Frag3 frag3_load(WS W, int lane, int mo_bytes, int mo_c, int u) {
    typedef Units3<L> U;
    const int off = quads_offset(L, AR_X3) + mo_c * U::TILE + u * U::UNIT;
    Frag3 f;
    NR_PRAGMA_UNROLL
    for (int pt = 0; pt < 3; ++pt) {
        if constexpr (U::HALF) {
            const nr_v2u h = wld2u(W, lane * 8 + mo_bytes, (off + pt * U::PART) * 4);
            f.p[pt][0] = h[0]; f.p[pt][1] = h[1]; f.p[pt][2] = 0u; f.p[pt][3] = 0u;
        } else {
            f.p[pt] = wld4u(W, lane * 16 + mo_bytes, (off + pt * U::PART) * 4);
        }
    }
    return f;
}
// six products (weight part, activation part) per slot, smallest first, back to back on the slot's accumulator: a chain of
// v_mfma_f32_16x16x32_bf16 on ONE accumulator issues every 18.1 cycles, alternating between two accumulators costs 20.2 and between
// three 24.2 (tests/hw/split_arith_probe.hip, profiles/r06_c_*).  u = the unit's position inside the operand array handed in.
template <int L, int NT, int KQX>
__device__ __forceinline__ void mfma_unit3(const Frag3& a, int u, const Opnd3<NT, KQX>& x, v4f (&acc)[NT]) {
    constexpr int WI[6] = {2, 0, 1, 1, 0, 0}, XJ[6] = {0, 2, 1, 0, 1, 0};
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t)
        NR_PRAGMA_UNROLL
        for (int q = 0; q < 6; ++q) {
            if constexpr (Units3<L>::HALF) {
                nr_v2u av, bv;
                av[0] = a.p[WI[q]][0]; av[1] = a.p[WI[q]][1];
                bv[0] = x.p[XJ[q]][t][0]; bv[1] = x.p[XJ[q]][t][1];
                acc[t] = nr_mfma16x16_bf16(av, bv, acc[t]);
            } else {
                nr_v4u bv;
                bv[0] = x.p[XJ[q]][t][4 * u]; bv[1] = x.p[XJ[q]][t][4 * u + 1]; bv[2] = x.p[XJ[q]][t][4 * u + 2]; bv[3] = x.p[XJ[q]][t][4 * u + 3];
                acc[t] = nr_mfma16x32_bf16(a.p[WI[q]], bv, acc[t]);
            }
        }
}

template <int L> struct LayerPre3 {
    static constexpr int MT = kShape[L].mt_out, KQ = kShape[L].kq, K1 = kShape[L].k1, NU = KQ > 0 ? Units3<L>::NU : 0;
    static constexpr int NF = NU < NR_PREFETCH3 ? NU : NR_PREFETCH3;
    Frag3 q[NF > 0 ? NF : 1];
    float4 b[MT];
    float s1[MT * K1 > 0 ? MT * K1 : 1];
};
template <int L, int AR> struct layer_pre_of { typedef LayerPre<L> type; };
template <int L> struct layer_pre_of<L, AR_X3> { typedef LayerPre3<L> type; };
template <int L, int AR> using LayerPreT = typename layer_pre_of<L, AR>::type;

template <int L, class WS>
__device__ __forceinline__ void layer_prefetch(WS W, int lane, LayerPre3<L>& p) {
    static_assert(ws_ar<WS>::value == AR_X3, "a LayerPre3 is filled from an AR_X3 weight source");
    NR_PRAGMA_UNROLL
    for (int i = 0; i < LayerPre3<L>::NF; ++i) p.q[i] = frag3_load<L>(W, lane, 0, i / Units3<L>::UPM, i % Units3<L>::UPM);
    NR_PRAGMA_UNROLL
    for (int mo = 0; mo < LayerPre3<L>::MT; ++mo) p.b[mo] = wld4(W, (lane >> 4) * 16, (bias_offset(L, AR_X3) + mo * 16) * 4);
    NR_PRAGMA_UNROLL
    for (int i = 0; i < LayerPre3<L>::MT * LayerPre3<L>::K1; ++i) p.s1[i] = wld1(W, lane * 4, (single_offset(L, AR_X3) + i * 64) * 4);
}

template <int L, int NT, class WS, int KQX, int K1X, class PN>
__device__ __forceinline__ void layer_acc(WS W, int lane, const LayerPre3<L>& pre, const Opnd3<NT, KQX>& xq,
                                          const float (&x1)[NT][K1X], v4f (&acc)[NT][kShape[L].mt_out], PN& next) {
    constexpr int MT = kShape[L].mt_out, KQ = kShape[L].kq, K1 = kShape[L].k1;
    static_assert(KQX >= (KQ > 0 ? 4 * KQ : 1) && K1X >= (K1 > 0 ? K1 : 1), "operand arrays too small");
    if constexpr (KQ > 0) {
        typedef Units3<L> U;
        constexpr int NU = U::NU, NF = LayerPre3<L>::NF, D = NU < NR_PREFETCH3 + 1 ? NU : NR_PREFETCH3 + 1;
        Frag3 ring[D];
        NR_PRAGMA_UNROLL
        for (int i = 0; i < D; ++i) ring[i] = i < NF ? pre.q[i < NF ? i : 0] : frag3_load<L>(W, lane, 0, i / U::UPM, i % U::UPM);
        NR_PRAGMA_UNROLL
        for (int i = 0; i < NU; ++i) {
            const Frag3 cur = ring[i % D];
            if (i == (NU >= 2 ? NU - 2 : 0)) layer_prefetch(W, lane, next);
            NR_PIN();
            const int mo = i / U::UPM, u = i % U::UPM;
            v4f a[NT];
            NR_PRAGMA_UNROLL
            for (int t = 0; t < NT; ++t) a[t] = acc[t][mo];
            mfma_unit3<L, NT>(cur, u, xq, a);
            NR_PRAGMA_UNROLL
            for (int t = 0; t < NT; ++t) acc[t][mo] = a[t];
            if (i + D < NU) ring[i % D] = frag3_load<L>(W, lane, 0, (i + D) / U::UPM, (i + D) % U::UPM);
        }
    } else {
        layer_prefetch(W, lane, next);
        NR_PIN();
    }
    NR_PRAGMA_UNROLL
    for (int mo = 0; mo < MT; ++mo)
        NR_PRAGMA_UNROLL
        for (int k1 = 0; k1 < K1; ++k1)
            NR_PRAGMA_UNROLL
            for (int t = 0; t < NT; ++t) acc[t][mo] = nr_mfma16(pre.s1[mo * K1 + k1], x1[t][k1], acc[t][mo]);
}

template <int L, int NT, int A, class WS, int KQX, int K1X, class PN>
__device__ __forceinline__ void layer_fwd(WS W, int lane, const LayerPre3<L>& pre, const Opnd3<NT, KQX>& xq,
                                          const float (&x1)[NT][K1X], float (&y)[NT][kShape[L].mt_out * 4], PN& next) {
    constexpr int MT = kShape[L].mt_out;
    v4f acc[NT][MT];
    NR_PRAGMA_UNROLL
    for (int mo = 0; mo < MT; ++mo)
        NR_PRAGMA_UNROLL
        for (int t = 0; t < NT; ++t) {
            acc[t][mo][0] = pre.b[mo].x; acc[t][mo][1] = pre.b[mo].y; acc[t][mo][2] = pre.b[mo].z; acc[t][mo][3] = pre.b[mo].w;
        }
    layer_acc<L, NT>(W, lane, pre, xq, x1, acc, next);
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t)
        NR_PRAGMA_UNROLL
        for (int mo = 0; mo < MT; ++mo)
            NR_PRAGMA_UNROLL
            for (int r = 0; r < 4; r += 2) {
                apply_act2<A, L>(acc[t][mo][r], acc[t][mo][r + 1], y[t][4 * mo + r], y[t][4 * mo + r + 1]);
            }
}
// y = act(scale_t * (W x) + b): a layer whose input is the operand times one scalar per (point, view) column - W (s x) = s (W x) - so that
// ONE split of x serves this layer and the layers that consume x itself (vis_fc2.0 on x * vis' next to rgb_fc.0 on x: ibrnet.py:347-349,
// 363).  Differs from evaluating W (s x) by fp32 rounding only.  No single K-steps.
template <int L, int NT, int A, class WS, int KQX, class PN>
__device__ __forceinline__ void layer_fwd_scaled(WS W, int lane, const LayerPre3<L>& pre, const Opnd3<NT, KQX>& xq, const float (&scale)[NT],
                                                 float (&y)[NT][kShape[L].mt_out * 4], PN& next) {
    constexpr int MT = kShape[L].mt_out;
    static_assert(kShape[L].k1 == 0, "scaled form: quad K-steps only");
    v4f acc[NT][MT];
    float none1[NT][1];
    NR_PRAGMA_UNROLL
    for (int mo = 0; mo < MT; ++mo)
        NR_PRAGMA_UNROLL
        for (int t = 0; t < NT; ++t) { acc[t][mo][0] = 0.0f; acc[t][mo][1] = 0.0f; acc[t][mo][2] = 0.0f; acc[t][mo][3] = 0.0f; }
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t) none1[t][0] = 0.0f;
    layer_acc<L, NT>(W, lane, pre, xq, none1, acc, next);
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t)
        NR_PRAGMA_UNROLL
        for (int mo = 0; mo < MT; ++mo) {
            const float b[4] = {pre.b[mo].x, pre.b[mo].y, pre.b[mo].z, pre.b[mo].w};
            NR_PRAGMA_UNROLL
            for (int r = 0; r < 4; r += 2)
                apply_act2<A, L>(fmaf(acc[t][mo][r], scale[t], b[r]), fmaf(acc[t][mo][r + 1], scale[t], b[r + 1]), y[t][4 * mo + r], y[t][4 * mo + r + 1]);
        }
}

template <int L, int NT, int A, class WS, int KQX, int K1X>
__device__ __forceinline__ void layer_fwd(WS W, int lane, const Opnd3<NT, KQX>& xq,
                                          const float (&x1)[NT][K1X], float (&y)[NT][kShape[L].mt_out * 4]) {
    LayerPre3<L> pre;
    NoLayer none;
    layer_prefetch<L>(W, lane, pre);
    layer_fwd<L, NT, A>(W, lane, pre, xq, x1, y, none);
}

// K-slice of output tile `mo` (run-time) of layer L: quads [KQ0, KQ0 + KQN) = units [KQ0 / 2, (KQ0 + KQN) / 2); the operand holds the slice
template <int L, int NT, int KQ0, int KQN, int K10, int K1N, class WS, int KQX, int K1X>
__device__ __forceinline__ void layer_tile_slice(WS W, int lane, int mo,
                                                 const Opnd3<NT, KQX>& xq, const float (&x1)[NT][K1X], v4f (&acc)[NT]) {
    constexpr int KQ = kShape[L].kq, K1 = kShape[L].k1;
    static_assert(KQ0 % 2 == 0 && KQN % 2 == 0 && KQ % 2 == 0, "AR_X3 slices are whole quad pairs");
    static_assert(KQ0 + KQN <= KQ && K10 + K1N <= K1, "slice outside the layer");
    static_assert(KQX >= (KQN > 0 ? 4 * KQN : 1) && K1X >= (K1N > 0 ? K1N : 1), "operand arrays too small");
    const int v1 = lane * 4 + mo * (K1 * 256), mo_bytes = mo * (Units3<L>::TILE * 4);
    float s1[K1N > 0 ? K1N : 1];
    NR_PRAGMA_UNROLL
    for (int k1 = 0; k1 < K1N; ++k1) s1[k1] = wld1(W, v1, (single_offset(L, AR_X3) + (K10 + k1) * 64) * 4);
    if constexpr (KQN > 0) {
        constexpr int U0 = KQ0 / 2, UN = KQN / 2;
        Frag3 cur = frag3_load<L>(W, lane, mo_bytes, 0, U0);
        NR_PRAGMA_UNROLL
        for (int u = 0; u < UN; ++u) {
            Frag3 nxt = cur;
            if (u + 1 < UN) nxt = frag3_load<L>(W, lane, mo_bytes, 0, U0 + u + 1);
            NR_PIN();
            mfma_unit3<L, NT>(cur, u, xq, acc);
            cur = nxt;
        }
    }
    NR_PRAGMA_UNROLL
    for (int k1 = 0; k1 < K1N; ++k1)
        NR_PRAGMA_UNROLL
        for (int t = 0; t < NT; ++t) acc[t] = nr_mfma16(s1[k1], x1[t][k1], acc[t]);
}

// ---------------------------------------------------------------------------------------------
// Weight staging (nr_layout.h kPhase): the stage is two LDS regions.  phase_enter<PH> is called by every wave when it
// is done with the previous phase: one barrier (its fence makes each wave wait for its own DMA pieces first) after
// which phase PH is complete in its region and nobody reads the other region any more, so the copy of the NEXT phase
// is issued into that one right away and lands while PH is being computed.
// ---------------------------------------------------------------------------------------------
template <int PH, class WS>
__device__ __forceinline__ void stage_issue(float* dst, WS W, int wave, int nw, int lane) {
    constexpr int AR = ws_ar<WS>::value;
    constexpr int begin = phase_begin(PH, AR) * 4, bytes = phase_floats(PH, AR) * 4, NCH = (bytes + 1023) / 1024;
    static_assert(bytes % 16 == 0 && bytes <= stage_region_bytes(AR), "phase does not fit a stage region");
    // 1 KiB per wave instruction.  The last piece runs past the end of the phase up to the next KiB boundary: the source
    // bytes exist (later layers of the packed buffer; the buffer descriptor range-checks anyway) and the region is a
    // whole number of KiB, so no lane needs masking and the only per-lane address is lane * 16.
    for (int c = wave; c < NCH; c += nw) nr_dma16(ws_raw(W), dst + c * 256, lane, lane * 16, begin + c * 1024);
}

// seq0 = (tiles this workgroup has finished) * phase_count(VIS): the regions alternate along the phase sequence
template <int PH, bool VIS, class WS>
__device__ __forceinline__ typename ws_lds<WS>::type phase_enter(float* wl, WS W, int seq0, bool issue_next, int wave, int nw, int lane) {
    constexpr int AR = ws_ar<WS>::value, RF = stage_region_bytes(AR) / 4;
    NR_BLOCK_SYNC();
    const int r = (seq0 + phase_seq(PH, VIS)) & 1;
    if (issue_next) stage_issue<phase_next(PH, VIS)>(wl + (r ^ 1) * RF, W, wave, nw, lane);
    return typename ws_lds<WS>::type{wl + r * RF, phase_begin(PH, AR) * 4};
}

// ---------------------------------------------------------------------------------------------
// block-wide all-reduce over the view-waves (one wave per reference view).
//   red: LDS scratch of (nw + 1) * RMAX * 64 floats.  Deterministic (views summed in order 0..nw-1),
//   identical result in every wave.  Two barriers per round (reduce-scatter, then all-gather).
// ---------------------------------------------------------------------------------------------
// OP = RED_MAX0: rows whose index is a multiple of MAXSTRIDE take the maximum, all other rows the sum
enum RedOp { RED_SUM, RED_MAX, RED_MAX0 };
template <int OP, int MAXSTRIDE> __device__ __forceinline__ float red_combine(float a, float b, int row) {
    if (OP == RED_SUM) return a + b;
    if (OP == RED_MAX) return fmaxf(a, b);
    return (row % MAXSTRIDE == 0) ? fmaxf(a, b) : a + b;
}
template <int R, int RMAX, int OP, int MAXSTRIDE = 1>
__device__ __forceinline__ void block_allreduce(float (&v)[R], float* red, int wave, int nw, int lane) {
    static_assert(R <= RMAX, "allreduce scratch too small");
    NR_PRIO_HI();
    NR_PRAGMA_UNROLL
    for (int r = 0; r < R; ++r) red[(wave * RMAX + r) * 64 + lane] = v[r];
    NR_BLOCK_SYNC();
    for (int r = wave; r < R; r += nw) {
        float s = red[r * 64 + lane];
        for (int w = 1; w < nw; ++w) s = red_combine<OP, MAXSTRIDE>(s, red[(w * RMAX + r) * 64 + lane], r);
        red[(nw * RMAX + r) * 64 + lane] = s;
    }
    NR_BLOCK_SYNC();
    NR_PRAGMA_UNROLL
    for (int r = 0; r < R; ++r) v[r] = red[(nw * RMAX + r) * 64 + lane];
    NR_PRIO_LO();
}

}  // namespace nr
