// Fused InstanceNorm2d(affine) + activation (+ residual add) + reflection padding for the per-image encoders
// (network/ops.py:150-230 ResUNetLight, :43-75 ResidualBlock, network/vis_encoder.py:6-21).
//
// Every convolution of the encoders is followed by InstanceNorm -> ReLU / ELU (-> + skip -> ReLU) and the result is
// reflect-padded for the next 3x3 convolution.  Through PyTorch that is 4-6 full passes over the activation
// (MIOpen batch-norm, the activation, the add, the second activation, reflection_pad2d) and as many again in the backward:
// by rocprofv3 (profiles/r03_f_encoder_kernel_stats.csv) padding, normalisation and element-wise kernels are 31 % of the
// encoders' forward + backward time, the convolutions themselves 46 %.  These kernels do the whole chain in two passes:
//   stats :  per (image, channel) plane  sum (x - K), sum (x - K)^2  with K = the plane's first element (a shift keeps the
//            one-pass variance well conditioned)                                                            [reads x once]
//   apply :  out_pad = reflect_pad(act(gamma (x - mean) rstd + beta [+ residual]))  written once, already padded for the next
//            convolution (pad 0 / 1); the un-padded activation is the interior view of the same buffer       [reads x once]
// and the backward in two more (plane sums of g and g * xhat, then dx / d residual), where g is the incoming gradient of the
// PADDED output folded back onto the interior (the adjoint of the reflection) and taken through the activation.
// Tensors are NCHW contiguous fp32 (the layout MIOpen's fp32 Winograd kernels run in); purely HBM-bound: the roofline is
// bytes moved / 8 TB/s.
#pragma once
#include "nr_platform.h"

namespace nr {

enum NormAct { NORM_ACT_NONE = 0, NORM_ACT_RELU = 1, NORM_ACT_ELU = 2 };

__device__ __forceinline__ int reflect_index(int i, int n) {            // reflection without repeating the edge (pad < n)
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// raw [planes][2] (zeroed): sum (x - K), sum (x - K)^2 over the plane, K = x[plane][0].  grid = (chunks, planes)
__global__ void __launch_bounds__(256) inorm_stats_kernel(const float* __restrict__ x, int hw, float* __restrict__ raw) {
    const int plane = blockIdx.y;
    const float* px = x + (size_t)plane * hw;
    const float K = px[0];
    float s1 = 0.0f, s2 = 0.0f;
    if ((hw & 3) == 0) {                      // planes are 16-byte aligned then: one dwordx4 per thread and iteration
        const float4* p4 = reinterpret_cast<const float4*>(px);
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw / 4; i += gridDim.x * blockDim.x) {
            const float4 v = p4[i];
            const float a = v.x - K, b = v.y - K, c = v.z - K, d = v.w - K;
            s1 += (a + b) + (c + d);
            s2 = fmaf(a, a, fmaf(b, b, fmaf(c, c, fmaf(d, d, s2))));
        }
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
            const float d = px[i] - K;
            s1 += d;
            s2 = fmaf(d, d, s2);
        }
    }
    __shared__ float sh[2][4];
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh[0][wave] = s1; sh[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.0f, b = 0.0f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += sh[0][w]; b += sh[1][w]; }
        atomicAdd(raw + 2 * plane, a);
        atomicAdd(raw + 2 * plane + 1, b);
    }
}

struct NormApplyParams {
    const float* x;        // [n][c][h][w]
    const float* raw;      // [n*c][2] from inorm_stats_kernel
    const float* gamma;    // [c]
    const float* beta;     // [c]
    const float* res;      // residual, or null; element (n_, c_, y, x) at res[n_ * rs_n + c_ * rs_c + y * rs_h + x]
    float* out;            // [n][c][h + 2 pad][w + 2 pad]; image i at out + i * out_stride_n (the leading channels of a wider buffer)
    float* stats;          // [n*c][2]: mean, rstd (kept for the backward)
    long long rs_n, rs_c, rs_h, out_stride_n;
    int n, c, h, w, pad, act;
    float eps;
};

__device__ __forceinline__ float norm_act_fwd(float v, int act) {
    if (act == NORM_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == NORM_ACT_ELU) return v > 0.0f ? v : expm1f(v);
    return v;
}
// d act / d pre-activation from the OUTPUT z
__device__ __forceinline__ float norm_act_grad(float z, int act) {
    if (act == NORM_ACT_RELU) return z > 0.0f ? 1.0f : 0.0f;
    if (act == NORM_ACT_ELU) return z > 0.0f ? 1.0f : z + 1.0f;
    return 1.0f;
}

// i / d and i % d for 0 <= i < 2^23 through a float reciprocal (no integer division in the element loop)
__device__ __forceinline__ void fast_divmod(int i, int d, float inv_d, int& q, int& r) {
    q = (int)((float)i * inv_d);
    r = i - q * d;
    if (r >= d) { ++q; r -= d; }
    if (r < 0) { --q; r += d; }
}

// grid = (chunks, planes): everything that depends on the plane (mean, 1 / std, gamma, beta, base pointers) is uniform per workgroup
__global__ void __launch_bounds__(256) inorm_apply_kernel(NormApplyParams p) {
    const int plane = blockIdx.y;
    const int hp = p.h + 2 * p.pad, wp = p.w + 2 * p.pad, hw = p.h * p.w;
    const int ch = plane % p.c, img = plane / p.c;
    const float* px = p.x + (size_t)plane * hw;
    const float inv_hw = 1.0f / (float)hw, inv_wp = 1.0f / (float)wp;
    const float m1 = p.raw[2 * plane] * inv_hw, m2 = p.raw[2 * plane + 1] * inv_hw;
    const float mean = px[0] + m1;
    const float rstd = 1.0f / sqrtf(fmaxf(m2 - m1 * m1, 0.0f) + p.eps);
    const float sc = rstd * p.gamma[ch], sh = p.beta[ch] - mean * sc;            // gamma (x - mean) rstd + beta = x sc + sh
    const float* pr = p.res ? p.res + img * p.rs_n + ch * p.rs_c : nullptr;
    float* po = p.out + (size_t)img * p.out_stride_n + (size_t)ch * hp * wp;
    if (blockIdx.x == 0 && threadIdx.x == 0) { p.stats[2 * plane] = mean; p.stats[2 * plane + 1] = rstd; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hp * wp; i += gridDim.x * blockDim.x) {
        int yp, xp;
        fast_divmod(i, wp, inv_wp, yp, xp);
        const int y = reflect_index(yp - p.pad, p.h), xx = reflect_index(xp - p.pad, p.w);
        float v = fmaf(px[y * p.w + xx], sc, sh);
        if (pr) v += pr[y * p.rs_h + xx];
        po[i] = norm_act_fwd(v, p.act);
    }
}

struct NormBwdParams {
    const float* x;        // [n][c][h][w] forward input
    const float* out;      // [n][c][hp][wp] forward output (padded); image i at out + i * out_stride_n
    const float* d_out;    // [n][c][hp][wp] gradient of the padded output; image i at d_out + i * d_out_stride_n
    const float* stats;    // [n*c][2] mean, rstd
    const float* gamma;    // [c]
    float* raw;            // [n*c][2]: sum g, sum g xhat (zeroed; reduce kernel adds, apply kernel reads)
    float* d_gamma;        // [c] or null: sum over the images of the planes' sum g xhat (written by the apply kernel)
    float* d_beta;         // [c] or null: ... of sum g
    float* dx;             // [n][c][h][w]
    float* d_res;          // [n][c][h][w] or null
    long long out_stride_n, d_out_stride_n;
    int n, c, h, w, pad, act;
};

// g(y, x): gradient of the padded output folded back onto the interior point (adjoint of the reflection padding), through the
// activation
__device__ __forceinline__ float norm_bwd_g(const NormBwdParams& p, const float* d_plane, const float* o_plane, int y, int xx) {
    const int wp = p.w + 2 * p.pad;
    float g = d_plane[(y + p.pad) * wp + xx + p.pad];
    if (p.pad > 0 && (y <= p.pad || xx <= p.pad || y >= p.h - 1 - p.pad || xx >= p.w - 1 - p.pad)) {      // near a border only
        // padded rows / columns that mirror onto (y, x): distance k <= pad from an edge mirrors to the row k outside it
        const int ya = (y >= 1 && y <= p.pad) ? p.pad - y : -1;                          // top border row index in the padded tensor
        const int yb = (y <= p.h - 2 && y >= p.h - 1 - p.pad) ? p.pad + 2 * (p.h - 1) - y : -1;
        const int xa = (xx >= 1 && xx <= p.pad) ? p.pad - xx : -1;
        const int xb = (xx <= p.w - 2 && xx >= p.w - 1 - p.pad) ? p.pad + 2 * (p.w - 1) - xx : -1;
        const int ys[3] = {y + p.pad, ya, yb}, xs[3] = {xx + p.pad, xa, xb};
        g = 0.0f;
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                if (ys[a] >= 0 && xs[b] >= 0) g += d_plane[ys[a] * wp + xs[b]];
    }
    return g * norm_act_grad(o_plane[(y + p.pad) * wp + xx + p.pad], p.act);
}

// grid = (chunks, planes)
__global__ void __launch_bounds__(256) inorm_backward_reduce_kernel(NormBwdParams p) {
    const int plane = blockIdx.y, hw = p.h * p.w, hpwp = (p.h + 2 * p.pad) * (p.w + 2 * p.pad);
    const int img = plane / p.c, ch = plane - img * p.c;
    const float* px = p.x + (size_t)plane * hw;
    const float* dpl = p.d_out + (size_t)img * p.d_out_stride_n + (size_t)ch * hpwp;
    const float* opl = p.out + (size_t)img * p.out_stride_n + (size_t)ch * hpwp;
    const float mean = p.stats[2 * plane], rstd = p.stats[2 * plane + 1];
    float s1 = 0.0f, s2 = 0.0f;
    const float inv_w = 1.0f / (float)p.w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        int y, xx;
        fast_divmod(i, p.w, inv_w, y, xx);
        const float g = norm_bwd_g(p, dpl, opl, y, xx);
        s1 += g;
        s2 = fmaf(g, (px[i] - mean) * rstd, s2);
    }
    __shared__ float sh[2][4];
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sh[0][wave] = s1; sh[1][wave] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.0f, b = 0.0f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += sh[0][w]; b += sh[1][w]; }
        atomicAdd(p.raw + 2 * plane, a);
        atomicAdd(p.raw + 2 * plane + 1, b);
    }
}

// grid = (chunks, planes)
__global__ void __launch_bounds__(256) inorm_backward_apply_kernel(NormBwdParams p) {
    const int plane = blockIdx.y, hw = p.h * p.w, hpwp = (p.h + 2 * p.pad) * (p.w + 2 * p.pad);
    const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)p.w;
    const int img = plane / p.c, ch = plane - img * p.c;
    const float* px = p.x + (size_t)plane * hw;
    const float* dpl = p.d_out + (size_t)img * p.d_out_stride_n + (size_t)ch * hpwp;
    const float* opl = p.out + (size_t)img * p.out_stride_n + (size_t)ch * hpwp;
    const float mean = p.stats[2 * plane], rstd = p.stats[2 * plane + 1];
    const float mg = p.raw[2 * plane] * inv_hw, mgx = p.raw[2 * plane + 1] * inv_hw;
    const float gr = p.gamma[ch] * rstd;
    // the affine parameters' gradients: the planes' sums added over the images, in image order, by the first workgroup of image 0's
    // plane (PyTorch: one more reduction kernel per call; same-address atomics from every plane of a channel cost more than that)
    if (img == 0 && blockIdx.x == 0 && threadIdx.x == 0 && p.d_gamma && p.d_beta) {
        float sb = 0.0f, sg = 0.0f;
        for (int i = 0; i < p.n; ++i) { sb += p.raw[2 * (i * p.c + ch)]; sg += p.raw[2 * (i * p.c + ch) + 1]; }
        p.d_beta[ch] = sb;
        p.d_gamma[ch] = sg;
    }
    float* pdx = p.dx + (size_t)plane * hw;
    float* pdr = p.d_res ? p.d_res + (size_t)plane * hw : nullptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        int y, xx;
        fast_divmod(i, p.w, inv_w, y, xx);
        const float g = norm_bwd_g(p, dpl, opl, y, xx);
        const float xhat = (px[i] - mean) * rstd;
        pdx[i] = gr * (g - mg - xhat * mgx);
        if (pdr) pdr[i] = g;
    }
}

// ---- bilinear x2 up-sampling (align_corners) + reflection padding of the decoder half of the image encoder ------------------------
// network/ops.py:150-230 (ResUNetLight.upconv3 / upconv2: F.interpolate(scale_factor=2, mode='bilinear', align_corners=True), then a
// reflect-padded 3 x 3 convolution).  PyTorch: the up-sampling kernel (0.4 TB/s), reflection_pad2d, and in the backward one thread
// per OUTPUT-gradient element with four float atomics each.  Here: one kernel writes the up-sampled plane already padded for the
// convolution, and the backward is a gather - one thread per INPUT element sums the (<= 8 x 8, typically 4 x 4) padded output
// gradients that touch it, from per-axis tables the host builds with the forward's own fp32 arithmetic (source index = scale *
// output index, truncated) - no atomics, deterministic.
struct UpsampleParams {
    const float* x;      // [planes][h][w]
    float* out;          // [planes][2h + 2 pad][2w + 2 pad]
    int planes, h, w, pad;
    float sy, sx;        // (h - 1) / (2h - 1), (w - 1) / (2w - 1) in fp32, computed by the host
};

// grid = (chunks, planes)
__global__ void __launch_bounds__(256) upsample2x_pad_kernel(UpsampleParams p) {
    const int plane = blockIdx.y, H = 2 * p.h, W = 2 * p.w, hp = H + 2 * p.pad, wp = W + 2 * p.pad;
    const float* px = p.x + (size_t)plane * p.h * p.w;
    float* po = p.out + (size_t)plane * hp * wp;
    const float inv_wp = 1.0f / (float)wp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hp * wp; i += gridDim.x * blockDim.x) {
        int yp, xp;
        fast_divmod(i, wp, inv_wp, yp, xp);
        const int oy = reflect_index(yp - p.pad, H), ox = reflect_index(xp - p.pad, W);
        const float fy = p.sy * (float)oy, fx = p.sx * (float)ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
        const float* r0 = px + y0 * p.w;
        const float* r1 = px + y1 * p.w;
        po[i] = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
    }
}

constexpr int kUpTaps = 8;       // padded output rows (columns) that read one input row (column): at most 3 + 3 + 2 mirrored
struct UpsampleBwdParams {
    const float* d_out;  // [planes][hp][wp]
    float* dx;           // [planes][h][w]
    const int* cnt_y;    // [h]
    const int* idx_y;    // [h][kUpTaps] padded output rows
    const float* wgt_y;  // [h][kUpTaps]
    const int* cnt_x;    // [w]
    const int* idx_x;    // [w][kUpTaps]
    const float* wgt_x;  // [w][kUpTaps]
    int planes, h, w, hp, wp;
};

constexpr int kUpRows = 4;       // input rows per workgroup of the backward
// grid = (ceil(h / kUpRows), planes), dynamic LDS kUpRows * wp floats.  Separable: for each of the workgroup's input rows the padded
// output-gradient rows that read it are combined first (threads along the padded width: coalesced reads, the row's table entries are
// uniform), the result goes to LDS, and the columns are gathered from there.
__global__ void __launch_bounds__(256) upsample2x_pad_backward_kernel(UpsampleBwdParams p) {
    NR_DYNAMIC_SMEM(float, vs);
    const int plane = blockIdx.y, y0 = blockIdx.x * kUpRows;
    const float* dpl = p.d_out + (size_t)plane * p.hp * p.wp;
    float* pdx = p.dx + (size_t)plane * p.h * p.w;
    for (int r = 0; r < kUpRows && y0 + r < p.h; ++r) {
        const int y = y0 + r, ny = p.cnt_y[y];
        const int* iy = p.idx_y + y * kUpTaps;
        const float* wy = p.wgt_y + y * kUpTaps;
        for (int ox = threadIdx.x; ox < p.wp; ox += blockDim.x) {
            float v = 0.0f;
            for (int a = 0; a < ny; ++a) v = fmaf(wy[a], dpl[iy[a] * p.wp + ox], v);
            vs[r * p.wp + ox] = v;
        }
    }
    __syncthreads();
    for (int xx = threadIdx.x; xx < p.w; xx += blockDim.x) {
        const int nx = p.cnt_x[xx];
        int ix[kUpTaps];
        float wx[kUpTaps];
        NR_PRAGMA_UNROLL
        for (int b = 0; b < kUpTaps; ++b) { ix[b] = p.idx_x[xx * kUpTaps + b]; wx[b] = p.wgt_x[xx * kUpTaps + b]; }
        for (int r = 0; r < kUpRows && y0 + r < p.h; ++r) {
            const float* row = vs + r * p.wp;
            float acc = 0.0f;
            NR_PRAGMA_UNROLL
            for (int b = 0; b < kUpTaps; ++b)
                if (b < nx) acc = fmaf(wx[b], row[ix[b]], acc);
            pdx[(y0 + r) * p.w + xx] = acc;
        }
    }
}

}  // namespace nr
