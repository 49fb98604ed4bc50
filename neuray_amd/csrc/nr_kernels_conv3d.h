// The two 3-D convolutions of MVSNet's cost regularisation that MIOpen runs far off any roofline (SURVEY.md 8(f) f-3;
// reference network/mvsnet/mvsnet.py:29-69 CostRegNet, frozen and evaluation-only inside CostVolumeInitNet, network/init_net.py:113-160):
//
//   conv0   ConvBnReLU3D(32 -> 8, 3 x 3 x 3, pad 1) on the plane-sweep variance volume [n, 32, D, H, W] at full volume resolution:
//           68 % of the U-Net's MACs.  MIOpen: 23.1 ms for 8 x 64 x 160 x 160 (7.8 TFLOP/s); here an implicit GEMM on the fp32 MFMA,
//           the frozen batch norm folded into weights and bias, leaky ReLU in the epilogue.
//   prob    Conv3d(8 -> 1, 3 x 3 x 3, pad 1): 216 MACs per voxel over a 420 MB input - memory bound; MIOpen: 6.2 ms (0.9 TFLOP/s).
//
// conv0 mapping (v_mfma_f32_16x16x4_f32, the layout conventions of nr_layout.h): the weights are the A operand - M = output channel
// (8 real rows, 8 zero rows) -, 16 consecutive voxels along x are the N columns, K = 27 taps x 32 channels.  The input is channels-last
// ([n][D][H][W][32], written that way by warp_variance_kernel): a voxel's channels are one 128-byte line, lane group g loads channels
// 8 g .. 8 g + 7 of its column's voxel as two 16-byte loads per tap and supplies channel 8 g + j in K-step j - the point kernel's
// "gathered order".  The packed weights (27 taps x 2 quads x 64 lanes x float4 = 55 KB) live in LDS for the whole launch.  Out-of-volume
// taps are buffer loads beyond the descriptor's range: they return 0, which is the zero padding.  (Two strips per wave sharing every A
// fragment - two accumulator chains - measured the same: 13.3 vs 13.1 ms for the U-Net; not kept.)
#pragma once
#include "nr_device.h"

namespace nr {

struct Conv0Params {
    const float* x;        // [n][D][H][W][32]
    const float* wpack;    // [27][2][64] float4: tap (kz, ky, kx), quad q, lane (m = l & 15, g = l >> 4), component i -> W'[m][8 g + 4 q + i][tap] (0 for m >= 8)
    const float* bias;     // [8] (batch norm folded)
    float* out;            // [n][8][D][H][W]
    int n, d, h, w;
    float slope;           // leaky ReLU
};

constexpr int kConv0Waves = 8;
constexpr int kConv0PackFloats = 27 * 2 * 64 * 4;

__global__ void __launch_bounds__(64 * kConv0Waves) costreg_conv0_kernel(Conv0Params p) {
    __shared__ __attribute__((aligned(16))) float wl[kConv0PackFloats];
    const int lane = threadIdx.x & 63;
    const int wave = NR_UNIFORM((int)(threadIdx.x >> 6));
    for (int i = threadIdx.x; i < kConv0PackFloats / 4; i += blockDim.x)
        reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(p.wpack)[i];
    __syncthreads();
    const int c = lane & 15, g = lane >> 4;
    const long long vol = (long long)p.d * p.h * p.w;
    const int sx = (p.w + 15) / 16;                                  // strips of 16 voxels per row
    const long long strips = (long long)p.n * p.d * p.h * sx;
    const float4* wq = reinterpret_cast<const float4*>(wl) + lane + nr_opaque_zero();
    const float b0 = p.bias[(4 * g + 0) & 7], b1 = p.bias[(4 * g + 1) & 7], b2 = p.bias[(4 * g + 2) & 7], b3 = p.bias[(4 * g + 3) & 7];
    for (long long s = (long long)blockIdx.x * kConv0Waves + wave; s < strips; s += (long long)gridDim.x * kConv0Waves) {
        const int xs = (int)(s % sx);
        long long t = s / sx;
        const int y = (int)(t % p.h);
        t /= p.h;
        const int z = (int)(t % p.d), img = (int)(t / p.d);
        const int x = xs * 16 + c;
        const nr_mbuf X = nr_make_mbuf(p.x + (size_t)img * vol * 32, sizeof(float) * 32 * (size_t)vol);      // this image's volume (< 2^31 bytes)
        v4f acc;
        acc[0] = 0.0f; acc[1] = 0.0f; acc[2] = 0.0f; acc[3] = 0.0f;
        NR_PRAGMA_UNROLL
        for (int kz = 0; kz < 3; ++kz) {
            const int zz = z + kz - 1;
            NR_PRAGMA_UNROLL
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = y + ky - 1;
                const bool row_ok = zz >= 0 && zz < p.d && yy >= 0 && yy < p.h;          // wave-uniform
                const long long row = ((long long)zz * p.h + yy) * p.w;
                float4 q0[3], q1[3];
                NR_PRAGMA_UNROLL
                for (int kx = 0; kx < 3; ++kx) {
                    const int xx = x + kx - 1;
                    // (a byte offset past the buffer's range reads as 0: the zero padding)
                    const int voff = (row_ok && xx >= 0 && xx < p.w) ? (int)((row + xx) * 128 + 32 * g) : 0x7ffffff0;
                    q0[kx] = mld4(X, voff, 0);
                    q1[kx] = mld4(X, voff, 16);
                }
                NR_PRAGMA_UNROLL
                for (int kx = 0; kx < 3; ++kx) {
                    const int tap = (kz * 3 + ky) * 3 + kx;
                    const float4 a0 = wq[(tap * 2 + 0) * 64], a1 = wq[(tap * 2 + 1) * 64];
                    acc = nr_mfma16(a0.x, q0[kx].x, acc); acc = nr_mfma16(a0.y, q0[kx].y, acc);
                    acc = nr_mfma16(a0.z, q0[kx].z, acc); acc = nr_mfma16(a0.w, q0[kx].w, acc);
                    acc = nr_mfma16(a1.x, q1[kx].x, acc); acc = nr_mfma16(a1.y, q1[kx].y, acc);
                    acc = nr_mfma16(a1.z, q1[kx].z, acc); acc = nr_mfma16(a1.w, q1[kx].w, acc);
                }
            }
        }
        if (g < 2 && x < p.w) {                                     // D layout: lane (column c, group g), register r = output channel 4 g + r
            float* o = p.out + ((long long)img * 8 + 4 * g) * vol + ((long long)z * p.h + y) * p.w + x;
            const float v0 = acc[0] + b0, v1 = acc[1] + b1, v2 = acc[2] + b2, v3 = acc[3] + b3;
            o[0] = v0 > 0.0f ? v0 : v0 * p.slope;
            o[vol] = v1 > 0.0f ? v1 : v1 * p.slope;
            o[2 * vol] = v2 > 0.0f ? v2 : v2 * p.slope;
            o[3 * vol] = v3 > 0.0f ? v3 : v3 * p.slope;
        }
    }
}

// prob: out[n][z][y][x] = bias + sum_{c < 8, taps} w[c][tap] x[n][c][z + dz][y + dy][x + dx], zero padding.  One thread per voxel, x fastest:
// every load of a warp is 64 consecutive floats of one row.
struct ProbParams {
    const float* x;     // [n][8][D][H][W]
    const float* w;     // [8][27]
    float* out;         // [n][D][H][W]
    int n, d, h, w_;
    float bias;
};

__global__ void __launch_bounds__(256) costreg_prob_kernel(ProbParams p) {
    __shared__ float ws[8 * 27];
    for (int i = threadIdx.x; i < 8 * 27; i += blockDim.x) ws[i] = p.w[i];
    __syncthreads();
    const long long vol = (long long)p.d * p.h * p.w_, total = vol * p.n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % p.w_);
        long long t = i / p.w_;
        const int y = (int)(t % p.h);
        t /= p.h;
        const int z = (int)(t % p.d), img = (int)(t / p.d);
        float acc = p.bias;
        for (int c = 0; c < 8; ++c) {
            const float* base = p.x + ((long long)img * 8 + c) * vol;
            NR_PRAGMA_UNROLL
            for (int kz = 0; kz < 3; ++kz) {
                const int zz = z + kz - 1;
                NR_PRAGMA_UNROLL
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = y + ky - 1;
                    const bool ok = zz >= 0 && zz < p.d && yy >= 0 && yy < p.h;
                    const float* row = base + ((long long)(ok ? zz : z) * p.h + (ok ? yy : y)) * p.w_;
                    const float l = (ok && x > 0) ? row[x - 1] : 0.0f, m = ok ? row[x] : 0.0f, r = (ok && x + 1 < p.w_) ? row[x + 1] : 0.0f;
                    const float* wk = ws + c * 27 + (kz * 3 + ky) * 3;
                    acc = fmaf(wk[0], l, acc); acc = fmaf(wk[1], m, acc); acc = fmaf(wk[2], r, acc);
                }
            }
        }
        p.out[i] = acc;
    }
}

}  // namespace nr
