// The two 3-D convolutions of MVSNet's cost regularisation that MIOpen runs far off any roofline (SURVEY.md 8(f) f-3;
// reference network/mvsnet/mvsnet.py:29-69 CostRegNet, frozen and evaluation-only inside CostVolumeInitNet, network/init_net.py:113-160):
//
//   conv0   ConvBnReLU3D(32 -> 8, 3 x 3 x 3, pad 1) on the plane-sweep variance volume [n, 32, D, H, W] at full volume resolution:
//           68 % of the U-Net's MACs.  MIOpen: 23.1 ms for 8 x 64 x 160 x 160 (7.8 TFLOP/s); here an implicit GEMM on the fp32 MFMA,
//           the frozen batch norm folded into weights and bias, leaky ReLU in the epilogue.
//   prob    Conv3d(8 -> 1, 3 x 3 x 3, pad 1): 216 MACs per voxel over a 420 MB input - memory bound; MIOpen: 6.2 ms (0.9 TFLOP/s).
//
// conv0 mapping (v_mfma_f32_16x16x4_f32, the layout conventions of nr_layout.h): the weights are the A operand, 16 consecutive voxels
// along x are the N columns, K = taps x 32 channels.  M = 16 rows = 8 output channels x TWO output rows (y, y + 1) of the same
// x-strip: an input row yy at depth zz feeds output row y through tap ky = yy - y + 1 and output row y + 1 through tap ky - 1, so the A
// fragment of "row slot" r = yy - y + 1 in 0 .. 3 carries W[kz][r] in its rows 0 .. 7 and W[kz][r - 1] in its rows 8 .. 15 (zero where
// that tap does not exist): 36 (kz, r, kx) slots for two output rows instead of 2 x 27 half-empty ones - 1.5 x fewer MFMAs - and four
// input rows loaded per kz for two output rows instead of six.  The three kx taps of a row slot are ONE load: the input is
// channels-last ([n][D][H][W][32], written that way by warp_variance_kernel), lane (column c, group g) loads channels 8 g .. 8 g + 7 of
// voxel x0 + c as two 16-byte loads and supplies channel 8 g + j in K-step j (the point kernel's "gathered order"); the kx = 0 / 2
// operands are the same registers moved one lane inside the 16-lane group (DPP row shift), and the two voxels just outside the strip
// come with one more load pair that only lanes c = 0 and c = 15 aim at memory.  12 load pairs + 12 edge pairs per two output rows where
// the first version issued 108.  The packed weights (36 slots x 2 quads x 64 lanes x float4 = 72 KB) live in LDS for the whole launch
// (two workgroups per CU).  Out-of-volume rows / voxels are buffer loads beyond the descriptor's range: they return 0, the zero padding;
// a row slot that lies outside the volume is skipped (wave-uniform).
// (First version, one output row per wave and a load per tap: 5.0 ms for 8 x 64 x 160 x 160, 0.46 of its own MFMA bound; two strips per
// wave sharing every A fragment measured the same and was not kept.)
#pragma once
#include "nr_device.h"

namespace nr {

struct Conv0Params {
    const float* x;        // [n][D][H][W][32]
    const float* wpack;    // [3 kz][4 r][3 kx][2 q][64 lanes] float4: lane (m = l & 15, g = l >> 4), component i -> channel 8 g + 4 q + i of
                           // W'[m][.][kz][r][kx] (m < 8, r <= 2) / W'[m - 8][.][kz][r - 1][kx] (m >= 8, r >= 1), else 0
    const float* bias;     // [8] (batch norm folded)
    float* out;            // [n][8][D][H][W]
    int n, d, h, w;
    float slope;           // leaky ReLU
};

constexpr int kConv0Waves = 8;
constexpr int kConv0Slots = 3 * 4 * 3;
constexpr int kConv0PackFloats = kConv0Slots * 2 * 64 * 4;

__global__ void __launch_bounds__(64 * kConv0Waves) costreg_conv0_kernel(Conv0Params p) {
    NR_DYNAMIC_SMEM(float, wl);
    const int lane = threadIdx.x & 63;
    const int wave = NR_UNIFORM((int)(threadIdx.x >> 6));
    for (int i = threadIdx.x; i < kConv0PackFloats / 4; i += blockDim.x)
        reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(p.wpack)[i];
    __syncthreads();
    const int c = lane & 15, g = lane >> 4;
    const long long vol = (long long)p.d * p.h * p.w;
    const int sx = (p.w + 15) / 16, hy = (p.h + 1) / 2;              // strips of 16 voxels per row, pairs of rows
    const long long strips = (long long)p.n * p.d * hy * sx;
    const float4* wq = reinterpret_cast<const float4*>(wl) + lane + nr_opaque_zero();
    const float b0 = p.bias[(4 * g + 0) & 7], b1 = p.bias[(4 * g + 1) & 7], b2 = p.bias[(4 * g + 2) & 7], b3 = p.bias[(4 * g + 3) & 7];
    for (long long s = (long long)blockIdx.x * kConv0Waves + wave; s < strips; s += (long long)gridDim.x * kConv0Waves) {
        const int xs = (int)(s % sx);
        long long t = s / sx;
        const int y = 2 * (int)(t % hy);
        t /= hy;
        const int z = (int)(t % p.d), img = (int)(t / p.d);
        const int x = xs * 16 + c;
        const int xe = c == 0 ? x - 1 : (c == 15 ? x + 1 : -1);       // the voxel just outside the strip (lanes 0 / 15 only)
        const nr_mbuf X = nr_make_mbuf(p.x + (size_t)img * vol * 32, sizeof(float) * 32 * (size_t)vol);      // this image's volume (< 2^31 bytes)
        v4f acc;
        acc[0] = 0.0f; acc[1] = 0.0f; acc[2] = 0.0f; acc[3] = 0.0f;
        // row slot `slot` = kz * 4 + r: its four loads (centre pair, edge pair) are issued one slot ahead of the MFMAs that consume them
        auto slot_ok = [&](int slot) {
            const int zz = z + slot / 4 - 1, yy = y + slot % 4 - 1;
            return zz >= 0 && zz < p.d && yy >= 0 && yy < p.h;          // wave-uniform; false: the slot lies in the zero padding
        };
        auto issue = [&](int slot, float4 (&q)[4]) {
            const int zz = z + slot / 4 - 1, yy = y + slot % 4 - 1;
            const long long row = ((long long)zz * p.h + yy) * p.w;
            const bool ok = slot_ok(slot);
            // (a byte offset past the buffer's range reads as 0: the zero padding)
            const int voc = (ok && x < p.w) ? (int)((row + x) * 128 + 32 * g) : 0x7ffffff0;
            const int voe = (ok && xe >= 0 && xe < p.w) ? (int)((row + xe) * 128 + 32 * g) : 0x7ffffff0;
            q[0] = mld4(X, voc, 0); q[1] = mld4(X, voc, 16); q[2] = mld4(X, voe, 0); q[3] = mld4(X, voe, 16);
        };
        float4 cur[4], nxt[4];
        issue(0, cur);
        NR_PRAGMA_UNROLL
        for (int slot = 0; slot < 12; ++slot) {
            if (slot + 1 < 12) issue(slot + 1, nxt);
            if (slot_ok(slot)) {
                const float bc[8] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w};
                const float be[8] = {cur[2].x, cur[2].y, cur[2].z, cur[2].w, cur[3].x, cur[3].y, cur[3].z, cur[3].w};
                float bl[8], br[8];
                NR_PRAGMA_UNROLL
                for (int j = 0; j < 8; ++j) { bl[j] = nr_row_from_left(bc[j], be[j]); br[j] = nr_row_from_right(bc[j], be[j]); }
                NR_PRAGMA_UNROLL
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 a0 = wq[((slot * 3 + kx) * 2 + 0) * 64], a1 = wq[((slot * 3 + kx) * 2 + 1) * 64];
                    const float* b = kx == 0 ? bl : (kx == 1 ? bc : br);
                    acc = nr_mfma16(a0.x, b[0], acc); acc = nr_mfma16(a0.y, b[1], acc);
                    acc = nr_mfma16(a0.z, b[2], acc); acc = nr_mfma16(a0.w, b[3], acc);
                    acc = nr_mfma16(a1.x, b[4], acc); acc = nr_mfma16(a1.y, b[5], acc);
                    acc = nr_mfma16(a1.z, b[6], acc); acc = nr_mfma16(a1.w, b[7], acc);
                }
            }
            NR_PRAGMA_UNROLL
            for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
        }
        const int yo = y + (g >> 1);                                  // D layout: lane (column c, group g), register r = row 4 g + r: output row y + g / 2, channel 4 (g & 1) + r
        if (x < p.w && yo < p.h) {
            float* o = p.out + ((long long)img * 8 + 4 * (g & 1)) * vol + ((long long)z * p.h + yo) * p.w + x;
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

// up11: the last decoder step of the cost regularisation, c0 + ConvTranspose3d(16 -> 8, 3, stride 2, padding 1, output_padding 1) + frozen
// batch norm + leaky ReLU (network/mvsnet/mvsnet.py:57-69 conv11 and the skip add; MIOpen: 2.9 ms for the transposed convolution to
// 8 x 8 x 64 x 160 x 160 plus three element-wise passes over that 420 MB tensor).  11 GFLOP against 0.95 GB of compulsory traffic: memory
// bound; one kernel that reads x and c0 once and writes the sum once.  Output o reads input i through tap k where o = 2 i - 1 + k: an even
// o has one tap (k = 1, i = o / 2), an odd o two (k = 2 at i = (o - 1) / 2, k = 0 at i = (o + 1) / 2 if that exists).  A thread owns the
// output pair (2 j, 2 j + 1) of one (z, y) row - uniform work: x[j] feeds both, x[j + 1] the odd one - and all 8 output channels.  A
// workgroup takes kUp11Rows output rows of one plane that share their y parity (oy, oy + 2, ...), laid end to end over its threads: which
// z / y taps exist, and their weights, are then uniform per workgroup - the weights are scalar loads, no LDS - and every lane has work
// whatever the row length.  (First version, one row per workgroup with the 14 KB of weights copied into LDS by each: 2.1 ms.)
struct Up11Params {
    const float* x;        // [n][16][d][h][w]
    const float* wpack;    // [3 kz][3 ky][16 ci][8 co][3 kx], batch norm folded
    const float* bias;     // [8]
    const float* skip;     // [n][8][2d][2h][2w] or null
    float* out;            // [n][8][2d][2h][2w]
    int n, d, h, w;
    float slope;
};

constexpr int kUp11Rows = 4;

// grid = (chunks of 256 threads over kUp11Rows * w) * (row groups per plane = 2 parities * ceil(h / kUp11Rows)) * 2d * n
__global__ void __launch_bounds__(256) costreg_up11_kernel(Up11Params p) {
    const int per_group = kUp11Rows * p.w, chunks = (per_group + (int)blockDim.x - 1) / (int)blockDim.x;
    const int groups_y = (p.h + kUp11Rows - 1) / kUp11Rows;                   // row groups of one parity
    long long g = (long long)blockIdx.x / chunks;
    const int t = ((int)(blockIdx.x % chunks)) * (int)blockDim.x + (int)threadIdx.x;
    const int gy = (int)(g % groups_y);
    g /= groups_y;
    const int par = (int)(g & 1);                                             // y parity of the group's rows
    g >>= 1;
    const int oz = (int)(g % (2 * p.d)), img = (int)(g / (2 * p.d));
    const int r = t / p.w, j = t - r * p.w;
    const int hy = gy * kUp11Rows + r;                                        // the row's input row index (oy = 2 hy + par)
    if (r >= kUp11Rows || hy >= p.h) return;
    const int oy = 2 * hy + par;
    const long long plane = (long long)p.h * p.w, vol = plane * p.d;
    // taps: index 0 = the tap every output has, index 1 = the second tap of an odd output (uniform per workgroup; the last odd row /
    // plane has no second tap: its loads are redirected to a valid address and multiplied by 0)
    const int nz = (oz & 1) ? 2 : 1, ny = par ? 2 : 1;
    const int iz0 = (oz & 1) ? (oz - 1) / 2 : oz / 2, kz0 = (oz & 1) ? 2 : 1;
    const int ky0 = par ? 2 : 1;
    const bool has_next = j + 1 < p.w;
    float a0[8], a1[8];
    NR_PRAGMA_UNROLL
    for (int co = 0; co < 8; ++co) { a0[co] = 0.0f; a1[co] = 0.0f; }
    const float* xi = p.x + (long long)img * 16 * vol + j;
    for (int tz = 0; tz < nz; ++tz) {
        const int iz = tz ? iz0 + 1 : iz0, kz = tz ? 0 : kz0;
        if (iz >= p.d) continue;                                              // uniform
        for (int ty = 0; ty < ny; ++ty) {
            const int iy = ty ? hy + 1 : hy, ky = ty ? 0 : ky0;
            const bool row_ok = iy < p.h;
            const float* src = xi + (long long)iz * plane + (long long)(row_ok ? iy : hy) * p.w;
            const float live = row_ok ? 1.0f : 0.0f;
            const float* __restrict__ wk = p.wpack + (kz * 3 + ky) * (16 * 8 * 3);   // uniform: scalar loads
            // (two input channels at a time: their 48 weights fit the scalar registers; unrolled over all 16 the 384 of them spilled)
#pragma unroll 2
            for (int ci = 0; ci < 16; ++ci) {
                const float xa = src[ci * vol] * live, xb = has_next ? src[ci * vol + 1] * live : 0.0f;
                NR_PRAGMA_UNROLL
                for (int co = 0; co < 8; ++co) {
                    const float w0 = wk[ci * 24 + 3 * co], w1 = wk[ci * 24 + 3 * co + 1], w2 = wk[ci * 24 + 3 * co + 2];
                    a0[co] = fmaf(xa, w1, a0[co]);                            // even output: kx = 1 at j
                    a1[co] = fmaf(xa, w2, fmaf(xb, w0, a1[co]));              // odd output: kx = 2 at j, kx = 0 at j + 1
                }
            }
        }
    }
    const long long oplane = 4 * plane, ovol = 8 * vol;
    const long long o = (long long)img * 8 * ovol + (long long)oz * oplane + (long long)oy * (2 * p.w) + 2 * j;
    NR_PRAGMA_UNROLL
    for (int co = 0; co < 8; ++co) {
        float v0 = a0[co] + p.bias[co], v1 = a1[co] + p.bias[co];
        v0 = v0 > 0.0f ? v0 : v0 * p.slope;
        v1 = v1 > 0.0f ? v1 : v1 * p.slope;
        if (p.skip) { const float2 sk = *reinterpret_cast<const float2*>(p.skip + o + co * ovol); v0 += sk.x; v1 += sk.y; }
        *reinterpret_cast<float2*>(p.out + o + co * ovol) = make_float2(v0, v1);
    }
}

}  // namespace nr
