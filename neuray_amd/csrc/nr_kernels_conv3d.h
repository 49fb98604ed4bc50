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

// prob: out[n][z][y][x] = bias + sum_{c < 8, taps} w[c][tap] x[n][c][z + dz][y + dy][x + dx], zero padding.  A thread owns one (y, x) column
// and WALKS ALONG z over a segment of kProbSeg planes: every input plane it reads (8 channels x 3 rows x 3 columns = 72 loads, a wave's
// load = 64 consecutive floats of one row) feeds the three outputs z - 1, z, z + 1 through three rotating accumulators, so an output costs 72
// (+ the segment's two halo planes: 81) L1 reads instead of the 216 of one thread per voxel - that version sat on the L1 bandwidth
// (27 x 420 MB per call: 0.57 ms = 0.82 TB/s of compulsory traffic, round 4).  The 216 weights sit in LDS in the order they are used
// ([ky][c][kz][kx]) and are read as 16-byte broadcasts.  Missing neighbours (image borders) are loaded from a clamped address and
// replaced by 0 with a select: no divergent control flow inside the walk.
struct ProbParams {
    const float* x;     // [n][8][D][H][W]
    const float* w;     // [3 ky][8 c][3 kz][3 kx]
    float* out;         // [n][D][H][W]
    int n, d, h, w_;
    float bias;
};

constexpr int kProbSeg = 16;        // output planes per thread

// one row slot of one input plane: tap kz = 2 of output z - 1, kz = 1 of output z, kz = 0 of output z + 1.  ws: the row slot's 72 weights
__device__ __forceinline__ void prob_row(const float* ws, const float* __restrict__ row, long long vol, int dl, int dr, bool okl, bool okm, bool okr,
                                         float& acc_m, float& acc_c, float& acc_p) {
    NR_PRAGMA_UNROLL
    for (int c = 0; c < 8; ++c) {
        const float* q = row + c * vol;
        const float lv = q[dl], mv = q[0], rv = q[dr];
        const float l = okl ? lv : 0.0f, m = okm ? mv : 0.0f, r = okr ? rv : 0.0f;
        float w[12];                                           // [kz][kx] + 3 unused (three 16-byte broadcast reads)
        NR_PRAGMA_UNROLL
        for (int i = 0; i < 3; ++i) {
            const float4 t = ld4(ws + 12 * c + 4 * i);
            w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
        }
        acc_p = fmaf(w[0], l, acc_p); acc_p = fmaf(w[1], m, acc_p); acc_p = fmaf(w[2], r, acc_p);
        acc_c = fmaf(w[3], l, acc_c); acc_c = fmaf(w[4], m, acc_c); acc_c = fmaf(w[5], r, acc_c);
        acc_m = fmaf(w[6], l, acc_m); acc_m = fmaf(w[7], m, acc_m); acc_m = fmaf(w[8], r, acc_m);
    }
}

// grid = ceil(h * w / 256) * ceil(d / kProbSeg) * n
__global__ void __launch_bounds__(256) costreg_prob_kernel(ProbParams p) {
    __shared__ __attribute__((aligned(16))) float ws[3 * 8 * 12];             // [ky][c][9 weights + 3 pad]
    for (int i = threadIdx.x; i < 3 * 8 * 12; i += blockDim.x) ws[i] = (i % 12) < 9 ? p.w[(i / 12) * 9 + i % 12] : 0.0f;
    __syncthreads();
    const int plane = p.h * p.w_, chunks = (plane + (int)blockDim.x - 1) / (int)blockDim.x, segs = (p.d + kProbSeg - 1) / kProbSeg;
    int b = (int)blockIdx.x;
    const int chunk = b % chunks;
    b /= chunks;
    const int seg = b % segs, img = b / segs;
    const int pix0 = chunk * (int)blockDim.x + (int)threadIdx.x;
    const bool pvalid = pix0 < plane;
    const int pix = pvalid ? pix0 : plane - 1;                 // (no early exit: every lane runs the same walk)
    const int y = pix / p.w_, x = pix - y * p.w_;
    const int z0 = seg * kProbSeg, z1 = (z0 + kProbSeg < p.d) ? z0 + kProbSeg : p.d;      // outputs [z0, z1)
    const long long vol = (long long)p.d * plane;
    const float* base = p.x + (long long)img * 8 * vol + pix;
    const bool xl = x > 0, xr = x + 1 < p.w_, yu = y > 0, yd = y + 1 < p.h;
    const int dl = xl ? -1 : 0, dr = xr ? 1 : 0, dy0 = yu ? -p.w_ : 0, dy2 = yd ? p.w_ : 0;
    // acc_m: output zi - 1 (complete after this plane), acc_c: output zi, acc_p: output zi + 1
    float acc_m = p.bias, acc_c = p.bias, acc_p = p.bias;
    float* o = p.out + (long long)img * vol + pix;
#pragma unroll 1
    for (int zi = z0 - 1; zi <= z1; ++zi) {
        if (zi >= 0 && zi < p.d) {                             // uniform
            const float* pl = base + (long long)zi * plane;
            // (an address the compiler cannot see through, re-made per plane: the weight reads are loop invariant and would otherwise be
            // hoisted out of the walk - 216 registers; the sched barriers keep one row slot's reads from being issued ahead of another's)
            const float* wz = ws + nr_opaque_zero();
            prob_row(wz, pl + dy0, vol, dl, dr, xl && yu, yu, xr && yu, acc_m, acc_c, acc_p); NR_PIN();
            prob_row(wz + 96, pl, vol, dl, dr, xl, true, xr, acc_m, acc_c, acc_p); NR_PIN();
            prob_row(wz + 192, pl + dy2, vol, dl, dr, xl && yd, yd, xr && yd, acc_m, acc_c, acc_p); NR_PIN();
        }
        if (zi - 1 >= z0 && pvalid) o[(long long)(zi - 1) * plane] = acc_m;
        acc_m = acc_c; acc_c = acc_p; acc_p = p.bias;
    }
}

// up11: the last decoder step of the cost regularisation, c0 + ConvTranspose3d(16 -> 8, 3, stride 2, padding 1, output_padding 1) + frozen
// batch norm + leaky ReLU (network/mvsnet/mvsnet.py:57-69 conv11 and the skip add; MIOpen: 2.9 ms for the transposed convolution to
// 8 x 8 x 64 x 160 x 160 plus three element-wise passes over that 420 MB tensor).  11 GFLOP against 0.95 GB of compulsory traffic: memory
// bound; one kernel that reads x and c0 once and writes the sum once.  Output o reads input i through tap k where o = 2 i - 1 + k: an even
// o has one tap (k = 1, i = o / 2), an odd o two (k = 2 at i = (o - 1) / 2, k = 0 at i = (o + 1) / 2 if that exists).
// Round 5: a thread owns the 2 x 2 x 2 OUTPUT BLOCK of input voxel (iz, iy, j) - outputs (2 iz + a, 2 iy + b, 2 j + c) - and all 8 output
// channels: it reads the 8 input voxels (iz .. iz + 1, iy .. iy + 1, j .. j + 1) of each of the 16 input channels once and spends each of the
// 27 taps exactly once per output channel (216 FMAs per input channel: the work is the same for every thread, where the round-4 mapping -
// a thread per output pair of one (z, y) row, row groups of one parity per workgroup - ran workgroups of 1, 2 and 4 tap pairs side by
// side and read every input row up to four times).  Threads are laid end to end over a plane's (iy, j), so a wave's loads are 64
// consecutive floats of each of the 2 x 2 (z, y) neighbours; the weights are uniform per instruction: 16-byte LDS broadcasts (as
// scalar operands, 216 per input channel against ~100 scalar registers, hipcc spilled them to VGPR lanes inside the loop).
struct Up11Params {         // (C_in, C_out) = (16, 8): conv11, or (32, 16): conv9 (the same kernel one level down, round 5)
    const float* x;        // [n][C_in][d][h][w]
    const float* wpack;    // [3 kz][3 ky][C_in ci][C_out co][3 kx], batch norm folded
    const float* bias;     // [C_out]
    const float* skip;     // [n][C_out][2d][2h][2w] or null
    float* out;            // [n][C_out][2d][2h][2w]
    int n, d, h, w;
    float slope;
};

#ifndef NR_UP11_CO
#define NR_UP11_CO 4               // output channels per thread: 4 = two threads per input voxel (blockIdx.y = the channel half), 32 accumulators, 121 VGPRs,
                                   // 4 waves per SIMD: 0.42 ms on 8 x 64 x 160 x 160; 8 = one thread per voxel, 230 VGPRs, 2 waves per SIMD: 0.55 ms
#endif
constexpr int kUp11Co = NR_UP11_CO;
// one (kz, ky) tap pair of one input channel: output parity a = (KZ != 1), read from input plane iz + (KZ == 0); the same along y.
// w: the pair's weights [co][kx] of this thread's channels in LDS (16-byte broadcast reads)
template <int KZ, int KY>
__device__ __forceinline__ void up11_tap(const float* w, const float (&v)[2][2][2], float (&acc)[2][2][2][kUp11Co]) {
    constexpr int A = KZ != 1, DA = KZ == 0, B = KY != 1, DB = KY == 0;
    const float x0 = v[DA][DB][0], x1 = v[DA][DB][1];
    float wr[3 * kUp11Co];
    NR_PRAGMA_UNROLL
    for (int i = 0; i < 3 * kUp11Co / 4; ++i) {
        const float4 t = ld4(w + 4 * i);
        wr[4 * i] = t.x; wr[4 * i + 1] = t.y; wr[4 * i + 2] = t.z; wr[4 * i + 3] = t.w;
    }
    NR_PRAGMA_UNROLL
    for (int co = 0; co < kUp11Co; ++co) {
        const float w0 = wr[3 * co], w1 = wr[3 * co + 1], w2 = wr[3 * co + 2];
        acc[A][B][0][co] = fmaf(x0, w1, acc[A][B][0][co]);                          // even output: kx = 1 at j
        acc[A][B][1][co] = fmaf(x0, w2, fmaf(x1, w0, acc[A][B][1][co]));            // odd output: kx = 2 at j, kx = 0 at j + 1
    }
}

// grid = ceil(h w / 256) * d * n
#ifndef NR_UP11_MINW
#define NR_UP11_MINW 4
#endif
template <int CIN, int COUT>
__global__ void __launch_bounds__(256, NR_UP11_MINW) costreg_up11_kernel(Up11Params p) {
    constexpr int WB = COUT * 3;           // weights of one (tap pair, input channel): [co][kx]
    __shared__ __attribute__((aligned(16))) float wl[9 * CIN * WB];           // the folded weights (13.5 KB at 16 -> 8, 54 KB at 32 -> 16): 54 B per
    for (int i = threadIdx.x; i < 9 * CIN * WB / 4; i += blockDim.x)          // thread of a workgroup whose threads each spend them on 1728 multiply-adds
        reinterpret_cast<float4*>(wl)[i] = reinterpret_cast<const float4*>(p.wpack)[i];
    __syncthreads();
    const int plane = p.h * p.w, chunks = (plane + (int)blockDim.x - 1) / (int)blockDim.x;
    int b = (int)blockIdx.x;
    const int chunk = b % chunks;
    b /= chunks;
    const int iz = b % p.d, img = b / p.d;
    const int pix = chunk * (int)blockDim.x + (int)threadIdx.x;
    if (pix >= plane) return;
    const int iy = pix / p.w, j = pix - iy * p.w;
    const long long vol = (long long)plane * p.d;
    // neighbours that do not exist (last plane / row / column) are read at the voxel itself and multiplied by 0
    const bool zn = iz + 1 < p.d, yn = iy + 1 < p.h, xn = j + 1 < p.w;
    const long long dz = zn ? plane : 0;
    const int dy = yn ? p.w : 0, dx = xn ? 1 : 0;
    const float fz = zn ? 1.0f : 0.0f, fy = yn ? 1.0f : 0.0f, fx = xn ? 1.0f : 0.0f;
    const int cog = (int)blockIdx.y * kUp11Co;                 // first output channel of this thread
    float acc[2][2][2][kUp11Co];
    NR_PRAGMA_UNROLL
    for (int a = 0; a < 2; ++a)
        NR_PRAGMA_UNROLL
        for (int b_ = 0; b_ < 2; ++b_)
            NR_PRAGMA_UNROLL
            for (int c = 0; c < 2; ++c)
                NR_PRAGMA_UNROLL
                for (int co = 0; co < kUp11Co; ++co) acc[a][b_][c][co] = 0.0f;
    const float* xi = p.x + (long long)img * CIN * vol + (long long)iz * plane + pix;
    const float f01 = fx, f10 = fy, f11 = fy * fx, g00 = fz, g01 = fz * fx, g10 = fz * fy, g11 = fz * fy * fx;
    // the 8 inputs of channel ci + 1 are loaded while channel ci is being spent (one memory round trip ahead): with the loads at the top
    // of their own iteration hipcc holds a tap's weights in registers until the value they multiply arrives - 230 VGPRs
    float raw[8];
    auto issue = [&](int ci) {
        const float* s_ = xi + ci * vol;
        raw[0] = s_[0]; raw[1] = s_[dx]; raw[2] = s_[dy]; raw[3] = s_[dy + dx];
        raw[4] = s_[dz]; raw[5] = s_[dz + dx]; raw[6] = s_[dz + dy]; raw[7] = s_[dz + dy + dx];
    };
    issue(0);
#pragma unroll 1
    for (int ci = 0; ci < CIN; ++ci) {
        float v[2][2][2];
        v[0][0][0] = raw[0];        v[0][0][1] = raw[1] * f01;
        v[0][1][0] = raw[2] * f10;  v[0][1][1] = raw[3] * f11;
        v[1][0][0] = raw[4] * g00;  v[1][0][1] = raw[5] * g01;
        v[1][1][0] = raw[6] * g10;  v[1][1][1] = raw[7] * g11;
        NR_PRAGMA_UNROLL
        for (int i = 0; i < 8; ++i) NR_KEEP(v[i >> 2][(i >> 1) & 1][i & 1]);
        issue(ci + 1 < CIN ? ci + 1 : CIN - 1);
        const float* wk = wl + ci * WB + 3 * cog;                                   // + (kz * 3 + ky) * CIN * WB
        constexpr int TS = CIN * WB;
        // (sched barriers: left alone the scheduler issues all 54 weight reads of the channel up front - 216 registers)
        up11_tap<0, 0>(wk + 0 * TS, v, acc); NR_PIN(); up11_tap<0, 1>(wk + 1 * TS, v, acc); NR_PIN(); up11_tap<0, 2>(wk + 2 * TS, v, acc); NR_PIN();
        up11_tap<1, 0>(wk + 3 * TS, v, acc); NR_PIN(); up11_tap<1, 1>(wk + 4 * TS, v, acc); NR_PIN(); up11_tap<1, 2>(wk + 5 * TS, v, acc); NR_PIN();
        up11_tap<2, 0>(wk + 6 * TS, v, acc); NR_PIN(); up11_tap<2, 1>(wk + 7 * TS, v, acc); NR_PIN(); up11_tap<2, 2>(wk + 8 * TS, v, acc); NR_PIN();
    }
    const long long orow = 2 * p.w, oplane = 4 * (long long)plane, ovol = 8 * vol;
    const long long o000 = (long long)img * COUT * ovol + (long long)(2 * iz) * oplane + (long long)(2 * iy) * orow + 2 * j;
    NR_PRAGMA_UNROLL
    for (int co_ = 0; co_ < kUp11Co; ++co_) {
        const int co = cog + co_;
        const float bc = p.bias[co];
        NR_PRAGMA_UNROLL
        for (int a = 0; a < 2; ++a)
            NR_PRAGMA_UNROLL
            for (int b_ = 0; b_ < 2; ++b_) {
                const long long o = o000 + co * ovol + a * oplane + b_ * orow;
                float v0 = acc[a][b_][0][co_] + bc, v1 = acc[a][b_][1][co_] + bc;
                v0 = v0 > 0.0f ? v0 : v0 * p.slope;
                v1 = v1 > 0.0f ? v1 : v1 * p.slope;
                if (p.skip) { const float2 sk = *reinterpret_cast<const float2*>(p.skip + o); v0 += sk.x; v1 += sk.y; }
                *reinterpret_cast<float2*>(p.out + o) = make_float2(v0, v1);
            }
    }
}

// The interior layers of the U-Net's encoder half - conv1 (8 -> 16, stride 2), conv2 (16 -> 16), conv3 (16 -> 32, stride 2), conv4 (32 -> 32):
// ConvBnReLU3D(C_in, C_out, 3, stride, pad 1), network/mvsnet/mvsnet.py:29-69, modules.py:16-23 - as an implicit GEMM on the fp32 MFMA with
// the frozen batch norm folded into weights and bias and the leaky ReLU in the epilogue (MIOpen / CK: 0.95 + 1.54 + 0.30 + 0.51 ms per
// 8 x 800 x 800 at 20-35 TFLOP/s, plus two element-wise passes each).  NCDHW in and out (the neighbouring layers are MIOpen's).
//   A = weights (M = 16 output channels per tile, MT = C_out / 16 tiles), N = 16 consecutive OUTPUT voxels along x, K = 4 input channels
//   of one tap per MFMA: lane (column c, group g) supplies channel 4 q + g at input voxel S (x0 + c) + dx - 1.  A wave owns kC3Rows = 4
//   output rows of one x-strip: per (dz, channel quad q) it loads the S * 3 + 3 input rows it needs ONCE -
//     stride 1: one dword per lane (+ the voxel just outside the strip on lanes 0 / 15), the dx = 0 / 2 operands by DPP row shifts;
//     stride 2: the even and the odd voxel of the lane's pair (dx = 1 / 2), the dx = 0 operand = the odd voxel of the lane to the left -
//   and spends them on 3 dy x 3 dx x 4 rows x MT MFMAs.  Out-of-volume rows / voxels are buffer loads beyond the descriptor's range
//   (0 = the zero padding).
//   wpack [3 dz][C_in / 4 q][3 dy][3 dx][MT][64 lanes]: lane (m = l & 15, g = l >> 4) -> W[16 mt + m][4 q + g][dz][dy][dx] * scale[16 mt + m].
// Measured (tools/time_costreg.py, two views per call): conv2 0.452 -> 0.062 ms = 92 TFLOP/s = 0.58 of the fp32 MFMA peak, conv4 0.166 ->
// 0.042 ms; conv6 (64 -> 64 on 8 x 25 x 25: 224 wave tasks) is slower than the library's kernel and stays with it.
struct Conv3dParams {
    const float* x;        // [n][C_in][D][H][W]
    const float* wpack;
    const float* bias;     // [COUT] (batch norm folded; padded to the kernel's COUT)
    float* out;            // [n][C_out][OD][OH][OW], O* = (* - 1) / S + 1
    int n, d, h, w;        // input volume
    float slope;
    int cin, cout;         // the tensors' channel counts: <= the kernel's CIN / COUT (a multiple of 4 / 16); the pack carries zeros for the rest
};

constexpr int kC3Rows = 4, kC3Waves = 4;

template <int CIN, int COUT, int S>
__global__ void __launch_bounds__(64 * kC3Waves) conv3d_kernel(Conv3dParams p) {
    constexpr int MT = COUT / 16, NQ = CIN / 4, NY = kC3Rows, NR = S * (NY - 1) + 3;
    static_assert(CIN % 4 == 0 && COUT % 16 == 0 && (S == 1 || S == 2), "shape");
    const int lane = threadIdx.x & 63;
    const int wave = NR_UNIFORM((int)(threadIdx.x >> 6));
    const int c = lane & 15, g = lane >> 4;
    const int od = (p.d - 1) / S + 1, oh = (p.h - 1) / S + 1, ow = (p.w - 1) / S + 1;
    const long long plane = (long long)p.h * p.w, vol = plane * p.d, oplane = (long long)oh * ow, ovol = oplane * od;
    const int sx = (ow + 15) / 16, gy = (oh + NY - 1) / NY;
    const long long tasks = (long long)p.n * od * gy * sx;
    const nr_wbuf WP = nr_make_wbuf(p.wpack, sizeof(float) * 27 * NQ * MT * 64);
    for (long long s = (long long)blockIdx.x * kC3Waves + wave; s < tasks; s += (long long)gridDim.x * kC3Waves) {
        const int xs = (int)(s % sx);
        long long t = s / sx;
        const int y0 = NY * (int)(t % gy);
        t /= gy;
        const int z = (int)(t % od), img = (int)(t / od);
        const int x = xs * 16 + c;                                        // output column
        // input columns: S = 1: centre xi, edge = the voxel just outside the strip (lanes 0 / 15); S = 2: the pair (xi, xi + 1), edge = xi - 1 (lane 0)
        const int xi = S * x;
        const int xe = S == 1 ? (c == 0 ? xi - 1 : (c == 15 ? xi + 1 : -1)) : (c == 0 ? xi - 1 : -1);
        const nr_mbuf X = nr_make_mbuf(p.x + (size_t)img * p.cin * vol, sizeof(float) * p.cin * (size_t)vol);
        v4f acc[NY][MT];
        NR_PRAGMA_UNROLL
        for (int oy = 0; oy < NY; ++oy)
            NR_PRAGMA_UNROLL
            for (int mt = 0; mt < MT; ++mt) { acc[oy][mt][0] = 0.0f; acc[oy][mt][1] = 0.0f; acc[oy][mt][2] = 0.0f; acc[oy][mt][3] = 0.0f; }
        for (int dz = 0; dz < 3; ++dz) {
            const int zz = S * z + dz - 1;
            if (zz < 0 || zz >= p.d) continue;                           // wave-uniform: the plane lies in the zero padding
#pragma unroll 1
            for (int q = 0; q < NQ; ++q) {
                const long long chan = ((long long)(4 * q + g) * p.d + zz) * plane;
                const bool chok = 4 * q + g < p.cin;                      // (a padded input channel: reads 0, its weights are 0 as well)
                float b0[NR], b1[NR], b2[NR];                            // the dx = 0 / 1 / 2 operands of input row r
                NR_PRAGMA_UNROLL
                for (int r = 0; r < NR; ++r) {
                    const int yy = S * y0 + r - 1;
                    const bool ok = chok && yy >= 0 && yy < p.h;          // (row test: uniform)
                    const long long row = chan + (long long)yy * p.w;
                    const int voc = (ok && xi < p.w) ? (int)((row + xi) * 4) : 0x7ffffff0;          // (past the buffer's range: reads 0)
                    const int voe = (ok && xe >= 0 && xe < p.w) ? (int)((row + xe) * 4) : 0x7ffffff0;
                    const float ctr = nr_buf_ld1(X, voc, 0), be = nr_buf_ld1(X, voe, 0);
                    if constexpr (S == 1) {
                        b1[r] = ctr;
                        b0[r] = nr_row_from_left(ctr, be);
                        b2[r] = nr_row_from_right(ctr, be);
                    } else {
                        const int voo = (ok && xi + 1 < p.w) ? (int)((row + xi + 1) * 4) : 0x7ffffff0;
                        const float odd = nr_buf_ld1(X, voo, 0);
                        b1[r] = ctr;
                        b2[r] = odd;
                        b0[r] = nr_row_from_left(odd, be);
                    }
                }
                const int wbase = ((dz * NQ + q) * 9) * MT * 64 + lane;
                NR_PRAGMA_UNROLL
                for (int dy = 0; dy < 3; ++dy)
                    NR_PRAGMA_UNROLL
                    for (int dx = 0; dx < 3; ++dx) {
                        float a[MT];
                        NR_PRAGMA_UNROLL
                        for (int mt = 0; mt < MT; ++mt) a[mt] = nr_buf_ld1(WP, (wbase + ((dy * 3 + dx) * MT + mt) * 64) * 4, 0);
                        NR_PRAGMA_UNROLL
                        for (int oy = 0; oy < NY; ++oy) {
                            const float b = dx == 0 ? b0[S * oy + dy] : (dx == 1 ? b1[S * oy + dy] : b2[S * oy + dy]);
                            NR_PRAGMA_UNROLL
                            for (int mt = 0; mt < MT; ++mt) acc[oy][mt] = nr_mfma16(a[mt], b, acc[oy][mt]);
                        }
                    }
            }
        }
        // D layout: lane (column c, group g), register r of tile mt = output channel 16 mt + 4 g + r of output voxel x0 + c
        if (x < ow) {
            NR_PRAGMA_UNROLL
            for (int mt = 0; mt < MT; ++mt) {
                const float4 b4 = ld4(p.bias + 16 * mt + 4 * g);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
                NR_PRAGMA_UNROLL
                for (int oy = 0; oy < NY; ++oy) {
                    const int y = y0 + oy;
                    if (y >= oh) continue;
                    float* o = p.out + ((long long)img * p.cout + 16 * mt + 4 * g) * ovol + (long long)z * oplane + (long long)y * ow + x;
                    NR_PRAGMA_UNROLL
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[oy][mt][r] + bb[r];
                        if (16 * mt + 4 * g + r < p.cout) o[r * ovol] = v > 0.0f ? v : v * p.slope;
                    }
                }
            }
        }
    }
}

// Frozen activated batch norm of MVSNet (inplace_abn.ABN in evaluation mode: network/mvsnet/modules.py:7-23 `self.bn(self.conv(x))`,
// mvsnet.py:7-69): y = leaky_relu(x * scale[c] + shift[c]) with scale = gamma / sqrt(var + eps), shift = beta - mean * scale, IN PLACE on the
// convolution's output [n][c][inner] (inner = h w or d h w) - one read and one write per element where batch_norm + leaky_relu are two of
// each (the 2-D feature net runs its first layers on 8 x 800 x 800 x 12 views: 0.9 of its 2.7 ms were these passes).  HBM bound.
struct ScaleShiftLeakyParams {
    float* x;              // [n][c][inner], in place
    const float* scale;    // [c]
    const float* shift;    // [c]
    long long inner;
    int n, c;
    float slope;
};

// grid = (chunks, n * c): a workgroup stays inside one (image, channel) plane, so scale / shift are scalar loads
__global__ void __launch_bounds__(256) scale_shift_leaky_kernel(ScaleShiftLeakyParams p) {
    const int plane = blockIdx.y, ch = plane % p.c;
    const float a = p.scale[ch], b = p.shift[ch], sl = p.slope;
    float* px = p.x + (size_t)plane * p.inner;
    const long long stride = (long long)gridDim.x * blockDim.x;
    if ((p.inner & 3) == 0) {
        float4* p4 = reinterpret_cast<float4*>(px);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.inner / 4; i += stride) {
            float4 v = p4[i];
            v.x = fmaf(v.x, a, b); v.y = fmaf(v.y, a, b); v.z = fmaf(v.z, a, b); v.w = fmaf(v.w, a, b);
            v.x = v.x > 0.0f ? v.x : v.x * sl; v.y = v.y > 0.0f ? v.y : v.y * sl;
            v.z = v.z > 0.0f ? v.z : v.z * sl; v.w = v.w > 0.0f ? v.w : v.w * sl;
            p4[i] = v;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.inner; i += stride) {
            const float v = fmaf(px[i], a, b);
            px[i] = v > 0.0f ? v : v * sl;
        }
    }
}

// ---- the variance volume in the layout conv0 reads, second mapping (round 5) --------------------------------------------------------
// warp_variance_kernel (nr_kernels.h) gives a voxel to ONE thread, which walks the 32 channels of each of its 4 x n_num taps as eight
// 16-byte loads: one load instruction of a wave touches 64 different 128-byte lines and uses an eighth of each, and a line has to survive
// in the 32 KB L1 until the eighth pass over it (a wave's taps alone are ~17 KB per source view).  Here EIGHT consecutive lanes own a
// voxel and lane q of them the channels 4 q .. 4 q + 3: a tap is one 128-byte line read once, by one instruction, whole; the sums are
// 8 registers instead of 64 and the variance leaves as one 128-byte line per voxel.  With the loads in order the kernel is bound by its
// VALU work (the IEEE divisions of grid_sample's un-normalisation: ~120 instructions per source view), so the eight lanes of a voxel do
// not repeat it: lane q evaluates the homography, the taps and the bilinear weights of source view q, and the group reads them from
// that lane (eight ds_bpermute per source view).  Same expressions, same roundings as warp_variance_kernel: the results are
// bit-identical - every channel's sums are formed by the same operations in the same order (division by V = n_num + 1 is a
// multiplication when V is a power of two, which is exact).  Reference view and depth plane are the workgroup's (blockIdx.y), so the
// 3 x 4 transforms and the depth are scalar loads.  grid = (ceil(fh fw / 32), rfn dn), 256 threads: a workgroup takes 32 consecutive
// pixels of its plane.  (A 16 x 16 pixel tile per workgroup, walked two rows at a time so that a row's source texels are met again by
// the next row while still in L1 - 1.85 x fewer lines from L2 - measured the same: 1.564 vs 1.554 ms,
// profiles/r05_z_warpvar_kernel_ab.log.  The kernel is not waiting for L2.  Nor does it want its loads earlier: an instantiation for
// three source views with all twelve taps in flight before the first is spent needs 102 VGPRs instead of 58 - four waves per SIMD
// instead of eight - and takes 1.27 ms where this one takes 1.12, profiles/r05_z_warpvar_kernel_ab3.log.)
__global__ void __launch_bounds__(256) warp_variance_cl_kernel(WarpVarParams p) {
    const int hw = p.fh * p.fw;
    const int plane = blockIdx.y, r = plane / p.dn;
    const int lane = threadIdx.x & 63, q = lane & 7, group0 = lane & ~7;
    const int pix_raw = (int)blockIdx.x * 32 + (int)(threadIdx.x >> 3);
    const int pix = pix_raw < hw ? pix_raw : hw - 1;           // (every lane stays in for the exchanges; a padding voxel is not stored)
    const int y = pix / p.fw, x = pix - y * p.fw;
    const float wm1 = (float)(p.fw - 1), hm1 = (float)(p.fh - 1);
    const float half_w = rn_div(wm1, 2.0f), half_h = rn_div(hm1, 2.0f);
    const float dv = p.depth_vals[plane];
    const float gx = rn_mul((float)x, dv), gy = rn_mul((float)y, dv), gz = dv;
    float4 sum = reinterpret_cast<const float4*>(p.ref_feats + ((long long)r * hw + pix) * 32)[q];
    float4 sq = make_float4(sum.x * sum.x, sum.y * sum.y, sum.z * sum.z, sum.w * sum.w);
    for (int j0 = 0; j0 < p.n_num; j0 += 8) {
        // this lane's source view of the round (lanes beyond n_num repeat the last one: no divergence, their values are not read)
        const int jm = j0 + q < p.n_num ? j0 + q : p.n_num - 1;
        const float* M = p.transforms + ((long long)r * p.n_num + jm) * 12;
        const float X = rn_add(dot3(M[0], M[1], M[2], gx, gy, gz), M[3]);
        const float Y = rn_add(dot3(M[4], M[5], M[6], gx, gy, gz), M[7]);
        float Z = rn_add(dot3(M[8], M[9], M[10], gx, gy, gz), M[11]);
        if (Z < 1e-4f) Z = 1e-4f;
        // grid_sample(align_corners=True) un-normalisation of  g = s / ((size-1)/2) - 1
        const float ix = rn_mul(rn_div(rn_add(rn_sub(rn_div(rn_div(X, Z), half_w), 1.0f), 1.0f), 2.0f), wm1);
        const float iy = rn_mul(rn_div(rn_add(rn_sub(rn_div(rn_div(Y, Z), half_h), 1.0f), 1.0f), 2.0f), hm1);
        const float x0f = floorf(ix), y0f = floorf(iy);
        const float wx1 = ix - x0f, wy1 = iy - y0f, wx0 = (x0f + 1.0f) - ix, wy0 = (y0f + 1.0f) - iy;
        // (NaN / huge coordinates fail every bounds test below: zero contribution, as in grid_sample)
        const bool okx0 = x0f >= 0.0f && x0f <= wm1, okx1 = x0f + 1.0f >= 0.0f && x0f + 1.0f <= wm1;
        const bool oky0 = y0f >= 0.0f && y0f <= hm1, oky1 = y0f + 1.0f >= 0.0f && y0f + 1.0f <= hm1;
        const int x0 = okx0 ? (int)x0f : 0, x1 = okx1 ? (int)x0f + 1 : 0, y0 = oky0 ? (int)y0f : 0, y1 = oky1 ? (int)y0f + 1 : 0;
        const float mw00 = (okx0 && oky0) ? wx0 * wy0 : 0.0f, mw10 = (okx1 && oky0) ? wx1 * wy0 : 0.0f;
        const float mw01 = (okx0 && oky1) ? wx0 * wy1 : 0.0f, mw11 = (okx1 && oky1) ? wx1 * wy1 : 0.0f;
        const int mo00 = (y0 * p.fw + x0) * 8, mo10 = (y0 * p.fw + x1) * 8, mo01 = (y1 * p.fw + x0) * 8, mo11 = (y1 * p.fw + x1) * 8;
        const int nj = p.n_num - j0 < 8 ? p.n_num - j0 : 8;
        for (int jj = 0; jj < nj; ++jj) {
            const int from = group0 + jj;
            const float w00 = __shfl(mw00, from), w10 = __shfl(mw10, from), w01 = __shfl(mw01, from), w11 = __shfl(mw11, from);
            const int o00 = __shfl(mo00, from), o10 = __shfl(mo10, from), o01 = __shfl(mo01, from), o11 = __shfl(mo11, from);
            const float4* m = reinterpret_cast<const float4*>(p.src_feats + (long long)p.nn_ids[r * p.n_num + j0 + jj] * hw * 32) + q;
            const float4 a = m[o00], b = m[o10], c = m[o01], e = m[o11];
            const float v0 = a.x * w00 + b.x * w10 + c.x * w01 + e.x * w11;
            const float v1 = a.y * w00 + b.y * w10 + c.y * w01 + e.y * w11;
            const float v2 = a.z * w00 + b.z * w10 + c.z * w01 + e.z * w11;
            const float v3 = a.w * w00 + b.w * w10 + c.w * w01 + e.w * w11;
            sum.x += v0; sum.y += v1; sum.z += v2; sum.w += v3;
            sq.x += v0 * v0; sq.y += v1 * v1; sq.z += v2 * v2; sq.w += v3 * v3;
        }
    }
    if (pix_raw >= hw) return;
    const int vi = p.n_num + 1;
    const float V = (float)vi;
    float4 o;
    if ((vi & (vi - 1)) == 0) {               // V = 2^k: x / V = x * (1 / V), both exact up to the same single rounding
        const float iv = 1.0f / V;
        const float m0 = rn_mul(sum.x, iv), m1 = rn_mul(sum.y, iv), m2 = rn_mul(sum.z, iv), m3 = rn_mul(sum.w, iv);
        o = make_float4(rn_sub(rn_mul(sq.x, iv), rn_mul(m0, m0)), rn_sub(rn_mul(sq.y, iv), rn_mul(m1, m1)),
                        rn_sub(rn_mul(sq.z, iv), rn_mul(m2, m2)), rn_sub(rn_mul(sq.w, iv), rn_mul(m3, m3)));
    } else {
        const float m0 = rn_div(sum.x, V), m1 = rn_div(sum.y, V), m2 = rn_div(sum.z, V), m3 = rn_div(sum.w, V);
        o = make_float4(rn_sub(rn_div(sq.x, V), rn_mul(m0, m0)), rn_sub(rn_div(sq.y, V), rn_mul(m1, m1)),
                        rn_sub(rn_div(sq.z, V), rn_mul(m2, m2)), rn_sub(rn_div(sq.w, V), rn_mul(m3, m3)));
    }
    reinterpret_cast<float4*>(p.out + ((long long)plane * hw + pix) * 32)[q] = o;
}

}  // namespace nr
