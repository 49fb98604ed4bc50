// 3 x 3 stride-1 convolutions of the per-image encoders (reference network/ops.py:86-148,150-230 ResUNetLight's BasicBlock /
// conv layers, network/vis_encoder.py:6-21, the res_net of network/init_net.py:13-61) at fp32 grade on the K = 32 bf16 MFMA: the arithmetic of
// the point kernel's AR_X3 (nr_layout.h).  Every operand is split EXACTLY into three bf16 parts x = h + m + l; a product is the six
// MFMAs lh, hl, mm, mh, hm, hh (smallest first, fp32 accumulate; the three dropped terms are below 2^-23 of the product) = 6 x 16 cycles per
// 16 x 16 x 32 block where the fp32 MFMA spends 8 x 32.  Weights are split once per call by conv2d_x3_pack_kernel, activations once per element
// when a workgroup stages them to LDS - not per tap, and not per output tile.
//
// Index space.  NCHW in and out (the layout of the fused norm kernels around these layers, nr_kernels_norm.h).  The batch is ONE tall image of
// n * hp rows (hp = h + 2 pad: the input with `pad` rings of zeros, pad = 0 for the encoders' pre-padded activations, 2 for the data gradient =
// the full correlation with the flipped kernel), cut into column bands of tw output columns; inside a band the rows are lw = tw + 2 wide and a
// position is q = R * lw + x.  The output at q reads the inputs q + dy * lw + dx, so in q the 9 taps are pure shifts: an MFMA column tile is 16
// CONSECUTIVE q (rows wrap inside a tile), and the positions x >= tw / the last two rows of every image are computed and dropped (2 / lw of the
// work with lw = 52 on the 50-, 100- and 200-pixel maps of an 800 x 800 view: no per-image or per-row rounding to 16).
//   workgroup = 4 waves x NT tiles = 64 NT consecutive q  x  16 MTW output channels (blockIdx.y: channel group, blockIdx.z: band); with
//               WC = 2 (layers of 128 output channels) eight waves: two channel groups share the staged block (half the staging per output)
//   LDS       = the 64 NT + 2 lw + 2 input positions the workgroup reads, per 32-channel block: [3 parts][position][32 channels bf16]
//               (a position's 64 bytes are 4 lane groups x 8 channels: the B operand of tile t for tap (dy, dx) is ONE ds_read_b128 per part
//               at position t * 16 + c + dy * lw + dx - a contiguous 1 KB per wave, no bank conflicts)
//   A operand = wpack [tap][kb][mt][part][lane][4 dwords] straight from L2 / L1 (the four waves read the same 1 KB pieces at the same time),
//               reused over the wave's NT tiles.
#pragma once
#include "nr_platform.h"

namespace nr {

constexpr int kC2Waves = 4;
constexpr int conv2d_x3_positions(int nt, int lw) { return (kC2Waves * nt * 16 + 2 * lw + 2 + 63) / 64 * 64; }      // whole staging passes
constexpr int conv2d_x3_smem_bytes(int nt, int lw) { return conv2d_x3_positions(nt, lw) * 64 * 3; }
constexpr int conv2d_x3_max_passes(int nt) { return nt <= 4 ? 7 : 12; }   // staging passes of 64 positions a kernel instance holds in registers
#ifndef NR_C2_PROBE
#define NR_C2_PROBE 0                 // timing probes with WRONG results: 1 no staging, 2 no A loads, 3 no B loads, 4 no MFMAs
#endif

// w [cout][cin][3][3] (fp32) -> wpack.  Lane l = (m = l & 15, g = l >> 4) of tile mt holds, in dword d of part p, the bf16 pair of input channels
// 32 kb + 8 g + 2 d (+ 1) of output channel 16 mt + m.
struct Conv2dPackParams {
    const float* w;         // [cout][cin][3][3]: the layer's weight
    unsigned* wpack;        // the layer's pack, or null
    unsigned* wpack_t;      // the pack of the layer's data gradient (W'[ci][co][2 - dy][2 - dx]: a convolution from cout to cin channels), or null
    int cout, cin;
};

__device__ __forceinline__ void conv2d_x3_pack_one(const float* w, int ld_o, int ld_k, int flip, int kbn, int mtn, int i, unsigned* out) {
    // element (o, k, tap) of the packed matrix = w[o * ld_o + k * ld_k + (flip ? 8 - tap : tap)]
    const int d = i & 3, l = (i >> 2) & 63;
    int t = i >> 8;
    const int mt = t % mtn;
    t /= mtn;
    const int kb = t % kbn, tap = t / kbn;
    const int o = 16 * mt + (l & 15), k = 32 * kb + 8 * (l >> 4) + 2 * d;
    const int tp = flip ? 8 - tap : tap;
    unsigned h, m, lo;
    nr_split3(w[(size_t)o * ld_o + (size_t)k * ld_k + tp], w[(size_t)o * ld_o + (size_t)(k + 1) * ld_k + tp], h, m, lo);
    const size_t base = ((((size_t)tap * kbn + kb) * mtn + mt) * 3) * 256 + l * 4 + d;
    out[base] = h;
    out[base + 256] = m;
    out[base + 512] = lo;
}

// one thread per (tap, kb, mt, lane, d) of either pack (both have 9 * cin * cout / 2 dwords per part)
__global__ void __launch_bounds__(256) conv2d_x3_pack_kernel(Conv2dPackParams p) {
    const int total = 9 * (p.cin / 32) * (p.cout / 16) * 256;
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= total) return;
    if (p.wpack) conv2d_x3_pack_one(p.w, p.cin * 9, 9, 0, p.cin / 32, p.cout / 16, i, p.wpack);
    if (p.wpack_t) conv2d_x3_pack_one(p.w, 9, p.cin * 9, 1, p.cout / 32, p.cin / 16, i, p.wpack_t);
}

struct Conv2dX3Params {
    const float* x;          // [n][cin][h][w]
    const unsigned* wpack;   // conv2d_x3_pack_kernel
    const float* bias;       // [cout] or null
    float* out;              // [n][cout][oh][ow], oh = h + 2 pad - 2, ow = w + 2 pad - 2
    int n, cin, cout, h, w, pad;
    int tw;                  // output columns per band (blockIdx.z); lw = tw + 2
};

template <int NT, int MTW, int WC = 1>
__global__ void __launch_bounds__(64 * kC2Waves * WC) conv2d_x3_kernel(Conv2dX3Params p) {
    constexpr int MAXP = (conv2d_x3_max_passes(NT) + WC - 1) / WC;
    NR_DYNAMIC_SMEM(unsigned char, lds);
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave_all = NR_UNIFORM(tid >> 6);
    const int wave = wave_all & 3, wave_c = wave_all >> 2;         // position quarter, output-channel group of the workgroup
    const int c = lane & 15, g = lane >> 4;
    const int hp = p.h + 2 * p.pad, oh = hp - 2, ow = p.w + 2 * p.pad - 2;
    const int lw = p.tw + 2, band_x0 = (int)blockIdx.z * p.tw;
    const int kbn = p.cin / 32, mtn = p.cout / 16, mt0 = ((int)blockIdx.y * WC + wave_c) * MTW;
    const int rows = p.n * hp;                                     // (the host checks n * hp * lw < 2^31)
    const int q0 = (int)blockIdx.x * (kC2Waves * NT * 16);
    const int px = conv2d_x3_positions(NT, lw), part_bytes = px * 64;
    const size_t plane = (size_t)p.h * p.w;
    const nr_mbuf X = nr_make_mbuf(p.x, sizeof(float) * (size_t)p.n * p.cin * plane);
    const nr_wbuf W = nr_make_wbuf(reinterpret_cast<const float*>(p.wpack), (size_t)9 * kbn * mtn * 3 * 1024);

    // staging: thread (slot = tid >> 2, octet = tid & 3) moves channels 8 octet .. + 7 of position pass * 64 + slot.  The source offset of
    // channel 0 of the block (bytes; out of the padded image: past the buffer's range = reads 0) is the same for every channel block.
    const int slot = tid >> 2, oct = tid & 3;
    const int npass = px / (64 * WC);
    int src[MAXP];
    NR_PRAGMA_UNROLL
    for (int ps = 0; ps < MAXP; ++ps) {
        src[ps] = -1;
        const unsigned q = (unsigned)(q0 + ps * 64 * WC + slot);
        const int R = (int)(q / (unsigned)lw);
        const int xx = (int)q - R * lw;
        if (ps < npass && R < rows) {
            const int img = (int)((unsigned)R / (unsigned)hp);
            const int y = R - img * hp - p.pad, xs = band_x0 + xx - p.pad;
            if (y >= 0 && y < p.h && xs >= 0 && xs < p.w) src[ps] = (int)((((size_t)img * p.cin) * plane + (size_t)y * p.w + xs) * 4);
        }
    }
    const int plane_b = (int)(plane * 4);
    // the loads of ALL passes of a channel block are issued back to back (one round of memory latency per block), and the next block's
    // while this one is being multiplied
    float v[MAXP][8];
    auto fetch = [&](int kb, int ps0, int ps1) NR_LAMBDA_INLINE {
        const int koff = (32 * kb + 8 * oct) * plane_b;
        NR_PRAGMA_UNROLL
        for (int ps = 0; ps < MAXP; ++ps) {
            if (ps < ps0 || ps >= ps1) continue;                     // (wave-uniform)
            NR_PRAGMA_UNROLL
            for (int j = 0; j < 8; ++j) v[ps][j] = (NR_C2_PROBE == 1) ? 0.0f : nr_buf_ld1(X, src[ps] >= 0 ? src[ps] + koff + j * plane_b : 0x7ffffff0, 0);
        }
    };
    auto split_store = [&]() NR_LAMBDA_INLINE {
        NR_PRAGMA_UNROLL
        for (int ps = 0; ps < MAXP; ++ps) {
            if (ps >= npass) break;
            nr_v4u ph, pm, pl;
            NR_PRAGMA_UNROLL
            for (int d = 0; d < 4; ++d) {
                unsigned a, b, e;
                nr_split3(v[ps][2 * d], v[ps][2 * d + 1], a, b, e);
                ph[d] = a; pm[d] = b; pl[d] = e;
            }
            unsigned char* dst = lds + (ps * 64 * WC + slot) * 64 + oct * 16;
            *reinterpret_cast<nr_v4u*>(dst) = ph;
            *reinterpret_cast<nr_v4u*>(dst + part_bytes) = pm;
            *reinterpret_cast<nr_v4u*>(dst + 2 * part_bytes) = pl;
        }
    };

    v4f acc[NT][MTW];
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t)
        NR_PRAGMA_UNROLL
        for (int mt = 0; mt < MTW; ++mt) { acc[t][mt][0] = 0.0f; acc[t][mt][1] = 0.0f; acc[t][mt][2] = 0.0f; acc[t][mt][3] = 0.0f; }

    // A operands: one (step s = kb * 9 + tap, tile mt) fragment triple at a time, fetched one micro-step ahead into the other register set
    const int steps = kbn * 9;
    auto load_a = [&](int s, int mt, nr_v4u (&A)[3]) NR_LAMBDA_INLINE {
        const int kb = s / 9, tap = s - 9 * kb;
        const int wb = (((tap * kbn + kb) * mtn + mt0 + mt) * 3) * 1024 + lane * 16;
        NR_PRAGMA_UNROLL
        for (int pt = 0; pt < 3; ++pt) {
#if NR_C2_PROBE == 2
            A[pt] = nr_v4u{(unsigned)(wb + mt), (unsigned)pt, 1u, 2u};
#else
            A[pt] = nr_buf_ld4u(W, s < steps ? wb + pt * 1024 : 0x7ffffff0, 0);
#endif
        }
    };
    const int rd0 = (wave * NT * 16 + c) * 64 + g * 16;            // the wave's tile 0, tap (0, 0)
    auto load_b = [&](int tap, nr_v4u (&B)[NT][3]) NR_LAMBDA_INLINE {
        const int dy = tap / 3, dx = tap - 3 * dy;
        const unsigned char* rb = lds + rd0 + (dy * lw + dx) * 64;
        NR_PRAGMA_UNROLL
        for (int t = 0; t < NT; ++t)
            NR_PRAGMA_UNROLL
            for (int pt = 0; pt < 3; ++pt) {
#if NR_C2_PROBE == 3
                B[t][pt] = nr_v4u{(unsigned)(t + tap), (unsigned)pt, 1u, 2u};
#else
                B[t][pt] = *reinterpret_cast<const nr_v4u*>(rb + t * 1024 + pt * part_bytes);
#endif
            }
    };

    static_assert(MTW % 2 == 0, "the A register sets alternate per output-channel tile");
    nr_v4u A0[3], A1[3];
    int kb = 0, tap = 0, s = 0;
    fetch(0, 0, MAXP);
    load_a(0, 0, A0);
    constexpr int PPT = (MAXP + 8) / 9;                              // staging passes fetched per tap
    // products smallest first: (weight part, activation part) = (l, h) (h, l) (m, m) (m, h) (h, m) (h, h)
    auto products = [&](const nr_v4u (&A)[3], const nr_v4u (&B)[NT][3], int mt) NR_LAMBDA_INLINE {
        constexpr int WI[6] = {2, 0, 1, 1, 0, 0}, XJ[6] = {0, 2, 1, 0, 1, 0};
        NR_PRAGMA_UNROLL
        for (int pr = 0; pr < 6; ++pr)
            NR_PRAGMA_UNROLL
            for (int t = 0; t < NT; ++t) {
#if NR_C2_PROBE == 4
                acc[t][mt][pr & 3] += __builtin_bit_cast(float, A[WI[pr]][0] ^ B[t][XJ[pr]][1]);
#else
                acc[t][mt] = nr_mfma16x32_bf16(A[WI[pr]], B[t][XJ[pr]], acc[t][mt]);
#endif
            }
    };
#pragma unroll 1
    for (; s < steps; ++s) {
        if (tap == 0) {
            if (kb) NR_BLOCK_SYNC();                                // (everyone is done reading the previous block)
            split_store();
            NR_BLOCK_SYNC();
        }
        nr_v4u B[NT][3];
        load_b(tap, B);
        NR_PRAGMA_UNROLL
        for (int mt = 0; mt < MTW; mt += 2) {
            load_a(s, mt + 1, A1);
            // the next block's activations, a pass or two per tap: behind the A loads a step is about to wait for (vmcnt counts in order)
            if (mt == 0 && kb + 1 < kbn) fetch(kb + 1, tap * PPT, tap * PPT + PPT);
            products(A0, B, mt);
            if (mt + 2 < MTW) load_a(s, mt + 2, A0);
            else load_a(s + 1, 0, A0);
            products(A1, B, mt + 1);
        }
        if (++tap == 9) { tap = 0; ++kb; }
    }

    // D layout: lane (column c, group g), register r of tile mt = output channel 16 (mt0 + mt) + 4 g + r at position q0 + (wave NT + t) 16 + c
    float bias[MTW][4];
    NR_PRAGMA_UNROLL
    for (int mt = 0; mt < MTW; ++mt)
        NR_PRAGMA_UNROLL
        for (int r = 0; r < 4; ++r) bias[mt][r] = p.bias ? p.bias[16 * (mt0 + mt) + 4 * g + r] : 0.0f;
    const size_t oplane = (size_t)oh * ow;
    NR_PRAGMA_UNROLL
    for (int t = 0; t < NT; ++t) {
        const unsigned q = (unsigned)(q0 + (wave * NT + t) * 16 + c);
        const int R = (int)(q / (unsigned)lw);
        const int xx = (int)q - R * lw;
        const int img = (int)((unsigned)R / (unsigned)hp);
        const int y = R - img * hp;
        if (R >= rows || xx >= p.tw || band_x0 + xx >= ow || y >= oh) continue;
        float* o = p.out + (((size_t)img * p.cout + 16 * mt0 + 4 * g) * oh + y) * ow + band_x0 + xx;
        NR_PRAGMA_UNROLL
        for (int mt = 0; mt < MTW; ++mt)
            NR_PRAGMA_UNROLL
            for (int r = 0; r < 4; ++r) o[(size_t)(16 * mt + r) * oplane] = acc[t][mt][r] + bias[mt][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layers: dW[co][ci][dy][dx] = sum over (image, y < oh, x < ow) of dY[co][y][x] * xp[ci][y + dy][x + dx], in the same
// split arithmetic.  In NCHW the contraction index - positions - is the contiguous one, so BOTH operands come straight from global memory as
// two (unaligned) 16-byte loads per lane, no LDS and no transposes (the library's weight-gradient kernels ask for NHWC copies of both
// tensors): K runs over q = y * wp + x of the PADDED row pitch, 32 positions per MFMA, lane (channel i, group g) owning q0 = qb + 8 g .. + 7.
//   B = xp[ci][q + dy * wp + dx]: a tap is a shift of the load address (the three dx taps of a row share one ten-element window);
//   A = dY[co] at q: dY's rows are ow = wp - 2 long, so its linear index is q - 2 y; the two dropped columns of a row are zeros, and the
//       elements of a lane's run that lie in the next row are the loaded run shifted by two (wp even: a run never starts on the last column).
// A wave owns a 32 x 32 (co, ci) block for all nine taps (144 accumulator registers) and every (4 * ksplit)-th K block; the four waves of a
// workgroup are summed through LDS, the workgroup's partial goes to a workspace, conv2d_x3_wrw_reduce_kernel adds the partials in a fixed
// order (deterministic: no atomics).
struct Conv2dWrwParams {
    const float* dy;         // [n][cout][oh][ow]
    const float* xp;         // [n][cin][hp][wp], hp = oh + 2, wp = ow + 2
    float* ws;               // [ksplit][pairs][144][64] partial sums (register dumps)
    float* dw;               // [cout][cin][3][3]
    int n, cin, cout, hp, wp;
    int ksplit;
};

constexpr int kWrwAcc = 9 * 2 * 2 * 4;

__global__ void __launch_bounds__(256, 2) conv2d_x3_wrw_kernel(Conv2dWrwParams p) {
    NR_DYNAMIC_SMEM(float, red);                                   // [2][kWrwAcc][64]
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wave = NR_UNIFORM(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int oh = p.hp - 2, ow = p.wp - 2;
    const int cib = p.cin / 32, pair = (int)blockIdx.y, co0 = 32 * (pair / cib), ci0 = 32 * (pair % cib);
    const int kpi = (oh * p.wp + 31) / 32, nkb = p.n * kpi;        // K blocks per image, in all
    const nr_mbuf DY = nr_make_mbuf(p.dy, sizeof(float) * (size_t)p.n * p.cout * oh * ow);
    const nr_mbuf X = nr_make_mbuf(p.xp, sizeof(float) * (size_t)p.n * p.cin * p.hp * p.wp);
    v4f acc[9][2][2];
    NR_PRAGMA_UNROLL
    for (int t = 0; t < 9; ++t)
        NR_PRAGMA_UNROLL
        for (int a = 0; a < 2; ++a)
            NR_PRAGMA_UNROLL
            for (int b = 0; b < 2; ++b) { acc[t][a][b][0] = 0.0f; acc[t][a][b][1] = 0.0f; acc[t][a][b][2] = 0.0f; acc[t][a][b][3] = 0.0f; }
#pragma unroll 1
    for (int kblk = (int)blockIdx.x * 4 + wave; kblk < nkb; kblk += p.ksplit * 4) {
        const int img = kblk / kpi;
        const int q0 = (kblk - img * kpi) * 32 + 8 * g;
        const int y0 = q0 / p.wp, x0 = q0 - y0 * p.wp, jw = p.wp - x0;
        nr_v4u A[2][3];
        NR_PRAGMA_UNROLL
        for (int mt = 0; mt < 2; ++mt) {
            const int off = (((img * p.cout + co0 + 16 * mt + i) * oh) * ow + q0 - 2 * y0) * 4;
            const float4 lo = nr_buf_ld4(DY, off, 0), hi = nr_buf_ld4(DY, off + 16, 0);
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            float a[8];
            NR_PRAGMA_UNROLL
            for (int j = 0; j < 8; ++j) {
                const float same = (x0 + j < ow && y0 < oh) ? v[j] : 0.0f;
                const float next = (j >= 2 && y0 + 1 < oh) ? v[j >= 2 ? j - 2 : 0] : 0.0f;
                a[j] = j < jw ? same : next;
            }
            NR_PRAGMA_UNROLL
            for (int d = 0; d < 4; ++d) {
                unsigned h, m, l;
                nr_split3(a[2 * d], a[2 * d + 1], h, m, l);
                A[mt][0][d] = h; A[mt][1][d] = m; A[mt][2][d] = l;
            }
        }
        const int xoff = ((img * p.cin + ci0 + i) * p.hp) * p.wp + q0;
        NR_PRAGMA_UNROLL
        for (int dy = 0; dy < 3; ++dy) {
            // the ten positions q0 + dy wp .. + 9 of the lane's channel: the runs of the three dx taps are windows of them (three loads per
            // row and channel tile instead of six - the kernel is bound by the cache lines its loads touch, 16 channel planes per instruction)
            float win[2][10];
            NR_PRAGMA_UNROLL
            for (int nt = 0; nt < 2; ++nt) {
                const int off = (xoff + 16 * nt * p.hp * p.wp + dy * p.wp) * 4;
                const float4 lo = nr_buf_ld4(X, off, 0), hi = nr_buf_ld4(X, off + 16, 0);
                const float2 tl = nr_buf_ld2(X, off + 32, 0);
                win[nt][0] = lo.x; win[nt][1] = lo.y; win[nt][2] = lo.z; win[nt][3] = lo.w;
                win[nt][4] = hi.x; win[nt][5] = hi.y; win[nt][6] = hi.z; win[nt][7] = hi.w;
                win[nt][8] = tl.x; win[nt][9] = tl.y;
            }
            // the window's even pairs (0,1) .. (8,9) serve dx = 0 (pairs 0..3) and dx = 2 (pairs 1..4), its odd pairs (1,2) .. (7,8) dx = 1:
            // nine pair splits per window instead of twelve
            unsigned pe[2][3][5], po[2][3][4];
            NR_PRAGMA_UNROLL
            for (int nt = 0; nt < 2; ++nt) {
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 5; ++k) nr_split3(win[nt][2 * k], win[nt][2 * k + 1], pe[nt][0][k], pe[nt][1][k], pe[nt][2][k]);
                NR_PRAGMA_UNROLL
                for (int k = 0; k < 4; ++k) nr_split3(win[nt][2 * k + 1], win[nt][2 * k + 2], po[nt][0][k], po[nt][1][k], po[nt][2][k]);
            }
            NR_PRAGMA_UNROLL
            for (int dx = 0; dx < 3; ++dx) {
                const int tap = 3 * dy + dx;
                nr_v4u B[2][3];
                NR_PRAGMA_UNROLL
                for (int nt = 0; nt < 2; ++nt)
                    NR_PRAGMA_UNROLL
                    for (int pt = 0; pt < 3; ++pt)
                        NR_PRAGMA_UNROLL
                        for (int d = 0; d < 4; ++d) B[nt][pt][d] = dx == 1 ? po[nt][pt][d] : pe[nt][pt][d + (dx >> 1)];
                constexpr int WI[6] = {2, 0, 1, 1, 0, 0}, XJ[6] = {0, 2, 1, 0, 1, 0};
                NR_PRAGMA_UNROLL
                for (int pr = 0; pr < 6; ++pr)
                    NR_PRAGMA_UNROLL
                    for (int mt = 0; mt < 2; ++mt)
                        NR_PRAGMA_UNROLL
                        for (int nt = 0; nt < 2; ++nt) acc[tap][mt][nt] = nr_mfma16x32_bf16(A[mt][WI[pr]], B[nt][XJ[pr]], acc[tap][mt][nt]);
            }
        }
    }
    // waves 2, 3 -> 0, 1; wave 1 -> 0; wave 0 writes the workgroup's partial (register dump: [register][lane])
    auto dump = [&](float* dst) NR_LAMBDA_INLINE {
        NR_PRAGMA_UNROLL
        for (int t = 0; t < 9; ++t)
            NR_PRAGMA_UNROLL
            for (int a = 0; a < 2; ++a)
                NR_PRAGMA_UNROLL
                for (int b = 0; b < 2; ++b)
                    NR_PRAGMA_UNROLL
                    for (int r = 0; r < 4; ++r) dst[((((t * 2 + a) * 2 + b) * 4) + r) * 64 + lane] = acc[t][a][b][r];
    };
    auto gather = [&](const float* src) NR_LAMBDA_INLINE {
        NR_PRAGMA_UNROLL
        for (int t = 0; t < 9; ++t)
            NR_PRAGMA_UNROLL
            for (int a = 0; a < 2; ++a)
                NR_PRAGMA_UNROLL
                for (int b = 0; b < 2; ++b)
                    NR_PRAGMA_UNROLL
                    for (int r = 0; r < 4; ++r) acc[t][a][b][r] += src[((((t * 2 + a) * 2 + b) * 4) + r) * 64 + lane];
    };
    if (wave >= 2) dump(red + (wave - 2) * kWrwAcc * 64);
    NR_BLOCK_SYNC();
    if (wave < 2) gather(red + wave * kWrwAcc * 64);
    NR_BLOCK_SYNC();
    if (wave == 1) dump(red);
    NR_BLOCK_SYNC();
    if (wave == 0) {
        gather(red);
        dump(p.ws + ((size_t)blockIdx.x * gridDim.y + pair) * kWrwAcc * 64);
    }
}

// dw[co][ci][tap] = the partials of its (co, ci) block, added in a fixed order: a workgroup per (pair, register) row of 64 lanes, wave w adds
// the partials w, w + 8, ... and wave 0 the eight sums.
constexpr int kWrwRedWaves = 8;
__global__ void __launch_bounds__(64 * kWrwRedWaves) conv2d_x3_wrw_reduce_kernel(Conv2dWrwParams p) {
    __shared__ float part[kWrwRedWaves][64];
    const int cib = p.cin / 32, pairs = cib * (p.cout / 32);
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    const int reg = (int)blockIdx.x % kWrwAcc, pair = (int)blockIdx.x / kWrwAcc;
    float s = 0.0f;
    for (int k = wave; k < p.ksplit; k += kWrwRedWaves) s += p.ws[((size_t)k * pairs + pair) * kWrwAcc * 64 + reg * 64 + lane];
    part[wave][lane] = s;
    NR_BLOCK_SYNC();
    if (wave) return;
    for (int w = 1; w < kWrwRedWaves; ++w) s += part[w][lane];
    const int r = reg & 3, nt = (reg >> 2) & 1, mt = (reg >> 3) & 1, tap = reg >> 4;
    const int co = 32 * (pair / cib) + 16 * mt + 4 * (lane >> 4) + r, ci = 32 * (pair % cib) + 16 * nt + (lane & 15);
    p.dw[((size_t)co * p.cin + ci) * 9 + tap] = s;
}

}  // namespace nr
