// Hardware probe (not a test): the ARITHMETIC of the point kernel's MLP chain.
//
// The fp32 point kernel is closed at 0.51 of the fp32 MFMA peak by its instruction mix; what is left is arithmetic the
// MI355X does faster than v_mfma_f32_16x16x4_f32 at fp32 GRADE.  This probe runs the same per-(point, view) work - a 32 -> 32 -> 32
// scaled-ELU chain out of LDS-resident weight fragments, a 2-row vector head on the VALU, optionally the point kernel's two narrow
// layers (32 -> 8, 32 -> 16), one deterministic LDS all-reduce over the four view waves - in the point kernel's shape (a wave = 2 view
// slots x 16 points, 4 waves per workgroup, 3 workgroups per CU) on six arithmetics:
//
//   F32   v_mfma_f32_16x16x4_f32, 8 MFMAs per K = 32                                   (today's product kernel)
//   X3    v_mfma_f32_16x16x32_bf16, every operand split THREE ways x = h + m + l (bf16 each, round-to-nearest residuals: the split is
//         exact for an fp32 operand); products hh, hm, mh, hl, lh, mm = 6 MFMAs per K = 32; dropped m l + l m + l l <= 2^-23 |x w|
//   X2    v_mfma_f32_16x16x32_bf16, two-way split (hh, hl, lh: 3 MFMAs per K = 32; 2^-16 grade)      (today's bf16x3 on the new opcode)
//   X2L   v_mfma_f32_16x16x16_bf16 (_1k), two-way split: 3 MFMAs per K = 16 = 6 per K = 32           (today's bf16x3 library)
//   H2    v_mfma_f32_16x16x32_f16, two-way f16 split (3 MFMAs per K = 32; 2^-22 grade)                (the cheap comparison)
//   B1    v_mfma_f32_16x16x32_bf16, plain bf16 operands (1 MFMA per K = 32)                           (today's bf16 on the new opcode)
//
// The D registers of a layer are the B operands of the next one in every arithmetic: a lane's 8 K-values of a K = 32 step are its 4
// D registers of output tile 2 kp and its 4 of tile 2 kp + 1 (the weights' K order is permuted at pack time to match).
// Weights are split at pack time, activations in the kernel (v_cvt_pk_bf16_f32 + unpack + subtract: 11 VALU per register PAIR for
// three parts) - that VALU cost, not the MFMAs, decides the outcome, so it is inside the timed region.
//
// Reported per arithmetic: SIMD cycles per (point, view) at full occupancy (with / without narrow layers and all-reduce), VGPRs (from
// -Rpass-analysis=kernel-resource-usage), the chain's error against a float64 evaluation of the same network on the same fp32
// inputs (max and rms over 64 tiles), and - for the split forms - the worst relative error of SINGLE products a * b against the
// exact product, over random and adversarial operands (values whose split residuals are as large as they can be).
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tests/hw/split_arith_probe.hip -o /tmp/split_arith_probe && /tmp/split_arith_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef short v4s __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

enum Arith { F32, X3, X2, X2L, H2, B1 };
constexpr int parts_of(int ar) { return ar == X3 ? 3 : (ar == X2 || ar == X2L || ar == H2) ? 2 : 1; }

constexpr float kL = 1.4426950408889634f;
__device__ __forceinline__ float elu_s(float t) { return __builtin_amdgcn_fmed3f(t, fmaf(__builtin_amdgcn_exp2f(t), kL, -kL), 0.0f); }

__device__ __forceinline__ float group_sum(float t) {         // the four 16-lane groups: (g0 + g1) + (g2 + g3)
    const unsigned u = __builtin_bit_cast(unsigned, t);
    const v2f a = __builtin_bit_cast(v2f, (v2u)__builtin_amdgcn_permlane16_swap(u, u, false, false));
    const float s = a.x + a.y;
    const unsigned v = __builtin_bit_cast(unsigned, s);
    const v2f b = __builtin_bit_cast(v2f, (v2u)__builtin_amdgcn_permlane32_swap(v, v, false, false));
    return b.x + b.y;
}

// ---- operand splits ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    v2bf v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ unsigned pk_f16(float lo, float hi) {
    v2h v; v[0] = (_Float16)lo; v[1] = (_Float16)hi;
    return __builtin_bit_cast(unsigned, v);
}
// P parts of a register pair (x0, x1): part[i] = bf16 pair; x = part0 + part1 (+ part2), residuals exact in fp32
template <int P> __device__ __forceinline__ void split_bf16(float x0, float x1, unsigned (&part)[3]) {
    part[0] = pk_bf16(x0, x1);
    if (P >= 2) {
        const float r0 = x0 - bf_lo(part[0]), r1 = x1 - bf_hi(part[0]);
        part[1] = pk_bf16(r0, r1);
        if (P >= 3) part[2] = pk_bf16(r0 - bf_lo(part[1]), r1 - bf_hi(part[1]));
    }
}
__device__ __forceinline__ void split_f16(float x0, float x1, unsigned (&part)[3]) {
    part[0] = pk_f16(x0, x1);
    const v2h h = __builtin_bit_cast(v2h, part[0]);
    part[1] = pk_f16(x0 - (float)h[0], x1 - (float)h[1]);
}

// B operand of one K = 32 step of one slot: parts x 4 registers (8 values: D registers 0..3 of tile 2 kp, then of tile 2 kp + 1)
template <int AR> struct BOp { v4u p[parts_of(AR)]; };
template <int AR> __device__ __forceinline__ BOp<AR> make_b(const float* x /*8 values*/) {
    BOp<AR> b;
    unsigned q[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (AR == H2) split_f16(x[2 * i], x[2 * i + 1], q[i]);
        else split_bf16<parts_of(AR)>(x[2 * i], x[2 * i + 1], q[i]);
    }
#pragma unroll
    for (int p = 0; p < parts_of(AR); ++p) { b.p[p][0] = q[0][p]; b.p[p][1] = q[1][p]; b.p[p][2] = q[2][p]; b.p[p][3] = q[3][p]; }
    return b;
}
__device__ __forceinline__ v4f mma_bf(v4u a, v4u b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0); }
__device__ __forceinline__ v4f mma_h(v4u a, v4u b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0); }
__device__ __forceinline__ v4f mma_bfl(unsigned a0, unsigned a1, unsigned b0, unsigned b1, v4f c) {
    const v2u a = {a0, a1}, b = {b0, b1};
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(v4s, a), __builtin_bit_cast(v4s, b), c, 0, 0, 0);
}

// ---- the network (host, natural layout) -------------------------------------------------------------------------------------------
struct Net { float W1[32][32], W2[32][32], b1[32], b2[32], Wh[2][32], N1[8][32], N2[16][32]; };

// packed buffer (32-bit words).  F32: quads as in nr_layout.h ([mo][kq][lane] float4, component j = W[16 mo + m][16 kq + 4 g + j]).
// Split forms: [mo][part][lane] uint4 for the single K = 32 step of a 32-input layer: 8 x 16-bit = W[16 mo + m][16 q + 4 g + j], q = 0, 1.
constexpr int LW = 2 * 3 * 64 * 4;            // words reserved per 32-row layer (3 parts)
constexpr int O_W1 = 0, O_W2 = LW, O_N1 = 2 * LW, O_N2 = 3 * LW, O_B1 = 4 * LW, O_B2 = O_B1 + 32, O_WH = O_B2 + 32, O_END = O_WH + 64;

static unsigned short f2bf(float f) {         // round to nearest even
    unsigned u; memcpy(&u, &f, 4);
    const unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(r >> 16);
}
static float bf2f(unsigned short b) { const unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2h(float f) { const _Float16 h = (_Float16)f; unsigned short s; memcpy(&s, &h, 2); return s; }
static float h2f(unsigned short s) { _Float16 h; memcpy(&h, &s, 2); return (float)h; }
static void split_host(int ar, float w, unsigned short (&p)[3]) {
    p[0] = p[1] = p[2] = 0;
    if (ar == H2) { p[0] = f2h(w); p[1] = f2h(w - h2f(p[0])); return; }
    p[0] = f2bf(w);
    if (parts_of(ar) >= 2) { const float r = w - bf2f(p[0]); p[1] = f2bf(r); if (parts_of(ar) >= 3) p[2] = f2bf(r - bf2f(p[1])); }
}

static void pack(int ar, const Net& n, std::vector<unsigned>& p) {
    p.assign(O_END, 0u);
    auto layer = [&](int off, const float (*W)[32], int rows) {
        for (int mo = 0; mo < (rows + 15) / 16; ++mo)
            for (int l = 0; l < 64; ++l) {
                const int m = l & 15, g = l >> 4, row = 16 * mo + m;
                if (ar == F32) {
                    for (int kq = 0; kq < 2; ++kq)
                        for (int j = 0; j < 4; ++j) {
                            const float w = row < rows ? W[row][16 * kq + 4 * g + j] : 0.0f;
                            memcpy(&p[off + ((mo * 2 + kq) * 64 + l) * 4 + j], &w, 4);
                        }
                } else {
                    for (int i = 0; i < 8; ++i) {
                        const float w = row < rows ? W[row][16 * (i / 4) + 4 * g + i % 4] : 0.0f;
                        unsigned short s[3];
                        split_host(ar, w, s);
                        for (int part = 0; part < parts_of(ar); ++part) {
                            unsigned& word = p[off + ((mo * 3 + part) * 64 + l) * 4 + i / 2];
                            word |= (unsigned)s[part] << (16 * (i & 1));
                        }
                    }
                }
            }
    };
    layer(O_W1, n.W1, 32); layer(O_W2, n.W2, 32); layer(O_N1, n.N1, 8); layer(O_N2, n.N2, 16);
    for (int i = 0; i < 32; ++i) { memcpy(&p[O_B1 + i], &n.b1[i], 4); memcpy(&p[O_B2 + i], &n.b2[i], 4); }
    for (int j = 0; j < 2; ++j) for (int f = 0; f < 32; ++f) memcpy(&p[O_WH + j * 32 + f], &n.Wh[j][f], 4);
}

__host__ __device__ inline float feat(int tile, int view, int point, int f) {
    const unsigned x = (unsigned)(tile * 7919 + view * 104729 + point * 1299709 + f * 15485863);
    return (float)((x * 2654435761u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

template <int R, int NW>
__device__ __forceinline__ void block_allreduce(float (&v)[R], float* red, int wave, int lane) {
#pragma unroll
    for (int r = 0; r < R; ++r) red[(wave * R + r) * 64 + lane] = v[r];
    __syncthreads();
    for (int r = wave; r < R; r += NW) {
        float s = red[r * 64 + lane];
        for (int w = 1; w < NW; ++w) s += red[(w * R + r) * 64 + lane];
        red[(NW * R + r) * 64 + lane] = s;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = red[(NW * R + r) * 64 + lane];
}

// one 32-input layer with MT output tiles for NS slots; acc holds the bias on entry
template <int AR, int MT, int NS>
__device__ __forceinline__ void layer(const unsigned* w, int lane, const float (&x)[NS][8], v4f (&acc)[NS][MT]) {
    if constexpr (AR == F32) {
#pragma unroll
        for (int mo = 0; mo < MT; ++mo)
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {
                const float4 a = *reinterpret_cast<const float4*>(w + ((mo * 2 + kq) * 64 + lane) * 4);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x[s][4 * kq + 0], acc[s][mo], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x[s][4 * kq + 1], acc[s][mo], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x[s][4 * kq + 2], acc[s][mo], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x[s][4 * kq + 3], acc[s][mo], 0, 0, 0);
            }
    } else {
        constexpr int P = parts_of(AR);
        BOp<AR> b[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) b[s] = make_b<AR>(x[s]);
        v4u a[MT][P];
#pragma unroll
        for (int mo = 0; mo < MT; ++mo)
#pragma unroll
            for (int p = 0; p < P; ++p) a[mo][p] = *reinterpret_cast<const v4u*>(w + ((mo * 3 + p) * 64 + lane) * 4);
        // products (weight part i, activation part j) with i + j < P, smallest first; the accumulators of the MT x NS (tile, slot)
        // pairs are independent chains, so consecutive MFMAs never wait for each other
#pragma unroll
        for (int o = P - 1; o >= 0; --o)
#pragma unroll
            for (int i = 0; i <= o; ++i)
#pragma unroll
                for (int mo = 0; mo < MT; ++mo)
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int j = o - i;
                        if constexpr (AR == H2) acc[s][mo] = mma_h(a[mo][i], b[s].p[j], acc[s][mo]);
                        else if constexpr (AR == X2L) {
                            acc[s][mo] = mma_bfl(a[mo][i][0], a[mo][i][1], b[s].p[j][0], b[s].p[j][1], acc[s][mo]);
                            acc[s][mo] = mma_bfl(a[mo][i][2], a[mo][i][3], b[s].p[j][2], b[s].p[j][3], acc[s][mo]);
                        } else acc[s][mo] = mma_bf(a[mo][i], b[s].p[j], acc[s][mo]);
                    }
    }
}

template <int AR, bool NARROW, bool ALLRED>
__global__ void __launch_bounds__(256, 3) chain(const unsigned* packed, float* out, int tiles_per_wg, int write_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned smem[];
    unsigned* wl = smem;
    float* red = reinterpret_cast<float*>(smem + O_END);
    const float* wf = reinterpret_cast<const float*>(wl);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    for (int i = threadIdx.x; i < O_END; i += 256) wl[i] = packed[i];
    __syncthreads();
    float keep = 0.0f;
    for (int it = 0; it < tiles_per_wg; ++it) {
        const int tile = blockIdx.x * tiles_per_wg + it;
        float x[2][8];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[s][i] = feat(tile, wave * 2 + s, c, 16 * (i / 4) + 4 * g + i % 4);
        float h1[2][8], h2[2][8];
        {
            v4f acc[2][2];
            const float4 b0 = *reinterpret_cast<const float4*>(wf + O_B1 + 4 * g), b1 = *reinterpret_cast<const float4*>(wf + O_B1 + 16 + 4 * g);
#pragma unroll
            for (int s = 0; s < 2; ++s) { acc[s][0] = (v4f){b0.x, b0.y, b0.z, b0.w}; acc[s][1] = (v4f){b1.x, b1.y, b1.z, b1.w}; }
            layer<AR, 2, 2>(wl + O_W1, lane, x, acc);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) h1[s][i] = elu_s(acc[s][i / 4][i % 4]);
        }
        {
            v4f acc[2][2];
            const float4 b0 = *reinterpret_cast<const float4*>(wf + O_B2 + 4 * g), b1 = *reinterpret_cast<const float4*>(wf + O_B2 + 16 + 4 * g);
#pragma unroll
            for (int s = 0; s < 2; ++s) { acc[s][0] = (v4f){b0.x, b0.y, b0.z, b0.w}; acc[s][1] = (v4f){b1.x, b1.y, b1.z, b1.w}; }
            layer<AR, 2, 2>(wl + O_W2, lane, h1, acc);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) h2[s][i] = elu_s(acc[s][i / 4][i % 4]);
        }
        float head[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 w0 = *reinterpret_cast<const float4*>(wf + O_WH + j * 32 + 4 * g), w1 = *reinterpret_cast<const float4*>(wf + O_WH + j * 32 + 16 + 4 * g);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float a = h2[s][0] * w0.x;
                a = fmaf(h2[s][1], w0.y, a); a = fmaf(h2[s][2], w0.z, a); a = fmaf(h2[s][3], w0.w, a);
                a = fmaf(h2[s][4], w1.x, a); a = fmaf(h2[s][5], w1.y, a); a = fmaf(h2[s][6], w1.z, a); a = fmaf(h2[s][7], w1.w, a);
                head[s][j] = group_sum(a);
            }
        }
        if (tile < write_tiles && g == 0) {          // the chain's own outputs, for the error report: [tile][view][point][32 + 2]
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float* o = out + ((size_t)(tile * 8 + wave * 2 + s) * 16 + c) * 34;
                o[32] = head[s][0]; o[33] = head[s][1];
            }
        }
        if (tile < write_tiles) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) out[((size_t)(tile * 8 + wave * 2 + s) * 16 + c) * 34 + 16 * (i / 4) + 4 * g + i % 4] = h2[s][i];
        }
        float n1[2][4], n2[2][4];
        if constexpr (NARROW) {
            v4f acc[2][1];
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[s][0] = (v4f){0.f, 0.f, 0.f, 0.f};
            layer<AR, 1, 2>(wl + O_N1, lane, h2, acc);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) n1[s][i] = elu_s(acc[s][0][i]);
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[s][0] = (v4f){0.f, 0.f, 0.f, 0.f};
            layer<AR, 1, 2>(wl + O_N2, lane, h1, acc);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) n2[s][i] = elu_s(acc[s][0][i]);
        }
        float st[12];
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = h2[0][i] * head[0][0] + h2[1][i] * head[1][0];
        st[8] = head[0][1] + head[1][1]; st[9] = head[0][0] + head[1][0]; st[10] = head[0][1] * head[0][0] + head[1][1] * head[1][0];
        st[11] = 2.0f;
        if constexpr (NARROW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { st[i] += n1[0][i] + n1[1][i]; st[4 + i] += n2[0][i] + n2[1][i]; }
        }
        if constexpr (ALLRED) block_allreduce<12, 4>(st, red, wave, lane);
#pragma unroll
        for (int i = 0; i < 12; ++i) keep += st[i];
    }
    if (keep == 123.456f) out[threadIdx.x] = keep;
}

// ---- single products: D[m][n] = A[m][n % 32] * b[n] (B = diag pattern), every operand split in the kernel's way ------------------
template <int AR>
__global__ void products(const float* a /*[16][32]*/, const float* bdiag /*[16]*/, float* d /*[16][16]*/) {
    const int lane = threadIdx.x, m = lane & 15, g = lane >> 4;
    float av[8], bv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = 8 * g + i;
        av[i] = a[m * 32 + k];
        bv[i] = k == m ? bdiag[m] : 0.0f;      // this lane's column is n = lane & 15: only k = n carries b[n]
    }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (AR == F32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {      // the same sum on the fp32 MFMA: K-step i, lane group g supplies k = 8 g + i
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc, 0, 0, 0);
        }
    } else {
        constexpr int P = parts_of(AR);
        const BOp<AR> A = make_b<AR>(av), B = make_b<AR>(bv);
#pragma unroll
        for (int o = P - 1; o >= 0; --o)
#pragma unroll
            for (int i = 0; i <= o; ++i) {
                if constexpr (AR == H2) acc = mma_h(A.p[i], B.p[o - i], acc);
                else acc = mma_bf(A.p[i], B.p[o - i], acc);
            }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) d[(4 * g + r) * 16 + m] = acc[r];
}

// ---- host float64 evaluation ------------------------------------------------------------------------------------------------------
static double elu_s_d(double t) { return t > 0.0 ? t : 1.4426950408889634074 * (exp2(t) - 1.0); }
static void host_chain(const Net& n, int tile, int view, int point, double (&h2)[32], double (&head)[2]) {
    double x[32], h1[32];
    for (int f = 0; f < 32; ++f) x[f] = feat(tile, view, point, f);
    for (int o = 0; o < 32; ++o) { double a = n.b1[o]; for (int f = 0; f < 32; ++f) a += (double)n.W1[o][f] * x[f]; h1[o] = elu_s_d(a); }
    for (int o = 0; o < 32; ++o) { double a = n.b2[o]; for (int f = 0; f < 32; ++f) a += (double)n.W2[o][f] * h1[f]; h2[o] = elu_s_d(a); }
    for (int j = 0; j < 2; ++j) { double a = 0; for (int f = 0; f < 32; ++f) a += (double)n.Wh[j][f] * h2[f]; head[j] = a; }
}

template <class K>
static double time_kernel(K launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch(); launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}


// ---- issue / dependent latency of v_mfma_f32_16x16x32_bf16: NACC independent accumulator chains, one wave per SIMD ------------------
template <int NACC, int GAPV>
__global__ void __launch_bounds__(256) mfma_chain(float* out, int n) {
    v4u a = {threadIdx.x * 3u + 1u, threadIdx.x * 5u + 2u, threadIdx.x * 7u + 3u, threadIdx.x * 11u + 4u}, b = a;
    b[0] ^= 0x01010101u;
    v4f acc[NACC];
    float filler = (float)threadIdx.x;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = mma_bf(a, b, acc[i]);
#pragma unroll
                for (int v = 0; v < GAPV; ++v) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(filler));
            }
    }
    float s = filler;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int NACC, int GAPV>
static void run_chain(float* dout, int cus, double ghz) {
    const int n = 2000;
    const double ms = time_kernel([&] { hipLaunchKernelGGL((mfma_chain<NACC, GAPV>), cus, 256, 0, 0, dout, n); }, 3);
    printf("    16x16x32 bf16, %d accumulator chain(s), %d v_fma between MFMAs: %6.1f SIMD cycles per MFMA\n", NACC, GAPV, ms * 1e-3 * ghz * 1e9 / ((double)n * 8 * NACC));
}

static const char* kName[] = {"F32  16x16x4 f32, 8 MFMA / K32", "X3   16x16x32 bf16 3-way, 6 MFMA / K32", "X2   16x16x32 bf16 2-way, 3 MFMA / K32",
                              "X2L  16x16x16 bf16 2-way, 6 MFMA / K32", "H2   16x16x32 f16 2-way, 3 MFMA / K32", "B1   16x16x32 bf16 plain, 1 MFMA / K32"};

template <int AR>
static void run(const Net& net, float* dout, int cus, double ghz) {
    std::vector<unsigned> p;
    pack(AR, net, p);
    unsigned* dp;
    CHECK(hipMalloc(&dp, p.size() * 4));
    CHECK(hipMemcpy(dp, p.data(), p.size() * 4, hipMemcpyHostToDevice));
    const size_t smem = (O_END + 5 * 12 * 64) * 4;
    // ---- error of the chain against float64, 64 tiles ----
    const int WT = 64;
    std::vector<float> out((size_t)WT * 8 * 16 * 34);
    hipLaunchKernelGGL((chain<AR, false, true>), WT, 256, smem, 0, dp, dout, 1, WT);
    CHECK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    double mx = 0, sq = 0, mxh = 0, sqh = 0; size_t cnt = 0, cnth = 0;
    for (int t = 0; t < WT; ++t)
        for (int v = 0; v < 8; ++v)
            for (int pt = 0; pt < 16; ++pt) {
                double h2[32], head[2];
                host_chain(net, t, v, pt, h2, head);
                const float* o = &out[((size_t)(t * 8 + v) * 16 + pt) * 34];
                for (int f = 0; f < 32; ++f) { const double e = fabs(o[f] - h2[f]); mx = fmax(mx, e); sq += e * e; ++cnt; }
                for (int j = 0; j < 2; ++j) { const double e = fabs(o[32 + j] - head[j]); mxh = fmax(mxh, e); sqh += e * e; ++cnth; }
            }
    printf("%-42s chain vs float64: h2 max %.3e rms %.3e | head max %.3e rms %.3e\n", kName[AR], mx, sqrt(sq / cnt), mxh, sqrt(sqh / cnth));
    // ---- timing at full occupancy ----
    const int tiles = 400, reps = 5, grid = cus * 3 * 4;
    const double pv = (double)grid * tiles * 16 * 8;
    auto report = [&](const char* name, double ms) {
        printf("    %-40s %8.3f ms  %7.2f SIMD-cycles/pv\n", name, ms, ms * 1e-3 * ghz * 1e9 * cus * 4 / pv);
    };
    report("chain+head+allreduce", time_kernel([&] { hipLaunchKernelGGL((chain<AR, false, true>), grid, 256, smem, 0, dp, dout, tiles, 0); }, reps));
    report("... + narrow layers (32->8, 32->16)", time_kernel([&] { hipLaunchKernelGGL((chain<AR, true, true>), grid, 256, smem, 0, dp, dout, tiles, 0); }, reps));
    report("chain+head, no allreduce", time_kernel([&] { hipLaunchKernelGGL((chain<AR, false, false>), grid, 256, smem, 0, dp, dout, tiles, 0); }, reps));
    report("chain+head+narrow, no allreduce", time_kernel([&] { hipLaunchKernelGGL((chain<AR, true, false>), grid, 256, smem, 0, dp, dout, tiles, 0); }, reps));
    CHECK(hipFree(dp));
}

// worst relative error of single products, in units of 2^-24
template <int AR>
static void run_products(const char* name) {
    float *da, *db, *dd;
    CHECK(hipMalloc(&da, 512 * 4)); CHECK(hipMalloc(&db, 16 * 4)); CHECK(hipMalloc(&dd, 256 * 4));
    std::vector<float> a(512), b(16), d(256);
    double worst_rand = 0, worst_adv = 0;
    auto rndm = [] { return (float)((double)rand() / RAND_MAX * 2.0 - 1.0) * exp2f((float)(rand() % 41 - 20)); };
    // adversarial: significands whose bf16 residuals are maximal at both levels: 1 + 2^-8 - 2^-16 + 2^-9 ... patterns around the
    // rounding ties of each 8-bit field
    auto adv = [] {
        const unsigned f1 = (rand() & 0x7f), t2 = (rand() & 1) ? 0x7f : 0x80, t3 = (rand() & 1) ? 0x7f : 0x80;
        unsigned u = (127u + (rand() % 9) - 4u) << 23 | (f1 << 16) | (t2 << 8) | t3;
        u ^= (rand() & 3);
        if (rand() & 1) u |= 0x80000000u;
        float f; memcpy(&f, &u, 4); return f;
    };
    for (int rep = 0; rep < 4000; ++rep) {
        const bool is_adv = rep & 1;
        for (auto& v : a) v = is_adv ? adv() : rndm();
        for (auto& v : b) v = is_adv ? adv() : rndm();
        CHECK(hipMemcpy(da, a.data(), 512 * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(db, b.data(), 16 * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL((products<AR>), 1, 64, 0, 0, da, db, dd);
        CHECK(hipMemcpy(d.data(), dd, 256 * 4, hipMemcpyDeviceToHost));
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                const double exact = (double)a[m * 32 + n] * (double)b[n];
                const double rel = fabs((double)d[m * 16 + n] - exact) / fabs(exact) * 16777216.0;
                if (is_adv) worst_adv = fmax(worst_adv, rel); else worst_rand = fmax(worst_rand, rel);
            }
    }
    printf("%-42s single products a*b: worst relative error %.3f x 2^-24 (random), %.3f x 2^-24 (adversarial)\n", name, worst_rand, worst_adv);
    CHECK(hipFree(da)); CHECK(hipFree(db)); CHECK(hipFree(dd));
}

int main() {
    Net* net = new Net;
    srand(1);
    auto rnd = [] { return ((float)rand() / RAND_MAX - 0.5f) * 0.5f; };
    for (auto& r : net->W1) for (auto& v : r) v = rnd();
    for (auto& r : net->W2) for (auto& v : r) v = rnd();
    for (auto& v : net->b1) v = rnd();
    for (auto& v : net->b2) v = rnd();
    for (auto& r : net->Wh) for (auto& v : r) v = rnd();
    for (auto& r : net->N1) for (auto& v : r) v = rnd();
    for (auto& r : net->N2) for (auto& v : r) v = rnd();
    float* dout;
    CHECK(hipMalloc(&dout, (size_t)64 * 8 * 16 * 34 * 4 + 4096));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("%d CUs, %.2f GHz; a wave = 2 view slots x 16 points, 4 waves per workgroup, 3 workgroups per CU\n", cus, ghz);
    run<F32>(*net, dout, cus, ghz);
    run<X3>(*net, dout, cus, ghz);
    run<X2>(*net, dout, cus, ghz);
    run<X2L>(*net, dout, cus, ghz);
    run<H2>(*net, dout, cus, ghz);
    run<B1>(*net, dout, cus, ghz);
    printf("MFMA issue / dependency (one wave per SIMD):\n");
    run_chain<1, 0>(dout, cus, ghz); run_chain<2, 0>(dout, cus, ghz); run_chain<3, 0>(dout, cus, ghz); run_chain<4, 0>(dout, cus, ghz);
    run_chain<1, 2>(dout, cus, ghz); run_chain<2, 2>(dout, cus, ghz); run_chain<2, 4>(dout, cus, ghz); run_chain<4, 4>(dout, cus, ghz); run_chain<4, 8>(dout, cus, ghz);
    run_products<F32>(kName[F32]);
    run_products<X3>(kName[X3]);
    run_products<X2>(kName[X2]);
    run_products<H2>(kName[H2]);
    return 0;
}
