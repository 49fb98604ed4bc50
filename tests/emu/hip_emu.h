// CPU emulation of the small HIP/CDNA subset used by neuray_amd/csrc/*.  TEST INFRASTRUCTURE ONLY.
//
// The product kernels are written against a handful of wrappers (nr_mfma16, nr_shfl*, NR_LAUNCH,
// __syncthreads, __shared__).  Building the same sources with -DNEURAY_EMU and this header gives a
// host library (tests/emu/_build/libneuray_emu.so) in which every HIP thread is a ucontext fiber:
// a wave is 64 fibers, cross-lane operations and barriers are rendezvous points, and the fp32 MFMA
// is the k-ordered fmaf chain the hardware implements (cdna_hip_programming.md section 3).  It exists so
// that kernel logic can be checked against the oracle in the CPU-only container; nothing under
// neuray_amd/ ever loads it and it is never timed.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#include <atomic>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct int2 { int x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ inline __attribute__((noinline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __shared__ static thread_local

namespace emu {

constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 256 * 1024;

struct WaveState {
    int active = 0;     // lanes not yet finished
    int arrived = 0;
    unsigned gen = 0;
    // double-buffered operand slots (parity = gen & 1)
    float fa[2][WAVE], fb[2][WAVE];
    uint64_t bits[2];
};

struct Lane {
    ucontext_t ctx;
    char* stack = nullptr;
    unsigned tid = 0;
    bool done = false;
    int wait_kind = 0;       // 0 runnable, 1 wave, 2 block
    unsigned wait_gen = 0;
};

struct Block {
    dim3 bidx, bdim, gdim;
    std::vector<Lane> lanes;
    std::vector<WaveState> waves;
    int blk_active = 0, blk_arrived = 0;
    unsigned blk_gen = 0;
    char* dyn_smem = nullptr;
    ucontext_t main_ctx;
    const std::function<void()>* body = nullptr;
};

extern thread_local Block* g_blk;
extern thread_local Lane* g_lane;

struct TidProxy { unsigned y = 0, z = 0; struct X { operator unsigned() const { return g_lane->tid; } } x; };
struct BidProxy { struct X { operator unsigned() const { return g_blk->bidx.x; } } x;
                  struct Y { operator unsigned() const { return g_blk->bidx.y; } } y;
                  struct Z { operator unsigned() const { return g_blk->bidx.z; } } z; };
struct BdimProxy { struct X { operator unsigned() const { return g_blk->bdim.x; } } x; unsigned y = 1, z = 1; };
struct GdimProxy { struct X { operator unsigned() const { return g_blk->gdim.x; } } x;
                   struct Y { operator unsigned() const { return g_blk->gdim.y; } } y;
                   struct Z { operator unsigned() const { return g_blk->gdim.z; } } z; };

void yield_lane();
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);

static inline WaveState& my_wave() { return g_blk->waves[g_lane->tid / WAVE]; }
static inline int my_lane() { return g_lane->tid % WAVE; }

// rendezvous of the (still active) lanes of this wave; returns the parity used for this round
static inline int wave_sync() {
    WaveState& w = my_wave();
    unsigned gen = w.gen;
    if (++w.arrived >= w.active) { w.arrived = 0; w.gen = gen + 1; return gen & 1; }
    g_lane->wait_kind = 1; g_lane->wait_gen = gen;
    yield_lane();
    return gen & 1;
}
static inline void block_sync() {
    Block* b = g_blk;
    unsigned gen = b->blk_gen;
    if (++b->blk_arrived >= b->blk_active) { b->blk_arrived = 0; b->blk_gen = gen + 1; return; }
    g_lane->wait_kind = 2; g_lane->wait_gen = gen;
    yield_lane();
}

}  // namespace emu

static emu::TidProxy threadIdx;
static emu::BidProxy blockIdx;
static emu::BdimProxy blockDim;
static emu::GdimProxy gridDim;

static inline void __syncthreads() { emu::block_sync(); }

// ---- cross-lane primitives -------------------------------------------------------------------
static inline float emu_shfl_f(float v, int src) {
    emu::WaveState& w = emu::my_wave();
    int par = w.gen & 1;
    w.fa[par][emu::my_lane()] = v;
    emu::wave_sync();
    return w.fa[par][src & 63];
}
static inline float __shfl(float v, int src) { return emu_shfl_f(v, src); }
static inline float __shfl_xor(float v, int m) { return emu_shfl_f(v, emu::my_lane() ^ m); }
static inline float __shfl_up(float v, int d) { int l = emu::my_lane(); return emu_shfl_f(v, l - d < 0 ? l : l - d); }
static inline float __shfl_down(float v, int d) { int l = emu::my_lane(); return emu_shfl_f(v, l + d > 63 ? l : l + d); }
static inline int __shfl(int v, int src) { float f; memcpy(&f, &v, 4); f = emu_shfl_f(f, src); memcpy(&v, &f, 4); return v; }
static inline int __shfl_xor(int v, int m) { return __shfl(v, emu::my_lane() ^ m); }
static inline int __shfl_up(int v, int d) { int l = emu::my_lane(); return __shfl(v, l - d < 0 ? l : l - d); }
static inline int __shfl_down(int v, int d) { int l = emu::my_lane(); return __shfl(v, l + d > 63 ? l : l + d); }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }

// float atomicAdd on LDS or global memory (blocks may run on different host threads: compare-and-swap loop)
static inline float atomicAdd(float* p, float v) {
    unsigned int* u = reinterpret_cast<unsigned int*>(p);
    unsigned int old = __atomic_load_n(u, __ATOMIC_RELAXED), neu;
    float f;
    do {
        memcpy(&f, &old, 4);
        const float g = f + v;
        memcpy(&neu, &g, 4);
    } while (!__atomic_compare_exchange_n(u, &old, neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return f;
}

static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) {
    return __atomic_fetch_add(p, v, __ATOMIC_RELAXED);
}

static inline unsigned long long __ballot(int pred) {
    emu::WaveState& w = emu::my_wave();
    int par = w.gen & 1;
    w.fa[par][emu::my_lane()] = pred ? 1.0f : 0.0f;
    emu::wave_sync();
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) if (w.fa[par][i] != 0.0f) m |= (1ull << i);
    return m;
}

// ---- fp32 MFMA 16x16x4: D = A(16x4) * B(4x16) + C, k-ordered fmaf chain -----------------------
typedef float v4f __attribute__((vector_size(16)));
static inline v4f nr_mfma16(float a, float b, v4f c) {
    emu::WaveState& w = emu::my_wave();
    int par = w.gen & 1;
    int l = emu::my_lane();
    w.fa[par][l] = a;   // A[m = l&15][k = l>>4]
    w.fb[par][l] = b;   // B[k = l>>4][n = l&15]
    emu::wave_sync();
    int n = l & 15, g = l >> 4;
    v4f d = c;
    for (int r = 0; r < 4; ++r) {
        int m = 4 * g + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.fa[par][k * 16 + m], w.fb[par][k * 16 + n], acc);
        d[r] = acc;
    }
    return d;
}

// ---- bf16-operand quad MFMA 16x16x16 (NR_BF16_QUADS builds): A values arrive as packed bf16 pairs, B values are rounded
// to bf16 (round to nearest even) here; fp32 accumulation over k = 4*(lane>>4) + j in ascending order.
static inline float emu_bf16_round(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return f;
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    float r; memcpy(&r, &u, 4); return r;
}
static inline void emu_bf16_unpack(float packed, float& lo, float& hi) {
    unsigned u; memcpy(&u, &packed, 4);
    unsigned l = u << 16, h = u & 0xffff0000u;
    memcpy(&lo, &l, 4); memcpy(&hi, &h, 4);
}
static inline v4f nr_mfma16_bf16q(float a01, float a23, float b0, float b1, float b2, float b3, v4f c) {
    float a[4];
    emu_bf16_unpack(a01, a[0], a[1]); emu_bf16_unpack(a23, a[2], a[3]);
    const float b[4] = {emu_bf16_round(b0), emu_bf16_round(b1), emu_bf16_round(b2), emu_bf16_round(b3)};
    v4f d = c;
    for (int j = 0; j < 4; ++j) {       // four K-steps with k = 4*kk + j: every (m, n) sums all 16 products; order differs
        v4f z = {0.0f, 0.0f, 0.0f, 0.0f};                                  // from the hardware's only in rounding noise
        v4f t = nr_mfma16(a[j], b[j], z);
        for (int r = 0; r < 4; ++r) d[r] += t[r];
    }
    return d;
}

// split library (NR_BF16_SPLIT): hi * hi + hi * lo + lo * hi with bf16 halves, fp32 accumulation (cross terms first)
static inline v4f nr_mfma16_bf16q3(float a01h, float a23h, float a01l, float a23l, float b0, float b1, float b2, float b3, v4f c) {
    float ah[4], al[4];
    emu_bf16_unpack(a01h, ah[0], ah[1]); emu_bf16_unpack(a23h, ah[2], ah[3]);
    emu_bf16_unpack(a01l, al[0], al[1]); emu_bf16_unpack(a23l, al[2], al[3]);
    const float b[4] = {b0, b1, b2, b3};
    float bh[4], bl[4];
    for (int j = 0; j < 4; ++j) { bh[j] = emu_bf16_round(b[j]); bl[j] = emu_bf16_round(b[j] - bh[j]); }
    v4f d = c;
    const float* As[3] = {al, ah, ah};
    const float* Bs[3] = {bh, bl, bh};
    for (int term = 0; term < 3; ++term)
        for (int j = 0; j < 4; ++j) {
            v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
            v4f t = nr_mfma16(As[term][j], Bs[term][j], z);
            for (int r = 0; r < 4; ++r) d[r] += t[r];
        }
    return d;
}

// ---- AR_X3 (nr_layout.h): the K = 32 / K = 16 bf16 MFMAs on packed operands and the three-way operand split ---------------
typedef unsigned int nr_v4u __attribute__((vector_size(16)));
typedef unsigned int nr_v2u __attribute__((vector_size(8)));
static inline float emu_bf16_word(unsigned w, int hi) { const unsigned u = hi ? (w & 0xffff0000u) : (w << 16); float f; memcpy(&f, &u, 4); return f; }
// every (m, n) sums its K products in fp32; the order differs from the hardware's only in rounding noise
static inline v4f nr_mfma16x32_bf16(nr_v4u a, nr_v4u b, v4f c) {
    v4f d = c;
    for (int i = 0; i < 8; ++i) {
        v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
        v4f t = nr_mfma16(emu_bf16_word(a[i / 2], i & 1), emu_bf16_word(b[i / 2], i & 1), z);
        for (int r = 0; r < 4; ++r) d[r] += t[r];
    }
    return d;
}
static inline v4f nr_mfma16x16_bf16(nr_v2u a, nr_v2u b, v4f c) {
    v4f d = c;
    for (int i = 0; i < 4; ++i) {
        v4f z = {0.0f, 0.0f, 0.0f, 0.0f};
        v4f t = nr_mfma16(emu_bf16_word(a[i / 2], i & 1), emu_bf16_word(b[i / 2], i & 1), z);
        for (int r = 0; r < 4; ++r) d[r] += t[r];
    }
    return d;
}
static inline unsigned emu_pk_bf16(float lo, float hi) {
    unsigned a, b; const float rl = emu_bf16_round(lo), rh = emu_bf16_round(hi);
    memcpy(&a, &rl, 4); memcpy(&b, &rh, 4);
    return (a >> 16) | (b & 0xffff0000u);
}
static inline void nr_split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = emu_pk_bf16(x0, x1);
    const float r0 = x0 - emu_bf16_word(h, 0), r1 = x1 - emu_bf16_word(h, 1);
    m = emu_pk_bf16(r0, r1);
    l = emu_pk_bf16(r0 - emu_bf16_word(m, 0), r1 - emu_bf16_word(m, 1));
}

// ---- exact-rounding helpers (same names as the HIP device intrinsics) --------------------------
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }

#define NR_DYNAMIC_SMEM(type, name) type* name = reinterpret_cast<type*>(emu::g_blk->dyn_smem)
#define NR_LAUNCH(kernel, grid, block, smem, stream, ...)                                   \
    do { std::function<void()> _body = [=]() { kernel(__VA_ARGS__); };                       \
         emu::launch((grid), (block), (smem), _body); } while (0)
