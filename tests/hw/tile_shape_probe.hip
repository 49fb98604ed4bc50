// Hardware probe (not a test): the TILE SHAPE of the point kernel.
//
//   shape A (today):   a wave = 2 view slots x 16 points on v_mfma_f32_16x16x4_f32, 4 waves per workgroup (8 views)
//   shape B (probed):  a wave = 1 view slot  x 32 points on v_mfma_f32_32x32x2_f32, 8 waves per workgroup (8 views)
//
// Both run the same per-(point, view) arithmetic out of LDS-resident weight fragments: a 32 -> 32 -> 32 scaled-ELU chain (the shape
// of a dist-decoder head / vis_fc), a 2-row vector head on the VALU (lane-group sum), optionally the point kernel's two NARROW layers
// (32 -> 8: neuray_fc.0; 32 -> 16: rgb_fc.0 - output rows that fill a 16-row MFMA tile but only half / a quarter of a 32-row one), and
// one deterministic LDS all-reduce over the view waves of the 35 + 1 per-point statistics rows (reduce-scatter + all-gather, two
// barriers: nr_device.h block_allreduce).  The D registers of a layer are the B operands of the next one in both shapes (K packed in D
// order); the host checks that against a plain CPU evaluation of the same network, so the 32x32x2 operand maps written below are the
// ones a kernel would need.
//
// Reported: ns and SIMD cycles per (point, view) at full occupancy, per shape, with and without the narrow layers; VGPRs from
// `hipcc -Rpass-analysis=kernel-resource-usage`.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tests/hw/tile_shape_probe.hip -o tests/hw/tile_shape_probe && tests/hw/tile_shape_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr float kL = 1.4426950408889634f;
__device__ __forceinline__ float elu_s(float t) { return __builtin_amdgcn_fmed3f(t, fmaf(__builtin_amdgcn_exp2f(t), kL, -kL), 0.0f); }
static float elu_s_host(float t) { return t > 0.0f ? t : kL * (exp2f(t) - 1.0f); }

__device__ __forceinline__ float half_sum(float t) {          // lanes l, l ^ 32: every lane receives lo + hi
    const unsigned v = __builtin_bit_cast(unsigned, t);
    const v2f b = __builtin_bit_cast(v2f, (v2u)__builtin_amdgcn_permlane32_swap(v, v, false, false));
    return b.x + b.y;
}
__device__ __forceinline__ float group_sum(float t) {         // the four 16-lane groups: (g0 + g1) + (g2 + g3)
    const unsigned u = __builtin_bit_cast(unsigned, t);
    const v2f a = __builtin_bit_cast(v2f, (v2u)__builtin_amdgcn_permlane16_swap(u, u, false, false));
    return half_sum(a.x + a.y);
}

// ---- natural network (host side): W1, W2 [32][32], b1, b2 [32], Wh [2][32], N1 [8][32], N2 [16][32] --------------------------------
struct Net { float W1[32][32], W2[32][32], b1[32], b2[32], Wh[2][32], N1[8][32], N2[16][32]; };

// ---- packed fragments -------------------------------------------------------------------------------------------------------------
// shape A: quad (mo, kq), component j, lane (m = l & 15, g = l >> 4):  W[16 mo + m][16 kq + 4 g + j]         (nr_layout.h)
// shape B: quad kq, component j, lane (m = l & 31, h = l >> 5):        W[m][8 kq + 4 h + j]
//   (32x32x2: lane l supplies A[m = l & 31][k = l >> 5] and B[k = l >> 5][n = l & 31]; register i of lane (n, h) receives
//    D[8 (i / 4) + 4 h + i % 4][n] - so K-step s = 4 kq + j takes D register s of the previous layer as its B operand)
// float offsets inside the packed buffer
constexpr int A_W1 = 0, A_W2 = 1024, A_B1 = 2048, A_B2 = 2048 + 32, A_WH = 2112, A_N1 = 2112 + 64, A_N2 = A_N1 + 512, A_END = A_N2 + 512;
constexpr int B_W1 = 0, B_W2 = 1024, B_B1 = 2048, B_B2 = 2048 + 32, B_WH = 2112, B_N1 = 2112 + 64, B_N2 = B_N1 + 1024, B_END = B_N2 + 1024;

static void pack_a(const Net& n, std::vector<float>& p) {
    p.assign(A_END, 0.0f);
    auto quads = [&](int off, const float (*W)[32], int rows) {
        for (int mo = 0; mo < (rows + 15) / 16; ++mo)
            for (int kq = 0; kq < 2; ++kq)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 4; ++j) {
                        const int m = l & 15, g = l >> 4, row = 16 * mo + m;
                        p[off + ((mo * 2 + kq) * 64 + l) * 4 + j] = row < rows ? W[row][16 * kq + 4 * g + j] : 0.0f;
                    }
    };
    quads(A_W1, n.W1, 32); quads(A_W2, n.W2, 32); quads(A_N1, n.N1, 8); quads(A_N2, n.N2, 16);
    for (int i = 0; i < 32; ++i) { p[A_B1 + i] = n.b1[i]; p[A_B2 + i] = n.b2[i]; }       // [mo][g][r] = natural order 16 mo + 4 g + r
    for (int j = 0; j < 2; ++j)
        for (int f = 0; f < 32; ++f) p[A_WH + j * 32 + f] = n.Wh[j][f];                    // [j][t][g][r] = natural
}
static void pack_b(const Net& n, std::vector<float>& p) {
    p.assign(B_END, 0.0f);
    auto quads = [&](int off, const float (*W)[32], int rows) {
        for (int kq = 0; kq < 4; ++kq)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 4; ++j) {
                    const int m = l & 31, h = l >> 5;
                    p[off + (kq * 64 + l) * 4 + j] = m < rows ? W[m][8 * kq + 4 * h + j] : 0.0f;
                }
    };
    quads(B_W1, n.W1, 32); quads(B_W2, n.W2, 32); quads(B_N1, n.N1, 8); quads(B_N2, n.N2, 16);
    for (int i = 0; i < 32; ++i) { p[B_B1 + i] = n.b1[i]; p[B_B2 + i] = n.b2[i]; }       // register i of half h: row 8 (i / 4) + 4 h + i % 4
    for (int j = 0; j < 2; ++j)
        for (int f = 0; f < 32; ++f) p[B_WH + j * 32 + f] = n.Wh[j][f];
}

// input feature f of (tile, view, point): a cheap deterministic function, the same on the host
__host__ __device__ inline float feat(int tile, int view, int point, int f) {
    const unsigned x = (unsigned)(tile * 7919 + view * 104729 + point * 1299709 + f * 15485863);
    return (float)((x * 2654435761u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

// ---- deterministic all-reduce over the waves of the workgroup (nr_device.h block_allreduce) -------------------------------------
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

// =====================================================================================================================================
// shape A: 2 slots x 16 points, v_mfma_f32_16x16x4_f32, 4 waves
// =====================================================================================================================================
template <int KQ, int MT, int NS>
__device__ __forceinline__ void layer_a(const float* w, int lane, const float (&x)[NS][8], v4f (&acc)[NS][MT]) {
#pragma unroll
    for (int mo = 0; mo < MT; ++mo)
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const float4 a = *reinterpret_cast<const float4*>(w + ((mo * KQ + kq) * 64 + lane) * 4);
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s][mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x[s][4 * kq + 0], acc[s][mo], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s][mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x[s][4 * kq + 1], acc[s][mo], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s][mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x[s][4 * kq + 2], acc[s][mo], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s][mo] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x[s][4 * kq + 3], acc[s][mo], 0, 0, 0);
        }
}

template <bool NARROW, bool ALLRED>
__global__ void __launch_bounds__(256, 3) shape_a(const float* packed, float* out, int tiles_per_wg, int write_tile0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                       // A_END floats
    float* red = smem + A_END;              // (4 + 1) * 12 * 64
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    for (int i = threadIdx.x; i < A_END; i += 256) wl[i] = packed[i];
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
            const float4 b0 = *reinterpret_cast<const float4*>(wl + A_B1 + 4 * g), b1 = *reinterpret_cast<const float4*>(wl + A_B1 + 16 + 4 * g);
#pragma unroll
            for (int s = 0; s < 2; ++s) { acc[s][0] = (v4f){b0.x, b0.y, b0.z, b0.w}; acc[s][1] = (v4f){b1.x, b1.y, b1.z, b1.w}; }
            layer_a<2, 2, 2>(wl + A_W1, lane, x, acc);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) h1[s][i] = elu_s(acc[s][i / 4][i % 4]);
        }
        {
            v4f acc[2][2];
            const float4 b0 = *reinterpret_cast<const float4*>(wl + A_B2 + 4 * g), b1 = *reinterpret_cast<const float4*>(wl + A_B2 + 16 + 4 * g);
#pragma unroll
            for (int s = 0; s < 2; ++s) { acc[s][0] = (v4f){b0.x, b0.y, b0.z, b0.w}; acc[s][1] = (v4f){b1.x, b1.y, b1.z, b1.w}; }
            layer_a<2, 2, 2>(wl + A_W2, lane, h1, acc);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i) h2[s][i] = elu_s(acc[s][i / 4][i % 4]);
        }
        // vector head: 2 rows, per lane 8 products, then the group sum
        float head[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 w0 = *reinterpret_cast<const float4*>(wl + A_WH + j * 32 + 4 * g), w1 = *reinterpret_cast<const float4*>(wl + A_WH + j * 32 + 16 + 4 * g);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float a = h2[s][0] * w0.x;
                a = fmaf(h2[s][1], w0.y, a); a = fmaf(h2[s][2], w0.z, a); a = fmaf(h2[s][3], w0.w, a);
                a = fmaf(h2[s][4], w1.x, a); a = fmaf(h2[s][5], w1.y, a); a = fmaf(h2[s][6], w1.z, a); a = fmaf(h2[s][7], w1.w, a);
                head[s][j] = group_sum(a);
            }
        }
        float n1[2][4], n2[2][4];
        if constexpr (NARROW) {
            v4f acc[2][1];
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[s][0] = (v4f){0.f, 0.f, 0.f, 0.f};
            layer_a<2, 1, 2>(wl + A_N1, lane, h2, acc);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) n1[s][i] = elu_s(acc[s][0][i]);
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[s][0] = (v4f){0.f, 0.f, 0.f, 0.f};
            layer_a<2, 1, 2>(wl + A_N2, lane, h1, acc);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) n2[s][i] = elu_s(acc[s][0][i]);
        }
        // statistics rows: per lane the two slots are summed, then the waves: 8 channels (h2 * head0) + 3 "rgb" + 1 "mask"
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
        if (write_tile0 && tile == 0 && wave == 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) out[i * 64 + lane] = st[i];
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) keep += st[i];
    }
    if (keep == 123.456f) out[4096 + threadIdx.x] = keep;
}

// =====================================================================================================================================
// shape B: 1 slot x 32 points, v_mfma_f32_32x32x2_f32, 8 waves
// =====================================================================================================================================
template <int KQ>
__device__ __forceinline__ void layer_b(const float* w, int lane, const float (&x)[16], v16f& acc) {
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
        const float4 a = *reinterpret_cast<const float4*>(w + (kq * 64 + lane) * 4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, x[4 * kq + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, x[4 * kq + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, x[4 * kq + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, x[4 * kq + 3], acc, 0, 0, 0);
    }
}

template <bool NARROW, bool ALLRED, int NW>
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 2 : 3) shape_b(const float* packed, float* out, int tiles_per_wg, int write_tile0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                       // B_END floats
    float* red = smem + B_END;              // (NW + 1) * 20 * 64
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, c = lane & 31;
    for (int i = threadIdx.x; i < B_END; i += NW * 64) wl[i] = packed[i];
    __syncthreads();
    float keep = 0.0f;
    for (int it = 0; it < tiles_per_wg; ++it) {
        const int tile = blockIdx.x * tiles_per_wg + it;
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = feat(tile, wave, c, 8 * (i / 4) + 4 * h + i % 4);
        float h1[16], h2[16];
        {
            v16f acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = *reinterpret_cast<const float4*>(wl + B_B1 + 8 * q + 4 * h);
                acc[4 * q] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
            }
            layer_b<4>(wl + B_W1, lane, x, acc);
#pragma unroll
            for (int i = 0; i < 16; ++i) h1[i] = elu_s(acc[i]);
        }
        {
            v16f acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = *reinterpret_cast<const float4*>(wl + B_B2 + 8 * q + 4 * h);
                acc[4 * q] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
            }
            layer_b<4>(wl + B_W2, lane, h1, acc);
#pragma unroll
            for (int i = 0; i < 16; ++i) h2[i] = elu_s(acc[i]);
        }
        float head[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float a = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = *reinterpret_cast<const float4*>(wl + B_WH + j * 32 + 8 * q + 4 * h);
                a = fmaf(h2[4 * q], w.x, a); a = fmaf(h2[4 * q + 1], w.y, a); a = fmaf(h2[4 * q + 2], w.z, a); a = fmaf(h2[4 * q + 3], w.w, a);
            }
            head[j] = half_sum(a);
        }
        float n1[16], n2[16];
        if constexpr (NARROW) {
            // 8 / 16 useful rows of a 32-row tile: registers 0..3 (rows 0..7) / 0..7 (rows 0..15) carry data
            v16f acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            layer_b<4>(wl + B_N1, lane, h2, acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) n1[i] = elu_s(acc[i]);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
            layer_b<4>(wl + B_N2, lane, h1, acc);
#pragma unroll
            for (int i = 0; i < 8; ++i) n2[i] = elu_s(acc[i]);
        }
        // statistics rows of 32 points: 16 channels + 3 "rgb" + 1 "mask"
        float st[20];
#pragma unroll
        for (int i = 0; i < 16; ++i) st[i] = h2[i] * head[0];
        st[16] = head[1]; st[17] = head[0]; st[18] = head[1] * head[0]; st[19] = 1.0f;
        if constexpr (NARROW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) st[i] += n1[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) st[4 + i] += n2[i];
        }
        if constexpr (ALLRED) block_allreduce<20, NW>(st, red, wave, lane);
        if (write_tile0 && tile == 0 && wave == 0) {
#pragma unroll
            for (int i = 0; i < 20; ++i) out[i * 64 + lane] = st[i];
        }
#pragma unroll
        for (int i = 0; i < 20; ++i) keep += st[i];
    }
    if (keep == 123.456f) out[4096 + threadIdx.x] = keep;
}

// ---- host reference of tile 0 (no narrow layers): per point the all-reduced statistics ----------------------------------------------
static void host_tile0(const Net& n, int points, std::vector<float>& ch, std::vector<float>& extra) {
    ch.assign(points * 32, 0.0f); extra.assign(points * 3, 0.0f);
    for (int p = 0; p < points; ++p)
        for (int v = 0; v < 8; ++v) {
            float x[32], h1[32], h2[32], head[2];
            for (int f = 0; f < 32; ++f) x[f] = feat(0, v, p, f);
            for (int o = 0; o < 32; ++o) { float a = n.b1[o]; for (int f = 0; f < 32; ++f) a = fmaf(n.W1[o][f], x[f], a); h1[o] = elu_s_host(a); }
            for (int o = 0; o < 32; ++o) { float a = n.b2[o]; for (int f = 0; f < 32; ++f) a = fmaf(n.W2[o][f], h1[f], a); h2[o] = elu_s_host(a); }
            for (int j = 0; j < 2; ++j) { float a = 0; for (int f = 0; f < 32; ++f) a = fmaf(n.Wh[j][f], h2[f], a); head[j] = a; }
            for (int f = 0; f < 32; ++f) ch[p * 32 + f] += h2[f] * head[0];
            extra[p * 3] += head[1]; extra[p * 3 + 1] += head[0]; extra[p * 3 + 2] += head[1] * head[0];
        }
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
    std::vector<float> pa, pb;
    pack_a(*net, pa); pack_b(*net, pb);
    float *da, *db, *dout;
    CHECK(hipMalloc(&da, pa.size() * 4)); CHECK(hipMalloc(&db, pb.size() * 4)); CHECK(hipMalloc(&dout, 8192 * 4));
    CHECK(hipMemcpy(da, pa.data(), pa.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, pb.data(), pb.size() * 4, hipMemcpyHostToDevice));
    const size_t smem_a = (A_END + 5 * 12 * 64) * 4, smem_b8 = (B_END + 9 * 20 * 64) * 4, smem_b4 = (B_END + 5 * 20 * 64) * 4;
    CHECK(hipFuncSetAttribute((const void*)shape_b<false, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b8));
    CHECK(hipFuncSetAttribute((const void*)shape_b<true, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b8));
    CHECK(hipFuncSetAttribute((const void*)shape_b<true, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b8));
    CHECK(hipFuncSetAttribute((const void*)shape_b<false, false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b8));

    // ---- operand maps: tile 0 of both shapes against the host ----
    std::vector<float> out(8192), ch, extra;
    double worst_a = 0, worst_b = 0;
    host_tile0(*net, 32, ch, extra);
    hipLaunchKernelGGL((shape_a<false, true>), 1, 256, smem_a, 0, da, dout, 1, 1);
    CHECK(hipMemcpy(out.data(), dout, 8192 * 4, hipMemcpyDeviceToHost));
    for (int l = 0; l < 64; ++l) {
        const int c = l & 15, g = l >> 4;
        for (int i = 0; i < 8; ++i) worst_a = fmax(worst_a, fabs(out[i * 64 + l] - ch[c * 32 + 16 * (i / 4) + 4 * g + i % 4]));
        for (int j = 0; j < 3; ++j) worst_a = fmax(worst_a, fabs(out[(8 + j) * 64 + l] - extra[c * 3 + j]));
        worst_a = fmax(worst_a, fabs(out[11 * 64 + l] - 8.0f));
    }
    hipLaunchKernelGGL((shape_b<false, true, 8>), 1, 512, smem_b8, 0, db, dout, 1, 1);
    CHECK(hipMemcpy(out.data(), dout, 8192 * 4, hipMemcpyDeviceToHost));
    for (int l = 0; l < 64; ++l) {
        const int c = l & 31, h = l >> 5;
        for (int i = 0; i < 16; ++i) worst_b = fmax(worst_b, fabs(out[i * 64 + l] - ch[c * 32 + 8 * (i / 4) + 4 * h + i % 4]));
        for (int j = 0; j < 3; ++j) worst_b = fmax(worst_b, fabs(out[(16 + j) * 64 + l] - extra[c * 3 + j]));
        worst_b = fmax(worst_b, fabs(out[19 * 64 + l] - 8.0f));
    }
    printf("operand maps vs host (tile 0, 8 views all-reduced): shape A max abs err %.2e, shape B %.2e\n", worst_a, worst_b);

    // ---- timing at full occupancy ----
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    const int tiles = 400, reps = 5;
    printf("%d CUs, %.2f GHz; per configuration: ms per launch, ns per (point, view) chip-wide, SIMD cycles per (point, view)\n", cus, ghz);
    auto report = [&](const char* name, double ms, double pv) {
        printf("%-58s %8.3f ms  %9.5f ns/pv  %7.2f SIMD-cycles/pv\n", name, ms, ms * 1e6 / pv, ms * 1e-3 * ghz * 1e9 * cus * 4 / pv);
    };
    {
        const int grid = cus * 3 * 4;                              // 3 workgroups per CU resident, 4 rounds
        const double pv = (double)grid * tiles * 16 * 8;
        report("A 2x16 pts 16x16x4, 4 waves, chain+head+allreduce", time_kernel([&] { hipLaunchKernelGGL((shape_a<false, true>), grid, 256, smem_a, 0, da, dout, tiles, 0); }, reps), pv);
        report("A ... + narrow layers (32->8, 32->16)", time_kernel([&] { hipLaunchKernelGGL((shape_a<true, true>), grid, 256, smem_a, 0, da, dout, tiles, 0); }, reps), pv);
        report("A chain+head, no allreduce", time_kernel([&] { hipLaunchKernelGGL((shape_a<false, false>), grid, 256, smem_a, 0, da, dout, tiles, 0); }, reps), pv);
        report("A chain+head+narrow, no allreduce", time_kernel([&] { hipLaunchKernelGGL((shape_a<true, false>), grid, 256, smem_a, 0, da, dout, tiles, 0); }, reps), pv);
    }
    {
        const int grid = cus * 2 * 4;                              // 2 workgroups of 8 waves per CU resident
        const double pv = (double)grid * (tiles / 2) * 32 * 8;
        report("B 1x32 pts 32x32x2, 8 waves, chain+head+allreduce", time_kernel([&] { hipLaunchKernelGGL((shape_b<false, true, 8>), grid, 512, smem_b8, 0, db, dout, tiles / 2, 0); }, reps), pv);
        report("B ... + narrow layers (32->8, 32->16)", time_kernel([&] { hipLaunchKernelGGL((shape_b<true, true, 8>), grid, 512, smem_b8, 0, db, dout, tiles / 2, 0); }, reps), pv);
        report("B chain+head, no allreduce", time_kernel([&] { hipLaunchKernelGGL((shape_b<false, false, 8>), grid, 512, smem_b8, 0, db, dout, tiles / 2, 0); }, reps), pv);
        report("B chain+head+narrow, no allreduce", time_kernel([&] { hipLaunchKernelGGL((shape_b<true, false, 8>), grid, 512, smem_b8, 0, db, dout, tiles / 2, 0); }, reps), pv);
    }
    {   // 4 waves per workgroup (4 views per tile): isolates the MFMA shape from the 8-wave barrier
        const int grid = cus * 3 * 4;
        const double pv = (double)grid * (tiles / 2) * 32 * 4;
        report("B 1x32 pts 32x32x2, 4 waves (4 views), chain+head+allreduce", time_kernel([&] { hipLaunchKernelGGL((shape_b<false, true, 4>), grid, 256, smem_b4, 0, db, dout, tiles / 2, 0); }, reps), pv);
        report("B 4 waves ... + narrow layers", time_kernel([&] { hipLaunchKernelGGL((shape_b<true, true, 4>), grid, 256, smem_b4, 0, db, dout, tiles / 2, 0); }, reps), pv);
    }
    return 0;
}
