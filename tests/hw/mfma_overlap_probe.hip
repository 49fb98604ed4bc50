// Hardware probe (not a test): can VALU work of ONE wave hide behind fp32 MFMAs of ANOTHER wave on the same SIMD?
// (MI355X_MICROARCH.md: "a MFMA-only wave and a VALU-only wave on the same CU run concurrently" - measured there with
// bf16 MFMAs.)  Workgroups of 512 threads = 2 waves per SIMD.  Roles by wave index parity... no: by wave >> 2, so that every
// SIMD hosts one wave of each role (waves are dealt round-robin to the 4 SIMDs).
//   role M: independent v_mfma chains (NCH accumulators, back to back)       role V: independent v_fma chains (NV accumulators)
// Configurations: M alone (V waves exit at once), V alone, M + V together, and M+V mixed inside every wave (the shape of
// the point kernel: each wave interleaves its own MFMAs and VALU).  If the pipes overlap: t(M+V) ~ max(t(M), t(V)); if the
// fp32 MFMA shares the VALU's issue/ALU: t(M+V) ~ t(M) + t(V).  Same for the bf16 MFMA as the control.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));

template <bool BF16>
__device__ __forceinline__ v4f mfma(float a, float b, v4f c) {
    if constexpr (BF16) {
        const v4s av = {(short)__float_as_int(a), 1, 2, 3}, bv = {(short)__float_as_int(b), 3, 2, 1};
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
}

// MODE 0: roles split over the two waves of a SIMD (do_m / do_v select which roles run)    MODE 1: every wave does both, interleaved
template <bool BF16, int MODE>
__global__ void __launch_bounds__(512) probe(const float* in, float* out, int iters, int do_m, int do_v, int n_mfma, int n_valu) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool role_m = MODE == 1 || (wave >> 2) == 0, role_v = MODE == 1 || (wave >> 2) == 1;
    v4f acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = (v4f){in[lane], 0.f, 0.f, 0.f};
    float f[8];
    for (int k = 0; k < 8; ++k) f[k] = in[lane + k];
    const float a = in[lane], b = in[lane + 64];
    if (MODE == 0 && !((role_m && do_m) || (role_v && do_v))) return;
    for (int it = 0; it < iters; ++it) {
        if (role_m && do_m) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {       // n_mfma is a multiple of 4: 4 independent chains, interleaved
                if (q * 4 >= n_mfma) break;
                acc[0] = mfma<BF16>(a, b, acc[0]); acc[1] = mfma<BF16>(a, b, acc[1]);
                acc[2] = mfma<BF16>(a, b, acc[2]); acc[3] = mfma<BF16>(a, b, acc[3]);
                if (MODE == 1 && do_v) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) { if (j * 8 >= n_valu / 8 * 1) break; }
                }
            }
        }
        if (role_v && do_v) {
            for (int j = 0; j < n_valu / 8; ++j) {
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = __builtin_fmaf(f[k], 1.0001f, 0.5f);      // 8 independent chains
            }
        }
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    for (int k = 0; k < 8; ++k) s += f[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MODE 1 proper: per 4 MFMAs of a wave, VPM * 4 independent VALU ops in the same wave
template <bool BF16, int VPM>
__global__ void __launch_bounds__(512) mixed(const float* in, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    v4f acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = (v4f){in[lane], 0.f, 0.f, 0.f};
    float f[8];
    for (int k = 0; k < 8; ++k) f[k] = in[lane + k];
    const float a = in[lane], b = in[lane + 64];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            acc[0] = mfma<BF16>(a, b, acc[0]);
#pragma unroll
            for (int k = 0; k < VPM; ++k) f[(4 * q + k) & 7] = __builtin_fmaf(f[(4 * q + k) & 7], 1.0001f, 0.5f);
            acc[1] = mfma<BF16>(a, b, acc[1]);
#pragma unroll
            for (int k = 0; k < VPM; ++k) f[(4 * q + k + 2) & 7] = __builtin_fmaf(f[(4 * q + k + 2) & 7], 1.0001f, 0.5f);
            acc[2] = mfma<BF16>(a, b, acc[2]);
#pragma unroll
            for (int k = 0; k < VPM; ++k) f[(4 * q + k + 4) & 7] = __builtin_fmaf(f[(4 * q + k + 4) & 7], 1.0001f, 0.5f);
            acc[3] = mfma<BF16>(a, b, acc[3]);
#pragma unroll
            for (int k = 0; k < VPM; ++k) f[(4 * q + k + 6) & 7] = __builtin_fmaf(f[(4 * q + k + 6) & 7], 1.0001f, 0.5f);
        }
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    for (int k = 0; k < 8; ++k) s += f[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// cost of one VALU instruction of a given kind issued between the MFMAs of the same wave (OP: 0 v_fma_f32, 1 v_exp_f32,
// 2 v_pk_fma_f32 (two results), 3 v_med3_f32, 4 v_rcp_f32), NV of them per MFMA, independent chains
template <int OP, int NV>
__global__ void __launch_bounds__(512) opcost(const float* in, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    v4f acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = (v4f){in[lane], 0.f, 0.f, 0.f};
    typedef float v2f __attribute__((ext_vector_type(2)));
    float f[8]; v2f p[8];
    for (int k = 0; k < 8; ++k) { f[k] = in[lane + k]; p[k] = (v2f){in[lane + k], in[lane + k + 8]}; }
    const float a = in[lane], b = in[lane + 64];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[m] = mfma<false>(a, b, acc[m]);
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int j = (4 * q + m * NV + k) & 7;
                    if (OP == 0) f[j] = __builtin_fmaf(f[j], 1.0001f, 0.5f);
                    if (OP == 1) f[j] = __builtin_amdgcn_exp2f(f[j]);
                    if (OP == 2) p[j] = __builtin_elementwise_fma(p[j], (v2f){1.0001f, 1.0001f}, (v2f){0.5f, 0.5f});
                    if (OP == 3) f[j] = __builtin_amdgcn_fmed3f(f[j], 0.25f, 0.0f);
                    if (OP == 4) f[j] = __builtin_amdgcn_rcpf(f[j]);
                }
            }
        }
    }
    float s = 0.f;
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    for (int k = 0; k < 8; ++k) s += f[k] + p[k].x + p[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
float time_ms(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(10);
    hipEventRecord(e0); launch(4000); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <bool BF16>
void roles(const float* in, float* out, int n_valu) {
    const int n_mfma = 32;
    auto run = [&](int dm, int dv) {
        return time_ms([&](int it) { probe<BF16, 0><<<256, 512>>>(in, out, it, dm, dv, n_mfma, n_valu); });
    };
    const float tm = run(1, 0), tv = run(0, 1), tb = run(1, 1);
    printf("%s MFMA wave + VALU wave per SIMD, %3d independent v_fma per 32 MFMA:  M alone %6.2f ms  V alone %6.2f ms  together %6.2f ms"
           "  (max %.2f, sum %.2f) -> overlap %.0f%%\n", BF16 ? "bf16" : "fp32", n_valu, tm, tv, tb, tm > tv ? tm : tv, tm + tv,
           100.0 * (tm + tv - tb) / (tm < tv ? tm : tv));
}

template <bool BF16, int VPM>
void mix(const float* in, float* out, float base) {
    const float t = time_ms([&](int it) { mixed<BF16, VPM><<<256, 512>>>(in, out, it); });
    printf("%s same wave, %d independent v_fma per MFMA (2 waves/SIMD): %6.2f ms = %.2fx the MFMA-only time\n", BF16 ? "bf16" : "fp32", VPM, t, t / base);
}

int main() {
    float *in, *out; hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 512 * 4); hipMemset(in, 0, 4096 * 4);
    for (int nv : {32, 64, 128, 256}) roles<false>(in, out, nv);
    for (int nv : {32, 128, 256}) roles<true>(in, out, nv);
    const float b32 = time_ms([&](int it) { mixed<false, 0><<<256, 512>>>(in, out, it); });
    printf("fp32 MFMA only, 2 waves/SIMD: %.2f ms (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", b32, b32 * 1e-3 * 2.4e9 / (4000.0 * 32 * 2));
    mix<false, 1>(in, out, b32); mix<false, 2>(in, out, b32); mix<false, 4>(in, out, b32); mix<false, 8>(in, out, b32);
    {
        const float base = time_ms([&](int it) { opcost<0, 0><<<256, 512>>>(in, out, it); });
        const double mf = 4000.0 * 32 * 2;                       // MFMAs per SIMD (2 waves)
        auto cost = [&](const char* name, float t, int nv) {
            printf("%-14s %d per MFMA: %6.2f ms -> %.1f SIMD cycles per instruction\n", name, nv, t, (t - base) * 1e-3 * 2.4e9 / (mf * nv));
        };
        cost("v_fma_f32", time_ms([&](int it) { opcost<0, 4><<<256, 512>>>(in, out, it); }), 4);
        cost("v_exp_f32", time_ms([&](int it) { opcost<1, 4><<<256, 512>>>(in, out, it); }), 4);
        cost("v_pk_fma_f32", time_ms([&](int it) { opcost<2, 4><<<256, 512>>>(in, out, it); }), 4);
        cost("v_med3_f32", time_ms([&](int it) { opcost<3, 4><<<256, 512>>>(in, out, it); }), 4);
        cost("v_rcp_f32", time_ms([&](int it) { opcost<4, 4><<<256, 512>>>(in, out, it); }), 4);
    }
    const float b16 = time_ms([&](int it) { mixed<true, 0><<<256, 512>>>(in, out, it); });
    printf("bf16 MFMA only, 2 waves/SIMD: %.2f ms (%.1f cycles per MFMA per SIMD)\n", b16, b16 * 1e-3 * 2.4e9 / (4000.0 * 32 * 2));
    mix<true, 1>(in, out, b16); mix<true, 2>(in, out, b16); mix<true, 4>(in, out, b16);
    return 0;
}
