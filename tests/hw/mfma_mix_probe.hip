// Hardware probe (not a test): fp32 MFMA 16x16x4 throughput when mixed with LDS fragment reads and VALU filler,
// at a chosen number of waves per SIMD.  Prints % of the 32-cycle-per-MFMA peak for each mix.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

// MODE 0: A operand from registers (loop invariant)         MODE 1: A from ds_read_b128, 1 quad prefetched
// MODE 2: MODE 1 + VALU filler (VF plain VALU ops per MFMA)  MODE 3: MODE 0 + VALU filler
// MODE 4: MODE 1 but two independent accumulator chains      MODE 5: A from registers, NACC chains, ELU epilogue every 16 MFMAs
template <int MODE, int VF, int NACC>
__global__ void __launch_bounds__(256) probe(const float* in, float* out, int iters) {
    __shared__ float4 lds[64 * 32];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) lds[i] = make_float4(in[i & 255], 0.5f, 0.25f, 0.125f);
    __syncthreads();
    v4f acc[NACC];
    for (int k = 0; k < NACC; ++k) acc[k] = (v4f){0.f, 0.f, 0.f, 0.f};
    float x[4] = {in[lane], in[lane + 64], in[lane + 128], in[lane + 192]};
    float filler = in[lane];
    float4 a = lds[lane];
    float4 nxt = a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (MODE == 1 || MODE == 2 || MODE == 4) { nxt = lds[((q + 1 + it) & 31) * 64 + lane]; __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int k = 0; k < NACC; ++k) {
                acc[k] = MFMA(a.x, x[0], acc[k]);
                if (MODE == 2 || MODE == 3) { for (int f = 0; f < VF; ++f) filler = filler * 1.0001f + 0.5f; }
                acc[k] = MFMA(a.y, x[1], acc[k]);
                if (MODE == 2 || MODE == 3) { for (int f = 0; f < VF; ++f) filler = filler * 1.0001f + 0.5f; }
                acc[k] = MFMA(a.z, x[2], acc[k]);
                if (MODE == 2 || MODE == 3) { for (int f = 0; f < VF; ++f) filler = filler * 1.0001f + 0.5f; }
                acc[k] = MFMA(a.w, x[3], acc[k]);
                if (MODE == 2 || MODE == 3) { for (int f = 0; f < VF; ++f) filler = filler * 1.0001f + 0.5f; }
            }
            if (MODE == 1 || MODE == 2 || MODE == 4) a = nxt;
        }
        if (MODE == 5) {
#pragma unroll
            for (int k = 0; k < NACC; ++k)
                for (int r = 0; r < 4; ++r) { float v = acc[k][r]; acc[k][r] = __builtin_amdgcn_fmed3f(v, __builtin_amdgcn_exp2f(v * 1.44f) - 1.0f, 0.0f); }
            x[0] = acc[0][0]; x[1] = acc[0][1]; x[2] = acc[NACC - 1][2]; x[3] = acc[NACC - 1][3];
        }
    }
    float s = filler;
    for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int VF, int NACC>
void run(const char* name, int blocks_per_cu, const float* in, float* out) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, VF, NACC><<<grid, 256>>>(in, out, 10);
    hipEventRecord(e0); probe<MODE, VF, NACC><<<grid, 256>>>(in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 16 * 4 * NACC * blocks_per_cu;       // one wave per SIMD per block
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-44s waves/SIMD %d: %6.2f ms  %5.1f cycles/MFMA/SIMD  (%.0f%% of the 32-cycle peak at 2.4 GHz)\n", name, blocks_per_cu, ms,
           cyc / mfma_per_simd, 100.0 * 32.0 * mfma_per_simd / cyc);
}

int main() {
    float *in, *out; hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 8 * 256 * 4); hipMemset(in, 0, 4096 * 4);
    for (int w = 1; w <= 3; w += 2) {
        run<0, 0, 1>("regs A, 1 chain", w, in, out);
        run<0, 0, 2>("regs A, 2 chains", w, in, out);
        run<1, 0, 2>("LDS A (b128, prefetch 1), 2 chains", w, in, out);
        run<3, 2, 2>("regs A, 2 chains, 2 VALU/MFMA", w, in, out);
        run<3, 4, 2>("regs A, 2 chains, 4 VALU/MFMA", w, in, out);
        run<3, 6, 2>("regs A, 2 chains, 6 VALU/MFMA", w, in, out);
        run<2, 4, 2>("LDS A, 2 chains, 4 VALU/MFMA", w, in, out);
        run<5, 0, 2>("regs A, 2 chains, ELU epilogue /16 MFMA", w, in, out);
    }
    return 0;
}
