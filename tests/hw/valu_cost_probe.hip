// Hardware probe (not a test): issue cost of the VALU instructions the three-way operand split is made of (and of candidates that would
// shorten it), as independent instruction streams on one wave per SIMD:  hipcc --offload-arch=gfx950 -O3 tests/hw/valu_cost_probe.hip -o /tmp/vcp && /tmp/vcp
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int OP> __global__ void __launch_bounds__(256) k(float* out, int n) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned m = 0xffff0000u, sixteen = 16, neg1 = 0xbf80u;       // bf16 pair (-1.0, 0)
    for (int i = 0; i < n; ++i) {
#define R8(S) asm volatile(S :: ); 
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (OP == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %4\n v_cvt_pk_bf16_f32 %4, %4, %5\n v_cvt_pk_bf16_f32 %5, %5, %6\n v_cvt_pk_bf16_f32 %6, %6, %7\n v_cvt_pk_bf16_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (OP == 2) asm volatile("v_lshlrev_b32 %0, 16, %0\n v_lshlrev_b32 %1, 16, %1\n v_lshlrev_b32 %2, 16, %2\n v_lshlrev_b32 %3, 16, %3\n v_lshlrev_b32 %4, 16, %4\n v_lshlrev_b32 %5, 16, %5\n v_lshlrev_b32 %6, 16, %6\n v_lshlrev_b32 %7, 16, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (OP == 3) asm volatile("v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        if (OP == 4) asm volatile("v_and_b32 %0, 0xffff0000, %0\n v_and_b32 %1, 0xffff0000, %1\n v_and_b32 %2, 0xffff0000, %2\n v_and_b32 %3, 0xffff0000, %3\n v_and_b32 %4, 0xffff0000, %4\n v_and_b32 %5, 0xffff0000, %5\n v_and_b32 %6, 0xffff0000, %6\n v_and_b32 %7, 0xffff0000, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (OP == 5) asm volatile("v_sub_f32 %0, %0, %1\n v_sub_f32 %1, %1, %2\n v_sub_f32 %2, %2, %3\n v_sub_f32 %3, %3, %4\n v_sub_f32 %4, %4, %5\n v_sub_f32 %5, %5, %6\n v_sub_f32 %6, %6, %7\n v_sub_f32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (OP == 6) asm volatile("v_dot2_f32_bf16 %0, %1, %8, %0\n v_dot2_f32_bf16 %1, %2, %8, %1\n v_dot2_f32_bf16 %2, %3, %8, %2\n v_dot2_f32_bf16 %3, %4, %8, %3\n v_dot2_f32_bf16 %4, %5, %8, %4\n v_dot2_f32_bf16 %5, %6, %8, %5\n v_dot2_f32_bf16 %6, %7, %8, %6\n v_dot2_f32_bf16 %7, %0, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(neg1));
        if (OP == 7) asm volatile("v_perm_b32 %0, %0, %1, %8\n v_perm_b32 %1, %1, %2, %8\n v_perm_b32 %2, %2, %3, %8\n v_perm_b32 %3, %3, %4, %8\n v_perm_b32 %4, %4, %5, %8\n v_perm_b32 %5, %5, %6, %8\n v_perm_b32 %6, %6, %7, %8\n v_perm_b32 %7, %7, %0, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        if (OP == 8) asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)sixteen;
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int OP> int run(const char* name, float* d, int cus, double ghz) {
    const int n = 20000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, cus, 256, 0, 0, d, n); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k<OP>, cus, 256, 0, 0, d, n); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %5.2f SIMD cycles per instruction (one wave per SIMD, 8 independent chains)\n", name, ms * 1e-3 * ghz * 1e9 / ((double)n * 8));
    return 0;
}
int main() {
    float* d; CHECK(hipMalloc(&d, 4096));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    run<0>("v_fma_f32", d, cus, ghz); run<1>("v_cvt_pk_bf16_f32", d, cus, ghz); run<2>("v_lshlrev_b32 (inline 16)", d, cus, ghz);
    run<3>("v_and_b32 (mask in a VGPR)", d, cus, ghz); run<4>("v_and_b32 (32-bit literal)", d, cus, ghz); run<5>("v_sub_f32", d, cus, ghz);
    run<6>("v_dot2_f32_bf16", d, cus, ghz); run<7>("v_perm_b32", d, cus, ghz); run<8>("v_exp_f32", d, cus, ghz);
    return 0;
}
