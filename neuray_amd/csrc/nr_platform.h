// Platform layer: the only place that touches HIP builtins directly.
//
// Product build: hipcc --offload-arch=gfx950 (CDNA4, wave64).  The same kernel sources also build
// with -DNEURAY_EMU against tests/emu/hip_emu.h (a CPU fiber emulator used ONLY by the CPU test-suite
// to check kernel logic against the oracle; it is never shipped, loaded by neuray_amd, or timed).
#pragma once

#ifdef NEURAY_EMU
#include "hip_emu.h"
#define NR_UNIFORM(x) (x)
#define NR_PRAGMA_UNROLL
#else
#include <hip/hip_runtime.h>

typedef float v4f __attribute__((ext_vector_type(4)));

// fp32-in / fp32-accumulate matrix core op: D(16x16) = A(16x4) * B(4x16) + C.
//   lane l supplies A[m = l&15][k = l>>4] and B[k = l>>4][n = l&15];
//   lane l, reg r receives D[m = 4*(l>>4) + r][n = l&15].
// Exact fp32 (k-ordered fma chain), 32-cycle issue per SIMD: v_mfma_f32_16x16x4_f32.
__device__ __forceinline__ v4f nr_mfma16(float a, float b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// wave-uniform value -> SGPR (lets hipcc use scalar loads for per-view constants)
#define NR_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#define NR_PRAGMA_UNROLL _Pragma("unroll")
#define NR_DYNAMIC_SMEM(type, name) \
    extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; \
    type* name = reinterpret_cast<type*>(name##_raw)
#define NR_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, (grid), (block), (smem), (hipStream_t)(stream), __VA_ARGS__)
#endif

#define NR_WAVE 64

// ---- exactly rounded single operations (the "rounding contract") --------------------------------------
// HIP's __fadd_rn/__fmul_rn are plain `a + b` / `a * b` compiled with fp-contract=fast, so a product feeding a
// sum is still fused into an FMA after inlining.  These helpers are compiled with contraction off (the fmul and
// fadd carry no `contract` flag, so they are never fused), which makes the camera algebra bit-identical to the
// numpy oracle.  Division and sqrt are IEEE-correct under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt.
namespace nr {
#ifdef NEURAY_EMU
__device__ __forceinline__ float rn_mul(float a, float b) { volatile float r = a * b; return r; }
__device__ __forceinline__ float rn_add(float a, float b) { volatile float r = a + b; return r; }
__device__ __forceinline__ float rn_sub(float a, float b) { volatile float r = a - b; return r; }
__device__ __forceinline__ float rn_div(float a, float b) { volatile float r = a / b; return r; }
__device__ __forceinline__ float rn_sqrt(float a) { return sqrtf(a); }
#else
#pragma clang fp contract(off)
__device__ __forceinline__ float rn_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float rn_add(float a, float b) { return a + b; }
__device__ __forceinline__ float rn_sub(float a, float b) { return a - b; }
__device__ __forceinline__ float rn_div(float a, float b) { return a / b; }
__device__ __forceinline__ float rn_sqrt(float a) { return __builtin_sqrtf(a); }
#pragma clang fp contract(fast)
#endif
}  // namespace nr
