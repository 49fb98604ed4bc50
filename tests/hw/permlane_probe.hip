// Probe of v_permlane16_swap / v_permlane32_swap semantics on gfx950 and of nr_group_sum (run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 -I neuray_amd/csrc tests/hw/permlane_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
#include "nr_platform.h"
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out, float* fo, const float* fi) {
    const unsigned l = threadIdx.x;
    const v2u a = __builtin_amdgcn_permlane16_swap(l, l + 100, false, false);
    const v2u b = __builtin_amdgcn_permlane32_swap(l, l + 100, false, false);
    out[l] = a.x; out[64 + l] = a.y; out[128 + l] = b.x; out[192 + l] = b.y;
    float acc[4];
    for (int j = 0; j < 4; ++j) acc[j] = nr_group_sum(fi[j * 64 + l] * 1.5f);
    for (int j = 0; j < 4; ++j) fo[j * 64 + l] = acc[j] + 0.25f;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    float *fo, *fi; hipMalloc(&fo, 1024); hipMalloc(&fi, 1024);
    float hin[256]; for (int i = 0; i < 256; ++i) hin[i] = (float)((i * 37) % 101) * 0.125f;
    hipMemcpy(fi, hin, sizeof(hin), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, fo, fi);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    float hf[256]; hipMemcpy(hf, fo, sizeof(hf), hipMemcpyDeviceToHost);
    const char* names[4] = {"p16.x", "p16.y", "p32.x", "p32.y"};
    for (int j = 0; j < 4; ++j) { printf("%s:", names[j]); for (int i = 0; i < 64; ++i) printf(" %u", h[64 * j + i]); printf("\n"); }
    int bad = 0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 64; ++i) {
            const int c = i & 15;
            const float* x = hin + j * 64;
            const float want = ((x[c] * 1.5f + x[c + 16] * 1.5f) + (x[c + 32] * 1.5f + x[c + 48] * 1.5f)) + 0.25f;
            if (hf[j * 64 + i] != want) { if (bad < 8) printf("group_sum mismatch j=%d lane=%d got %g want %g\n", j, i, hf[j * 64 + i], want); ++bad; }
        }
    printf("group_sum mismatches: %d\n", bad);
    return bad != 0;
}
