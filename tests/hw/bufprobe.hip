// Hardware probe (not a test): semantics of raw buffer loads as the kernels use them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ void probe(const float* W, int nfloats, const int* voffs, const int* soffs, int n, float* out, float* out_ref, int flags_variant) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, nfloats * 4, 0x00020000);
    int lane = threadIdx.x & 63;
    for (int i = 0; i < n; ++i) {
        int vo = voffs[i] + lane * 16, so = __builtin_amdgcn_readfirstlane(soffs[i]);
        v4u u = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
        typedef float v4f_ __attribute__((ext_vector_type(4)));
        v4f_ f = __builtin_bit_cast(v4f_, u);
        out[(i * 64 + lane) * 4 + 0] = f.x; out[(i * 64 + lane) * 4 + 1] = f.y;
        out[(i * 64 + lane) * 4 + 2] = f.z; out[(i * 64 + lane) * 4 + 3] = f.w;
        const float* p = (const float*)((const char*)W + vo + so);
        for (int k = 0; k < 4; ++k) out_ref[(i * 64 + lane) * 4 + k] = p[k];
    }
}
// constant soffsets as the layer code uses them
template <int SOFF> __global__ void probe_const(const float* W, int nfloats, float* out, float* out_ref) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, nfloats * 4, 0x00020000);
    int lane = threadIdx.x & 63;
    v4u u = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, SOFF, 0);
    out[lane * 4 + 0] = __builtin_bit_cast(float, u.x); out[lane * 4 + 1] = __builtin_bit_cast(float, u.y);
    out[lane * 4 + 2] = __builtin_bit_cast(float, u.z); out[lane * 4 + 3] = __builtin_bit_cast(float, u.w);
    const float* p = (const float*)((const char*)W + lane * 16 + SOFF);
    for (int k = 0; k < 4; ++k) out_ref[lane * 4 + k] = p[k];
}
int main() {
    const int nf = 4 * 1024 * 1024;
    std::vector<float> h(nf); for (int i = 0; i < nf; ++i) h[i] = (float)i;
    float *W, *out, *ref; int *vo, *so;
    hipMalloc(&W, nf * 4); hipMemcpy(W, h.data(), nf * 4, hipMemcpyHostToDevice);
    int hv[] = {0, 0, 0, 0, 1024, 5000000, 12345 * 16, 0, 0};
    int hs[] = {0, 16, 4096, 4100 * 4, 170000, 16, 0, 100000 * 4, 4095};
    hs[8] = 4092;
    int n = 9;
    hipMalloc(&vo, n * 4); hipMalloc(&so, n * 4); hipMemcpy(vo, hv, n * 4, hipMemcpyHostToDevice); hipMemcpy(so, hs, n * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, n * 256 * 4); hipMalloc(&ref, n * 256 * 4);
    probe<<<1, 64>>>(W, nf, vo, so, n, out, ref, 0);
    std::vector<float> a(n * 256), b(n * 256);
    hipMemcpy(a.data(), out, n * 256 * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), ref, n * 256 * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) { int bad = 0; for (int k = 0; k < 256; ++k) bad += a[i * 256 + k] != b[i * 256 + k];
        printf("runtime voff=%d soff=%d : %d/256 mismatches\n", hv[i], hs[i], bad);
        if (i < 2) { for (int k = 0; k < 12; ++k) printf("  [%d] got %g want %g\n", k, a[i * 256 + k], b[i * 256 + k]); } }
    auto chk = [&](const char* name) { hipMemcpy(a.data(), out, 1024, hipMemcpyDeviceToHost); hipMemcpy(b.data(), ref, 1024, hipMemcpyDeviceToHost);
        int bad = 0; for (int k = 0; k < 256; ++k) bad += a[k] != b[k]; printf("const %s: %d/256 mismatches (got[0]=%g want %g)\n", name, bad, a[0], b[0]); };
    probe_const<0><<<1, 64>>>(W, nf, out, ref); chk("0");
    probe_const<1024><<<1, 64>>>(W, nf, out, ref); chk("1024");
    probe_const<4096><<<1, 64>>>(W, nf, out, ref); chk("4096");
    probe_const<69632><<<1, 64>>>(W, nf, out, ref); chk("69632");
    probe_const<171984><<<1, 64>>>(W, nf, out, ref); chk("171984");
    return 0;
}
