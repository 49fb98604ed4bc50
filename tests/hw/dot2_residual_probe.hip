#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, float* o) {
  const int t = threadIdx.x;
  float x0 = x[t], x1 = x[t+64];
  v2bf h; h[0] = (__bf16)x0; h[1] = (__bf16)x1;
  unsigned hb = __builtin_bit_cast(unsigned, h);
  // reference
  o[t] = x0 - __builtin_bit_cast(float, hb << 16); o[t+64] = x1 - __builtin_bit_cast(float, hb & 0xffff0000u);
  // builtin
  v2bf klo = __builtin_bit_cast(v2bf, 0x0000BF80u), khi = __builtin_bit_cast(v2bf, 0xBF800000u);
  o[t+128] = __builtin_amdgcn_fdot2_f32_bf16(h, klo, x0, false);
  o[t+192] = __builtin_amdgcn_fdot2_f32_bf16(h, khi, x1, false);
  // VOP3P with the constants in VGPRs
  unsigned vlo = 0x0000BF80u, vhi = 0xBF800000u; float r0, r1;
  asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r0) : "v"(hb), "v"(vlo), "v"(x0));
  asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r1) : "v"(hb), "v"(vhi), "v"(x1));
  o[t+256] = r0; o[t+320] = r1;
}
int main() {
  float hx[128], ho[384]; float *dx, *dout;
  srand(1); for (int i = 0; i < 128; ++i) hx[i] = ((float)rand()/RAND_MAX - 0.5f) * expf((float)(rand()%20-10));
  hipMalloc(&dx, 512); hipMalloc(&dout, 384*4); hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, 1, 64, 0, 0, dx, dout); hipMemcpy(ho, dout, 384*4, hipMemcpyDeviceToHost);
  int bad_b = 0, bad_v = 0;
  for (int i = 0; i < 128; ++i) { if (memcmp(&ho[i], &ho[128+i], 4)) ++bad_b; if (memcmp(&ho[i], &ho[256+i], 4)) ++bad_v; }
  printf("builtin (v_dot2c VOP2) mismatches: %d / 128; v_dot2_f32_bf16 VOP3P with VGPR constants: %d / 128\n", bad_b, bad_v);
  for (int i = 0; i < 4; ++i) printf("x %g ref %g builtin %g vop3p %g | hi half: x %g ref %g builtin %g vop3p %g\n", hx[i], ho[i], ho[128+i], ho[256+i], hx[64+i], ho[64+i], ho[192+i], ho[320+i]);
  return 0;
}
