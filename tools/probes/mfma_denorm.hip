// Does v_mfma_f32_32x32x16_f16 keep fp16 denormal operands (gfx950)?  A[i][k] = 2^-20 (a denormal half), B = 1: every C entry
// should be 16 * 2^-20 = 1.526e-5 if denormals are honoured, 0 if they are flushed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)9.5367431640625e-07f; b[e] = (_Float16)1.0f; }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  out[threadIdx.x] = c[0];
  f16x8 a2;
  for (int e = 0; e < 8; ++e) a2[e] = (_Float16)5.9604644775390625e-08f;   // 2^-24, the smallest denormal
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b, c, 0, 0, 0);
  out[64 + threadIdx.x] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("2^-20 operands: C = %g (expected %g if denormals are kept)\n", h[0], 16 * 9.5367431640625e-07);
  printf("2^-24 operands: C = %g (expected %g)\n", h[64], 16 * 5.9604644775390625e-08);
  return 0;
}
