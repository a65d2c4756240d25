// VALU issue rates on gfx950: v_fma_f32 / v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 / v_permlane32_swap, W waves per SIMD,
// 16 independent chains per wave.  Wall-clock (hipEvent) -> nanoseconds per wave-instruction per SIMD and TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/build/valu_rate tools/probes/valu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void rate(float* out, int iters) {
  v2f v[16];
  for (int i = 0; i < 16; ++i) v[i] = (v2f){threadIdx.x * 0.001f + i, 0.25f * i};
  v2f c = (v2f){1.0001f, 0.9999f}, d = (v2f){1e-6f, -1e-6f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i].x) : "v"(c.x), "v"(d.x)); }
      else if (KIND == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d)); }
      else if (KIND == 2) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(d)); }
      else if (KIND == 3) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c)); }
      else if (KIND == 4) { asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i].x), "+v"(v[i].y)); }
      else if (KIND == 5) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i].x) : "v"(d.x)); }
      else if (KIND == 6) { asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[i].x), "+v"(v[i].y)); }
      else { asm volatile("v_mov_b32 %0, %1" : "=v"(v[i].x) : "v"(v[(i + 1) & 15].y)); }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, float* out, double flops_per_lane) {
  const int iters = 4000;
  for (int w = 1; w <= 4; ++w) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(rate<KIND>, dim3(256), dim3(256 * w), 0, 0, out, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL(rate<KIND>, dim3(256), dim3(256 * w), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double inst_per_simd = (double)iters * 16 * w;      // wave-instructions per SIMD
    const double ns = ms * 1e6 / inst_per_simd;
    const double tf = flops_per_lane * 64 * inst_per_simd * 1024 / (ms * 1e-3) / 1e12;
    printf("%-22s %d waves/SIMD: %.3f ms, %.2f ns per wave-instruction per SIMD (= %.2f clocks at 2.4 GHz)%s", name, w, ms, ns,
           ns * 2.4, flops_per_lane > 0 ? "" : "\n");
    if (flops_per_lane > 0) printf(", %.1f TFLOP/s\n", tf);
  }
}

int main() {
  float* out;
  hipMalloc(&out, 1 << 22);
  run<0>("v_fma_f32", out, 2);
  run<1>("v_pk_fma_f32", out, 4);
  run<2>("v_pk_add_f32", out, 2);
  run<3>("v_pk_mul_f32", out, 2);
  run<5>("v_add_f32", out, 1);
  run<7>("v_mov_b32", out, 0);
  run<4>("v_permlane32_swap_b32", out, 0);
  run<6>("v_permlane16_swap_b32", out, 0);
  return 0;
}
