// Co-issue probe (gfx950): per wave a loop of one v_mfma_f32_32x32x16_f16 followed by F independent VALU fillers
// (v_fma_f32 / v_exp_f32 / v_cvt_pk), with W waves per SIMD.  Prints shader cycles per MFMA for each (W, F, kind).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/build/mfma_coissue tools/probes/mfma_coissue.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F, int KIND, int NACC>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  const float c = 1.0001f, d = 0.5f;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m % NACC], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < F; ++f) {
        float& x = v[(m * F + f) & 7];
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d));
        else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        else asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(c));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int F, int KIND, int NACC>
void run(int wpg /* workgroups per CU: 4-wave workgroups, 1 wave per SIMD each */, float* out, long long* cyc) {
  const int iters = 2000, nb = 256 * wpg;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<F, KIND, NACC>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<F, KIND, NACC>), dim3(nb), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(nb);
  hipMemcpy(h.data(), cyc, nb * sizeof(long long), hipMemcpyDeviceToHost);
  double avg = 0;
  for (auto x : h) avg += x;
  avg /= nb;
  // s_memtime / readcyclecounter ticks at 100 MHz on gfx9: report wall time per MFMA per SIMD instead
  const double ns_per_mfma_simd = ms * 1e6 / (iters * 8.0 * wpg);
  printf("waves/SIMD %d  fillers %d kind %d acc-chains %d : %7.2f ns per MFMA per SIMD (%.1f cycles at 2.1 GHz), kernel %.3f ms\n",
         wpg, F, KIND, NACC, ns_per_mfma_simd, ns_per_mfma_simd * 2.1, ms);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 256 * sizeof(float));
  hipMalloc(&cyc, 1024 * sizeof(long long));
  for (int w = 1; w <= 2; ++w) {
    run<0, 0, 4>(w, out, cyc);
    run<0, 0, 1>(w, out, cyc);
    run<2, 0, 4>(w, out, cyc);
    run<4, 0, 4>(w, out, cyc);
    run<5, 0, 4>(w, out, cyc);
    run<6, 0, 4>(w, out, cyc);
    run<8, 0, 4>(w, out, cyc);
    run<12, 0, 4>(w, out, cyc);
    run<4, 1, 4>(w, out, cyc);
    run<4, 2, 4>(w, out, cyc);
    run<6, 0, 1>(w, out, cyc);
  }
  return 0;
}
