// Ping-pong probe (gfx950): 8-wave workgroups, one per CU; waves 0-3 issue only v_mfma_f32_32x32x16_f16 (chains of NCH
// accumulators held in VGPRs or AGPRs), waves 4-7 only VALU (v_fma_f32, 8 independent chains).  Times each role alone and
// both together: does a wave of MFMAs share its SIMD with a wave of VALU work?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool AGPR, bool SWAP, int PM = 0, int PV = 0, int KIND = 0>
__global__ __launch_bounds__(512, 1) void probe(float* out, int n_mfma, int n_valu, int mode) {
  const int wave = threadIdx.x >> 6;
  bool mrole = SWAP ? wave >= 4 : wave < 4;
  __shared__ int cnt[4];
  __shared__ int simd_of[8];
  if (PM == 9) {         // roles by rank among the waves that share the SIMD (HW_ID bits 5:4), not by wave index
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const int simd = (hw >> 4) & 3;
    int rank = 0;
    if ((threadIdx.x & 63) == 0) { rank = atomicAdd(&cnt[simd], 1); simd_of[wave] = simd; }
    rank = __builtin_amdgcn_readfirstlane(rank);
    mrole = (rank & 1) == (SWAP ? 1 : 0);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0 && mode == 3) {
      for (int i = 0; i < 8; ++i) out[512 * 256 + i] = (float)simd_of[i];
    }
  }
  if (PM != 9) { if (mrole) __builtin_amdgcn_s_setprio(PM); else __builtin_amdgcn_s_setprio(PV); }
  float s = 0.f;
  if (mrole) {
    if (mode & 1) {
      f16x8 a, b;
      for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
      f32x16 c0, c1;
      for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
      for (int it = 0; it < n_mfma; it += 8) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if (AGPR) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
          } else {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
          }
        }
      }
      for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    }
  } else {
    if (mode & 2) {
      float v[8];
      for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
      float4 q4 = {0, 0, 0, 0};
      const unsigned ldsaddr = (threadIdx.x & 63) * 16;
      const float c = 1.0001f, d = 0.5f;
      for (int it = 0; it < n_valu; it += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
          else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
          else if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
          else if (KIND == 3) asm volatile("v_fma_mixlo_f16 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(c), "v"(d));
          else if (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
          else if (KIND == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
          else if (KIND == 6) { if (i == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(q4) : "v"(ldsaddr)); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d)); }
        }
        if (KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)");
      }
      for (int i = 0; i < 8; ++i) s += v[i];
      s += q4.x;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool AGPR, bool SWAP, int PM = 0, int PV = 0, int KIND = 0>
float run(float* out, int nm, int nv, int mode) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<AGPR, SWAP, PM, PV, KIND>), dim3(256), dim3(512), 0, 0, out, nm, nv, mode);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<AGPR, SWAP, PM, PV, KIND>), dim3(256), dim3(512), 0, 0, out, nm, nv, mode);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

template <bool AGPR, bool SWAP, int PM = 0, int PV = 0, int KIND = 0>
void report(float* out, const char* tag) {
  const int nm = 24 * 2000, nv = 110 * 2000;      // per wave: the MFMAs / VALU instructions of 2000 attention tiles
  const float tm = run<AGPR, SWAP, PM, PV, KIND>(out, nm, nv, 1), tv = run<AGPR, SWAP, PM, PV, KIND>(out, nm, nv, 2), tb = run<AGPR, SWAP, PM, PV, KIND>(out, nm, nv, 3);
  printf("%s: MFMA waves alone %7.1f us (%.1f ns / MFMA), VALU waves alone %7.1f us (%.2f ns / instr), together %7.1f us\n", tag,
         tm, tm * 1e3 / nm, tv, tv * 1e3 / nv, tb);
}

int main() {
  float* out;
  hipMalloc(&out, (256 * 512 + 64) * sizeof(float));
  report<false, false, 9, 0, 0>(out, "v_fma_f32      | MFMA = first wave of the SIMD ");
  report<false, true, 9, 0, 0>(out, "v_fma_f32      | MFMA = second wave of the SIMD");
  report<false, false, 9, 0, 1>(out, "v_exp_f32      | MFMA = first wave of the SIMD ");
  report<false, true, 9, 0, 1>(out, "v_exp_f32      | MFMA = second wave of the SIMD");
  report<false, false, 9, 0, 2>(out, "v_cvt_pk_f16   | MFMA = first wave of the SIMD ");
  report<false, true, 9, 0, 2>(out, "v_cvt_pk_f16   | MFMA = second wave of the SIMD");
  report<false, false, 9, 0, 3>(out, "v_fma_mixlo    | MFMA = first wave of the SIMD ");
  report<false, true, 9, 0, 3>(out, "v_fma_mixlo    | MFMA = second wave of the SIMD");
  report<false, false, 9, 0, 4>(out, "v_max3_f32     | MFMA = first wave of the SIMD ");
  report<false, true, 9, 0, 4>(out, "v_max3_f32     | MFMA = second wave of the SIMD");
  report<false, false, 9, 0, 6>(out, "ds_read + fma  | MFMA = first wave of the SIMD ");
  report<false, true, 9, 0, 6>(out, "ds_read + fma  | MFMA = second wave of the SIMD");
  return 0;
}
