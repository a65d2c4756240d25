// Measured-peak micro-benchmarks so that roofline fractions can be quoted against what THIS chip sustains,
// next to the datasheet figures: a register-only fp32 MFMA loop and a float4 streaming copy.
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

__global__ __launch_bounds__(256) void peak_mfma_f32_kernel(float* __restrict__ sink, int iters) {
  f32x16 a0, a1, a2, a3;
#pragma unroll
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
  const float x = 1.0f + (float)(threadIdx.x & 7) * 0.125f, y = 0.5f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  if (s == 12345.678f) sink[0] = s;      // keeps the loop alive, never true
}

typedef _Float16 pk_f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void peak_mfma_f16_kernel(float* __restrict__ sink, int iters) {
  f32x16 a0, a1, a2, a3;
#pragma unroll
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
  pk_f16x8 x, y;
#pragma unroll
  for (int e = 0; e < 8; ++e) {        // full-range random-looking operands (zero operands clock higher: DVFS)
    x[e] = (_Float16)(0.37f * (float)(((threadIdx.x * 7 + e * 13) % 23) - 11));
    y[e] = (_Float16)(0.11f * (float)(((threadIdx.x * 5 + e * 3) % 19) - 9));
  }
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a3, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(256) void peak_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) dst[i] = src[i];
}

// tuning variants of the streaming copy (eav_peak_copy_variant): U independent 16-byte loads in flight per lane,
// optionally non-temporal
template <int U, bool NT>
__global__ __launch_bounds__(256) void peak_copy_u_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                          int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f32x4 v[U];
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s4 + i + u * stride) : s4[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) __builtin_nontemporal_store(v[u], d4 + i + u * stride);
      else d4[i + u * stride] = v[u];
    }
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}

// L2 -> CU read ceilings: every wave reads 1-KB chunks (16 B per lane) of an L2-resident footprint, 8 loads in flight.
// MODE 0: global_load_dwordx4 into VGPRs; MODE 1: global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip)
template <int MODE>
__global__ __launch_bounds__(256) void peak_l2_read_kernel(const unsigned char* __restrict__ src, int footprint_kb,
                                                           int iters, float* __restrict__ sink) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 8 * 1024];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned chunk = (blockIdx.x * 4 + wave) * 8u;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)((chunk + u) % (unsigned)footprint_kb) * 1024 + lane * 16);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(src + (size_t)((chunk + u) % (unsigned)footprint_kb) * 1024 + lane * 16),
            (__attribute__((address_space(3))) void*)(lds + (wave * 8 + u) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    chunk += gridDim.x * 32u;
  }
  if (MODE == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc[0] = reinterpret_cast<const float*>(lds)[threadIdx.x];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

}  // namespace

// bytes read = blocks * 4 waves * iters * 8 KB
extern "C" int eav_peak_l2_read(const void* src, int footprint_kb, int mode, int iters, int blocks, float* sink,
                                void* stream) {
  EAV_REQUIRE(src && sink && footprint_kb > 0 && iters > 0 && blocks > 0, "eav_peak_l2_read: bad arguments");
  if (mode == 0)
    hipLaunchKernelGGL(peak_l2_read_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)src, footprint_kb, iters, sink);
  else
    hipLaunchKernelGGL(peak_l2_read_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)src, footprint_kb, iters, sink);
  EAV_CHECK_LAUNCH("eav_peak_l2_read");
  return EAV_OK;
}

extern "C" int eav_peak_copy_variant(const float* src, float* dst, int64_t n, int variant, int blocks, void* stream) {
  EAV_REQUIRE(src && dst && n > 0 && (n & 3) == 0 && blocks > 0, "eav_peak_copy_variant: bad arguments");
  const float4* s = reinterpret_cast<const float4*>(src);
  float4* d = reinterpret_cast<float4*>(dst);
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case 1: hipLaunchKernelGGL((peak_copy_u_kernel<4, false>), dim3(blocks), dim3(256), 0, st, s, d, n / 4); break;
    case 2: hipLaunchKernelGGL((peak_copy_u_kernel<4, true>), dim3(blocks), dim3(256), 0, st, s, d, n / 4); break;
    case 3: hipLaunchKernelGGL((peak_copy_u_kernel<8, false>), dim3(blocks), dim3(256), 0, st, s, d, n / 4); break;
    case 4: hipLaunchKernelGGL((peak_copy_u_kernel<8, true>), dim3(blocks), dim3(256), 0, st, s, d, n / 4); break;
    case 5: hipLaunchKernelGGL((peak_copy_u_kernel<2, false>), dim3(blocks), dim3(256), 0, st, s, d, n / 4); break;
    default: hipLaunchKernelGGL((peak_copy_u_kernel<1, false>), dim3(blocks), dim3(256), 0, st, s, d, n / 4); break;
  }
  EAV_CHECK_LAUNCH("eav_peak_copy_variant");
  return EAV_OK;
}

// launches `blocks` x 4 waves, each issuing 4*iters v_mfma_f32_32x32x2_f32: FLOP = blocks*4*iters*4*4096
extern "C" int eav_peak_mfma_f32(float* sink, int blocks, int iters, void* stream) {
  EAV_REQUIRE(sink && blocks > 0 && iters > 0, "eav_peak_mfma_f32: bad arguments");
  hipLaunchKernelGGL(peak_mfma_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters);
  EAV_CHECK_LAUNCH("eav_peak_mfma_f32");
  return EAV_OK;
}

// `blocks` x 4 waves, each issuing 4*iters v_mfma_f32_32x32x16_f16: FLOP = blocks*4*iters*4*32768
extern "C" int eav_peak_mfma_f16(float* sink, int blocks, int iters, void* stream) {
  EAV_REQUIRE(sink && blocks > 0 && iters > 0, "eav_peak_mfma_f16: bad arguments");
  hipLaunchKernelGGL(peak_mfma_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, iters);
  EAV_CHECK_LAUNCH("eav_peak_mfma_f16");
  return EAV_OK;
}

extern "C" int eav_peak_copy(const float* src, float* dst, int64_t n, void* stream) {
  EAV_REQUIRE(src && dst && n > 0 && (n & 3) == 0, "eav_peak_copy: bad arguments");
  // one block per CU, four non-temporal 16-byte loads in flight per lane: the best of tools/copy_bench.py's sweep
  // (6.2 TB/s read+write; thousands of blocks in flight spread the accesses over too many DRAM pages: 4.5 TB/s)
  hipLaunchKernelGGL((peak_copy_u_kernel<4, true>), dim3(256), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n / 4);
  EAV_CHECK_LAUNCH("eav_peak_copy");
  return EAV_OK;
}
