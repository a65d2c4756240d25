// v_permlane32_swap / v_permlane16_swap (gfx950): which lanes move, and what a swap costs beside packed fp32 adds.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/build/permlane_probe tools/probes/permlane_probe.hip ; run on the GPU box
// Expectation used by eegnet_fir_fft.hip (quad transpose of the third FFT stage): swap32(a, b) exchanges a[lanes 32-63]
// with b[lanes 0-31]; swap16(a, b) exchanges a[lanes 16-31, 48-63] with b[lanes 0-15, 32-47].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void swap32(float& a, float& b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16(float& a, float& b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}

__global__ void semantics(float* out) {
  const int l = threadIdx.x;
  float a = (float)l, b = 100.f + l;
  swap32(a, b);
  out[l] = a; out[64 + l] = b;
  a = (float)l; b = 100.f + l;
  swap16(a, b);
  out[128 + l] = a; out[192 + l] = b;
  // 4 x 4 transpose between lane bits 5:4 and the register index
  float r[4];
  for (int i = 0; i < 4; ++i) r[i] = 1000.f * i + l;      // reg i of lane l
  swap32(r[0], r[2]); swap32(r[1], r[3]);
  swap16(r[0], r[1]); swap16(r[2], r[3]);
  for (int i = 0; i < 4; ++i) out[256 + 64 * i + l] = r[i];
}

__device__ __forceinline__ void swap32v(v2f& a, v2f& b) {
  float ax = a.x, ay = a.y, bx = b.x, by = b.y;
  swap32(ax, bx); swap32(ay, by);
  a = (v2f){ax, ay}; b = (v2f){bx, by};
}
__device__ __forceinline__ void swap16v(v2f& a, v2f& b) {
  float ax = a.x, ay = a.y, bx = b.x, by = b.y;
  swap16(ax, bx); swap16(ay, by);
  a = (v2f){ax, ay}; b = (v2f){bx, by};
}

template <int KIND>
__global__ __launch_bounds__(512, 1) void rate(float* out, long long* cyc, int iters) {
  v2f v[16];
  for (int i = 0; i < 16; ++i) v[i] = (v2f){threadIdx.x * 0.01f + i, 0.5f * i};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {      // 32 swaps (the quad transposes of one transform)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        swap32v(v[4 * m], v[4 * m + 2]); swap32v(v[4 * m + 1], v[4 * m + 3]);
        swap16v(v[4 * m], v[4 * m + 1]); swap16v(v[4 * m + 2], v[4 * m + 3]);
      }
    } else {              // 32 packed adds
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] += v[(i + 5) & 15];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] -= v[(i + 3) & 15] * 0.5f;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 4096 * 8);
  hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, out);
  std::vector<float> h(512);
  hipMemcpy(h.data(), out, 512 * 4, hipMemcpyDeviceToHost);
  bool ok32 = true, ok16 = true, okT = true;
  for (int l = 0; l < 64; ++l) {
    const float ea = l >= 32 ? 100.f + (l - 32) : (float)l, eb = l < 32 ? (float)(l + 32) : 100.f + l;
    ok32 &= h[l] == ea && h[64 + l] == eb;
    const bool odd = (l >> 4) & 1;
    const float fa = odd ? 100.f + (l - 16) : (float)l, fb = odd ? 100.f + l : (float)(l + 16);
    ok16 &= h[128 + l] == fa && h[192 + l] == fb;
    for (int i = 0; i < 4; ++i) okT &= h[256 + 64 * i + l] == 1000.f * (l >> 4) + (16 * i + (l & 15));
  }
  printf("swap32 as expected: %d   swap16 as expected: %d   4x4 transpose (reg i of lane 16 a + k = old reg a of lane 16 i + k): %d\n",
         ok32, ok16, okT);
  if (!ok32 || !ok16 || !okT) {
    for (int r = 0; r < 8; ++r) { for (int l = 0; l < 64; ++l) printf("%g ", h[64 * r + l]); printf("\n"); }
  }
  const int iters = 2000;
  for (int kind = 0; kind < 2; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
      else hipLaunchKernelGGL(rate<1>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
      hipDeviceSynchronize();
    }
    std::vector<long long> c(256);
    hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : c) m += x; m /= 256;
    printf("%s: %.1f shader-clock ticks per iteration (32 instructions per wave, 2 waves per SIMD)\n",
           kind == 0 ? "32 permlane swaps" : "32 v_pk ops      ", m / iters);
  }
  return 0;
}
