// EEGNet "separableConv": a dense 64 -> 64 channel, 16-tap temporal convolution, 'same' padding
// (7 left / 8 right) - nn.Conv2d(64, 64, (1,16), padding='same', bias=False), EEGNet_tor.py:37,59 -
// forward, data gradient and weight gradient as implicit GEMMs on v_mfma_f32_32x32x2_f32.
//
//   fwd   out[b,o,t] = sum_{i,k} wT[(i*16+k)][o] * inpad[b,i,t+k]          inpad[u] = in[u-padl]
//   dgrad is the same kernel with wT_bwd[(o*16+k')][i] = W[o,i,15-k'] and padl = 8
//   wgrad dW[o,i,k]  = sum_{b,t} du[b,o,t] * inpad[b,i,t+k]                 (padl = 7)
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

constexpr int NCH = 64, KT = 16, KD = NCH * KT;  // 1024 = contraction length of the forward GEMM
constexpr int TT = 128;                          // output samples per block
constexpr int INS = TT + 16;                     // LDS row stride of the input tile (144)
constexpr int WCH = 64;                          // contraction rows per weight chunk (4 input channels)

// ------------------------------------------------------------------------------------------ fwd
// grid (ntile, B); 4 waves: wave w owns samples [32w, 32w+32) of the tile and all 64 outputs.
__global__ __launch_bounds__(256, 2) void conv64_fwd_kernel(const float* __restrict__ in,
                                                            const float* __restrict__ wT, float* __restrict__ out,
                                                            float* __restrict__ part, int T, int padl) {
  __shared__ __attribute__((aligned(16))) float ins[NCH * INS];
  __shared__ __attribute__((aligned(16))) float wsb[2][WCH * NCH];
  __shared__ float red[4 * 128];
  const int tile = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, kk = lane >> 5;
  const int t0 = tile * TT;
  const float* src = in + (int64_t)b * NCH * T;
  // weight-chunk prefetch registers (explicit scalars: keeps them out of scratch)
  float4 rw0, rw1, rw2, rw3;
  static_assert(WCH * NCH / 4 / 256 == 4, "4 float4 per thread per weight chunk");
#define FETCH_W(ic)                                                                         \
  {                                                                                         \
    const float4* wp = reinterpret_cast<const float4*>(wT + (int64_t)(ic) * WCH * NCH) + threadIdx.x; \
    rw0 = wp[0]; rw1 = wp[256]; rw2 = wp[512]; rw3 = wp[768];                               \
  }
#define COMMIT_W(buf)                                                                       \
  {                                                                                         \
    float4* wq = reinterpret_cast<float4*>(wsb[buf]) + threadIdx.x;                         \
    wq[0] = rw0; wq[256] = rw1; wq[512] = rw2; wq[768] = rw3;                               \
  }
  FETCH_W(0);
  for (int idx = threadIdx.x; idx < NCH * INS; idx += 256) {
    const int i = idx / INS, u = idx - i * INS;
    const int t = t0 + u - padl;
    ins[idx] = (t >= 0 && t < T) ? src[(int64_t)i * T + t] : 0.f;
  }
  COMMIT_W(0);
  __syncthreads();
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  constexpr int NCHUNK = KD / WCH;                     // 16
  for (int ic = 0; ic < NCHUNK; ++ic) {
    const float* ws = wsb[ic & 1];
    if (ic + 1 < NCHUNK) FETCH_W(ic + 1);             // next chunk's weights in flight during the MFMAs
    // software pipeline: operands of step ks+1 are read while the two MFMAs of step ks run
    auto ldb = [&](int ks) { return ins[(ic * (WCH / KT) + (ks >> 3)) * INS + wave * 32 + n + 2 * (ks & 7) + kk]; };
    float bv = ldb(0), a0 = ws[kk * NCH + n], a1 = ws[kk * NCH + 32 + n];
#pragma unroll 8
    for (int ks = 0; ks < WCH / 2; ++ks) {
      const int kn = (ks + 1 < WCH / 2) ? ks + 1 : ks;
      const float bn = ldb(kn);
      const float a0n = ws[(2 * kn + kk) * NCH + n], a1n = ws[(2 * kn + kk) * NCH + 32 + n];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc1, 0, 0, 0);
      bv = bn; a0 = a0n; a1 = a1n;
    }
    if (ic + 1 < NCHUNK) COMMIT_W((ic + 1) & 1);      // the other buffer was last read in iteration ic-1
    __syncthreads();
  }
  // C layout: col = n (sample), row = (reg&3) + 8*(reg>>2) + 4*kk (output channel within the 32-tile)
  const int t = t0 + wave * 32 + n;
  const bool ok = t < T;
  float* dst = out + (int64_t)b * NCH * T + t;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = (r & 3) + 8 * (r >> 2) + 4 * kk;
    float v0 = ok ? acc0[r] : 0.f, v1 = ok ? acc1[r] : 0.f;
    if (ok) {
      dst[(int64_t)o * T] = v0;
      dst[(int64_t)(o + 32) * T] = v1;
    }
    if (part) {
      float s0 = half_sum(v0), q0 = half_sum(v0 * v0), s1 = half_sum(v1), q1 = half_sum(v1 * v1);
      if (n == 0) {
        red[wave * 128 + o] = s0;
        red[wave * 128 + 32 + o] = s1;
        red[wave * 128 + 64 + o] = q0;
        red[wave * 128 + 96 + o] = q1;
      }
    }
  }
  if (part) {
    __syncthreads();
    if (threadIdx.x < 128)
      part[((int64_t)b * gridDim.x + tile) * 128 + threadIdx.x] =
          (red[threadIdx.x] + red[128 + threadIdx.x]) + (red[256 + threadIdx.x] + red[384 + threadIdx.x]);
  }
}

// ---------------------------------------------------------------------------------------- wgrad
// grid (G, 4): blockIdx.y = group of 16 input channels, blockIdx.x = slice of the (b, tile) items.
// dW[o, i, k] = sum_{b,t} du[b,o,t] * inpad[b,i,t+k]:  A[o][t] = du, B[t][(i,k)] = inpad[i][t+k].
constexpr int DUS = TT + 1;   // 129: odd stride -> 32 rows hit 32 banks
constexpr int P2S = 144;      // == 16 (mod 32): the two channels of a column tile use disjoint banks

__global__ __launch_bounds__(256, 3) void conv64_wgrad_kernel(const float* __restrict__ du,
                                                              const float* __restrict__ in,
                                                              float* __restrict__ part, int B, int T, int padl,
                                                              int ntile) {
  __shared__ float dus[NCH * DUS];
  __shared__ float p2s[16 * P2S];
  const int ig = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, kk = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  const int nitems = B * ntile;
  const bool vec = (T & 3) == 0;
  // next item's tiles are fetched into registers before the MFMA phase of the current one
  float4 rdu[8];
  float rp[9];
  auto fetch = [&](int item) {
    const int b = item / ntile, tile = item - b * ntile;
    const int t0 = tile * TT;
    const float* dsrc = du + (int64_t)b * NCH * T;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = threadIdx.x + 256 * i;
      const int o = f >> 5, t = t0 + 4 * (f & 31);
      const float* p = dsrc + (int64_t)o * T + t;
      if (vec && t + 3 < T) rdu[i] = *reinterpret_cast<const float4*>(p);
      else rdu[i] = make_float4(t < T ? p[0] : 0.f, t + 1 < T ? p[1] : 0.f, t + 2 < T ? p[2] : 0.f, t + 3 < T ? p[3] : 0.f);
    }
    const float* isrc = in + ((int64_t)b * NCH + ig * 16) * T;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int ch = idx / P2S, u = idx - ch * P2S;
      const int t = t0 + u - padl;
      rp[i] = (idx < 16 * P2S && t >= 0 && t < T) ? isrc[(int64_t)ch * T + t] : 0.f;
    }
  };
  if ((int)blockIdx.x < nitems) fetch(blockIdx.x);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int f = threadIdx.x + 256 * i;
      float* d = dus + (f >> 5) * DUS + 4 * (f & 31);
      d[0] = rdu[i].x; d[1] = rdu[i].y; d[2] = rdu[i].z; d[3] = rdu[i].w;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int idx = threadIdx.x + 256 * i;
      if (idx < 16 * P2S) p2s[idx] = rp[i];
    }
    __syncthreads();
    if (item + (int)gridDim.x < nitems) fetch(item + gridDim.x);
    // wave w owns input channels [4w, 4w+4) of the group: two column tiles of (2 channels x 16 taps)
    const float* bp0 = p2s + (4 * wave + (n >> 4)) * P2S + (n & 15) + kk;
    const float* bp1 = bp0 + 2 * P2S;
    const float* ap0 = dus + n * DUS + kk;
    const float* ap1 = ap0 + 32 * DUS;
    float a0 = ap0[0], a1 = ap1[0], b0 = bp0[0], b1 = bp1[0];
#pragma unroll 8
    for (int ks = 0; ks < TT / 2; ++ks) {
      const int kn = (ks + 1 < TT / 2) ? ks + 1 : ks;     // prefetch the next step's operands
      const float a0n = ap0[2 * kn], a1n = ap1[2 * kn], b0n = bp0[2 * kn], b1n = bp1[2 * kn];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      a0 = a0n; a1 = a1n; b0 = b0n; b1 = b1n;
    }
  }
  // part[slice][o][i][k]; C layout: col = n -> (channel n>>4, tap n&15), row -> o
  float* dst = part + (int64_t)blockIdx.x * (NCH * KD);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        const int i = ig * 16 + 4 * wave + 2 * ct + (n >> 4);
        dst[(int64_t)o * KD + i * KT + (n & 15)] = acc[mt][ct][r];
      }
}

// wT_fwd[(i*16+k)][o] = W[o][i][k];  wT_bwd[(o*16+k')][i] = W[o][i][15-k']
__global__ void conv64_prep_kernel(const float* __restrict__ w, float* __restrict__ wT_fwd,
                                   float* __restrict__ wT_bwd) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= NCH * KD) return;
  const int o = idx / KD, rem = idx - o * KD, i = rem >> 4, k = rem & 15;
  const float v = w[idx];
  wT_fwd[(i * KT + k) * NCH + o] = v;
  wT_bwd[(o * KT + (15 - k)) * NCH + i] = v;
}

}  // namespace

extern "C" int eav_conv64_prep_weights(const float* w, float* wT_fwd, float* wT_bwd, void* stream) {
  EAV_REQUIRE(w && wT_fwd && wT_bwd, "eav_conv64_prep_weights: bad arguments");
  hipLaunchKernelGGL(conv64_prep_kernel, dim3(NCH * KD / 256), dim3(256), 0, (hipStream_t)stream, w, wT_fwd, wT_bwd);
  EAV_CHECK_LAUNCH("eav_conv64_prep_weights");
  return EAV_OK;
}

extern "C" int eav_conv64_ntiles(int T) { return cdiv(T, TT); }

extern "C" int eav_conv64_fwd(const float* in, const float* wT, float* out, float* stat_part, int B, int T, int padl,
                              void* stream) {
  EAV_REQUIRE(in && wT && out && B > 0 && T > 0 && padl >= 0 && padl <= 15, "eav_conv64_fwd: bad arguments");
  dim3 grid(cdiv(T, TT), B);
  hipLaunchKernelGGL(conv64_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, wT, out, stat_part, T, padl);
  EAV_CHECK_LAUNCH("eav_conv64_fwd");
  return EAV_OK;
}

extern "C" int eav_conv64_wgrad_nparts(int B, int T) {
  int nitems = B * cdiv(T, TT);
  return nitems < 192 ? nitems : 192;
}

extern "C" int eav_conv64_wgrad(const float* du, const float* in, float* part, int B, int T, int padl,
                                void* stream) {
  EAV_REQUIRE(du && in && part && B > 0 && T > 0 && padl >= 0 && padl <= 15, "eav_conv64_wgrad: bad arguments");
  dim3 grid(eav_conv64_wgrad_nparts(B, T), 4);
  hipLaunchKernelGGL(conv64_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, du, in, part, B, T, padl,
                     cdiv(T, TT));
  EAV_CHECK_LAUNCH("eav_conv64_wgrad");
  return EAV_OK;
}
