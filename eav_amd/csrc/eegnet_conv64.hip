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

// ------------------------------------------------------------------------------------------ fwd
// Weight-stationary schedule: 8 waves per block, wave (ot, ic) keeps the weights of output tile ot (32 channels)
// x input-channel chunk ic (16 channels = 256 contraction rows) in 128 VGPRs for the whole kernel - no weight
// staging, no per-chunk barriers.  Per 32-sample sub-tile a wave issues 128 MFMAs whose B operands come from the
// LDS input tile; the 4 partial tiles of an output tile (one per input chunk) are summed through LDS in a fixed
// order.  Blocks are persistent over (image, 128-sample tile) items with the next input tile prefetched into
// registers; BatchNorm statistics accumulate in registers and are written once per block.
constexpr int V2_THREADS = 512;

// TT_ = output samples per work item: 128 normally; 32 when the launch would otherwise have fewer items than CUs
// (the reference's own [32,1,30,500] batches: T = 125 -> one 128-sample item per image = 32 blocks on 256 CUs).
template <int TT_>
__global__ __launch_bounds__(V2_THREADS, 1) void conv64_fwd_kernel(const float* __restrict__ in,
                                                                      const float* __restrict__ wT,
                                                                      float* __restrict__ out, float* __restrict__ part,
                                                                      int B, int T, int padl, int ntile) {
  constexpr int INS_ = TT_ + 16;                                      // LDS row stride of the input tile
  constexpr int V2_NLD = (NCH * INS_ + V2_THREADS - 1) / V2_THREADS;  // floats per thread per input tile
  __shared__ __attribute__((aligned(16))) float ins[NCH * INS_];
  __shared__ float red[2][8][16 * 64];                               // [buffer][wave][reg*64 + lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, kk = lane >> 5;
  const int ot = wave & 1, ic = wave >> 1;
  float areg[128];
#pragma unroll
  for (int p = 0; p < 128; ++p) areg[p] = wT[(int64_t)(ic * 256 + 2 * p + kk) * NCH + 32 * ot + n];
  float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};
  float rin[V2_NLD];
  const int nitems = B * ntile;
  auto fetch = [&](int item) {
    const int b = item / ntile, tile = item - b * ntile;
    const int t0 = tile * TT_;
    const float* src = in + (int64_t)b * NCH * T;
#pragma unroll
    for (int i = 0; i < V2_NLD; ++i) {
      const int idx = threadIdx.x + V2_THREADS * i;
      const int ch = idx / INS_, u = idx - ch * INS_;
      const int t = t0 + u - padl;
      rin[i] = (idx < NCH * INS_ && t >= 0 && t < T) ? src[(int64_t)ch * T + t] : 0.f;
    }
  };
  if ((int)blockIdx.x < nitems) fetch(blockIdx.x);
  int rb = 0;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / ntile, tile = item - b * ntile;
    const int t0 = tile * TT_;
    __syncthreads();                       // every wave is done with the previous input tile
#pragma unroll
    for (int i = 0; i < V2_NLD; ++i) {
      const int idx = threadIdx.x + V2_THREADS * i;
      if (idx < NCH * INS_) ins[idx] = rin[i];
    }
    __syncthreads();
    if (item + (int)gridDim.x < nitems) fetch(item + gridDim.x);
    for (int sub = 0; sub < TT_ / 32; ++sub) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* bp = ins + (16 * ic) * INS_ + 32 * sub + n + kk;
#pragma unroll
      for (int p = 0; p < 128; ++p)        // contraction row 2p+kk of the chunk: channel p>>3, tap 2(p&7)+kk
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[p], bp[(p >> 3) * INS_ + 2 * (p & 7)], acc, 0, 0, 0);
      float* rw = red[rb][wave];
#pragma unroll
      for (int r = 0; r < 16; ++r) rw[r * 64 + lane] = acc[r];
      __syncthreads();
      // wave (ot, ic) finalises registers r in [4 ic, 4 ic + 4) of output tile ot: sum of the 4 chunk partials
      const int t = t0 + 32 * sub + n;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * ic + q;
        float v = (red[rb][ot][r * 64 + lane] + red[rb][ot + 2][r * 64 + lane]) +
                  (red[rb][ot + 4][r * 64 + lane] + red[rb][ot + 6][r * 64 + lane]);
        const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (t < T) out[((int64_t)b * NCH + o) * T + t] = v; else v = 0.f;
        st_s[q] += v;
        st_q[q] += v * v;
      }
      rb ^= 1;                             // the other buffer was last read two barriers ago
    }
  }
  if (part) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float s1 = half_sum(st_s[q]), s2 = half_sum(st_q[q]);
      if (n == 0) {
        const int r = 4 * ic + q;
        const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * kk;
        part[(int64_t)blockIdx.x * 128 + o] = s1;
        part[(int64_t)blockIdx.x * 128 + 64 + o] = s2;
      }
    }
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
// c0..c3 (optional): step counters incremented by one thread of the launch (eav_eegnet_step_prologue)
__global__ void conv64_prep_kernel(const float* __restrict__ w, float* __restrict__ wT_fwd,
                                   float* __restrict__ wT_bwd, int64_t* c0, int64_t* c1, int64_t* c2, int64_t* c3) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx == 0) {
    if (c0) *c0 += 1;
    if (c1) *c1 += 1;
    if (c2) *c2 += 1;
    if (c3) *c3 += 1;
  }
  if (idx >= NCH * KD) return;
  const int o = idx / KD, rem = idx - o * KD, i = rem >> 4, k = rem & 15;
  const float v = w[idx];
  wT_fwd[(i * KT + k) * NCH + o] = v;
  wT_bwd[(o * KT + (15 - k)) * NCH + i] = v;
}

}  // namespace

extern "C" int eav_conv64_prep_weights(const float* w, float* wT_fwd, float* wT_bwd, void* stream) {
  EAV_REQUIRE(w && wT_fwd && wT_bwd, "eav_conv64_prep_weights: bad arguments");
  return eav_eegnet_step_prologue(w, wT_fwd, wT_bwd, nullptr, nullptr, nullptr, nullptr, stream);
}

extern "C" int eav_eegnet_step_prologue(const float* w, float* wT_fwd, float* wT_bwd, int64_t* c0, int64_t* c1,
                                        int64_t* c2, int64_t* c3, void* stream) {
  EAV_REQUIRE(w && wT_fwd && wT_bwd, "eav_eegnet_step_prologue: bad arguments");
  hipLaunchKernelGGL(conv64_prep_kernel, dim3(NCH * KD / 256), dim3(256), 0, (hipStream_t)stream, w, wT_fwd, wT_bwd,
                     c0, c1, c2, c3);
  EAV_CHECK_LAUNCH("eav_eegnet_step_prologue");
  return EAV_OK;
}

extern "C" int eav_conv64_ntiles(int T) { return cdiv(T, TT); }

// few 128-sample items -> 32-sample items (4x the blocks; every block still keeps its weight slabs in registers)
static bool conv64_small(int B, int T) { return B * cdiv(T, TT) < 128; }

// number of statistics partials eav_conv64_fwd writes ([nparts][128])
extern "C" int eav_conv64_fwd_nparts(int B, int T) {
  const int nitems = conv64_small(B, T) ? B * cdiv(T, 32) : B * cdiv(T, TT);
  return nitems < 256 ? nitems : 256;
}

extern "C" int eav_conv64_fwd(const float* in, const float* wT, float* out, float* stat_part, int B, int T, int padl,
                              void* stream) {
  EAV_REQUIRE(in && wT && out && B > 0 && T > 0 && padl >= 0 && padl <= 15, "eav_conv64_fwd: bad arguments");
  if (conv64_small(B, T))
    hipLaunchKernelGGL(conv64_fwd_kernel<32>, dim3(eav_conv64_fwd_nparts(B, T)), dim3(V2_THREADS), 0,
                       (hipStream_t)stream, in, wT, out, stat_part, B, T, padl, cdiv(T, 32));
  else
    hipLaunchKernelGGL(conv64_fwd_kernel<TT>, dim3(eav_conv64_fwd_nparts(B, T)), dim3(V2_THREADS), 0,
                       (hipStream_t)stream, in, wT, out, stat_part, B, T, padl, cdiv(T, TT));
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
