// EEGNet "separableConv" (dense 64 -> 64 channels, 16 taps, EEGNet_tor.py:37,59) on the fp16 matrix cores with
// split operands - the opt-in fp32-grade fast path that goes with eegnet_fir_split.hip (same representation:
// sigma v = hi + 2^-11 lo, three MFMAs per product, fp32 accumulation; see that file's header).
//
// Formulation.  The contraction index of the exact-fp32 kernel is (input channel, tap), which makes a lane's eight
// consecutive contraction rows eight consecutive TIME samples starting at an arbitrary offset - unaligned for 16-byte
// LDS reads of fp16.  Here the taps are taken apart instead:  out[o,t] = sum_k ( W_k . in[:, t+k] )  with W_k a 64x64
// matrix, so the contraction runs over CHANNELS only and the input tile is kept time-major / channel-minor in LDS
// ([u][64 ch], row stride 144 B): a lane's eight contraction rows are eight consecutive channels of one time row -
// always 16-byte aligned, and the 144-byte stride spreads 16 consecutive rows over all 64 banks.
//   fwd / dgrad : wave (ot, tg) = (32-channel output tile, group of 4 taps) keeps its 4 x 4 K-step weight pieces in
//                 128 VGPRs for the whole launch; 48 MFMAs per 32-sample sub-tile (exact kernel: 128 at twice the
//                 cycles each); the four tap-group partials of an output tile are summed through LDS in fixed order.
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int NCH = 64, KT = 16;
constexpr int TT = 128;            // output samples per item
constexpr int ROWS = TT + 16;      // time rows of the LDS tile (t0 - padl ... t0 + 127 + 15 - padl)
constexpr int RS = 72;             // halfs per row: 64 channels + 8 pad = 144 B
constexpr int THREADS = 512;
constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;

__device__ __forceinline__ void split2(float v, _Float16& hi, _Float16& lo) {
  v = __builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
  hi = (_Float16)v;
  lo = (_Float16)((v - (float)hi) * LO_SCALE);
}
__device__ __forceinline__ uint32_t pack2(_Float16 a, _Float16 b) {
  union { _Float16 h[2]; uint32_t u; } v;
  v.h[0] = a; v.h[1] = b;
  return v.u;
}

constexpr int NLD = (32 * ROWS + THREADS - 1) / THREADS;     // (channel pair, time row) entries per thread: 9

__global__ __launch_bounds__(THREADS, 1) void conv64_fwd_split_kernel(const float* __restrict__ in,
                                                                      const float* __restrict__ wT,
                                                                      const float* __restrict__ sx,
                                                                      const float* __restrict__ sw,
                                                                      float* __restrict__ out, float* __restrict__ part,
                                                                      int B, int T, int padl, int ntile) {
  __shared__ __attribute__((aligned(16))) _Float16 xh[ROWS * RS];
  __shared__ __attribute__((aligned(16))) _Float16 xl[ROWS * RS];
  __shared__ float red[2][8][16 * 64];                               // [buffer][wave][reg*64 + lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, g = lane >> 5;
  const int ot = wave & 1, tg = wave >> 1;
  const float sigx = sx[0], sigw = sw[0], post = sx[1] * sw[1];
  // weight pieces of taps 4 tg .. 4 tg + 3: row o = 32 ot + n, contraction rows = channels 16 ks + 8 g + e
  h8 ah[4][4], al[4][4];
#pragma unroll
  for (int tk = 0; tk < 4; ++tk)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = 16 * ks + 8 * g + e, k = 4 * tg + tk;
        _Float16 hi, lo;
        split2(sigw * wT[(int64_t)(i * KT + k) * NCH + 32 * ot + n], hi, lo);
        ah[tk][ks][e] = hi;
        al[tk][ks][e] = lo;
      }
  float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};
  float2 rin[NLD];
  const int nitems = B * ntile;
  auto fetch = [&](int item) {
    const int b = item / ntile, tile = item - b * ntile;
    const int t0 = tile * TT;
    const float* src = in + (int64_t)b * NCH * T;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = threadIdx.x + THREADS * i;
      const int chp = idx / ROWS, u = idx - chp * ROWS;
      const int t = t0 + u - padl;
      const bool ok = idx < 32 * ROWS && t >= 0 && t < T;
      rin[i].x = ok ? src[(int64_t)(2 * chp) * T + t] : 0.f;
      rin[i].y = ok ? src[(int64_t)(2 * chp + 1) * T + t] : 0.f;
    }
  };
  if ((int)blockIdx.x < nitems) fetch(blockIdx.x);
  int rb = 0;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / ntile, tile = item - b * ntile;
    const int t0 = tile * TT;
    __syncthreads();                       // every wave is done with the previous input tile
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = threadIdx.x + THREADS * i;
      if (idx < 32 * ROWS) {
        const int chp = idx / ROWS, u = idx - chp * ROWS;
        _Float16 h0, l0, h1, l1;
        split2(sigx * rin[i].x, h0, l0);
        split2(sigx * rin[i].y, h1, l1);
        *reinterpret_cast<uint32_t*>(&xh[u * RS + 2 * chp]) = pack2(h0, h1);
        *reinterpret_cast<uint32_t*>(&xl[u * RS + 2 * chp]) = pack2(l0, l1);
      }
    }
    __syncthreads();
    if (item + (int)gridDim.x < nitems) fetch(item + gridDim.x);
    for (int sub = 0; sub < TT / 32; ++sub) {
      f32x16 c1, c2;
#pragma unroll
      for (int r = 0; r < 16; ++r) { c1[r] = 0.f; c2[r] = 0.f; }
      const int base = (32 * sub + n + 4 * tg) * RS + 8 * g;
#pragma unroll
      for (int tk = 0; tk < 4; ++tk) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const h8 bh = *reinterpret_cast<const h8*>(&xh[base + tk * RS + 16 * ks]);
          const h8 bl = *reinterpret_cast<const h8*>(&xl[base + tk * RS + 16 * ks]);
          c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[tk][ks], bh, c2, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tk][ks], bh, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[tk][ks], bl, c2, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the operand reads of later taps from being hoisted (registers)
      }
      float* rw = red[rb][wave];
#pragma unroll
      for (int r = 0; r < 16; ++r) rw[r * 64 + lane] = post * fmaf(LO_INV, c2[r], c1[r]);
      __syncthreads();
      // wave (ot, tg) finalises registers r in [4 tg, 4 tg + 4) of output tile ot: sum of the 4 tap-group partials
      const int t = t0 + 32 * sub + n;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 4 * tg + q;
        float v = (red[rb][ot][r * 64 + lane] + red[rb][ot + 2][r * 64 + lane]) +
                  (red[rb][ot + 4][r * 64 + lane] + red[rb][ot + 6][r * 64 + lane]);
        const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * g;
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
        const int r = 4 * tg + q;
        const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * g;
        part[(int64_t)blockIdx.x * 128 + o] = s1;
        part[(int64_t)blockIdx.x * 128 + 64 + o] = s2;
      }
    }
  }
}

}  // namespace

// Same contract as eav_conv64_fwd (statistics partials: eav_conv64_fwd_nparts(B, T) rows of 128) plus the operand
// scales: scale_x / scale_w = device float[3] from eav_absmax_scale over `in` and over the 65 536 weights.
extern "C" int eav_conv64_fwd_split(const float* in, const float* wT, const float* scale_x, const float* scale_w,
                                    float* out, float* stat_part, int B, int T, int padl, void* stream) {
  EAV_REQUIRE(in && wT && scale_x && scale_w && out && B > 0 && T > 0 && padl >= 0 && padl <= 15,
              "eav_conv64_fwd_split: bad arguments");
  hipLaunchKernelGGL(conv64_fwd_split_kernel, dim3(eav_conv64_fwd_nparts(B, T)), dim3(THREADS), 0, (hipStream_t)stream,
                     in, wT, scale_x, scale_w, out, stat_part, B, T, padl, cdiv(T, TT));
  EAV_CHECK_LAUNCH("eav_conv64_fwd_split");
  return EAV_OK;
}
