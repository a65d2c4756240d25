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

// ---------------------------------------------------------------------------------------- wgrad
// dW[o,i,k] = sum_{b,t} du[b,o,t] * in[b,i,t+k-padl]:  per tap a 64x64 product contracted over time.  grid (G, 4):
// blockIdx.y = (row tile rt, column tile ct) of the 64x64 output; the block's 4 waves own 4 taps each and share the LDS
// images of du[32 rows][128] and in[32 channels][128+16] (channel-major, time-minor - the contraction now runs over
// time).  A lane's eight contraction rows are eight consecutive time samples starting at 16 ks + 8 g + k: the two
// aligned 16-byte blocks around them are read once per K-step and the four taps' operands are cut out in registers
// (dword select for even k, v_alignbit for odd k) - the shift is a compile-time constant per (tap quad parity, tap).
constexpr int WT = 128;             // time samples per item
constexpr int DRS = WT + 8;         // halfs per du row (272 B)
constexpr int IRS = WT + 24;        // halfs per in row (304 B): 128 + 15 taps + alignment slack

template <int HALF>                 // HALF = tap quad parity: shifts 4*HALF + tk
__device__ __forceinline__ void wgrad_steps(const _Float16* __restrict__ dh, const _Float16* __restrict__ dl,
                                            const _Float16* __restrict__ ih, const _Float16* __restrict__ il,
                                            f32x16 (&c1)[4], f32x16 (&c2)[4]) {
#pragma unroll 2
  for (int ks = 0; ks < WT / 16; ++ks) {
    const h8 ahi = *reinterpret_cast<const h8*>(dh + 16 * ks);
    const h8 alo = *reinterpret_cast<const h8*>(dl + 16 * ks);
    uint32_t H[8], L[8];
    {
      const uint4 a = *reinterpret_cast<const uint4*>(ih + 16 * ks), b = *reinterpret_cast<const uint4*>(ih + 16 * ks + 8);
      H[0] = a.x; H[1] = a.y; H[2] = a.z; H[3] = a.w; H[4] = b.x; H[5] = b.y; H[6] = b.z; H[7] = b.w;
      const uint4 c = *reinterpret_cast<const uint4*>(il + 16 * ks), d = *reinterpret_cast<const uint4*>(il + 16 * ks + 8);
      L[0] = c.x; L[1] = c.y; L[2] = c.z; L[3] = c.w; L[4] = d.x; L[5] = d.y; L[6] = d.z; L[7] = d.w;
    }
#pragma unroll
    for (int tk = 0; tk < 4; ++tk) {
      const int sh = 4 * HALF + tk, d0 = sh >> 1;
      union { uint32_t u[4]; h8 h; } bh, bl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (sh & 1) {
          bh.u[j] = __builtin_amdgcn_alignbit(H[d0 + j + 1], H[d0 + j], 16);
          bl.u[j] = __builtin_amdgcn_alignbit(L[d0 + j + 1], L[d0 + j], 16);
        } else {
          bh.u[j] = H[d0 + j];
          bl.u[j] = L[d0 + j];
        }
      }
      c2[tk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bh.h, c2[tk], 0, 0, 0);
      c1[tk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bh.h, c1[tk], 0, 0, 0);
      c2[tk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bl.h, c2[tk], 0, 0, 0);
    }
  }
}

__global__ __launch_bounds__(256, 2) void conv64_wgrad_split_kernel(const float* __restrict__ du,
                                                                    const float* __restrict__ in,
                                                                    const float* __restrict__ sdu,
                                                                    const float* __restrict__ sin_,
                                                                    float* __restrict__ part, int B, int T, int padl,
                                                                    int ntile) {
  __shared__ __attribute__((aligned(16))) _Float16 duh[32 * DRS];
  __shared__ __attribute__((aligned(16))) _Float16 dul[32 * DRS];
  __shared__ __attribute__((aligned(16))) _Float16 inh[32 * IRS];
  __shared__ __attribute__((aligned(16))) _Float16 inl[32 * IRS];
  const int rt = blockIdx.y >> 1, ct = blockIdx.y & 1;
  const int lane = threadIdx.x & 63, tq = threadIdx.x >> 6;
  const int n = lane & 31, g = lane >> 5;
  const float sigd = sdu[0], sigi = sin_[0], post = sdu[1] * sin_[1];
  f32x16 c1[4], c2[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) { c1[a][r] = 0.f; c2[a][r] = 0.f; }
  const int nitems = B * ntile;
  constexpr int NDU = 32 * (WT / 2) / 256;                 // sample pairs of the du tile per thread: 8
  constexpr int NIN = (32 * ((WT + 16) / 2) + 255) / 256;  // sample pairs of the in tile per thread: 9
  float2 rdu[NDU], rin[NIN];
  auto fetch = [&](int item) {
    const int b = item / ntile, tile = item - b * ntile;
    const int t0 = tile * WT;
    const float* dsrc = du + ((int64_t)b * NCH + 32 * rt) * T;
#pragma unroll
    for (int i = 0; i < NDU; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int o = idx / (WT / 2), t = t0 + 2 * (idx - o * (WT / 2));
      const float* p = dsrc + (int64_t)o * T + t;
      rdu[i].x = t < T ? p[0] : 0.f;
      rdu[i].y = t + 1 < T ? p[1] : 0.f;
    }
    const float* isrc = in + ((int64_t)b * NCH + 32 * ct) * T;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int ch = idx / ((WT + 16) / 2), u = 2 * (idx - ch * ((WT + 16) / 2));
      const int t = t0 + u - padl;
      const bool ok = ch < 32;
      rin[i].x = (ok && t >= 0 && t < T) ? isrc[(int64_t)ch * T + t] : 0.f;
      rin[i].y = (ok && t + 1 >= 0 && t + 1 < T) ? isrc[(int64_t)ch * T + t + 1] : 0.f;
    }
  };
  if ((int)blockIdx.x < nitems) fetch(blockIdx.x);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NDU; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int o = idx / (WT / 2), tl = 2 * (idx - o * (WT / 2));
      _Float16 h0, l0, h1, l1;
      split2(sigd * rdu[i].x, h0, l0);
      split2(sigd * rdu[i].y, h1, l1);
      *reinterpret_cast<uint32_t*>(&duh[o * DRS + tl]) = pack2(h0, h1);
      *reinterpret_cast<uint32_t*>(&dul[o * DRS + tl]) = pack2(l0, l1);
    }
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int ch = idx / ((WT + 16) / 2), u = 2 * (idx - ch * ((WT + 16) / 2));
      if (ch < 32) {
        _Float16 h0, l0, h1, l1;
        split2(sigi * rin[i].x, h0, l0);
        split2(sigi * rin[i].y, h1, l1);
        *reinterpret_cast<uint32_t*>(&inh[ch * IRS + u]) = pack2(h0, h1);
        *reinterpret_cast<uint32_t*>(&inl[ch * IRS + u]) = pack2(l0, l1);
      }
    }
    __syncthreads();
    if (item + (int)gridDim.x < nitems) fetch(item + gridDim.x);
    // taps 4 tq .. 4 tq + 3: aligned block 8 * (2 ks + g + (tq >> 1)), shift 4 (tq & 1) + tk inside the block pair
    const _Float16* dh = duh + n * DRS + 8 * g;
    const _Float16* dl = dul + n * DRS + 8 * g;
    const _Float16* ih = inh + n * IRS + 8 * (g + (tq >> 1));
    const _Float16* il = inl + n * IRS + 8 * (g + (tq >> 1));
    if (tq & 1) wgrad_steps<1>(dh, dl, ih, il, c1, c2);
    else wgrad_steps<0>(dh, dl, ih, il, c1, c2);
  }
  // part[slice][o][i][k]; C layout: col = n -> input channel 32 ct + n, row -> output channel
  float* dst = part + (int64_t)blockIdx.x * (NCH * NCH * KT);
#pragma unroll
  for (int tk = 0; tk < 4; ++tk)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * g;
      dst[(int64_t)o * (NCH * KT) + (32 * ct + n) * KT + 4 * tq + tk] = post * fmaf(LO_INV, c2[tk][r], c1[tk][r]);
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


// part [eav_conv64_wgrad_nparts(B, T)][64*64*16]; sum over parts = dL/dW[o,i,k].  scale_du / scale_in: device float[3].
extern "C" int eav_conv64_wgrad_split(const float* du, const float* in, const float* scale_du, const float* scale_in,
                                      float* part, int B, int T, int padl, void* stream) {
  EAV_REQUIRE(du && in && scale_du && scale_in && part && B > 0 && T > 0 && padl >= 0 && padl <= 15,
              "eav_conv64_wgrad_split: bad arguments");
  dim3 grid(eav_conv64_wgrad_nparts(B, T), 4);
  hipLaunchKernelGGL(conv64_wgrad_split_kernel, grid, dim3(256), 0, (hipStream_t)stream, du, in, scale_du, scale_in,
                     part, B, T, padl, cdiv(T, WT));
  EAV_CHECK_LAUNCH("eav_conv64_wgrad_split");
  return EAV_OK;
}
