// EEGNet element-wise / reduction kernels around the three convolutions (HBM-bound, fp32).
//
//  dw_fwd   : firstBN -> ELU -> depthwiseConv (30 channels -> D=8 per filter)   EEGNet_tor.py:52-54
//  pool_fwd : BN -> ELU -> AvgPool(1,P) -> Dropout                               :55-58 / :60-63
//  pool_bwd : backward of pool_fwd (two passes: BN-backward sums, then apply)
//  dw_bwd   : backward of dw_fwd: depthwise weight grad, g = dL/d(firstBN out), BN-backward sums
// All tensors are channels-first, time innermost; lanes run along time, 16 B per lane where the
// row length allows (row bases are then 16-B aligned), scalar fallback otherwise.
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

constexpr int F1 = 8, DD = 8, CHMAX = 32;

__device__ __forceinline__ void ld4(const float* p, int i, int n, bool vec, float (&v)[4]) {
  if (vec && i + 3 < n) {
    float4 a = *reinterpret_cast<const float4*>(p + i);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (i + e < n) ? p[i + e] : 0.f;
  }
}
// streaming forms: the big activation tensors are touched once per kernel - non-temporal loads / stores keep them from
// evicting the small reused operands (measured on the float4 copy: +10 % at the same geometry)
__device__ __forceinline__ void ld4s(const float* p, int i, int n, bool vec, float (&v)[4]) {
  if (vec && i + 3 < n) {
    const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + i));
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (i + e < n) ? p[i + e] : 0.f;
  }
}
__device__ __forceinline__ void st4s(float* p, int i, int n, bool vec, const float (&v)[4]) {
  if (vec && i + 3 < n) {
    __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(p + i));
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i + e < n) p[i + e] = v[e];
  }
}
__device__ __forceinline__ void st4(float* p, int i, int n, bool vec, const float (&v)[4]) {
  if (vec && i + 3 < n) {
    *reinterpret_cast<float4*>(p + i) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i + e < n) p[i + e] = v[e];
  }
}

// block_sum_256 for NT-thread blocks (NT / 64 waves; the wave sums are added in wave order)
template <int NT, int NV>
__device__ __forceinline__ void block_sum_nt(float (&v)[NV], float* red) {
  if constexpr (NT == 256) {
    block_sum_256<NV>(v, red);
  } else {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const float s = wave_sum(v[k]);
      if (lane == 0) red[wave * NV + k] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < NV) {
      float t = red[threadIdx.x];
#pragma unroll
      for (int w = 1; w < NT / 64; ++w) t += red[w * NV + threadIdx.x];
      v[0] = t;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------- dw_fwd
// grid (nchunk, F1, B); block 256 threads x 4 samples = 1024 samples of one (b, f).
// z[b, f*8+d, t] = sum_c w2[f*8+d, c] * ELU(scale1[f]*y1[b,f,c,t] + shift1[f])
// part[(b*nchunk+chunk)][2*64]: per-channel sum z / sum z^2 (slots of this block's 8 channels).
// CG = channel groups, NT = threads: for short rows (S <= 4 NT / CG, one chunk - the reference's own S = 500) the block is
// CG groups of NT / CG threads, group j walks channels j, j + CG, ... and the groups' partial sums meet in LDS (fixed
// order): every wave works instead of half of them idling past the row's end, and the serial channel loop - a chain of
// exposed instruction latencies when a CU holds one block - is CG times shorter.
template <int CG, int NT>
__global__ __launch_bounds__(NT) void dw_fwd_kernel(const float* __restrict__ y1, const float* __restrict__ bn1,
                                                     const float* __restrict__ w2, float* __restrict__ z,
                                                     float* __restrict__ part, int C, int S,
                                                     const float* __restrict__ bn2, float* __restrict__ p2) {
  __shared__ float wsh[DD * CHMAX];
  __shared__ float red[NT / 64 * 16];
  __shared__ float xsum[CG > 1 ? (CG - 1) * DD * 4 * (NT / CG) : 1];
  constexpr int TPG = NT / CG;            // threads per channel group
  const int chunk = blockIdx.x, f = blockIdx.y, b = blockIdx.z;
  const int cg = CG > 1 ? threadIdx.x / TPG : 0, tl = CG > 1 ? threadIdx.x % TPG : threadIdx.x;
  for (int i = threadIdx.x; i < DD * C; i += NT) wsh[i] = w2[f * DD * C + i];
  __syncthreads();
  const float sc = bn1[16 + f], sh = bn1[24 + f];  // layout: mean, invstd, scale, shift (8 each)
  const bool vec = (S & 3) == 0;
  const int t = chunk * (4 * TPG) + tl * 4;
  float acc[DD][4];
#pragma unroll
  for (int d = 0; d < DD; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[d][e] = 0.f;
  if (t < S) {
    const float* src = y1 + ((int64_t)b * F1 + f) * C * S;
    float nx[2][4];                       // two channel rows in flight ahead of the one being consumed
    ld4s(src + (int64_t)(CG > 1 ? min(cg, C - 1) : 0) * S, t, S, vec, nx[0]);
    ld4s(src + (int64_t)min(cg + CG, C - 1) * S, t, S, vec, nx[1]);
    for (int c = cg; c < C; c += CG) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = nx[0][e]; nx[0][e] = nx[1][e]; }
      ld4s(src + (int64_t)min(c + 2 * CG, C - 1) * S, t, S, vec, nx[1]);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = elu_f(sc * v[e] + sh);
#pragma unroll
      for (int d = 0; d < DD; ++d) {
        const float w = wsh[d * C + c];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[d][e] += w * v[e];
      }
    }
  }
  if (CG > 1) {                           // group 0 collects the other groups' partial sums, in group order
    if (cg > 0)
#pragma unroll
      for (int d = 0; d < DD; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) xsum[(((cg - 1) * DD + d) * 4 + e) * TPG + tl] = acc[d][e];
    __syncthreads();
    if (cg == 0)
#pragma unroll
      for (int g = 1; g < CG; ++g)
#pragma unroll
        for (int d = 0; d < DD; ++d)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[d][e] += xsum[(((g - 1) * DD + d) * 4 + e) * TPG + tl];
  }
  const bool mine = CG == 1 || cg == 0;   // the group that holds the sums writes z and the statistics
  float st[16];
#pragma unroll
  for (int d = 0; d < DD; ++d) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = (mine && t + e < S) ? acc[d][e] : 0.f;
      s += v;
      q += v * v;
    }
    st[d] = s;
    st[8 + d] = q;
    if (mine && t < S) st4s(z + ((int64_t)b * F1 * DD + f * DD + d) * S, t, S, vec, acc[d]);
    // eval-mode forward (bn2 = depthwiseBN on its RUNNING statistics, known before this launch; no dropout): the thread's four
    // samples are exactly one AvgPool(1,4) window, so depthwiseBN -> ELU -> pool leaves here too - pool_fwd_kernel<4> and its
    // read of z (164 MB at the bench shape) disappear from eval-mode steps.  The same arithmetic as pool_fwd_kernel<4>.
    if (p2 && mine && t + 3 < S) {
      const int ch = f * DD + d, CH = F1 * DD;
      const float sc2 = bn2[2 * CH + ch], sh2 = bn2[3 * CH + ch];
      float ps = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) ps += elu_f(sc2 * acc[d][e] + sh2);
      p2[((int64_t)b * CH + ch) * (S / 4) + t / 4] = ps * 0.25f;
    }
  }
  block_sum_nt<NT, 16>(st, red);
  if (threadIdx.x < 16) {
    const int nch = F1 * DD;
    float* dst = part + ((int64_t)b * gridDim.x + chunk) * 2 * nch;
    const int d = threadIdx.x & 7;
    dst[(threadIdx.x < 8 ? 0 : nch) + f * DD + d] = st[0];
  }
}

// ------------------------------------------------------------------------------------ pool_fwd
// one block per (b, ch) row: out[b,ch,to] = drop * mean_{j<P} ELU(scale*in[b,ch,to*P+j] + shift)
template <int P>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ in, const float* __restrict__ bn,
                                                       float* __restrict__ out, int CH, int T, float drop_p,
                                                       uint64_t seed_in,
    const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev) {
  const uint64_t seed = dropout_seed(seed_in, seed_dev);
  const int row = blockIdx.x, ch = row % CH;
  const float sc = bn[2 * CH + ch], sh = bn[3 * CH + ch];
  const int To = T / P;
  const float* src = in + (int64_t)row * T;
  const bool vec = (T & 3) == 0;
  for (int to = threadIdx.x; to < To; to += 256) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < P; j += 4) {
      float v[4];
      ld4(src, to * P + j, T, vec, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += elu_f(sc * v[e] + sh);
    }
    const uint64_t oi = (uint64_t)row * To + to;
    out[oi] = s * (1.0f / P) * dropout_mult_row(drop_p, seed, mask, oi, (uint64_t)row);
  }
}

// ------------------------------------------------------------------------------------ pool_bwd
// g[b,ch,t] = dp[b,ch,t/P]/P * drop * ELU'(v),  v = scale*u + shift,  (0 for the dropped tail)
// pass 1: part[b][ch] = sum_t g, part[b][CH+ch] = sum_t g*uhat
template <int P>
__global__ __launch_bounds__(256) void pool_bwd_reduce_kernel(const float* __restrict__ dp,
                                                              const float* __restrict__ u,
                                                              const float* __restrict__ bn, float* __restrict__ part,
                                                              int CH, int T, float drop_p, uint64_t seed_in,
    const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev) {
  const uint64_t seed = dropout_seed(seed_in, seed_dev);
  __shared__ float red[8];
  const int row = blockIdx.x, ch = row % CH, b = row / CH;
  const float mean = bn[ch], invstd = bn[CH + ch], sc = bn[2 * CH + ch], sh = bn[3 * CH + ch];
  const int To = T / P;
  const float* src = u + (int64_t)row * T;
  const bool vec = (T & 3) == 0;
  float st[2] = {0.f, 0.f};
  for (int to = threadIdx.x; to < To; to += 256) {
    const uint64_t oi = (uint64_t)row * To + to;
    const float go = dp[oi] * (1.0f / P) * dropout_mult_row(drop_p, seed, mask, oi, (uint64_t)row);
#pragma unroll
    for (int j = 0; j < P; j += 4) {
      float v[4];
      ld4(src, to * P + j, T, vec, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pre = sc * v[e] + sh;
        const float g = go * elu_grad_from_out(pre, elu_f(pre));
        st[0] += g;
        st[1] += g * ((v[e] - mean) * invstd);
      }
    }
  }
  block_sum_256<2>(st, red);
  if (threadIdx.x < 2) part[(int64_t)b * 2 * CH + threadIdx.x * CH + ch] = st[0];
}

// pass 2: du = scale*(g - m1 - uhat*m2) for every t < T (tail included: g = 0 there)
// m12 NULL (BatchNorm on its running statistics: m1 = m2 = 0, du = scale g needs no sums first): the ONE pass of an eval-mode
// step - the sums of pass 1 (the gradients of the BatchNorm weight and bias) go to `part` from here.
template <int P>
__global__ __launch_bounds__(256) void pool_bwd_apply_kernel(const float* __restrict__ dp,
                                                             const float* __restrict__ u,
                                                             const float* __restrict__ bn,
                                                             const float* __restrict__ m12, float* __restrict__ du,
                                                             int CH, int T, float drop_p, uint64_t seed_in,
    const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev, float* __restrict__ part) {
  const uint64_t seed = dropout_seed(seed_in, seed_dev);
  __shared__ float red[8];
  const int row = blockIdx.x, ch = row % CH;
  const float mean = bn[ch], invstd = bn[CH + ch], sc = bn[2 * CH + ch], sh = bn[3 * CH + ch];
  const float m1 = m12 ? m12[ch] : 0.f, m2 = m12 ? m12[CH + ch] : 0.f;
  float st[2] = {0.f, 0.f};
  const int To = T / P;
  const float* src = u + (int64_t)row * T;
  float* dst = du + (int64_t)row * T;
  const bool vec = (T & 3) == 0;
  for (int q = threadIdx.x; q < (T + 3) / 4; q += 256) {
    const int t = 4 * q;
    const int to = t / P;  // 4 | P: the quad lies inside one pooling window
    float go = 0.f;
    if (to < To) {
      const uint64_t oi = (uint64_t)row * To + to;
      go = dp[oi] * (1.0f / P) * dropout_mult_row(drop_p, seed, mask, oi, (uint64_t)row);
    }
    float v[4], o[4];
    ld4(src, t, T, vec, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pre = sc * v[e] + sh;
      const float g = go * elu_grad_from_out(pre, elu_f(pre));
      o[e] = sc * (g - m1 - (v[e] - mean) * invstd * m2);
      if (t + e < T) {                       // (go = 0 on the dropped tail: it adds nothing, as in pass 1)
        st[0] += g;
        st[1] += g * ((v[e] - mean) * invstd);
      }
    }
    st4(dst, t, T, vec, o);
  }
  if (part) {
    block_sum_256<2>(st, red);
    if (threadIdx.x < 2) part[(int64_t)(row / CH) * 2 * CH + threadIdx.x * CH + ch] = st[0];
  }
}

// -------------------------------------------------------------------------------------- dw_bwd
// grid (nchunk, F1, B) as dw_fwd.  For every (c, t):
//   v = scale1*y1 + shift1, a1 = ELU(v), da1 = sum_d w2[f*8+d,c]*dz[b,f*8+d,t], g = da1*ELU'(v)
//   g1[b,f,c,t] = g;  stats: sum g, sum g*yhat;  dW2[f*8+d,c] += sum_t dz[d,t]*a1[c,t]
// part_st[(b*nchunk+chunk)][2*8] (slot f), part_w[(b*nchunk+chunk)][64*C] (this block's 8*C slice)
// FUSED: dz is not read from memory but formed in the prologue from z (depthwiseConv output), dp2 (gradient of the pooled
// block-1 output) and the depthwiseBN backward coefficients - the arithmetic of pool_bwd_apply_kernel<4>, whose launch
// and whose dz round trip through HBM (2 x 164 MB at the bench shape) disappear.
// DPP move of src under control CTRL into the lanes of the banks in BANK_MASK (the others keep `old`)
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, BANK_MASK, false));
}

// Sum of a over the wave, wave-uniform, by DPP only (in-row inclusive scan, the two row broadcasts, lane 63): no LDS round
// trips and no barrier (eegnet_fir_fft.hip's wave_total)
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_row(float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, ROW_MASK, 0xf, BOUND));
}
__device__ __forceinline__ float wave_total_dpp(float a) {
  a += dpp_row<0x111, 0xf, true>(a);      // row_shr:1
  a += dpp_row<0x112, 0xf, true>(a);      // row_shr:2
  a += dpp_row<0x114, 0xf, true>(a);      // row_shr:4
  a += dpp_row<0x118, 0xf, true>(a);      // row_shr:8
  a += dpp_row<0x142, 0xa, false>(a);     // row_bcast:15 into rows 1 and 3
  a += dpp_row<0x143, 0xc, false>(a);     // row_bcast:31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
}

struct DwFuse {
  const float* z; const float* dp2; const float* bn2;   // bn2: mean, invstd, scale, shift, m1, m2 (64 each)
  float drop_p; uint64_t seed; const uint8_t* mask; const uint64_t* seed_dev;
  float* part2;     // eval-mode step (depthwiseBN on its running statistics: m1 = m2 = 0): the sums of pool_bwd_reduce_kernel<4>
                    // leave from here, part2[(b * nchunk + chunk)][2 * 64] - that launch and its read of z and dp2 disappear
};

// CG: channel groups for short rows, as in dw_fwd_kernel - every group forms dz for the row's samples (the same loads: L1
// hits), group j walks channels j, j + CG, ...; g1 and the depthwise weight gradient are per channel, the two statistics
// are block sums anyway.
#ifndef DWB_WPE          // A/B builds: minimum waves per SIMD asked of the register allocator (0 = none), rows fetched ahead
#define DWB_WPE 0
#endif
#ifndef DWB_PF
#define DWB_PF 2
#endif
template <bool FUSED, int CG, int NT, bool SUMS2 = false>      // SUMS2: the eval-mode form (fu.part2)
__global__ __launch_bounds__(NT, DWB_WPE) void dw_bwd_kernel(const float* __restrict__ y1, const float* __restrict__ dz,
                                                     const float* __restrict__ bn1, const float* __restrict__ w2,
                                                     float* __restrict__ g1, float* __restrict__ part_st,
                                                     float* __restrict__ part_w, int C, int S, DwFuse fu) {
  __shared__ float wsh[DD * CHMAX];
  __shared__ float red[NT / 64 * 8];
  __shared__ float red2[SUMS2 ? NT / 64 * 2 * DD : 1];
  __shared__ float wacc[NT / 64 * DD * CHMAX];
  constexpr int TPG = NT / CG;            // threads per channel group (a multiple of the wave size)
  // samples in DESCENDING order (the FFT weight-gradient kernel behind this one walks them ascending): measured -3 .. -8 us
  // on the step on one box, alternating runs (profiles/r05_eeg_variants.txt); dw_fwd walking backwards costs +5 us
#ifdef DWB_ASC
  const int chunk = blockIdx.x, f = blockIdx.y, b = blockIdx.z;
#else
  const int chunk = blockIdx.x, f = blockIdx.y, b = gridDim.z - 1 - blockIdx.z;
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cg = CG > 1 ? threadIdx.x / TPG : 0, tl = CG > 1 ? threadIdx.x % TPG : threadIdx.x;
  for (int i = threadIdx.x; i < DD * C; i += NT) wsh[i] = w2[f * DD * C + i];
  if (CG > 1)                             // a wave only fills its own group's channels of its wacc slot
    for (int i = threadIdx.x; i < NT / 64 * DD * CHMAX; i += NT) wacc[i] = 0.f;
  __syncthreads();
  const float mean = bn1[f], invstd = bn1[8 + f], sc = bn1[16 + f], sh = bn1[24 + f];
  const bool vec = (S & 3) == 0;
  const int t = chunk * (4 * TPG) + tl * 4;
  float dzv[DD][4];
  if (FUSED) {
    float st2[2 * DD];
#pragma unroll
    for (int k = 0; k < 2 * DD; ++k) st2[k] = 0.f;
    const uint64_t seed = dropout_seed(fu.seed, fu.seed_dev);
    const int To = S / 4, to = t / 4;
#pragma unroll
    for (int d = 0; d < DD; ++d) {
      const int ch = f * DD + d, CH = F1 * DD;
      const int64_t row = (int64_t)b * CH + ch;
      const float mean2 = fu.bn2[ch], invstd2 = fu.bn2[CH + ch], sc2 = fu.bn2[2 * CH + ch], sh2 = fu.bn2[3 * CH + ch];
      const float m1 = SUMS2 ? 0.f : fu.bn2[4 * CH + ch], m2 = SUMS2 ? 0.f : fu.bn2[5 * CH + ch];
      float go = 0.f;
      if (to < To) {
        const uint64_t oi = (uint64_t)row * To + to;
        go = fu.dp2[oi] * 0.25f * dropout_mult_row(fu.drop_p, seed, fu.mask, oi, (uint64_t)row);
      }
      float v[4];
      ld4s(fu.z + row * S, t < S ? t : S, S, vec, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pre = sc2 * v[e] + sh2;
        const float g = go * elu_grad_from_out(pre, elu_f(pre));
        dzv[d][e] = (t + e < S) ? sc2 * (g - m1 - (v[e] - mean2) * invstd2 * m2) : 0.f;
        if (SUMS2 && t + e < S) {
          st2[d] += g;
          st2[DD + d] += g * ((v[e] - mean2) * invstd2);
        }
      }
    }
    if constexpr (SUMS2) {
      // wave sums by DPP into this wave's LDS row - no barrier here (one in front of the y1 loads below cost 40 us of the
      // 310: the loads could no longer start under the prologue); the rows are added at the end of the kernel.  With
      // channel groups every group formed the same dz: the waves of group 0 are the ones summed.
#pragma unroll
      for (int k = 0; k < 2 * DD; ++k) {
        const float tot = wave_total_dpp(st2[k]);
        if (lane == 0) red2[wave * 2 * DD + k] = tot;
      }
    }
  } else {
#pragma unroll
    for (int d = 0; d < DD; ++d) ld4(dz + ((int64_t)b * F1 * DD + f * DD + d) * S, t < S ? t : S, S, vec, dzv[d]);
  }
  float st[2] = {0.f, 0.f};
  const int64_t base = ((int64_t)b * F1 + f) * C * S;
  // y1 rows in flight per thread ahead of the one being worked on (16 bytes each).  (Round 5: 4 / 6 rows ahead measured
  // 399 / 350 us against 349 us with 2 at [64,1,30,10000] - the pass is not limited by bytes in flight.)
  constexpr int PF = DWB_PF;
  float nx[PF][4];
#pragma unroll
  for (int q = 0; q < PF; ++q)
    ld4s(y1 + base + (int64_t)min(cg + q * CG, C - 1) * S, t < S ? t : S, S, vec, nx[q]);
  for (int c = cg; c < C; c += CG) {
    float v[4], g[4], a[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = nx[0][e];
#pragma unroll
      for (int q = 0; q + 1 < PF; ++q) nx[q][e] = nx[q + 1][e];
    }
    ld4s(y1 + base + (int64_t)min(c + PF * CG, C - 1) * S, t < S ? t : S, S, vec, nx[PF - 1]);
    float wd[DD];
#pragma unroll
    for (int d = 0; d < DD; ++d) wd[d] = wsh[d * C + c];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pre = sc * v[e] + sh;
      a[e] = elu_f(pre);
      float da = 0.f;
#pragma unroll
      for (int d = 0; d < DD; ++d) da += wd[d] * dzv[d][e];
      g[e] = da * elu_grad_from_out(pre, a[e]);
      st[0] += g[e];
      st[1] += g[e] * ((v[e] - mean) * invstd);
    }
    if (t < S) st4s(g1 + base + (int64_t)c * S, t, S, vec, g);
    // depthwise weight gradient: 8 values per thread -> transposing butterfly over the wave
    float r[DD];
#pragma unroll
    for (int d = 0; d < DD; ++d)
      // (explicit fmas: left to -ffp-contract, two instantiations of this template may fuse the four products differently
      // - the eval-mode form must reproduce the train-mode form's bits, tests/test_eegnet_kernels_gpu.py)
      r[d] = __builtin_fmaf(dzv[d][0], a[0], dzv[d][1] * a[1]) + __builtin_fmaf(dzv[d][2], a[2], dzv[d][3] * a[3]);
    // step xor 1: keep 4 values; xor 2: keep 2; xor 4: keep 1; then plain reduce over the rest
    float r4[4], r2[2], r1;
#ifdef DWB_ABL_NORED      // timing-only ablation (results garbage): the pass without its cross-lane reductions
    r1 = r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7];
#else
    // (the partner values travel by DPP quad permutations / masked row shifts / v_permlane swaps: the same pairs and the same
    // order of additions as the __shfl_xor butterfly this replaces - ten dependent ds_bpermute round trips per channel)
    {
      const bool hi = lane & 1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float mine = hi ? r[4 + k] : r[k], other = hi ? r[k] : r[4 + k];
        r4[k] = mine + dpp_f<0xB1, 0xf>(0.f, other);                 // quad_perm:[1,0,3,2] = lane ^ 1
      }
    }
    {
      const bool hi = lane & 2;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float mine = hi ? r4[2 + k] : r4[k], other = hi ? r4[k] : r4[2 + k];
        r2[k] = mine + dpp_f<0x4E, 0xf>(0.f, other);                 // quad_perm:[2,3,0,1] = lane ^ 2
      }
    }
    {
      const bool hi = lane & 4;
      float mine = hi ? r2[1] : r2[0], other = hi ? r2[0] : r2[1];
      float part = dpp_f<0x104, 0x5>(0.f, other);                    // row_shl:4 into lanes 0-3 / 8-11 of every row
      part = dpp_f<0x114, 0xA>(part, other);                         // row_shr:4 into lanes 4-7 / 12-15: lane ^ 4
      r1 = mine + part;
    }
    r1 += dpp_f<0x128, 0xf>(0.f, r1);                                // row_ror:8 = lane ^ 8
    {
      auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(r1), __float_as_uint(r1), false, false);
      r1 = __uint_as_float(q[0]) + __uint_as_float(q[1]);            // own + lane ^ 16 (the two rows side by side)
      auto h = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1), __float_as_uint(r1), false, false);
      r1 = __uint_as_float(h[0]) + __uint_as_float(h[1]);            // own + lane ^ 32
    }
#endif
    // lane l (l < 8) now holds the wave total of d = 4*(l&1) + 2*((l>>1)&1) + ((l>>2)&1)
    if (lane < 8) {
      const int d = 4 * (lane & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 2) & 1);
      wacc[wave * DD * CHMAX + d * C + c] = r1;  // one slot per wave: summed in a fixed order below
    }
  }
  block_sum_nt<NT, 2>(st, red);
  if (threadIdx.x < 2) part_st[((int64_t)b * gridDim.x + chunk) * 16 + threadIdx.x * 8 + f] = st[0];
  __syncthreads();
  if (FUSED && SUMS2 && threadIdx.x < 2 * DD) {      // (red2 was written before the barriers above)
    constexpr int NWG = NT / 64 / CG;                   // waves of channel group 0: threads 0 .. NT / CG - 1
    float t = red2[threadIdx.x];
#pragma unroll
    for (int w = 1; w < NWG; ++w) t += red2[w * 2 * DD + threadIdx.x];
    const int nch = F1 * DD;
    float* dst2 = fu.part2 + ((int64_t)b * gridDim.x + chunk) * 2 * nch;
    dst2[(threadIdx.x < DD ? 0 : nch) + f * DD + (threadIdx.x & (DD - 1))] = t;
  }
  float* dst = part_w + ((int64_t)b * gridDim.x + chunk) * (F1 * DD * C) + f * DD * C;
  for (int i = threadIdx.x; i < DD * C; i += NT) {
    float w = (wacc[i] + wacc[DD * CHMAX + i]) + (wacc[2 * DD * CHMAX + i] + wacc[3 * DD * CHMAX + i]);
    if constexpr (NT == 512)
      w += (wacc[4 * DD * CHMAX + i] + wacc[5 * DD * CHMAX + i]) + (wacc[6 * DD * CHMAX + i] + wacc[7 * DD * CHMAX + i]);
    dst[i] = w;
  }
}

}  // namespace

static int dw_fwd_launch(const float* y1, const float* bn1, const float* w2, float* z, float* stat_part, int B, int C,
                         int S, const float* bn2, float* p2, void* stream) {
  dim3 grid(cdiv(S, 1024), F1, B);
  if (S <= 256)         // short rows: channel groups (see the kernel)
    hipLaunchKernelGGL((dw_fwd_kernel<4, 256>), grid, dim3(256), 0, (hipStream_t)stream, y1, bn1, w2, z, stat_part, C, S,
                       bn2, p2);
  else if (S <= 512)
    hipLaunchKernelGGL((dw_fwd_kernel<4, 512>), grid, dim3(512), 0, (hipStream_t)stream, y1, bn1, w2, z, stat_part, C, S,
                       bn2, p2);
  else
    hipLaunchKernelGGL((dw_fwd_kernel<1, 256>), grid, dim3(256), 0, (hipStream_t)stream, y1, bn1, w2, z, stat_part, C, S,
                       bn2, p2);
  return EAV_OK;
}

extern "C" int eav_eegnet_dw_fwd(const float* y1, const float* bn1, const float* w2, float* z, float* stat_part,
                                 int B, int C, int S, void* stream) {
  EAV_REQUIRE(y1 && bn1 && w2 && z && stat_part && B > 0 && C > 0 && C <= CHMAX && S > 0,
              "eav_eegnet_dw_fwd: bad arguments (Chans must be <= %d)", CHMAX);
  dw_fwd_launch(y1, bn1, w2, z, stat_part, B, C, S, nullptr, nullptr, stream);
  EAV_CHECK_LAUNCH("eav_eegnet_dw_fwd");
  return EAV_OK;
}

// eav_eegnet_dw_fwd that also leaves p2 [B,64,S/4] = AvgPool(1,4)(ELU(depthwiseBN(z))) (EEGNet_tor.py:55-57) for a
// depthwiseBN in EVAL mode: bn2 = its finalised parameter block (eav_bn_finalize with training = 0: scale / shift from the
// running statistics), no dropout (nn.Dropout is the identity in eval mode).  S % 4 == 0.
extern "C" int eav_eegnet_dw_fwd_pool_eval(const float* y1, const float* bn1, const float* w2, float* z, float* stat_part,
                                           const float* bn2, float* p2, int B, int C, int S, void* stream) {
  EAV_REQUIRE(y1 && bn1 && w2 && z && stat_part && bn2 && p2 && B > 0 && C > 0 && C <= CHMAX && S > 0 && (S & 3) == 0,
              "eav_eegnet_dw_fwd_pool_eval: bad arguments (Chans <= %d, Samples %% 4 == 0)", CHMAX);
  dw_fwd_launch(y1, bn1, w2, z, stat_part, B, C, S, bn2, p2, stream);
  EAV_CHECK_LAUNCH("eav_eegnet_dw_fwd_pool_eval");
  return EAV_OK;
}

extern "C" int eav_eegnet_dw_bwd(const float* y1, const float* dz, const float* bn1, const float* w2, float* g1,
                                 float* stat_part, float* w_part, int B, int C, int S, void* stream) {
  EAV_REQUIRE(y1 && dz && bn1 && w2 && g1 && stat_part && w_part && B > 0 && C > 0 && C <= CHMAX && S > 0,
              "eav_eegnet_dw_bwd: bad arguments (Chans must be <= %d)", CHMAX);
  dim3 grid(cdiv(S, 1024), F1, B);
  hipLaunchKernelGGL((dw_bwd_kernel<false, 1, 256>), grid, dim3(256), 0, (hipStream_t)stream, y1, dz, bn1, w2, g1,
                     stat_part, w_part, C, S, DwFuse{});
  EAV_CHECK_LAUNCH("eav_eegnet_dw_bwd");
  return EAV_OK;
}

// dw_bwd with the depthwiseBN -> ELU -> AvgPool(1,4) -> Dropout backward folded into its prologue (see DwFuse): takes z and
// dp2 [B,64,S/4] instead of dz.  bn2: mean, invstd, scale, shift, m1, m2 (64 floats each, m1/m2 from eav_bn_bwd_finalize).
static int dw_bwd_fused_launch(const char* name, const float* y1, const float* z, const float* dp2, const float* bn2,
                               const float* bn1, const float* w2, float* g1, float* stat_part, float* w_part, float* bn2_part,
                               int B, int C, int S, float drop_p, uint64_t seed, const uint8_t* mask,
                               const uint64_t* seed_dev, void* stream) {
  EAV_REQUIRE(y1 && z && dp2 && bn2 && bn1 && w2 && g1 && stat_part && w_part && B > 0 && C > 0 && C <= CHMAX && S >= 4,
              "%s: bad arguments (Chans must be <= %d)", name, CHMAX);
  EAV_REQUIRE(drop_p > -1.f && drop_p < 1.f, "%s: dropout %f outside (-1,1)", name, drop_p);
  dim3 grid(cdiv(S, 1024), F1, B);
  DwFuse fu{z, dp2, bn2, drop_p, seed, mask, seed_dev, bn2_part};
#define DWB_LAUNCH(CG, NT, SUMS)                                                                                        \
  hipLaunchKernelGGL((dw_bwd_kernel<true, CG, NT, SUMS>), grid, dim3(NT), 0, (hipStream_t)stream, y1, nullptr, bn1, w2, g1, \
                     stat_part, w_part, C, S, fu)
  if (bn2_part) {
    if (S <= 256) DWB_LAUNCH(4, 256, true);
    else if (S <= 512) DWB_LAUNCH(4, 512, true);
    else DWB_LAUNCH(1, 256, true);
  } else {
    if (S <= 256) DWB_LAUNCH(4, 256, false);
    else if (S <= 512) DWB_LAUNCH(4, 512, false);
    else DWB_LAUNCH(1, 256, false);
  }
#undef DWB_LAUNCH
  EAV_CHECK_LAUNCH(name);
  return EAV_OK;
}

extern "C" int eav_eegnet_dw_bwd_fused(const float* y1, const float* z, const float* dp2, const float* bn2,
                                       const float* bn1, const float* w2, float* g1, float* stat_part, float* w_part,
                                       int B, int C, int S, float drop_p, uint64_t seed, const uint8_t* mask,
                                       const uint64_t* seed_dev, void* stream) {
  return dw_bwd_fused_launch("eav_eegnet_dw_bwd_fused", y1, z, dp2, bn2, bn1, w2, g1, stat_part, w_part, nullptr, B, C, S,
                             drop_p, seed, mask, seed_dev, stream);
}

// The same with depthwiseBN on its RUNNING statistics (an eval-mode step): dz = scale2 g needs no batch sums first, so the
// sums themselves (sum g, sum g zhat per channel: the BatchNorm weight / bias gradients) leave from this pass -
// bn2_part[(b * ceil(S / 1024) + chunk)][2 * 64], to be finished by eav_bn_bwd_finalize(training = 0) AFTER this launch; no
// eav_bn_elu_pool_bwd_reduce pass.  bn2's m1 / m2 slots are not read.
extern "C" int eav_eegnet_dw_bwd_fused_eval(const float* y1, const float* z, const float* dp2, const float* bn2,
                                            const float* bn1, const float* w2, float* g1, float* stat_part, float* w_part,
                                            float* bn2_part, int B, int C, int S, float drop_p, uint64_t seed,
                                            const uint8_t* mask, const uint64_t* seed_dev, void* stream) {
  EAV_REQUIRE(bn2_part, "eav_eegnet_dw_bwd_fused_eval: bad arguments");
  return dw_bwd_fused_launch("eav_eegnet_dw_bwd_fused_eval", y1, z, dp2, bn2, bn1, w2, g1, stat_part, w_part, bn2_part, B, C,
                             S, drop_p, seed, mask, seed_dev, stream);
}

extern "C" int eav_bn_elu_pool_fwd(const float* in, const float* bn, float* out, int B, int CH, int T, int P,
                                   float drop_p, uint64_t seed, const uint8_t* mask, const uint64_t* seed_dev, void* stream) {
  EAV_REQUIRE(in && bn && out && B > 0 && CH > 0 && T >= P, "eav_bn_elu_pool_fwd: bad arguments");
  EAV_REQUIRE(P == 4 || P == 8, "eav_bn_elu_pool_fwd: pool %d not in {4,8}", P);
  EAV_REQUIRE(drop_p > -1.f && drop_p < 1.f, "eav_bn_elu_pool_fwd: dropout %f outside (-1,1)", drop_p);
  if (P == 4)
    hipLaunchKernelGGL(pool_fwd_kernel<4>, dim3(B * CH), dim3(256), 0, (hipStream_t)stream, in, bn, out, CH, T,
                       drop_p, seed, mask, seed_dev);
  else
    hipLaunchKernelGGL(pool_fwd_kernel<8>, dim3(B * CH), dim3(256), 0, (hipStream_t)stream, in, bn, out, CH, T,
                       drop_p, seed, mask, seed_dev);
  EAV_CHECK_LAUNCH("eav_bn_elu_pool_fwd");
  return EAV_OK;
}

extern "C" int eav_bn_elu_pool_bwd_reduce(const float* dp, const float* u, const float* bn, float* part, int B,
                                          int CH, int T, int P, float drop_p, uint64_t seed, const uint8_t* mask,
                                          const uint64_t* seed_dev, void* stream) {
  EAV_REQUIRE(dp && u && bn && part && B > 0 && CH > 0 && T >= P, "eav_bn_elu_pool_bwd_reduce: bad arguments");
  EAV_REQUIRE(P == 4 || P == 8, "eav_bn_elu_pool_bwd_reduce: pool %d not in {4,8}", P);
  if (P == 4)
    hipLaunchKernelGGL(pool_bwd_reduce_kernel<4>, dim3(B * CH), dim3(256), 0, (hipStream_t)stream, dp, u, bn, part,
                       CH, T, drop_p, seed, mask, seed_dev);
  else
    hipLaunchKernelGGL(pool_bwd_reduce_kernel<8>, dim3(B * CH), dim3(256), 0, (hipStream_t)stream, dp, u, bn, part,
                       CH, T, drop_p, seed, mask, seed_dev);
  EAV_CHECK_LAUNCH("eav_bn_elu_pool_bwd_reduce");
  return EAV_OK;
}

static int pool_bwd_apply_launch(const char* name, const float* dp, const float* u, const float* bn, const float* m12,
                                 float* du, float* part, int B, int CH, int T, int P, float drop_p, uint64_t seed,
                                 const uint8_t* mask, const uint64_t* seed_dev, void* stream) {
  EAV_REQUIRE(dp && u && bn && (m12 || part) && du && B > 0 && CH > 0 && T >= P, "%s: bad arguments", name);
  EAV_REQUIRE(P == 4 || P == 8, "%s: pool %d not in {4,8}", name, P);
  if (P == 4)
    hipLaunchKernelGGL(pool_bwd_apply_kernel<4>, dim3(B * CH), dim3(256), 0, (hipStream_t)stream, dp, u, bn, m12, du,
                       CH, T, drop_p, seed, mask, seed_dev, part);
  else
    hipLaunchKernelGGL(pool_bwd_apply_kernel<8>, dim3(B * CH), dim3(256), 0, (hipStream_t)stream, dp, u, bn, m12, du,
                       CH, T, drop_p, seed, mask, seed_dev, part);
  EAV_CHECK_LAUNCH(name);
  return EAV_OK;
}

extern "C" int eav_bn_elu_pool_bwd_apply(const float* dp, const float* u, const float* bn, const float* m12,
                                         float* du, int B, int CH, int T, int P, float drop_p, uint64_t seed,
                                         const uint8_t* mask, const uint64_t* seed_dev, void* stream) {
  EAV_REQUIRE(m12, "eav_bn_elu_pool_bwd_apply: bad arguments");
  return pool_bwd_apply_launch("eav_bn_elu_pool_bwd_apply", dp, u, bn, m12, du, nullptr, B, CH, T, P, drop_p, seed, mask,
                               seed_dev, stream);
}

// BatchNorm on its running statistics (an eval-mode step): du = scale g and the sums of eav_bn_elu_pool_bwd_reduce
// (part [B][2*CH], finished by eav_bn_bwd_finalize(training = 0)) in ONE pass over u and dp.
extern "C" int eav_bn_elu_pool_bwd_eval(const float* dp, const float* u, const float* bn, float* du, float* part, int B,
                                        int CH, int T, int P, float drop_p, uint64_t seed, const uint8_t* mask,
                                        const uint64_t* seed_dev, void* stream) {
  EAV_REQUIRE(part, "eav_bn_elu_pool_bwd_eval: bad arguments");
  return pool_bwd_apply_launch("eav_bn_elu_pool_bwd_eval", dp, u, bn, nullptr, du, part, B, CH, T, P, drop_p, seed, mask,
                               seed_dev, stream);
}
