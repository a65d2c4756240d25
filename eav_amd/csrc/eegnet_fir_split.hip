// EEGNet temporal FIR on the fp16 matrix cores with SPLIT operands - an opt-in fp32-grade fast path
// (EEGNet_tor.fir_precision = "split"; the default stays the exact-fp32 MFMA kernels of eegnet_fir.hip).
//
// Every fp32 operand v is represented by two fp16 pieces of the pre-scaled value sigma*v (sigma a power of two chosen
// from the tensor's absolute maximum so that the pieces stay inside fp16's normal range):
//     hi = fp16(sigma v),   lo = fp16((sigma v - hi) * 2^11)        =>  sigma v = hi + 2^-11 lo + O(2^-22 |sigma v|)
// and a product by three MFMAs with fp32 accumulation (fp16 x fp16 products are exact in fp32):
//     (sx x)(sw w) ~= hi_x hi_w + 2^-11 (hi_x lo_w + lo_x hi_w)      dropped term: 2^-22 lo_x lo_w <= 2^-24 |sx x sw w|
// i.e. the per-product error is at the level of one fp32 rounding.  v_mfma_f32_32x32x16_f16 retires 16 taps per 32
// cycles against 2 taps per 64 cycles of v_mfma_f32_32x32x2_f32: 3 split MFMAs cost 5.3x less matrix time, which
// leaves the forward bound by the y1 write and the weight gradient by the y1/g1 reads.
//
// Same Toeplitz formulation as eegnet_fir.hip: C[(f,s), n] = sum_j w[f, j-s] * xpad[t0 + 4n + j] = y[f, t0 + 4n + s].
// The weight pieces stay in registers (2 x NKS x 4 VGPRs); the x pieces are read from a linear fp16 LDS image, eight
// consecutive taps per lane (two 8-byte reads per piece).  Feed budget: one fresh 1 KB operand per 32-cycle MFMA is
// what a CU's LDS can deliver to four SIMDs, so only the x pieces may come from LDS.
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int F1 = 8;
constexpr int TILE = 128;   // samples per MFMA tile
constexpr int TPS = 16;     // tiles per LDS segment
constexpr float LO_SCALE = 2048.f, LO_INV = 1.f / 2048.f;

// The scales put every tensor's maximum at <= 2^12; the clamp only matters for a dy outlier beyond the 16-sigma bound
// eav_fir_dy_scale assumes for the normalised activations (it then saturates instead of turning into inf/NaN).
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

// ------------------------------------------------------------------------------------------ scales
// One launch: every block writes max |v| of its slice to part[1 + blk]; the block that finishes last (device counter
// at part[0], left at zero again for the next call) reduces the partials and writes
// scale[0] = sigma = 2^(12 - ceil(log2 max)), scale[1] = 1/sigma, scale[2] = max.
__global__ __launch_bounds__(256) void absmax_scale_kernel(const float* __restrict__ v, int64_t n, float extra,
                                                           float* __restrict__ part, float* __restrict__ scale) {
  __shared__ float red[4];
  __shared__ bool last;
  const int nparts = gridDim.x;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(v[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[1 + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __threadfence();
    unsigned* counter = reinterpret_cast<unsigned*>(part);
    last = atomicAdd(counter, 1u) == (unsigned)nparts - 1u;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  m = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, __builtin_nontemporal_load(part + 1 + i));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * extra;
    int e = 0;
    if (m > 0.f && isfinite(m)) frexpf(m, &e);       // m = f * 2^e, f in [0.5, 1)
    const float s = ldexpf(1.f, 12 - e);             // |sigma v| <= 2^12: 4 binades below fp16's maximum
    scale[0] = s;
    scale[1] = 1.f / s;
    scale[2] = m;
    *reinterpret_cast<unsigned*>(part) = 0u;
  }
}

__device__ __forceinline__ float block_max_256(float m, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ void write_scale(float m, float* __restrict__ out) {
  int e = 0;
  if (m > 0.f && isfinite(m)) frexpf(m, &e);
  const float s = ldexpf(1.f, 12 - e);
  out[0] = s;
  out[1] = 1.f / s;
  out[2] = m;
}
// scale from per-block maxima emitted by the producing kernel (no extra pass over the tensor)
__global__ __launch_bounds__(256) void absmax_finish_kernel(const float* __restrict__ part, int nparts, float extra,
                                                            float* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) m = fmaxf(m, part[i]);
  m = block_max_256(m, red);
  if (threadIdx.x == 0) write_scale(m * extra, out);
}
// sigma for dy = scale_f * (g - m1_f - xhat * m2_f) where g = ELU'(.) * sum_d w2[f*8+d, c] * dz[f*8+d, t]:
//   |g| <= max_c sum_d |w2[f*8+d, c]| * max|dz|     (ELU' <= 1),     |xhat| <= 16 assumed (saturating clamp beyond)
__global__ __launch_bounds__(256) void dy_scale_kernel(const float* __restrict__ bnp, const float* __restrict__ dzpart,
                                                       int nparts, const float* __restrict__ w2, int C,
                                                       float* __restrict__ out) {
  __shared__ float red[4];
  __shared__ float wn[F1];
  float dzmax = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) dzmax = fmaxf(dzmax, dzpart[i]);
  dzmax = block_max_256(dzmax, red);
  if (threadIdx.x < F1) {
    float m = 0.f;
    for (int c = 0; c < C; ++c) {
      float s = 0.f;
      for (int d = 0; d < 8; ++d) s += fabsf(w2[(threadIdx.x * 8 + d) * C + c]);
      m = fmaxf(m, s);
    }
    wn[threadIdx.x] = m;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float m = 0.f;
  for (int f = 0; f < F1; ++f)
    m = fmaxf(m, fabsf(bnp[16 + f]) * (wn[f] * dzmax + fabsf(bnp[32 + f]) + 16.f * fabsf(bnp[40 + f])));
  write_scale(m, out);
}

// ------------------------------------------------------------------------------------------ fwd
template <int NKS>
__global__ __launch_bounds__(256, 2) void fir_fwd_split_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                               const float* __restrict__ sx, const float* __restrict__ sw,
                                                               float* __restrict__ y1, float* __restrict__ part, int rows,
                                                               int C, int S, int klen, int padl, int nseg, int ntiles) {
  constexpr int HALO = 16 * NKS;
  constexpr int SEG = TPS * TILE + HALO;
  __shared__ __attribute__((aligned(16))) _Float16 xh[SEG];
  __shared__ __attribute__((aligned(16))) _Float16 xl[SEG];
  __shared__ float red[4 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, g = lane >> 5;
  const float sigx = sx[0], sigw = sw[0];
  const float post = sx[1] * sw[1];

  // A operand pieces, resident: row i = f*4 + s, taps j = 16 ks + 8 g + e  ->  w[f, j - s]
  h8 ah[NKS], al[NKS];
  {
    const int f = n >> 2, s = n & 3;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 16 * ks + 8 * g + e - s;
        const float v = (j >= 0 && j < klen) ? sigw * w1[f * klen + j] : 0.f;
        _Float16 hi, lo;
        split2(v, hi, lo);
        ah[ks][e] = hi;
        al[ks][e] = lo;
      }
    }
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (S & 3) == 0;
  const int nwork = rows * nseg;
  constexpr int NLD = (SEG / 2 + 255) / 256;          // sample PAIRS per thread (packed 4-byte LDS stores)
  float2 xr[NLD];
  auto fetch = [&](int work) {
    const int row = work / nseg, seg = work - row * nseg;
    const int tile0 = seg * TPS;
    const int nt = min(TPS, ntiles - tile0);
    const int useg0 = tile0 * TILE;
    const float* xrow = x + (int64_t)row * S;
    const int nload = nt * TILE + HALO;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = 2 * (threadIdx.x + 256 * i);
      const int t = useg0 + idx - padl;
      xr[i].x = (idx < nload && t >= 0 && t < S) ? xrow[t] : 0.f;
      xr[i].y = (idx + 1 < nload && t + 1 >= 0 && t + 1 < S) ? xrow[t + 1] : 0.f;
    }
  };
  if ((int)blockIdx.x < nwork) fetch(blockIdx.x);
  for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
    const int row = work / nseg, seg = work - row * nseg;
    const int tile0 = seg * TPS;
    const int nt = min(TPS, ntiles - tile0);
    const int useg0 = tile0 * TILE;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = 2 * (threadIdx.x + 256 * i);
      if (idx < SEG) {
        _Float16 h0, l0, h1, l1;
        split2(sigx * xr[i].x, h0, l0);
        split2(sigx * xr[i].y, h1, l1);
        *reinterpret_cast<uint32_t*>(&xh[idx]) = pack2(h0, h1);
        *reinterpret_cast<uint32_t*>(&xl[idx]) = pack2(l0, l1);
      }
    }
    __syncthreads();
    if (work + (int)gridDim.x < nwork) fetch(work + gridDim.x);
    const int b = row / C, c = row - b * C;
    for (int tile = wave; tile < nt; tile += 4) {
      const int u0 = tile * TILE + 4 * n + 8 * g;
      f32x16 acc, acc2;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
        const h4 h0 = *reinterpret_cast<const h4*>(&xh[u0 + 16 * ks]);
        const h4 h1 = *reinterpret_cast<const h4*>(&xh[u0 + 16 * ks + 4]);
        const h4 l0 = *reinterpret_cast<const h4*>(&xl[u0 + 16 * ks]);
        const h4 l1 = *reinterpret_cast<const h4*>(&xl[u0 + 16 * ks + 4]);
        const h8 bh = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        const h8 bl = {l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl, acc2, 0, 0, 0);
      }
      // C layout: col = lane&31 (n), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = f*4 + s
      const int t = useg0 + tile * TILE + 4 * n;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = 2 * q + g;
        float* dst = y1 + (((int64_t)b * F1 + f) * C + c) * S + t;
        float v0 = post * fmaf(LO_INV, acc2[4 * q + 0], acc[4 * q + 0]);
        float v1 = post * fmaf(LO_INV, acc2[4 * q + 1], acc[4 * q + 1]);
        float v2 = post * fmaf(LO_INV, acc2[4 * q + 2], acc[4 * q + 2]);
        float v3 = post * fmaf(LO_INV, acc2[4 * q + 3], acc[4 * q + 3]);
        if (vec && t + 3 < S) {
          *reinterpret_cast<float4*>(dst) = make_float4(v0, v1, v2, v3);
        } else {
          if (t + 0 < S) dst[0] = v0; else v0 = 0.f;
          if (t + 1 < S) dst[1] = v1; else v1 = 0.f;
          if (t + 2 < S) dst[2] = v2; else v2 = 0.f;
          if (t + 3 < S) dst[3] = v3; else v3 = 0.f;
        }
        s1[q] += (v0 + v1) + (v2 + v3);
        s2[q] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float a1 = half_sum(s1[q]), a2 = half_sum(s2[q]);
    if (n == 0) {
      red[wave * 16 + 2 * q + g] = a1;
      red[wave * 16 + 8 + 2 * q + g] = a2;
    }
  }
  __syncthreads();
  if (threadIdx.x < 16)
    part[blockIdx.x * 16 + threadIdx.x] =
        (red[threadIdx.x] + red[16 + threadIdx.x]) + (red[32 + threadIdx.x] + red[48 + threadIdx.x]);
}

// ---------------------------------------------------------------------------------------- wgrad
// dW[f, k = 2n+s] = sum_u dy[f, u-s] * xpad[u + 2n] on v_mfma_f32_16x16x32_f16: rows i = f*2+s, columns n (NT tiles of
// 16), contraction over 32 time samples per MFMA.  A block is a PAIR of waves sharing one LDS image of an item =
// (row, 256-sample chunk): each wave stages half of the filters (lane = sample quad) and owns half of the column
// tiles, so its accumulators (2 x TPW x 4 VGPRs) and operand window stay small enough for 3 waves per SIMD - the matrix
// phase of an item is only ~2k cycles, so it is the number of items in flight that hides the HBM latency.
// Two operand-reuse facts keep the LDS out of the way: the x operand of (K-step ks, column tile nt) depends on ks+nt
// only, so a sliding register window needs ONE new operand pair per K-step; the odd-shift rows (s=1) are made in
// registers from the aligned 16-byte read plus the preceding dword (v_alignbit), no second LDS copy.
constexpr int WCH = 256, WKS = WCH / 32;
constexpr int DYS = WCH + 8;        // halfs per dy row; [0,8) is the left pad, [7] holds dy[c0-1]

__device__ __forceinline__ h8 as_h8(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  union { uint32_t u[4]; h8 h; } v;
  v.u[0] = a; v.u[1] = b; v.u[2] = c; v.u[3] = d;
  return v.h;
}

template <int NT>
__global__ __launch_bounds__(128, 3) void fir_wgrad_split_kernel(
    const float* __restrict__ x, const float* __restrict__ y1, const float* __restrict__ g1,
    const float* __restrict__ bnp, const float* __restrict__ sx, const float* __restrict__ sdy,
    float* __restrict__ part, int rows, int C, int S, int klen, int padl, int nchunk) {
  constexpr int LAGS = 32 * NT;
  static_assert(NT % 2 == 0, "the two waves of a block own NT/2 column tiles each");
  constexpr int TPW = NT / 2;                          // column tiles per wave
  constexpr int XW = WCH + LAGS + 8;                   // halfs of the x window of one item (even)
  constexpr int NXP = (XW / 2 + 127) / 128;            // x sample PAIRS per thread
  __shared__ __attribute__((aligned(16))) _Float16 dyh[F1 * DYS];
  __shared__ __attribute__((aligned(16))) _Float16 dyl[F1 * DYS];
  __shared__ __attribute__((aligned(16))) _Float16 xh[XW];
  __shared__ __attribute__((aligned(16))) _Float16 xl[XW];
  __shared__ float aff[3 * F1];                        // dy = aff0*g + aff1*y + aff2 (BatchNorm backward, pre-scaled)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int af = col >> 1, as = col & 1;               // A row i = f*2 + s
  const int nt0 = wave * TPW;                          // this wave's column tiles [nt0, nt0 + TPW)
  constexpr int ntn = TPW;
  const float sigx = sx[0];
  if (threadIdx.x < F1) {
    const int f = threadIdx.x;
    const float mean = bnp[f], invstd = bnp[8 + f], sc = sdy[0] * bnp[16 + f], m1 = bnp[32 + f], m2 = bnp[40 + f];
    aff[f] = sc;
    aff[F1 + f] = -sc * invstd * m2;
    aff[2 * F1 + f] = sc * (mean * invstd * m2 - m1);
  }
  f32x4 c1[TPW], c2[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) { c1[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; c2[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const bool vec = (S & 3) == 0;
  const int nwork = rows * nchunk;
  float4 ry[4], rg[4];
  float ryh = 0.f, rgh = 0.f;
  float2 rx[NXP];
  auto fetch = [&](int work) {
    const int row = work / nchunk, chunk = work - row * nchunk;
    const int c0 = chunk * WCH;
    const int b = row / C, c = row - b * C;
    const int t = c0 + 4 * lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t base = (((int64_t)b * F1 + 4 * wave + i) * C + c) * S;
      if (vec && t + 3 < S) {
        ry[i] = *reinterpret_cast<const float4*>(y1 + base + t);
        rg[i] = *reinterpret_cast<const float4*>(g1 + base + t);
      } else {
        float yv[4], gv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = t + e < S;
          yv[e] = ok ? y1[base + t + e] : 0.f;
          gv[e] = ok ? g1[base + t + e] : 0.f;
        }
        ry[i] = make_float4(yv[0], yv[1], yv[2], yv[3]);
        rg[i] = make_float4(gv[0], gv[1], gv[2], gv[3]);
      }
    }
    if (lane < 4) {                                     // the left halo sample dy[f, c0-1], f = 4*wave + lane
      const int64_t base = (((int64_t)b * F1 + 4 * wave + lane) * C + c) * S;
      const bool ok = c0 >= 1 && c0 - 1 < S;
      ryh = ok ? y1[base + c0 - 1] : 0.f;
      rgh = ok ? g1[base + c0 - 1] : 0.f;
    }
    const float* xrow = x + (int64_t)row * S;
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
      const int idx = 2 * (threadIdx.x + 128 * i);
      const int tx = c0 + idx - padl;
      rx[i].x = (idx < XW && tx >= 0 && tx < S) ? xrow[tx] : 0.f;
      rx[i].y = (idx < XW && tx + 1 >= 0 && tx + 1 < S) ? xrow[tx + 1] : 0.f;
    }
  };
  auto commit = [&](int work) {
    const int chunk = work % nchunk;
    const int c0 = chunk * WCH;
    const int t = c0 + 4 * lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = 4 * wave + i;
      const float a0 = aff[f], a1 = aff[F1 + f], a2 = aff[2 * F1 + f];
      const float yv[4] = {ry[i].x, ry[i].y, ry[i].z, ry[i].w};
      const float gv[4] = {rg[i].x, rg[i].y, rg[i].z, rg[i].w};
      h4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        _Float16 ph, pl;
        split2(t + e < S ? fmaf(a0, gv[e], fmaf(a1, yv[e], a2)) : 0.f, ph, pl);
        hi[e] = ph;
        lo[e] = pl;
      }
      *reinterpret_cast<h4*>(&dyh[f * DYS + 8 + 4 * lane]) = hi;
      *reinterpret_cast<h4*>(&dyl[f * DYS + 8 + 4 * lane]) = lo;
    }
    if (lane < 4) {
      const int f = 4 * wave + lane;
      _Float16 ph, pl;
      split2((c0 >= 1 && c0 - 1 < S) ? fmaf(aff[f], rgh, fmaf(aff[F1 + f], ryh, aff[2 * F1 + f])) : 0.f, ph, pl);
      dyh[f * DYS + 7] = ph;
      dyl[f * DYS + 7] = pl;
    }
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
      const int idx = 2 * (threadIdx.x + 128 * i);
      if (idx < XW) {
        _Float16 h0, l0, h1, l1;
        split2(sigx * rx[i].x, h0, l0);
        split2(sigx * rx[i].y, h1, l1);
        *reinterpret_cast<uint32_t*>(&xh[idx]) = pack2(h0, h1);
        *reinterpret_cast<uint32_t*>(&xl[idx]) = pack2(l0, l1);
      }
    }
  };
  if ((int)blockIdx.x < nwork) fetch(blockIdx.x);
  __syncthreads();                                      // aff[] visible
  const uint32_t* xhw = reinterpret_cast<const uint32_t*>(xh) + 4 * g + col + 16 * nt0;
  const uint32_t* xlw = reinterpret_cast<const uint32_t*>(xl) + 4 * g + col + 16 * nt0;
  for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
    commit(work);
    __syncthreads();
    if (work + (int)gridDim.x < nwork) fetch(work + gridDim.x);
    // sliding operand window: slot j holds the x operand of index m = ks + j; per K-step the window moves down by
    // one slot (register moves, hidden under the MFMAs) and ONE new operand pair is read.  The loop is kept rolled so
    // that the live set stays at window + accumulators (a full unroll lets the scheduler hoist all 12 operand pairs).
    h8 bh[TPW], bl[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      if (j < ntn) {
        bh[j] = as_h8(xhw[16 * j], xhw[16 * j + 1], xhw[16 * j + 2], xhw[16 * j + 3]);
        bl[j] = as_h8(xlw[16 * j], xlw[16 * j + 1], xlw[16 * j + 2], xlw[16 * j + 3]);
      }
    }
    const _Float16* ah_p = dyh + af * DYS + 8 + 8 * g;
    const _Float16* al_p = dyl + af * DYS + 8 + 8 * g;
#pragma unroll 1
    for (int ks = 0; ks < WKS; ++ks) {
      // A pieces of this K-step: 8 halfs at row[8 + 32 ks + 8 g - s]
      const uint4 dh = *reinterpret_cast<const uint4*>(ah_p + 32 * ks);
      const uint4 dl = *reinterpret_cast<const uint4*>(al_p + 32 * ks);
      const uint32_t ph = *reinterpret_cast<const uint32_t*>(ah_p + 32 * ks - 2);
      const uint32_t pl = *reinterpret_cast<const uint32_t*>(al_p + 32 * ks - 2);
      const h8 ahi = as ? as_h8(__builtin_amdgcn_alignbit(dh.x, ph, 16), __builtin_amdgcn_alignbit(dh.y, dh.x, 16),
                                __builtin_amdgcn_alignbit(dh.z, dh.y, 16), __builtin_amdgcn_alignbit(dh.w, dh.z, 16))
                           : as_h8(dh.x, dh.y, dh.z, dh.w);
      const h8 alo = as ? as_h8(__builtin_amdgcn_alignbit(dl.x, pl, 16), __builtin_amdgcn_alignbit(dl.y, dl.x, 16),
                                __builtin_amdgcn_alignbit(dl.z, dl.y, 16), __builtin_amdgcn_alignbit(dl.w, dl.z, 16))
                           : as_h8(dl.x, dl.y, dl.z, dl.w);
      // the operand the NEXT K-step will need in its last slot travels while this step's MFMAs run
      const int wn = 16 * (ks + ntn);
      const h8 nh = as_h8(xhw[wn], xhw[wn + 1], xhw[wn + 2], xhw[wn + 3]);
      const h8 nl = as_h8(xlw[wn], xlw[wn + 1], xlw[wn + 2], xlw[wn + 3]);
      // three passes over the wave's tiles, so that the two MFMAs into c2[j] are TPW instructions apart
#pragma unroll
      for (int j = 0; j < TPW; ++j)
        if (j < ntn) c1[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bh[j], c1[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TPW; ++j)
        if (j < ntn) c2[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bh[j], c2[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TPW; ++j)
        if (j < ntn) c2[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bl[j], c2[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j + 1 < TPW; ++j) {
        if (j + 1 < ntn) { bh[j] = bh[j + 1]; bl[j] = bl[j + 1]; }
      }
#pragma unroll
      for (int j = 0; j < TPW; ++j)
        if (j == ntn - 1) { bh[j] = nh; bl[j] = nl; }
    }
    __syncthreads();                                    // both waves are done reading the image
  }
  // C layout 16x16: col = lane&15, row = (lane>>4)*4 + reg = f*2 + s.  The two waves own disjoint lags.
  const float post = sx[1] * sdy[1];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    if (j < ntn) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = g * 4 + r;
        const int k = 2 * (16 * (nt0 + j) + col) + (i & 1);
        if (k < klen) part[(int64_t)blockIdx.x * (F1 * klen) + (i >> 1) * klen + k] = post * fmaf(LO_INV, c2[j][r], c1[j][r]);
      }
    }
  }
}

}  // namespace

static int fir_grid(int nwork) { return nwork < 512 ? nwork : 512; }

// floats the caller provides in `part` (zero-initialised ONCE; part[0] is the kernel's self-resetting counter)
static int absmax_blocks(int64_t n) {
  int64_t b = (n + 256 * 16 - 1) / (256 * 16);
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
extern "C" int eav_absmax_scale_nparts(int64_t n) { return absmax_blocks(n) + 1; }

// scale[0] = sigma (power of two with |sigma v| <= 2^12 for every element, allowing for the factor `extra` >= 1 the
// caller expects on top of the tensor's own maximum), scale[1] = 1 / sigma, scale[2] = max.  part:
// eav_absmax_scale_nparts(n) floats, ZERO-INITIALISED once by the caller (part[0] is a self-resetting counter).
extern "C" int eav_absmax_scale(const float* v, int64_t n, float extra, float* part, float* scale, void* stream) {
  EAV_REQUIRE(v && part && scale && n > 0 && extra >= 1.f, "eav_absmax_scale: bad arguments");
  hipLaunchKernelGGL(absmax_scale_kernel, dim3(absmax_blocks(n)), dim3(256), 0, (hipStream_t)stream, v, n, extra, part,
                     scale);
  EAV_CHECK_LAUNCH("eav_absmax_scale");
  return EAV_OK;
}

extern "C" int eav_eegnet_fir_fwd_split(const float* x, const float* w1, const float* scale_x, const float* scale_w,
                                        float* y1, float* stat_part, int B, int C, int S, int klen, void* stream) {
  EAV_REQUIRE(x && w1 && scale_x && scale_w && y1 && stat_part && B > 0 && C > 0 && S > 0,
              "eav_eegnet_fir_fwd_split: bad arguments");
  EAV_REQUIRE(klen >= 1 && klen <= 300, "eav_eegnet_fir_fwd_split: kernLength %d outside [1,300]", klen);
  const int ntiles = cdiv(S, TILE), nseg = cdiv(ntiles, TPS);
  const int nwork = B * C * nseg;
#define EAV_FIRS(NKS)                                                                                              \
  hipLaunchKernelGGL(fir_fwd_split_kernel<NKS>, dim3(fir_grid(nwork)), dim3(256), 0, (hipStream_t)stream, x, w1,   \
                     scale_x, scale_w, y1, stat_part, B * C, C, S, klen, (klen - 1) / 2, nseg, ntiles)
  if (klen <= 77) EAV_FIRS(5);
  else if (klen <= 141) EAV_FIRS(9);
  else EAV_FIRS(19);
#undef EAV_FIRS
  EAV_CHECK_LAUNCH("eav_eegnet_fir_fwd_split");
  return EAV_OK;
}

extern "C" int eav_absmax_finish(const float* part, int nparts, float extra, float* scale, void* stream) {
  EAV_REQUIRE(part && nparts > 0 && extra >= 1.f && scale, "eav_absmax_finish: bad arguments");
  hipLaunchKernelGGL(absmax_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nparts, extra, scale);
  EAV_CHECK_LAUNCH("eav_absmax_finish");
  return EAV_OK;
}

extern "C" int eav_fir_dy_scale(const float* bn_params, const float* dzmax_part, int nparts, const float* w2, int C,
                                float* out, void* stream) {
  EAV_REQUIRE(bn_params && dzmax_part && nparts > 0 && w2 && C > 0 && out, "eav_fir_dy_scale: bad arguments");
  hipLaunchKernelGGL(dy_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, bn_params, dzmax_part, nparts, w2, C,
                     out);
  EAV_CHECK_LAUNCH("eav_fir_dy_scale");
  return EAV_OK;
}

static int wgrad_split_grid(int64_t nwork) { return nwork < 1536 ? (int)nwork : 1536; }

extern "C" int eav_eegnet_fir_wgrad_split_nparts(int B, int C, int S) {
  return wgrad_split_grid((int64_t)B * C * cdiv(S + 1, WCH));
}

extern "C" int eav_eegnet_fir_wgrad_split(const float* x, const float* y1, const float* g1, const float* bn_params,
                                          const float* scale_x, const float* scale_dy, float* part, int B, int C, int S,
                                          int klen, void* stream) {
  EAV_REQUIRE(x && y1 && g1 && bn_params && scale_x && scale_dy && part && B > 0 && C > 0 && S > 0,
              "eav_eegnet_fir_wgrad_split: bad arguments");
  EAV_REQUIRE(klen >= 1 && klen <= 300, "eav_eegnet_fir_wgrad_split: kernLength %d outside [1,300]", klen);
  const int nchunk = cdiv(S + 1, WCH), nwork = B * C * nchunk;
#define EAV_FWS(NT)                                                                                               \
  hipLaunchKernelGGL(fir_wgrad_split_kernel<NT>, dim3(wgrad_split_grid(nwork)), dim3(128), 0, (hipStream_t)stream, \
                     x, y1, g1, bn_params, scale_x, scale_dy, part, B * C, C, S, klen, (klen - 1) / 2, nchunk)
  if (klen <= 64) EAV_FWS(2);
  else if (klen <= 192) EAV_FWS(6);
  else EAV_FWS(10);
#undef EAV_FWS
  EAV_CHECK_LAUNCH("eav_eegnet_fir_wgrad_split");
  return EAV_OK;
}
