// Row-wise / element-wise kernels of the AST and ViT encoders (fp32, HBM-bound, 16 B per lane):
// LayerNorm fwd/bwd (eps 1e-12, HF config), softmax fwd/bwd over attention rows, erf-GELU backward,
// bias-gradient column sums, patch im2col, token/position embedding fwd/bwd, token-row gather/scatter.
// Reference arithmetic: Hugging Face modeling_audio_spectrogram_transformer.py / modeling_vit.py as
// instantiated at Transformer_Audio.py:22 and Transformer_Vision.py:29 (see oracle/vit_oracle.py).
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

constexpr int LN_MAXQ = 4;  // float4 per lane: D <= 1024

// ---------------------------------------------------------------------------------- LayerNorm
// one wave per row; two-pass statistics in registers (mean, then sum (x-mean)^2), biased variance
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                            int M, int D, float eps, unsigned* __restrict__ amax,
                                                            unsigned char* __restrict__ planes,
                                                            const float* __restrict__ pslot, int64_t ldp, float lomul) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nq = D >> 2;
  const float4* src = reinterpret_cast<const float4*>(x + (int64_t)row * D);
  float4 v[LN_MAXQ];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    const int q = lane + 64 * i;
    v[i] = q < nq ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    const int q = lane + 64 * i;
    if (q < nq) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      ss += (a * a + b * b) + (c * c + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)D + eps);
  float4* dst = reinterpret_cast<float4*>(y + (int64_t)row * D);
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  const float4* b4 = reinterpret_cast<const float4*>(beta);
  float vmax = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    const int q = lane + 64 * i;
    if (q < nq) {
      const float4 g = g4[q], b = b4[q];
      const float4 o = make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                                   (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
      if (y) dst[q] = o;
      vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
      if (planes) {
        // row planes of the split-operand GEMMs straight from the registers (D % 8 == 0: lanes 2p, 2p + 1 hold the two
        // halves of the group of 8 columns p; they swap halves so that each stores one whole 16-byte piece - hi / lo).
        // The scale is a bound of |y| known before the launch (eav_tf_forward_scales), not a measured maximum.
        const float sg = pslot[EAV_SLOT_SIGMA];
        const float t[4] = {o.x * sg, o.y * sg, o.z * sg, o.w * sg};
        _Float16 h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h[e] = (_Float16)t[e];
          l[e] = (_Float16)((t[e] - (float)h[e]) * lomul);
        }
        const uint2 hh = *reinterpret_cast<const uint2*>(h), ll = *reinterpret_cast<const uint2*>(l);
        const bool first = (q & 1) == 0;
        const uint2 send = first ? ll : hh;
        uint2 recv;
        recv.x = __shfl_xor(send.x, 1, 64);
        recv.y = __shfl_xor(send.y, 1, 64);
        *reinterpret_cast<uint4*>(planes + (int64_t)row * ldp + (int64_t)(q >> 1) * 32 + (first ? 0 : 16)) =
            first ? make_uint4(hh.x, hh.y, recv.x, recv.y) : make_uint4(recv.x, recv.y, ll.x, ll.y);
      }
    }
  }
  if (lane == 0) {
    if (mean_o) mean_o[row] = mean;
    if (rstd_o) rstd_o[row] = rstd;
    // max over the rows of rstd, for the bound of the backward's output (eav_layernorm_bwd_bound): word 1 of the row's shard
    // line in whichever scale slot this launch has (the line is this row's atomic target anyway / unused in a-priori mode)
    unsigned* rs = amax ? amax : reinterpret_cast<unsigned*>(const_cast<float*>(pslot));
    if (rs && rstd == rstd) atomicMax(rs + EAV_SLOT_SHARD(row) + 1, __float_as_uint(rstd));
  }
  if (amax) {   // max |y| of the row into one of the 64 shards of the operand-scale slot (gemm_sp.hip)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if (lane == 0 && vmax == vmax) atomicMax(amax + EAV_SLOT_SHARD(row), __float_as_uint(vmax));
  }
}

__device__ __forceinline__ float sigma_of_bound(float b) {
  // the power of two that puts b in [2^14, 2^15) (as sigma_from_bits in gemm_sp.hip); 1 for 0 / non-finite
  const unsigned bits = __float_as_uint(b);
  const int e = (int)((bits >> 23) & 0xff);
  if (b <= 0.f || e == 0xff) return 1.f;
  int se = 14 - (e - 127);
  se = max(-126, min(126, se));
  return __uint_as_float((unsigned)(se + 127) << 23);
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  dgamma += dy*xhat, dbeta += dy per column.
// Persistent blocks; part[blk][2*D] = (dgamma, dbeta) partials of the rows this block handled.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_i,
                                                            const float* __restrict__ rstd_i, float* __restrict__ dx,
                                                            int accumulate, float* __restrict__ part, int M, int D,
                                                            unsigned* __restrict__ amax, unsigned char* __restrict__ planes,
                                                            int64_t ldp, float lomul, const float* __restrict__ slot_old,
                                                            const float* __restrict__ slot_dy,
                                                            const float* __restrict__ slot_rstd) {
  // planes (optional): the STORED value (dx, after the accumulation) also leaves as row planes scaled by the sigma in the
  // amax slot - a bound set before the launch (eav_layernorm_bwd_bound) - and its column sums (the bias gradient of the
  // linear layer that produced the residual branch) as a third section of the partials: part[blk][3*D]
  extern __shared__ float sh[];  // [4][2*D] (planes: [4][3*D])
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nq = D >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(gamma);
  float4 gm[LN_MAXQ], ag[LN_MAXQ], ab[LN_MAXQ];
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    const int q = lane + 64 * i;
    gm[i] = q < nq ? g4[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float4 ac[LN_MAXQ];              // planes: column sums of the stored value
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) ac[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // planes: the scale.  slot_dy given: the bound of eav_layernorm_bwd_bound formed HERE, by every wave from the same numbers
  // (max|gamma| over the lanes' own gains, the three slots' shard words - bit-identical in every wave), and published by
  // block 0 for the consumers; a separate one-block launch waited 15-40 us per call for a CU between the persistent GEMMs
  // of the two streams.  slot_dy NULL: the sigma already in the slot.
  float psg = 0.f;
  if (planes) {
    if (slot_dy) {
      float gmx = 0.f;
#pragma unroll
      for (int i = 0; i < LN_MAXQ; ++i)
        gmx = fmaxf(gmx, fmaxf(fmaxf(fabsf(gm[i].x), fabsf(gm[i].y)), fmaxf(fabsf(gm[i].z), fabsf(gm[i].w))));
      float rmx = slot_rstd[32 * lane + 1];
      float old = slot_old ? slot_old[32 * lane] : 0.f, dmx = slot_dy[32 * lane];      // (non-negative floats: bit order = value order)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        gmx = fmaxf(gmx, __shfl_xor(gmx, o, 64));
        rmx = fmaxf(rmx, __shfl_xor(rmx, o, 64));
        old = fmaxf(old, __shfl_xor(old, o, 64));
        dmx = fmaxf(dmx, __shfl_xor(dmx, o, 64));
      }
      if (rmx == 0.f) {      // the slot was not written by the matching forward LayerNorm (rstd > 0 always): never trust a
                             // zero - walk the saved rstd instead (slow path, the same value in every wave)
        for (int i = lane; i < M; i += 64) rmx = fmaxf(rmx, rstd_i[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) rmx = fmaxf(rmx, __shfl_xor(rmx, o, 64));
      }
      const float bound = (old + (2.f + sqrtf((float)D)) * gmx * rmx * dmx) * 1.0001f;
      psg = sigma_of_bound(bound);
      if (blockIdx.x == 0 && threadIdx.x == 0) {
        reinterpret_cast<float*>(amax)[EAV_SLOT_SIGMA] = psg;
        reinterpret_cast<float*>(amax)[EAV_SLOT_ISIGMA] = 1.f / psg;
      }
    } else {
      psg = reinterpret_cast<const float*>(amax)[EAV_SLOT_SIGMA];
    }
  }
  float vmax = 0.f;
  // the rows of a wave are independent: the NEXT row's x, dy (and dx when accumulating) are loaded before the current
  // row's reductions, so a wave always has a row in flight (in the encoder step this kernel runs beside a weight-gradient
  // GEMM on the side stream, where an un-prefetched row costs a fully loaded memory system's latency each time)
  float4 xn[LN_MAXQ], dn[LN_MAXQ], pn[LN_MAXQ];
  float mean_n = 0.f, rstd_n = 0.f;
  auto fetch = [&](int row) {
    const float4* xs = reinterpret_cast<const float4*>(x + (int64_t)row * D);
    const float4* ds = reinterpret_cast<const float4*>(dy + (int64_t)row * D);
    const float4* ps = reinterpret_cast<const float4*>(dx + (int64_t)row * D);
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      if (q < nq) {
        xn[i] = xs[q];
        dn[i] = ds[q];
        if (accumulate) pn[i] = ps[q];
      }
    }
    mean_n = mean_i[row];
    rstd_n = rstd_i[row];
  };
  // A wave takes the rows in groups of 16 consecutive ones (group g = gw, gw + GW, ...): a group lies inside ONE 32-row
  // block, so the block maximum (EAV_SLOT_BMAX) costs one wave reduction and one atomic per 16 rows - one atomic per ROW
  // to ~200 addresses doubled the kernel's time (ViT B=128: 95 us against 46 us without the maxima)
#ifndef EAV_LN_RG
#define EAV_LN_RG 16
#endif
  constexpr int RG = EAV_LN_RG;
  const int GW = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
  auto next_row = [&](int r) { return ((r + 1) & (RG - 1)) ? r + 1 : r + 1 + (GW - 1) * RG; };
  int row = gw * RG;
  float gmax = 0.f;           // per lane: maximum over the rows of the current group
  if (row < M) fetch(row);
  for (; row < M; row = next_row(row)) {
    const float mean = mean_n, rstd = rstd_n;
    float4 xh[LN_MAXQ], gd[LN_MAXQ], pv[LN_MAXQ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      if (q < nq) {
        const float4 xv = xn[i], dv = dn[i];
        pv[i] = pn[i];
        xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        gd[i] = make_float4(dv.x * gm[i].x, dv.y * gm[i].y, dv.z * gm[i].z, dv.w * gm[i].w);
        ag[i].x += dv.x * xh[i].x; ag[i].y += dv.y * xh[i].y; ag[i].z += dv.z * xh[i].z; ag[i].w += dv.w * xh[i].w;
        ab[i].x += dv.x; ab[i].y += dv.y; ab[i].z += dv.z; ab[i].w += dv.w;
        s1 += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
        s2 += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
      } else {
        xh[i] = gd[i] = pv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const int next = next_row(row);
    if (next < M) fetch(next);
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
    float rmax = 0.f;
    float4* dst = reinterpret_cast<float4*>(dx + (int64_t)row * D);
#pragma unroll
    for (int i = 0; i < LN_MAXQ; ++i) {
      const int q = lane + 64 * i;
      if (q < nq) {
        float4 o = make_float4(rstd * (gd[i].x - m1 - xh[i].x * m2), rstd * (gd[i].y - m1 - xh[i].y * m2),
                               rstd * (gd[i].z - m1 - xh[i].z * m2), rstd * (gd[i].w - m1 - xh[i].w * m2));
        if (accumulate) {
          o.x += pv[i].x; o.y += pv[i].y; o.z += pv[i].z; o.w += pv[i].w;
        }
        dst[q] = o;
        rmax = fmaxf(rmax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
        if (planes) {     // (as layernorm_fwd_kernel: lanes 2p, 2p + 1 swap halves, each stores one whole 16-byte piece)
          ac[i].x += o.x; ac[i].y += o.y; ac[i].z += o.z; ac[i].w += o.w;
          const float t[4] = {o.x * psg, o.y * psg, o.z * psg, o.w * psg};
          _Float16 h[4], l[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            h[e] = (_Float16)t[e];
            l[e] = (_Float16)((t[e] - (float)h[e]) * lomul);
          }
          const uint2 hh = *reinterpret_cast<const uint2*>(h), ll = *reinterpret_cast<const uint2*>(l);
          const bool first = (q & 1) == 0;
          const uint2 send = first ? ll : hh;
          uint2 recv;
          recv.x = __shfl_xor(send.x, 1, 64);
          recv.y = __shfl_xor(send.y, 1, 64);
          *reinterpret_cast<uint4*>(planes + (int64_t)row * ldp + (int64_t)(q >> 1) * 32 + (first ? 0 : 16)) =
              first ? make_uint4(hh.x, hh.y, recv.x, recv.y) : make_uint4(recv.x, recv.y, ll.x, ll.y);
        }
      }
    }
    if (amax && !planes) {       // the group's maximum into its 32-row block's entry (per-row-block operand scales, eav_common.h)
      gmax = fmaxf(gmax, rmax);
      if (((row + 1) & (RG - 1)) == 0 || row + 1 >= M) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
        if (lane == 0) eav_slot_blockmax(amax, row, gmax);
        vmax = fmaxf(vmax, gmax);
        gmax = 0.f;
      }
    }
    if (planes) vmax = fmaxf(vmax, rmax);      // (a-priori scale: no block entries, the tensor-wide measured maximum only -
                                               // the next bound starts from it)
  }
  if (amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if (lane == 0 && vmax == vmax) atomicMax(amax + EAV_SLOT_SHARD(blockIdx.x * 4 + wave), __float_as_uint(vmax));
  }
  if (!part) return;
  const int NS = planes ? 3 : 2;
  float4* shw = reinterpret_cast<float4*>(sh + wave * NS * D);
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    const int q = lane + 64 * i;
    if (q < nq) {
      shw[q] = ag[i];
      shw[nq + q] = ab[i];
      if (planes) shw[2 * nq + q] = ac[i];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NS * D; i += 256)
    part[(int64_t)blockIdx.x * NS * D + i] = (sh[i] + sh[NS * D + i]) + (sh[2 * NS * D + i] + sh[3 * NS * D + i]);
}

// ------------------------------------------------------------------------------------ softmax
// in place over rows of length N (leading dimension ld); one wave per row, row held in registers
template <int NPL>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(float* __restrict__ s, int64_t rows, int N, int ld) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* p = s + row * ld;
  float v[NPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < N ? p[c] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    v[i] = (lane + 64 * i) < N ? expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  const float inv = 1.0f / wave_sum(sum);
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int c = lane + 64 * i;
    if (c < N) p[c] = v[i] * inv;
  }
}

// dS = P o (dP - sum_k dP*P), written in place over dP
template <int NPL>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP,
                                                          int64_t rows, int N, int ld) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = P + row * ld;
  float* d = dP + row * ld;
  float pv[NPL], dv[NPL];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int c = lane + 64 * i;
    pv[i] = c < N ? p[c] : 0.f;
    dv[i] = c < N ? d[c] : 0.f;
    dot += pv[i] * dv[i];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < NPL; ++i) {
    const int c = lane + 64 * i;
    if (c < N) d[c] = pv[i] * (dv[i] - dot);
  }
}

// ------------------------------------------------------------------------------- GELU backward
// d/dx [0.5 x (1 + erf(x/sqrt2))] = 0.5 (1 + erf(x/sqrt2)) + x exp(-x^2/2)/sqrt(2 pi);  in place on dact
__global__ __launch_bounds__(256) void gelu_bwd_kernel(float* __restrict__ dact, const float* __restrict__ pre,
                                                       int64_t n4, unsigned* __restrict__ amax) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  float vmax = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 d = reinterpret_cast<float4*>(dact)[i];
    const float4 x = reinterpret_cast<const float4*>(pre)[i];
    auto gp = [](float v) {
      return 0.5f * (1.0f + erff(v * 0.70710678118654752f)) + v * expf(-0.5f * v * v) * 0.3989422804014327f;
    };
    d.x *= gp(x.x); d.y *= gp(x.y); d.z *= gp(x.z); d.w *= gp(x.w);
    reinterpret_cast<float4*>(dact)[i] = d;
    vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w))));
  }
  if (amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if ((threadIdx.x & 63) == 0 && vmax == vmax)
      atomicMax(amax + EAV_SLOT_SHARD(blockIdx.x * 4 + (threadIdx.x >> 6)), __float_as_uint(vmax));
  }
}

// --------------------------------------------------------------------------- column sums (bias grad)
// grid (ceil(N/64), nslab): part[slab][n] = sum over the slab's rows of dy[r][n]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, float* __restrict__ part, int M,
                                                     int N, int ld, int rows_per_slab) {
  __shared__ float sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rs = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_slab, r1 = min(M, r0 + rows_per_slab);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rs; r < r1; r += 4) s += dy[(int64_t)r * ld + c];
  sh[rs][threadIdx.x & 63] = s;
  __syncthreads();
  if (threadIdx.x < 64 && c < N)
    part[(int64_t)blockIdx.y * N + c] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------ im2col
// col[(b, py, px), (c, i, j)] = x[b, c, py*sy + i, px*sx + j] for a [C, H, W] image with strides;
// the AST view (HF :57-59) is C=1, H=mel, W=frames of the transposed input: x[b, frame, mel], i.e.
// element (h, w) lives at b*H*W + w*H + h  ->  `transposed` = 1.
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int C,
                                                     int H, int W, int P, int sy, int sx, int ny, int nx,
                                                     int transposed) {
  const int KP = C * P * P;
  const int64_t total = (int64_t)B * ny * nx * KP;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    const int k = (int)(idx % KP);
    const int64_t r = idx / KP;
    const int px = (int)(r % nx), py = (int)((r / nx) % ny), b = (int)(r / ((int64_t)nx * ny));
    const int j = k % P, i = (k / P) % P, c = k / (P * P);
    const int h = py * sy + i, w = px * sx + j;
    const int64_t src = transposed ? ((int64_t)b * C + c) * H * W + (int64_t)w * H + h
                                   : (((int64_t)b * C + c) * H + h) * W + w;
    col[idx] = x[src];
  }
}

// h[b,t,:] = (t < nextra ? tokens[t,:] : h[b,t,:]) + pos[t,:]
__global__ __launch_bounds__(256) void embed_finish_kernel(float* __restrict__ h, const float* __restrict__ cls,
                                                           const float* __restrict__ dist,
                                                           const float* __restrict__ pos, int B, int ntok, int D,
                                                           int nextra) {
  const int64_t total = (int64_t)B * ntok * (D >> 2);
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    const int q = (int)(idx % (D >> 2));
    const int t = (int)((idx / (D >> 2)) % ntok);
    float4 v;
    if (t < nextra) v = reinterpret_cast<const float4*>(t == 0 ? cls : dist)[q];
    else v = reinterpret_cast<float4*>(h)[idx];
    const float4 p = reinterpret_cast<const float4*>(pos)[(int64_t)t * (D >> 2) + q];
    reinterpret_cast<float4*>(h)[idx] = make_float4(v.x + p.x, v.y + p.y, v.z + p.z, v.w + p.w);
  }
}

// dpos[t,:] = sum_b dh[b,t,:];  demb[b*np + p,:] = dh[b, nextra+p, :]  (contiguous rows for the patch GEMM)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dh, float* __restrict__ dpos,
                                                        float* __restrict__ demb, int B, int ntok, int D, int nextra) {
  const int64_t total = (int64_t)ntok * (D >> 2);
  const int np = ntok - nextra;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int q = (int)(idx % (D >> 2)), t = (int)(idx / (D >> 2));
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < B; ++b) {
      const float4 v = reinterpret_cast<const float4*>(dh)[((int64_t)b * ntok + t) * (D >> 2) + q];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      if (t >= nextra) reinterpret_cast<float4*>(demb)[((int64_t)b * np + (t - nextra)) * (D >> 2) + q] = v;
    }
    reinterpret_cast<float4*>(dpos)[idx] = s;
  }
}

// out[b*nextra + e, :] = h[b, e, :] (gather, dir = 0)  /  h[b, e, :] = out[...] (scatter, dir = 1)
__global__ __launch_bounds__(256) void token_rows_kernel(float* __restrict__ h, float* __restrict__ rows, int B,
                                                         int ntok, int D, int nextra, int dir) {
  const int64_t total = (int64_t)B * nextra * (D >> 2);
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int q = (int)(idx % (D >> 2));
    const int e = (int)((idx / (D >> 2)) % nextra), b = (int)(idx / ((int64_t)(D >> 2) * nextra));
    float4* hp = reinterpret_cast<float4*>(h) + ((int64_t)b * ntok + e) * (D >> 2) + q;
    float4* rp = reinterpret_cast<float4*>(rows) + idx;
    if (dir == 0) *rp = *hp; else *hp = *rp;
  }
}

// AST pooled output (cls + dist)/2 (HF AST :304): fwd pooled[b] = (seq[2b] + seq[2b+1])/2; bwd dseq[2b] = dseq[2b+1] = dpooled[b]/2
__global__ __launch_bounds__(256) void pair_mean_kernel(float* __restrict__ seq, float* __restrict__ pooled, int B,
                                                        int D, int dir) {
  const int64_t total = (int64_t)B * D;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int d = (int)(idx % D), b = (int)(idx / D);
    if (dir == 0) {
      pooled[idx] = (seq[(int64_t)(2 * b) * D + d] + seq[(int64_t)(2 * b + 1) * D + d]) * 0.5f;
    } else {
      const float v = pooled[idx] * 0.5f;
      seq[(int64_t)(2 * b) * D + d] = v;
      seq[(int64_t)(2 * b + 1) * D + d] = v;
    }
  }
}

int grid_for(int64_t n) { return (int)(n < 1 ? 1 : (cdiv64(n, 256) > 4096 ? 4096 : cdiv64(n, 256))); }

// ------------------------------------------------------------------------------------ a-priori operand scales
// max over the rows of ||w_r||_2 (bits, atomicMax - zero `out` first): one wave per row at a time, 16-byte loads, all
// loads of a row in flight at once for rows of up to 1024 floats; a few persistent blocks and ONE atomic per block (one
// atomic per row to the same address serialised in the L2: 34 us for 3072 rows)
__device__ __forceinline__ void rownorm_max_body(const float* __restrict__ w, int R, int C, int64_t ld,
                                                 unsigned* __restrict__ out, int bx, int nbx) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool vec = (C & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)w & 15) == 0;
  float best = 0.f;
  for (int r = bx * 4 + wave; r < R; r += nbx * 4) {
    const float* src = w + (int64_t)r * ld;
    float s = 0.f;
    if (vec) {
      const float4* s4 = reinterpret_cast<const float4*>(src);
      const int n4 = C >> 2;
      for (int c0 = 0; c0 < n4; c0 += 256) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c0 + lane + 64 * u;
          v[u] = c < n4 ? s4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s = fmaf(v[u].x, v[u].x, fmaf(v[u].y, v[u].y, fmaf(v[u].z, v[u].z, fmaf(v[u].w, v[u].w, s))));
      }
    } else {
      for (int c = lane; c < C; c += 64) s = fmaf(src[c], src[c], s);
    }
    best = fmaxf(best, sqrtf(wave_sum(s)));
  }
  if (lane == 0) red[wave] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    best = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (best == best) atomicMax(out, __float_as_uint(best));
  }
}
__global__ __launch_bounds__(256) void rownorm_max_kernel(const float* __restrict__ w, int R, int C, int64_t ld,
                                                          unsigned* __restrict__ out) {
  rownorm_max_body(w, R, C, ld, out, blockIdx.x, gridDim.x);
}

// max over the COLUMNS of ||w[:, c]||_2 (bits, atomicMax - zero `out` first).  A block owns 64 columns: thread (rs, cq)
// sums the squares of columns 4 cq .. + 3 over the rows rs, rs + 16, ... (a row segment = 256 contiguous bytes per 16
// threads), the 16 row slices are added through LDS, one atomic per block.
__device__ __forceinline__ void colnorm_max_body(const float* __restrict__ w, int R, int C, int64_t ld,
                                                 unsigned* __restrict__ out, int bx) {
  __shared__ float4 part[16][17];
  const int cq = threadIdx.x & 15, rs = threadIdx.x >> 4;
  const int c = bx * 64 + 4 * cq;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C) {
    int r = rs;
    for (; r + 48 < R; r += 64) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(w + (int64_t)(r + 16 * u) * ld + c);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s.x = fmaf(v[u].x, v[u].x, s.x); s.y = fmaf(v[u].y, v[u].y, s.y);
        s.z = fmaf(v[u].z, v[u].z, s.z); s.w = fmaf(v[u].w, v[u].w, s.w);
      }
    }
    for (; r < R; r += 16) {
      const float4 v = *reinterpret_cast<const float4*>(w + (int64_t)r * ld + c);
      s.x = fmaf(v.x, v.x, s.x); s.y = fmaf(v.y, v.y, s.y); s.z = fmaf(v.z, v.z, s.z); s.w = fmaf(v.w, v.w, s.w);
    }
  }
  part[rs][cq] = s;
  __syncthreads();
  if (threadIdx.x < 16) {
    float4 t = part[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 q = part[k][threadIdx.x];
      t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
    }
    float best = sqrtf(fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)));
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
    if (threadIdx.x == 0 && best == best) atomicMax(out, __float_as_uint(best));
  }
}
__global__ __launch_bounds__(256) void colnorm_max_kernel(const float* __restrict__ w, int R, int C, int64_t ld,
                                                          unsigned* __restrict__ out) {
  colnorm_max_body(w, R, C, ld, out, blockIdx.x);
}

// A table of row / column norm maxima in ONE launch (eav_norm_max_multi): job y = blockIdx.y, its blocks blockIdx.x <
// (row job: min(ceil(R / 4), 128); column job: ceil(C / 64)).  The weight refresh after an optimiser step issued 36 such
// launches of ~10 us on the side stream before the forward's a-priori scales could be formed.
struct NormJob { const float* w; int64_t ld; unsigned* out; int R, C, cols, pad; };
__global__ __launch_bounds__(256) void norm_max_multi_kernel(const NormJob* __restrict__ jobs) {
  const NormJob j = jobs[blockIdx.y];
  if (j.cols) {
    if ((int)blockIdx.x < (j.C + 63) / 64) colnorm_max_body(j.w, j.R, j.C, j.ld, j.out, blockIdx.x);
  } else {
    const int nb = min((j.R + 3) / 4, 128);
    if ((int)blockIdx.x < nb) rownorm_max_body(j.w, j.R, j.C, j.ld, j.out, blockIdx.x, nb);
  }
}


// Operand scales of one encoder layer's forward from RIGOROUS bounds of the tensors, so that their producers can emit
// the fp16 hi / lo planes directly (no measured maximum, no conversion pass).  With xhat the normalised row of a
// LayerNorm, |xhat_k| <= sqrt(D - 1) and ||xhat||_2 <= sqrt(D), hence for y = gamma xhat + beta:
//     |y_k| <= sqrt(D) max|gamma| + max|beta|,        ||y||_2 <= sqrt(D) max|gamma| + ||beta||_2,
// and for the MLP's hidden activation a = GELU(y2 W1^T + b1), |GELU(x)| <= |x|:
//     |a| <= ||y2||_2 max_n ||W1_n||_2 + max|b1|,
// and for the fused q/k/v projection of y1 (k_qkv >= 0): |qkv| <= ||y1||_2 max_n ||Wqkv_n||_2 + max|b_qkv|.
// The bounds overshoot the actual maxima by a factor ~sqrt(D) / 4-5 (3 bits at D = 768); the planes keep full precision
// for elements down to 2^-29 of the SCALE, so nothing is lost.  One block per layer; layer l's parameters start at
// p0 + l stride (the flat parameter buffer lays the layers out identically), its slots at slots + l slot_stride.
__global__ __launch_bounds__(256) void tf_forward_scales_kernel(const float* __restrict__ p0, int64_t stride, int off_g1,
                                                                int off_b1, int off_g2, int off_b2, int off_bfc1, int D,
                                                                int FF, const float* __restrict__ wnorm_fc1,
                                                                float* __restrict__ slots, int64_t slot_stride,
                                                                int k_y1, int k_y2, int k_act, int slot_floats,
                                                                int off_bqkv, const float* __restrict__ wnorm_qkv,
                                                                int k_qkv) {
  __shared__ float red[4 * 8];
  const int l = blockIdx.x;
  const float* p = p0 + (int64_t)l * stride;
  // max|g1| max|b1| max|g2| max|b2| sum b2^2 max|bfc1| sum b1^2 max|bqkv|
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < D; i += 256) {
    v[0] = fmaxf(v[0], fabsf(p[off_g1 + i]));
    v[1] = fmaxf(v[1], fabsf(p[off_b1 + i]));
    v[2] = fmaxf(v[2], fabsf(p[off_g2 + i]));
    v[3] = fmaxf(v[3], fabsf(p[off_b2 + i]));
    v[4] += p[off_b2 + i] * p[off_b2 + i];
    v[6] += p[off_b1 + i] * p[off_b1 + i];
  }
  for (int i = threadIdx.x; i < FF; i += 256) v[5] = fmaxf(v[5], fabsf(p[off_bfc1 + i]));
  if (k_qkv >= 0)
    for (int i = threadIdx.x; i < 3 * D; i += 256) v[7] = fmaxf(v[7], fabsf(p[off_bqkv + i]));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float a = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float b = __shfl_xor(a, o, 64);
      a = (k == 4 || k == 6) ? a + b : fmaxf(a, b);
    }
    if (lane == 0) red[wave * 8 + k] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      t[k] = (k == 4 || k == 6) ? (red[k] + red[8 + k]) + (red[16 + k] + red[24 + k])
                                : fmaxf(fmaxf(red[k], red[8 + k]), fmaxf(red[16 + k], red[24 + k]));
    const float sd = sqrtf((float)D);
    const float b_y1 = sd * t[0] + t[1], b_y2 = sd * t[2] + t[3];
    const float n_y2 = sd * t[2] + sqrtf(t[4]), n_y1 = sd * t[0] + sqrtf(t[6]);
    const float b_act = n_y2 * wnorm_fc1[l] + t[5];
    const float b_qkv = k_qkv >= 0 ? n_y1 * wnorm_qkv[l] + t[7] : 0.f;
    float* s = slots + (int64_t)l * slot_stride;
    const float bounds[4] = {b_y1, b_y2, b_act, b_qkv};
    const int ks[4] = {k_y1, k_y2, k_act, k_qkv};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (ks[k] < 0) continue;
      // 1.0001: the bound is evaluated in fp32 - keep it a bound under its own rounding
      const float sg = sigma_of_bound(bounds[k] * 1.0001f);
      s[(int64_t)ks[k] * slot_floats + EAV_SLOT_SIGMA] = sg;
      s[(int64_t)ks[k] * slot_floats + EAV_SLOT_ISIGMA] = 1.f / sg;
    }
  }
}

}  // namespace

extern "C" int eav_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                 float* rstd, int M, int D, float eps, void* stream) {
  EAV_REQUIRE(x && gamma && beta && y && M > 0 && D > 0 && (D & 3) == 0 && D <= 1024,
              "eav_layernorm_fwd: need D %% 4 == 0 and D <= 1024");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                     mean, rstd, M, D, eps, (unsigned*)nullptr, (unsigned char*)nullptr, (const float*)nullptr,
                     (int64_t)0, 0.f);
  EAV_CHECK_LAUNCH("eav_layernorm_fwd");
  return EAV_OK;
}

// the *_amax forms also accumulate max |output| into an operand-scale slot (EAV_SP_SLOT floats, zeroed by the caller) for
// the split-operand GEMM that consumes the output
extern "C" int eav_layernorm_fwd_amax(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                      float* rstd, int M, int D, float eps, float* amax_slot, void* stream) {
  EAV_REQUIRE(x && gamma && beta && y && amax_slot && M > 0 && D > 0 && (D & 3) == 0 && D <= 1024,
              "eav_layernorm_fwd_amax: need D %% 4 == 0 and D <= 1024");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                     mean, rstd, M, D, eps, reinterpret_cast<unsigned*>(amax_slot), (unsigned char*)nullptr,
                     (const float*)nullptr, (int64_t)0, 0.f);
  EAV_CHECK_LAUNCH("eav_layernorm_fwd_amax");
  return EAV_OK;
}

// LayerNorm whose output leaves as the row planes [M][Dp/8][2][8] of the split-operand GEMM that consumes it (no fp32
// copy, no conversion pass; y may be given as well).  scale_slot[EAV_SLOT_SIGMA] is a scale known before the launch
// (eav_tf_forward_scales); D % 8 == 0; the pad columns / rows of the planes are not written (allocate zero-filled).
extern "C" int eav_layernorm_fwd_planes(const float* x, const float* gamma, const float* beta, float* y, void* planes,
                                        const float* scale_slot, float* mean, float* rstd, int M, int D, float eps,
                                        void* stream) {
  EAV_REQUIRE(x && gamma && beta && planes && scale_slot && M > 0 && D > 0 && (D & 7) == 0 && D <= 1024 &&
                  ((uintptr_t)planes & 15) == 0,
              "eav_layernorm_fwd_planes: need D %% 8 == 0 and D <= 1024");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                     mean, rstd, M, D, eps, (unsigned*)nullptr, (unsigned char*)planes, scale_slot,
                     (int64_t)((D + 31) / 32 * 32) * 4, 2048.f);
  EAV_CHECK_LAUNCH("eav_layernorm_fwd_planes");
  return EAV_OK;
}

#ifndef EAV_LN_BLOCKS
#define EAV_LN_BLOCKS 512
#endif
extern "C" int eav_layernorm_bwd_nparts(int M) { return cdiv(M, 4) < EAV_LN_BLOCKS ? cdiv(M, 4) : EAV_LN_BLOCKS; }

extern "C" int eav_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                 const float* rstd, float* dx, int accumulate, float* part, int M, int D,
                                 void* stream) {
  EAV_REQUIRE(dy && x && gamma && mean && rstd && dx && M > 0 && D > 0 && (D & 3) == 0 && D <= 1024,
              "eav_layernorm_bwd: need D %% 4 == 0 and D <= 1024");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(eav_layernorm_bwd_nparts(M)), dim3(256), 8 * D * sizeof(float),
                     (hipStream_t)stream, dy, x, gamma, mean, rstd, dx, accumulate, part, M, D, (unsigned*)nullptr,
                     (unsigned char*)nullptr, (int64_t)0, 0.f, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr);
  EAV_CHECK_LAUNCH("eav_layernorm_bwd");
  return EAV_OK;
}

extern "C" int eav_layernorm_bwd_amax(const float* dy, const float* x, const float* gamma, const float* mean,
                                      const float* rstd, float* dx, int accumulate, float* part, int M, int D,
                                      float* amax_slot, void* stream) {
  EAV_REQUIRE(dy && x && gamma && mean && rstd && dx && amax_slot && M > 0 && D > 0 && (D & 3) == 0 && D <= 1024,
              "eav_layernorm_bwd_amax: need D %% 4 == 0 and D <= 1024");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(eav_layernorm_bwd_nparts(M)), dim3(256), 8 * D * sizeof(float),
                     (hipStream_t)stream, dy, x, gamma, mean, rstd, dx, accumulate, part, M, D,
                     reinterpret_cast<unsigned*>(amax_slot), (unsigned char*)nullptr, (int64_t)0, 0.f, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr);
  EAV_CHECK_LAUNCH("eav_layernorm_bwd_amax");
  return EAV_OK;
}

// eav_layernorm_bwd_amax whose stored value (dx, after the accumulation) ALSO leaves as the row planes [M][Dp/8][2][8] of the
// products that consume it - no conversion pass - scaled by slot's sigma: with slot_dy the kernel forms the bound of
// eav_layernorm_bwd_bound itself (slot_old / slot_dy / slot_rstd as there) and publishes it in slot; slot_dy NULL: the sigma
// already in slot.  part is
// [nparts][3*D]: dgamma | dbeta | column sums of the stored value (the bias gradient of the layer that fed the residual).
// slot's shards receive the measured tensor-wide maximum (no 32-row block entries: the scale is a-priori).
extern "C" int eav_layernorm_bwd_planes(const float* dy, const float* x, const float* gamma, const float* mean,
                                        const float* rstd, float* dx, int accumulate, float* part, int M, int D,
                                        float* slot, void* planes, const float* slot_old, const float* slot_dy,
                                        const float* slot_rstd, void* stream) {
  EAV_REQUIRE(dy && x && gamma && mean && rstd && dx && part && slot && planes && M > 0 && D > 0 && (D & 7) == 0 &&
                  D <= 1024 && ((uintptr_t)planes & 15) == 0 && (!slot_dy || slot_rstd) && slot_old != slot,
              "eav_layernorm_bwd_planes: need D %% 8 == 0, D <= 1024, 16-byte aligned planes");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(eav_layernorm_bwd_nparts(M)), dim3(256), 12 * D * sizeof(float),
                     (hipStream_t)stream, dy, x, gamma, mean, rstd, dx, accumulate, part, M, D,
                     reinterpret_cast<unsigned*>(slot), (unsigned char*)planes, (int64_t)eav_sp_kpad(D) * 4, 2048.f, slot_old,
                     slot_dy, slot_rstd);
  EAV_CHECK_LAUNCH("eav_layernorm_bwd_planes");
  return EAV_OK;
}


extern "C" int eav_softmax_fwd(float* s, int64_t rows, int N, int ld, void* stream) {
  EAV_REQUIRE(s && rows > 0 && N > 0 && ld >= N && N <= 2048, "eav_softmax_fwd: need N <= 2048");
  dim3 grid((unsigned)cdiv64(rows, 4));
  if (N <= 256) hipLaunchKernelGGL(softmax_fwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, s, rows, N, ld);
  else if (N <= 1280) hipLaunchKernelGGL(softmax_fwd_kernel<20>, grid, dim3(256), 0, (hipStream_t)stream, s, rows, N, ld);
  else hipLaunchKernelGGL(softmax_fwd_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, s, rows, N, ld);
  EAV_CHECK_LAUNCH("eav_softmax_fwd");
  return EAV_OK;
}

extern "C" int eav_softmax_bwd(const float* P, float* dP, int64_t rows, int N, int ld, void* stream) {
  EAV_REQUIRE(P && dP && rows > 0 && N > 0 && ld >= N && N <= 2048, "eav_softmax_bwd: need N <= 2048");
  dim3 grid((unsigned)cdiv64(rows, 4));
  if (N <= 256) hipLaunchKernelGGL(softmax_bwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, P, dP, rows, N, ld);
  else if (N <= 1280) hipLaunchKernelGGL(softmax_bwd_kernel<20>, grid, dim3(256), 0, (hipStream_t)stream, P, dP, rows, N, ld);
  else hipLaunchKernelGGL(softmax_bwd_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, P, dP, rows, N, ld);
  EAV_CHECK_LAUNCH("eav_softmax_bwd");
  return EAV_OK;
}

extern "C" int eav_gelu_bwd(float* dact, const float* pre, int64_t n, void* stream) {
  EAV_REQUIRE(dact && pre && n > 0 && (n & 3) == 0, "eav_gelu_bwd: n must be a positive multiple of 4");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, dact, pre, n / 4,
                     (unsigned*)nullptr);
  EAV_CHECK_LAUNCH("eav_gelu_bwd");
  return EAV_OK;
}

extern "C" int eav_gelu_bwd_amax(float* dact, const float* pre, int64_t n, float* amax_slot, void* stream) {
  EAV_REQUIRE(dact && pre && amax_slot && n > 0 && (n & 3) == 0, "eav_gelu_bwd_amax: n must be a positive multiple of 4");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, dact, pre, n / 4,
                     reinterpret_cast<unsigned*>(amax_slot));
  EAV_CHECK_LAUNCH("eav_gelu_bwd_amax");
  return EAV_OK;
}

extern "C" int eav_colsum_nparts(int M) { return cdiv(M, 256); }

extern "C" int eav_colsum(const float* dy, float* part, int M, int N, int ld, void* stream) {
  EAV_REQUIRE(dy && part && M > 0 && N > 0 && ld >= N, "eav_colsum: bad arguments");
  dim3 grid(cdiv(N, 64), cdiv(M, 256));
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, part, M, N, ld, 256);
  EAV_CHECK_LAUNCH("eav_colsum");
  return EAV_OK;
}

extern "C" int eav_im2col(const float* x, float* col, int B, int C, int H, int W, int P, int sy, int sx,
                          int transposed, void* stream) {
  EAV_REQUIRE(x && col && B > 0 && C > 0 && H >= P && W >= P && P > 0 && sy > 0 && sx > 0, "eav_im2col: bad arguments");
  const int ny = (H - P) / sy + 1, nx = (W - P) / sx + 1;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((int64_t)B * ny * nx * C * P * P)), dim3(256), 0,
                     (hipStream_t)stream, x, col, B, C, H, W, P, sy, sx, ny, nx, transposed);
  EAV_CHECK_LAUNCH("eav_im2col");
  return EAV_OK;
}

extern "C" int eav_embed_finish(float* h, const float* cls, const float* dist, const float* pos, int B, int ntok,
                                int D, int nextra, void* stream) {
  EAV_REQUIRE(h && cls && pos && B > 0 && ntok > nextra && (D & 3) == 0 && nextra >= 1 && nextra <= 2 &&
                  (nextra == 1 || dist), "eav_embed_finish: bad arguments");
  hipLaunchKernelGGL(embed_finish_kernel, dim3(grid_for((int64_t)B * ntok * D / 4)), dim3(256), 0,
                     (hipStream_t)stream, h, cls, dist, pos, B, ntok, D, nextra);
  EAV_CHECK_LAUNCH("eav_embed_finish");
  return EAV_OK;
}

extern "C" int eav_embed_bwd(const float* dh, float* dpos, float* demb, int B, int ntok, int D, int nextra,
                             void* stream) {
  EAV_REQUIRE(dh && dpos && demb && B > 0 && ntok > nextra && (D & 3) == 0, "eav_embed_bwd: bad arguments");
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for((int64_t)ntok * D / 4)), dim3(256), 0, (hipStream_t)stream, dh,
                     dpos, demb, B, ntok, D, nextra);
  EAV_CHECK_LAUNCH("eav_embed_bwd");
  return EAV_OK;
}

extern "C" int eav_token_rows(float* h, float* rows, int B, int ntok, int D, int nextra, int scatter, void* stream) {
  EAV_REQUIRE(h && rows && B > 0 && ntok >= nextra && nextra > 0 && (D & 3) == 0, "eav_token_rows: bad arguments");
  hipLaunchKernelGGL(token_rows_kernel, dim3(grid_for((int64_t)B * nextra * D / 4)), dim3(256), 0,
                     (hipStream_t)stream, h, rows, B, ntok, D, nextra, scatter);
  EAV_CHECK_LAUNCH("eav_token_rows");
  return EAV_OK;
}

extern "C" int eav_pair_mean(float* seq, float* pooled, int B, int D, int backward, void* stream) {
  EAV_REQUIRE(seq && pooled && B > 0 && D > 0, "eav_pair_mean: bad arguments");
  hipLaunchKernelGGL(pair_mean_kernel, dim3(grid_for((int64_t)B * D)), dim3(256), 0, (hipStream_t)stream, seq, pooled,
                     B, D, backward);
  EAV_CHECK_LAUNCH("eav_pair_mean");
  return EAV_OK;
}

// max_r ||w_r||_2 of a [R, C] matrix into *out (a float whose bits are combined with atomicMax: zero it first)
extern "C" int eav_rownorm_max(const float* w, int R, int C, int64_t ld, float* out, void* stream) {
  EAV_REQUIRE(w && out && R > 0 && C > 0 && ld >= C, "eav_rownorm_max: bad arguments");
  const int blocks = cdiv(R, 4) < 128 ? cdiv(R, 4) : 128;
  hipLaunchKernelGGL(rownorm_max_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, R, C, ld,
                     reinterpret_cast<unsigned*>(out));
  EAV_CHECK_LAUNCH("eav_rownorm_max");
  return EAV_OK;
}

extern "C" int eav_colnorm_max(const float* w, int R, int C, int64_t ld, float* out, void* stream) {
  EAV_REQUIRE(w && out && R > 0 && C > 0 && ld >= C && (C & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)w & 15) == 0,
              "eav_colnorm_max: bad arguments (C, ld multiples of 4, w 16-byte aligned)");
  hipLaunchKernelGGL(colnorm_max_kernel, dim3(cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, w, R, C, ld,
                     reinterpret_cast<unsigned*>(out));
  EAV_CHECK_LAUNCH("eav_colnorm_max");
  return EAV_OK;
}

// jobs: device table of njobs rows of 5 int64 {w, ld, out, R | C << 32, cols (0: row norms, 1: column norms)} (the outputs
// zeroed by the caller, as for eav_rownorm_max / eav_colnorm_max); max_blocks >= the largest job's block count
extern "C" int eav_norm_max_multi(const void* jobs, int njobs, int max_blocks, void* stream) {
  EAV_REQUIRE(jobs && njobs > 0 && max_blocks > 0, "eav_norm_max_multi: bad arguments");
  static_assert(sizeof(NormJob) == 40, "EavNormJob layout");
  hipLaunchKernelGGL(norm_max_multi_kernel, dim3(max_blocks, njobs), dim3(256), 0, (hipStream_t)stream, (const NormJob*)jobs);
  EAV_CHECK_LAUNCH("eav_norm_max_multi");
  return EAV_OK;
}

// sigma / 1 / sigma of slot_out from a bound of the tensor eav_layernorm_bwd_planes is about to store,
//   |dx_old + LN'(dy)| <= max|dx_old| + (2 + sqrt(D)) max|gamma| max_rows(rstd) max|dy|
// (dx = rstd (g - mean g - xhat mean(g xhat)), g = gamma o dy: |g - mean g| <= 2 max|g|, |xhat| <= sqrt(D), mean|xhat| <= 1).
// slot_old: the slot whose shards hold the MEASURED max|dx_old| (0: no accumulation); slot_dy: shards of max|dy|.  One block.
__global__ __launch_bounds__(256) void layernorm_bwd_bound_kernel(float* __restrict__ slot_out, const float* __restrict__ slot_old,
                                                                  const float* __restrict__ slot_dy,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ rstd, int M, int D,
                                                                  const float* __restrict__ slot_rstd) {
  __shared__ float red[8];
  float gmx = 0.f, rmx = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) gmx = fmaxf(gmx, fabsf(gamma[i]));
  // max rstd: from the forward's slot (word 1 of the 64 shard lines, layernorm_fwd_kernel) when there is one - a walk over the
  // M-long rstd array by one block cost 69 us per launch at M = 25216
  if (slot_rstd) {
    if (threadIdx.x < 64) rmx = slot_rstd[32 * threadIdx.x + 1];
  } else {
    for (int i = threadIdx.x; i < M; i += 256) rmx = fmaxf(rmx, rstd[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    gmx = fmaxf(gmx, __shfl_xor(gmx, o, 64));
    rmx = fmaxf(rmx, __shfl_xor(rmx, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = gmx; red[4 + (threadIdx.x >> 6)] = rmx; }
  __syncthreads();
  if (slot_rstd && rstd && fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])) == 0.f) {
    // the slot's word 1 was never written by the matching forward (rstd > 0 always): walk the saved rstd instead
    __syncthreads();
    rmx = 0.f;
    for (int i = threadIdx.x; i < M; i += 256) rmx = fmaxf(rmx, rstd[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rmx = fmaxf(rmx, __shfl_xor(rmx, o, 64));
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = rmx;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    gmx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    rmx = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    const float old = slot_old ? __uint_as_float(eav_slot_bits(slot_old)) : 0.f;
    const float dy = __uint_as_float(eav_slot_bits(slot_dy));
    const float bound = (old + (2.f + sqrtf((float)D)) * gmx * rmx * dy) * 1.0001f;
    const float sg = sigma_of_bound(bound);
    slot_out[EAV_SLOT_SIGMA] = sg;
    slot_out[EAV_SLOT_ISIGMA] = 1.f / sg;
  }
}

// sigma / 1 / sigma of slot_out from the bound factor * max|x| * *norm, max|x| = the maximum in amax_slot's shards: the
// operand scale of a tensor that is about to be produced, from a bound of its magnitude (e.g. the MLP's hidden-state
// gradient dact = (dh W2) o gelu'(pre): |dact| <= 1.13 sqrt(D) max|dh| max_j ||W2[:, j]||_2) - its producer then writes
// the planes directly (eav_gemm_sp_ex with planes_out).  One wave.
__global__ __launch_bounds__(64) void bound_scale_kernel(float* __restrict__ slot_out, const float* __restrict__ amax_slot,
                                                         const float* __restrict__ norm, float factor) {
  const unsigned bits = eav_slot_bits(amax_slot);
  if (threadIdx.x == 0) {
    const float bound = factor * __uint_as_float(bits) * norm[0] * 1.0001f;
    const float sg = sigma_of_bound(bound);
    slot_out[EAV_SLOT_SIGMA] = sg;
    slot_out[EAV_SLOT_ISIGMA] = 1.f / sg;
  }
}

extern "C" int eav_layernorm_bwd_bound(float* slot_out, const float* slot_old, const float* slot_dy, const float* gamma,
                                       const float* rstd, int M, int D, const float* slot_rstd, void* stream) {
  EAV_REQUIRE(slot_out && slot_dy && gamma && (rstd || slot_rstd) && M > 0 && D > 0, "eav_layernorm_bwd_bound: bad arguments");
  hipLaunchKernelGGL(layernorm_bwd_bound_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, slot_out, slot_old, slot_dy, gamma,
                     rstd, M, D, slot_rstd);
  EAV_CHECK_LAUNCH("eav_layernorm_bwd_bound");
  return EAV_OK;
}

extern "C" int eav_sp_bound_scale(float* slot_out, const float* amax_slot, const float* norm, float factor, void* stream) {
  EAV_REQUIRE(slot_out && amax_slot && norm && factor > 0.f, "eav_sp_bound_scale: bad arguments");
  hipLaunchKernelGGL(bound_scale_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slot_out, amax_slot, norm, factor);
  EAV_CHECK_LAUNCH("eav_sp_bound_scale");
  return EAV_OK;
}

// sigma / 1 / sigma of the forward operand slots y1 (LayerNorm-before output), y2 (LayerNorm-after output) and act
// (GELU output) of every layer from rigorous bounds (see tf_forward_scales_kernel).  params = first float of layer 0 in
// the flat parameter buffer, layer_stride floats per layer, off_* = offsets of layernorm_before.{weight,bias},
// layernorm_after.{weight,bias}, mlp.fc1.bias inside a layer; wnorm_fc1 [layers] = eav_rownorm_max of mlp.fc1.weight;
// slots = slot of layer 0's first forward operand, slot_stride floats per layer, k_* = slot index within a layer.
static int tf_forward_scales_impl(const float* params, int64_t layer_stride, int layers, int off_g1, int off_b1,
                                  int off_g2, int off_b2, int off_bfc1, int D, int FF, const float* wnorm_fc1,
                                  float* slots, int64_t slot_stride, int k_y1, int k_y2, int k_act, int off_bqkv,
                                  const float* wnorm_qkv, int k_qkv, void* stream) {
  EAV_REQUIRE(params && wnorm_fc1 && slots && layers > 0 && D > 0 && FF > 0 && (k_qkv < 0 || wnorm_qkv),
              "eav_tf_forward_scales: bad arguments");
  hipLaunchKernelGGL(tf_forward_scales_kernel, dim3(layers), dim3(256), 0, (hipStream_t)stream, params, layer_stride,
                     off_g1, off_b1, off_g2, off_b2, off_bfc1, D, FF, wnorm_fc1, slots, slot_stride, k_y1, k_y2, k_act,
                     EAV_SP_SLOT, off_bqkv, wnorm_qkv, k_qkv);
  EAV_CHECK_LAUNCH("eav_tf_forward_scales");
  return EAV_OK;
}

extern "C" int eav_tf_forward_scales(const float* params, int64_t layer_stride, int layers, int off_g1, int off_b1,
                                     int off_g2, int off_b2, int off_bfc1, int D, int FF, const float* wnorm_fc1,
                                     float* slots, int64_t slot_stride, int k_y1, int k_y2, int k_act, void* stream) {
  return tf_forward_scales_impl(params, layer_stride, layers, off_g1, off_b1, off_g2, off_b2, off_bfc1, D, FF, wnorm_fc1,
                                slots, slot_stride, k_y1, k_y2, k_act, 0, nullptr, -1, stream);
}

// the same plus the slot of the fused q/k/v projection's output (k_qkv) from ||y1||_2 max_n ||Wqkv_n||_2 + max|b_qkv|:
// off_bqkv = offset of the fused [3D] bias in a layer, wnorm_qkv [layers] = eav_rownorm_max of the fused [3D, D] weight
extern "C" int eav_tf_forward_scales_qkv(const float* params, int64_t layer_stride, int layers, int off_g1, int off_b1,
                                         int off_g2, int off_b2, int off_bfc1, int off_bqkv, int D, int FF,
                                         const float* wnorm_fc1, const float* wnorm_qkv, float* slots, int64_t slot_stride,
                                         int k_y1, int k_y2, int k_act, int k_qkv, void* stream) {
  return tf_forward_scales_impl(params, layer_stride, layers, off_g1, off_b1, off_g2, off_b2, off_bfc1, D, FF, wnorm_fc1,
                                slots, slot_stride, k_y1, k_y2, k_act, off_bqkv, wnorm_qkv, k_qkv, stream);
}
