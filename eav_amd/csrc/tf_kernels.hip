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
                                                            int M, int D, float eps, unsigned* __restrict__ amax) {
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
      dst[q] = o;
      vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
    }
  }
  if (lane == 0) {
    if (mean_o) mean_o[row] = mean;
    if (rstd_o) rstd_o[row] = rstd;
  }
  if (amax) {   // max |y| of the row into one of the 64 shards of the operand-scale slot (gemm_sp.hip)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if (lane == 0 && vmax == vmax) atomicMax(amax + EAV_SLOT_SHARD(row), __float_as_uint(vmax));
  }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat));  dgamma += dy*xhat, dbeta += dy per column.
// Persistent blocks; part[blk][2*D] = (dgamma, dbeta) partials of the rows this block handled.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean_i,
                                                            const float* __restrict__ rstd_i, float* __restrict__ dx,
                                                            int accumulate, float* __restrict__ part, int M, int D,
                                                            unsigned* __restrict__ amax) {
  extern __shared__ float sh[];  // [4][2*D]
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
  int row = blockIdx.x * 4 + wave;
  if (row < M) fetch(row);
  for (; row < M; row += gridDim.x * 4) {
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
    const int next = row + gridDim.x * 4;
    if (next < M) fetch(next);
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
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
        vmax = fmaxf(vmax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
      }
    }
  }
  if (amax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if (lane == 0 && vmax == vmax) atomicMax(amax + EAV_SLOT_SHARD(blockIdx.x * 4 + wave), __float_as_uint(vmax));
  }
  if (!part) return;
  float4* shw = reinterpret_cast<float4*>(sh + wave * 2 * D);
#pragma unroll
  for (int i = 0; i < LN_MAXQ; ++i) {
    const int q = lane + 64 * i;
    if (q < nq) {
      shw[q] = ag[i];
      shw[nq + q] = ab[i];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256)
    part[(int64_t)blockIdx.x * 2 * D + i] = (sh[i] + sh[2 * D + i]) + (sh[4 * D + i] + sh[6 * D + i]);
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

}  // namespace

extern "C" int eav_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                 float* rstd, int M, int D, float eps, void* stream) {
  EAV_REQUIRE(x && gamma && beta && y && M > 0 && D > 0 && (D & 3) == 0 && D <= 1024,
              "eav_layernorm_fwd: need D %% 4 == 0 and D <= 1024");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y,
                     mean, rstd, M, D, eps, (unsigned*)nullptr);
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
                     mean, rstd, M, D, eps, reinterpret_cast<unsigned*>(amax_slot));
  EAV_CHECK_LAUNCH("eav_layernorm_fwd_amax");
  return EAV_OK;
}

extern "C" int eav_layernorm_bwd_nparts(int M) { return cdiv(M, 4) < 256 ? cdiv(M, 4) : 256; }

extern "C" int eav_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                 const float* rstd, float* dx, int accumulate, float* part, int M, int D,
                                 void* stream) {
  EAV_REQUIRE(dy && x && gamma && mean && rstd && dx && M > 0 && D > 0 && (D & 3) == 0 && D <= 1024,
              "eav_layernorm_bwd: need D %% 4 == 0 and D <= 1024");
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(eav_layernorm_bwd_nparts(M)), dim3(256), 8 * D * sizeof(float),
                     (hipStream_t)stream, dy, x, gamma, mean, rstd, dx, accumulate, part, M, D, (unsigned*)nullptr);
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
                     reinterpret_cast<unsigned*>(amax_slot));
  EAV_CHECK_LAUNCH("eav_layernorm_bwd_amax");
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
