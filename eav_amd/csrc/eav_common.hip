// Error reporting + tiny generic kernels (partial-sum reduction, BatchNorm finalisers).
#include "eav_common.h"

#include <stdarg.h>
#include <stdio.h>

#include "../../include/eav_hip.h"

static thread_local char g_err[512] = "";

int eav_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* eav_last_error(void) { return g_err; }
extern "C" int eav_abi_version(void) { return EAV_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------
// out[i] = scale * sum_p part[p*stride + i]   (fp64 accumulation; deterministic order)
// block = 16 outputs x 16 part-lanes; every lane sums its parts in fp64, the 16 lane sums are
// added in a fixed order -> bit-reproducible.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int nparts,
                                                              int64_t stride, int n, float scale,
                                                              float* __restrict__ out) {
  __shared__ double sh[16][17];
  const int ol = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + ol;
  double s = 0.0;
  if (i < n) {
    const float* src = part + i;
    int p = pl;
    for (; p + 48 < nparts; p += 64) {
      float a = src[(int64_t)p * stride], b = src[(int64_t)(p + 16) * stride];
      float c = src[(int64_t)(p + 32) * stride], d = src[(int64_t)(p + 48) * stride];
      s += ((double)a + (double)b) + ((double)c + (double)d);
    }
    for (; p < nparts; p += 16) s += (double)src[(int64_t)p * stride];
  }
  sh[pl][ol] = s;
  __syncthreads();
  if (threadIdx.x < 16 && i < n) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][threadIdx.x];
    out[i] = (float)(t * (double)scale);
  }
}

// The same reduction for 16-byte-aligned operands: a block is CL column lanes (one float4 = 4 outputs each) x 256 / CL
// part lanes; a wave instruction reads whole 16 CL-byte row segments (128 B for CL = 8) with four independent loads in
// flight per lane.  Every lane sums its parts p = pl, pl + PL, ... in fp64 in that order, the PL lane sums are added in
// lane order: fixed order, bit-reproducible.  (The scalar kernel above read 64-byte segments with one load in flight and
// ran at 45-80 GB/s on the [~400, 3072] bias-gradient partials of the encoders; this one is latency-bound at a few us.)
template <int CL>
__device__ __forceinline__ void reduce_partials_vec_body(const float* __restrict__ part, int nparts, int64_t stride, int n4,
                                                         float scale, float* __restrict__ out, int bid) {
  constexpr int PL = 256 / CL;
  __shared__ double sh[PL][4 * CL + 1];
  const int cl = threadIdx.x % CL, pl = threadIdx.x / CL;
  const int c4 = bid * CL + cl;
  double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
  if (c4 < n4) {
    const float* src = part + 4 * (int64_t)c4;
    int p = pl;
    for (; p + 3 * PL < nparts; p += 4 * PL) {
      const float4 v0 = *reinterpret_cast<const float4*>(src + (int64_t)p * stride);
      const float4 v1 = *reinterpret_cast<const float4*>(src + (int64_t)(p + PL) * stride);
      const float4 v2 = *reinterpret_cast<const float4*>(src + (int64_t)(p + 2 * PL) * stride);
      const float4 v3 = *reinterpret_cast<const float4*>(src + (int64_t)(p + 3 * PL) * stride);
      a += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
      b += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
      c += ((double)v0.z + (double)v1.z) + ((double)v2.z + (double)v3.z);
      d += ((double)v0.w + (double)v1.w) + ((double)v2.w + (double)v3.w);
    }
    for (; p < nparts; p += PL) {
      const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)p * stride);
      a += (double)v.x; b += (double)v.y; c += (double)v.z; d += (double)v.w;
    }
  }
  sh[pl][4 * cl] = a; sh[pl][4 * cl + 1] = b; sh[pl][4 * cl + 2] = c; sh[pl][4 * cl + 3] = d;
  __syncthreads();
  if (threadIdx.x < 4 * CL && 4 * bid * CL + threadIdx.x < 4 * n4) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < PL; ++k) t += sh[k][threadIdx.x];
    out[4 * (int64_t)bid * CL + threadIdx.x] = (float)(t * (double)scale);
  }
}

template <int CL>
__global__ __launch_bounds__(256) void reduce_partials_vec_kernel(const float* __restrict__ part, int nparts,
                                                                  int64_t stride, int n4, float scale,
                                                                  float* __restrict__ out) {
  reduce_partials_vec_body<CL>(part, nparts, stride, n4, scale, out, blockIdx.x);
}

extern "C" int eav_reduce_partials(const float* part, int nparts, int64_t stride, int n, float scale, float* out,
                                   void* stream) {
  EAV_REQUIRE(part && out && nparts > 0 && n > 0, "eav_reduce_partials: bad arguments");
  const bool vec = (n & 3) == 0 && (stride & 3) == 0 && (((uintptr_t)part | (uintptr_t)out) & 15) == 0;
  if (vec && n >= 8192)        // wide outputs (weight-gradient partials): 64-float column groups, >= 128 blocks
    hipLaunchKernelGGL(reduce_partials_vec_kernel<16>, dim3(cdiv(n / 4, 16)), dim3(256), 0, (hipStream_t)stream, part,
                       nparts, stride, n / 4, scale, out);
  else if (vec)                // narrow outputs (bias / LayerNorm gradients): more part lanes per output
    hipLaunchKernelGGL(reduce_partials_vec_kernel<8>, dim3(cdiv(n / 4, 8)), dim3(256), 0, (hipStream_t)stream, part,
                       nparts, stride, n / 4, scale, out);
  else
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(n, 16)), dim3(256), 0, (hipStream_t)stream, part, nparts,
                       stride, n, scale, out);
  EAV_CHECK_LAUNCH("eav_reduce_partials");
  return EAV_OK;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm forward finaliser.  part[p][0..nch) = sum x, part[p][nch..2nch) = sum x^2.
// training: batch mean / biased var normalise; running stats updated with the unbiased var
// (momentum 0.1 in the reference's nn.BatchNorm2d, EEGNet_tor.py:25,29,38).  eval: running stats.
// Emits mean, invstd and the fused affine  y = scale*x + shift.
__global__ void bn_finalize_kernel(const float* __restrict__ part, int nparts, int nch, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, int training, float momentum,
                                   float eps, float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                   float* __restrict__ scale_o, float* __restrict__ shift_o) {
  const int c = blockIdx.x;
  __shared__ double shs[256], shq[256];
  float mean, var;
  if (training) {
    double s = 0.0, q = 0.0;
    for (int p = threadIdx.x; p < nparts; p += 256) {
      s += (double)part[(int64_t)p * 2 * nch + c];
      q += (double)part[(int64_t)p * 2 * nch + nch + c];
    }
    shs[threadIdx.x] = s;
    shq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (threadIdx.x < o) {
        shs[threadIdx.x] += shs[threadIdx.x + o];
        shq[threadIdx.x] += shq[threadIdx.x + o];
      }
      __syncthreads();
    }
    if (threadIdx.x != 0) return;
    s = shs[0];
    q = shq[0];
    double m = s / count;
    double v = q / count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m;
    var = (float)v;
    double unb = count > 1.0 ? v * (count / (count - 1.0)) : v;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  } else {
    if (threadIdx.x != 0) return;
    mean = rmean[c];
    var = rvar[c];
  }
  float invstd = 1.0f / sqrtf(var + eps);
  float sc = gamma[c] * invstd;
  mean_o[c] = mean;
  invstd_o[c] = invstd;
  scale_o[c] = sc;
  shift_o[c] = beta[c] - mean * sc;
}

extern "C" int eav_bn_finalize(const float* part, int nparts, int nch, double count, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, int training,
                               float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                               void* stream) {
  EAV_REQUIRE(nch > 0 && gamma && beta && running_mean && running_var && mean && invstd && scale && shift,
              "eav_bn_finalize: bad arguments");
  EAV_REQUIRE(!training || (part && nparts > 0 && count > 0), "eav_bn_finalize: training needs partials");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(nch), dim3(256), 0, (hipStream_t)stream, part, nparts, nch,
                     count, gamma, beta, running_mean, running_var, training, momentum, eps, mean, invstd, scale,
                     shift);
  EAV_CHECK_LAUNCH("eav_bn_finalize");
  return EAV_OK;
}

// BatchNorm backward finaliser.  part[p][0..nch) = sum g, part[p][nch..2nch) = sum g*xhat
// (g = gradient w.r.t. the BN output).  dbeta = sum g, dgamma = sum g*xhat; in training mode the
// input gradient is scale*(g - m1 - xhat*m2) with m1 = mean g, m2 = mean g*xhat; in eval mode
// (running statistics are constants) m1 = m2 = 0.
__device__ __forceinline__ void bn_bwd_finalize_body(const float* __restrict__ part, int nparts, int nch, double count,
                                                     int training, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ m1, float* __restrict__ m2, int c) {
  __shared__ double shs[256], shq[256];
  double s = 0.0, q = 0.0;
  for (int p = threadIdx.x; p < nparts; p += 256) {
    s += (double)part[(int64_t)p * 2 * nch + c];
    q += (double)part[(int64_t)p * 2 * nch + nch + c];
  }
  shs[threadIdx.x] = s;
  shq[threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      shs[threadIdx.x] += shs[threadIdx.x + o];
      shq[threadIdx.x] += shq[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  s = shs[0];
  q = shq[0];
  dbeta[c] = (float)s;
  dgamma[c] = (float)q;
  m1[c] = training ? (float)(s / count) : 0.f;
  m2[c] = training ? (float)(q / count) : 0.f;
}

__global__ void bn_bwd_finalize_kernel(const float* __restrict__ part, int nparts, int nch, double count,
                                       int training, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                       float* __restrict__ m1, float* __restrict__ m2) {
  bn_bwd_finalize_body(part, nparts, nch, count, training, dgamma, dbeta, m1, m2, blockIdx.x);
}

// One launch for the small finishing work behind a gradient pass: a fixed-order reduction of partial rows (a weight gradient)
// and up to two BatchNorm-backward finalisers - the same bodies as eav_reduce_partials / eav_bn_bwd_finalize on disjoint block
// ranges, hence the same bits.  Each was a graph node of its own at the ~4.7 us floor of a dependent node (EEGNet step:
// depthwiseConv.weight + firstBN after eav_eegnet_dw_bwd_fused; + depthwiseBN in the eval-mode step).
struct BnBwdJob {
  const float* part;
  int nparts, nch, training;
  double count;
  float *dgamma, *dbeta, *m1, *m2;
};

__global__ __launch_bounds__(256) void finish_jobs_kernel(const float* __restrict__ part, int nparts, int64_t stride, int n4,
                                                          float scale, float* __restrict__ out, int nred, BnBwdJob a,
                                                          BnBwdJob b) {
  const int bid = blockIdx.x;
  if (bid < nred) {
    reduce_partials_vec_body<8>(part, nparts, stride, n4, scale, out, bid);
    return;
  }
  const int c = bid - nred;
  const BnBwdJob& j = c < a.nch ? a : b;
  bn_bwd_finalize_body(j.part, j.nparts, j.nch, j.count, j.training, j.dgamma, j.dbeta, j.m1, j.m2, c < a.nch ? c : c - a.nch);
}

extern "C" int eav_reduce_and_bn_bwd_finalize(const float* part, int nparts, int64_t stride, int n, float* out,
                                              const float* part_a, int nparts_a, int nch_a, double count_a, int training_a,
                                              float* dgamma_a, float* dbeta_a, float* m1_a, float* m2_a,
                                              const float* part_b, int nparts_b, int nch_b, double count_b, int training_b,
                                              float* dgamma_b, float* dbeta_b, float* m1_b, float* m2_b, void* stream) {
  EAV_REQUIRE(part && out && nparts > 0 && n > 0, "eav_reduce_and_bn_bwd_finalize: bad reduction arguments");
  EAV_REQUIRE(part_a && nparts_a > 0 && nch_a > 0 && count_a > 0 && dgamma_a && dbeta_a && m1_a && m2_a,
              "eav_reduce_and_bn_bwd_finalize: bad BatchNorm arguments");
  EAV_REQUIRE(!part_b || (nparts_b > 0 && nch_b > 0 && count_b > 0 && dgamma_b && dbeta_b && m1_b && m2_b),
              "eav_reduce_and_bn_bwd_finalize: bad second BatchNorm arguments");
  const bool vec8 = (n & 3) == 0 && (stride & 3) == 0 && (((uintptr_t)part | (uintptr_t)out) & 15) == 0 && n < 8192;
  if (!vec8) {      // shapes eav_reduce_partials serves with its other kernels: the separate launches (same results)
    int rc = eav_reduce_partials(part, nparts, stride, n, 1.f, out, stream);
    if (rc == EAV_OK)
      rc = eav_bn_bwd_finalize(part_a, nparts_a, nch_a, count_a, training_a, dgamma_a, dbeta_a, m1_a, m2_a, stream);
    if (rc == EAV_OK && part_b)
      rc = eav_bn_bwd_finalize(part_b, nparts_b, nch_b, count_b, training_b, dgamma_b, dbeta_b, m1_b, m2_b, stream);
    return rc;
  }
  BnBwdJob a{part_a, nparts_a, nch_a, training_a, count_a, dgamma_a, dbeta_a, m1_a, m2_a};
  BnBwdJob b{part_b, nparts_b, part_b ? nch_b : 0, training_b, count_b, dgamma_b, dbeta_b, m1_b, m2_b};
  const int nred = cdiv(n / 4, 8);
  hipLaunchKernelGGL(finish_jobs_kernel, dim3(nred + a.nch + b.nch), dim3(256), 0, (hipStream_t)stream, part, nparts, stride,
                     n / 4, 1.f, out, nred, a, b);
  EAV_CHECK_LAUNCH("eav_reduce_and_bn_bwd_finalize");
  return EAV_OK;
}

extern "C" int eav_bn_bwd_finalize(const float* part, int nparts, int nch, double count, int training,
                                   float* dgamma, float* dbeta, float* m1, float* m2, void* stream) {
  EAV_REQUIRE(part && nparts > 0 && nch > 0 && count > 0 && dgamma && dbeta && m1 && m2,
              "eav_bn_bwd_finalize: bad arguments");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(nch), dim3(256), 0, (hipStream_t)stream, part, nparts,
                     nch, count, training, dgamma, dbeta, m1, m2);
  EAV_CHECK_LAUNCH("eav_bn_bwd_finalize");
  return EAV_OK;
}

// ---------------------------------------------------------------------------------------------
// Tensor.renorm_(p=2, dim=0, maxnorm) on a [rows, cols] matrix (EEGNet_tor.py:34,48):
// rows with ||w||_2 > maxnorm are scaled by maxnorm / (norm + 1e-7).
// (second matrix w_b [rows_b, cols_b]: blocks rows .. rows + rows_b - 1; eav_renorm_rows2)
__global__ void renorm_rows_kernel(float* __restrict__ w, int rows, int cols, float maxnorm, float* __restrict__ w_b,
                                   int cols_b) {
  __shared__ float red[8];
  int r = blockIdx.x;
  if (r >= rows) {
    r -= rows;
    w = w_b;
    cols = cols_b;
  }
  float* row = w + (int64_t)r * cols;
  // long rows (the dense layer's 5 x 19 968): 16 bytes per lane and four loads in flight - one block per row is a chain of
  // load latencies (78 dependent trips of 4 bytes took 10 us of the EEGNet step)
  const bool vec = (cols & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0;
  float v[1] = {0.f};
  if (vec) {
    const float4* row4 = reinterpret_cast<const float4*>(row);
    const int n4 = cols >> 2;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = threadIdx.x; i0 < n4; i0 += 1024) {
      float4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = i0 + 256 * u < n4 ? row4[i0 + 256 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += (q[u].x * q[u].x + q[u].y * q[u].y) + (q[u].z * q[u].z + q[u].w * q[u].w);
    }
    v[0] = (a[0] + a[1]) + (a[2] + a[3]);
  } else {
    for (int i = threadIdx.x; i < cols; i += 256) v[0] += row[i] * row[i];
  }
  block_sum_256<1>(v, red);
  __shared__ float sc;
  if (threadIdx.x == 0) {
    float norm = sqrtf(v[0]);
    sc = norm > maxnorm ? maxnorm / (norm + 1e-7f) : 1.f;
  }
  __syncthreads();
  float s = sc;
  if (s == 1.f) return;
  if (vec) {
    float4* row4 = reinterpret_cast<float4*>(row);
    const int n4 = cols >> 2;
    for (int i0 = threadIdx.x; i0 < n4; i0 += 1024) {
      float4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = i0 + 256 * u < n4 ? row4[i0 + 256 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i0 + 256 * u < n4) row4[i0 + 256 * u] = make_float4(q[u].x * s, q[u].y * s, q[u].z * s, q[u].w * s);
    }
  } else {
    for (int i = threadIdx.x; i < cols; i += 256) row[i] *= s;
  }
}

extern "C" int eav_renorm_rows(float* w, int rows, int cols, float maxnorm, void* stream) {
  EAV_REQUIRE(w && rows > 0 && cols > 0, "eav_renorm_rows: bad arguments");
  hipLaunchKernelGGL(renorm_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, w, rows, cols, maxnorm,
                     (float*)nullptr, 0);
  EAV_CHECK_LAUNCH("eav_renorm_rows");
  return EAV_OK;
}

extern "C" int eav_renorm_rows2(float* w_a, int rows_a, int cols_a, float* w_b, int rows_b, int cols_b, float maxnorm,
                                void* stream) {
  EAV_REQUIRE(w_a && rows_a > 0 && cols_a > 0 && w_b && rows_b > 0 && cols_b > 0, "eav_renorm_rows2: bad arguments");
  hipLaunchKernelGGL(renorm_rows_kernel, dim3(rows_a + rows_b), dim3(256), 0, (hipStream_t)stream, w_a, rows_a, cols_a,
                     maxnorm, w_b, cols_b);
  EAV_CHECK_LAUNCH("eav_renorm_rows2");
  return EAV_OK;
}
