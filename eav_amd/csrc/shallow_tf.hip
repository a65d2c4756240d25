// ShallowConvNet + single-head transformer (Transformer_torch/Transformer_EEG.py:14-148): the pieces that are not
// plain GEMM / LayerNorm / attention (those reuse gemm_f32.hip, tf_kernels.hip, attention.hip):
//   * conv(1->NF,(1,KC)) fused with the per-filter channel projection of PatchEmbedding (:117,:28-35) - the
//     reference materialises the [B,NF,Chans,T] conv output; here v[b,t,i] = sum_j w[i,j] * u[b,i,t+j] with
//     u[b,i,s] = sum_c wv[i,c] x[b,c,s], which is the same sum re-associated (NF*Chans*S + NF*KC*T MACs per sample)
//   * ReLU+Dropout, Dropout+residual, strided row add (the "+ V" residual of MultiHeadAttention, :74-76)
//   * head: BatchNorm over tokens -> square -> AvgPool(1,35)/7 -> log(clamp) -> Dropout (:135-144)
// All HBM-bound element-wise / small-reduction kernels; reductions are two-stage and ordered.
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

constexpr int ETS = 128;   // embed forward: time samples per block
constexpr int BTS = 64;    // embed backward: time samples per block (four staged tiles must fit the 64 KB of LDS)
constexpr int NFMAX = 48, KCMAX = 16, CHMAX = 32;

// ------------------------------------------------------------------------------------ embed fwd
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wc,
                                                        const float* __restrict__ wv, float* __restrict__ u,
                                                        float* __restrict__ v, int C, int S, int NF, int KC,
                                                        int ldv) {
  __shared__ float xs[CHMAX][ETS + KCMAX];
  __shared__ float us[NFMAX][ETS + KCMAX + 1];
  __shared__ float wvs[NFMAX][CHMAX + 1];
  __shared__ float wcs[NFMAX][KCMAX + 1];
  const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, s0 = tile * ETS, T = S - KC + 1;
  const int W = ETS + KC - 1;
  for (int i = tid; i < C * W; i += 256) {
    const int c = i / W, sl = i - c * W, s = s0 + sl;
    xs[c][sl] = s < S ? x[((int64_t)b * C + c) * S + s] : 0.f;
  }
  for (int i = tid; i < NF * C; i += 256) wvs[i / C][i % C] = wv[(i / C) * ldv + i % C];
  for (int i = tid; i < NF * KC; i += 256) wcs[i / KC][i % KC] = wc[i];
  __syncthreads();
  for (int i = tid; i < NF * W; i += 256) {
    const int f = i / W, sl = i - f * W, s = s0 + sl;
    float a = 0.f;
    for (int c = 0; c < C; ++c) a = fmaf(wvs[f][c], xs[c][sl], a);
    us[f][sl] = a;
    if (sl < ETS && s < S) u[((int64_t)b * NF + f) * S + s] = a;
  }
  __syncthreads();
  for (int i = tid; i < ETS * NF; i += 256) {
    const int tl = i / NF, f = i - tl * NF, t = s0 + tl;
    if (t < T) {
      float a = 0.f;
      for (int j = 0; j < KC; ++j) a = fmaf(wcs[f][j], us[f][tl + j], a);
      v[((int64_t)b * T + t) * NF + f] = a;
    }
  }
}

// ------------------------------------------------------------------------------------ embed bwd
// part_c[blk][f*KC+j] = sum_t dv[b,t,f] * u[b,f,t+j];  part_v[blk][f*C+c] = sum_s e[f,s] * x[b,c,s],
// e[f,s] = sum_j wc[f,j] * dv[b,s-j,f].
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dv, const float* __restrict__ x,
                                                        const float* __restrict__ u, const float* __restrict__ wc,
                                                        float* __restrict__ part_c, float* __restrict__ part_v, int C,
                                                        int S, int NF, int KC) {
  __shared__ float xs[CHMAX][BTS + 1];
  __shared__ float us[NFMAX][BTS + KCMAX + 1];
  __shared__ float dvs[NFMAX][BTS + KCMAX + 1];   // dvs[f][KC-1 + tl] = dv[t0+tl], with KC-1 samples of left halo
  __shared__ float es[NFMAX][BTS + 1];
  __shared__ float wcs[NFMAX][KCMAX + 1];
  const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, s0 = tile * BTS, T = S - KC + 1;
  const int blk = b * gridDim.x + tile, W = BTS + KC - 1;
  for (int i = tid; i < C * BTS; i += 256) {
    const int c = i / BTS, sl = i - c * BTS, s = s0 + sl;
    xs[c][sl] = s < S ? x[((int64_t)b * C + c) * S + s] : 0.f;
  }
  for (int i = tid; i < NF * W; i += 256) {
    const int f = i / W, sl = i - f * W, s = s0 + sl;
    us[f][sl] = s < S ? u[((int64_t)b * NF + f) * S + s] : 0.f;
  }
  for (int i = tid; i < W * NF; i += 256) {
    const int hl = i / NF, f = i - hl * NF, t = s0 + hl - (KC - 1);
    dvs[f][hl] = (t >= 0 && t < T) ? dv[((int64_t)b * T + t) * NF + f] : 0.f;
  }
  for (int i = tid; i < NF * KC; i += 256) wcs[i / KC][i % KC] = wc[i];
  __syncthreads();
  for (int i = tid; i < NF * BTS; i += 256) {
    const int f = i / BTS, sl = i - f * BTS;
    float a = 0.f;
    for (int j = 0; j < KC; ++j) a = fmaf(wcs[f][j], dvs[f][KC - 1 + sl - j], a);
    es[f][sl] = a;
  }
  __syncthreads();
  for (int i = tid; i < NF * KC; i += 256) {
    const int f = i / KC, j = i - f * KC;
    float a = 0.f;
    for (int tl = 0; tl < BTS; ++tl) a = fmaf(dvs[f][KC - 1 + tl], us[f][tl + j], a);
    part_c[(int64_t)blk * NF * KC + i] = a;
  }
  for (int i = tid; i < NF * C; i += 256) {
    const int f = i / C, c = i - f * C;
    float a = 0.f;
    for (int sl = 0; sl < BTS; ++sl) a = fmaf(es[f][sl], xs[c][sl], a);
    part_v[(int64_t)blk * NF * C + i] = a;
  }
}

// ------------------------------------------------------------------------------------ element-wise
__global__ void relu_dropout_kernel(float* __restrict__ h, int64_t n, float drop_p, uint64_t seed_in,
                                    const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev) {
  const uint64_t seed = dropout_seed(seed_in, seed_dev);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = h[i];
    h[i] = v > 0.f ? v * dropout_mult(drop_p, seed, mask, i) : 0.f;
  }
}
// act is the OUTPUT of the forward (relu(pre) * mult): act > 0  <=>  pre > 0 and kept
__global__ void relu_dropout_bwd_kernel(float* __restrict__ dact, const float* __restrict__ act, int64_t n,
                                        float scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    dact[i] = act[i] > 0.f ? dact[i] * scale : 0.f;
}
// out = resid + y * mult   (out may alias resid)
__global__ void dropout_add_kernel(const float* __restrict__ y, const float* __restrict__ resid,
                                   float* __restrict__ out, int64_t n, float drop_p, uint64_t seed_in,
                                   const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev) {
  const uint64_t seed = dropout_seed(seed_in, seed_dev);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = (resid ? resid[i] : 0.f) + y[i] * dropout_mult(drop_p, seed, mask, i);
}
// out[m*ldo+j] = a[m*lda+j] (+ b[m*ldb+j]),  j < n
__global__ void add_strided_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                   float* __restrict__ out, int ldo, int64_t M, int n) {
  const int64_t total = M * n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / n;
    const int j = (int)(i - m * n);
    out[m * ldo + j] = a[m * lda + j] + (b ? b[m * ldb + j] : 0.f);
  }
}

// ------------------------------------------------------------------------------------ column statistics
// part[blk][0..N) = sum_m x[m,n], [N..2N) = sum_m x[m,n]^2 over the block's rows (N <= 256)
constexpr int CS_ROWS = 256;
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                       int64_t M, int N, int ld) {
  __shared__ float sh[2][256];
  const int tid = threadIdx.x, per = 256 / N;          // row lanes per column
  const int n = tid % N, r = tid / N;
  const int64_t m0 = (int64_t)blockIdx.x * CS_ROWS;
  float s = 0.f, q = 0.f;
  if (r < per)
    for (int64_t m = m0 + r; m < m0 + CS_ROWS && m < M; m += per) {
      const float v = x[m * ld + n];
      s += v;
      q += v * v;
    }
  sh[0][tid] = s;
  sh[1][tid] = q;
  __syncthreads();
  if (tid < N) {
    float a = 0.f, c = 0.f;
    for (int k = 0; k < per; ++k) {
      a += sh[0][k * N + tid];
      c += sh[1][k * N + tid];
    }
    part[(int64_t)blockIdx.x * 2 * N + tid] = a;
    part[(int64_t)blockIdx.x * 2 * N + N + tid] = c;
  }
}

// ------------------------------------------------------------------------------------ head
// o = scale*v + shift;  m[b,f,p] = mean_{k<WIN} o[b, p*STR+k, f]^2;  out = log(clamp(m, lo, hi)) * dropout
__global__ __launch_bounds__(256) void sqpool_log_fwd_kernel(const float* __restrict__ v, const float* __restrict__ bn,
                                                             float* __restrict__ pooled, float* __restrict__ out, int T,
                                                             int NF, int NP, int WIN, int STR, float lo, float hi,
                                                             float drop_p, uint64_t seed_in,
                                                             const uint8_t* __restrict__ mask,
                                                             const uint64_t* __restrict__ seed_dev) {
  const uint64_t seed = dropout_seed(seed_in, seed_dev);
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < NP * NF; i += 256) {
    const int p = i / NF, f = i - p * NF;               // f fastest: coalesced reads of v[b,t,:]
    const float sc = bn[2 * NF + f], sh = bn[3 * NF + f];
    float a = 0.f;
    for (int k = 0; k < WIN; ++k) {
      const float o = fmaf(sc, v[((int64_t)b * T + p * STR + k) * NF + f], sh);
      a = fmaf(o, o, a);
    }
    a /= (float)WIN;
    const int64_t oi = ((int64_t)b * NF + f) * NP + p;
    pooled[oi] = a;
    out[oi] = logf(fminf(fmaxf(a, lo), hi)) * dropout_mult(drop_p, seed, mask, oi);
  }
}
// g[b,t,f] = dL/do = (2/WIN) * o * sum_{p: window p covers t} dy[b,f,p] * drop / m[b,f,p]  (0 outside the clamp range)
// part[b][f] = sum_t g, part[b][NF+f] = sum_t g*xhat
__global__ __launch_bounds__(256) void sqpool_log_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pooled,
                                                             const float* __restrict__ v, const float* __restrict__ bn,
                                                             float* __restrict__ g, float* __restrict__ part, int T,
                                                             int NF, int NP, int WIN, int STR, float lo, float hi,
                                                             float drop_p, uint64_t seed_in,
                                                             const uint8_t* __restrict__ mask,
                                                             const uint64_t* __restrict__ seed_dev) {
  const uint64_t seed = dropout_seed(seed_in, seed_dev);
  extern __shared__ float dm[];                          // [NF][NP] then reduction scratch [2][256]
  float* red = dm + NF * NP;
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < NF * NP; i += 256) {
    const int64_t oi = (int64_t)b * NF * NP + i;
    const float m = pooled[oi];
    dm[i] = (m >= lo && m <= hi) ? dy[oi] * dropout_mult(drop_p, seed, mask, oi) / m : 0.f;
  }
  __syncthreads();
  const int per = 256 / NF, f = tid % NF, r = tid / NF;
  float s = 0.f, q = 0.f;
  if (r < per) {
    const float mean = bn[f], invstd = bn[NF + f], sc = bn[2 * NF + f], sh = bn[3 * NF + f];
    for (int t = r; t < T; t += per) {
      // windows p with p*STR <= t < p*STR + WIN
      int p1 = t / STR;
      if (p1 > NP - 1) p1 = NP - 1;
      int p0 = (t - WIN + STR) / STR;                    // ceil((t - WIN + 1) / STR) for t-WIN+1 >= 0
      if (t - WIN + 1 <= 0) p0 = 0;
      float a = 0.f;
      for (int p = p0; p <= p1; ++p) a += dm[f * NP + p];
      const float x = v[((int64_t)b * T + t) * NF + f];
      const float gv = a * (2.f / (float)WIN) * fmaf(sc, x, sh);
      g[((int64_t)b * T + t) * NF + f] = gv;
      s += gv;
      q += gv * (x - mean) * invstd;
    }
  }
  red[tid] = s;
  red[256 + tid] = q;
  __syncthreads();
  if (tid < NF) {
    float a = 0.f, c = 0.f;
    for (int k = 0; k < per; ++k) {
      a += red[k * NF + tid];
      c += red[256 + k * NF + tid];
    }
    part[(int64_t)b * 2 * NF + tid] = a;
    part[(int64_t)b * 2 * NF + NF + tid] = c;
  }
}
// BatchNorm input gradient on token-major rows: dx[m,f] = scale_f * (g - m1_f - xhat * m2_f)
__global__ void bn_rows_bwd_kernel(const float* __restrict__ g, const float* __restrict__ v, const float* __restrict__ bn,
                                   float* __restrict__ dx, int64_t n, int NF) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int f = (int)(i % NF);
    const float xh = (v[i] - bn[f]) * bn[NF + f];
    dx[i] = bn[2 * NF + f] * (g[i] - bn[4 * NF + f] - xh * bn[5 * NF + f]);
  }
}

inline int ew_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

// =============================================================================================== C ABI
static int embed_ok(const char* who, int B, int C, int S, int NF, int KC) {
  if (!(B > 0 && B <= 65535 && C >= 1 && C <= CHMAX && NF >= 1 && NF <= NFMAX && KC >= 1 && KC <= KCMAX && S >= KC))
    return eav_set_error(EAV_EINVAL, "%s: need Chans<=32, filters<=48, taps<=16, Samples>=taps (B=%d C=%d S=%d NF=%d KC=%d)",
                         who, B, C, S, NF, KC);
  return EAV_OK;
}

extern "C" int eav_shallow_embed_nparts(int B, int S) { return B * cdiv(S, BTS); }

extern "C" int eav_shallow_embed_fwd(const float* x, const float* wc, const float* wv, int ldv, float* u, float* v, int B,
                                     int C, int S, int NF, int KC, void* stream) {
  EAV_REQUIRE(x && wc && wv && u && v && ldv >= C, "eav_shallow_embed_fwd: null pointer or ldv < Chans");
  if (int rc = embed_ok("eav_shallow_embed_fwd", B, C, S, NF, KC)) return rc;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(cdiv(S, ETS), B), dim3(256), 0, (hipStream_t)stream, x, wc, wv, u, v, C, S,
                     NF, KC, ldv);
  EAV_CHECK_LAUNCH("eav_shallow_embed_fwd");
  return EAV_OK;
}

extern "C" int eav_shallow_embed_bwd(const float* dv, const float* x, const float* u, const float* wc, float* part_c,
                                     float* part_v, int B, int C, int S, int NF, int KC, void* stream) {
  EAV_REQUIRE(dv && x && u && wc && part_c && part_v, "eav_shallow_embed_bwd: null pointer");
  if (int rc = embed_ok("eav_shallow_embed_bwd", B, C, S, NF, KC)) return rc;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(cdiv(S, BTS), B), dim3(256), 0, (hipStream_t)stream, dv, x, u, wc, part_c,
                     part_v, C, S, NF, KC);
  EAV_CHECK_LAUNCH("eav_shallow_embed_bwd");
  return EAV_OK;
}

extern "C" int eav_relu_dropout(float* h, int64_t n, float drop_p, uint64_t seed, const uint8_t* mask,
                                const uint64_t* seed_dev, void* stream) {
  EAV_REQUIRE(h && n > 0 && drop_p >= 0.f && drop_p < 1.f, "eav_relu_dropout: bad arguments");
  hipLaunchKernelGGL(relu_dropout_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, h, n, drop_p, seed,
                     mask, seed_dev);
  EAV_CHECK_LAUNCH("eav_relu_dropout");
  return EAV_OK;
}

extern "C" int eav_relu_dropout_bwd(float* dact, const float* act, int64_t n, float drop_p, void* stream) {
  EAV_REQUIRE(dact && act && n > 0 && drop_p >= 0.f && drop_p < 1.f, "eav_relu_dropout_bwd: bad arguments");
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dact, act, n,
                     1.f / (1.f - drop_p));
  EAV_CHECK_LAUNCH("eav_relu_dropout_bwd");
  return EAV_OK;
}

extern "C" int eav_dropout_add(const float* y, const float* resid, float* out, int64_t n, float drop_p, uint64_t seed,
                               const uint8_t* mask, const uint64_t* seed_dev, void* stream) {
  EAV_REQUIRE(y && out && n > 0 && drop_p >= 0.f && drop_p < 1.f, "eav_dropout_add: bad arguments");
  hipLaunchKernelGGL(dropout_add_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, y, resid, out, n,
                     drop_p, seed, mask, seed_dev);
  EAV_CHECK_LAUNCH("eav_dropout_add");
  return EAV_OK;
}

extern "C" int eav_add_strided(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int64_t M, int n,
                               void* stream) {
  EAV_REQUIRE(a && out && M > 0 && n > 0 && lda >= n && ldo >= n && (!b || ldb >= n), "eav_add_strided: bad arguments");
  hipLaunchKernelGGL(add_strided_kernel, dim3(ew_blocks(M * n)), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, out,
                     ldo, M, n);
  EAV_CHECK_LAUNCH("eav_add_strided");
  return EAV_OK;
}

extern "C" int eav_colstats_nparts(int64_t M) { return (int)cdiv64(M, CS_ROWS); }

extern "C" int eav_colstats(const float* x, float* part, int64_t M, int N, int ld, void* stream) {
  EAV_REQUIRE(x && part && M > 0 && N >= 1 && N <= 256 && ld >= N, "eav_colstats: need 1 <= N <= 256");
  hipLaunchKernelGGL(colstats_kernel, dim3(eav_colstats_nparts(M)), dim3(256), 0, (hipStream_t)stream, x, part, M, N,
                     ld);
  EAV_CHECK_LAUNCH("eav_colstats");
  return EAV_OK;
}

static int pool_ok(const char* who, int B, int T, int NF, int NP, int WIN, int STR) {
  if (!(B > 0 && NF >= 1 && NF <= 256 && WIN >= 1 && STR >= 1 && NP >= 1 && (NP - 1) * STR + WIN <= T &&
        (size_t)(NF * NP + 512) * sizeof(float) <= 64 * 1024))
    return eav_set_error(EAV_EINVAL, "%s: bad pooling geometry (B=%d T=%d NF=%d NP=%d win=%d stride=%d)", who, B, T, NF,
                         NP, WIN, STR);
  return EAV_OK;
}

extern "C" int eav_sqpool_log_fwd(const float* v, const float* bn, float* pooled, float* out, int B, int T, int NF,
                                  int NP, int win, int stride, float lo, float hi, float drop_p, uint64_t seed,
                                  const uint8_t* mask, const uint64_t* seed_dev, void* stream) {
  EAV_REQUIRE(v && bn && pooled && out, "eav_sqpool_log_fwd: null pointer");
  if (int rc = pool_ok("eav_sqpool_log_fwd", B, T, NF, NP, win, stride)) return rc;
  hipLaunchKernelGGL(sqpool_log_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, v, bn, pooled, out, T, NF, NP,
                     win, stride, lo, hi, drop_p, seed, mask, seed_dev);
  EAV_CHECK_LAUNCH("eav_sqpool_log_fwd");
  return EAV_OK;
}

extern "C" int eav_sqpool_log_bwd(const float* dy, const float* pooled, const float* v, const float* bn, float* g,
                                  float* part, int B, int T, int NF, int NP, int win, int stride, float lo, float hi,
                                  float drop_p, uint64_t seed, const uint8_t* mask, const uint64_t* seed_dev,
                                  void* stream) {
  EAV_REQUIRE(dy && pooled && v && bn && g && part, "eav_sqpool_log_bwd: null pointer");
  if (int rc = pool_ok("eav_sqpool_log_bwd", B, T, NF, NP, win, stride)) return rc;
  hipLaunchKernelGGL(sqpool_log_bwd_kernel, dim3(B), dim3(256), (NF * NP + 512) * sizeof(float), (hipStream_t)stream,
                     dy, pooled, v, bn, g, part, T, NF, NP, win, stride, lo, hi, drop_p, seed, mask, seed_dev);
  EAV_CHECK_LAUNCH("eav_sqpool_log_bwd");
  return EAV_OK;
}

extern "C" int eav_bn_rows_bwd(const float* g, const float* v, const float* bn, float* dx, int64_t M, int NF,
                               void* stream) {
  EAV_REQUIRE(g && v && bn && dx && M > 0 && NF > 0, "eav_bn_rows_bwd: bad arguments");
  hipLaunchKernelGGL(bn_rows_bwd_kernel, dim3(ew_blocks(M * NF)), dim3(256), 0, (hipStream_t)stream, g, v, bn, dx,
                     M * NF, NF);
  EAV_CHECK_LAUNCH("eav_bn_rows_bwd");
  return EAV_OK;
}
