// Canonical EEGNet (CNN_torch/CNN_EEG.py:7-67) - the blocks that differ from EEGNet_tor.py:
//   block1: Conv2d(1,F1,(1,K1),'same') -> BN -> depthwise Conv2d(F1,D*F1,(Chans,1),groups=F1) -> BN -> ELU -> pool4
//   block2: depthwise Conv2d(C2,C2,(1,K2),'same',groups=C2) -> pointwise Conv2d(C2,F2,1) -> BN -> ELU -> pool8
// with run-time F1 <= 16, D <= 8, F2 <= 64, K1 <= 1024, K2 <= 32, Chans <= 256 (high-density montages, 2 kHz recordings).  The model is a few MFLOP per sample
// at its default size (64 x 128 input), so these are direct LDS-tiled kernels, not MFMA ones; BN -> ELU -> pool ->
// dropout and the classifier reuse eegnet_block.hip / head_optim.hip.  Reductions are two-stage and ordered
// (per-block partials + eav_reduce_partials / eav_bn_finalize): no float atomics, bit-reproducible.
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

constexpr int TT = 1024;   // tconv forward: time samples per block
constexpr int WT = 512;    // tconv wgrad: time samples per work item
constexpr int KMAX = 1024;
constexpr int CGMAX = 256;   // electrodes of the spatial (depthwise) kernels: their weight rows live in LDS
constexpr int ST = 128;    // separable conv forward: time samples per block
constexpr int PT = 64;     // pointwise backward: time samples per work item

// ------------------------------------------------------------------------------------ tconv_fwd
// y1[b,f,c,t] = sum_j w[f,j] * x[b,c,t+j-padl]; part[blk][0..F1) = sum y1, [F1..2F1) = sum y1^2 over the block.
template <int NG, int KM = KMAX>      // KM: taps the LDS images are sized for (512: the occupancy of the usual filters)
__global__ __launch_bounds__(256) void tconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ y1, float* __restrict__ part, int C,
                                                        int S, int F1, int K, int padl) {
  __shared__ float xs[TT + KM + 4];
  __shared__ __attribute__((aligned(16))) float ws[8 * NG][KM + 4];
  __shared__ float red[4 * 16];
  const int tile = blockIdx.x, c = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
  const int t0 = tile * TT, K4 = (K + 3) & ~3;
  const float* xr = x + ((int64_t)b * C + c) * S;
  for (int i = tid; i < TT + K4 + 3; i += 256) {
    int t = t0 + i - padl;
    xs[i] = (t >= 0 && t < S) ? xr[t] : 0.f;
  }
  for (int i = tid; i < 8 * NG * K4; i += 256) {
    int f = i / K4, j = i - f * K4;
    ws[f][j] = (f < F1 && j < K) ? w[f * K + j] : 0.f;
  }
  __syncthreads();
  const int blk = (b * C + c) * gridDim.x + tile;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    float acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int f = 0; f < 8; ++f) acc[i][f] = 0.f;
    for (int j = 0; j < K4; j += 4) {
      float xv[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[i][e] = xs[tid + 256 * i + j + e];
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const f32x4 wv = *(const f32x4*)&ws[g * 8 + f][j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][f] = fmaf(wv[e], xv[i][e], acc[i][f]);
      }
    }
    float st[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) st[k] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = t0 + tid + 256 * i;
      if (t < S) {
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          const int fg = g * 8 + f;
          if (fg < F1) {
            y1[(((int64_t)b * F1 + fg) * C + c) * S + t] = acc[i][f];
            st[f] += acc[i][f];
            st[8 + f] += acc[i][f] * acc[i][f];
          }
        }
      }
    }
    block_sum_256<16>(st, red);
    if (tid < 16) {
      const int fg = g * 8 + (tid & 7);
      if (fg < F1) part[(int64_t)blk * 2 * F1 + (tid >> 3) * F1 + fg] = st[0];
    }
  }
}

// ------------------------------------------------------------------------------------ tconv_wgrad
// dW[f,j] = sum_{b,c,t} dy[b,f,c,t] * x[b,c,t+j-padl] with the BatchNorm backward folded into the staging:
// dy = scale_f * (g1 - m1_f - xhat * m2_f), xhat = (y1 - mean_f) * invstd_f.   bn = mean, invstd, scale, shift,
// m1, m2 (F1 each).  Thread = (lag lane jl, time slice); accumulators live across the block's work items.
template <int NG, int NI>
__global__ __launch_bounds__(256) void tconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ y1,
                                                          const float* __restrict__ g1, const float* __restrict__ bn,
                                                          float* __restrict__ part, int B, int C, int S, int F1, int K,
                                                          int padl, int JW, int nitems) {
  constexpr int FW = 8 * NG;
  __shared__ float xs[WT + KMAX + 4];
  __shared__ __attribute__((aligned(16))) float dys[WT * FW];
  const int tid = threadIdx.x, jl = tid % JW, slice = tid / JW, nsl = 256 / JW;
  const int ntile = (S + WT - 1) / WT;
  float acc[NI][FW];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int f = 0; f < FW; ++f) acc[i][f] = 0.f;
  for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
    const int tile = it % ntile, bc = it / ntile, c = bc % C, b = bc / C;
    const int t0 = tile * WT;
    const float* xr = x + ((int64_t)b * C + c) * S;
    for (int i = tid; i < WT + KMAX; i += 256) {
      int t = t0 + i - padl;
      xs[i] = (t >= 0 && t < S) ? xr[t] : 0.f;
    }
    // dys[t][f]: thread = (t, quad of filters) so that the LDS stores are 16-byte and conflict-free
    for (int i = tid; i < WT * 2 * NG; i += 256) {
      const int q = i % (2 * NG), tl = i / (2 * NG), t = t0 + tl;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (t < S) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int f = 4 * q + e;
          if (f < F1) {
            const int64_t o = (((int64_t)b * F1 + f) * C + c) * S + t;
            const float xh = (y1[o] - bn[f]) * bn[F1 + f];
            v[e] = bn[2 * F1 + f] * (g1[o] - bn[4 * F1 + f] - xh * bn[5 * F1 + f]);
          }
        }
      }
      *(f32x4*)&dys[tl * FW + 4 * q] = v;
    }
    __syncthreads();
    const int tb = slice * (WT / nsl), te = tb + WT / nsl;
    for (int t = tb; t < te; ++t) {
      float dv[FW];
#pragma unroll
      for (int q = 0; q < 2 * NG; ++q) {
        const f32x4 v = *(const f32x4*)&dys[t * FW + 4 * q];
#pragma unroll
        for (int e = 0; e < 4; ++e) dv[4 * q + e] = v[e];
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const float xv = xs[t + jl + JW * i];
#pragma unroll
        for (int f = 0; f < FW; ++f) acc[i][f] = fmaf(dv[f], xv, acc[i][f]);
      }
    }
    __syncthreads();
  }
  // combine the time slices in a fixed order, then one partial row per block - two lag groups at a time: the exchange buffer
  // (the dy tile, WT * FW floats) holds 256 * 2 * FW
  float* red = dys;
#pragma unroll
  for (int ib = 0; ib < NI; ib += 2) {
#pragma unroll
    for (int i = ib; i < ib + 2 && i < NI; ++i)
#pragma unroll
      for (int f = 0; f < FW; ++f) red[((i - ib) * FW + f) * 256 + tid] = acc[i][f];
    __syncthreads();
    if (slice == 0) {
#pragma unroll
      for (int i = ib; i < ib + 2 && i < NI; ++i) {
        const int j = jl + JW * i;
#pragma unroll
        for (int f = 0; f < FW; ++f) {
          float s = 0.f;
          for (int sl = 0; sl < nsl; ++sl) s += red[((i - ib) * FW + f) * 256 + sl * JW + jl];
          if (f < F1 && j < K) part[(int64_t)blockIdx.x * F1 * K + f * K + j] = s;
        }
      }
    }
    if (ib + 2 < NI) __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ spatial_fwd
// z[b,f*D+d,t] = sum_c wd[f*D+d,c] * (scale_f * y1[b,f,c,t] + shift_f)      (BN1 affine folded in)
// part[(b,tile)][fd] = sum z, [C2+fd] = sum z^2.   A thread owns 4 consecutive samples (float4 loads and stores).
constexpr int SPT = 1024;   // samples per block of the spatial kernels

__device__ __forceinline__ void ld4(const float* p, int i, int n, bool vec, float (&v)[4]) {
  if (vec && i + 3 < n) {
    const float4 a = *reinterpret_cast<const float4*>(p + i);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (i + e < n) ? p[i + e] : 0.f;
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

__global__ __launch_bounds__(256) void spatial_fwd_kernel(const float* __restrict__ y1, const float* __restrict__ bn1,
                                                          const float* __restrict__ wd, float* __restrict__ z,
                                                          float* __restrict__ part, int C, int S, int F1, int D,
                                                          int elu) {
  __shared__ float wl[8][CGMAX];
  __shared__ float red[4 * 16];
  const int tile = blockIdx.x, f = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, C2 = F1 * D;
  for (int i = tid; i < 8 * C; i += 256) {
    int d = i / C, c = i - d * C;
    wl[d][c] = d < D ? wd[(f * D + d) * C + c] : 0.f;
  }
  __syncthreads();
  const float sc = bn1[2 * F1 + f], sh = bn1[3 * F1 + f];
  const int t = tile * SPT + 4 * tid;
  const bool vec = (S & 3) == 0;
  const float* src = y1 + ((int64_t)b * F1 + f) * C * S;
  float acc[8][4];
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[d][e] = 0.f;
  if (t < S) {
    for (int c = 0; c < C; ++c) {
      float v[4];
      ld4(src + (int64_t)c * S, t, S, vec, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float o = fmaf(sc, v[e], sh);
        v[e] = (t + e < S) ? (elu ? elu_f(o) : o) : 0.f;      // elu: EEGNet_tor.py:53 (BN -> ELU -> depthwise)
      }
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const float w = wl[d][c];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[d][e] = fmaf(w, v[e], acc[d][e]);
      }
    }
  }
  float st[16];
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    if (t < S && d < D) st4(z + ((int64_t)b * C2 + f * D + d) * S, t, S, vec, acc[d]);
    st[d] = (acc[d][0] + acc[d][1]) + (acc[d][2] + acc[d][3]);
    st[8 + d] = (acc[d][0] * acc[d][0] + acc[d][1] * acc[d][1]) + (acc[d][2] * acc[d][2] + acc[d][3] * acc[d][3]);
  }
  block_sum_256<16>(st, red);
  if (tid < 16 && (tid & 7) < D)
    part[((int64_t)b * gridDim.x + tile) * 2 * C2 + (tid >> 3) * C2 + f * D + (tid & 7)] = st[0];
}

// ------------------------------------------------------------------------------------ spatial_bwd
// g1[b,f,c,t] = sum_d wd[fd,c] * dz[b,fd,t]  (gradient w.r.t. the BN1 output);
// stat_part[(b,tile)][f] = sum g1, [F1+f] = sum g1*xhat;  w_part[(b,tile)][fd*C+c] = sum_t dz[fd,t] * bn1out[c,t].
// The per-(d,c) sums are taken over the thread's 4 samples first, then across the wave, then across the 4 waves.
__global__ __launch_bounds__(256) void spatial_bwd_kernel(const float* __restrict__ y1, const float* __restrict__ dz,
                                                          const float* __restrict__ bn1, const float* __restrict__ wd,
                                                          float* __restrict__ g1, float* __restrict__ stat_part,
                                                          float* __restrict__ w_part, int C, int S, int F1, int D,
                                                          int elu) {
  __shared__ float wl[8][CGMAX];
  __shared__ float wred[4][8][CGMAX];
  __shared__ float red[4 * 2];
  const int tile = blockIdx.x, f = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, C2 = F1 * D;
  const int lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 8 * C; i += 256) {
    int d = i / C, c = i - d * C;
    wl[d][c] = d < D ? wd[(f * D + d) * C + c] : 0.f;
  }
  __syncthreads();
  const float mean = bn1[f], invstd = bn1[F1 + f], sc = bn1[2 * F1 + f], sh = bn1[3 * F1 + f];
  const int t = tile * SPT + 4 * tid;
  const bool vec = (S & 3) == 0, live = t < S;
  float dzv[8][4];
#pragma unroll
  for (int d = 0; d < 8; ++d) {
#pragma unroll
    for (int e = 0; e < 4; ++e) dzv[d][e] = 0.f;
    if (live && d < D) ld4(dz + ((int64_t)b * C2 + f * D + d) * S, t, S, vec, dzv[d]);
  }
  const int64_t base = ((int64_t)b * F1 + f) * C * S;
  float st[2] = {0.f, 0.f};
  for (int c = 0; c < C; ++c) {
    float y[4] = {0.f, 0.f, 0.f, 0.f}, g[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
    if (live) ld4(y1 + base + (int64_t)c * S, t, S, vec, y);
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = fmaf(sc, y[e], sh);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float w = wl[d][c];
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = fmaf(w, dzv[d][e], g[e]);
    }
    if (elu) {      // the conv saw a = ELU(o): the weight gradient pairs dz with a, and g passes back through ELU'
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = elu_f(o[e]);
        g[e] *= elu_grad_from_out(o[e], a);
        o[e] = a;
      }
    }
    if (live) st4(g1 + base + (int64_t)c * S, t, S, vec, g);
#pragma unroll
    for (int e = 0; e < 4; ++e) {       // dz is 0 beyond S, so g and the products below vanish there
      st[0] += g[e];
      st[1] += g[e] * (y[e] - mean) * invstd;
    }
    for (int d = 0; d < D; ++d) {
      const float s = wave_sum((dzv[d][0] * o[0] + dzv[d][1] * o[1]) + (dzv[d][2] * o[2] + dzv[d][3] * o[3]));
      if (lane == 0) wred[wave][d][c] = s;
    }
  }
  block_sum_256<2>(st, red);   // contains the barriers that also publish wred
  const int64_t row = (int64_t)b * gridDim.x + tile;
  if (tid < 2) stat_part[row * 2 * F1 + tid * F1 + f] = st[0];
  for (int i = tid; i < D * C; i += 256) {
    const int d = i / C, c = i - d * C;
    w_part[row * C2 * C + (int64_t)(f * D + d) * C + c] = wred[0][d][c] + wred[1][d][c] + wred[2][d][c] + wred[3][d][c];
  }
}

// ------------------------------------------------------------------------------------ sepconv_fwd
// d3[b,ch,t] = sum_k wdw[ch,k] * a[b,ch,t+k-padl];  z[b,o,t] = sum_ch wp[o,ch] * d3[b,ch,t];
// part[(b,tile)][o] = sum z, [F2+o] = sum z^2.
__global__ __launch_bounds__(256) void sepconv_fwd_kernel(const float* __restrict__ a, const float* __restrict__ wdw,
                                                          const float* __restrict__ wp, float* __restrict__ d3,
                                                          float* __restrict__ z, float* __restrict__ part, int C2,
                                                          int F2, int T, int K2, int padl) {
  __shared__ float d3s[64][ST];
  __shared__ float wps[64][65];
  __shared__ float wds[64][32];
  __shared__ float red[4][64][2];
  const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, t0 = tile * ST;
  for (int i = tid; i < C2 * K2; i += 256) wds[i / K2][i % K2] = wdw[i];
  for (int i = tid; i < 64 * 64; i += 256) {
    int o = i >> 6, ch = i & 63;
    wps[o][ch] = (o < F2 && ch < C2) ? wp[o * C2 + ch] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < C2 * ST; i += 256) {
    const int tl = i % ST, ch = i / ST, t = t0 + tl;
    const float* src = a + ((int64_t)b * C2 + ch) * T;
    float s = 0.f;
    for (int k = 0; k < K2; ++k) {
      const int u = t + k - padl;
      if (u >= 0 && u < T) s = fmaf(wds[ch][k], src[u], s);
    }
    if (t < T) d3[((int64_t)b * C2 + ch) * T + t] = s;
    d3s[ch][tl] = t < T ? s : 0.f;
  }
  __syncthreads();
  const int tl = tid % ST, oh = tid / ST, t = t0 + tl, lane = tid & 63, wave = tid >> 6;
  for (int ob = oh * 4; ob < F2; ob += 8) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < C2; ++ch) {
      const float v = d3s[ch][tl];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = fmaf(wps[ob + i][ch], v, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (ob + i < F2) {
        if (t < T) z[((int64_t)b * F2 + ob + i) * T + t] = acc[i];
        const float s = wave_sum(acc[i]), q = wave_sum(acc[i] * acc[i]);
        if (lane == 0) {
          red[wave][ob + i][0] = s;
          red[wave][ob + i][1] = q;
        }
      }
    }
  }
  __syncthreads();
  // output channel o was handled by waves {0,1} (o%8 < 4) or {2,3}
  for (int i = tid; i < 2 * F2; i += 256) {
    const int o = i % F2, which = i / F2, w0 = ((o >> 2) & 1) * 2;
    part[((int64_t)b * gridDim.x + tile) * 2 * F2 + which * F2 + o] = red[w0][o][which] + red[w0 + 1][o][which];
  }
}

// ------------------------------------------------------------------------------------ pointwise_bwd
// dd3[b,ch,t] = sum_o wp[o,ch] * du[b,o,t];   part[blk][o*C2+ch] = sum over the block's items of du[o,t] * d3[ch,t]
__global__ __launch_bounds__(256) void pointwise_bwd_kernel(const float* __restrict__ du, const float* __restrict__ d3,
                                                            const float* __restrict__ wp, float* __restrict__ dd3,
                                                            float* __restrict__ part, int C2, int F2, int T,
                                                            int nitems) {
  __shared__ float dus[64][PT + 1];
  __shared__ float d3s[64][PT + 1];
  __shared__ float wps[64][65];
  const int tid = threadIdx.x, ntile = (T + PT - 1) / PT;
  for (int i = tid; i < 64 * 64; i += 256) {
    int o = i >> 6, ch = i & 63;
    wps[o][ch] = (o < F2 && ch < C2) ? wp[o * C2 + ch] : 0.f;
  }
  float acc[16];
#pragma unroll
  for (int m = 0; m < 16; ++m) acc[m] = 0.f;
  const int npair = F2 * C2;
  for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
    const int tile = it % ntile, b = it / ntile, t0 = tile * PT;
    __syncthreads();
    for (int i = tid; i < 64 * PT; i += 256) {
      const int tl = i % PT, r = i / PT, t = t0 + tl;
      dus[r][tl] = (r < F2 && t < T) ? du[((int64_t)b * F2 + r) * T + t] : 0.f;
      d3s[r][tl] = (r < C2 && t < T) ? d3[((int64_t)b * C2 + r) * T + t] : 0.f;
    }
    __syncthreads();
    {
      const int tl = tid % PT, q = tid / PT, t = t0 + tl;
      for (int ch = q; ch < C2; ch += 256 / PT) {
        float s = 0.f;
        for (int o = 0; o < F2; ++o) s = fmaf(wps[o][ch], dus[o][tl], s);
        if (t < T) dd3[((int64_t)b * C2 + ch) * T + t] = s;
      }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int p = tid + 256 * m;
      if (p < npair) {
        const int o = p / C2, ch = p - o * C2;
        float s = acc[m];
        for (int tl = 0; tl < PT; ++tl) s = fmaf(dus[o][tl], d3s[ch][tl], s);
        acc[m] = s;
      }
    }
  }
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const int p = tid + 256 * m;
    if (p < npair) part[(int64_t)blockIdx.x * npair + p] = acc[m];
  }
}

// ------------------------------------------------------------------------------------ dwt_bwd
// one block per (b, ch):  da[b,ch,t] = sum_k wdw[ch,k] * dd3[b,ch,t-k+padl];
// part[b][ch*K2+k] = sum_t dd3[b,ch,t] * a[b,ch,t+k-padl]
__global__ __launch_bounds__(256) void dwt_bwd_kernel(const float* __restrict__ dd3, const float* __restrict__ a,
                                                      const float* __restrict__ wdw, float* __restrict__ da,
                                                      float* __restrict__ part, int C2, int T, int K2, int padl) {
  __shared__ float red[4 * 32];
  __shared__ float wk[32];
  const int row = blockIdx.x, ch = row % C2, b = row / C2, tid = threadIdx.x;
  if (tid < 32) wk[tid] = tid < K2 ? wdw[ch * K2 + tid] : 0.f;
  __syncthreads();
  const float* g = dd3 + (int64_t)row * T;
  const float* src = a + (int64_t)row * T;
  float acc[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) acc[k] = 0.f;
  for (int t = tid; t < T; t += 256) {
    float s = 0.f;
    const float gv = g[t];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (k < K2) {
        const int ug = t - k + padl, ua = t + k - padl;
        if (ug >= 0 && ug < T) s = fmaf(wk[k], g[ug], s);
        if (ua >= 0 && ua < T) acc[k] = fmaf(gv, src[ua], acc[k]);
      }
    }
    da[(int64_t)row * T + t] = s;
  }
  block_sum_256<32>(acc, red);
  if (tid < K2) part[((int64_t)b * C2 + ch) * K2 + tid] = acc[0];
}

// ------------------------------------------------------------------------------------ dense temporal conv (generic)
// EEGNet_tor's "separableConv" is a DENSE Conv2d(F1*D -> F2, (1,16), groups=1, padding='same') (EEGNet_tor.py:37,59).
// The reference configuration (64 -> 64) runs on the fp32 matrix cores (eegnet_conv64.hip); any other width takes
// these direct kernels.  out[b,o,t] = sum_ci sum_k W(o,ci,k) in[b,ci,t+k-padl] with W(o,ci,k) = w[o so + ci si + k'],
// k' = kflip ? K-1-k : k - the data gradient is the same kernel on the transposed, tap-flipped weights (so = K,
// si = Cout K, kflip = 1, padl' = K-1-padl).  part[(b,tile)][o] = sum out, [Cout+o] = sum out^2 (BatchNorm statistics).
constexpr int DT = 64;    // output samples per block

__global__ __launch_bounds__(256) void dconv_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                        float* __restrict__ out, float* __restrict__ part, int Cin,
                                                        int Cout, int T, int K, int padl, int so, int si, int kflip) {
  __shared__ float in_s[64][DT + 16];
  __shared__ float w_s[64][8][17];
  const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, t0 = tile * DT;
  const int o = tid >> 2, q = tid & 3;
  for (int i = tid; i < Cin * (DT + 16); i += 256) {
    const int ci = i / (DT + 16), tl = i - ci * (DT + 16), u = t0 + tl - padl;
    in_s[ci][tl] = (u >= 0 && u < T) ? in[((int64_t)b * Cin + ci) * T + u] : 0.f;
  }
  float acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int c0 = 0; c0 < Cin; c0 += 8) {
    __syncthreads();      // in_s complete (first pass) / previous weight chunk consumed
    for (int i = tid; i < 64 * 8 * 16; i += 256) {
      const int oo = i >> 7, cc = (i >> 4) & 7, k = i & 15;
      w_s[oo][cc][k] = (oo < Cout && c0 + cc < Cin && k < K)
                           ? w[(int64_t)oo * so + (int64_t)(c0 + cc) * si + (kflip ? K - 1 - k : k)] : 0.f;
    }
    __syncthreads();
    for (int cc = 0; cc < 8 && c0 + cc < Cin; ++cc) {
      float win[31], wv[16];
#pragma unroll
      for (int e = 0; e < 31; ++e) win[e] = in_s[c0 + cc][16 * q + e];
#pragma unroll
      for (int k = 0; k < 16; ++k) wv[k] = w_s[o][cc][k];
#pragma unroll
      for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = fmaf(wv[k], win[e + k], acc[e]);
    }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int t = t0 + 16 * q + e;
    if (o < Cout && t < T) {
      out[((int64_t)b * Cout + o) * T + t] = acc[e];
      s1 += acc[e];
      s2 += acc[e] * acc[e];
    }
  }
  if (part) {
    s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
    s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
    if (q == 0 && o < Cout) {
      float* pr = part + ((int64_t)b * gridDim.x + tile) * 2 * Cout;
      pr[o] = s1;
      pr[Cout + o] = s2;
    }
  }
}

// weight gradient: part[b][o][ci][k] = sum_t dy[b,o,t] x[b,ci,t+k-padl]; a block owns (b, 4 output channels), a thread
// (ci, 4 taps); time is walked in tiles of 128 samples staged in LDS.
constexpr int DWT = 128;

__global__ __launch_bounds__(256) void dconv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          float* __restrict__ part, int Cin, int Cout, int T, int K,
                                                          int padl) {
  __shared__ float x_s[64][DWT + 16];
  __shared__ float dy_s[4][DWT];
  const int og = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int ci = tid >> 2, kq = tid & 3;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int t0 = 0; t0 < T; t0 += DWT) {
    __syncthreads();
    for (int i = tid; i < Cin * (DWT + 16); i += 256) {
      const int c = i / (DWT + 16), tl = i - c * (DWT + 16), u = t0 + tl - padl;
      x_s[c][tl] = (u >= 0 && u < T) ? x[((int64_t)b * Cin + c) * T + u] : 0.f;
    }
    for (int i = tid; i < 4 * DWT; i += 256) {
      const int oo = i / DWT, tl = i - oo * DWT, o = 4 * og + oo;
      dy_s[oo][tl] = (o < Cout && t0 + tl < T) ? dy[((int64_t)b * Cout + o) * T + t0 + tl] : 0.f;
    }
    __syncthreads();
    if (ci < Cin) {
      float xw[4];
#pragma unroll
      for (int j = 0; j < 3; ++j) xw[j + 1] = x_s[ci][4 * kq + j];
      for (int tl = 0; tl < DWT; ++tl) {
        xw[0] = xw[1]; xw[1] = xw[2]; xw[2] = xw[3];
        xw[3] = x_s[ci][tl + 4 * kq + 3];
#pragma unroll
        for (int oo = 0; oo < 4; ++oo) {
          const float g = dy_s[oo][tl];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[oo][j] = fmaf(g, xw[j], acc[oo][j]);
        }
      }
    }
  }
  if (ci < Cin)
#pragma unroll
    for (int oo = 0; oo < 4; ++oo) {
      const int o = 4 * og + oo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * kq + j;
        if (o < Cout && k < K) part[(((int64_t)b * Cout + o) * Cin + ci) * K + k] = acc[oo][j];
      }
    }
}

}  // namespace

// =============================================================================================== C ABI
static int tconv_ok(const char* who, int B, int C, int S, int F1, int K) {
  if (!(B > 0 && C > 0 && S > 0 && F1 >= 1 && F1 <= 16 && K >= 1 && K <= KMAX))
    return eav_set_error(EAV_EINVAL, "%s: need F1<=16, kernLength<=1024 (got B=%d C=%d S=%d F1=%d K=%d)", who, B, C, S,
                         F1, K);
  return EAV_OK;
}

// F1 == 8 with <= 300 taps runs on the fp32 matrix cores (the Toeplitz-GEMM kernels of eegnet_fir.hip, templated on
// the tap count); every other configuration takes the direct kernels of this file.
static bool use_mfma_fir(int F1, int K) { return F1 == 8 && K <= 300; }

extern "C" int eav_tconv_fwd_nparts(int B, int C, int S, int F1, int K) {
  return use_mfma_fir(F1, K) ? eav_eegnet_fir_fwd_nparts(B, C, S) : B * C * cdiv(S, TT);
}

extern "C" int eav_tconv_fwd(const float* x, const float* w, float* y1, float* stat_part, int B, int C, int S, int F1,
                             int K, void* stream) {
  EAV_REQUIRE(x && w && y1 && stat_part, "eav_tconv_fwd: null pointer");
  if (int rc = tconv_ok("eav_tconv_fwd", B, C, S, F1, K)) return rc;
  if (use_mfma_fir(F1, K)) return eav_eegnet_fir_fwd(x, w, y1, stat_part, B, C, S, K, stream);
  EAV_REQUIRE(C <= 65535 && B <= 65535, "eav_tconv_fwd: grid too large");
  const dim3 grid(cdiv(S, TT), C, B);
  const int padl = (K - 1) / 2;
#define EAV_TF(NG, KM)                                                                                                \
  hipLaunchKernelGGL((tconv_fwd_kernel<NG, KM>), grid, dim3(256), 0, (hipStream_t)stream, x, w, y1, stat_part, C, S, F1, \
                     K, padl)
  if (F1 <= 8 && K <= 512) EAV_TF(1, 512);
  else if (F1 <= 8) EAV_TF(1, 1024);
  else if (K <= 512) EAV_TF(2, 512);
  else EAV_TF(2, 1024);
#undef EAV_TF
  EAV_CHECK_LAUNCH("eav_tconv_fwd");
  return EAV_OK;
}

extern "C" int eav_tconv_wgrad_nparts(int B, int C, int S, int F1, int K) {
  if (use_mfma_fir(F1, K)) return eav_eegnet_fir_wgrad_nparts(B, C, S);
  const int64_t items = (int64_t)B * C * cdiv(S, WT);
  return (int)(items < 1024 ? items : 1024);
}

extern "C" int eav_tconv_wgrad(const float* x, const float* y1, const float* g1, const float* bn_params, float* part,
                               int B, int C, int S, int F1, int K, void* stream) {
  EAV_REQUIRE(x && y1 && g1 && bn_params && part, "eav_tconv_wgrad: null pointer");
  if (int rc = tconv_ok("eav_tconv_wgrad", B, C, S, F1, K)) return rc;
  if (use_mfma_fir(F1, K)) return eav_eegnet_fir_wgrad(x, y1, g1, bn_params, part, B, C, S, K, stream);
  const int nitems = B * C * cdiv(S, WT), nblk = eav_tconv_wgrad_nparts(B, C, S, F1, K);
  const int JW = K <= 64 ? 64 : (K <= 128 ? 128 : 256), NI = cdiv(K, JW), padl = (K - 1) / 2;
#define EAV_TW(NG, NI_)                                                                                              \
  hipLaunchKernelGGL((tconv_wgrad_kernel<NG, NI_>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, y1, g1,         \
                     bn_params, part, B, C, S, F1, K, padl, JW, nitems)
  if (F1 <= 8 && NI == 1) EAV_TW(1, 1);
  else if (F1 <= 8 && NI == 2) EAV_TW(1, 2);
  else if (F1 <= 8) EAV_TW(1, 4);             // (513 .. 1024 taps)
  else if (NI == 1) EAV_TW(2, 1);
  else if (NI == 2) EAV_TW(2, 2);
  else EAV_TW(2, 4);
#undef EAV_TW
  EAV_CHECK_LAUNCH("eav_tconv_wgrad");
  return EAV_OK;
}

static int spatial_ok(const char* who, int B, int C, int S, int F1, int D) {
  if (!(B > 0 && C >= 1 && C <= CGMAX && S > 0 && F1 >= 1 && F1 <= 16 && D >= 1 && D <= 8))
    return eav_set_error(EAV_EINVAL, "%s: need Chans<=256, F1<=16, D<=8 (got B=%d C=%d S=%d F1=%d D=%d)", who, B, C,
                         S, F1, D);
  return EAV_OK;
}

extern "C" int eav_spatial_nparts(int B, int S) { return B * cdiv(S, SPT); }

extern "C" int eav_spatial_fwd(const float* y1, const float* bn1, const float* wd, float* z, float* stat_part, int B,
                               int C, int S, int F1, int D, int elu, void* stream) {
  EAV_REQUIRE(y1 && bn1 && wd && z && stat_part, "eav_spatial_fwd: null pointer");
  if (int rc = spatial_ok("eav_spatial_fwd", B, C, S, F1, D)) return rc;
  hipLaunchKernelGGL(spatial_fwd_kernel, dim3(cdiv(S, SPT), F1, B), dim3(256), 0, (hipStream_t)stream, y1, bn1, wd, z,
                     stat_part, C, S, F1, D, elu);
  EAV_CHECK_LAUNCH("eav_spatial_fwd");
  return EAV_OK;
}

extern "C" int eav_spatial_bwd(const float* y1, const float* dz, const float* bn1, const float* wd, float* g1,
                               float* stat_part, float* w_part, int B, int C, int S, int F1, int D, int elu,
                               void* stream) {
  EAV_REQUIRE(y1 && dz && bn1 && wd && g1 && stat_part && w_part, "eav_spatial_bwd: null pointer");
  if (int rc = spatial_ok("eav_spatial_bwd", B, C, S, F1, D)) return rc;
  hipLaunchKernelGGL(spatial_bwd_kernel, dim3(cdiv(S, SPT), F1, B), dim3(256), 0, (hipStream_t)stream, y1, dz, bn1, wd,
                     g1, stat_part, w_part, C, S, F1, D, elu);
  EAV_CHECK_LAUNCH("eav_spatial_bwd");
  return EAV_OK;
}

static int sep_ok(const char* who, int B, int C2, int F2, int T, int K2) {
  if (!(B > 0 && C2 >= 1 && C2 <= 64 && F2 >= 1 && F2 <= 64 && T > 0 && K2 >= 1 && K2 <= 32))
    return eav_set_error(EAV_EINVAL, "%s: need D*F1<=64, F2<=64, taps<=32 (got B=%d C2=%d F2=%d T=%d K2=%d)", who, B,
                         C2, F2, T, K2);
  return EAV_OK;
}

extern "C" int eav_sepconv_fwd_nparts(int B, int T) { return B * cdiv(T, ST); }

extern "C" int eav_sepconv_fwd(const float* a, const float* wdw, const float* wp, float* d3, float* z,
                               float* stat_part, int B, int C2, int F2, int T, int K2, void* stream) {
  EAV_REQUIRE(a && wdw && wp && d3 && z && stat_part, "eav_sepconv_fwd: null pointer");
  if (int rc = sep_ok("eav_sepconv_fwd", B, C2, F2, T, K2)) return rc;
  hipLaunchKernelGGL(sepconv_fwd_kernel, dim3(cdiv(T, ST), B), dim3(256), 0, (hipStream_t)stream, a, wdw, wp, d3, z,
                     stat_part, C2, F2, T, K2, (K2 - 1) / 2);
  EAV_CHECK_LAUNCH("eav_sepconv_fwd");
  return EAV_OK;
}

extern "C" int eav_pointwise_bwd_nparts(int B, int T) {
  const int64_t items = (int64_t)B * cdiv(T, PT);
  return (int)(items < 512 ? items : 512);
}

extern "C" int eav_pointwise_bwd(const float* du, const float* d3, const float* wp, float* dd3, float* w_part, int B,
                                 int C2, int F2, int T, void* stream) {
  EAV_REQUIRE(du && d3 && wp && dd3 && w_part, "eav_pointwise_bwd: null pointer");
  if (int rc = sep_ok("eav_pointwise_bwd", B, C2, F2, T, 1)) return rc;
  hipLaunchKernelGGL(pointwise_bwd_kernel, dim3(eav_pointwise_bwd_nparts(B, T)), dim3(256), 0, (hipStream_t)stream, du,
                     d3, wp, dd3, w_part, C2, F2, T, B * cdiv(T, PT));
  EAV_CHECK_LAUNCH("eav_pointwise_bwd");
  return EAV_OK;
}

extern "C" int eav_dwt_bwd(const float* dd3, const float* a, const float* wdw, float* da, float* w_part, int B, int C2,
                           int T, int K2, void* stream) {
  EAV_REQUIRE(dd3 && a && wdw && da && w_part, "eav_dwt_bwd: null pointer");
  if (int rc = sep_ok("eav_dwt_bwd", B, C2, 1, T, K2)) return rc;
  hipLaunchKernelGGL(dwt_bwd_kernel, dim3(B * C2), dim3(256), 0, (hipStream_t)stream, dd3, a, wdw, da, w_part, C2, T,
                     K2, (K2 - 1) / 2);
  EAV_CHECK_LAUNCH("eav_dwt_bwd");
  return EAV_OK;
}

static int dconv_ok(const char* who, int B, int Cin, int Cout, int T, int K) {
  if (!(B > 0 && B <= 65535 && Cin >= 1 && Cin <= 64 && Cout >= 1 && Cout <= 64 && T > 0 && K >= 1 && K <= 16))
    return eav_set_error(EAV_EINVAL, "%s: need channels<=64, taps<=16 (got B=%d Cin=%d Cout=%d T=%d K=%d)", who, B, Cin,
                         Cout, T, K);
  return EAV_OK;
}

extern "C" int eav_dconv_fwd_nparts(int B, int T) { return B * cdiv(T, DT); }

extern "C" int eav_dconv_fwd(const float* in, const float* w, float* out, float* stat_part, int B, int Cin, int Cout,
                             int T, int K, int transposed, void* stream) {
  EAV_REQUIRE(in && w && out, "eav_dconv_fwd: null pointer");
  if (int rc = dconv_ok("eav_dconv_fwd", B, Cin, Cout, T, K)) return rc;
  // forward: w [Cout][Cin][K], left pad (K-1)/2 (torch 'same').  transposed (data gradient): `in` = dL/dout with Cin :=
  // the conv's output channels, `out` = dL/din with Cout := its input channels; w is still the FORWARD weight
  // [Cin][Cout][K] read transposed with flipped taps, left pad K-1-(K-1)/2
  const int padl = (K - 1) / 2;
  if (transposed)
    hipLaunchKernelGGL(dconv_fwd_kernel, dim3(cdiv(T, DT), B), dim3(256), 0, (hipStream_t)stream, in, w, out, stat_part,
                       Cin, Cout, T, K, K - 1 - padl, K, Cout * K, 1);
  else
    hipLaunchKernelGGL(dconv_fwd_kernel, dim3(cdiv(T, DT), B), dim3(256), 0, (hipStream_t)stream, in, w, out, stat_part,
                       Cin, Cout, T, K, padl, Cin * K, K, 0);
  EAV_CHECK_LAUNCH("eav_dconv_fwd");
  return EAV_OK;
}

// part [B][Cout*Cin*K]: finish with eav_reduce_partials(part, B, Cout*Cin*K, Cout*Cin*K, ...)
extern "C" int eav_dconv_wgrad(const float* dy, const float* x, float* part, int B, int Cin, int Cout, int T, int K,
                               void* stream) {
  EAV_REQUIRE(dy && x && part, "eav_dconv_wgrad: null pointer");
  if (int rc = dconv_ok("eav_dconv_wgrad", B, Cin, Cout, T, K)) return rc;
  hipLaunchKernelGGL(dconv_wgrad_kernel, dim3(cdiv(Cout, 4), B), dim3(256), 0, (hipStream_t)stream, dy, x, part, Cin, Cout,
                     T, K, (K - 1) / 2);
  EAV_CHECK_LAUNCH("eav_dconv_wgrad");
  return EAV_OK;
}
