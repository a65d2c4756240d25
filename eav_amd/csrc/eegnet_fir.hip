// EEGNet temporal FIR (firstConv, K<=300 taps, 1 -> 8 filters) on the gfx950 fp32 matrix cores.
//
// Reference op: nn.Conv2d(1, F1=8, (1, kernLength=300), padding='same', bias=False)
// (CNN_torch/EEGNet_tor.py:24,51) and its weight gradient (autograd of the same call, :109).
// 'same' with an even kernel pads (K-1)/2 = 149 zeros left and 150 right.
//
// Forward as a Toeplitz GEMM on v_mfma_f32_32x32x2_f32 (exact f32 fma chain):
//   C[(f,s), n] = sum_j A[(f,s), j] * B[j, n]
//   A[(f,s), j] = w[f, j-s]            (8 filters x 4 shifts = 32 rows, j in [0,304))
//   B[j, n]     = xpad[t0 + 4n + j]    (32 columns, time stride 4)
//   => C[(f,s), n] = y[f, t0 + 4n + s]: one 32x32 tile = 8 filters x 128 consecutive samples,
//   152 MFMAs per tile (300/304 = 98.7 % useful).  A stays in registers for the whole kernel;
//   B is read from a polyphase (u mod 4) LDS image of the padded input row so that the 32 lanes
//   of a half-wave hit 32 consecutive banks.  The accumulator layout gives every lane four
//   consecutive samples of four filters -> float4 stores of y1, coalesced 512 B per half-wave.
//   BatchNorm batch statistics (sum, sum of squares per filter) are taken in the epilogue.
//
// Weight gradient as a GEMM on v_mfma_f32_16x16x4_f32:
//   C[(f,s), n] = sum_u A[(f,s), u] * B[u, n],  A[(f,s),u] = dy[f, u-s], B[u,n] = xpad[u + 2n]
//   => C[(f,s), n] = dW[f, k = 2n + s]; 8 filters x 2 shifts = 16 rows, 150 of 160 columns used.
//   dy (gradient w.r.t. the FIR output) is formed on the fly while staging, from the saved FIR
//   output y1 and g = dL/d(BN1 output): dy = scale*(g - m1 - yhat*m2)  (BatchNorm backward).
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

constexpr int F1 = 8;
constexpr int TILE = 128;         // samples per MFMA tile
constexpr int TPS = 16;           // tiles per LDS segment
// NSTEP = MFMA K-steps of 2 taps: (klen + 3 shifts) / 2 rounded up to even - 152 for the reference's 300 taps; shorter
// kernels (the canonical EEGNet's 64 taps, CNN_EEG.py:13) run the 34- or 66-step instantiation.

// ------------------------------------------------------------------------------------------ fwd
template <int NSTEP>
__global__ __launch_bounds__(256, 2) void fir_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                         float* __restrict__ y1, float* __restrict__ part, int rows,
                                                         int C, int S, int klen, int padl, int nseg, int ntiles,
                                                         const int64_t* __restrict__ xidx) {
  constexpr int HALO = 2 * NSTEP;                 // taps + shifts covered by one tile's window
  constexpr int SEG_M4 = TPS * 32 + NSTEP / 2;    // floats per polyphase plane
  __shared__ __attribute__((aligned(16))) float xs[4 * SEG_M4];
  __shared__ float red[4 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, kk = lane >> 5;

  // A operand, resident for the whole kernel: row i = f*4 + s, column j = 2p + kk
  float a[NSTEP];
  {
    const int f = n >> 2, s = n & 3;
#pragma unroll
    for (int p = 0; p < NSTEP; ++p) {
      int j = 2 * p + kk - s;
      a[p] = (j >= 0 && j < klen) ? w1[f * klen + j] : 0.f;
    }
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const bool vec = (S & 3) == 0;
  const int nwork = rows * nseg;
  // the padded input segment of the NEXT work item is fetched into registers before the MFMA phase of
  // the current one, so its HBM latency is hidden behind ~39k cycles of matrix work
  constexpr int NLD = (TPS * TILE + HALO + 255) / 256;
  float xr[NLD];
  auto fetch = [&](int work) {
    const int row = work / nseg, seg = work - row * nseg;
    const int tile0 = seg * TPS;
    const int nt = min(TPS, ntiles - tile0);
    const int useg0 = tile0 * TILE;
    // xidx: the batch is rows xidx[0..B) of a larger resident array (the trainer's data set) - no gathered copy
    const float* xrow = x + (xidx ? (xidx[row / C] * C + row % C) : (int64_t)row) * S;
    const int nload = nt * TILE + HALO;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int t = useg0 + idx - padl;
      xr[i] = (idx < nload && t >= 0 && t < S) ? xrow[t] : 0.f;
    }
  };
  if ((int)blockIdx.x < nwork) fetch(blockIdx.x);
  for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
    const int row = work / nseg, seg = work - row * nseg;
    const int tile0 = seg * TPS;
    const int nt = min(TPS, ntiles - tile0);
    const int useg0 = tile0 * TILE;
    // polyphase image: xs[(u&3)][u>>2], u local to the segment
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = threadIdx.x + 256 * i;
      if (idx < TPS * TILE + HALO) xs[(idx & 3) * SEG_M4 + (idx >> 2)] = xr[i];
    }
    __syncthreads();
    if (work + (int)gridDim.x < nwork) fetch(work + gridDim.x);
    const int b = row / C, c = row - b * C;
    for (int tile = wave; tile < nt; tile += 4) {
      const float* pe = xs + kk * SEG_M4 + tile * 32 + n;
      const float* po = xs + (2 + kk) * SEG_M4 + tile * 32 + n;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int p = 0; p < NSTEP; ++p) {
        float bv = (p & 1) ? po[p >> 1] : pe[p >> 1];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p], bv, acc, 0, 0, 0);
      }
      // C layout: col = lane&31 (n), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = f*4 + s
      const int t = useg0 + tile * TILE + 4 * n;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f = 2 * g + kk;
        float* dst = y1 + (((int64_t)b * F1 + f) * C + c) * S + t;
        float v0 = acc[4 * g + 0], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
        if (vec && t + 3 < S) {
          *reinterpret_cast<float4*>(dst) = make_float4(v0, v1, v2, v3);
        } else {
          if (t + 0 < S) dst[0] = v0; else v0 = 0.f;
          if (t + 1 < S) dst[1] = v1; else v1 = 0.f;
          if (t + 2 < S) dst[2] = v2; else v2 = 0.f;
          if (t + 3 < S) dst[3] = v3; else v3 = 0.f;
        }
        s1[g] += (v0 + v1) + (v2 + v3);
        s2[g] += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      }
    }
    __syncthreads();
  }
  // per-block partial statistics: lanes of one half hold filters f = 2g + kk
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float a1 = half_sum(s1[g]), a2 = half_sum(s2[g]);
    if (n == 0) {
      red[wave * 16 + 2 * g + kk] = a1;
      red[wave * 16 + 8 + 2 * g + kk] = a2;
    }
  }
  __syncthreads();
  if (threadIdx.x < 16)
    part[blockIdx.x * 16 + threadIdx.x] =
        (red[threadIdx.x] + red[16 + threadIdx.x]) + (red[32 + threadIdx.x] + red[48 + threadIdx.x]);
}

// ---------------------------------------------------------------------------------------- wgrad
// Wave-independent schedule: a work item = (row, 252-sample chunk) belongs to ONE wave, which stages its
// own LDS images (wave-private region, no block barrier in the loop), prefetches the next item's y1/g1/x
// into registers before its MFMA phase, and keeps its 16x160 accumulator tile in registers for its whole
// work list.  Waves of a CU drift apart, so one wave's staging overlaps the others' matrix work.
// WG_NT column tiles of 16 -> lags k = 2n+s, n < 16*WG_NT: 10 tiles for the reference's 300 taps, 2 / 4 for <= 64 / 128.
constexpr int WG_CW = 248;                // max u-samples per wave item (62 K-steps of 4; an even number of K-steps)
constexpr int WG_QS = 260;                // dy row stride: 260 == 4 (mod 32), rows 16-B aligned; >= WG_CW + 4 + 8

// PLAIN: BatchNorm in eval mode (running statistics are constants, m1 = m2 = 0): dy = scale * g, y1 is not read at all -
// the training mode of every epoch after the first in the reference's Trainer_uni.train().
template <int WG_NT, bool PLAIN>
__global__ __launch_bounds__(256, 3) void fir_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ y1, const float* __restrict__ g1,
    const float* __restrict__ bnp /* mean, invstd, scale, shift, m1, m2 (8 each) */, float* __restrict__ part, int rows,
    int C, int S, int klen, int padl, int nchunk, int CH, const int64_t* __restrict__ xidx) {
  constexpr int LAGS = 32 * WG_NT;                    // lags covered (>= klen)
  constexpr int WG_XW = WG_CW + 8 + LAGS;             // x window per item (incl. the look-ahead K-step)
  constexpr int WG_WAVE_LDS = F1 * WG_QS + WG_XW;     // floats per wave
  constexpr int RED = F1 * LAGS > 4 * WG_WAVE_LDS ? F1 * LAGS : 4 * WG_WAVE_LDS;
  __shared__ __attribute__((aligned(16))) float lds[RED];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* dyl = lds + wave * WG_WAVE_LDS;
  float* xl = dyl + F1 * WG_QS;
  const int col = lane & 15, kq = lane >> 4;
  const int af = col >> 1, as = col & 1;  // A row i = f*2 + s
  f32x4 acc[WG_NT];
#pragma unroll
  for (int i = 0; i < WG_NT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool vec = (S & 3) == 0;
  const int nq1 = CH / 4 + 1;               // quads per filter row incl. the one-quad left halo (<= 64)
  const int nwork = rows * nchunk;
  const int nwaves = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
  constexpr int NX = (WG_XW + 63) / 64;
  float4 ry[F1], rg[F1];
  float rx[NX];
  auto fetch = [&](int work) {
    const int row = work / nchunk, chunk = work - row * nchunk;
    const int c0 = chunk * CH;
    const int b = row / C, c = row - b * C;
    const int t = c0 - 4 + 4 * lane;        // lane = quad index q; register i = filter f
#pragma unroll
    for (int f = 0; f < F1; ++f) {
      ry[f] = rg[f] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane < nq1) {
        const int64_t base = (((int64_t)b * F1 + f) * C + c) * S;
        if (vec && t >= 0 && t + 3 < S) {
          if (!PLAIN) ry[f] = *reinterpret_cast<const float4*>(y1 + base + t);
          rg[f] = *reinterpret_cast<const float4*>(g1 + base + t);
        } else {
          float yv[4], gv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int te = t + e;
            const bool ok = te >= 0 && te < S;
            yv[e] = (ok && !PLAIN) ? y1[base + te] : 0.f;
            gv[e] = ok ? g1[base + te] : 0.f;
          }
          ry[f] = make_float4(yv[0], yv[1], yv[2], yv[3]);
          rg[f] = make_float4(gv[0], gv[1], gv[2], gv[3]);
        }
      }
    }
    const float* xrow = x + (xidx ? (xidx[b] * C + c) : (int64_t)row) * S;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = lane + 64 * i;
      const int tx = c0 + idx - padl;
      rx[i] = (idx < CH + LAGS && tx >= 0 && tx < S) ? xrow[tx] : 0.f;
    }
  };
  auto commit = [&](int work) {
    const int chunk = work % nchunk;
    const int t = chunk * CH - 4 + 4 * lane;
    if (lane < nq1) {
#pragma unroll
      for (int f = 0; f < F1; ++f) {
        const float mean = bnp[f], invstd = bnp[8 + f], sc = bnp[16 + f], m1 = bnp[32 + f], m2 = bnp[40 + f];
        const float yv[4] = {ry[f].x, ry[f].y, ry[f].z, ry[f].w};
        const float gv[4] = {rg[f].x, rg[f].y, rg[f].z, rg[f].w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int te = t + e;
          o[e] = (te >= 0 && te < S) ? (PLAIN ? sc * gv[e] : sc * (gv[e] - m1 - (yv[e] - mean) * invstd * m2)) : 0.f;
        }
        *reinterpret_cast<float4*>(&dyl[f * WG_QS + 4 * lane]) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int idx = lane + 64 * i;
      if (idx < CH + LAGS) xl[idx] = rx[i];
    }
  };
  if (gw < nwork) fetch(gw);
  const int ksteps = CH / 4;
  const float* ap = dyl + af * WG_QS + 4 + kq - as;
  const float* bp = xl + kq + 2 * col;
  for (int work = gw; work < nwork; work += nwaves) {
    commit(work);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS image is complete (wave-private)
    __builtin_amdgcn_wave_barrier();
    if (work + nwaves < nwork) fetch(work + nwaves);
    // software pipeline, two register sets: the 11 operands of the next K-step are read while the
    // 10 MFMAs of the current one run; sched_barrier pins loads-before-MFMAs.
    // ksteps is even (wgrad_geometry): the body is straight-line - a mid-loop exit makes the compiler's wait-count
    // pass fall back to lgkmcnt(0) at the loop head, which waits for the operands just requested for the NEXT step
    // as well.  The last iteration's look-ahead reads one K-step past the chunk (inside the wave's LDS image, unused).
    // Two operand sets, read one K-step ahead.  `landed()` (an empty asm that consumes the registers) makes the
    // compiler place the LDS wait for a set right AFTER the ten MFMAs of the other set have been issued - 320 cycles
    // after the reads went out - instead of in front of the MFMAs that follow the next set's reads, where its
    // conservative loop-carried lgkmcnt(0) would also wait for the reads just requested.
    float av0 = ap[0], bv0[WG_NT], av1, bv1[WG_NT];
#pragma unroll
    for (int nt = 0; nt < WG_NT; ++nt) bv0[nt] = bp[32 * nt];
    auto landed = [](float& a, float (&b)[WG_NT]) {
      asm volatile("" : "+v"(a));
#pragma unroll
      for (int nt = 0; nt < WG_NT; ++nt) asm volatile("" : "+v"(b[nt]));
    };
    landed(av0, bv0);
    for (int ks = 0; ks < ksteps; ks += 2) {
      av1 = ap[4 * (ks + 1)];
#pragma unroll
      for (int nt = 0; nt < WG_NT; ++nt) bv1[nt] = bp[4 * (ks + 1) + 32 * nt];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < WG_NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv0[nt], acc[nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      landed(av1, bv1);
      av0 = ap[4 * (ks + 2)];
#pragma unroll
      for (int nt = 0; nt < WG_NT; ++nt) bv0[nt] = bp[4 * (ks + 2) + 32 * nt];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < WG_NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv1[nt], acc[nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      landed(av0, bv0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all reads of this image retired before it is overwritten
    __builtin_amdgcn_wave_barrier();
  }
  // ---- combine the 4 waves' accumulators through LDS (fixed order), then one partial per block
  // C layout 16x16: col = lane&15, row = (lane>>4)*4 + reg = f*2 + s
  __syncthreads();
  float* red = lds;  // needs 8*LAGS floats
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int nt = 0; nt < WG_NT; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = kq * 4 + r;            // row = f*2 + s
          const int k = 2 * (16 * nt + col) + (i & 1);
          const int f = i >> 1;
          float* dst = red + f * LAGS + k;
          if (w == 0) *dst = acc[nt][r]; else *dst += acc[nt][r];
        }
      }
    }
    __syncthreads();
  }
  for (int idx = threadIdx.x; idx < F1 * klen; idx += 256) {
    const int f = idx / klen, k = idx - f * klen;
    part[(int64_t)blockIdx.x * (F1 * klen) + idx] = red[f * LAGS + k];
  }
}

// -------------------------------------------------------------------------------- inference: x -> block-1 output
// No-grad evaluation (Trainer_uni.validate, EEGNet_tor.py:118-135; BatchNorm on running statistics, no dropout):
//     p2[b, f*8+d, t/4] = mean_{e<4} ELU(bn2(sum_c w2[f*8+d, c] ELU(bn1(FIR_f(x[b,c,:]))[4(t/4)+e])))
// in ONE pass - the FIR output y1 [B,8,C,S] (614 MB at the benchmark shape), its ELU and the depthwise output z
// [B,64,S] never exist in memory.  Same Toeplitz MFMA tile as fir_fwd_kernel (8 filters x 128 samples per wave), but a
// work item is (sample b, 4 tiles = 512 samples, one per wave) and walks the C electrodes: each lane keeps the 4 filters x
// 8 depth multipliers x 4 samples of z it owns in 128 registers (one wave per SIMD, 512-register budget; the f32 MFMA
// reaches its issue rate from a single dependent accumulator chain), and since a lane's four samples are exactly one
// AvgPool(1,4) window, BN2 -> ELU -> pool is lane-local.  x rows are double-buffered in LDS (one barrier per electrode)
// with the next row's loads in flight during the MFMA phase.
// ELU for the inference kernel: expm1 as its Taylor polynomial on (-0.5, 0] (truncation < 6e-9) and v_exp_f32 - 1 below
// (absolute error < 1e-7, the result is in (-1, -0.39]) - a dozen VALU instructions against libm's ~40 with branches
__device__ __forceinline__ float elu_fast(float v) {
  const float x = fminf(v, 0.f);
  float p = 1.f / 40320.f;
  p = fmaf(p, x, 1.f / 5040.f);
  p = fmaf(p, x, 1.f / 720.f);
  p = fmaf(p, x, 1.f / 120.f);
  p = fmaf(p, x, 1.f / 24.f);
  p = fmaf(p, x, 1.f / 6.f);
  p = fmaf(p, x, 0.5f);
  p = fmaf(p, x, 1.f);
  p *= x;
  const float q = __expf(x) - 1.f;
  return v > 0.f ? v : (x > -0.5f ? p : q);
}

template <int NSTEP>
__global__ __launch_bounds__(256, 1) void fir_dw_infer_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                              const float* __restrict__ bn1, const float* __restrict__ w2,
                                                              const float* __restrict__ bn2, float* __restrict__ p2,
                                                              int B, int C, int S, int klen, int padl, int nseg,
                                                              const int64_t* __restrict__ xidx) {
  constexpr int TPB = 4;                          // tiles per work item: one per wave
  constexpr int HALO = 2 * NSTEP;
  constexpr int SEG_M4 = TPB * 32 + NSTEP / 2;    // floats per polyphase plane
  __shared__ __attribute__((aligned(16))) float xs[2][4 * SEG_M4];
  __shared__ __attribute__((aligned(16))) float w2s[32 * 64];      // [c][f*8+d]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, kk = lane >> 5;
  float a[NSTEP];
  {
    const int f = n >> 2, sft = n & 3;
#pragma unroll
    for (int p = 0; p < NSTEP; ++p) {
      const int j = 2 * p + kk - sft;
      a[p] = (j >= 0 && j < klen) ? w1[f * klen + j] : 0.f;
    }
  }
  for (int i = threadIdx.x; i < C * 64; i += 256) {
    const int c = i >> 6, fd = i & 63;
    w2s[i] = w2[fd * C + c];
  }
  float sc1[4], sh1[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    sc1[g] = bn1[2 * F1 + 2 * g + kk];
    sh1[g] = bn1[3 * F1 + 2 * g + kk];
  }
  const int T2 = S >> 2;
  const int nwork = B * nseg;
  constexpr int NLD = (TPB * TILE + HALO + 255) / 256;
  float xr[NLD];
  auto fetch = [&](int work, int c) {
    const int b = work / nseg, seg = work - b * nseg;
    const int useg0 = seg * TPB * TILE;
    const float* xrow = x + ((xidx ? xidx[b] : (int64_t)b) * C + c) * S;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = threadIdx.x + 256 * i;
      const int t = useg0 + idx - padl;
      xr[i] = (idx < TPB * TILE + HALO && t >= 0 && t < S) ? xrow[t] : 0.f;
    }
  };
  int it = 0;                                     // electrode iterations so far: LDS buffer = it & 1
  if ((int)blockIdx.x < nwork) fetch(blockIdx.x, 0);
  for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
    const int b = work / nseg, seg = work - b * nseg;
    const int t0 = (seg * TPB + wave) * TILE;      // first sample of this wave's tile
    float z[4][8][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int d = 0; d < 8; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) z[g][d][e] = 0.f;
    // One wave per SIMD: nothing else hides the VALU tail of an electrode (firstBN -> ELU -> 8 depth multipliers: 16 ELUs +
    // 128 FMAs per lane), so it is software-pipelined by hand - the tail of electrode c - 1 (accumulator accp) is cut into
    // 32 half-units of <= 12 VALU instructions, one after every 4th MFMA of electrode c: each sits in the 64-cycle shadow
    // of the dependent MFMA chain.
    f32x16 accp;
    float wv[8], ev = 0.f;
    auto tail_half = [&](int u, int cprev) {      // u in [0, 32): (g, e) = (u >> 3, (u >> 1) & 3), half = u & 1
      const int g = u >> 3, e = (u >> 1) & 3;
      if ((u & 7) == 0) {
        const float4 wa = *reinterpret_cast<const float4*>(w2s + cprev * 64 + (2 * g + kk) * 8);
        const float4 wb = *reinterpret_cast<const float4*>(w2s + cprev * 64 + (2 * g + kk) * 8 + 4);
        wv[0] = wa.x; wv[1] = wa.y; wv[2] = wa.z; wv[3] = wa.w; wv[4] = wb.x; wv[5] = wb.y; wv[6] = wb.z; wv[7] = wb.w;
      }
      if ((u & 1) == 0) {
        ev = elu_fast(fmaf(sc1[g], accp[4 * g + e], sh1[g]));
      } else {
#pragma unroll
        for (int d = 0; d < 8; ++d) z[g][d][e] = fmaf(wv[d], ev, z[g][d][e]);
      }
    };
    for (int c = 0; c < C; ++c, ++it) {
      float* buf = xs[it & 1];
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < TPB * TILE + HALO) buf[(idx & 3) * SEG_M4 + (idx >> 2)] = xr[i];
      }
      __syncthreads();      // (also orders the w2s fill before its first use; the other buffer was read an iteration ago)
      if (c + 1 < C) fetch(work, c + 1);
      else if (work + (int)gridDim.x < nwork) fetch(work + gridDim.x, 0);
      const float* pe = buf + kk * SEG_M4 + wave * 32 + n;
      const float* po = buf + (2 + kk) * SEG_M4 + wave * 32 + n;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      constexpr int EVERY = NSTEP >= 128 ? 4 : (NSTEP >= 64 ? 2 : 1);
#pragma unroll
      for (int p = 0; p < NSTEP; ++p) {
        const float bv = (p & 1) ? po[p >> 1] : pe[p >> 1];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p], bv, acc, 0, 0, 0);
        if (c > 0 && p % EVERY == EVERY - 1 && p / EVERY < 32) {
          __builtin_amdgcn_sched_barrier(0);
          tail_half(p / EVERY, c - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      accp = acc;
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) tail_half(u, C - 1);     // the last electrode's tail: nothing left to hide it behind
    // depthwiseBN (running statistics) -> ELU -> AvgPool(1,4): the lane's four samples are one pooling window
    const int tp = (t0 >> 2) + n;
    if (tp < T2) {
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          const int fd = (2 * g + kk) * 8 + d;
          const float s2 = bn2[2 * 64 + fd], h2 = bn2[3 * 64 + fd];
          const float v = (elu_f(fmaf(s2, z[g][d][0], h2)) + elu_f(fmaf(s2, z[g][d][1], h2))) +
                          (elu_f(fmaf(s2, z[g][d][2], h2)) + elu_f(fmaf(s2, z[g][d][3], h2)));
          p2[((int64_t)b * 64 + fd) * T2 + tp] = 0.25f * v;
        }
    }
  }
}

}  // namespace

static int fir_grid(int64_t nwork) { return nwork < 512 ? (int)nwork : 512; }
static int wgrad_grid(int64_t nwork) { return nwork < 768 ? (int)nwork : 768; }

extern "C" int eav_eegnet_fir_fwd_nparts(int B, int C, int S) {
  int ntiles = cdiv(S, TILE), nseg = cdiv(ntiles, TPS);
  return fir_grid((int64_t)B * C * nseg);
}

extern "C" int eav_eegnet_fir_fwd_indexed(const float* x, const int64_t* xidx, const float* w1, float* y1,
                                          float* stat_part, int B, int C, int S, int klen, void* stream);

extern "C" int eav_eegnet_fir_fwd(const float* x, const float* w1, float* y1, float* stat_part, int B, int C, int S,
                                  int klen, void* stream) {
  return eav_eegnet_fir_fwd_indexed(x, nullptr, w1, y1, stat_part, B, C, S, klen, stream);
}

// the batch = samples xidx[0..B) of x [*, C, S] (xidx == NULL: x is the batch itself)
extern "C" int eav_eegnet_fir_fwd_indexed(const float* x, const int64_t* xidx, const float* w1, float* y1,
                                          float* stat_part, int B, int C, int S, int klen, void* stream) {
  EAV_REQUIRE(x && w1 && y1 && stat_part && B > 0 && C > 0 && S > 0, "eav_eegnet_fir_fwd: bad arguments");
  EAV_REQUIRE(klen >= 1 && klen <= 300, "eav_eegnet_fir_fwd: kernLength %d outside [1,300]", klen);
  const int ntiles = cdiv(S, TILE), nseg = cdiv(ntiles, TPS);
  const int nwork = B * C * nseg;
#define EAV_FIR_FWD(NS)                                                                                         \
  hipLaunchKernelGGL(fir_fwd_kernel<NS>, dim3(fir_grid(nwork)), dim3(256), 0, (hipStream_t)stream, x, w1, y1,  \
                     stat_part, B * C, C, S, klen, (klen - 1) / 2, nseg, ntiles, xidx)
  if (klen <= 65) EAV_FIR_FWD(34);
  else if (klen <= 129) EAV_FIR_FWD(66);
  else EAV_FIR_FWD(152);
#undef EAV_FIR_FWD
  EAV_CHECK_LAUNCH("eav_eegnet_fir_fwd");
  return EAV_OK;
}

static void wgrad_geometry(int S, int* nchunk, int* CH) {
  int n = cdiv(S + 1, WG_CW);
  int ch = cdiv(cdiv(S + 1, n), 8) * 8;      // <= 248, multiple of 8: an even number of K-steps
  *nchunk = n;
  *CH = ch;
}

extern "C" int eav_eegnet_fir_wgrad_nparts(int B, int C, int S) {
  int nchunk, CH;
  wgrad_geometry(S, &nchunk, &CH);
  return wgrad_grid((int64_t)B * C * nchunk);
}

// part: [nparts][8][klen] floats; sum over parts = dL/d(firstConv.weight)
extern "C" int eav_eegnet_fir_wgrad_indexed(const float* x, const int64_t* xidx, const float* y1, const float* g1,
                                            const float* bn_params, float* part, int B, int C, int S, int klen,
                                            void* stream);

extern "C" int eav_eegnet_fir_wgrad(const float* x, const float* y1, const float* g1, const float* bn_params,
                                    float* part, int B, int C, int S, int klen, void* stream) {
  return eav_eegnet_fir_wgrad_indexed(x, nullptr, y1, g1, bn_params, part, B, C, S, klen, stream);
}

extern "C" int eav_eegnet_fir_wgrad_indexed(const float* x, const int64_t* xidx, const float* y1, const float* g1,
                                            const float* bn_params, float* part, int B, int C, int S, int klen,
                                            void* stream) {
  EAV_REQUIRE(x && g1 && bn_params && part && B > 0 && C > 0 && S > 0, "eav_eegnet_fir_wgrad: bad arguments");
  EAV_REQUIRE(klen >= 1 && klen <= 300, "eav_eegnet_fir_wgrad: kernLength %d outside [1,300]", klen);
  int nchunk, CH;
  wgrad_geometry(S, &nchunk, &CH);
  const int nwork = B * C * nchunk;
  // y1 == NULL selects the eval-mode form (dy = scale * g: the BatchNorm-backward means are zero, y1 is not needed)
#define EAV_FIR_WG(NT)                                                                                                \
  do {                                                                                                                \
    if (y1)                                                                                                           \
      hipLaunchKernelGGL((fir_wgrad_kernel<NT, false>), dim3(wgrad_grid(nwork)), dim3(256), 0, (hipStream_t)stream,   \
                         x, y1, g1, bn_params, part, B * C, C, S, klen, (klen - 1) / 2, nchunk, CH, xidx);            \
    else                                                                                                              \
      hipLaunchKernelGGL((fir_wgrad_kernel<NT, true>), dim3(wgrad_grid(nwork)), dim3(256), 0, (hipStream_t)stream, x, \
                         y1, g1, bn_params, part, B * C, C, S, klen, (klen - 1) / 2, nchunk, CH, xidx);               \
  } while (0)
  if (klen <= 64) EAV_FIR_WG(2);
  else if (klen <= 128) EAV_FIR_WG(4);
  else EAV_FIR_WG(10);
#undef EAV_FIR_WG
  EAV_CHECK_LAUNCH("eav_eegnet_fir_wgrad");
  return EAV_OK;
}

// No-grad forward of block 1 in eval mode: x [B,C,S] (or rows xidx of a resident data set) -> p2 [B,64,S/4] =
// AvgPool4(ELU(depthwiseBN(depthwiseConv(ELU(firstBN(firstConv(x))))))) with both BatchNorms on their running statistics
// (bn1 / bn2: mean, invstd, scale, shift as eav_bn_finalize writes them in eval mode).  S % 4 == 0, C <= 32.
extern "C" int eav_eegnet_block1_infer(const float* x, const int64_t* xidx, const float* w1, const float* bn1,
                                       const float* w2, const float* bn2, float* p2, int B, int C, int S, int klen,
                                       void* stream) {
  EAV_REQUIRE(x && w1 && bn1 && w2 && bn2 && p2 && B > 0 && C > 0 && C <= 32 && S >= 4 && (S & 3) == 0,
              "eav_eegnet_block1_infer: bad arguments (need C <= 32, S %% 4 == 0)");
  EAV_REQUIRE(klen >= 1 && klen <= 300, "eav_eegnet_block1_infer: kernLength %d outside [1,300]", klen);
  const int ntiles = cdiv(S, TILE), nseg = cdiv(ntiles, 4);
  const int64_t nwork = (int64_t)B * nseg;
  const int grid = nwork < 256 ? (int)nwork : 256;
#define EAV_FIR_INF(NS)                                                                                          \
  hipLaunchKernelGGL(fir_dw_infer_kernel<NS>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w1, bn1, w2, bn2, p2, \
                     B, C, S, klen, (klen - 1) / 2, nseg, xidx)
  if (klen <= 65) EAV_FIR_INF(34);
  else if (klen <= 129) EAV_FIR_INF(66);
  else EAV_FIR_INF(152);
#undef EAV_FIR_INF
  EAV_CHECK_LAUNCH("eav_eegnet_block1_infer");
  return EAV_OK;
}
