// EEGNet "separableConv" (dense 64 -> 64 channels, 16 taps, 'same' padding 7 / 8 - nn.Conv2d(64, 64, (1,16), padding='same',
// bias=False), CNN_torch/EEGNet_tor.py:37,59) and its data gradient in the frequency domain, exact fp32 arithmetic.
//
// The direct form is a K = 1024 contraction per output (64 input channels x 16 taps): 20.97 GFLOP per pass at [64,64,2500],
// 0.18 ms on the fp32 matrix cores (eegnet_conv64.hip, 0.73-0.76 of the 157 TFLOP/s peak).  With overlap-save blocks of 64
// samples (49 valid outputs) the tap dimension disappears: per frequency bin the 64 x 64 channel mixing is ONE complex
// matrix-vector product,
//     Y_o[m] = sum_i conj(W_oi[m]) X_i[m]                (W_oi = FFT of the 16 taps of filter (o, i))
// i.e. 64 bins x a [128 x 128] real matrix instead of a [64 x 1024] one: 6 x fewer multiply-adds, and they are still
// plain GEMMs for the fp32 MFMA.  Two blocks of the same sample travel as ONE complex signal (block A real, block B
// imaginary): every step - FFT, complex-linear mixing, inverse FFT - keeps them apart (real filters), no unpacking.
//
// Three kernels + the filter spectra:
//   c64_spectra_kernel   BmT[bin][(ri_i, i)][(ri_o, o)] = the transpose of [[Gr, -Gi], [Gi, Gr]], G = conj(W) / 64 (forward), or the
//                        flipped / transposed filters of the data gradient
//   c64_pack_fft_kernel  a wave per column (sample b, block pair): coalesced row loads -> LDS transpose -> lane = channel,
//                        64-point complex FFT entirely in registers (8 x 8, no exchange) -> Z[bin][col][re | im][64 ch]
//   c64_bin_gemm_kernel  per bin C[cols x 128] = Z[cols x 128] . Bm^T on v_mfma_f32_32x32x2_f32, weight-stationary (a wave
//                        keeps its 32 outputs x 128 contraction rows of Bm in 64 VGPRs), column tiles through LDS
//   c64_ifft_unpack_kernel  lane = output channel: inverse FFT in registers, BatchNorm sums per channel (= per lane),
//                        LDS transpose, coalesced row stores
// Spectra layout [bin][col][2][64]: every global access of all three kernels is a whole 256-byte row segment.
#include <algorithm>

#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

constexpr int NCH = 64, KT = 16, NB = 64, LV = NB - KT + 1;      // 49 valid outputs per 64-sample block
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr float C64[50] = {1.0f, 0.995184727f, 0.98078528f, 0.956940336f, 0.923879533f, 0.881921264f, 0.831469612f, 0.773010453f, 0.707106781f, 0.634393284f, 0.555570233f, 0.471396737f, 0.382683432f, 0.290284677f, 0.195090322f, 0.0980171403f, 0.0f, -0.0980171403f, -0.195090322f, -0.290284677f, -0.382683432f, -0.471396737f, -0.555570233f, -0.634393284f, -0.707106781f, -0.773010453f, -0.831469612f, -0.881921264f, -0.923879533f, -0.956940336f, -0.98078528f, -0.995184727f, -1.0f, -0.995184727f, -0.98078528f, -0.956940336f, -0.923879533f, -0.881921264f, -0.831469612f, -0.773010453f, -0.707106781f, -0.634393284f, -0.555570233f, -0.471396737f, -0.382683432f, -0.290284677f, -0.195090322f, -0.0980171403f, 0.0f, 0.0980171403f};
constexpr float S64[50] = {0.0f, 0.0980171403f, 0.195090322f, 0.290284677f, 0.382683432f, 0.471396737f, 0.555570233f, 0.634393284f, 0.707106781f, 0.773010453f, 0.831469612f, 0.881921264f, 0.923879533f, 0.956940336f, 0.98078528f, 0.995184727f, 1.0f, 0.995184727f, 0.98078528f, 0.956940336f, 0.923879533f, 0.881921264f, 0.831469612f, 0.773010453f, 0.707106781f, 0.634393284f, 0.555570233f, 0.471396737f, 0.382683432f, 0.290284677f, 0.195090322f, 0.0980171403f, 0.0f, -0.0980171403f, -0.195090322f, -0.290284677f, -0.382683432f, -0.471396737f, -0.555570233f, -0.634393284f, -0.707106781f, -0.773010453f, -0.831469612f, -0.881921264f, -0.923879533f, -0.956940336f, -0.98078528f, -0.995184727f, -1.0f, -0.995184727f};

__device__ __forceinline__ v2f cmul(v2f a, v2f b) { return a.xx * b + a.yy * (v2f){-b.y, b.x}; }
__device__ __forceinline__ v2f cmulc(v2f a, v2f b) { return a.xx * (v2f){b.x, -b.y} + a.yy * (v2f){b.y, b.x}; }
template <bool INV>
__device__ __forceinline__ v2f twmul(v2f a, v2f w) { return INV ? cmulc(a, w) : cmul(a, w); }
template <bool INV>
__device__ __forceinline__ v2f rot(v2f a) { return INV ? (v2f){-a.y, a.x} : (v2f){a.y, -a.x}; }      // x (-i) / x (+i)

template <bool INV>
__device__ __forceinline__ void dft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
  const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = rot<INV>(a1 - a3);
  a0 = t0 + t2; a1 = t1 + t3; a2 = t0 - t2; a3 = t1 - t3;
}

// 8-point DFT, natural order in and out
template <bool INV>
__device__ __forceinline__ void dft8(v2f& a0, v2f& a1, v2f& a2, v2f& a3, v2f& a4, v2f& a5, v2f& a6, v2f& a7) {
  constexpr float R = 0.70710678118654752f;
  dft4<INV>(a0, a2, a4, a6);      // E0..E3 in a0, a2, a4, a6
  dft4<INV>(a1, a3, a5, a7);      // O0..O3 in a1, a3, a5, a7
  const v2f t0 = a1, t1 = twmul<INV>(a3, (v2f){R, -R}), t2 = rot<INV>(a5), t3 = twmul<INV>(a7, (v2f){-R, -R});
  const v2f e0 = a0, e1 = a2, e2 = a4, e3 = a6;
  a0 = e0 + t0; a1 = e1 + t1; a2 = e2 + t2; a3 = e3 + t3;
  a4 = e0 - t0; a5 = e1 - t1; a6 = e2 - t2; a7 = e3 - t3;
}

// 64-point complex FFT of one lane's registers: x[n] in, X[k] out at x[pos64(k)] (8 x 8, digit-reversed output: all
// indices are compile-time constants, the permutation costs nothing).  INV: conjugate twiddles, no 1/N.
__host__ __device__ constexpr int pos64(int k) { return 8 * (k & 7) + (k >> 3); }

template <bool INV>
__device__ __forceinline__ void fft64(v2f (&x)[64]) {
#pragma unroll
  for (int n2 = 0; n2 < 8; ++n2)
    dft8<INV>(x[n2], x[8 + n2], x[16 + n2], x[24 + n2], x[32 + n2], x[40 + n2], x[48 + n2], x[56 + n2]);
#pragma unroll
  for (int k1 = 1; k1 < 8; ++k1)
#pragma unroll
    for (int n2 = 1; n2 < 8; ++n2) x[8 * k1 + n2] = twmul<INV>(x[8 * k1 + n2], (v2f){C64[n2 * k1], -S64[n2 * k1]});
#pragma unroll
  for (int k1 = 0; k1 < 8; ++k1)
    dft8<INV>(x[8 * k1], x[8 * k1 + 1], x[8 * k1 + 2], x[8 * k1 + 3], x[8 * k1 + 4], x[8 * k1 + 5], x[8 * k1 + 6],
              x[8 * k1 + 7]);
}

struct Geo {
  int B, T, padl, nblk, npair, ncol, ncolp;      // ncolp = columns rounded up to 128: whole 32-column GEMM tiles and 8 equal
                                                 // weight-gradient chunks of a multiple of 16 columns; columns >= ncol are ZERO
};

Geo geometry(int B, int T, int padl) {
  Geo g;
  g.B = B; g.T = T; g.padl = padl;
  g.nblk = cdiv(T, LV);
  g.npair = cdiv(g.nblk, 2);
  g.ncol = B * g.npair;
  g.ncolp = cdiv(g.ncol, 128) * 128;
  return g;
}

// ---------------------------------------------------------------------------------------------------------- filter spectra
// BmT[bin][k = (ri_i, in)][n = (ri_o, out)] (contraction index major: the GEMM's weight-stationary waves then load their
// operand registers as whole 128-byte rows).  One workgroup (one wave) per input channel `in`, lane = output channel `out`.
// Table 0 (forward): filter (out, in) = w[out][in][:]; table 1 (data gradient: dp2[i] = sum_o w'[i][o] * du[o],
// w'[i][o][k'] = w[o][i][15 - k']): out = i, in = o.
__global__ __launch_bounds__(64) void c64_spectra_kernel(const float* __restrict__ w, float* __restrict__ Bm0, int bwd0) {
  const int in = blockIdx.x, lane = threadIdx.x;
  const int bwd = bwd0 + blockIdx.y;                      // grid.y = 2: the forward's table, then the data gradient's
  float* Bm = Bm0 + (int64_t)blockIdx.y * 64 * 128 * 128;
  v2f x[64];
#pragma unroll
  for (int n = 0; n < 64; ++n) {
    float t = 0.f;
    if (n < KT) t = bwd ? w[((int64_t)in * NCH + lane) * KT + (KT - 1 - n)] : w[((int64_t)lane * NCH + in) * KT + n];
    x[n] = (v2f){t, 0.f};
  }
  fft64<false>(x);
#pragma unroll
  for (int m = 0; m < 64; ++m) {
    const v2f W = x[pos64(m)];
    const float gr = W.x * (1.0f / NB), gi = -W.y * (1.0f / NB);      // G = conj(W) / 64
    float* row0 = Bm + ((int64_t)m * 128 + in) * 128;                 // contraction row (re, in)
    float* row1 = Bm + ((int64_t)m * 128 + 64 + in) * 128;            // contraction row (im, in)
    row0[lane] = gr;   row0[64 + lane] = gi;                          // y_re += gr z_re,  y_im += gi z_re
    row1[lane] = -gi;  row1[64 + lane] = gr;                          // y_re -= gi z_im,  y_im += gr z_im
  }
}

// ------------------------------------------------------------------------------------------------------------ pack + FFT
// A wave per column.  Rows of 64 samples are loaded coalesced (lane = sample), transposed through a [64][65] LDS tile and
// read back with lane = channel (bank (lane + n) mod 32: conflict-free); block A -> real parts, block B -> imaginary parts.
constexpr int TS = 65;

// vonly: the block is its 49 valid samples, zero-padded (the du operand of the weight gradient), instead of 64 samples
// starting padl before the block (forward / data gradient inputs).
__global__ __launch_bounds__(256) void c64_pack_fft_kernel(const float* __restrict__ in, float* __restrict__ Z, Geo g,
                                                           int vonly) {
  __shared__ float tiles[4][64 * TS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* tile = tiles[wave];
  const int nwaves = gridDim.x * 4;
  for (int col = blockIdx.x * 4 + wave; col < g.ncolp; col += nwaves) {
    if (col >= g.ncol) {                       // padding column: zero spectra (the weight gradient contracts over columns)
      float* dz = Z + (int64_t)col * 128 + lane;
#pragma unroll 8
      for (int m = 0; m < 64; ++m) {
        dz[(int64_t)m * g.ncolp * 128] = 0.f;
        dz[(int64_t)m * g.ncolp * 128 + 64] = 0.f;
      }
      continue;
    }
    const int b = col / g.npair, pr = col - b * g.npair;
    v2f x[64];
    // all 128 row loads of the column are issued before the first one is consumed (8 at a time left the kernel waiting on
    // HBM latency 16 times per column: 45 us; x[] doubles as the landing zone)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int blk = 2 * pr + half;
      const int t = blk * LV - (vonly ? 0 : g.padl) + lane;
      const bool ok = blk < g.nblk && t >= 0 && t < g.T && (!vonly || lane < LV);
      const float* src = in + (int64_t)b * NCH * g.T + t;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const float v = ok ? src[(int64_t)ch * g.T] : 0.f;
        if (half == 0) x[ch].x = v; else x[ch].y = v;
      }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) tile[ch * TS + lane] = half == 0 ? x[ch].x : x[ch].y;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float r[64];
#pragma unroll
      for (int n = 0; n < 64; ++n) r[n] = tile[lane * TS + n];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int n = 0; n < 64; ++n) {
        if (half == 0) x[n].x = r[n]; else x[n].y = r[n];
      }
    }
    fft64<false>(x);
    float* dst = Z + (int64_t)col * 128 + lane;
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      dst[(int64_t)m * g.ncolp * 128] = x[pos64(m)].x;
      dst[(int64_t)m * g.ncolp * 128 + 64] = x[pos64(m)].y;
    }
  }
}

// ------------------------------------------------------------------------------------------------------- per-bin GEMM
// C[bin][col][n] = sum_k Z[bin][col][k] BmT[bin][k][n], n, k in [0,128).  grid (wgs per bin, 64 bins), 4 waves; wave w owns
// outputs [32 w, 32 w + 32): its Bm rows stay in 64 VGPRs (B operand of v_mfma_f32_32x32x2_f32: lane = (k & 1, n)).  Column
// tiles of 32 are staged into LDS (row stride 129: the A-operand reads hit 32 banks), next tile prefetched in registers.
constexpr int XS = 129;

#ifndef C64V_WPE        // A/B builds: minimum waves per SIMD asked of the register allocator for the per-bin GEMM (0 = none)
#define C64V_WPE 0
#endif
__global__ __launch_bounds__(256, C64V_WPE) void c64_bin_gemm_kernel(const float* __restrict__ Z, const float* __restrict__ Bm,
                                                           float* __restrict__ C, int ncolp) {
  __shared__ float xs[2][32 * XS];
  const int bin = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, kk = lane >> 5;
  float wreg[64];
  {
    const float* bp = Bm + ((int64_t)bin * 128 + kk) * 128 + 32 * wave + n;      // BmT[bin][k][n]: 128-byte rows per half-wave
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) wreg[ks] = bp[(int64_t)2 * ks * 128];
  }
  const int ntiles = ncolp >> 5;
  const float* zb = Z + (int64_t)bin * ncolp * 128;
  float* cb = C + (int64_t)bin * ncolp * 128;
  float4 pre[4];
  auto fetch = [&](int tile) {
    const float4* src = reinterpret_cast<const float4*>(zb + (int64_t)tile * 32 * 128);
#pragma unroll
    for (int i = 0; i < 4; ++i) pre[i] = src[threadIdx.x + 256 * i];
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx4 = threadIdx.x + 256 * i, col = idx4 >> 5, k4 = idx4 & 31;
      float* d = xs[buf] + col * XS + 4 * k4;
      d[0] = pre[i].x; d[1] = pre[i].y; d[2] = pre[i].z; d[3] = pre[i].w;
    }
  };
  int tile = blockIdx.x, buf = 0;
  if (tile < ntiles) {
    fetch(tile);
    commit(0);
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int nxt = tile + gridDim.x;
#ifndef C64V_ABL_NOFETCH
    if (nxt < ntiles) fetch(nxt);
#endif
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // the tile's 64 A operands are read before the first MFMA (left to itself hipcc reads them pairwise just in time:
    // an LDS round trip in front of every second MFMA)
    const float* ap = xs[buf] + n * XS + kk;
    float av[64];
#pragma unroll
#ifdef C64V_ABL_NOLDS
    for (int ks = 0; ks < 64; ++ks) av[ks] = 1.0f + ks;
#else
    for (int ks = 0; ks < 64; ++ks) av[ks] = ap[2 * ks];
#endif
    __builtin_amdgcn_sched_barrier(0);            // (the scheduler otherwise sinks the reads back between the MFMAs)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks], wreg[ks], acc, 0, 0, 0);
    // D[m = column][n = output]: lane holds output n, registers = columns (r & 3) + 8 (r >> 2) + 4 kk
    float* dst = cb + ((int64_t)tile * 32) * 128 + 32 * wave + n;
#ifdef C64V_ABL_NOSTORE      // (timing-only ablation)
    if (acc[0] == 12345.f)
#endif
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * kk) * 128] = acc[r];
#ifndef C64V_ABL_NOFETCH
    if (nxt < ntiles) commit(buf ^ 1);
#endif
    // raw barrier: __syncthreads() would also wait for this tile's 16 global stores per lane (vmcnt(0)) at every tile
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    buf ^= 1;
  }
}

// ------------------------------------------------------------------------------------------------------- IFFT + unpack
// A wave per column, lane = output channel.  stat_part (optional): [waves][128] - sums and sums of squares per channel.
__global__ __launch_bounds__(256) void c64_ifft_unpack_kernel(const float* __restrict__ Y, float* __restrict__ out,
                                                              float* __restrict__ part, Geo g) {
  __shared__ float tiles[4][64 * TS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* tile = tiles[wave];
  const int nwaves = gridDim.x * 4, wg = blockIdx.x * 4 + wave;
  float s1 = 0.f, s2 = 0.f;
  for (int col = wg; col < g.ncol; col += nwaves) {
    const int b = col / g.npair, pr = col - b * g.npair;
    v2f x[64];
    const float* src = Y + (int64_t)col * 128 + lane;
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      x[m].x = src[(int64_t)m * g.ncolp * 128];
      x[m].y = src[(int64_t)m * g.ncolp * 128 + 64];
    }
    fft64<true>(x);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int blk = 2 * pr + half, t0 = blk * LV;
      if (blk >= g.nblk) break;
#pragma unroll
      for (int j = 0; j < LV; ++j) {
        const float v = half == 0 ? x[pos64(j)].x : x[pos64(j)].y;
        tile[lane * TS + j] = v;
        if (t0 + j < g.T) {
          s1 += v;
          s2 += v * v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float* dst = out + (int64_t)b * NCH * g.T + t0 + lane;
      const bool ok = lane < LV && t0 + lane < g.T;
#pragma unroll 8
      for (int ch = 0; ch < NCH; ++ch) {
        const float v = tile[ch * TS + lane];
        if (ok) dst[(int64_t)ch * g.T] = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if (part) {
    part[(int64_t)wg * 128 + lane] = s1;
    part[(int64_t)wg * 128 + 64 + lane] = s2;
  }
}


// ------------------------------------------------------------------------------------------------------- weight gradient
// dW[o][i][k] = sum_{b,t} du[b,o,t] in[b,i,t+k-7] = Re IFFT_m( sum_cols conj(D_o[m]) Z_i[m] )[k], D = spectra of the
// zero-padded 49-sample blocks of du (two blocks per column: real + imaginary), Z = the forward's input spectra.  Per bin
// one real [128 x cols]^T [cols x 128] product P[a][b] = sum_col D[col][a] Z[col][b] (a = (re|im, o), b = (re|im, i)):
//   Re Acc[o][i] = P[(re,o)][(re,i)] + P[(im,o)][(im,i)],   Im Acc[o][i] = P[(re,o)][(im,i)] - P[(im,o)][(re,i)].
// c64_bin_wgemm_kernel: grid (column chunks, 64 bins), 4 waves; wave w = rows a in [32 w, 32 w + 32), all 128 columns b
// (four 32 x 32 accumulators); both operands are read straight from global memory - for a fixed column the 32 lanes of a
// half-wave read 128 consecutive bytes - eight K-steps ahead; the chunk's partial product goes to Pp[chunk][bin].
// c64_wfinish_kernel: lane = i, workgroup = o: sums the chunks in order, forms Acc, inverse FFT over the bins, taps 0..15.
#ifndef C64V_WCH
#define C64V_WCH 4
#endif
#ifndef C64V_PF
#define C64V_PF 8
#endif
#ifndef C64V_GW
#define C64V_GW 8
#endif
constexpr int WCH = C64V_WCH;      // column chunks (split-K) of the weight-gradient GEMM

__global__ __launch_bounds__(256) void c64_bin_wgemm_kernel(const float* __restrict__ D, const float* __restrict__ Z,
                                                            float* __restrict__ Pp, int ncolp, int cpc) {
  const int bin = blockIdx.y, chunk = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, kk = lane >> 5;
  const int c0 = chunk * cpc;                  // cpc: a multiple of 16 columns, all inside the zero-padded buffers
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float* dp = D + ((int64_t)bin * ncolp + c0 + kk) * 128 + 32 * wave + n;
  const float* zp = Z + ((int64_t)bin * ncolp + c0 + kk) * 128 + n;
  const int nks = cpc >> 1;                    // K-steps of 2 columns: a multiple of PF
  constexpr int PF = C64V_PF;
  float ra[PF], rb[PF][4];
#pragma unroll
  for (int p = 0; p < PF; ++p) {
    ra[p] = dp[(int64_t)p * 256];
#pragma unroll
    for (int j = 0; j < 4; ++j) rb[p][j] = zp[(int64_t)p * 256 + 32 * j];
  }
  for (int ks0 = 0; ks0 < nks; ks0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const float a = ra[p];
      const float b0 = rb[p][0], b1 = rb[p][1], b2 = rb[p][2], b3 = rb[p][3];
      // unconditional reload, clamped to the chunk's last K-step (a branch here made hipcc copy the whole register window)
#ifndef C64V_ABL_NOLOAD      // (timing-only ablation: the MFMA side alone)
      const int64_t o = (int64_t)min(ks0 + PF + p, nks - 1) * 256;
      ra[p] = dp[o];
      rb[p][0] = zp[o]; rb[p][1] = zp[o + 32]; rb[p][2] = zp[o + 64]; rb[p][3] = zp[o + 96];
#endif
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b3, acc[3], 0, 0, 0);
    }
  }
  // D-layout: lane holds column b = 32 j + n, rows a = 32 wave + (r & 3) + 8 (r >> 2) + 4 kk
  float* out = Pp + (((int64_t)chunk * 64 + bin) * 128 + 32 * wave) * 128 + n;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * kk) * 128 + 32 * j] = acc[j][r];
}

// (bin m, output o) per workgroup, lane = i: the chunk sums in a fixed order and the complex combination -> Acc[m][o][i]
__global__ __launch_bounds__(64) void c64_wsum_kernel(const float* __restrict__ Pp, float* __restrict__ Acc, int nchunk) {
  const int m = blockIdx.x, o = blockIdx.y, lane = threadIdx.x;
  float prr = 0.f, pii = 0.f, pri = 0.f, pir = 0.f;
  for (int c = 0; c < nchunk; ++c) {                // fixed order: bit-reproducible
    const float* p = Pp + (((int64_t)c * 64 + m) * 128) * 128;
    prr += p[(int64_t)o * 128 + lane];
    pri += p[(int64_t)o * 128 + 64 + lane];
    pir += p[(int64_t)(64 + o) * 128 + lane];
    pii += p[(int64_t)(64 + o) * 128 + 64 + lane];
  }
  *reinterpret_cast<v2f*>(Acc + 2 * (((int64_t)m * 64 + o) * 64 + lane)) = (v2f){prr + pii, pri - pir};
}

// workgroup = o, lane = i: inverse FFT over the bins, real parts of lags 0 .. 15
__global__ __launch_bounds__(64) void c64_wfinish_kernel(const float* __restrict__ Acc, float* __restrict__ dW) {
  const int o = blockIdx.x, lane = threadIdx.x;
  v2f x[64];
#pragma unroll
  for (int m = 0; m < 64; ++m) x[m] = *reinterpret_cast<const v2f*>(Acc + 2 * (((int64_t)m * 64 + o) * 64 + lane));
  fft64<true>(x);
#pragma unroll
  for (int k = 0; k < KT; ++k) dW[((int64_t)o * NCH + lane) * KT + k] = x[pos64(k)].x * (1.0f / NB);
}

int pack_grid(const Geo& g) { return std::max(1, std::min(512, cdiv(g.ncolp, 4))); }

}  // namespace

// floats of the workspace of eav_conv64_fft_*: two filter-spectrum tables (forward, data gradient), three spectra buffers
// [64 bins][columns][128] (input of the forward - kept for the weight gradient -, input of the data gradient, a scratch one:
// GEMM output / du blocks of the weight gradient) and the weight gradient's split-K partial products
extern "C" int64_t eav_conv64_fft_ws_floats(int B, int T) {
  const Geo g = geometry(B, T, 7);
  return (int64_t)2 * 64 * 128 * 128 + (int64_t)3 * 64 * g.ncolp * 128 + (int64_t)WCH * 64 * 128 * 128 +
         (int64_t)64 * 64 * 64 * 2;
}

// rows of stat_part ([rows][128]: 64 sums, 64 sums of squares) eav_conv64_fft_fwd writes
extern "C" int eav_conv64_fft_nparts(int B, int T) {
  const Geo g = geometry(B, T, 7);
  return pack_grid(g) * 4;
}

// out [B,64,T] = separableConv(in) (bwd = 0, 'same' padding 7 / 8; stat_part as eav_conv64_fwd's, may be NULL) or its data
// gradient (bwd = 1: in = d loss / d out, out = d loss / d in; bwd = 2: the same, re-using the filter spectra the forward
// call of this step prepared in ws - it prepares both tables); w [64,64,16] = separableConv.weight.
extern "C" int eav_conv64_fft_fwd(const float* in, const float* w, float* out, float* stat_part, float* ws, int B, int T,
                                  int bwd, void* stream) {
  EAV_REQUIRE(in && w && out && ws && B > 0 && T > 0, "eav_conv64_fft_fwd: bad arguments");
  EAV_REQUIRE(((uintptr_t)ws & 15) == 0, "eav_conv64_fft_fwd: the workspace must be 16-byte aligned");
  const Geo g = geometry(B, T, bwd ? 8 : 7);
  EAV_REQUIRE(bwd >= 0 && bwd <= 2, "eav_conv64_fft_fwd: bwd must be 0, 1 or 2");
  hipStream_t st = (hipStream_t)stream;
  float* Bm = ws + (bwd ? (int64_t)64 * 128 * 128 : 0);
  const bool prepared = bwd == 2;                        // 2: the forward call of this step left the table in ws
  if (bwd == 2) bwd = 1;
  float* Z = ws + (int64_t)2 * 64 * 128 * 128 + (bwd ? (int64_t)64 * g.ncolp * 128 : 0);
  float* Y = ws + (int64_t)2 * 64 * 128 * 128 + (int64_t)2 * 64 * g.ncolp * 128;
  if (!prepared) {      // forward: both tables (weights do not change between the forward and the backward of a step)
    hipLaunchKernelGGL(c64_spectra_kernel, dim3(NCH, bwd ? 1 : 2), dim3(64), 0, st, w, Bm, bwd);
    EAV_CHECK_LAUNCH("eav_conv64_fft_fwd(spectra)");
  }
  hipLaunchKernelGGL(c64_pack_fft_kernel, dim3(pack_grid(g)), dim3(256), 0, st, in, Z, g, 0);
  EAV_CHECK_LAUNCH("eav_conv64_fft_fwd(fft)");
  hipLaunchKernelGGL(c64_bin_gemm_kernel, dim3(std::min(C64V_GW, g.ncolp / 32), 64), dim3(256), 0, st, Z, Bm, Y, g.ncolp);
  EAV_CHECK_LAUNCH("eav_conv64_fft_fwd(gemm)");
  hipLaunchKernelGGL(c64_ifft_unpack_kernel, dim3(pack_grid(g)), dim3(256), 0, st, Y, out, stat_part, g);
  EAV_CHECK_LAUNCH("eav_conv64_fft_fwd(ifft)");
  return EAV_OK;
}

// dW [64,64,16] = d loss / d separableConv.weight (WRITTEN, no partials to reduce) from du = d loss / d(conv output)
// [B,64,T].  The forward input's spectra must still be in `ws`: call after eav_conv64_fft_fwd(in, ..., ws, B, T, 0) of the
// same step (a data-gradient call in between does not disturb them).  Bit-reproducible.
extern "C" int eav_conv64_fft_wgrad(const float* du, float* dW, float* ws, int B, int T, void* stream) {
  EAV_REQUIRE(du && dW && ws && B > 0 && T > 0, "eav_conv64_fft_wgrad: bad arguments");
  const Geo g = geometry(B, T, 7);
  hipStream_t st = (hipStream_t)stream;
  float* Z = ws + (int64_t)2 * 64 * 128 * 128;
  float* D = Z + (int64_t)2 * 64 * g.ncolp * 128;
  float* Pp = Z + (int64_t)3 * 64 * g.ncolp * 128;
  float* Acc = Pp + (int64_t)WCH * 64 * 128 * 128;
  hipLaunchKernelGGL(c64_pack_fft_kernel, dim3(pack_grid(g)), dim3(256), 0, st, du, D, g, 1);
  EAV_CHECK_LAUNCH("eav_conv64_fft_wgrad(fft)");
  const int cpc = g.ncolp / WCH;                            // columns per chunk: a multiple of 16
  hipLaunchKernelGGL(c64_bin_wgemm_kernel, dim3(WCH, 64), dim3(256), 0, st, D, Z, Pp, g.ncolp, cpc);
  EAV_CHECK_LAUNCH("eav_conv64_fft_wgrad(gemm)");
  hipLaunchKernelGGL(c64_wsum_kernel, dim3(64, 64), dim3(64), 0, st, Pp, Acc, WCH);
  EAV_CHECK_LAUNCH("eav_conv64_fft_wgrad(sum)");
  hipLaunchKernelGGL(c64_wfinish_kernel, dim3(NCH), dim3(64), 0, st, Acc, dW);
  EAV_CHECK_LAUNCH("eav_conv64_fft_wgrad(finish)");
  return EAV_OK;
}
