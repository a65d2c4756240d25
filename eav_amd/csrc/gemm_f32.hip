// Batched fp32 GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 fma chain).
//
//   C[z][m,n] = epilogue( alpha * sum_k opA(A[z])[m,k] * opB(B[z])[k,n] )
//   TA = 0: A stored [M,K] (K contiguous)     TA = 1: A stored [K,M] (M contiguous)
//   TB = 0: B stored [N,K] (K contiguous, i.e. an nn.Linear weight [out,in])   TB = 1: B stored [K,N]
//
// It carries every dense contraction of the AST / ViT path: nn.Linear forward (TA0,TB0), its data
// gradient (TA0,TB1) and weight gradient (TA1,TB1), the patch-embedding convolution (im2col rows),
// Q.K^T, P.V and the four attention-backward products (batched over (image, head) with two-level
// strides into the [tokens, hidden] activations - no head split/merge copies).
// Why fp32 and not bf16 MFMA: with bf16 operands the 12-layer logits drift 5e-3 from the fp32
// reference (measured), 5x the 1e-3 parity bound; the f32 MFMA is bit-exact
// fp32 at 157 TFLOP/s peak.
//
// Tile: BM x BN x 16 (BM = 128, or 64 when a 128-row grid would under-fill the chip; BN = 128 or 64; BK = 16
// keeps the double-buffered LDS at 33 KB -> 3 blocks per CU), 256 threads = 2x2 waves, each wave 64 x BN/2 as 32x32 MFMA
// tiles.  Both operands are staged K-major in LDS (As[k][m], Bs[k][n]) so that the 32 lanes of a
// half-wave read 32 consecutive floats (conflict-free ds_read_b32); the transposing store of a
// K-contiguous operand uses an odd row stride (129) and is conflict-free as well.  Global loads are
// 16 B per lane, prefetched into registers one K-tile ahead; LDS is double-buffered (one barrier
// per K-tile).
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

#ifndef EAV_GEMM_BK
#define EAV_GEMM_BK 16
#endif
constexpr int BK = EAV_GEMM_BK;

struct GemmArgs {
  const float* A; const float* B; float* C;
  const float* bias;    // [N] or null
  const float* resid;   // [M,N] (ldr) or null, added after the activation
  float* pre;           // [M,N] (ldc) or null: value before the activation
  int M, N, K, lda, ldb, ldc, ldr;
  int H;                // heads per image: z = zb*H + zh
  int64_t sAb, sAh, sBb, sBh, sCb, sCh;
  float alpha;
  int gelu;             // 1: erf-GELU after bias
  int accumulate;       // 1: C += result
  int ksplit;           // > 0: split-K mode - slice z covers k in [z*ksplit, min(K, (z+1)*ksplit)), C[z] = partial
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// load a float4 of 4 consecutive elements along the contiguous axis with bounds handling
__device__ __forceinline__ float4 ld_guard(const float* p, int64_t off, int i, int n, bool rowok) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!rowok) return v;
  if (i + 3 < n) return *reinterpret_cast<const float4*>(p + off);
  if (i + 0 < n) v.x = p[off + 0];
  if (i + 1 < n) v.y = p[off + 1];
  if (i + 2 < n) v.z = p[off + 2];
  return v;
}

template <int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256, 3) void gemm_f32_kernel(GemmArgs g) {
  constexpr int WM = BM / 2;                     // wave tile height: 64 or 32
  constexpr int MT = WM / 32;                    // 32-high MFMA row tiles per wave: 2 or 1
  constexpr int SA = TA ? (BM + 4) : (BM + 1);   // LDS row strides (floats) of the K-major images
  constexpr int SB = TB ? (BN + 4) : (BN + 1);
  constexpr int NA4 = BM * BK / 4 / 256;         // float4 loads per thread for A
  constexpr int NB4 = BN * BK / 4 / 256;         // float4 loads per thread for B
  constexpr int KQ = BK / 4;                     // float4 per K-contiguous row
  constexpr int WN = BN / 2;                     // wave tile width
  constexpr int NT = WN / 32;                    // 32-wide MFMA column tiles per wave: 2 or 1
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                              // [2][BK][SA]
  float* Bs = smem + 2 * BK * SA;                // [2][BK][SB]

  const int z = blockIdx.z, zb = z / g.H, zh = z - zb * g.H;
  const float* A = g.A + zb * g.sAb + zh * g.sAh;
  const float* B = g.B + zb * g.sBb + zh * g.sBh;
  float* C = g.C + zb * g.sCb + zh * g.sCh;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n = lane & 31, kk = lane >> 5;
  const int M = g.M, N = g.N;
  int K = g.K;
  if (g.ksplit > 0) {   // split-K: this z owns a K slice and its own partial-C slab
    const int kbase = z * g.ksplit;
    K = min(g.ksplit, g.K - kbase);
    A = g.A + (TA ? (int64_t)kbase * g.lda : (int64_t)kbase);
    B = g.B + (TB ? (int64_t)kbase * g.ldb : (int64_t)kbase);
    C = g.C + (int64_t)z * g.M * g.ldc;
  }
  // interior tiles take the unguarded path (no per-element bounds logic in the hot loop)
  const bool interior = (m0 + BM <= M) && (n0 + BN <= N) && (K % BK == 0);

  float4 ra[NA4], rb[NB4];
  auto load_tiles = [&](int k0) {
    if (interior) {
#pragma unroll
      for (int i = 0; i < NA4; ++i) {
        const int f = t + 256 * i;
        if (!TA) ra[i] = *reinterpret_cast<const float4*>(A + (int64_t)(m0 + f / KQ) * g.lda + k0 + 4 * (f % KQ));
        else ra[i] = *reinterpret_cast<const float4*>(A + (int64_t)(k0 + f / (BM / 4)) * g.lda + m0 + 4 * (f % (BM / 4)));
      }
#pragma unroll
      for (int i = 0; i < NB4; ++i) {
        const int f = t + 256 * i;
        if (!TB) rb[i] = *reinterpret_cast<const float4*>(B + (int64_t)(n0 + f / KQ) * g.ldb + k0 + 4 * (f % KQ));
        else rb[i] = *reinterpret_cast<const float4*>(B + (int64_t)(k0 + f / (BN / 4)) * g.ldb + n0 + 4 * (f % (BN / 4)));
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
      const int f = t + 256 * i;
      if (!TA) {  // A[M,K]: KQ float4 per row
        const int row = f / KQ, kq = f % KQ;
        ra[i] = ld_guard(A, (int64_t)(m0 + row) * g.lda + k0 + 4 * kq, k0 + 4 * kq, K, m0 + row < M);
      } else {    // A[K,M]: BM/4 float4 per k-row
        const int kr = f / (BM / 4), mq = f % (BM / 4);
        ra[i] = ld_guard(A, (int64_t)(k0 + kr) * g.lda + m0 + 4 * mq, m0 + 4 * mq, M, k0 + kr < K);
      }
    }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int f = t + 256 * i;
      if (!TB) {  // B[N,K]
        const int row = f / KQ, kq = f % KQ;
        rb[i] = ld_guard(B, (int64_t)(n0 + row) * g.ldb + k0 + 4 * kq, k0 + 4 * kq, K, n0 + row < N);
      } else {    // B[K,N]: BN/4 float4 per k-row
        const int kr = f / (BN / 4), nq = f % (BN / 4);
        rb[i] = ld_guard(B, (int64_t)(k0 + kr) * g.ldb + n0 + 4 * nq, n0 + 4 * nq, N, k0 + kr < K);
      }
    }
  };
  auto store_tiles = [&](int buf) {
    float* as = As + buf * BK * SA;
    float* bs = Bs + buf * BK * SB;
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
      const int f = t + 256 * i;
      if (!TA) {
        const int row = f / KQ, kq = f % KQ;
        as[(4 * kq + 0) * SA + row] = ra[i].x;
        as[(4 * kq + 1) * SA + row] = ra[i].y;
        as[(4 * kq + 2) * SA + row] = ra[i].z;
        as[(4 * kq + 3) * SA + row] = ra[i].w;
      } else {
        const int kr = f / (BM / 4), mq = f % (BM / 4);
        *reinterpret_cast<float4*>(&as[kr * SA + 4 * mq]) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NB4; ++i) {
      const int f = t + 256 * i;
      if (!TB) {
        const int row = f / KQ, kq = f % KQ;
        bs[(4 * kq + 0) * SB + row] = rb[i].x;
        bs[(4 * kq + 1) * SB + row] = rb[i].y;
        bs[(4 * kq + 2) * SB + row] = rb[i].z;
        bs[(4 * kq + 3) * SB + row] = rb[i].w;
      } else {
        const int kr = f / (BN / 4), nq = f % (BN / 4);
        *reinterpret_cast<float4*>(&bs[kr * SB + 4 * nq]) = rb[i];
      }
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = (K + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tiles((kt + 1) * BK);
    const float* as = As + buf * BK * SA + wm * WM + n;
    const float* bs = Bs + buf * BK * SB + wn * WN + n;
    // operands of k-step ks+1 are read while the MFMAs of step ks execute
    float av[MT], bv[NT];
#pragma unroll
    for (int a = 0; a < MT; ++a) av[a] = as[kk * SA + 32 * a];
#pragma unroll
    for (int b = 0; b < NT; ++b) bv[b] = bs[kk * SB + 32 * b];
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float an[MT], bn[NT];
      if (ks + 1 < BK / 2) {
        const int k = 2 * (ks + 1) + kk;
#pragma unroll
        for (int a = 0; a < MT; ++a) an[a] = as[k * SA + 32 * a];
#pragma unroll
        for (int b = 0; b < NT; ++b) bn[b] = bs[k * SB + 32 * b];
      }
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int a = 0; a < MT; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[b], acc[a][b], 0, 0, 0);
      if (ks + 1 < BK / 2) {
#pragma unroll
        for (int a = 0; a < MT; ++a) av[a] = an[a];
#pragma unroll
        for (int b = 0; b < NT; ++b) bv[b] = bn[b];
      }
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Per 32 x 32 block every run-time option
  // is tested once and all residual / accumulate loads are issued before the arithmetic (element by element, each load
  // was followed by its own s_waitcnt vmcnt(0): 16 serial memory round trips per block)
  float* prep = g.pre ? g.pre + zb * g.sCb + zh * g.sCh : nullptr;
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    const int col = n0 + wn * WN + 32 * b + n;
    const bool colok = col < N;
    const float bias = (g.bias && colok) ? g.bias[col] : 0.f;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      const int row0 = m0 + wm * WM + 32 * a + 4 * kk;
      auto rowof = [&](int r) { return row0 + (r & 3) + 8 * (r >> 2); };
      float v[16], rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] = g.alpha * acc[a][b][r] + bias;
        rv[r] = 0.f;
      }
      if (g.resid) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (colok && rowof(r) < M) rv[r] = g.resid[(int64_t)rowof(r) * g.ldr + col];
      }
      if (g.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (colok && rowof(r) < M) rv[r] += C[(int64_t)rowof(r) * g.ldc + col];
      }
      if (prep) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (colok && rowof(r) < M) prep[(int64_t)rowof(r) * g.ldc + col] = v[r];
      }
      if (g.gelu) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = gelu_erf(v[r]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (colok && rowof(r) < M) C[(int64_t)rowof(r) * g.ldc + col] = v[r] + rv[r];
    }
  }
}

template <int BM, int BN, bool TA, bool TB>
int launch_bm(const GemmArgs& g, int batch, hipStream_t st) {
  constexpr int SA = TA ? (BM + 4) : (BM + 1), SB = TB ? (BN + 4) : (BN + 1);
  const size_t lds = (size_t)2 * BK * (SA + SB) * sizeof(float);
  // once per instantiation, race-free (function-local static initialisation is thread-safe in C++11): the library
  // keeps no mutable global state
  static const hipError_t attr_rc = hipFuncSetAttribute(
      reinterpret_cast<const void*>(&gemm_f32_kernel<BM, BN, TA, TB>), hipFuncAttributeMaxDynamicSharedMemorySize,
      (int)((size_t)2 * BK * (SA + SB) * sizeof(float)));
  (void)attr_rc;
  dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), batch);
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, TA, TB>), grid, dim3(256), lds, st, g);
  return 0;
}

// 128-row tiles by default; 64-row tiles when the 128-row grid would leave the chip under-filled (e.g. the
// N = 768 projections at M = 9712: 456 blocks for 768 resident slots) and for outputs of at most 64 rows (the 40 x 160
// weight gradients of the ShallowConvNet transformer)
template <int BN, bool TA, bool TB>
int launch(const GemmArgs& g, int batch, hipStream_t st) {
  const int64_t blocks128 = (int64_t)cdiv(g.M, 128) * cdiv(g.N, BN) * batch;
  if (blocks128 < 640 || g.M <= 64) return launch_bm<64, BN, TA, TB>(g, batch, st);
  return launch_bm<128, BN, TA, TB>(g, batch, st);
}

}  // namespace

extern "C" int eav_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                            int transA, int transB, int batch, int heads, int64_t sAb, int64_t sAh, int64_t sBb,
                            int64_t sBh, int64_t sCb, int64_t sCh, float alpha, const float* bias, int gelu,
                            float* pre, const float* resid, int ldr, int accumulate, void* stream) {
  EAV_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && heads > 0 && batch % heads == 0,
              "eav_gemm_f32: bad arguments");
  EAV_REQUIRE((lda & 3) == 0 && (ldb & 3) == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && (sAb & 3) == 0 &&
                  (sAh & 3) == 0 && (sBb & 3) == 0 && (sBh & 3) == 0,
              "eav_gemm_f32: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  EAV_REQUIRE(!(resid && batch > 1), "eav_gemm_f32: residual epilogue is not batched");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.resid = resid; g.pre = pre;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.H = heads;
  g.sAb = sAb; g.sAh = sAh; g.sBb = sBb; g.sBh = sBh; g.sCb = sCb; g.sCh = sCh;
  g.alpha = alpha; g.gelu = gelu; g.accumulate = accumulate; g.ksplit = 0;
  hipStream_t st = (hipStream_t)stream;
  const bool narrow = N <= 64;
  const int v = (narrow ? 4 : 0) | (transA ? 2 : 0) | (transB ? 1 : 0);
  switch (v) {
    case 0: launch<128, false, false>(g, batch, st); break;
    case 1: launch<128, false, true>(g, batch, st); break;
    case 2: launch<128, true, false>(g, batch, st); break;
    case 3: launch<128, true, true>(g, batch, st); break;
    case 4: launch<64, false, false>(g, batch, st); break;
    case 5: launch<64, false, true>(g, batch, st); break;
    case 6: launch<64, true, false>(g, batch, st); break;
    default: launch<64, true, true>(g, batch, st); break;
  }
  EAV_CHECK_LAUNCH("eav_gemm_f32");
  return EAV_OK;
}

// Split-K form for the weight gradients (few output tiles, very long contraction over tokens):
// nsplit K-slices write partial products to `ws` ([nsplit][M][N] floats), which are then summed in a
// fixed order (fp64) into C - deterministic, no atomics.  eav_gemm_f32_splitk_plan returns nsplit.
extern "C" int eav_gemm_f32_splitk_plan(int M, int N, int K) {
  const int tiles = cdiv(M, 128) * cdiv(N, N <= 64 ? 64 : 128);
  int ns = cdiv(1024, tiles);
  const int maxs = cdiv(K, 256);
  if (ns > maxs) ns = maxs;
  if (ns > 64) ns = 64;
  return ns < 1 ? 1 : ns;
}

__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int nsplit, int64_t n, float* __restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  double a = 0, b = 0, c = 0, d = 0;
  for (int s = 0; s < nsplit; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)s * n + i);
    a += v.x; b += v.y; c += v.z; d += v.w;
  }
  *reinterpret_cast<float4*>(out + i) = make_float4((float)a, (float)b, (float)c, (float)d);
}

// small outputs with many slices (the ShallowConvNet weight gradients: 40 x 160 outputs, up to 64 slices): 8 lanes share
// one float4 of the output and walk the slices 8 apart - the serial chain of `nsplit` dependent loads was pure latency -
// then combine in a fixed butterfly order (fp64, deterministic)
__global__ void splitk_reduce_wide_kernel(const float* __restrict__ ws, int nsplit, int64_t n, float* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int sub = (int)(gid & 7);
  const int64_t i = (gid >> 3) * 4;
  double a = 0, b = 0, c = 0, d = 0;
  if (i < n)
    for (int s = sub; s < nsplit; s += 8) {
      const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)s * n + i);
      a += v.x; b += v.y; c += v.z; d += v.w;
    }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); d += __shfl_xor(d, o, 64);
  }
  if (sub == 0 && i < n) *reinterpret_cast<float4*>(out + i) = make_float4((float)a, (float)b, (float)c, (float)d);
}

extern "C" int eav_gemm_f32_splitk(const float* A, const float* B, float* C, float* ws, int M, int N, int K, int lda,
                                   int ldb, int transA, int transB, void* stream) {
  EAV_REQUIRE(A && B && C && ws && M > 0 && N > 0 && K > 0, "eav_gemm_f32_splitk: bad arguments");
  EAV_REQUIRE((lda & 3) == 0 && (ldb & 3) == 0 && (N & 3) == 0 && (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)ws) & 15) == 0,
              "eav_gemm_f32_splitk: operands must be 16-byte aligned, leading dimensions and N multiples of 4");
  const int nsplit = eav_gemm_f32_splitk_plan(M, N, K);
  GemmArgs g;
  g.A = A; g.B = B; g.C = nsplit > 1 ? ws : C; g.bias = nullptr; g.resid = nullptr; g.pre = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = N; g.ldr = 0; g.H = 1;
  g.sAb = g.sAh = g.sBb = g.sBh = g.sCb = g.sCh = 0;
  g.alpha = 1.f; g.gelu = 0; g.accumulate = 0;
  g.ksplit = nsplit > 1 ? cdiv(cdiv(K, nsplit), BK) * BK : 0;
  const int nz = nsplit > 1 ? cdiv(K, g.ksplit) : 1;
  hipStream_t st = (hipStream_t)stream;
  const int v = (N <= 64 ? 4 : 0) | (transA ? 2 : 0) | (transB ? 1 : 0);
  switch (v) {
    case 0: launch<128, false, false>(g, nz, st); break;
    case 1: launch<128, false, true>(g, nz, st); break;
    case 2: launch<128, true, false>(g, nz, st); break;
    case 3: launch<128, true, true>(g, nz, st); break;
    case 4: launch<64, false, false>(g, nz, st); break;
    case 5: launch<64, false, true>(g, nz, st); break;
    case 6: launch<64, true, false>(g, nz, st); break;
    default: launch<64, true, true>(g, nz, st); break;
  }
  EAV_CHECK_LAUNCH("eav_gemm_f32_splitk");
  if (nsplit > 1) {
    const int64_t n = (int64_t)M * N;
    if (nz >= 16 && n <= (1 << 18))
      hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3((unsigned)cdiv64(n * 2, 256)), dim3(256), 0, st, ws, nz, n, C);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(n, 1024)), dim3(256), 0, st, ws, nz, n, C);
    EAV_CHECK_LAUNCH("eav_gemm_f32_splitk(reduce)");
  }
  return EAV_OK;
}
