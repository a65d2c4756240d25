// Batched GEMM with bf16 MFMA operands (v_mfma_f32_32x32x16_bf16), fp32 in HBM, fp32 accumulate/output.
//
// Same contract as eav_gemm_f32 (gemm_f32.hip): C[z] = epilogue(alpha * opA(A[z]) . opB(B[z])), all four
// operand layouts, two-level batch strides, bias / erf-GELU / residual / accumulate epilogues, split-K.
// The operands stay fp32 in memory; they are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while being staged
// into LDS, so no bf16 copies of activations or weights exist.  This is the *fast* precision mode of the
// encoders: 16x the MFMA rate of the fp32 path, but 12-layer logits drift ~5e-3 from the fp32 reference
// (DESIGN.md section 7), outside north_star's 1e-3 bound - hence opt-in (Encoder.precision).
//
// Tile 128 x BN x 32, 4 waves (2x2), LDS images [row][k] bf16 with an 80-byte row stride: every
// ds_read_b128 fragment read (8 consecutive k of one row) is conflict-free across its 16-lane groups.
#include "eav_common.h"
#include "../../../include/eav_hip_extras.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BM = 128, BK = 32, LDSK = BK + 8;   // 40 bf16 = 80 B per row

struct GemmArgs {
  const float* A; const float* B; float* C;
  const float* bias; const float* resid; float* pre;
  int M, N, K, lda, ldb, ldc, ldr, H;
  int64_t sAb, sAh, sBb, sBh, sCb, sCh;
  float alpha;
  int gelu, accumulate, ksplit;
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// K-contiguous operand (stored [R,K]): 8 consecutive k of row r -> bf16x8
__device__ __forceinline__ bf16x8 load8_rowmajor(const float* P, int ld, int r, int k, int R, int K, bool interior) {
  float v[8];
  const float* p = P + (int64_t)r * ld + k;
  if (interior || (r < R && k + 7 < K)) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (r < R && k + e < K) ? p[e] : 0.f;
  }
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
  return o;
}

// K-major operand (stored [K,R]): a 4(k) x 4(r) block as four float4 rows (16 B per lane, lanes along r)
__device__ __forceinline__ void load4x4_kmajor(const float* P, int ld, int r, int k, int R, int K, bool interior,
                                               float4 (&v)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float* p = P + (int64_t)(k + e) * ld + r;
    if (interior || (k + e < K && r + 3 < R)) {
      v[e] = *reinterpret_cast<const float4*>(p);
    } else {
      const bool kok = k + e < K;
      v[e] = make_float4(kok && r < R ? p[0] : 0.f, kok && r + 1 < R ? p[1] : 0.f, kok && r + 2 < R ? p[2] : 0.f,
                         kok && r + 3 < R ? p[3] : 0.f);
    }
  }
}
// transposing commit of that block: row r+i receives k..k+3 as 4 bf16 (8-byte store)
__device__ __forceinline__ void store4x4_kmajor(__bf16* tile, int r, int k, const float4 (&v)[4]) {
  const float c[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x}, {v[0].y, v[1].y, v[2].y, v[3].y},
                         {v[0].z, v[1].z, v[2].z, v[3].z}, {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (__bf16)c[i][e];
    *reinterpret_cast<bf16x4*>(&tile[(r + i) * LDSK + k]) = o;
  }
}

template <int BN, bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs g) {
  constexpr int WN = BN / 2, NT = WN / 32;
  constexpr int NA = BM * (BK / 8) / 256;     // bf16x8 pieces per thread (K-contiguous A): 2
  constexpr int NB = BN * (BK / 8) / 256;     // 2 or 1
  __shared__ __attribute__((aligned(16))) __bf16 As[2][BM * LDSK];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2][BN * LDSK];

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
  if (g.ksplit > 0) {
    const int kbase = z * g.ksplit;
    K = min(g.ksplit, g.K - kbase);
    A = g.A + (TA ? (int64_t)kbase * g.lda : (int64_t)kbase);
    B = g.B + (TB ? (int64_t)kbase * g.ldb : (int64_t)kbase);
    C = g.C + (int64_t)z * g.M * g.ldc;
  }
  const bool interior = (m0 + BM <= M) && (n0 + BN <= N) && (K % BK == 0);
  // K-contiguous operand: piece = (row p>>2, k-octet p&3), 2 (or 1) pieces per thread.
  // K-major operand: one 4x4 block per thread: k-quad (8 per tile) x row-quad (BM/4 or BN/4 per tile).
  bf16x8 ra[NA], rb[NB];
  float4 ta[4], tb[4];
  const int a_rq = t & (BM / 4 - 1), a_kq = t / (BM / 4);            // 32 row-quads x 8 k-quads
  const int b_rq = t & (BN / 4 - 1), b_kq = t / (BN / 4);
  const bool b_active = b_kq < BK / 4;                                // BN = 64: threads >= 128 idle for B
  auto fetch = [&](int k0) {
    if (!TA) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int p = t + 256 * i;
        ra[i] = load8_rowmajor(A, g.lda, m0 + (p >> 2), k0 + 8 * (p & 3), M, K, interior);
      }
    } else {
      load4x4_kmajor(A, g.lda, m0 + 4 * a_rq, k0 + 4 * a_kq, M, K, interior, ta);
    }
    if (!TB) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int p = t + 256 * i;
        rb[i] = load8_rowmajor(B, g.ldb, n0 + (p >> 2), k0 + 8 * (p & 3), N, K, interior);
      }
    } else if (b_active) {
      load4x4_kmajor(B, g.ldb, n0 + 4 * b_rq, k0 + 4 * b_kq, N, K, interior, tb);
    }
  };
  auto commit = [&](int buf) {
    if (!TA) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int p = t + 256 * i;
        *reinterpret_cast<bf16x8*>(&As[buf][(p >> 2) * LDSK + 8 * (p & 3)]) = ra[i];
      }
    } else {
      store4x4_kmajor(As[buf], 4 * a_rq, 4 * a_kq, ta);
    }
    if (!TB) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int p = t + 256 * i;
        *reinterpret_cast<bf16x8*>(&Bs[buf][(p >> 2) * LDSK + 8 * (p & 3)]) = rb[i];
      }
    } else if (b_active) {
      store4x4_kmajor(Bs[buf], 4 * b_rq, 4 * b_kq, tb);
    }
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int nk = (K + BK - 1) / BK;
  fetch(0);
  commit(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) fetch((kt + 1) * BK);
    const __bf16* as = &As[buf][(wm * 64 + n) * LDSK + 8 * kk];
    const __bf16* bs = &Bs[buf][(wn * WN + n) * LDSK + 8 * kk];
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(as + 16 * ks);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(as + 32 * LDSK + 16 * ks);
      bf16x8 bv[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b) bv[b] = *reinterpret_cast<const bf16x8*>(bs + 32 * b * LDSK + 16 * ks);
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bv[b], acc[0][b], 0, 0, 0);
        acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bv[b], acc[1][b], 0, 0, 0);
      }
    }
    if (kt + 1 < nk) commit(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int b = 0; b < NT; ++b) {
    const int col = n0 + wn * WN + 32 * b + n;
    if (col >= N) continue;
    const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (row >= M) continue;
        float v = g.alpha * acc[a][b][r] + bias;
        const int64_t o = (int64_t)row * g.ldc + col;
        if (g.pre) g.pre[zb * g.sCb + zh * g.sCh + o] = v;
        if (g.gelu) v = gelu_erf(v);
        if (g.resid) v += g.resid[(int64_t)row * g.ldr + col];
        if (g.accumulate) v += C[o];
        C[o] = v;
      }
    }
  }
}

template <int BN, bool TA, bool TB>
void launch(const GemmArgs& g, int batch, hipStream_t st) {
  dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), batch);
  hipLaunchKernelGGL((gemm_bf16_kernel<BN, TA, TB>), grid, dim3(256), 0, st, g);
}

void dispatch(const GemmArgs& g, int batch, int transA, int transB, hipStream_t st) {
  const int v = (g.N <= 64 ? 4 : 0) | (transA ? 2 : 0) | (transB ? 1 : 0);
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

}  // namespace

extern "C" int eav_gemm_bf16(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                             int transA, int transB, int batch, int heads, int64_t sAb, int64_t sAh, int64_t sBb,
                             int64_t sBh, int64_t sCb, int64_t sCh, float alpha, const float* bias, int gelu,
                             float* pre, const float* resid, int ldr, int accumulate, void* stream) {
  EAV_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && batch > 0 && heads > 0 && batch % heads == 0,
              "eav_gemm_bf16: bad arguments");
  EAV_REQUIRE((lda & 3) == 0 && (ldb & 3) == 0 && (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && (sAb & 3) == 0 &&
                  (sAh & 3) == 0 && (sBb & 3) == 0 && (sBh & 3) == 0,
              "eav_gemm_bf16: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
  EAV_REQUIRE(!(resid && batch > 1), "eav_gemm_bf16: residual epilogue is not batched");
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.resid = resid; g.pre = pre;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = ldr; g.H = heads;
  g.sAb = sAb; g.sAh = sAh; g.sBb = sBb; g.sBh = sBh; g.sCb = sCb; g.sCh = sCh;
  g.alpha = alpha; g.gelu = gelu; g.accumulate = accumulate; g.ksplit = 0;
  dispatch(g, batch, transA, transB, (hipStream_t)stream);
  EAV_CHECK_LAUNCH("eav_gemm_bf16");
  return EAV_OK;
}

// slice count: the formula of eav_gemm_f32_splitk_plan (gemm_f32.hip), which the caller sizes `ws` with - this file is linked
// into its own library (libeav_extras.so), so it carries its own copy
static int splitk_plan(int M, int N, int K) {
  const int tiles = cdiv(M, 128) * cdiv(N, N <= 64 ? 64 : 128);
  int ns = cdiv(1024, tiles);
  const int maxs = cdiv(K, 256);
  if (ns > maxs) ns = maxs;
  if (ns > 64) ns = 64;
  return ns < 1 ? 1 : ns;
}

extern "C" int eav_gemm_bf16_splitk(const float* A, const float* B, float* C, float* ws, int M, int N, int K, int lda,
                                    int ldb, int transA, int transB, void* stream) {
  EAV_REQUIRE(A && B && C && ws && M > 0 && N > 0 && K > 0, "eav_gemm_bf16_splitk: bad arguments");
  EAV_REQUIRE((lda & 3) == 0 && (ldb & 3) == 0 && (N & 3) == 0 &&
                  (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)ws) & 15) == 0,
              "eav_gemm_bf16_splitk: operands must be 16-byte aligned, leading dimensions and N multiples of 4");
  const int nsplit = splitk_plan(M, N, K);
  GemmArgs g;
  g.A = A; g.B = B; g.C = nsplit > 1 ? ws : C; g.bias = nullptr; g.resid = nullptr; g.pre = nullptr;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = N; g.ldr = 0; g.H = 1;
  g.sAb = g.sAh = g.sBb = g.sBh = g.sCb = g.sCh = 0;
  g.alpha = 1.f; g.gelu = 0; g.accumulate = 0;
  g.ksplit = nsplit > 1 ? cdiv(cdiv(K, nsplit), BK) * BK : 0;
  const int nz = nsplit > 1 ? cdiv(K, g.ksplit) : 1;
  hipStream_t st = (hipStream_t)stream;
  dispatch(g, nz, transA, transB, st);
  EAV_CHECK_LAUNCH("eav_gemm_bf16_splitk");
  if (nsplit > 1) {
    const int64_t n = (int64_t)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(n, 1024)), dim3(256), 0, st, ws, nz, n, C);
    EAV_CHECK_LAUNCH("eav_gemm_bf16_splitk(reduce)");
  }
  return EAV_OK;
}
