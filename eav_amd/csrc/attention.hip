// Fused multi-head self-attention for the AST / ViT encoders (head_dim 64), exact fp32 on the fp32 matrix
// cores (v_mfma_f32_32x32x2_f32), flash-style: the [N, N] score matrix never leaves the CU.
// Reference arithmetic: HF eager_attention_forward (modeling_audio_spectrogram_transformer.py:102-127, the same
// code in modeling_vit.py): softmax(Q K^T * hd^-0.5) V per (image, head), no mask, dropout 0.
//
// Layout trick (all three kernels): the score tile is computed TRANSPOSED, S^T[key][q] = K-tile . Q^T, so that in
// the MFMA accumulator layout every lane owns ONE query column and its 16 registers are 16 keys of that query:
// the softmax row reductions are register-local (plus one exchange with lane^32), and the probabilities are
// already in the B-operand layout of the next product if the contraction visits the keys in the accumulator's
// own order kappa(ks,kk) = (ks&3) + 8*(ks>>2) + 4*kk - no shuffles, no LDS round trip for P.
//
//   fwd   : O^T[d][q]  += V^T[d][kappa] * P^T[kappa][q]                 per 32q x 32key tile: 64 MFMAs
//   bwd_q : dQ^T[d][q] += K^T[d][kappa] * dS^T[kappa][q]   (query tile stationary)            96 MFMAs
//   bwd_kv: dV^T[d][k] += dO^T[d][kappa] * P[kappa][k], dK^T[d][k] += Q^T[d][kappa] * dS[kappa][k]
//           (key tile stationary, S[q][key] = Q-tile . K^T in the untransposed orientation)   128 MFMAs
// Inputs are the fused projection output qkv [B*N, 3*D] (Q | K | V, head h at columns h*64) and, for the
// backward, dO [B*N, D], the saved log-sum-exp and delta = rowsum(dO * O).
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

constexpr int HD = 64;          // head dim
constexpr int KS = HD + 1;      // LDS row stride for tiles read row-per-lane (conflict-free)
constexpr int TQ = 128;         // rows per block (4 waves x 32)

__device__ __forceinline__ int kappa(int ks, int kk) { return (ks & 3) + 8 * (ks >> 2) + 4 * kk; }

// The softmax runs in base 2: scores are produced pre-multiplied by log2(e) (folded into the staged Q), so that each
// probability costs one subtract and one native v_exp_f32 instead of the ~12-instruction expf expansion - the VALU
// work between the MFMA phases was a third of a key-tile iteration.  lse stays in natural-log units at the ABI.
constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// stage a [32][64] tile (rows r0.., zero beyond nrows) of a [*, ld] matrix into LDS with row stride `stride`
__device__ __forceinline__ void fetch_tile(const float* __restrict__ base, int ld, int r0, int nrows, float4 (&reg)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = threadIdx.x + 256 * i;
    const int r = f >> 4, c4 = f & 15;
    reg[i] = (r0 + r < nrows) ? *reinterpret_cast<const float4*>(base + (int64_t)(r0 + r) * ld + 4 * c4)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void commit_tile(float* __restrict__ lds, int stride, const float4 (&reg)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int f = threadIdx.x + 256 * i;
    float* d = lds + (f >> 4) * stride + 4 * (f & 15);
    d[0] = reg[i].x; d[1] = reg[i].y; d[2] = reg[i].z; d[3] = reg[i].w;
  }
}

// ------------------------------------------------------------------------------------------ forward
// grid (ceil(N/128), B*H).  out: ao [B*N, D] (head columns), lse [B*H, N] = m + log(l) of the scaled scores.
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ ao,
                                                          float* __restrict__ lse, int N, int H, float scale) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 32 * KS + 2 * 32 * HD];   // one array: carved below
  float (*Ks)[32 * KS] = reinterpret_cast<float (*)[32 * KS]>(smem);
  float (*Vs)[32 * HD] = reinterpret_cast<float (*)[32 * HD]>(smem + 2 * 32 * KS);
  const int D = H * HD, ld = 3 * D;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, kk = lane >> 5;
  const int q = blockIdx.x * TQ + wave * 32 + j;                 // this lane's query row
  const float* Qb = qkv + (int64_t)b * N * ld + h * HD;
  const float* Kb = Qb + D;
  const float* Vb = Qb + 2 * D;
  float qreg[HD / 2];
#pragma unroll
  for (int ks = 0; ks < HD / 2; ++ks) qreg[ks] = q < N ? (scale * LOG2E) * Qb[(int64_t)q * ld + 2 * ks + kk] : 0.f;
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const int nkt = (N + 31) / 32;
  float4 rk[2], rv[2];
  fetch_tile(Kb, ld, 0, N, rk);
  fetch_tile(Vb, ld, 0, N, rv);
  commit_tile(Ks[0], KS, rk);
  commit_tile(Vs[0], HD, rv);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) {
      fetch_tile(Kb, ld, 32 * (kt + 1), N, rk);
      fetch_tile(Vb, ld, 32 * (kt + 1), N, rv);
    }
    // S^T[key][q] = K-tile . Q^T
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* kp = Ks[buf] + j * KS + kk;
#pragma unroll
    for (int ks = 0; ks < HD / 2; ++ks) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * ks], qreg[ks], s, 0, 0, 0);
    // online softmax over this lane's 16 keys (+ the 16 of lane^32)
    float mx = -INFINITY;
    if (32 * kt + 32 > N) {                      // only the last key tile can be ragged (block-uniform branch)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (32 * kt + kappa(r, kk) >= N) s[r] = -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = ex2(m - mn);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = ex2(s[r] - mn);
      rs += s[r];
    }
    rs += __shfl_xor(rs, 32, 64);
    l = l * alpha + rs;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    // O^T[d][q] += V^T[d][kappa] * P^T[kappa][q]
    const float* vp = Vs[buf] + j;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int kr = kappa(ks, kk) * HD;
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[kr], s[ks], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[kr + 32], s[ks], o1, 0, 0, 0);
    }
    if (kt + 1 < nkt) {
      commit_tile(Ks[buf ^ 1], KS, rk);
      commit_tile(Vs[buf ^ 1], HD, rv);
    }
    __syncthreads();
  }
  // O[q][d] = O^T[d][q] / l, transposed through LDS so that every row is written as 256 contiguous bytes
  const float inv = 1.0f / l;
  if (kk == 0 && q < N) lse[(int64_t)bh * N + q] = (m + log2f(l)) * LN2;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    // write d in [32*half, 32*half+32): lane (q=j) holds d = 32*half + (r&3)+8(r>>2)+4kk
    float* ow = smem + wave * (32 * 33);          // 4 x 1056 floats <= the 8256-float tile area
#pragma unroll
    for (int r = 0; r < 16; ++r) ow[j * 33 + kappa(r, kk)] = (half ? o1[r] : o0[r]) * inv;
    __syncthreads();
    // read back row-major: 32 q x 32 d per wave = 1024 floats / 64 lanes = 16 each
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = lane + 64 * i;              // (row, quad): 32 rows x 8 quads
      const int row = idx >> 3, c4 = idx & 7;
      const int qq = blockIdx.x * TQ + wave * 32 + row;
      if (qq < N) {
        const float* src = ow + row * 33 + 4 * c4;
        *reinterpret_cast<float4*>(ao + ((int64_t)b * N + qq) * D + h * HD + 32 * half + 4 * c4) =
            make_float4(src[0], src[1], src[2], src[3]);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------ backward
// delta[bh][q] = sum_d dO[q, h*64+d] * O[q, h*64+d]   (one wave per (row, head))
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ o, const float* __restrict__ dout,
                                                         float* __restrict__ delta, int B, int N, int H) {
  const int lane = threadIdx.x & 63;
  const int64_t id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // (b*N + q)*H + h
  if (id >= (int64_t)B * N * H) return;
  const int h = (int)(id % H);
  const int64_t row = id / H;
  const int b = (int)(row / N), q = (int)(row - (int64_t)b * N);
  const int64_t off = row * (H * HD) + h * HD + lane;
  const float v = wave_sum(o[off] * dout[off]);
  if (lane == 0) delta[((int64_t)b * H + h) * N + q] = v;
}

// write a per-wave transposed accumulator pair (acc^T[d][row], lane = row) as rows of 64 contiguous floats:
// out[(row0 + r) * ld + 32*half + ...] through a [32][33] LDS patch per wave
__device__ __forceinline__ void store_rows_T(float* __restrict__ patch, const f32x16& a0, const f32x16& a1, float mul,
                                             float* __restrict__ out, int ld, int row0, int nrows, int lane) {
  const int j = lane & 31, kk = lane >> 5;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) patch[j * 33 + kappa(r, kk)] = (half ? a1[r] : a0[r]) * mul;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = lane + 64 * i;
      const int row = idx >> 3, c4 = idx & 7;
      if (row0 + row < nrows) {
        const float* src = patch + row * 33 + 4 * c4;
        *reinterpret_cast<float4*>(out + (int64_t)(row0 + row) * ld + 32 * half + 4 * c4) =
            make_float4(src[0], src[1], src[2], src[3]);
      }
    }
  }
}

// dQ: query tile stationary.  grid (ceil(N/128), B*H)
__global__ __launch_bounds__(256, 2) void attn_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            float* __restrict__ dqkv, int N, int H, float scale) {
  __shared__ __attribute__((aligned(16))) float smem[4 * 32 * KS];      // Ks[2], Vs[2], stride KS
  float (*Ks)[32 * KS] = reinterpret_cast<float (*)[32 * KS]>(smem);
  float (*Vs)[32 * KS] = reinterpret_cast<float (*)[32 * KS]>(smem + 2 * 32 * KS);
  const int D = H * HD, ld = 3 * D;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, kk = lane >> 5;
  const int q = blockIdx.x * TQ + wave * 32 + j;
  const float* Qb = qkv + (int64_t)b * N * ld + h * HD;
  const float* Kb = Qb + D;
  const float* Vb = Qb + 2 * D;
  const float* dOb = dout + (int64_t)b * N * D + h * HD;
  float qreg[HD / 2], doreg[HD / 2];
#pragma unroll
  for (int ks = 0; ks < HD / 2; ++ks) {
    qreg[ks] = q < N ? (scale * LOG2E) * Qb[(int64_t)q * ld + 2 * ks + kk] : 0.f;
    doreg[ks] = q < N ? dOb[(int64_t)q * D + 2 * ks + kk] : 0.f;
  }
  const float lq = q < N ? LOG2E * lse[(int64_t)bh * N + q] : 0.f;
  const float dq_ = q < N ? delta[(int64_t)bh * N + q] : 0.f;
  f32x16 g0, g1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; }
  const int nkt = (N + 31) / 32;
  float4 rk[2], rv[2];
  fetch_tile(Kb, ld, 0, N, rk);
  fetch_tile(Vb, ld, 0, N, rv);
  commit_tile(Ks[0], KS, rk);
  commit_tile(Vs[0], KS, rv);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) {
      fetch_tile(Kb, ld, 32 * (kt + 1), N, rk);
      fetch_tile(Vb, ld, 32 * (kt + 1), N, rv);
    }
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    const float* kp = Ks[buf] + j * KS + kk;
    const float* vp = Vs[buf] + j * KS + kk;
#pragma unroll
    for (int ks = 0; ks < HD / 2; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * ks], qreg[ks], s, 0, 0, 0);       // S^T = K . Q^T
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[2 * ks], doreg[ks], dp, 0, 0, 0);    // dP^T = V . dO^T
    }
    if (32 * kt + 32 > N) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (32 * kt + kappa(r, kk) >= N) s[r] = -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = ex2(s[r] - lq) * (dp[r] - dq_);                   // dS^T = P^T o (dP^T - delta)
    const float* kd = Ks[buf] + j;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {                                                    // dQ^T += K^T . dS^T
      const int kr = kappa(ks, kk) * KS;
      g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kd[kr], s[ks], g0, 0, 0, 0);
      g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kd[kr + 32], s[ks], g1, 0, 0, 0);
    }
    if (kt + 1 < nkt) {
      commit_tile(Ks[buf ^ 1], KS, rk);
      commit_tile(Vs[buf ^ 1], KS, rv);
    }
    __syncthreads();
  }
  store_rows_T(smem + wave * (32 * 33), g0, g1, scale, dqkv + (int64_t)b * N * ld + h * HD, ld,
               blockIdx.x * TQ + wave * 32, N, lane);
}

// dK, dV: key tile stationary.  grid (ceil(N/128), B*H)
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             float* __restrict__ dqkv, int N, int H, float scale) {
  __shared__ __attribute__((aligned(16))) float smem[4 * 32 * KS + 4 * 32];   // Qs[2], dOs[2], lse[2][32], delta[2][32]
  float (*Qs)[32 * KS] = reinterpret_cast<float (*)[32 * KS]>(smem);
  float (*Os)[32 * KS] = reinterpret_cast<float (*)[32 * KS]>(smem + 2 * 32 * KS);
  float* Ls = smem + 4 * 32 * KS;           // [2][32]
  float* Ds = Ls + 64;                      // [2][32]
  const int D = H * HD, ld = 3 * D;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, kk = lane >> 5;
  const int key = blockIdx.x * TQ + wave * 32 + j;
  const float* Qb = qkv + (int64_t)b * N * ld + h * HD;
  const float* Kb = Qb + D;
  const float* Vb = Qb + 2 * D;
  const float* dOb = dout + (int64_t)b * N * D + h * HD;
  float kreg[HD / 2], vreg[HD / 2];
#pragma unroll
  for (int ks = 0; ks < HD / 2; ++ks) {
    kreg[ks] = key < N ? Kb[(int64_t)key * ld + 2 * ks + kk] : 0.f;
    vreg[ks] = key < N ? Vb[(int64_t)key * ld + 2 * ks + kk] : 0.f;
  }
  f32x16 gk0, gk1, gv0, gv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { gk0[r] = 0.f; gk1[r] = 0.f; gv0[r] = 0.f; gv1[r] = 0.f; }
  const int nqt = (N + 31) / 32;
  float4 rq[2], ro[2];
  float rl = 0.f, rd = 0.f;
  auto fetch = [&](int qt) {
    fetch_tile(Qb, ld, 32 * qt, N, rq);
    fetch_tile(dOb, D, 32 * qt, N, ro);
    if (threadIdx.x < 32) {
      const int qq = 32 * qt + threadIdx.x;
      rl = qq < N ? lse[(int64_t)bh * N + qq] : 0.f;
      rd = qq < N ? delta[(int64_t)bh * N + qq] : 0.f;
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {      // Q is staged pre-multiplied by scale * log2(e) (dK is rescaled by ln 2 at the end)
      rq[i].x *= scale * LOG2E; rq[i].y *= scale * LOG2E; rq[i].z *= scale * LOG2E; rq[i].w *= scale * LOG2E;
    }
    commit_tile(Qs[buf], KS, rq);
    commit_tile(Os[buf], KS, ro);
    if (threadIdx.x < 32) {
      Ls[buf * 32 + threadIdx.x] = rl * LOG2E;
      Ds[buf * 32 + threadIdx.x] = rd;
    }
  };
  fetch(0);
  commit(0);
  __syncthreads();
  for (int qt = 0; qt < nqt; ++qt) {
    const int buf = qt & 1;
    if (qt + 1 < nqt) fetch(qt + 1);
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
    const float* qp = Qs[buf] + j * KS + kk;
    const float* op = Os[buf] + j * KS + kk;
#pragma unroll
    for (int ks = 0; ks < HD / 2; ++ks) {
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(qp[2 * ks], kreg[ks], s, 0, 0, 0);        // S[q][key]
      dp = __builtin_amdgcn_mfma_f32_32x32x2f32(op[2 * ks], vreg[ks], dp, 0, 0, 0);      // dP[q][key]
    }
    // register r <-> query row kappa(r, kk) of this tile
    if (32 * qt + 32 > N) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (32 * qt + kappa(r, kk) >= N) s[r] = -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = kappa(r, kk);
      const float p = ex2(s[r] - Ls[buf * 32 + qi]);
      dp[r] = p * (dp[r] - Ds[buf * 32 + qi]);     // dS[q][key]
      s[r] = p;                                     // P[q][key]
    }
    const float* od = Os[buf] + j;
    const float* qd = Qs[buf] + j;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int qr = kappa(ks, kk) * KS;
      gv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(od[qr], s[ks], gv0, 0, 0, 0);           // dV^T += dO^T . P
      gv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(od[qr + 32], s[ks], gv1, 0, 0, 0);
      gk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qd[qr], dp[ks], gk0, 0, 0, 0);          // dK^T += (scale Q)^T . dS
      gk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qd[qr + 32], dp[ks], gk1, 0, 0, 0);
    }
    if (qt + 1 < nqt) commit(buf ^ 1);
    __syncthreads();
  }
  float* base = dqkv + (int64_t)b * N * ld + h * HD;
  store_rows_T(smem + wave * (32 * 33), gk0, gk1, LN2, base + D, ld, blockIdx.x * TQ + wave * 32, N, lane);
  store_rows_T(smem + wave * (32 * 33), gv0, gv1, 1.0f, base + 2 * D, ld, blockIdx.x * TQ + wave * 32, N, lane);
}

}  // namespace

extern "C" int eav_attn_fwd(const float* qkv, float* ao, float* lse, int B, int H, int N, int head_dim, float scale,
                            void* stream) {
  EAV_REQUIRE(qkv && ao && lse && B > 0 && H > 0 && N > 0, "eav_attn_fwd: bad arguments");
  EAV_REQUIRE(head_dim == HD, "eav_attn_fwd: head_dim %d unsupported by the fused kernel (needs %d)", head_dim, HD);
  dim3 grid(cdiv(N, TQ), B * H);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, ao, lse, N, H, scale);
  EAV_CHECK_LAUNCH("eav_attn_fwd");
  return EAV_OK;
}

// delta: scratch [B*H, N].  dqkv [B*N, 3*H*64] receives dQ | dK | dV in the layout of qkv.
extern "C" int eav_attn_bwd(const float* qkv, const float* ao, const float* dout, const float* lse, float* delta,
                            float* dqkv, int B, int H, int N, int head_dim, float scale, void* stream) {
  EAV_REQUIRE(qkv && ao && dout && lse && delta && dqkv && B > 0 && H > 0 && N > 0, "eav_attn_bwd: bad arguments");
  EAV_REQUIRE(head_dim == HD, "eav_attn_bwd: head_dim %d unsupported by the fused kernel (needs %d)", head_dim, HD);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)cdiv64((int64_t)B * N * H, 4)), dim3(256), 0, st, ao, dout, delta,
                     B, N, H);
  EAV_CHECK_LAUNCH("eav_attn_bwd(delta)");
  dim3 grid(cdiv(N, TQ), B * H);
  hipLaunchKernelGGL(attn_bwd_q_kernel, grid, dim3(256), 0, st, qkv, dout, lse, delta, dqkv, N, H, scale);
  EAV_CHECK_LAUNCH("eav_attn_bwd(dQ)");
  hipLaunchKernelGGL(attn_bwd_kv_kernel, grid, dim3(256), 0, st, qkv, dout, lse, delta, dqkv, N, H, scale);
  EAV_CHECK_LAUNCH("eav_attn_bwd(dK,dV)");
  return EAV_OK;
}
