// Fused multi-head self-attention (head_dim 64) on the fp16 matrix cores with split operands: the fp32-grade fast
// path of attention.hip (same arithmetic: HF eager_attention_forward, modeling_audio_spectrogram_transformer.py:102-127
// = modeling_vit.py, softmax(Q K^T hd^-0.5) V per (image, head), no mask, dropout 0; and its backward).
//
// Operands are fp16 hi + lo pieces of sigma*x (sigma = per-tensor power of two, gemm_sp.hip) and every product is three
// v_mfma_f32_32x32x16_f16 (hi.hi + lo.hi + hi.lo) into an fp32 accumulator - 24 / 36 / 48 MFMAs of 32 cycles per
// 32 x 32 score tile in forward / dQ / dK,dV, against 64 / 96 / 128 of 64 cycles on the exact-fp32 MFMA.
//
// Data layout.  eav_attn_sp_prep converts an fp32 activation [B*N, ncols] (qkv, or dO) ONCE into
//   row planes  [B*N][ncols/8][2][8] f16   (8 hi halves then 8 lo halves per 8 columns: contraction over head_dim)
//   T planes    [B][ncols/64][64][Npad/8][2][8] f16  (per 64-column head chunk, transposed: contraction over tokens;
//                                                    Npad = N rounded up to 32, zero beyond N)
// with lo = fp16(sigma x - hi) (no 2^11 lift: one accumulator per product).  Probabilities are split in registers
// (p 2^14); dS uses a safe power-of-two bound with a lifted lo in the dQ kernel, which also measures max|dS| so that the
// dK,dV kernel can use the exact scale with a single accumulator.
//
// Kernel structure (all three): a block is NW waves x 32 stationary rows (queries for fwd / dQ, keys for dK,dV); the
// streamed 32-row tiles go HBM/L2 -> LDS by global_load_lds_dwordx4 (double-buffered, one barrier per tile) in the
// XOR-swizzled 16-byte-slot images of gemm_sp.hip (conflict-free ds_read_b128 fragments).  The score tile is computed
// with the streamed rows on the MFMA M axis in the order pi(i) = i with bits 2 and 3 swapped: then in the accumulator
// layout every lane owns one stationary row and its registers 8s..8s+7 are 8 CONSECUTIVE streamed rows - exactly one
// B-operand fragment of the next product (no shuffles, no LDS round trip for P / dS), and softmax row statistics are
// register-local plus one exchange with lane^32.
#include <type_traits>

#include "eav_common.h"
#include "../../include/eav_hip.h"
#include "../../include/eav_hip_tuning.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef unsigned char u8;

constexpr float LOG2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
constexpr float SP = 16384.f;             // probabilities are split as p * 2^14
constexpr float DS_DOWN = 1.f / 4194304.f;  // 2^-22: |dS| in operand units is < 2^37 (see attn_bwd_q_sp_kernel)
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ int pi_row(int j) { return (j & ~12) | ((j & 4) << 1) | ((j & 8) >> 1); }

__device__ __forceinline__ float sigma_from_bits(unsigned bits) {
  const int e = (int)((bits >> 23) & 0xff);
  if (bits == 0u || e == 0xff) return 1.f;
  int se = 14 - (e - 127);
  se = max(-126, min(126, se));
  return __uint_as_float((unsigned)(se + 127) << 23);
}
__device__ __forceinline__ unsigned slot_bits(const float* slot) { return eav_slot_bits(slot); }

// 8 fp32 -> one hi and one lo operand fragment (8 halves each); lo = fp16((t - hi) * lomul)
__device__ __forceinline__ void split_frag(const float* t, float lomul, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hi[e] = (_Float16)t[e];
    lo[e] = (_Float16)((t[e] - (float)hi[e]) * lomul);
  }
}

#ifndef ATTN_ABL
#define ATTN_ABL 0     // timing-only ablations of the forward loop (results are garbage): 1 no softmax arithmetic, 2 no MFMAs, 4 no streaming
#endif
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

// ---- LDS images -----------------------------------------------------------------------------------------------
// "row tile": 32 rows x 64 head_dim = 32 x 256 B (16 pieces of 16 B per row), slot = row*16 + (p ^ (row & 15))
// "T tile"  : 64 head_dim rows x 32 tokens = 64 x 128 B (8 pieces per row), slot = d*8 + (p ^ ((d >> 1) & 7))
// both 8 KB = 8 global_load_lds instructions of 1 KB.  src_row: pointer to the row's first byte of this tile.
// chunk c of a row tile = rows 4c..4c+3; of a T tile = rows 8c..8c+7.
// The LDS-DMA is issued from inline asm (wave-uniform 64-bit base + a 32-bit per-lane offset), not through
// __builtin_amdgcn_global_load_lds: hipcc puts an `s_waitcnt vmcnt(0)` in front of the first LDS read that follows a DMA
// builtin it cannot disambiguate - in these loops right after the prefetch of the NEXT tile was issued, i.e. the prefetch
// was waited for before the current tile's products (no overlap at all).  An asm DMA is invisible to that pass; the waits
// that order DMA and fragment reads are the explicit ones at the top of every tile.
#define SBAR() __builtin_amdgcn_sched_barrier(0)
#define ATTN_DMA(voff, base, dst) \
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(dst) : "memory")
// chunk c (rows 4c .. 4c+3) of a 32-row tile whose first row is row0 (clamped to max_row) -> the row-tile image at lds_tile
__device__ __forceinline__ void dma_row_chunk(unsigned lds_tile, int c, const u8* base, unsigned row_stride, int row0,
                                              int max_row, int lane) {
  const int r = 4 * c + (lane >> 4);
  const unsigned vo = (unsigned)min(row0 + r, max_row) * row_stride + (unsigned)(((lane & 15) ^ (r & 15)) * 16);
  ATTN_DMA(vo, base, lds_tile + (unsigned)c * 1024u);
}
__device__ __forceinline__ void glds_row_chunk(u8* lds_tile, int c, const u8* base, int64_t row_stride, int row0,
                                               int max_row, int lane) {
  const int r = 4 * c + (lane >> 4);
  const int p = (lane & 15) ^ (r & 15);
  const u8* src = base + (int64_t)min(row0 + r, max_row) * row_stride + p * 16;
  __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(lds_tile + c * 1024), 16, 0, 0);
}
__device__ __forceinline__ void glds_t_chunk(u8* lds_tile, int c, const u8* base, int64_t row_stride, int lane) {
  const int d = 8 * c + (lane >> 3);
  const int p = (lane & 7) ^ ((d >> 1) & 7);
  const u8* src = base + (int64_t)d * row_stride + p * 16;
  __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(lds_tile + c * 1024), 16, 0, 0);
}
// A-operand fragment of a row tile: MFMA row i <-> tile row pi(i), head_dim 16s + 8h2 .. +7
__device__ __forceinline__ f16x8 frag_row(const u8* tile, int prow, int s, int h2, int hl) {
  return *reinterpret_cast<const f16x8*>(tile + prow * 256 + (((4 * s + 2 * h2 + hl) ^ (prow & 15)) << 4));
}
// A-operand fragment of a T tile: MFMA row i <-> head_dim row d, tokens 16s + 8h2 .. +7
__device__ __forceinline__ f16x8 frag_t(const u8* tile, int d, int s, int h2, int hl) {
  return *reinterpret_cast<const f16x8*>(tile + d * 128 + (((4 * s + 2 * h2 + hl) ^ ((d >> 1) & 7)) << 4));
}
// A-operand fragment "head_dim row d, tokens 16s + 8h2 .. +7" (what frag_t reads from a T tile) taken from a ROW tile with
// two transposing reads (ds_read_b64_tr_b16: within a group of 16 lanes, lane 4r + q supplies 8 bytes of row r - here
// token r of a block of four, features 4q .. 4q+3 - and lane j receives column j of the four rows; gemm_sp.hip).  Lane L
// = 16 G + 4 r + q: group G covers features dbase + 16 (G & 1) .. + 15 and the token half G >> 1, so the destination lane
// (d = L & 31, h2 = L >> 5) is exactly the MFMA's.  No transposed planes, no second copy of the tile in LDS.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
__device__ __forceinline__ f16x8 frag_tr(const u8* tile, int dbase, int s, int hl, int lane) {
  const int G = lane >> 4, r = (lane >> 2) & 3, q = lane & 3;
  const int piece = 2 * ((dbase >> 3) + 2 * (G & 1) + (q >> 1)) + hl;
  const int tok = 16 * s + 8 * (G >> 1) + r;
  f16x8 out;
  const f16x4 a = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4_ptr)(tile + tok * 256 + ((piece ^ (tok & 15)) << 4) + ((q & 1) << 3))));
  const f16x4 b = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (lds_s16x4_ptr)(tile + (tok + 4) * 256 + ((piece ^ ((tok + 4) & 15)) << 4) + ((q & 1) << 3))));
  out.lo = a;
  out.hi = b;
  return out;
}
// B-operand fragments of a stationary row straight from the row planes in global memory (4 head_dim steps, hi / lo)
__device__ __forceinline__ void load_row_frags(const u8* rowptr, int h2, f16x8 (&hi)[4], f16x8 (&lo)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4* p = reinterpret_cast<const uint4*>(rowptr + 64 * s + 32 * h2);
    const uint4 a = p[0], b = p[1];
    hi[s] = *reinterpret_cast<const f16x8*>(&a);
    lo[s] = *reinterpret_cast<const f16x8*>(&b);
  }
}

// Four consecutive values of a row -> this lane's 16-byte hi or lo piece of the GEMM operand planes (gemm_sp.hip:
// lo = fp16((t - hi) 2^11)); the neighbouring lane holds the other half of the group of 8 columns.
__device__ __forceinline__ uint4 gemm_plane_piece4(float t0, float t1, float t2, float t3, bool first) {
  _Float16 h[4] = {(_Float16)t0, (_Float16)t1, (_Float16)t2, (_Float16)t3};
  _Float16 l[4] = {(_Float16)((t0 - (float)h[0]) * 2048.f), (_Float16)((t1 - (float)h[1]) * 2048.f),
                   (_Float16)((t2 - (float)h[2]) * 2048.f), (_Float16)((t3 - (float)h[3]) * 2048.f)};
  const uint2 hh = *reinterpret_cast<const uint2*>(h), ll = *reinterpret_cast<const uint2*>(l);
  const uint2 send = first ? ll : hh;
  uint2 recv;
  recv.x = __shfl_xor(send.x, 1, 64);
  recv.y = __shfl_xor(send.y, 1, 64);
  return first ? make_uint4(hh.x, hh.y, recv.x, recv.y) : make_uint4(recv.x, recv.y, ll.x, ll.y);
}

// out[(row0 + r) * ld + 32*half + ..] = acc^T: lane = row, registers = 32 columns per accumulator; through an LDS patch.
// out may be null; planes (optional): the same rows also leave as GEMM operand planes scaled by psig - planes points at the
// first row's piece of column 0 of this head (row pitch ldp bytes).
// cs (optional): the sums of the wave's valid rows, per column, go to cs[0 .. 63] (a bias-gradient partial; fixed order).
__device__ __forceinline__ float store_rows_T(float* __restrict__ patch, const f32x16& a0, const f32x16& a1, float mul,
                                              float* __restrict__ out, int ld, int row0, int nrows, int lane,
                                              u8* __restrict__ planes = nullptr, int64_t ldp = 0, float psig = 0.f,
                                              float* __restrict__ cs = nullptr) {
  const int j = lane & 31, h2 = lane >> 5;
  float vmax = 0.f;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = (half ? a1[r] : a0[r]) * mul;
      patch[j * 33 + (r & 3) + 8 * (r >> 2) + 4 * h2] = v;
      if (row0 + j < nrows) vmax = fmaxf(vmax, fabsf(v));
    }
    __syncthreads();
    if (cs && lane < 32) {                  // column `lane` of this half over the valid rows, rows in order
      float t = 0.f;
      const int nv = min(32, nrows - row0);
      for (int r = 0; r < nv; ++r) t += patch[r * 33 + lane];
      cs[32 * half + lane] = t;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = lane + 64 * i;
      const int row = idx >> 3, c4 = idx & 7;
      const float* src = patch + row * 33 + 4 * c4;
      const float4 v4 = make_float4(src[0], src[1], src[2], src[3]);
      if (out && row0 + row < nrows) *reinterpret_cast<float4*>(out + (int64_t)(row0 + row) * ld + 32 * half + 4 * c4) = v4;
      if (planes) {      // (uniform branch: the exchange inside runs on every lane)
        const uint4 pc = gemm_plane_piece4(v4.x * psig, v4.y * psig, v4.z * psig, v4.w * psig, (c4 & 1) == 0);
        if (row0 + row < nrows)
          *reinterpret_cast<uint4*>(planes + (int64_t)(row0 + row) * ldp + (int64_t)((32 * half + 4 * c4) >> 3) * 32 +
                                    (c4 & 1) * 16) = pc;
      }
    }
  }
  return vmax;
}
// row0: first of the (up to) 32 output rows [row0, row0 + 32) of the [B*N, .] tensor this wave wrote, or < 0: the
// maximum also goes to the entries of the one or two 32-row blocks they lie in (per-row-block operand scales)
__device__ __forceinline__ void emit_amax(unsigned* slot, float vmax, int lane, int salt, int row0 = -1) {
  if (!slot) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
  if (lane == 0 && vmax == vmax) {
    atomicMax(slot + EAV_SLOT_SHARD(salt), __float_as_uint(vmax));
    if (row0 >= 0) {
      eav_slot_blockmax(slot, row0, vmax);
      if (((row0 + 31) >> EAV_BLK_SHIFT) != (row0 >> EAV_BLK_SHIFT)) eav_slot_blockmax(slot, row0 + 31, vmax);
    }
  }
}

// ------------------------------------------------------------------------------------------------ forward
// 1-D grid of ceil(N / (32 NW)) * B*H workgroups (attn_block_map: the row tiles of a head share an XCD).  rowp: row planes
// of qkv [B*N, 3D]; tp: T planes (chunk 2H + h = V of head h).
// Workgroup -> (row tile, image-head) with all row tiles of one (image, head) on ONE XCD: the tiles of a head stream the
// same K / V (or Q / dO) planes, and consecutive workgroup ids go round the 8 XCDs - with the plain (x = tile, y = head)
// grid the tiles of a head sat on different XCDs and every one of them pulled the head's planes through its own L2
// (ViT forward: 740 MB fetched per launch against 310 MB of planes).  1-D grid of ntile * nbh workgroups.
__device__ __forceinline__ void attn_block_map(int ntile, int nbh, int& tile, int& bh) {
  const int L = blockIdx.x, full = (nbh & ~7) * ntile;
  if (L < full) {
    const int idx = L >> 3;
    bh = (idx / ntile) * 8 + (L & 7);
    tile = idx - (idx / ntile) * ntile;
  } else {
    const int r = L - full;
    bh = (nbh & ~7) + r / ntile;
    tile = r - (r / ntile) * ntile;
  }
}

#if ATTN_ABL & 2
#define FMF(a, b, c) fake_mfma(a, b, c)
__device__ __forceinline__ f32x16 fake_mfma(f16x8 a, f16x8 b, f32x16 c) { c[0] += (float)a[0] * (float)b[0]; return c; }
#else
#define FMF(a, b, c) MFMA16(a, b, c)
#endif
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_sp_kernel(const u8* __restrict__ rowp, const u8* __restrict__ tp,
                                                                 const float* __restrict__ slot, float* __restrict__ ao,
                                                                 float* __restrict__ lse, unsigned* __restrict__ amax,
                                                                 int N, int Npad, int H, float scale, int ntile, int nbh,
                                                                 u8* __restrict__ aop, float* __restrict__ slot_ao) {
  constexpr int STAGE = 16384;   // K row tile | V^T tile
  __shared__ __attribute__((aligned(1024))) u8 smem[2 * STAGE > NW * 32 * 33 * 4 ? 2 * STAGE : NW * 32 * 33 * 4];
  const int D = H * 64;
  const int64_t ldrow = (int64_t)3 * D * 4;
  int tile_, bh;
  attn_block_map(ntile, nbh, tile_, bh);
  const int b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h2 = lane >> 5;
  const int q0 = tile_ * 32 * NW + wave * 32;
  const u8* rows_b = rowp + (int64_t)b * N * ldrow;
  f16x8 qh[4], ql[4];
  load_row_frags(rows_b + (int64_t)min(q0 + j, N - 1) * ldrow + h * 256, h2, qh, ql);
  const u8* kbase = rows_b + (int64_t)(D + h * 64) * 4;
  const u8* vbase = rows_b + (int64_t)(2 * D + h * 64) * 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  auto issue = [&](int kt, int buf) {
    const unsigned st = lds0 + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
      const int c = wave + NW * i;
      dma_row_chunk(st, c, kbase, (unsigned)ldrow, 32 * kt, N - 1, lane);
      dma_row_chunk(st + 8192, c, vbase, (unsigned)ldrow, 32 * kt, N - 1, lane);      // V ROWS: read transposed (frag_tr)
    }
  };
  const float isg = slot[EAV_SLOT_ISIGMA];
  const float c1 = scale * LOG2E * isg * isg;
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const int nkt = (N + 31) / 32;
  const int prow = pi_row(j);
  issue(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#if !(ATTN_ABL & 4)
    if (kt + 1 < nkt) issue(kt + 1, buf ^ 1);
#endif
    if (q0 >= N) continue;                 // wave-uniform: a wave without a valid row only helps to stage the tiles
#if ATTN_ABL & 4
    const u8* kt_ = smem;
#else
    const u8* kt_ = smem + buf * STAGE;
#endif
    const u8* vt_ = kt_ + 8192;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {       // S^T[key][q] = K-tile . Q^T
      const f16x8 kh = frag_row(kt_, prow, st, h2, 0), kl = frag_row(kt_, prow, st, h2, 1);
      s = FMF(kh, qh[st], s);
      s = FMF(kl, qh[st], s);
      s = FMF(kh, ql[st], s);
    }
    // register r <-> key 32 kt + (r&7) + 8 h2 + 16 (r>>3)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] *= c1;
    if (32 * kt + 32 > N) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (32 * kt + (r & 7) + 8 * h2 + 16 * (r >> 3) >= N) s[r] = -INFINITY;
    }
    f16x8 ph[2], pl[2];
#if ATTN_ABL & 1
    for (int e = 0; e < 8; ++e) { ph[0][e] = (_Float16)s[e]; ph[1][e] = (_Float16)s[8 + e]; pl[0][e] = ph[1][e]; pl[1][e] = ph[0][e]; }
    l += 1.f; m = 0.f;
#else
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = ex2(m - mn);
    float rs = 0.f;
    float pt[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = ex2(s[r] - mn);
      rs += p;
      pt[r] = p * SP;
    }
    rs += __shfl_xor(rs, 32, 64);
    l = l * alpha + rs;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    split_frag(pt, 1.f, ph[0], pl[0]);
    split_frag(pt + 8, 1.f, ph[1], pl[1]);
#endif
#pragma unroll
    for (int st = 0; st < 2; ++st) {       // O^T[d][q] += V^T[d][key] . P^T[key][q]
      const f16x8 v0h = frag_tr(vt_, 0, st, 0, lane), v0l = frag_tr(vt_, 0, st, 1, lane);
      const f16x8 v1h = frag_tr(vt_, 32, st, 0, lane), v1l = frag_tr(vt_, 32, st, 1, lane);
      o0 = FMF(v0h, ph[st], o0);
      o1 = FMF(v1h, ph[st], o1);
      o0 = FMF(v0l, ph[st], o0);
      o1 = FMF(v1l, ph[st], o1);
      o0 = FMF(v0h, pl[st], o0);
      o1 = FMF(v1h, pl[st], o1);
    }
  }
  const float inv = l > 0.f ? isg / (l * SP) : 0.f;
  if (h2 == 0 && q0 + j < N) lse[(int64_t)bh * N + q0 + j] = (m + log2f(l)) * LN2;
  __syncthreads();   // every wave is done with the tiles before the patch area is reused
  // aop: the output also (or only: ao = null) leaves as the planes of the o-proj products.  O is a convex combination of V
  // rows, |O| <= max|V| <= max|qkv|: the planes take qkv's own scale, published in slot_ao for their consumers.
  const float psig = aop ? slot[EAV_SLOT_SIGMA] : 0.f;
  if (aop && blockIdx.x == 0 && threadIdx.x == 0) {
    slot_ao[EAV_SLOT_SIGMA] = psig;
    slot_ao[EAV_SLOT_ISIGMA] = isg;
  }
  const float vmax = store_rows_T(reinterpret_cast<float*>(smem) + wave * (32 * 33), o0, o1, inv,
                                  ao ? ao + (int64_t)b * N * D + h * 64 : nullptr, D, q0, N, lane,
                                  aop ? aop + (int64_t)b * N * (D * 4) + h * 256 : nullptr, (int64_t)D * 4, psig);
  emit_amax(amax, vmax, lane, (int)blockIdx.x * NW + wave, q0 < N ? b * N + q0 : -1);
}

// ------------------------------------------------------------------------------------------------ forward, pipelined
// The same arithmetic as attn_fwd_sp_kernel with the tile loop software-pipelined INSIDE a wave (long sequences: the
// launcher picks it above g_fwd2_above keys).  Per key tile t two regions:
//   R1  the twelve score MFMAs of tile t + 1   ||  exponentials, row sums and hi / lo split of the probabilities of tile t
//   R2  the twelve P.V MFMAs of tile t         ||  row maxima of the scores of tile t + 1
// so the matrix pipe has work while the wave's VALU slots carry the softmax.  The running maximum is a REFERENCE m, moved
// (O and l rescaled) only when some row of the wave sees a score more than 1 (log2 units) above it - probabilities are
// then at most 2, times 2^14 still fp16 - which on all but the first tiles skips the 32-register rescale.  K and V rows
// stream through two rings of three 8-KB tiles (K two tiles ahead of its score product, V two ahead of its P.V product;
// LDS-DMA issued from asm with counted vmcnt waits, one barrier per tile); every LDS fragment address is a per-lane
// register + an immediate (the loop is unrolled over the three ring positions).  Row sums stay per lane half (keys
// 8 h2 .. of every 16) until the end.  Compiled with -fno-slp-vectorize (Makefile): the split of a probability pair is
// then v_cvt_pk_f16_f32 + v_fma_mixlo_f16 + v_fma_mixhi_f16, three instructions instead of seven.
struct FwdSt {
  f16x8 qh[4], ql[4];
  f32x16 o0, o1, s;
  float m, l;
};

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// 8 fp32 -> hi and lo fragments; neg1 = -1.0f held in a register the optimiser cannot see through (fma(hi, -1, t) would be
// rewritten as a subtraction of a converted value; as an fma it is one v_fma_mix*_f16 reading the packed hi directly)
__device__ __forceinline__ void split_frag_mix(const float* t, float neg1, f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const f32x2 tt = {t[e], t[e + 1]};
    const f16x2 h = __builtin_convertvector(tt, f16x2);
    hi[e] = h[0];
    hi[e + 1] = h[1];
    lo[e] = (_Float16)__builtin_fmaf((float)h[0], neg1, t[e]);
    lo[e + 1] = (_Float16)__builtin_fmaf((float)h[1], neg1, t[e + 1]);
  }
}


template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_sp2_kernel(const u8* __restrict__ rowp, const float* __restrict__ slot,
                                                                  float* __restrict__ ao, float* __restrict__ lse,
                                                                  unsigned* __restrict__ amax, int N, int H, float scale,
                                                                  int ntile, int nbh, u8* __restrict__ aop,
                                                                  float* __restrict__ slot_ao) {
  constexpr int TILE = 8192, CH = 8 / NW;
#ifdef ATTN_TRACE
  unsigned long long trc_last = __builtin_readcyclecounter();
  unsigned trc_sum[5] = {0, 0, 0, 0, 0};       // cycles ending at marker i: 0 = rebase/loop -> top, 1 = top, 2 = R1, 3 = R2, 4 = rebase
#define TRC(i) { const unsigned long long n_ = __builtin_readcyclecounter(); trc_sum[i] += (unsigned)(n_ - trc_last); trc_last = n_; }
#else
#define TRC(i)
#endif
#ifdef ATTN_TRACE     // debug build: per-workgroup start / end wall clock and hardware id behind the lse rows (tools/probes/attn_trace.py)
  unsigned long long trace_t0 = 0;
  unsigned trace_hw = 0, trace_xcc = 0;
  if (threadIdx.x == 0) {
    trace_t0 = wall_clock64();
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(trace_hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(trace_xcc));
  }
#endif
  __shared__ __attribute__((aligned(1024))) u8 smem[6 * TILE];   // K ring [0, 3), V ring [3, 6) (48 KB >= the epilogue patches)
  static_assert(6 * TILE >= NW * 32 * 33 * 4, "epilogue patches alias the rings");
  const int D = H * 64;
  const int ldrow = 3 * D * 4;
  int tile_, bh;
  attn_block_map(ntile, nbh, tile_, bh);
  const int b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h2 = lane >> 5;
  const int q0 = tile_ * 32 * NW + wave * 32;
  const bool active = q0 < N;                 // (wave-uniform) a wave without a valid row only helps to stage the tiles
  const u8* rows_b = rowp + (int64_t)b * N * ldrow;
  FwdSt f;
  load_row_frags(rows_b + (int64_t)min(q0 + j, N - 1) * ldrow + h * 256, h2, f.qh, f.ql);
  const u8* kbase = rows_b + (int64_t)(D + h * 64) * 4;
  const u8* vbase = rows_b + (int64_t)(2 * D + h * 64) * 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  const int nkt = (N + 31) / 32;
  float neg1;
  asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(neg1));
  // chunk c of a tile = rows 4c .. 4c+3; this lane's row and 16-byte piece (source side of the XOR swizzle)
  int crow[CH];
  unsigned cpo[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = wave + NW * i;
    crow[i] = 4 * c + (lane >> 4);
    cpo[i] = (unsigned)(((lane & 15) ^ (crow[i] & 15)) * 16);
  }
  auto issue_k = [&](int t) {                   // K rows of tile t -> K ring position t % 3
    const unsigned dst = lds0 + (unsigned)(t % 3) * TILE;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const unsigned vo = (unsigned)min(32 * t + crow[i], N - 1) * (unsigned)ldrow + cpo[i];
      ATTN_DMA(vo, kbase, dst + (unsigned)(wave + NW * i) * 1024u);
    }
  };
  auto issue_v = [&](int t) {
    const unsigned dst = lds0 + (unsigned)(3 + t % 3) * TILE;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const unsigned vo = (unsigned)min(32 * t + crow[i], N - 1) * (unsigned)ldrow + cpo[i];
      ATTN_DMA(vo, vbase, dst + (unsigned)(wave + NW * i) * 1024u);
    }
  };
  // fragment addresses inside a tile (per lane; ring position and token step are immediates)
  const int prow = pi_row(j);
  int kofs[4][2];                               // K row fragment [head_dim step][hl]
#pragma unroll
  for (int st = 0; st < 4; ++st)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) kofs[st][hl] = prow * 256 + (((4 * st + 2 * h2 + hl) ^ (prow & 15)) << 4);
  int vofs[2][2][2];                            // V transposing reads: [dbase / 32][hl][token + 4]
  {
    const int G = lane >> 4, r = (lane >> 2) & 3, q = lane & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int hl = 0; hl < 2; ++hl)
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) {
          const int piece = 2 * (4 * db + 2 * (G & 1) + (q >> 1)) + hl;
          const int tok = 8 * (G >> 1) + r + 4 * ab;
          vofs[db][hl][ab] = tok * 256 + ((piece ^ (tok & 15)) << 4) + ((q & 1) << 3);
        }
  }
  const float isg = slot[EAV_SLOT_ISIGMA];
  const float c1 = scale * LOG2E * isg * isg;
#pragma unroll
  for (int r = 0; r < 16; ++r) { f.o0[r] = 0.f; f.o1[r] = 0.f; f.s[r] = 0.f; }
  f.m = -INFINITY;
  f.l = 0.f;

  auto scores = [&](const u8* kt_, f32x16& s) {      // s = K-tile . Q^T (raw operand units)
    f16x8 kh[4], kl[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      kh[st] = *reinterpret_cast<const f16x8*>(kt_ + kofs[st][0]);
      kl[st] = *reinterpret_cast<const f16x8*>(kt_ + kofs[st][1]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      s = FMF(kh[st], f.qh[st], s);
      s = FMF(kl[st], f.qh[st], s);
      s = FMF(kh[st], f.ql[st], s);
    }
  };
  // row maximum (log2 units) of a score tile.  The ragged last tile needs no mask here: its rows beyond N are copies of
  // key N - 1 (the DMA clamps the row index), which cannot raise the maximum; their probabilities are zeroed in the last step.
  auto tile_max = [&](const f32x16& s) {
    float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
    mx = fmaxf(mx, s[15]);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * c1;
  };
  // move the reference maximum when some row of the wave needs it (wave-uniform branch)
  auto rebase = [&](float mx) {
    if (__builtin_amdgcn_ballot_w64(mx > f.m + 1.0f) != 0) {
      const float mn = fmaxf(f.m, mx);
      const float alpha = ex2(f.m - mn);
      f.m = mn;
      f.l *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { f.o0[r] *= alpha; f.o1[r] *= alpha; }
    }
  };

  // one tile.  RP = ring position of tile kt; MORE: a tile kt + 1 exists (else: the last, possibly ragged, tile).
  // The order below is the issue order (sched_barrier after every MFMA group pins it): one MFMA, then its fillers.
  auto step = [&](auto RPc, auto MOREc, int kt) {
    constexpr int RP = decltype(RPc)::value;
    constexpr bool MORE = decltype(MOREc)::value;
    const u8* kn_ = smem + ((RP + 1) % 3) * TILE;
    const u8* vt_ = smem + (3 + RP) * TILE;
    f32x16 sn;
    f16x8 kh[4], kl[4];
    f16x8 v[2][2][2];                            // V^T fragments [token step][dbase][hl]
    f16x8 ph[2], pl[2];
    float pt[16];
    float rs = 0.f;
    const float sh = 14.f - f.m;                 // probabilities leave the exponential already scaled by 2^14 (SP)
    SBAR();
    if constexpr (MORE) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        kh[st] = *reinterpret_cast<const f16x8*>(kn_ + kofs[st][0]);
        kl[st] = *reinterpret_cast<const f16x8*>(kn_ + kofs[st][1]);
      }
    }
    auto expo = [&](int r) {
#if ATTN_ABL & 1
      pt[r] = f.s[r];
#else
      pt[r] = ex2(fmaf(f.s[r], c1, sh));
      if constexpr (!MORE) {
        if (32 * kt + (r & 7) + 8 * h2 + 16 * (r >> 3) >= N) pt[r] = 0.f;
      }
      rs += pt[r];
#endif
    };
    auto split2 = [&](int pr) {                  // probabilities 2 pr, 2 pr + 1 -> their halves of the hi / lo fragments
      const int e = (2 * pr) & 7, fr = pr >> 2;
      const f32x2 tt = {pt[2 * pr], pt[2 * pr + 1]};
      const f16x2 hh = __builtin_convertvector(tt, f16x2);
      ph[fr][e] = hh[0];
      ph[fr][e + 1] = hh[1];
      pl[fr][e] = (_Float16)__builtin_fmaf((float)hh[0], neg1, pt[2 * pr]);
      pl[fr][e + 1] = (_Float16)__builtin_fmaf((float)hh[1], neg1, pt[2 * pr + 1]);
    };
    auto vread = [&](int i) {                    // i = 4 st + 2 db + hl
      const int st = i >> 2, db = (i >> 1) & 1, hl = i & 1;
      const f16x4 a = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4_ptr)(vt_ + vofs[db][hl][0] + st * 4096)));
      const f16x4 c = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4_ptr)(vt_ + vofs[db][hl][1] + st * 4096)));
      v[st][db][hl].lo = a;
      v[st][db][hl].hi = c;
    };
    TRC(1);
    // head: the first exponentials cover the latency of the K fragment reads
#pragma unroll
    for (int r = 0; r < 6; ++r) expo(r);
    SBAR();
    // R1: scores of tile kt + 1 || exponentials 6 .. 15, split, V fragments of token step 0
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      if constexpr (MORE) {
        const int st = g / 3, term = g - 3 * st;
        if (g == 0) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          sn = FMF(kh[0], f.qh[0], zero);
        } else if (term == 0) sn = FMF(kh[st], f.qh[st], sn);
        else if (term == 1) sn = FMF(kl[st], f.qh[st], sn);
        else sn = FMF(kh[st], f.ql[st], sn);
      }
      if (g < 10) expo(6 + g);
      if (g <= 4) split2(g);
      else if (g == 6) split2(5);
      else if (g == 8) split2(6);
      else if (g == 10) split2(7);
      if (g >= 4 && g < 8) vread(g - 4);
      SBAR();
    }
    f.l += rs;
    TRC(2);
    // R2: P.V of tile kt || V fragments of token step 1, row maxima of the scores of tile kt + 1
    float mxa = 0.f, mxb = 0.f;
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      const int st = g / 6, i = g - 6 * st;
      if (i == 0) f.o0 = FMF(v[st][0][0], ph[st], f.o0);
      else if (i == 1) f.o1 = FMF(v[st][1][0], ph[st], f.o1);
      else if (i == 2) f.o0 = FMF(v[st][0][1], ph[st], f.o0);
      else if (i == 3) f.o1 = FMF(v[st][1][1], ph[st], f.o1);
      else if (i == 4) f.o0 = FMF(v[st][0][0], pl[st], f.o0);
      else f.o1 = FMF(v[st][1][0], pl[st], f.o1);
      if (g < 4) vread(4 + g);
      if constexpr (MORE) {
#if !(ATTN_ABL & 1)
        if (g == 3) { mxa = fmaxf(fmaxf(sn[0], sn[1]), sn[2]); mxb = fmaxf(fmaxf(sn[3], sn[4]), sn[5]); }
        if (g == 4) { mxa = fmaxf(fmaxf(mxa, sn[6]), sn[7]); mxb = fmaxf(fmaxf(mxb, sn[8]), sn[9]); }
        if (g == 5) { mxa = fmaxf(fmaxf(mxa, sn[10]), sn[11]); mxb = fmaxf(fmaxf(mxb, sn[12]), sn[13]); }
        if (g == 6) { mxa = fmaxf(fmaxf(mxa, sn[14]), sn[15]); mxa = fmaxf(mxa, mxb); }
        if (g == 7) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mxa), __float_as_uint(mxa), false, false);
          mxa = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * c1;
        }
#endif
      }
      SBAR();
    }
    TRC(3);
    if constexpr (MORE) {
      f.s = sn;
#if !(ATTN_ABL & 1)
      rebase(mxa);
#endif
    }
    TRC(4);
  };

  // ---- prologue: K(0), V(0), K(1) | K(2), V(1); scores of tile 0
  issue_k(0);
  issue_v(0);
  if (nkt > 1) issue_k(1);
  if (nkt > 2) issue_k(2);
  if (nkt > 1) issue_v(1);
  if (nkt > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CH) : "memory");
  else if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // ---- tiles.  At the top of tile kt everything but the newest group (K(kt+2), V(kt+1)) has landed: K(kt+1), V(kt).
  auto top = [&](int kt) {
    TRC(0);
#if !(ATTN_ABL & 4)
    if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CH) : "memory");
    else if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if !(ATTN_ABL & 8)
    __builtin_amdgcn_s_barrier();
#endif
#if !(ATTN_ABL & 4)
    if (kt + 3 < nkt) issue_k(kt + 3);          // into the ring position of K(kt): its scores were formed in tile kt - 1
    if (kt + 2 < nkt) issue_v(kt + 2);          // into the position of V(kt - 1)
#endif
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using T = std::true_type;
  using F = std::false_type;
  if (!active) {                                 // (wave-uniform) no valid row: this wave only stages its share of the tiles
    for (int kt = 0; kt < nkt; ++kt) top(kt);
  } else {
    scores(smem, f.s);
    f.m = tile_max(f.s);                         // (O and l are still zero: nothing to rescale)
    int kt = 0;
    for (; kt + 3 < nkt; kt += 3) {              // ring position of tile kt is 0 at the top of every round
      top(kt);
      step(I0{}, T{}, kt);
      top(kt + 1);
      step(I1{}, T{}, kt + 1);
      top(kt + 2);
      step(I2{}, T{}, kt + 2);
    }
    const int rem = nkt - kt;                    // 1 .. 3 tiles left, the last one without a successor
    if (rem > 1) {
      top(kt);
      step(I0{}, T{}, kt);
      ++kt;
    }
    if (rem > 2) {
      top(kt);
      step(I1{}, T{}, kt);
      ++kt;
    }
    top(kt);
    if (rem == 1) step(I0{}, F{}, kt);
    else if (rem == 2) step(I1{}, F{}, kt);
    else step(I2{}, F{}, kt);
  }
  float l = f.l;
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
    l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  const float inv = l > 0.f ? isg / l : 0.f;     // l carries the 2^14 of the probabilities
  if (h2 == 0 && q0 + j < N) lse[(int64_t)bh * N + q0 + j] = (f.m + log2f(l * (1.f / 16384.f))) * LN2;
  __syncthreads();   // every wave is done with the tiles before the patch area is reused
  const float psig = aop ? slot[EAV_SLOT_SIGMA] : 0.f;
  if (aop && blockIdx.x == 0 && threadIdx.x == 0) {
    slot_ao[EAV_SLOT_SIGMA] = psig;
    slot_ao[EAV_SLOT_ISIGMA] = isg;
  }
  const float vmax = store_rows_T(reinterpret_cast<float*>(smem) + wave * (32 * 33), f.o0, f.o1, inv,
                                  ao ? ao + (int64_t)b * N * D + h * 64 : nullptr, D, q0, N, lane,
                                  aop ? aop + (int64_t)b * N * (D * 4) + h * 256 : nullptr, (int64_t)D * 4, psig);
  emit_amax(amax, vmax, lane, (int)blockIdx.x * NW + wave, q0 < N ? b * N + q0 : -1);
#ifdef ATTN_TRACE
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long* tr = reinterpret_cast<unsigned long long*>(lse + (((int64_t)nbh * N + 3) & ~3ll)) + 8 * (int64_t)blockIdx.x;
    tr[0] = trace_t0;
    tr[1] = wall_clock64();
    tr[2] = trace_hw | ((unsigned long long)trc_sum[0] << 32);
    tr[3] = trace_xcc | ((unsigned long long)trc_sum[1] << 32);
    tr[4] = trc_sum[2] | ((unsigned long long)trc_sum[3] << 32);
    tr[5] = trc_sum[4];
  }
#endif
}

// sigma of the planes the backward kernels write dqkv as: the bound of eav_attn_dqkv_bound (see there), formed by every wave
// from the same slot words - one more one-block launch on the main stream waited ~12 us for a CU between the persistent GEMMs
__device__ __forceinline__ float dqkv_sigma(const float* __restrict__ slot_do, const float* __restrict__ slot_qkv, int N,
                                            float scale, int lane) {
  float m = slot_do[32 * lane], q = slot_qkv[32 * lane];          // shard words (non-negative floats)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o, 64));
    q = fmaxf(q, __shfl_xor(q, o, 64));
  }
  if (q == 0.f) q = 32768.f * slot_qkv[EAV_SLOT_ISIGMA];
  const float bound = (float)N * m * fmaxf(1.f, 128.f * scale * q * q) * 1.0001f;
  return sigma_from_bits(__float_as_uint(bound));
}

// ------------------------------------------------------------------------------------------------ backward: dQ
// Query tile stationary.  Streams K rows, V rows and K^T.  dS^T = P^T o (dP^T - delta) in operand units
// (sigma_do sigma_qkv dS) is bounded by 2^37: |dP| <= 64 * 2^15 * 2^15 and |delta| = |dO . O| <= the same because O is
// a convex combination of V rows; it is split as t = dS 2^-22 with a lifted lo (second accumulator), and max |t| is
// published for the dK,dV kernel.
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_q_sp_kernel(
    const u8* __restrict__ rowp, const u8* __restrict__ tp, const u8* __restrict__ dorow, const float* __restrict__ slot,
    const float* __restrict__ slot_do, const float* __restrict__ lse, const float* __restrict__ ao,
    const float* __restrict__ dout, float* __restrict__ delta, float* __restrict__ dqkv,
    unsigned* __restrict__ amax_ds, unsigned* __restrict__ amax_out, int N, int Npad, int H, float scale, int ntile, int nbh,
    u8* __restrict__ gplanes, float* __restrict__ gslot, float* __restrict__ cs_part, const u8* __restrict__ aop,
    const float* __restrict__ slot_ao) {
  constexpr int STAGE = 16384;   // K rows | V rows (K^T for the last product is read transposed from the K rows: frag_tr)
  __shared__ __attribute__((aligned(1024))) u8 smem[2 * STAGE > NW * 32 * 33 * 4 ? 2 * STAGE : NW * 32 * 33 * 4];
  const int D = H * 64;
  const int64_t ldrow = (int64_t)3 * D * 4, lddo = (int64_t)D * 4;
  int tile_, bh;
  attn_block_map(ntile, nbh, tile_, bh);
  const int b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h2 = lane >> 5;
  const int q0 = tile_ * 32 * NW + wave * 32;
  const int q = min(q0 + j, N - 1);
  const u8* rows_b = rowp + (int64_t)b * N * ldrow;
  f16x8 qh[4], ql[4], gh[4], gl[4];
  load_row_frags(rows_b + (int64_t)q * ldrow + h * 256, h2, qh, ql);
  load_row_frags(dorow + ((int64_t)b * N + q) * lddo + h * 256, h2, gh, gl);
  const u8* kbase = rows_b + (int64_t)(D + h * 64) * 4;
  const u8* vrbase = rows_b + (int64_t)(2 * D + h * 64) * 4;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  auto issue = [&](int kt, int buf) {
    const unsigned st = lds0 + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
      const int c = wave + NW * i;
      dma_row_chunk(st, c, kbase, (unsigned)ldrow, 32 * kt, N - 1, lane);
      dma_row_chunk(st + 8192, c, vrbase, (unsigned)ldrow, 32 * kt, N - 1, lane);
    }
  };
  const float isg = slot[EAV_SLOT_ISIGMA], isd = slot_do[EAV_SLOT_ISIGMA];
  const float c1 = scale * LOG2E * isg * isg;
  const float lq = LOG2E * lse[(int64_t)bh * N + q];
  // delta[q] = dO[q, head] . O[q, head] (fp32): this lane's half of the row, the other half from lane ^ 32; published
  // for the dK,dV kernel, which runs after this one
  float dsum = 0.f;
  if (aop) {
    // ... from the planes: dO's row fragments are already in registers (gh / gl, unlifted lo), O's come from the o-proj
    // operand planes the forward wrote (same 16-byte pieces, lo lifted by 2^11) - no fp32 O tensor exists in the training
    // step and neither fp32 tensor is read (ViT: 154 of this kernel's 557 MB).  Products of two 22-bit values, fp32 sums.
    const u8* orow = aop + ((int64_t)b * N + q) * ((int64_t)D * 4) + h * 256;
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint4 ph = *reinterpret_cast<const uint4*>(orow + 64 * s + 32 * h2);
      const uint4 pl = *reinterpret_cast<const uint4*>(orow + 64 * s + 32 * h2 + 16);
      const f16x8 oh = *reinterpret_cast<const f16x8*>(&ph), ol = *reinterpret_cast<const f16x8*>(&pl);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ov = fmaf((float)ol[e], 1.f / 2048.f, (float)oh[e]);
        const float gv = (float)gh[s][e] + (float)gl[s][e];
        if (e & 1) a1 = fmaf(ov, gv, a1);
        else a0 = fmaf(ov, gv, a0);
      }
    }
    dsum = (a0 + a1) * (slot_ao[EAV_SLOT_ISIGMA] * isd);
    dsum += __shfl_xor(dsum, 32, 64);
    if (h2 == 0 && q0 + j < N) delta[(int64_t)bh * N + q] = dsum;
  } else {
    const float4* o4 = reinterpret_cast<const float4*>(ao + ((int64_t)b * N + q) * D + h * 64 + 32 * h2);
    const float4* g4 = reinterpret_cast<const float4*>(dout + ((int64_t)b * N + q) * D + h * 64 + 32 * h2);
    float4 ov[8], gv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ov[i] = o4[i]; gv[i] = g4[i]; }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a0 = fmaf(ov[i].x, gv[i].x, a0); a1 = fmaf(ov[i].y, gv[i].y, a1);
      a2 = fmaf(ov[i].z, gv[i].z, a2); a3 = fmaf(ov[i].w, gv[i].w, a3);
    }
    dsum = (a0 + a1) + (a2 + a3);
    dsum += __shfl_xor(dsum, 32, 64);
    if (h2 == 0 && q0 + j < N) delta[(int64_t)bh * N + q] = dsum;
  }
  const float dq_ = dsum * slot[EAV_SLOT_SIGMA] * slot_do[EAV_SLOT_SIGMA];   // delta in operand units
  f32x16 g0, g1, x0, x1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { g0[r] = 0.f; g1[r] = 0.f; x0[r] = 0.f; x1[r] = 0.f; }
  float tmax = 0.f;
  const int nkt = (N + 31) / 32;
  const int prow = pi_row(j);
  float neg2048;
  asm volatile("s_mov_b32 %0, 0xc5000000" : "=s"(neg2048));      // -2048.0f, opaque to the optimiser (see split_frag_mix)
  // fragment addresses inside a 32-row tile (per lane); the stage is an immediate (tile<BUF>)
  int kofs[4][2];
#pragma unroll
  for (int st = 0; st < 4; ++st)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) kofs[st][hl] = prow * 256 + (((4 * st + 2 * h2 + hl) ^ (prow & 15)) << 4);
  int tofs[2][2][2];                            // transposing reads of the K rows: [dbase / 32][hl][token + 4]
  {
    const int G = lane >> 4, r = (lane >> 2) & 3, qq = lane & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int hl = 0; hl < 2; ++hl)
#pragma unroll
        for (int ab = 0; ab < 2; ++ab) {
          const int piece = 2 * (4 * db + 2 * (G & 1) + (qq >> 1)) + hl;
          const int tok = 8 * (G >> 1) + r + 4 * ab;
          tofs[db][hl][ab] = tok * 256 + ((piece ^ (tok & 15)) << 4) + ((qq & 1) << 3);
        }
  }
  const float dqs = dq_ * DS_DOWN;              // t = p . [(dP - delta) 2^-22]: the down-scale rides on the fma of the bracket
                                                // (folded into the exponent it would cost the probability 2e-6 of relative accuracy)
  // one key tile.  LAST: the (possibly ragged) last tile - keys beyond N contribute nothing
  auto tile = [&](auto BUFc, auto LASTc, int kt) {
    constexpr int BUF = decltype(BUFc)::value;
    constexpr bool LAST = decltype(LASTc)::value;
    const u8* kt_ = smem + BUF * STAGE;
    const u8* vr_ = kt_ + 8192;
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const f16x8 kh = *reinterpret_cast<const f16x8*>(kt_ + kofs[st][0]), kl = *reinterpret_cast<const f16x8*>(kt_ + kofs[st][1]);
      const f16x8 vh = *reinterpret_cast<const f16x8*>(vr_ + kofs[st][0]), vl = *reinterpret_cast<const f16x8*>(vr_ + kofs[st][1]);
      s = MFMA16(kh, qh[st], s);          // S^T = K . Q^T
      dp = MFMA16(vh, gh[st], dp);        // dP^T = V . dO^T
      s = MFMA16(kl, qh[st], s);
      dp = MFMA16(vl, gh[st], dp);
      s = MFMA16(kh, ql[st], s);
      dp = MFMA16(vh, gl[st], dp);
      if (st & 1) SBAR();                 // (at most two steps' fragments in flight: the kernel sits at 256 registers)
    }
    // per token step (16 keys): dS and its hi / lo split, then the six dQ MFMAs of the step - the second step's
    // arithmetic is independent of the first step's MFMAs and fills their issue slots
#pragma unroll
    for (int fr = 0; fr < 2; ++fr) {
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 8 * fr + e;
        float p = ex2(fmaf(s[r], c1, -lq));
        if constexpr (LAST) {
          if (32 * kt + (r & 7) + 8 * h2 + 16 * (r >> 3) >= N) p = 0.f;
        }
        t[e] = p * fmaf(dp[r], DS_DOWN, -dqs);
        tmax = fmaxf(tmax, fabsf(t[e]));
      }
      f16x8 th, tl;
#pragma unroll
      for (int e = 0; e < 8; e += 2) {          // hi = fp16(t), lo = fp16((t - hi) 2^11): one v_cvt_pk + two v_mul + two v_fma_mix
        const f32x2 tt = {t[e], t[e + 1]};
        const f16x2 hh = __builtin_convertvector(tt, f16x2);
        th[e] = hh[0];
        th[e + 1] = hh[1];
        tl[e] = (_Float16)__builtin_fmaf((float)hh[0], neg2048, t[e] * 2048.f);
        tl[e + 1] = (_Float16)__builtin_fmaf((float)hh[1], neg2048, t[e + 1] * 2048.f);
      }
      f16x8 kf[2][2];                      // K^T fragments [dbase][hl]
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) {
          const f16x4 a_ = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4_ptr)(kt_ + tofs[db][hl][0] + fr * 4096)));
          const f16x4 c_ = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (lds_s16x4_ptr)(kt_ + tofs[db][hl][1] + fr * 4096)));
          kf[db][hl].lo = a_;
          kf[db][hl].hi = c_;
        }
      g0 = MFMA16(kf[0][0], th, g0);       // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
      g1 = MFMA16(kf[1][0], th, g1);
      g0 = MFMA16(kf[0][1], th, g0);
      g1 = MFMA16(kf[1][1], th, g1);
      x0 = MFMA16(kf[0][0], tl, x0);
      x1 = MFMA16(kf[1][0], tl, x1);
    }
  };
  auto top = [&](int kt, int buf) {             // tile kt has landed in stage buf; prefetch tile kt + 1 into the other one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nkt) issue(kt + 1, buf ^ 1);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using T = std::true_type;
  using F = std::false_type;
  issue(0, 0);
  if (q0 >= N) {                                 // (wave-uniform) no valid row: this wave only stages its share of the tiles
    for (int kt = 0; kt < nkt; ++kt) top(kt, kt & 1);
  } else {
    int kt = 0;
    for (; kt + 2 < nkt; kt += 2) {
      top(kt, 0);
      tile(I0{}, F{}, kt);
      top(kt + 1, 1);
      tile(I1{}, F{}, kt + 1);
    }
    if (kt + 1 < nkt) {
      top(kt, 0);
      tile(I0{}, F{}, kt);
      top(kt + 1, 1);
      tile(I1{}, T{}, kt + 1);
    } else {
      top(kt, 0);
      tile(I0{}, T{}, kt);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { g0[r] += x0[r] * (1.f / 2048.f); g1[r] += x1[r] * (1.f / 2048.f); }
  emit_amax(amax_ds, tmax, lane, (int)blockIdx.x * NW + wave);
  __syncthreads();
  const float mul = scale * (1.f / DS_DOWN) * isd * isg * isg;
  // gplanes (optional): dQ also (or only: dqkv = null) leaves as its section of the operand planes of the q/k/v projection's
  // gradient products, scaled by gslot's sigma - a BOUND of |dqkv| set before the launch (eav_attn_dqkv_bound) - and the
  // column sums of the wave's rows (bias-gradient partials) go to cs_part[(b, row tile)][3 D]
  const int nrb = (N + 31) / 32;
  const float gsig = gplanes ? dqkv_sigma(slot_do, slot, N, scale, lane) : 0.f;
  if (gplanes && blockIdx.x == 0 && threadIdx.x == 0) {          // published for the consumers of the planes
    gslot[EAV_SLOT_SIGMA] = gsig;
    gslot[EAV_SLOT_ISIGMA] = 1.f / gsig;
  }
  const float vmax = store_rows_T(reinterpret_cast<float*>(smem) + wave * (32 * 33), g0, g1, mul,
                                  dqkv ? dqkv + (int64_t)b * N * 3 * D + h * 64 : nullptr, 3 * D, q0, N, lane,
                                  gplanes ? gplanes + (int64_t)b * N * (3 * D * 4) + h * 256 : nullptr, (int64_t)3 * D * 4,
                                  gsig,
                                  cs_part && q0 < N ? cs_part + ((int64_t)b * nrb + q0 / 32) * (3 * D) + h * 64 : nullptr);
  emit_amax(amax_out, vmax, lane, (int)blockIdx.x * NW + wave, q0 < N ? b * N + q0 : -1);
}

// ------------------------------------------------------------------------------------------------ backward: dK, dV
// Key tile stationary.  Streams Q rows, dO rows, Q^T and dO^T.  S[q][key] = Q-tile . K^T (queries on the MFMA M axis in
// pi order), so register r of lane (key, h2) is query 32 qt + (r&7) + 8 h2 + 16 (r>>3).
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_bwd_kv_sp_kernel(
    const u8* __restrict__ rowp, const u8* __restrict__ tp, const u8* __restrict__ dorow, const u8* __restrict__ dotp,
    const float* __restrict__ slot, const float* __restrict__ slot_do, const float* __restrict__ slot_ds,
    const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dqkv,
    unsigned* __restrict__ amax_out, int N, int Npad, int H, float scale, int ntile, int nbh,
    u8* __restrict__ gplanes, const float* __restrict__ gslot, float* __restrict__ cs_part) {
  constexpr int STAGE = 16384 + 256;   // Q rows | dO rows | lse[32], delta[32] of the query tile (Q^T, dO^T: frag_tr)
  __shared__ __attribute__((aligned(1024))) u8 smem[2 * STAGE];
  const int D = H * 64;
  const int64_t ldrow = (int64_t)3 * D * 4, lddo = (int64_t)D * 4;
  int tile_, bh;
  attn_block_map(ntile, nbh, tile_, bh);
  const int b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, h2 = lane >> 5;
  const int k0 = tile_ * 32 * NW + wave * 32;
  const int key = min(k0 + j, N - 1);
  const u8* rows_b = rowp + (int64_t)b * N * ldrow;
  f16x8 kh[4], kl[4], vh[4], vl[4];
  load_row_frags(rows_b + (int64_t)key * ldrow + (D + h * 64) * 4, h2, kh, kl);
  load_row_frags(rows_b + (int64_t)key * ldrow + (2 * D + h * 64) * 4, h2, vh, vl);
  const u8* qbase = rows_b + (int64_t)(h * 64) * 4;
  const u8* gbase = dorow + (int64_t)b * N * lddo + (int64_t)(h * 64) * 4;
  const float* lse_b = lse + (int64_t)bh * N;
  const float* del_b = delta + (int64_t)bh * N;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  auto issue = [&](int qt, int buf) {
    const unsigned st = lds0 + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 8 / NW; ++i) {
      const int c = wave + NW * i;
      dma_row_chunk(st, c, qbase, (unsigned)ldrow, 32 * qt, N - 1, lane);
      dma_row_chunk(st + 8192, c, gbase, (unsigned)lddo, 32 * qt, N - 1, lane);
    }
    if (wave == 0) {   // lse | delta of the query tile through LDS as well (lanes 0-31 | 32-63: two wave-uniform bases)
      const unsigned vo = (unsigned)min(32 * qt + (lane & 31), N - 1) * 4u;
      const float* sb = lane < 32 ? lse_b : del_b;
      asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" ::"v"(reinterpret_cast<const u8*>(sb) + vo), "s"(st + 16384u) : "memory");
    }
  };
  const float isg = slot[EAV_SLOT_ISIGMA], isd = slot_do[EAV_SLOT_ISIGMA];
  const float c1 = scale * LOG2E * isg * isg;
  const float dsc = slot[EAV_SLOT_SIGMA] * slot_do[EAV_SLOT_SIGMA];
  const float s2 = sigma_from_bits(slot_bits(slot_ds));   // exact scale of t = dS 2^-22 (measured by the dQ kernel)
  f32x16 gk0, gk1, gv0, gv1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { gk0[r] = 0.f; gk1[r] = 0.f; gv0[r] = 0.f; gv1[r] = 0.f; }
  const int nqt = (N + 31) / 32;
  const int prow = pi_row(j);
  float neg1;
  asm volatile("s_mov_b32 %0, 0xbf800000" : "=s"(neg1));
  // fragment addresses inside a tile: the hi pieces; the lo piece of the same fragment is the address ^ 16 (the hl bit is
  // bit 0 of the XOR-swizzled piece index) - one v_xor at the read instead of eight more live registers (this kernel sits at
  // the 256-register limit of two waves per SIMD; with both tables resident hipcc spilled eight of them to scratch)
  int rofs[4];                                  // row fragments of the Q / dO tiles, per head_dim step
#pragma unroll
  for (int st = 0; st < 4; ++st) rofs[st] = prow * 256 + (((4 * st + 2 * h2) ^ (prow & 15)) << 4);
  int tofs[2][2];                               // transposing reads: [dbase / 32][token + 4]
  {
    const int G = lane >> 4, r = (lane >> 2) & 3, qq = lane & 3;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int ab = 0; ab < 2; ++ab) {
        const int piece = 2 * (4 * db + 2 * (G & 1) + (qq >> 1));
        const int tok = 8 * (G >> 1) + r + 4 * ab;
        tofs[db][ab] = tok * 256 + ((piece ^ (tok & 15)) << 4) + ((qq & 1) << 3);
      }
  }
  // t = dS 2^-22 s2 in operand units = p . [(dP - delta dsc) kq], kq = 2^-22 s2 (a power of two: exact)
  const float kq = DS_DOWN * s2, dk = dsc * kq;
  // one query tile (queries beyond N in the ragged last tile: their lse entry is +inf - see top - so p = 0 without a mask)
  auto tile = [&](auto BUFc, int qt) {
    constexpr int BUF = decltype(BUFc)::value;
    const u8* qr_ = smem + BUF * STAGE;
    const u8* gr_ = qr_ + 8192;
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const f16x8 ah = *reinterpret_cast<const f16x8*>(qr_ + rofs[st]), al = *reinterpret_cast<const f16x8*>(qr_ + (rofs[st] ^ 16));
      const f16x8 bh_ = *reinterpret_cast<const f16x8*>(gr_ + rofs[st]), bl_ = *reinterpret_cast<const f16x8*>(gr_ + (rofs[st] ^ 16));
      s = MFMA16(ah, kh[st], s);          // S[q][key] = Q . K^T
      dp = MFMA16(bh_, vh[st], dp);       // dP[q][key] = dO . V^T
      s = MFMA16(al, kh[st], s);
      dp = MFMA16(bl_, vh[st], dp);
      s = MFMA16(ah, kl[st], s);
      dp = MFMA16(bh_, vl[st], dp);
      if (st & 1) SBAR();                 // (at most two steps' fragments in flight: the kernel sits at 256 registers)
    }
    auto tfrag = [&](const u8* tile_, int st, int db, int hl) {
      f16x8 out;
      out.lo = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4_ptr)(tile_ + (tofs[db][0] ^ (16 * hl)) + st * 4096)));
      out.hi = __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (lds_s16x4_ptr)(tile_ + (tofs[db][1] ^ (16 * hl)) + st * 4096)));
      return out;
    };
    const float* ls = reinterpret_cast<const float*>(qr_ + 16384);
#pragma unroll
    for (int g = 0; g < 2; ++g) {       // sixteen queries at a time (token step g of the second products): their probabilities
                                        // and dS leave as fragments and go straight into the twelve MFMAs of the step
      const int ql = 16 * g + 8 * h2;                    // 8 consecutive queries <-> registers 8g .. 8g+7
      const float4 l0 = *reinterpret_cast<const float4*>(ls + ql), l1 = *reinterpret_cast<const float4*>(ls + ql + 4);
      const float4 d0 = *reinterpret_cast<const float4*>(ls + 32 + ql);
      const float4 d1 = *reinterpret_cast<const float4*>(ls + 36 + ql);
      const float lqv[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
      const float dlv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
      float pt[8], t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = 8 * g + e;
        // (the 2^14 of the probability and the 2^-22 s2 of dS are NOT folded into the exponent: an argument near 14 instead
        // of near 0 costs the exponential 7e-7 of relative accuracy - the spike case of test_attention_sp_is_fp32_grade)
        const float p = ex2(fmaf(s[r], c1, -lqv[e] * LOG2E));
        pt[e] = p * SP;
        t[e] = p * fmaf(dp[r], kq, -dlv[e] * dk);
      }
      f16x8 ph, pl, th, tl;
      split_frag_mix(pt, neg1, ph, pl);
      split_frag_mix(t, neg1, th, tl);
      {                                    // dV^T[d][key] += dO^T[d][q] . P[q][key]
        const f16x8 a0h = tfrag(gr_, g, 0, 0), a0l = tfrag(gr_, g, 0, 1);
        const f16x8 a1h = tfrag(gr_, g, 1, 0), a1l = tfrag(gr_, g, 1, 1);
        gv0 = MFMA16(a0h, ph, gv0);
        gv1 = MFMA16(a1h, ph, gv1);
        gv0 = MFMA16(a0l, ph, gv0);
        gv1 = MFMA16(a1l, ph, gv1);
        gv0 = MFMA16(a0h, pl, gv0);
        gv1 = MFMA16(a1h, pl, gv1);
      }
      {                                    // dK^T[d][key] += Q^T[d][q] . dS[q][key]
        const f16x8 a0h = tfrag(qr_, g, 0, 0), a0l = tfrag(qr_, g, 0, 1);
        const f16x8 a1h = tfrag(qr_, g, 1, 0), a1l = tfrag(qr_, g, 1, 1);
        gk0 = MFMA16(a0h, th, gk0);
        gk1 = MFMA16(a1h, th, gk1);
        gk0 = MFMA16(a0l, th, gk0);
        gk1 = MFMA16(a1l, th, gk1);
        gk0 = MFMA16(a0h, tl, gk0);
        gk1 = MFMA16(a1h, tl, gk1);
      }
      SBAR();
    }
  };
  auto top = [&](int qt, int buf) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == 0 && 32 * qt + 32 > N) {         // ragged last tile: lse = +inf for the queries beyond N (wave 0 staged the
      if (lane < 32 && 32 * qt + lane >= N)      // entries - its own DMA has landed - and every wave reads them after the barrier)
        reinterpret_cast<float*>(smem + buf * STAGE + 16384)[lane] = INFINITY;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (qt + 1 < nqt) issue(qt + 1, buf ^ 1);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  issue(0, 0);
  if (k0 >= N) {                                 // (wave-uniform) no valid key: this wave only stages its share of the tiles
    for (int qt = 0; qt < nqt; ++qt) top(qt, qt & 1);
  } else {
    int qt = 0;
    for (; qt + 2 <= nqt; qt += 2) {
      top(qt, 0);
      tile(I0{}, qt);
      top(qt + 1, 1);
      tile(I1{}, qt + 1);
    }
    if (qt < nqt) {
      top(qt, 0);
      tile(I0{}, qt);
    }
  }
  __syncthreads();
  float* base = dqkv ? dqkv + (int64_t)b * N * 3 * D + h * 64 : nullptr;
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * 33);
  const float mk = scale * (1.f / DS_DOWN) / s2 * isd * isg * isg;
  const float mv = isd / SP;
  const int nrb = (N + 31) / 32;
  u8* gp_ = gplanes ? gplanes + (int64_t)b * N * (3 * D * 4) + h * 256 : nullptr;     // (see attn_bwd_q_sp_kernel)
  float* cs_ = cs_part && k0 < N ? cs_part + ((int64_t)b * nrb + k0 / 32) * (3 * D) + h * 64 : nullptr;
  const float gsig = gplanes ? dqkv_sigma(slot_do, slot, N, scale, lane) : 0.f;     // (= what the dQ kernel published)
  float vmax = store_rows_T(patch, gk0, gk1, mk, base ? base + D : nullptr, 3 * D, k0, N, lane, gp_ ? gp_ + D * 4 : nullptr,
                            (int64_t)3 * D * 4, gsig, cs_ ? cs_ + D : nullptr);
  vmax = fmaxf(vmax, store_rows_T(patch, gv0, gv1, mv, base ? base + 2 * D : nullptr, 3 * D, k0, N, lane,
                                  gp_ ? gp_ + 2 * D * 4 : nullptr, (int64_t)3 * D * 4, gsig, cs_ ? cs_ + 2 * D : nullptr));
  emit_amax(amax_out, vmax, lane, (int)blockIdx.x * NW + wave, k0 < N ? b * N + k0 : -1);
}

// ------------------------------------------------------------------------------------------------ operand preparation
// src [B*N, ncols] fp32 -> row planes and / or T planes (see the header).  One 64-token x 64-column tile per block;
// grid (ncols/64, ceil(Npad/64), B).  tmask bit s set: write the T planes of column section s (sections of secw columns).
__global__ __launch_bounds__(256) void attn_sp_prep_kernel(const float* __restrict__ src, float* __restrict__ slot,
                                                           u8* __restrict__ rowp, u8* __restrict__ tp, int N, int Npad,
                                                           int ncols, int secw, unsigned tmask) {
  // hi / lo pieces leave through LDS images in linear order: whole 256-byte row segments per store instruction instead of
  // 16-byte chunks with 16-byte holes (the same change took eav_sp_convert from 4.3 to 5.4 TB/s)
  __shared__ float tile[64][65];
  __shared__ uint4 pimg[64 * 17];
  const float sigma = sigma_from_bits(slot_bits(slot));
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
    slot[EAV_SLOT_SIGMA] = sigma;
    slot[EAV_SLOT_ISIGMA] = 1.f / sigma;
  }
  const int chunk = blockIdx.x, t0 = blockIdx.y * 64, b = blockIdx.z;
  const int c0 = chunk * 64;
  const bool wantT = tp && ((tmask >> (c0 / secw)) & 1u);
  const int t = threadIdx.x;
  {
    const int cg = t & 7, rr = t >> 3;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int tok = t0 + rr + 32 * pass, col = c0 + 8 * cg;
      float tv[8];
      if (tok < N) {
        const float* p = src + ((int64_t)b * N + tok) * ncols + col;
        const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
        tv[0] = a.x; tv[1] = a.y; tv[2] = a.z; tv[3] = a.w; tv[4] = c.x; tv[5] = c.y; tv[6] = c.z; tv[7] = c.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) tv[e] *= sigma;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) tv[e] = 0.f;
      }
      if (rowp) {
        f16x8 hi, lo;
        split_frag(tv, 1.f, hi, lo);
        pimg[(rr + 32 * pass) * 17 + 2 * cg] = *reinterpret_cast<const uint4*>(&hi);
        pimg[(rr + 32 * pass) * 17 + 2 * cg + 1] = *reinterpret_cast<const uint4*>(&lo);
      }
      if (wantT) {
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[rr + 32 * pass][8 * cg + e] = tv[e];
      }
    }
  }
  __syncthreads();
  if (rowp) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = t + 256 * k, r = p >> 4, pc = p & 15;
      if (t0 + r < N)
        *reinterpret_cast<uint4*>(rowp + ((int64_t)b * N + t0 + r) * ncols * 4 + (c0 >> 3) * 32 + pc * 16) = pimg[r * 17 + pc];
    }
  }
  if (!wantT) return;
  __syncthreads();          // pimg is reused for the transposed image
  {
    const int rg = t & 7, cc = t >> 3;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int d = cc + 32 * pass;
      float tv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) tv[e] = tile[8 * rg + e][d];
      f16x8 hi, lo;
      split_frag(tv, 1.f, hi, lo);
      pimg[d * 17 + 2 * rg] = *reinterpret_cast<const uint4*>(&hi);
      pimg[d * 17 + 2 * rg + 1] = *reinterpret_cast<const uint4*>(&lo);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = t + 256 * k, d = p >> 4, pc = p & 15;
    if (t0 + 8 * (pc >> 1) < Npad)
      *reinterpret_cast<uint4*>(tp + (((int64_t)b * (ncols / 64) + chunk) * 64 + d) * Npad * 4 + (t0 >> 3) * 32 + pc * 16) =
          pimg[d * 17 + pc];
  }
}

// sigma / 1 / sigma of slot_out from a BOUND of |dqkv| (the scale of the planes eav_attn_bwd_sp_planes writes): with
// m = max|dO| (slot_do's shards) and q = max|qkv| (measured: slot_qkv's shards; or the bound 2^15 / sigma_qkv):  |dP|, |delta| <= 64 m q, hence
// |dS_ij| <= 128 P_ij m q;  |dQ_i| <= scale 128 m q^2 (row sums of P are 1), |dK_j| <= N scale 128 m q^2, |dV_j| <= N m.
// The bound is loose (by N and more); the planes keep fp32-grade ABSOLUTE accuracy regardless: fp16 denormal operands
// are honoured by the MFMA (tools/probes/mfma_denorm.hip), so an element far below the scale loses relative, not absolute,
// precision (error <= 2^-50 of the bound).
__global__ __launch_bounds__(64) void attn_dqkv_bound_kernel(float* __restrict__ slot_out, const float* __restrict__ slot_do,
                                                             const float* __restrict__ slot_qkv, int N, float scale) {
  const unsigned bits = eav_slot_bits(slot_do);
  if (threadIdx.x == 0) {
    // q: the measured maximum when the producer of qkv left it in the slot's shards (the fused projection does), else the
    // bound its scale stands for (2^15 / sigma)
    const unsigned qb = eav_slot_bits(slot_qkv);
    const float m = __uint_as_float(bits), q = qb ? __uint_as_float(qb) : 32768.f * slot_qkv[EAV_SLOT_ISIGMA];
    const float bound = (float)N * m * fmaxf(1.f, 128.f * scale * q * q) * 1.0001f;
    const float sg = sigma_from_bits(__float_as_uint(bound));
    slot_out[EAV_SLOT_SIGMA] = sg;
    slot_out[EAV_SLOT_ISIGMA] = 1.f / sg;
  }
}

}  // namespace

extern "C" int eav_attn_sp_npad(int N) { return (N + 31) / 32 * 32; }

extern "C" int eav_attn_sp_prep(const float* src, float* slot, void* rowp, void* tp, int B, int N, int ncols, int secw,
                                unsigned tmask, void* stream) {
  EAV_REQUIRE(src && slot && (rowp || tp) && B > 0 && N > 0 && ncols > 0 && ncols % 64 == 0 && secw > 0 &&
                  secw % 64 == 0 && ncols % secw == 0,
              "eav_attn_sp_prep: columns and sections must be multiples of the head dimension 64");
  const int Npad = eav_attn_sp_npad(N);
  hipLaunchKernelGGL(attn_sp_prep_kernel, dim3(ncols / 64, cdiv(Npad, 64), B), dim3(256), 0, (hipStream_t)stream, src,
                     slot, (u8*)rowp, (u8*)tp, N, Npad, ncols, secw, tmask);
  EAV_CHECK_LAUNCH("eav_attn_sp_prep");
  return EAV_OK;
}

// 4-wave (128-row) workgroups above this sequence length, 2-wave ones below: measured on ViT (N = 197) forward 106 -> 86 us,
// backward 418 -> 379 us with 4 waves (half as many re-reads of a head's K / V planes, one idle wave instead of one idle
// half-block); tuning hook
int g_nw4_above = 128;
int g_fwd2_from = 512;        // forward: the software-pipelined kernel from this sequence length on (negative argument of the hook)
extern "C" int eav_attn_sp_set_nw4_above(int n) {
  if (n < 0) g_fwd2_from = -n;
  else g_nw4_above = n;
  return 0;
}

// ao_planes (optional): the output as the GEMM operand planes [B*N][D/8][2][8] of the o-proj products, scaled with qkv's own
// sigma (|O| <= max|V|), which the kernel copies into ao_slot; ao may then be null (no fp32 copy: forward-only passes).
extern "C" int eav_attn_fwd_sp_planes(const void* rowp, const void* tp, const float* slot, float* ao, float* lse,
                                      float* amax_slot, void* ao_planes, float* ao_slot, int B, int H, int N, int head_dim,
                                      float scale, void* stream) {
  EAV_REQUIRE(rowp && slot && (ao || ao_planes) && lse && B > 0 && H > 0 && N > 0 && (!ao_planes || ao_slot),
              "eav_attn_fwd_sp: bad arguments");
  EAV_REQUIRE(head_dim == 64, "eav_attn_fwd_sp: head_dim %d unsupported (needs 64)", head_dim);
  const int Npad = eav_attn_sp_npad(N);
  hipStream_t st = (hipStream_t)stream;
  // long sequences (AST: 1214 tokens): the software-pipelined kernel, 152 -> 120 us per layer at B = 8; short ones (ViT: 197
  // tokens, 7 key tiles, bound by its HBM traffic and its per-workgroup prologue) stay on the plain loop (86 against 105 us)
  if (N >= g_fwd2_from) {
    hipLaunchKernelGGL(attn_fwd_sp2_kernel<4>, dim3(cdiv(N, 128) * B * H), dim3(256), 0, st, (const u8*)rowp, slot, ao, lse,
                       (unsigned*)amax_slot, N, H, scale, cdiv(N, 128), B * H, (u8*)ao_planes, ao_slot);
    EAV_CHECK_LAUNCH("eav_attn_fwd_sp");
    return EAV_OK;
  }
  if (N > g_nw4_above) {
    hipLaunchKernelGGL(attn_fwd_sp_kernel<4>, dim3(cdiv(N, 128) * B * H), dim3(256), 0, st, (const u8*)rowp,
                       (const u8*)tp, slot, ao, lse, (unsigned*)amax_slot, N, Npad, H, scale, cdiv(N, 128), B * H,
                       (u8*)ao_planes, ao_slot);
  } else {
    hipLaunchKernelGGL(attn_fwd_sp_kernel<2>, dim3(cdiv(N, 64) * B * H), dim3(128), 0, st, (const u8*)rowp,
                       (const u8*)tp, slot, ao, lse, (unsigned*)amax_slot, N, Npad, H, scale, cdiv(N, 64), B * H,
                       (u8*)ao_planes, ao_slot);
  }
  EAV_CHECK_LAUNCH("eav_attn_fwd_sp");
  return EAV_OK;
}

extern "C" int eav_attn_fwd_sp(const void* rowp, const void* tp, const float* slot, float* ao, float* lse,
                               float* amax_slot, int B, int H, int N, int head_dim, float scale, void* stream) {
  EAV_REQUIRE(ao, "eav_attn_fwd_sp: bad arguments");
  return eav_attn_fwd_sp_planes(rowp, tp, slot, ao, lse, amax_slot, nullptr, nullptr, B, H, N, head_dim, scale, stream);
}

extern "C" int eav_attn_dqkv_bound(float* slot_out, const float* slot_do, const float* slot_qkv, int N, float scale,
                                   void* stream) {
  EAV_REQUIRE(slot_out && slot_do && slot_qkv && N > 0 && scale > 0.f, "eav_attn_dqkv_bound: bad arguments");
  hipLaunchKernelGGL(attn_dqkv_bound_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slot_out, slot_do, slot_qkv, N, scale);
  EAV_CHECK_LAUNCH("eav_attn_dqkv_bound");
  return EAV_OK;
}

// delta: scratch [B*H, N].  slot_ds: scratch slot (zeroed by the caller).  dqkv [B*N, 3*H*64] fp32 (optional with planes).
// planes (optional): dqkv also (or only) leaves as the operand planes [B*N][3D/8][2][8] of the q/k/v projection's gradient
// products, scaled by the bound of eav_attn_dqkv_bound, which the kernels form themselves from slot_do's and slot's shard
// words and publish in planes_slot (sigma, 1 / sigma); colsum_part (optional)
// [B * ceil(N / 32)][3 D]: per 32-row tile the column sums of dqkv (finish the bias gradient with eav_reduce_partials).
// ao_planes + ao_slot (optional): the attention output as the forward's o-proj operand planes (eav_attn_fwd_sp_planes) - the
// row sums delta = dO . O are then formed from them and from dO's planes, and ao / dout (fp32) are not read and may be NULL.
extern "C" int eav_attn_bwd_sp_planes(const void* rowp, const void* tp, const void* dorow, const void* dotp,
                                      const float* slot, const float* slot_do, float* slot_ds, const float* ao,
                                      const float* dout, const float* lse, float* delta, float* dqkv, float* amax_slot,
                                      void* planes, float* planes_slot, float* colsum_part, const void* ao_planes,
                                      const float* ao_slot, int B, int H, int N, int head_dim, float scale, void* stream) {
  EAV_REQUIRE(rowp && dorow && slot && slot_do && slot_ds && ((ao && dout) || (ao_planes && ao_slot)) && lse && delta &&
                  (dqkv || planes) && B > 0 && H > 0 && N > 0 && (!planes || planes_slot), "eav_attn_bwd_sp: bad arguments");
  EAV_REQUIRE(head_dim == 64, "eav_attn_bwd_sp: head_dim %d unsupported (needs 64)", head_dim);
  const int Npad = eav_attn_sp_npad(N);
  hipStream_t st = (hipStream_t)stream;
  const int nbh = B * H, nt128 = cdiv(N, 128), nt64 = cdiv(N, 64);
  if (N > g_nw4_above) {
    hipLaunchKernelGGL(attn_bwd_q_sp_kernel<4>, dim3(nt128 * nbh), dim3(256), 0, st, (const u8*)rowp, (const u8*)tp,
                       (const u8*)dorow, slot, slot_do, lse, ao, dout, delta, dqkv, (unsigned*)slot_ds,
                       (unsigned*)amax_slot, N, Npad, H, scale, nt128, nbh, (u8*)planes, planes_slot, colsum_part,
                       (const u8*)ao_planes, ao_slot);
    EAV_CHECK_LAUNCH("eav_attn_bwd_sp(dQ)");
  } else {
    hipLaunchKernelGGL(attn_bwd_q_sp_kernel<2>, dim3(nt64 * nbh), dim3(128), 0, st, (const u8*)rowp, (const u8*)tp,
                       (const u8*)dorow, slot, slot_do, lse, ao, dout, delta, dqkv, (unsigned*)slot_ds,
                       (unsigned*)amax_slot, N, Npad, H, scale, nt64, nbh, (u8*)planes, planes_slot, colsum_part,
                       (const u8*)ao_planes, ao_slot);
    EAV_CHECK_LAUNCH("eav_attn_bwd_sp(dQ)");
  }
  // the dK,dV kernel holds 66 KB of tiles per block: 4-wave blocks keep 2 waves per SIMD at every N (a wave past the last
  // key only stages tiles)
  hipLaunchKernelGGL(attn_bwd_kv_sp_kernel<4>, dim3(nt128 * nbh), dim3(256), 0, st, (const u8*)rowp, (const u8*)tp,
                     (const u8*)dorow, (const u8*)dotp, slot, slot_do, slot_ds, lse, delta, dqkv, (unsigned*)amax_slot, N,
                     Npad, H, scale, nt128, nbh, (u8*)planes, planes_slot, colsum_part);
  EAV_CHECK_LAUNCH("eav_attn_bwd_sp(dK,dV)");
  return EAV_OK;
}

extern "C" int eav_attn_bwd_sp(const void* rowp, const void* tp, const void* dorow, const void* dotp, const float* slot,
                               const float* slot_do, float* slot_ds, const float* ao, const float* dout,
                               const float* lse, float* delta, float* dqkv, float* amax_slot, int B, int H, int N,
                               int head_dim, float scale, void* stream) {
  EAV_REQUIRE(dqkv, "eav_attn_bwd_sp: bad arguments");
  return eav_attn_bwd_sp_planes(rowp, tp, dorow, dotp, slot, slot_do, slot_ds, ao, dout, lse, delta, dqkv, amax_slot,
                                nullptr, nullptr, nullptr, nullptr, nullptr, B, H, N, head_dim, scale, stream);
}
