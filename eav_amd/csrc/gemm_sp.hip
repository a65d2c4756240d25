// fp32-grade GEMM on the gfx950 fp16 matrix cores with split operands ("sp16" planes).
//
// Every dense contraction of the AST / ViT training step (nn.Linear forward, data gradient, weight gradient, the
// patch-embedding projection: Transformer_Audio.py:72, Transformer_Vision.py:92 through the HF modelling code) runs
// here by default instead of on the exact-fp32 MFMA (gemm_f32.hip, 157 TFLOP/s peak).  An fp32 operand v is held as
//     hi = fp16(sigma v),  lo = fp16((sigma v - hi) 2^11)      sigma = 2^e with max|sigma v| in [2^14, 2^15)
// and a product is three v_mfma_f32_32x32x16_f16: hi.hi into one fp32 accumulator, lo.hi + hi.lo into a second one that
// is folded in with 2^-11 at the end (fp16 x fp16 products are exact in fp32; the dropped lo.lo term is <= 2^-22 of the
// product; the lift keeps lo a normal fp16 number for elements down to 2^-29 of the tensor maximum).  The result is
// fp32-grade (measured against float64 beside the exact-fp32 kernel: tests/test_split_kernels_gpu.py) at a third of the
// 2.5 PFLOP/s fp16 peak instead of 1/16.
//
// Operand format ("planes"): a matrix X[R, K] whose CONTRACTION index is the column index is stored as
//     uint16 planes[R][Kp/8][2][8]      Kp = K rounded up to 32, zero beyond K
// i.e. for every 8 consecutive k the 16-byte hi piece is followed by the 16-byte lo piece - one piece is exactly one
// lane's MFMA operand fragment, and a 32-deep K-tile of a row is one 128-byte line.  There is ONE GEMM layout
// (C[m,n] = sum_k A[m,k] B[n,k]); the data / weight gradient products use planes of the transposed tensors, which
// eav_sp_convert writes in the same pass (the tensor is converted once per step where it is produced, not re-rounded
// per tile).  sigma lives in a device "slot" of EAV_SP_SLOT floats (eav_common.h): 64 shards of the bits of max|v|, one
// per 128-byte line (producers atomicMax one shard each - integer max of non-negative floats is order-independent, hence
// deterministic; separate lines keep the L2 atomics from serialising), then sigma and 1/sigma, which eav_sp_convert
// writes; the GEMM folds 1/(sigma_A sigma_B) into alpha.  Nothing crosses to the host.
//
// Kernel: 256 x 128 x 32 tiles, 8 waves, three LDS stages, one workgroup per CU for products that fill the chip;
// 128 x 128 x 32, 4 waves, two stages, 2 workgroups per CU for smaller ones and for the token-contracting (weight-gradient)
// mode; each wave a 64 x 64 block of 2 x 2 MFMA tiles = 24 MFMAs per K-tile.  Operands go HBM/L2 -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write), ONE barrier per K-tile.  LDS image: row r, piece p (0..7) at
// 16-byte slot r*8 + (p ^ ((r>>1)&7)); the XOR is applied to the per-lane SOURCE address (the LDS-DMA destination is
// lane-linear) and again on the fragment read, which makes every ds_read_b128 conflict-free.  Workgroups are persistent:
// tile ids are remapped so that each XCD's L2 sees a compact group of tiles (8 tile-rows x all tile-columns at a time),
// and the next tile's first two stages are in flight under the current tile's epilogue.  The epilogue turns each
// accumulator block row-linear through a per-wave LDS patch (whole 128-byte row segments per global access).
#include <algorithm>
#include <type_traits>

#include "eav_common.h"
#include "../../include/eav_hip.h"
#include "../../include/eav_hip_tuning.h"

// timing-only ablation of the main loop (tools/probes/tr_ablate.sh builds variants; results are garbage): 1 = no fragment
// reads, 2 = no LDS-DMA, 4 = no MFMAs, 16 / 32 = every tile streams the A / B rows of tile 0 (L2-hot operand), 64 = no
// epilogue at all (the K loop alone: what bias / GELU / planes / maxima / stores cost a short-K product), 128 = the linear
// epilogue without its global stores (arithmetic, LDS patch and loads stay), 256 = its stores wrapped into a 1-MB window of
// the output (L2-resident: store issue and acknowledgement without the HBM write stream)
#ifndef EAV_ABL
#define EAV_ABL 0
#endif


namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ds_read_b64_tr_b16: within each group of 16 lanes the sixteen 8-byte reads form a [4][16] fp16 block - lane 4r + q
// supplies row r, columns 4q .. 4q+3 - and lane j receives column j of the four rows (measured on gfx950 with
// tools/probes/tr16_probe.hip).  With rows = contraction index (tokens) and columns = features, two such reads give a
// lane the eight consecutive-k values of an MFMA operand fragment from a TOKEN-major image: the weight-gradient products
// contract over tokens directly from the row planes, no transposed copy of any activation / gradient tensor exists.
__device__ __forceinline__ f16x4 lds_read_tr16(const unsigned char* p) {
  return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p));
}
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct SpArgs {
  const unsigned char* A;   // planes [M][Kp/8][2][8]
  const unsigned char* B;   // planes [N][Kp/8][2][8]
  float* C;
  const float* slotA;       // operand-scale slots (eav_common.h: 64 shards of max|x| bits, sigma, 1/sigma)
  const float* slotB;
  const float* bias;        // [N] or null
  const float* resid;       // [M,N] (ldr) or null, added after the activation
  float* pre;               // [M,N] (ldc) or null: value before the activation
  unsigned* amax;           // or null: atomicMax of the bits of |stored value|
  int noblk;                // amax: tensor-wide shards only, no 128-row block entries (EAV_GEMM_NO_BLOCKMAX)
  float* colsum;            // or null: [ceil(M / 64)][N] column sums of the stored value over blocks of 64 rows (bias gradient)
  unsigned char* planes;    // or null: the stored value (after the activation) also leaves as row planes [M][Np/8][2][8],
  const float* slotP;       //   scaled by slotP[EAV_SLOT_SIGMA] - a scale known BEFORE the launch (eav_tf_forward_scales);
  int64_t ldp;              //   row pitch of the planes in bytes; C may then be null (no fp32 copy of the value)
  float lomul;
  int M, N, nkt;            // nkt = Kp / 32
  int64_t ldA, ldB;         // row strides in bytes
  int ldc, ldr;
  int64_t sA, sC;           // batch strides (blockIdx.z): A in bytes, C / pre in floats (B is shared)
  float alpha;
  int gelu, accumulate;
  int order;                // tile order inside an XCD (see tile_origin)
  int kt_per_split;         // > 0: split-K, slice z covers K-tiles [z*kt_per_split, ...), C[z] = partial slab
  int tm, tn;
};

// erf-GELU without the libm erff (two divergent branches, ~60 instructions): with z = |x| / sqrt 2,
//   erfc(z) = t (a1 + t (a2 + ... + a6 t^5)) exp(-z^2),  t = 1 / (1 + 0.55 z)       (fit error 1.7e-8 on [0, 4.2])
// and gelu(x) = x erfc(z) / 2 for x < 0, x - x erfc(z) / 2 otherwise - no 1 + erf cancellation on the negative side.
// Measured over [-6, 6] against float64: max |error| 2.5e-7 (0.5 x (1 + erff(x / sqrt 2)) in fp32: 4.5e-7), max
// |error| / |x| 1.7e-7.  One v_rcp_f32, one v_exp_f32, 12 other VALU instructions.
__device__ __forceinline__ float gelu_erf(float x) {
#pragma clang fp contract(off)   // one rounding sequence wherever this is inlined (GEMM epilogue and conversion pass agree bit for bit)
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.55f, z, 1.0f));
  float p = 0.10732425004243851f;
  p = fmaf(p, t, -0.5180373191833496f);
  p = fmaf(p, t, 0.7597128748893738f);
  p = fmaf(p, t, -0.04142485186457634f);
  p = fmaf(p, t, 0.3908415734767914f);
  p = fmaf(p, t, 0.30158352851867676f);
  const float q = p * t * __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
  const float h = 0.5f * x * q;
  return x < 0.f ? h : x - h;
}

// d gelu / dx = Phi(x) + x phi(x) from the same erfc fit: Phi = erfc(z) / 2 for x < 0, 1 - erfc(z) / 2 otherwise
__device__ __forceinline__ float gelu_erf_grad(float x) {
#pragma clang fp contract(off)
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.55f, z, 1.0f));
  float p = 0.10732425004243851f;
  p = fmaf(p, t, -0.5180373191833496f);
  p = fmaf(p, t, 0.7597128748893738f);
  p = fmaf(p, t, -0.04142485186457634f);
  p = fmaf(p, t, 0.3908415734767914f);
  p = fmaf(p, t, 0.30158352851867676f);
  const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
  const float hq = 0.5f * p * t * e;
  return (x < 0.f ? hq : 1.0f - hq) + x * e * 0.3989422804014327f;
}

// v summed over the lanes l ^ 8, l ^ 16, l ^ 32 combinations (the 8 lanes that share l & 7) without LDS round trips:
// row_ror:8 pairs l with l ^ 8 inside a row of 16; v_permlane16_swap / v_permlane32_swap of the value with a copy of
// itself leave {even rows, odd rows} / {lower half, upper half} side by side in two registers (tools/probes/permlane_probe)
__device__ __forceinline__ float sum_over_lane_bits_3_4_5(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true));
  auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Four consecutive values of one row -> this lane's half of the 8-value hi / lo pieces; the partner lane (the other half of
// the group of 8 columns) gets the halves exchanged so that each lane stores one whole 16-byte piece: `first` (the lane
// holding columns 0-3 of the group) the hi piece, the other one the lo piece.  xorm = lane distance of the partner.
__device__ __forceinline__ uint4 plane_piece4(float t0, float t1, float t2, float t3, float lomul, bool first, int xorm) {
  _Float16 h[4] = {(_Float16)t0, (_Float16)t1, (_Float16)t2, (_Float16)t3};
  _Float16 l[4] = {(_Float16)((t0 - (float)h[0]) * lomul), (_Float16)((t1 - (float)h[1]) * lomul),
                   (_Float16)((t2 - (float)h[2]) * lomul), (_Float16)((t3 - (float)h[3]) * lomul)};
  const uint2 hh = *reinterpret_cast<const uint2*>(h), ll = *reinterpret_cast<const uint2*>(l);
  const uint2 send = first ? ll : hh;
  uint2 recv;
  if (xorm == 1) {      // neighbouring lanes: a DPP quad permutation [1,0,3,2] - no LDS round trip (ds_bpermute) in the epilogue
    recv.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.x, 0xB1, 0xf, 0xf, true);
    recv.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)send.y, 0xB1, 0xf, 0xf, true);
  } else {
    recv.x = __shfl_xor(send.x, xorm, 64);
    recv.y = __shfl_xor(send.y, xorm, 64);
  }
  return first ? make_uint4(hh.x, hh.y, recv.x, recv.y) : make_uint4(recv.x, recv.y, ll.x, ll.y);
}

// WM x WN waves, each a (32 RM) x (32 RN) block of RM x RN MFMA tiles.
//
// TR = false: C[m,n] = sum_k A[m,k] B[n,k] - both operands are row planes whose COLUMN index is contracted (forward and
// data-gradient products).  TR = true: C[m,n] = sum_t A[t,m] B[t,n] - the ROW index (token) of both row planes is
// contracted (weight-gradient products).  A K-tile is then 32 tokens x 128 features of each operand (512 B of every
// token row: 16 hi/lo piece pairs), staged token-major: piece (t, w = 2 fg + hl) at 16-byte slot t 32 + (w ^ s(t)),
// s(t) = (t & 1) | ((t & 2) << 2), applied to the per-lane SOURCE address of the LDS-DMA as in the other mode.  A
// fragment is two ds_read_b64_tr_b16 (tokens 4h .. 4h+3, h = 0, 1); the 32 lanes of a half-wave then touch the 16
// distinct 16-byte slots {row r = t & 3} x {4 feature groups} in both 8-byte halves: conflict-free.  Rows of the planes
// beyond the token count (up to the next multiple of 32) must be ZERO - they are contracted like real tokens.
//
// TERMS = 3: the fp32-grade product above.  TERMS = 1: hi.hi only - a plain fp16 product of the scaled operands (11-bit
// operand mantissas, fp32 accumulation), a third of the MFMA work on the same planes; the opt-in mode of the backward
// products (Encoder.grad_terms = 1; the forward - the logits - stays on three terms).  TERMS = 2 (token-contracting form
// only): hi_A.hi_B + lo_A.hi_B - operand B rounded to fp16, operand A at full split precision.  In a weight gradient A is the
// GRADIENT tensor (loosely scaled a-priori planes: it needs its lo piece) and B the ACTIVATION: two thirds of the matrix
// work, B's lo pieces are not even read from LDS, and the producers' fused gradient planes stay usable (a hi.hi-only product
// needs tightly scaled operands on BOTH sides, which switches them off: profiles/r05_term_budget.txt).
//
// NS = LDS stages.  2: a stage is in flight for one K-tile period.  3 (the 256 x 128 / 8-wave form, one workgroup per CU;
// the epilogue patches then alias stage 2): two periods, and 96 KB instead of 64 KB in flight per CU - the loop is bound by
// the latency of the L2 -> LDS stream, not by the matrix pipes (timing ablation, tools/probes/tr_ablate.py: DMA alone
// 290 us, MFMAs alone 199 us of the 342 us of a [25216 x 3072] x [768 x 3072]^T product).
template <int WM, int WN, int RM, int RN, bool TWOACC, bool TR, int TERMS = 3, int NS = 2>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN <= 4) ? 2 : 1) void gemm_sp_kernel(SpArgs g) {
  constexpr int NW = WM * WN, BM = 32 * RM * WM, BN = 32 * RN * WN;
  constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
  constexpr int NCH = (BM + BN) / 8;      // 1-KB chunks (8 rows x 128 B; TR: 2 tokens x 512 B) per stage
  constexpr int CPW = NCH / NW;           // chunks per wave
  constexpr int NT = RM * RN;             // MFMA tiles per wave
  constexpr int NMF = TERMS * NT;         // MFMAs per K-step of 16
  // fragment reads per K-step, in the order ah[], bh[], al[], bl[] (TERMS = 2 stops in front of bl)
  constexpr int NRD = (TR ? 2 : 1) * (TERMS == 1 ? RM + RN : TERMS == 2 ? 2 * RM + RN : 2 * (RM + RN));
  static_assert(TERMS == 3 || (TERMS == 1 && !TWOACC) || (TERMS == 2 && TWOACC && TR),
                "three terms, the hi.hi term alone in one accumulator, or hi.hi + lo_A.hi_B of the token-contracting form");
  static_assert(NCH % NW == 0 && CPW <= 2 * NMF && NRD <= (TR ? 2 : 1) * NMF,
                "stage chunks / fragment reads must fit the MFMA slots");
  static_assert(!TR || (BM == 128 && BN == 128 && RM == 2 && RN == 2), "the token-major image is laid out for 128 x 128");
  static_assert(NS == 2 || (NS == 3 && NW * 4096 <= STAGE), "two stages, or three with the patches inside the third");
  // + one 32 x 32 fp32 patch per wave: the epilogue turns accumulator blocks into row-linear order through it
  // (at offset 2 STAGE in both forms; NS = 3: that is stage 2, free while the epilogue runs)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NS == 2 ? 2 * STAGE + NW * 4096 : 3 * STAGE];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // ---- persistent workgroups: tile ids blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x is a multiple of 8 whenever a
  // workgroup gets more than one tile, so a workgroup stays on its XCD).  tile id -> tile: XCD-contiguous, then groups
  // of 8 tile-rows x all tile-columns, so that each XCD's L2 sees a compact set of operand panels.
  const int nb = g.tm * g.tn;
  auto tile_origin = [&](int id, int& m0, int& n0) {
    const int q = nb >> 3, r = nb & 7, xcd = id & 7, j = id >> 3;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    if (g.order == 1) {            // row-major: the tn tiles of a tile-row are consecutive (see gemm_sp_impl)
      m0 = (lin / g.tn) * BM;
      n0 = (lin % g.tn) * BN;
      return;
    }
    const int gsz = 8 * g.tn, grp = lin / gsz, first_m = grp * 8;
    const int gm = min(8, g.tm - first_m), rem = lin - grp * gsz;
    m0 = (first_m + rem % gm) * BM;
    n0 = (rem / gm) * BN;
  };

  const int z = blockIdx.z, tl0 = blockIdx.x, tstep = gridDim.x;
  const unsigned char* Ab = g.A;
  float* C = g.C;
  int kt0 = 0, kt1 = g.nkt;
  if (g.kt_per_split > 0) {
    kt0 = z * g.kt_per_split;
    kt1 = min(g.nkt, kt0 + g.kt_per_split);
    C += (int64_t)z * g.M * g.ldc;
  } else {
    Ab += z * g.sA;
    C += z * g.sC;
  }
  const int nk = kt1 - kt0;
  if (nk <= 0) return;

  // ---- operand sources: one SCALAR base per operand (the tile's first row at the first K-tile of this workgroup's K range)
  // and one 32-bit byte offset per lane and chunk - the saddr form of the LDS-DMA.  (Round 5 kept a 64-bit pointer per lane
  // and chunk and advanced each one by a VALU add pair per K-tile: 2 CPW registers and 2 CPW VALU instructions per K-tile
  // that the scalar unit now carries.)  Rows are clamped: ragged tiles re-read the last row.
  constexpr int ACH = BM / (8 * NW);                     // chunks wave + NW i with i < ACH belong to the A tile
  static_assert(BM % (8 * NW) == 0, "a wave's chunk i is an A chunk for every wave or for none");
  unsigned voff[CPW];
  const unsigned char *baseA = Ab, *baseB = g.B;
  auto set_sources = [&](int m0, int n0) {
    int lane = t & 63;                     // (laundered, like the epilogue's: the per-lane terms below are tile-invariant and
    asm volatile("" : "+v"(lane));         //  would otherwise be hoisted out of the tile loop and spilled around the K loop)
    if constexpr (TR) {
      baseA = Ab + (int64_t)kt0 * 32 * g.ldA;
      baseB = g.B + (int64_t)kt0 * 32 * g.ldB;
    } else {
      if (EAV_ABL & 16) m0 = 0;
      if (EAV_ABL & 32) n0 = 0;
      baseA = Ab + (int64_t)m0 * g.ldA + (int64_t)kt0 * 128;
      baseB = g.B + (int64_t)n0 * g.ldB + (int64_t)kt0 * 128;
    }
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int c = wave + NW * i;                       // wave-uniform chunk id
      const bool isA = i < ACH;
      if constexpr (TR) {
        const int tk = 2 * (isA ? c : c - BM / 8) + (lane >> 5);          // token within the K-tile
        const int w = (lane & 31) ^ ((tk & 1) | ((tk & 2) << 2));          // piece held by this lane's slot
        const int fg = w >> 1, hl = w & 1;
        const int ld = (int)(isA ? g.ldA : g.ldB);
        const int grp = min((isA ? m0 : n0) / 8 + fg, (ld >> 5) - 1);      // ragged feature tiles re-read the last group
        voff[i] = (unsigned)(tk * ld + grp * 32 + hl * 16);
      } else {
        const int row = 8 * c + (lane >> 3);               // row in the combined [A tile; B tile] image
        const int p = (lane & 7) ^ ((row >> 1) & 7);       // piece held by this lane's slot
        if (isA) voff[i] = (unsigned)(min(row, g.M - 1 - m0) * (int)g.ldA + p * 16);
        else voff[i] = (unsigned)(min(row - BM, g.N - 1 - n0) * (int)g.ldB + p * 16);
      }
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  // The LDS-DMA is issued from inline asm, not through __builtin_amdgcn_global_load_lds: hipcc orders every LDS access
  // it cannot disambiguate behind the outstanding DMA builtins with an `s_waitcnt vmcnt(0)` - the transposing reads of the
  // TR loop (each DMA was waited for right after its issue) and the epilogue's patch reads (the next tile's first stages
  // were drained before the epilogue could start).  An asm DMA is invisible to that pass; the waits that order DMA and
  // fragment reads are the explicit ones of this kernel (mid-tile vmcnt + barrier, tile start).
  // kt: K-tile (relative to this workgroup's K range) the stage holds.
  auto issue1 = [&](int buf, int i, int kt) {
    const int c = wave + NW * i;
    const bool isA = i < ACH;
    const unsigned dst = lds0 + buf * STAGE + c * 1024;
    const unsigned char* sb = (isA ? baseA : baseB) + (int64_t)kt * (TR ? 32 * (isA ? g.ldA : g.ldB) : (int64_t)128);
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff[i]), "s"(sb), "s"(dst) : "memory");   // (m0 is a reserved register: nothing hipcc emits for gfx950 in this kernel depends on it)
  };

  // ---- fragment addressing
  const int wm = wave / WN, wn = wave - wm * WN;
  const int r32 = lane & 31, kh = lane >> 5, gq = (r32 >> 1) & 7;
  int lowp[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) lowp[ks][hl] = ((2 * (2 * ks + kh) + hl) ^ gq) * 16;
  const int offA = (wm * 32 * RM + r32) * 128, offB = A_BYTES + (wn * 32 * RN + r32) * 128;
  // TR: lane = 16 g4 + 4 kk + c supplies token 8 (g4 >> 1) + kk (+ 4 h + 16 ks), features 16 (g4 & 1) + 4 c .. + 3 of tile i
  int trA[RM][2], trB[RN][2];
  if constexpr (TR) {
    const int g4 = lane >> 4, kk = (lane >> 2) & 3, c = lane & 3;
    const int lb = (8 * (g4 >> 1) + kk) * 512 + (g4 & 1) * 64 + (c >> 1) * 32 + (c & 1) * 8;
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
#pragma unroll
      for (int i = 0; i < RM; ++i) trA[i][hl] = lb + wm * 256 + ((i ^ (kk >> 1)) << 7) + ((hl ^ (kk & 1)) << 4);
#pragma unroll
      for (int j = 0; j < RN; ++j) trB[j][hl] = A_BYTES + lb + wn * 256 + ((j ^ (kk >> 1)) << 7) + ((hl ^ (kk & 1)) << 4);
    }
  }

  // ---- main loop: software-pipelined and explicitly interleaved (sched_barrier pins the order as written).
  // Two fragment register sets: f0 = K-step 0 of a K-tile, f1 = K-step 1.  Iteration t:
  //   phase A: MFMAs on f0(t)   || ds_read f1(t) from buffer t&1
  //   lgkmcnt(0), vmcnt(0) [stage t+1 landed], barrier
  //   phase B: MFMAs on f1(t)   || ds_read f0(t+1) from buffer (t+1)&1 || LDS-DMA of stage t+2 into buffer t&1
  // so LDS reads, the next stage's DMA issue and their address arithmetic all hide behind MFMA issue slots; a stage has
  // a whole K-tile's MFMA time to land.  Every wave's reads of buffer t&1 are complete (lgkmcnt(0)) before the barrier
  // that precedes its re-fill.  One MFMA slot = one MFMA, then (pinned behind it) at most one fragment read and one DMA.
  // In the LAST K-tile's phase B both buffers are free: the NEXT output tile's stages 0 and 1 are issued there, so they
  // land under this tile's epilogue (a K = 768 problem otherwise spends a quarter of its time in cold prologues).
  struct Frags { f16x8 ah[RM], al[RM], bh[RN], bl[RN]; };
  Frags f0, f1;
  f32x16 acc[RM][RN], acx[TWOACC ? RM : 1][TWOACC ? RN : 1];
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MM(c, a, b) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
  // fragment read number r of a K-step: ah[0..RM), bh[0..RN), al[0..RM), bl[0..RN)
  // TR: read r of a K-step (0 .. 4 (RM + RN)): fragment r >> 1 in the order ah[], bh[], al[], bl[], token half r & 1
  auto read_frag_tr = [&](Frags& f, const unsigned char* stage, int ks, int r) {
    const int fi = r >> 1, h = r & 1;
    const unsigned char* p = stage + ks * 8192 + h * 2048;
#define EAV_TRRD(dst, off) { const f16x4 v = lds_read_tr16(p + (off)); if (h) dst.hi = v; else dst.lo = v; }
    if (fi < RM) EAV_TRRD(f.ah[fi], trA[fi][0])
    else if (fi < RM + RN) EAV_TRRD(f.bh[fi - RM], trB[fi - RM][0])
    else if (fi < 2 * RM + RN) EAV_TRRD(f.al[fi - RM - RN], trA[fi - RM - RN][1])
    else EAV_TRRD(f.bl[fi - 2 * RM - RN], trB[fi - 2 * RM - RN][1])
#undef EAV_TRRD
  };
  auto read_frag = [&](Frags& f, const unsigned char* sa, const unsigned char* sb, int ks, int r) {
    if (r < RM) f.ah[r] = *reinterpret_cast<const f16x8*>(sa + r * 4096 + lowp[ks][0]);
    else if (r < RM + RN) f.bh[r - RM] = *reinterpret_cast<const f16x8*>(sb + (r - RM) * 4096 + lowp[ks][0]);
    else if (r < 2 * RM + RN) f.al[r - RM - RN] = *reinterpret_cast<const f16x8*>(sa + (r - RM - RN) * 4096 + lowp[ks][1]);
    else f.bl[r - 2 * RM - RN] = *reinterpret_cast<const f16x8*>(sb + (r - 2 * RM - RN) * 4096 + lowp[ks][1]);
  };
  // MFMA number m of a K-step: the hi.hi products of every tile, then lo.hi, then hi.lo
  auto mfma_slot = [&](const Frags& f, int m) {
    const int term = m / NT, tt = m - term * NT, i = tt / RN, j = tt - i * RN;
    // the B-tile fragment goes in as the MFMA's first operand: the accumulator block is the TRANSPOSE of the output
    // block - lane = one output row, registers 4q..4q+3 = four consecutive columns (16-byte epilogue accesses)
    if (term == 0) MM(acc[i][j], f.bh[j], f.ah[i]);
    else if (TWOACC) {
      if (term == 1) MM(acx[i][j], f.bh[j], f.al[i]);
      else MM(acx[i][j], f.bl[j], f.ah[i]);
    } else {
      if (term == 1) MM(acc[i][j], f.bh[j], f.al[i]);
      else MM(acc[i][j], f.bl[j], f.ah[i]);
    }
  };
  // more: a K-tile t+1 exists (read its first fragments); more2: a K-tile t+2 exists (issue its DMA);
  // next_m0 >= 0 (last K-tile only): issue the next output tile's first two stages instead
  // TR: boost exponents of the token blocks (= K-tiles: 32 tokens) of this workgroup's K range, one block per lane (a window
  // of 64 K-tiles, reloaded when the walk crosses it).  A boosted block's fragments are scaled back by 2^-k before their
  // MFMAs: the boost serves the products whose OUTPUT rows are these rows; in a contraction over the rows a small row's
  // share of the sum is small anyway, and un-boosting (fp16, may round into the subnormals) costs nothing next to the
  // large rows.  `anyb` (no lane holds a boost: the common case) keeps the whole mechanism out of the K loop.
  int kvecA = 0, kvecB = 0;
  bool anyb = false;
  auto load_boost_window = [&](int blk0) {
    if constexpr (TR) {
      const int b = (blk0 + lane) & (EAV_SLOT_NBLK - 1);
      kvecA = reinterpret_cast<const int*>(g.slotA)[EAV_SLOT_BEXP + b];
      kvecB = reinterpret_cast<const int*>(g.slotB)[EAV_SLOT_BEXP + b];
      anyb = __any((kvecA | kvecB) != 0);
    }
  };
  auto pow2h = [](int k) -> _Float16 {       // 2^-k as fp16 (subnormal for 15 <= k <= 24, 0 beyond)
    const unsigned short bits = k <= 14 ? (unsigned short)((15 - k) << 10) : (k <= 24 ? (unsigned short)(1u << (24 - k)) : 0);
    return __builtin_bit_cast(_Float16, bits);
  };
  auto unboost = [&](Frags& f, int ka, int kb) {
    if (ka) {
      const _Float16 sa = pow2h(ka);
#pragma unroll
      for (int i = 0; i < RM; ++i) {
        f.ah[i] *= sa;
        if constexpr (TERMS > 1) f.al[i] *= sa;
      }
    }
    if (kb) {
      const _Float16 sb = pow2h(kb);
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        f.bh[j] *= sb;
        if constexpr (TERMS == 3) f.bl[j] *= sb;
      }
    }
  };
  // flags of K-tile t: more = a K-tile t + 1 exists (read its first fragments); more2 = a K-tile t + NS exists (issue its
  // DMA into this K-tile's buffer); newer (NS = 3) = a K-tile t + 2 exists: its stage may stay in flight over the mid-tile
  // wait.  buf / nbuf = buffers of K-tiles t / t + 1.
  auto iter = [&](auto more_c, auto more2_c, auto newer_c, int t, int buf, int nbuf, int next_m0, int next_n0) {
    constexpr bool more = decltype(more_c)::value, more2 = decltype(more2_c)::value, newer = decltype(newer_c)::value;
    int ka = 0, kb = 0;
    if constexpr (TR) {
      if (t > 0 && (t & 63) == 0) load_boost_window(kt0 + t);
      if (anyb) {
        ka = __builtin_amdgcn_readlane(kvecA, t & 63);
        kb = __builtin_amdgcn_readlane(kvecB, t & 63);
        if (ka | kb) unboost(f0, ka, kb);
      }
    }
    const unsigned char* sa = smem + buf * STAGE + offA;
    const unsigned char* sb = smem + buf * STAGE + offB;
    const unsigned char* sa2 = smem + nbuf * STAGE + offA;
    const unsigned char* sb2 = smem + nbuf * STAGE + offB;
    SB();
#pragma unroll
    for (int m = 0; m < NMF; ++m) {          // phase A
      if (!(EAV_ABL & 4)) mfma_slot(f0, m);
      SB();
      if (!(EAV_ABL & 1)) {
        if constexpr (TR) {
          read_frag_tr(f1, smem + buf * STAGE, 1, m);
          if (m + NMF < NRD) read_frag_tr(f1, smem + buf * STAGE, 1, m + NMF);
        } else {
          if (m < NRD) read_frag(f1, sa, sb, 1, m);
        }
      }
      SB();
    }
    if constexpr (newer) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(CPW) : "memory");   // all but the newest stage
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    SB();
    if constexpr (TR) {
      if (ka | kb) unboost(f1, ka, kb);
    }
    if (!more && next_m0 >= 0) set_sources(next_m0, next_n0);
#pragma unroll
    for (int m = 0; m < NMF; ++m) {          // phase B
      if (!(EAV_ABL & 4)) mfma_slot(f1, m);
      SB();
      if (!(EAV_ABL & 1)) {
        if constexpr (TR) {
          if (more) {
            read_frag_tr(f0, smem + nbuf * STAGE, 0, m);
            if (m + NMF < NRD) read_frag_tr(f0, smem + nbuf * STAGE, 0, m + NMF);
          }
        } else {
          if (more && m < NRD) read_frag(f0, sa2, sb2, 0, m);
        }
      }
      if (more2 && m < CPW && !(EAV_ABL & 2)) issue1(buf, m, t + NS);
      if (more2 && m + NMF < CPW && !(EAV_ABL & 2)) issue1(buf, m + NMF, t + NS);          // (one term: 4 slots for 8 chunks)
      if (!more && next_m0 >= 0) {
        if (m < CPW) issue1(0, m, 0);
        else if (m < 2 * CPW && nk > 1) issue1(1, m - CPW, 1);
      }
      SB();
    }
    if (!more && next_m0 >= 0) {                 // the rest of the next tile's first two stages (2 CPW may exceed the slots)
#pragma unroll
      for (int m = NMF; m < 2 * CPW; ++m) {
        if (m < CPW) issue1(0, m, 0);
        else if (nk > 1) issue1(1, m - CPW, 1);
      }
    }
  };

  bool primed = false;
  for (int tl = tl0; tl < nb; tl += tstep) {
    int m0, n0, nm0 = -1, nn0 = -1;
    tile_origin(tl, m0, n0);
    if (tl + tstep < nb) tile_origin(tl + tstep, nm0, nn0);
    if (!primed) {
      set_sources(m0, n0);
#pragma unroll
      for (int i = 0; i < CPW; ++i) issue1(0, i, 0);
      if (nk > 1) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) issue1(1, i, 1);
      }
      primed = true;
    }
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[i][j][r] = 0.f;
          if (TWOACC) acx[i][j][r] = 0.f;
        }
    // stages 0 and 1 of this tile (issued in the prologue, or under the previous tile's last K-tile and epilogue)
    if constexpr (NS == 3) {
      __builtin_amdgcn_s_barrier();          // every wave is done with its epilogue patch (inside stage 2)
      if (nk > 2) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) issue1(2, i, 2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int r = 0; r < NRD; ++r) {
      if constexpr (TR) read_frag_tr(f0, smem, 0, r);
      else read_frag(f0, smem + offA, smem + offB, 0, r);
    }
    load_boost_window(kt0);
    {
      constexpr std::true_type Y{};
      constexpr std::false_type F{};
      int t = 0, b = 0;
      auto nx = [](int x) { return x + 1 == NS ? 0 : x + 1; };
      if constexpr (NS == 2) {
        for (; t + 2 < nk; ++t, b = nx(b)) iter(Y, Y, F, t, b, nx(b), -1, -1);
      } else {
        for (; t + 3 < nk; ++t, b = nx(b)) iter(Y, Y, Y, t, b, nx(b), -1, -1);
        if (t + 2 < nk) { iter(Y, F, Y, t, b, nx(b), -1, -1); ++t; b = nx(b); }
      }
      if (t + 1 < nk) { iter(Y, F, F, t, b, nx(b), -1, -1); ++t; b = nx(b); }
      if (t < nk) iter(F, F, F, t, b, nx(b), nm0, nn0);
    }

    // ---- epilogue of this tile (the next tile's first stages are in flight)
    // The epilogue's per-lane index arithmetic is tile-invariant: left alone, hipcc hoists it out of the persistent tile loop
    // and - the main loop sits at the 256-register budget - spills it once at kernel entry to reload it in every epilogue
    // (36-196 bytes of scratch per lane in round 5, none of it inside the K loop).  A laundered lane id makes it tile-local:
    // a handful of VALU instructions per tile instead of scratch traffic.
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int lane = lane_e, r32 = lane_e & 31, kh = lane_e >> 5;
    if (TWOACC) {
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += acx[TWOACC ? i : 0][TWOACC ? j : 0][r] * (1.f / 2048.f);
    }
    float alpha = g.alpha * g.slotA[EAV_SLOT_ISIGMA] * g.slotB[EAV_SLOT_ISIGMA];
    // row of the A planes this wave's output rows come from (batched launches: z selects a slab of sA / ldA rows)
    const int arow = (g.kt_per_split > 0 || g.ldA == 0 ? 0 : (int)(z * (g.sA / g.ldA))) + m0 + wm * 32 * RM;
    // per-row-block boosts of the operands (EAV_SLOT_BEXP, one entry per 32 rows = per MFMA tile; 0 unless a block was >= 2^8
    // below its tensor's maximum): the alpha of accumulator block (i, j) undoes 2^(kA[i] + kB[j])
    float alf[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
      for (int j = 0; j < RN; ++j) alf[i][j] = alpha;
    if constexpr (!TR) {
      int kA[RM], kB[RN];
#pragma unroll
      for (int i = 0; i < RM; ++i)
        kA[i] = reinterpret_cast<const int*>(g.slotA)[EAV_SLOT_BEXP + (((arow + 32 * i) >> EAV_BLK_SHIFT) & (EAV_SLOT_NBLK - 1))];
#pragma unroll
      for (int j = 0; j < RN; ++j)
        kB[j] = reinterpret_cast<const int*>(g.slotB)[EAV_SLOT_BEXP + (((n0 + wn * 32 * RN + 32 * j) >> EAV_BLK_SHIFT) & (EAV_SLOT_NBLK - 1))];
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) {
          const int kx = kA[i] + kB[j];
          if (kx) alf[i][j] = alpha * __uint_as_float((unsigned)(127 - min(kx, 126)) << 23);
        }
    }
    const float psig = g.planes ? g.slotP[EAV_SLOT_SIGMA] : 0.f;
    const int M = g.M, N = g.N;
    float vmax = 0.f;
    float* pre = g.pre ? g.pre + (g.kt_per_split > 0 ? 0 : z * g.sC) : nullptr;
    // Accumulator block (i, j) holds output rows rowb + (lane & 31), columns colb + 4 kh + 8 q + e in register 4 q + e.
    // Every run-time option (pre-activation store, GELU, residual, accumulate) is tested once per block, all loads of a
    // block are issued before its arithmetic.  Element form: ragged tiles and unaligned operands.
    auto epilogue_elements = [&]() {
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int col = n0 + wn * 32 * RN + 32 * j + 4 * kh;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = col + 8 * (r >> 2) + (r & 3);
          bv[r] = (g.bias && c < N) ? g.bias[c] : 0.f;
        }
        float csum[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) csum[r] = 0.f;
#pragma unroll
        for (int i = 0; i < RM; ++i) {
          const int row = m0 + wm * 32 * RM + 32 * i + r32;
          const bool rowok = row < M;
          auto ok = [&](int r) { return rowok && col + 8 * (r >> 2) + (r & 3) < N; };
          const int64_t o = (int64_t)row * g.ldc + col;
          f32x16& a = acc[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) a[r] = alf[i][j] * a[r] + bv[r];
          if (g.gelu == 2) {        // backward through GELU: scale by gelu'(pre), pre = the forward's stored pre-activation
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] *= gelu_erf_grad(ok(r) ? pre[o + 8 * (r >> 2) + (r & 3)] : 0.f);
          } else if (pre) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (ok(r)) pre[o + 8 * (r >> 2) + (r & 3)] = a[r];
          }
          if (g.gelu == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = gelu_erf(a[r]);
          }
          if (g.gelu == 3) {          // C receives the PRE-activation; only max|GELU| is published (eav_sp_convert_gelu
                                      // applies the activation while it splits - the activation tensor never exists in fp32)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float gv = fabsf(gelu_erf(a[r]));
              vmax = fmaxf(vmax, ok(r) ? gv : 0.f);
            }
          }
          if (g.resid) {
            const float* rp = g.resid + (int64_t)row * g.ldr + col;
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[r] = ok(r) ? rp[8 * (r >> 2) + (r & 3)] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] += rv[r];
          }
          if (g.accumulate) {
            float cv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) cv[r] = ok(r) ? C[o + 8 * (r >> 2) + (r & 3)] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] += cv[r];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (ok(r)) {
              if (g.C) C[o + 8 * (r >> 2) + (r & 3)] = a[r];
              if (g.gelu != 3) vmax = fmaxf(vmax, fabsf(a[r]));
            }
          }
          if (g.planes) {     // lanes l and l ^ 32 hold the two halves of every group of 8 columns (N % 8 == 0)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint4 pc = plane_piece4(a[4 * q] * psig, a[4 * q + 1] * psig, a[4 * q + 2] * psig, a[4 * q + 3] * psig,
                                            g.lomul, kh == 0, 32);
              const int cg = col - 4 * kh + 8 * q;          // first column of the group
              if (rowok && cg + 7 < N)
                *reinterpret_cast<uint4*>(g.planes + (int64_t)row * g.ldp + (int64_t)(cg >> 3) * 32 + kh * 16) = pc;
            }
          }
          if (g.colsum) {     // column sums over this block's 32 rows (one row per lane of a wave half), fixed order
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float t = ok(r) ? a[r] : 0.f;
#pragma unroll
              for (int o = 1; o < 32; o <<= 1) t += __shfl_xor(t, o, 64);
              csum[r] += t;
            }
          }
          __builtin_amdgcn_sched_barrier(0);   // keep the next block's loads behind this block's stores (register pressure)
        }
        if (g.colsum && r32 == 0 && m0 + wm * 32 * RM < M) {
          float* cp = g.colsum + (int64_t)((m0 + wm * 32 * RM) >> 6) * N;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = col + 8 * (r >> 2) + (r & 3);
            if (c < N) cp[c] = csum[r];
          }
        }
      }
    };
    // Row-linear form (tile inside the matrix, aligned rows): the B-tile fragment is the MFMA's first operand, so a lane
    // holds one output row x 16 columns; an accumulator block goes through the
    // wave's LDS patch (16-byte slots XOR-swizzled by row: conflict-free both ways) and comes back with lane l owning
    // columns 4 (l & 7) .. + 3 of rows (l >> 3) + 8 k - every global access of a wave instruction is then 8 whole
    // 128-byte row segments instead of 32 rows x 32 bytes (the planes conversion gained 25 % from the same change).
    // (EAV_ABL store ablations of the linear epilogue)
    auto st_off = [&](int64_t o) -> int64_t { return (EAV_ABL & 256) ? (o & ((int64_t)(1 << 18) - 1)) : o; };      // floats
    auto st_offb = [&](int64_t o) -> int64_t { return (EAV_ABL & 256) ? (o & ((int64_t)(1 << 20) - 1)) : o; };     // bytes
    auto epilogue_linear = [&]() {
      float* patch = reinterpret_cast<float*>(smem + 2 * STAGE + wave * 4096);
      const int prow = lane >> 3, pc4 = lane & 7;
#pragma unroll
      for (int j = 0; j < RN; ++j) {
        const int colb = n0 + wn * 32 * RN + 32 * j;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias) b4 = *reinterpret_cast<const float4*>(g.bias + colb + 4 * pc4);
        float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < RM; ++i) {
          const int rowb = m0 + wm * 32 * RM + 32 * i;
          f32x16& a = acc[i][j];
#pragma unroll
          for (int q = 0; q < 4; ++q)     // lane (row r32, kh) holds columns 4 kh + 8 q .. + 3: 16-byte slot kh + 2 q
            *reinterpret_cast<float4*>(patch + r32 * 32 + (((kh + 2 * q) ^ (r32 & 7)) << 2)) =
                make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          float v[16];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int r = prow + 8 * k;
            const float4 x = *reinterpret_cast<const float4*>(patch + r * 32 + ((pc4 ^ (r & 7)) << 2));
            v[4 * k] = alf[i][j] * x.x + b4.x; v[4 * k + 1] = alf[i][j] * x.y + b4.y;
            v[4 * k + 2] = alf[i][j] * x.z + b4.z; v[4 * k + 3] = alf[i][j] * x.w + b4.w;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the patch is free for the next block
          const int64_t o = (int64_t)(rowb + prow) * g.ldc + colb + 4 * pc4;     // + 8 k ldc per k
          if (g.gelu == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float4 p4 = *reinterpret_cast<const float4*>(pre + o + (int64_t)8 * k * g.ldc);
              v[4 * k] *= gelu_erf_grad(p4.x); v[4 * k + 1] *= gelu_erf_grad(p4.y);
              v[4 * k + 2] *= gelu_erf_grad(p4.z); v[4 * k + 3] *= gelu_erf_grad(p4.w);
            }
          } else if (pre && !(EAV_ABL & 128)) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              *reinterpret_cast<float4*>(pre + st_off(o + (int64_t)8 * k * g.ldc)) =
                  make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
          }
          if (g.gelu == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = gelu_erf(v[r]);
          }
          if (g.gelu == 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) vmax = fmaxf(vmax, fabsf(gelu_erf(v[r])));
          }
          if (g.resid) {
            const float* rp = g.resid + (int64_t)(rowb + prow) * g.ldr + colb + 4 * pc4;
            float4 r4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r4[k] = *reinterpret_cast<const float4*>(rp + (int64_t)8 * k * g.ldr);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v[4 * k] += r4[k].x; v[4 * k + 1] += r4[k].y; v[4 * k + 2] += r4[k].z; v[4 * k + 3] += r4[k].w;
            }
          }
          if (g.accumulate) {
            float4 c4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) c4[k] = *reinterpret_cast<const float4*>(C + o + (int64_t)8 * k * g.ldc);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v[4 * k] += c4[k].x; v[4 * k + 1] += c4[k].y; v[4 * k + 2] += c4[k].z; v[4 * k + 3] += c4[k].w;
            }
          }
          if (g.C && !(EAV_ABL & 128)) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              *reinterpret_cast<float4*>(C + st_off(o + (int64_t)8 * k * g.ldc)) =
                  make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
          }
          if (g.planes) {     // neighbouring lanes hold the two halves of a group of 8 columns: 16-byte piece stores
            const int64_t po = (int64_t)(rowb + prow) * g.ldp + (int64_t)((colb + 4 * pc4) >> 3) * 32 + (pc4 & 1) * 16;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              *reinterpret_cast<uint4*>((EAV_ABL & 128) ? reinterpret_cast<unsigned char*>(patch) + 64 * lane
                                                        : g.planes + st_offb(po + (int64_t)8 * k * g.ldp)) =
                  plane_piece4(v[4 * k] * psig, v[4 * k + 1] * psig, v[4 * k + 2] * psig, v[4 * k + 3] * psig, g.lomul,
                               (pc4 & 1) == 0, 1);
          }
          if (g.gelu != 3) {
#pragma unroll
            for (int r = 0; r < 16; ++r) vmax = fmaxf(vmax, fabsf(v[r]));
          }
          if (g.colsum) {     // lane = rows prow + 8 k, columns 4 pc4 .. + 3: four rows in the lane, then the 8 lanes of a column group
            float4 t = make_float4((v[0] + v[4]) + (v[8] + v[12]), (v[1] + v[5]) + (v[9] + v[13]),
                                   (v[2] + v[6]) + (v[10] + v[14]), (v[3] + v[7]) + (v[11] + v[15]));
            t.x = sum_over_lane_bits_3_4_5(t.x); t.y = sum_over_lane_bits_3_4_5(t.y);
            t.z = sum_over_lane_bits_3_4_5(t.z); t.w = sum_over_lane_bits_3_4_5(t.w);
            cs4.x += t.x; cs4.y += t.y; cs4.z += t.z; cs4.w += t.w;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (g.colsum && prow == 0)
          *reinterpret_cast<float4*>(g.colsum + (int64_t)((m0 + wm * 32 * RM) >> 6) * N + colb + 4 * pc4) = cs4;
      }
    };
    const bool aligned = (((uintptr_t)C | (uintptr_t)pre | (uintptr_t)g.resid | (uintptr_t)g.bias | (uintptr_t)g.planes |
                           (uintptr_t)g.colsum) & 15) == 0 &&
                         ((g.ldc | g.ldr | (int)(g.sC & 3) | (g.colsum ? N : 0)) & 3) == 0;
    if constexpr ((EAV_ABL & 64) != 0) {       // keep the accumulators alive, write nothing
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
      if (sacc == 12345.678f && C) C[0] = alf[0][0] + psig + (pre ? 1.f : 0.f);
    } else {
      if (aligned && m0 + BM <= M && n0 + BN <= N) epilogue_linear();
      else epilogue_elements();
    }
    if (g.amax) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
      if (lane == 0 && vmax == vmax) {
        atomicMax(g.amax + EAV_SLOT_SHARD(tl * NW + wave + 17 * z), __float_as_uint(vmax));
        // (rows of the OUTPUT: same numbering for z = 0; skipped on request - at N = 768 the 24 waves of a tile-row, resident
        // together, queue on one address: 93 -> 123 us for a [25216 x 768] x [768 x 768]^T product)
        if constexpr (!TR) {
          if (!g.noblk) {      // (the wave's maximum for each of its RM 32-row blocks: an over-estimate per block, which is safe)
#pragma unroll
            for (int i = 0; i < RM; ++i) eav_slot_blockmax(g.amax, arow + 32 * i, vmax);
          }
        }
      }
    }
  }
#undef MM
#undef SB
}

int g_order = 0;
int g_force_tile = 0;   // test / tuning hook: 0 = heuristic, 1 = 128x128, 2 = 256x128
int g_small_tile = 0;   // test / tuning hook: 0 = heuristic, 1 = 64x128 wherever it applies, 2 = never
// 11 (default): lo = fp16((t - hi) 2^11), cross terms in a second accumulator folded in with 2^-11 - both pieces stay
// normal fp16 numbers for elements down to 2^-29 of the tensor maximum.  0: lo = fp16(t - hi), one accumulator (64 fewer
// VGPRs, same speed at 2 waves per SIMD; full precision only down to 2^-15 of the maximum) - kept as a tuning hook.
int g_loshift = 11;
int g_persist = 1;      // tuning hook: 0 = one workgroup per output tile
int g_splitk_force = 0; // tuning hook (eav_gemm_sp_set_splitk): slices of eav_gemm_sp_splitk, 0 = the plan

// Returns false when no kernel is instantiated for the requested combination (the single-accumulator tuning mode
// g_loshift = 0 exists for the 128 x 128 two-stage column-contracting form only): the caller reports an error instead of
// returning EAV_OK over stale output.
template <int WM, int WN, int RM, int RN, bool TR = false, int NS = 2>
bool launch(SpArgs& g, int nz, hipStream_t st, int terms = 3) {
  g.tm = cdiv(g.M, 32 * RM * WM);
  g.tn = cdiv(g.N, 32 * RN * WN);
  // persistent workgroups: as many as stay resident (2 per CU for 4-wave tiles, 1 for 8-wave tiles), a multiple of 8 so
  // that every workgroup's tiles stay on one XCD; fewer tiles than that: one workgroup per tile
  const int resident = (WM * WN <= 4 ? 2 : 1) * 256 * (g_persist ? 1 : 1 << 20);
  const int nb = g.tm * g.tn, gx = nb <= resident ? nb : resident;
  if (terms == 1) {
    if constexpr (RM * RN >= 4) {      // (the one-term form has too few MFMA slots per K-step for the chunks of a 64-row tile)
      hipLaunchKernelGGL((gemm_sp_kernel<WM, WN, RM, RN, false, TR, 1, NS>), dim3(gx, 1, nz), dim3(64 * WM * WN), 0, st, g);
      return true;
    }
    return false;
  } else if (terms == 2) {
    if constexpr (TR) {
      if (!g_loshift) return false;
      hipLaunchKernelGGL((gemm_sp_kernel<WM, WN, RM, RN, true, true, 2, NS>), dim3(gx, 1, nz), dim3(64 * WM * WN), 0, st, g);
      return true;
    }
    return false;
  } else if (g_loshift) {
    if constexpr (RM * RN <= 4) {
      hipLaunchKernelGGL((gemm_sp_kernel<WM, WN, RM, RN, true, TR, 3, NS>), dim3(gx, 1, nz), dim3(64 * WM * WN), 0, st, g);
      return true;
    }
  } else {
    if constexpr (!TR && NS == 2) {
      hipLaunchKernelGGL((gemm_sp_kernel<WM, WN, RM, RN, false, false>), dim3(gx, 1, nz), dim3(64 * WM * WN), 0, st, g);
      return true;
    }
  }
  return false;
}

bool dispatch(SpArgs& g, int nz, hipStream_t st, int terms = 3, bool shared_gpu = false) {
  // 128 x 128 tiles x two workgroups per CU when the product has the GPU to itself (the forward): alone the two forms are
  // within +-5 % on the encoder shapes and the small one wins where the epilogue is heavy or the tile count quantises
  // badly (fc1 + GELU, o-proj; tools/gemm_sp_bench.py; forward-only step ViT B=128 17.35 against 18.1 ms).  256 x 128 / 8
  // waves / three LDS stages (one workgroup per CU, 3/4 of the L2 -> LDS bytes per flop, two stages in flight) when the
  // caller says another persistent GEMM is running beside this one (EAV_GEMM_SHARED_GPU: the backward's data-gradient
  // products next to the side stream's weight gradients) and the product fills the chip: its 144 KB of LDS keep the other
  // kernel's workgroups off the CUs it runs on - the two take turns instead of sharing every CU's LDS bandwidth and L2
  // (ViT B=128 step 57.6 -> 55.9 ms, tools/encoder_step_bench.py with SP_TILE=1 / 0 on one box; +10 % alone at 8192^3).
  const bool big = g_force_tile == 2 || (g_force_tile == 0 && shared_gpu && cdiv(g.M, 256) * cdiv(g.N, 128) >= 256);
  if (g_force_tile == 3) return launch<2, 4, 2, 2, false, 3>(g, nz, st, terms);
  if (big) return launch<4, 2, 2, 2, false, 3>(g, nz, st, terms);
  // Fewer 128 x 128 tiles than CUs (the N = 768 products at the per-rank batches of a data-parallel group: ViT B = 16 is
  // 25 x 6 = 150 tiles, a launch then lasts one tile's K loop on 150 of 256 CUs): 64 x 128 tiles, four waves of 32 x 64 -
  // twice the workgroups of half the length.  (Not with column sums: their partial rows are per 64 output rows of ONE wave.)
  if (g_force_tile == 0 && g_small_tile != 2 && terms == 3 && g_loshift && !g.colsum && g.M > 64 &&
      (g_small_tile == 1 || cdiv(g.M, 128) * cdiv(g.N, 128) * nz <= 256))
    return launch<2, 2, 1, 2>(g, nz, st, terms);
  return launch<2, 2, 2, 2>(g, nz, st, terms);
}

bool dispatch_tr(SpArgs& g, int nz, hipStream_t st, int terms = 3) { return launch<2, 2, 2, 2, true>(g, nz, st, terms); }

__global__ void sp_splitk_reduce_kernel(const float* __restrict__ ws, int nsplit, int64_t n, float* __restrict__ out,
                                        int accumulate) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  double a = 0, b = 0, c = 0, d = 0;
  for (int s = 0; s < nsplit; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(ws + (int64_t)s * n + i);
    a += v.x; b += v.y; c += v.z; d += v.w;
  }
  float4 o = make_float4((float)a, (float)b, (float)c, (float)d);
  if (accumulate) {
    const float4 p = *reinterpret_cast<const float4*>(out + i);
    o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
  }
  *reinterpret_cast<float4*>(out + i) = o;
}

// ------------------------------------------------------------------------------------------------------------------
// max |v| of a strided matrix into slot[0] (bits; atomicMax on non-negative floats = integer max: order-independent)
__device__ __forceinline__ void sp_absmax_body(const float* __restrict__ src, int R, int C4, int64_t ld,
                                               unsigned* __restrict__ slot, int slab) {
  // grid (column chunks of 256 float4, groups of 128 rows, 4 slabs of 32 rows): 16 B per lane along the row, four row loads
  // in flight; the maximum goes to a tensor-wide shard AND to the slab's 32-row block entry (EAV_SLOT_BMAX)
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * 128 + slab * 32, r1 = min(R, r0 + 32);
  if (r0 >= R || blockIdx.x * 256 >= C4) return;
  float m = 0.f;
  if (c < C4) {
    const float* p = src + 4 * (int64_t)c;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
      const float4 a = *reinterpret_cast<const float4*>(p + (int64_t)r * ld);
      const float4 b = *reinterpret_cast<const float4*>(p + (int64_t)(r + 1) * ld);
      const float4 cc = *reinterpret_cast<const float4*>(p + (int64_t)(r + 2) * ld);
      const float4 d = *reinterpret_cast<const float4*>(p + (int64_t)(r + 3) * ld);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
      m = fmaxf(m, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
      m = fmaxf(m, fmaxf(fmaxf(fabsf(cc.x), fabsf(cc.y)), fmaxf(fabsf(cc.z), fabsf(cc.w))));
      m = fmaxf(m, fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w))));
    }
    for (; r < r1; ++r) {
      const float4 a = *reinterpret_cast<const float4*>(p + (int64_t)r * ld);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m == m) {
    atomicMax(slot + EAV_SLOT_SHARD(((blockIdx.y * 4 + slab) * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)),
              __float_as_uint(m));
    eav_slot_blockmax(slot, r0, m);
  }
}

__global__ __launch_bounds__(256) void sp_absmax_kernel(const float* __restrict__ src, int R, int C4, int64_t ld,
                                                        unsigned* __restrict__ slot) {
  sp_absmax_body(src, R, C4, ld, slot, blockIdx.z);
}

// One launch for a table of dense matrices (the GEMM weights of an encoder: 49 matrices after every optimiser step).
struct PlaneJob {       // = EavPlaneJob of include/eav_hip.h
  const float* src;     // [R, C] dense
  unsigned char* dst;   // planes [R][Cp/8][2][8] or null
  unsigned char* dstT;  // planes of the transpose [C][Rp/8][2][8] or null
  float* slot;          // zeroed by the caller
  int R, C;
};

__global__ __launch_bounds__(256) void sp_absmax_multi_kernel(const PlaneJob* __restrict__ jobs) {
  const PlaneJob j = jobs[blockIdx.z >> 2];
  sp_absmax_body(j.src, j.R, j.C >> 2, j.C, reinterpret_cast<unsigned*>(j.slot), blockIdx.z & 3);
}

// sigma = 2^(14 - floor(log2 amax)): max|sigma v| in [2^14, 2^15); 1 for an all-zero / non-finite tensor
__device__ __forceinline__ float sigma_from_bits(unsigned bits) {
  const int e = (int)((bits >> 23) & 0xff);
  if (bits == 0u || e == 0xff) return 1.f;
  int se = 14 - (e - 127);
  se = max(-126, min(126, se));
  return __uint_as_float((unsigned)(se + 127) << 23);
}

__device__ __forceinline__ void split8(const float (&tv)[8], uint4& hi, uint4& lo, float lomul) {
  _Float16 h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = (_Float16)tv[e];
    l[e] = (_Float16)((tv[e] - (float)h[e]) * lomul);
  }
  hi = *reinterpret_cast<const uint4*>(h);
  lo = *reinterpret_cast<const uint4*>(l);
}

// src [R, C] fp32 (row stride ld) -> dst planes [R][Cp/8][2][8] (contraction over columns) and / or
// dstT planes [C][Rp/8][2][8] (contraction over rows).  64 x 64 tiles; persistent blocks walk the tiles (column index
// fastest) with the next tile's loads in flight while the current one is split, transposed through LDS and stored - a
// streaming pass keeps its rate with a few hundred resident blocks and drops with tens of thousands (tools/copy_bench.py).
__device__ __forceinline__ void sp_convert_body(const float* __restrict__ src, int R, int C, int64_t ld,
                                                float* __restrict__ slot, unsigned char* __restrict__ dst, int Cp,
                                                unsigned char* __restrict__ dstT, int Rp, float lomul,
                                                float* __restrict__ colsum_part, int ntx, int ntiles, int gelu, int bid,
                                                int nblocks) {
  // A thread's natural output - the hi and the lo piece of 8 values - is 32 contiguous bytes but a store carries 16: stored
  // directly, every store instruction writes 16-byte chunks with 16-byte holes (measured: 4.4 TB/s; with 32-byte holes
  // 2.5-3).  The pieces go through an LDS image of the tile's planes instead and leave in linear order: each store
  // instruction then writes whole 256-byte row segments.
  __shared__ float tile[64][65];
  __shared__ uint4 pimg[64 * 16 + 64];     // planes image: row r = 16 pieces (8 groups x hi, lo), one piece of padding per row
  __shared__ uint4 timg[64 * 16 + 64];     // the same for the transposed planes (row = tile column)
  const unsigned gbits = eav_slot_bits(slot);
  const float sigma0 = sigma_from_bits(gbits);
  if (bid == 0 && threadIdx.x == 0) {
    slot[EAV_SLOT_SIGMA] = sigma0;
    slot[EAV_SLOT_ISIGMA] = 1.f / sigma0;
  }
  const int t = threadIdx.x;
  const int cg = t & 7, rr = t >> 3;
  const bool vec = (ld & 3) == 0 && ((uintptr_t)src & 15) == 0;
  auto load_tile = [&](int id, float (&v)[2][8]) {
    const int by = id / ntx, bx = id - by * ntx;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int row = by * 64 + rr + 32 * pass, col = bx * 64 + 8 * cg;
      if (vec && row < R && col + 7 < C) {
        const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)row * ld + col);
        const float4 b = *reinterpret_cast<const float4*>(src + (int64_t)row * ld + col + 4);
        v[pass][0] = a.x; v[pass][1] = a.y; v[pass][2] = a.z; v[pass][3] = a.w;
        v[pass][4] = b.x; v[pass][5] = b.y; v[pass][6] = b.z; v[pass][7] = b.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[pass][e] = (row < R && col + e < C) ? src[(int64_t)row * ld + col + e] : 0.f;
      }
    }
  };
  // linear copy-out of an image: piece p = t + 256 k -> image row p >> 4, piece p & 15 (a wave = 4 rows x 256 bytes)
  auto copy_out = [&](const uint4* img, unsigned char* base, int64_t pitch, int row0, int nrows, int grp0, int ngrp) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = t + 256 * k, r = p >> 4, pc = p & 15;
      if (row0 + r < nrows && grp0 + (pc >> 1) < ngrp)
        *reinterpret_cast<uint4*>(base + (int64_t)(row0 + r) * pitch + (int64_t)grp0 * 32 + pc * 16) = img[r * 17 + pc];
    }
  };
  float cur[2][8];
  int id = bid;
  if (id < ntiles) load_tile(id, cur);
  for (; id < ntiles; id += nblocks) {
    float nxt[2][8];
    const int nid = id + nblocks;
    if (nid < ntiles) load_tile(nid, nxt);
    const int by = id / ntx, bx = id - by * ntx;
    const int r0 = by * 64, c0 = bx * 64;
    // Row blocks (32 rows: the two halves of this tile) whose maximum is >= 2^8 below the tensor's get their own power of two on
    // top of sigma (boost exponent kb, published in EAV_SLOT_BEXP for the consumers): rows keep fp32-grade RELATIVE precision
    // however small they are next to the largest row.  Not when transposed planes are written too (there the rows are contracted).
    float sig[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int kb = dstT ? 0 : min(eav_slot_boost(slot, 2 * by + pass, gbits), 253 - (int)(__float_as_uint(sigma0) >> 23));
      sig[pass] = __uint_as_float(__float_as_uint(sigma0) + ((unsigned)kb << 23));
      if (bx == 0 && t == 0 && !dstT)
        reinterpret_cast<int*>(slot)[EAV_SLOT_BEXP + ((2 * by + pass) & (EAV_SLOT_NBLK - 1))] = kb;
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const float sigma = sig[pass];
      float tv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) tv[e] = (gelu ? gelu_erf(cur[pass][e]) : cur[pass][e]) * sigma;
      if (dst) {
        uint4 hi, lo;
        split8(tv, hi, lo, lomul);
        pimg[(rr + 32 * pass) * 17 + 2 * cg] = hi;
        pimg[(rr + 32 * pass) * 17 + 2 * cg + 1] = lo;
      }
      if (dstT || colsum_part) {
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[rr + 32 * pass][8 * cg + e] = tv[e];
      }
    }
    __syncthreads();
    if (dst) copy_out(pimg, dst, (int64_t)Cp * 4, r0, R, c0 >> 3, Cp >> 3);
    if (colsum_part && t < 64 && c0 + t < C) {   // bias gradient partial: column sums of this 64-row tile (sigma is 2^e: exact)
      float hs[2];
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {       // each half of the tile carries its own scale
        float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
#pragma unroll
        for (int r = 32 * pass; r < 32 * pass + 32; r += 4) {
          a += tile[r][t]; b += tile[r + 1][t]; c += tile[r + 2][t]; d += tile[r + 3][t];
        }
        hs[pass] = ((a + b) + (c + d)) * (1.f / sig[pass]);
      }
      colsum_part[(int64_t)by * C + c0 + t] = hs[0] + hs[1];
    }
    if (dstT) {
      const int rg = t & 7, cc = t >> 3;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        float tv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) tv[e] = tile[8 * rg + e][cc + 32 * pass];
        uint4 hi, lo;
        split8(tv, hi, lo, lomul);
        timg[(cc + 32 * pass) * 17 + 2 * rg] = hi;
        timg[(cc + 32 * pass) * 17 + 2 * rg + 1] = lo;
      }
      __syncthreads();
      copy_out(timg, dstT, (int64_t)Rp * 4, c0, C, r0 >> 3, Rp >> 3);
    }
    __syncthreads();     // tile / pimg / timg are rewritten by the next iteration
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
      for (int e = 0; e < 8; ++e) cur[pass][e] = nxt[pass][e];
  }
}

__global__ __launch_bounds__(256) void sp_convert_kernel(const float* __restrict__ src, int R, int C, int64_t ld,
                                                         float* __restrict__ slot, unsigned char* __restrict__ dst,
                                                         int Cp, unsigned char* __restrict__ dstT, int Rp,
                                                         float lomul, float* __restrict__ colsum_part, int ntx,
                                                         int ntiles, int gelu) {
  sp_convert_body(src, R, C, ld, slot, dst, Cp, dstT, Rp, lomul, colsum_part, ntx, ntiles, gelu, blockIdx.x, gridDim.x);
}

// grid (blocks per matrix, matrices): every matrix of the table in one launch
__global__ __launch_bounds__(256) void sp_convert_multi_kernel(const PlaneJob* __restrict__ jobs, float lomul) {
  const PlaneJob j = jobs[blockIdx.y];
  const int Cp = (j.C + 31) / 32 * 32, Rp = (j.R + 31) / 32 * 32;
  const int ntx = (Cp + 63) / 64, nty = (Rp + 63) / 64;
  sp_convert_body(j.src, j.R, j.C, j.C, j.slot, j.dst, Cp, j.dstT, Rp, lomul, nullptr, ntx, ntx * nty, 0, blockIdx.x,
                  gridDim.x);
}

int g_convert_blocks = 512;   // resident-block cap of the conversion pass (tuning hook: eav_sp_set_convert_blocks)

void launch_convert(const float* src, int R, int C, int64_t ld, float* slot, void* dst, void* dstT, float* colsum_part,
                    hipStream_t st, int gelu = 0) {
  const int ntx = cdiv(eav_sp_kpad(C), 64), nty = cdiv(eav_sp_kpad(R), 64);
  const int ntiles = ntx * nty;
  hipLaunchKernelGGL(sp_convert_kernel, dim3(std::min(ntiles, g_convert_blocks)), dim3(256), 0, st, src, R, C, ld, slot,
                     (unsigned char*)dst, eav_sp_kpad(C), (unsigned char*)dstT, eav_sp_kpad(R),
                     g_loshift ? 2048.f : 1.f, colsum_part, ntx, ntiles, gelu);
}

}  // namespace

extern "C" int eav_sp_kpad(int K) { return (K + 31) / 32 * 32; }

extern "C" int eav_sp_absmax(const float* src, int R, int C, int64_t ld, float* slot, void* stream) {
  EAV_REQUIRE(src && slot && R > 0 && C > 0, "eav_sp_absmax: bad arguments");
  EAV_REQUIRE((C & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)src & 15) == 0,
              "eav_sp_absmax: columns / leading dimension must be multiples of 4, src 16-byte aligned");
  hipLaunchKernelGGL(sp_absmax_kernel, dim3(cdiv(C / 4, 256), cdiv(R, 128), 4), dim3(256), 0, (hipStream_t)stream, src, R,
                     C / 4, ld, reinterpret_cast<unsigned*>(slot));
  EAV_CHECK_LAUNCH("eav_sp_absmax");
  return EAV_OK;
}

extern "C" int eav_sp_convert(const float* src, int R, int C, int64_t ld, float* slot, void* dst, void* dstT,
                              void* stream) {
  EAV_REQUIRE(src && slot && R > 0 && C > 0 && (dst || dstT), "eav_sp_convert: bad arguments");
  EAV_REQUIRE((ld & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)dstT) & 15) == 0,
              "eav_sp_convert: leading dimension must be a multiple of 4, buffers 16-byte aligned");
  launch_convert(src, R, C, ld, slot, dst, dstT, nullptr, (hipStream_t)stream);
  EAV_CHECK_LAUNCH("eav_sp_convert");
  return EAV_OK;
}

// planes of GELU(src): src holds pre-activations (eav_gemm_sp with gelu = 3), slot the maximum of |GELU(src)|
extern "C" int eav_sp_convert_gelu(const float* src, int R, int C, int64_t ld, float* slot, void* dst, void* dstT,
                                   void* stream) {
  EAV_REQUIRE(src && slot && R > 0 && C > 0 && (dst || dstT), "eav_sp_convert_gelu: bad arguments");
  EAV_REQUIRE((ld & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)dstT) & 15) == 0,
              "eav_sp_convert_gelu: leading dimension must be a multiple of 4, buffers 16-byte aligned");
  launch_convert(src, R, C, ld, slot, dst, dstT, nullptr, (hipStream_t)stream, 1);
  EAV_CHECK_LAUNCH("eav_sp_convert_gelu");
  return EAV_OK;
}

// eav_sp_convert that also emits the bias-gradient partials of the same pass over src: colsum_part
// [eav_sp_convert_colsum_nparts(R)][C], row p = column sums of rows [64p, 64p+64) (finish with eav_reduce_partials).
extern "C" int eav_sp_convert_colsum_nparts(int R) { return cdiv(eav_sp_kpad(R), 64); }

extern "C" int eav_sp_convert_colsum(const float* src, int R, int C, int64_t ld, float* slot, void* dst, void* dstT,
                                     float* colsum_part, void* stream) {
  EAV_REQUIRE(src && slot && R > 0 && C > 0 && colsum_part, "eav_sp_convert_colsum: bad arguments");
  EAV_REQUIRE((ld & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)dstT) & 15) == 0,
              "eav_sp_convert_colsum: leading dimension must be a multiple of 4, buffers 16-byte aligned");
  launch_convert(src, R, C, ld, slot, dst, dstT, colsum_part, (hipStream_t)stream);
  EAV_CHECK_LAUNCH("eav_sp_convert_colsum");
  return EAV_OK;
}

// max|x| and planes (and / or planes of the transpose) of every matrix of a device-resident table in two launches: the
// weight refresh of an encoder after an optimiser step (49 matrices; one eav_sp_absmax + eav_sp_convert pair per matrix
// spent ~0.9 ms of launches per step on ~0.3 ms of memory traffic).  jobs: n x EavPlaneJob in device memory; every
// slot zeroed by the caller; C % 4 == 0, sources 16-byte aligned; maxR / maxC: the largest R and C of the table.
extern "C" int eav_sp_refresh_planes(const void* jobs, int n, int maxR, int maxC, void* stream) {
  EAV_REQUIRE(jobs && n > 0 && n <= 16383 && maxR > 0 && maxC > 0 && (maxC & 3) == 0, "eav_sp_refresh_planes: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sp_absmax_multi_kernel, dim3(cdiv(maxC / 4, 256), cdiv(maxR, 128), 4 * n), dim3(256), 0, st,
                     (const PlaneJob*)jobs);
  EAV_CHECK_LAUNCH("eav_sp_refresh_planes(absmax)");
  const int tiles = cdiv(eav_sp_kpad(maxC), 64) * cdiv(eav_sp_kpad(maxR), 64);
  hipLaunchKernelGGL(sp_convert_multi_kernel, dim3(std::min(tiles, std::max(8, 2048 / n)), n), dim3(256), 0, st,
                     (const PlaneJob*)jobs, g_loshift ? 2048.f : 1.f);
  EAV_CHECK_LAUNCH("eav_sp_refresh_planes(convert)");
  return EAV_OK;
}

extern "C" int eav_sp_set_convert_blocks(int n) { g_convert_blocks = n > 0 ? n : 1 << 30; return EAV_OK; }

extern "C" int eav_gemm_sp_set_splitk(int slices) {
  g_splitk_force = slices;
  return EAV_OK;
}

extern "C" int eav_gemm_sp_set_tile(int which) {
  g_force_tile = which & 3;
  g_small_tile = (which >> 6) & 3;      // +64: the 64 x 128 form wherever it applies, +128: never
  g_order = (which >> 4) & 3;      // +16: groups of 8 tile-rows, +32: row-major, 0: by shape
  g_loshift = (which & 4) ? 0 : 11;
  g_persist = (which & 8) ? 0 : 1;
  return EAV_OK;
}

// eav_gemm_sp that ALSO (or only: C = NULL) writes the stored value as row planes [M][Np/8][2][8] scaled by
// planes_slot[EAV_SLOT_SIGMA]: the consumer GEMM reads them directly, no conversion pass.  The scale must be known before
// the launch - a rigorous bound of |output| (eav_tf_forward_scales), not its measured maximum.  N % 8 == 0.
static int gemm_sp_impl(const void* A, const void* B, float* C, const float* slotA, const float* slotB, int M,
                        int N, int K, int ldc, int batch, int64_t sA_bytes, int64_t sC, float alpha,
                        const float* bias, int gelu, float* pre, const float* resid, int ldr, int accumulate,
                        float* amax_slot, void* planes_out, const float* planes_slot, void* stream, int flags,
                        float* colsum_part = nullptr) {
  const int terms = (flags & EAV_GEMM_ONE_TERM) ? 1 : 3;
  EAV_REQUIRE(A && B && (C || planes_out) && slotA && slotB && M > 0 && N > 0 && K > 0 && batch > 0,
              "eav_gemm_sp: bad arguments");
  EAV_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0 && (sA_bytes & 15) == 0,
              "eav_gemm_sp: operand planes must be 16-byte aligned");
  EAV_REQUIRE(!(resid && batch > 1), "eav_gemm_sp: residual epilogue is not batched");
  EAV_REQUIRE(gelu >= 0 && gelu <= 3 && (gelu != 2 || pre), "eav_gemm_sp: gelu = 2 (backward) reads the pre-activation from `pre`");
  EAV_REQUIRE(gelu != 3 || (!pre && !resid && !accumulate), "eav_gemm_sp: gelu = 3 stores the pre-activation only");
  EAV_REQUIRE(!planes_out || (planes_slot && batch == 1 && (N & 7) == 0 && gelu != 3 && ((uintptr_t)planes_out & 15) == 0),
              "eav_gemm_sp: plane output needs its scale slot, batch 1, N %% 8 == 0");
  EAV_REQUIRE(C || !accumulate, "eav_gemm_sp: accumulate needs C");
  EAV_REQUIRE(!colsum_part || (batch == 1 && gelu != 3), "eav_gemm_sp: column sums need batch 1 and a stored value");
  SpArgs g;
  const int Kp = eav_sp_kpad(K);
  g.A = (const unsigned char*)A; g.B = (const unsigned char*)B; g.C = C; g.slotA = slotA; g.slotB = slotB;
  g.bias = bias; g.resid = resid; g.pre = pre; g.amax = reinterpret_cast<unsigned*>(amax_slot);
  g.colsum = colsum_part;
  g.planes = (unsigned char*)planes_out; g.slotP = planes_slot; g.ldp = (int64_t)eav_sp_kpad(N) * 4;
  g.lomul = g_loshift ? 2048.f : 1.f;
  g.M = M; g.N = N; g.nkt = Kp / 32; g.ldA = (int64_t)Kp * 4; g.ldB = (int64_t)Kp * 4; g.ldc = ldc; g.ldr = ldr;
  g.sA = sA_bytes; g.sC = sC; g.alpha = alpha; g.gelu = gelu; g.accumulate = accumulate; g.kt_per_split = 0;
  // Tile order inside an XCD (tile_origin).  Few tile-columns (N <= 1024): row-major - the N / BN tiles that share a panel of A
  // are consecutive, i.e. resident together; with the 8-row groups a group (48 tiles at N = 768) straddled the rounds of 64
  // resident workgroups and every A panel came in from the fabric twice ([25216 x 3072] x [768 x 3072]^T: 8.8 M -> 6.1 M
  // 128-byte fabric reads per launch, TCC_EA0_RDREQ; 4.6 M in the 256 x 128 form; -2.5 % time, and the kernel runs at the
  // package power limit - tools/probes/clock_probe.py - where fabric traffic is paid for in clock).  Many tile-columns:
  // groups of 8 tile-rows x all columns (a round = 8 x 8 tiles) unless both operands are small enough to live in the
  // Infinity Cache, where the row-major walk measured 5-11 % faster (AST B=8: qkv, fc1).
  g.order = g_order ? g_order - 1
                    : (cdiv(N, 128) <= 8 || ((int64_t)M + N) * Kp * 4 <= (48ll << 20) ? 1 : 0);
  g.noblk = (flags & EAV_GEMM_NO_BLOCKMAX) ? 1 : 0;
  if (flags & EAV_GEMM_PLANES_NOLIFT) g.lomul = 1.f;      // lo = fp16(t - hi): the attention kernels' row planes
  EAV_REQUIRE(dispatch(g, batch, (hipStream_t)stream, terms, (flags & EAV_GEMM_SHARED_GPU) != 0),
              "eav_gemm_sp: no kernel for this tile form in the single-accumulator tuning mode (eav_gemm_sp_set_tile(4))");
  EAV_CHECK_LAUNCH("eav_gemm_sp");
  return EAV_OK;
}

extern "C" int eav_gemm_sp_planes(const void* A, const void* B, float* C, const float* slotA, const float* slotB, int M,
                                  int N, int K, int ldc, int batch, int64_t sA_bytes, int64_t sC, float alpha,
                                  const float* bias, int gelu, float* pre, const float* resid, int ldr, int accumulate,
                                  float* amax_slot, void* planes_out, const float* planes_slot, void* stream) {
  return gemm_sp_impl(A, B, C, slotA, slotB, M, N, K, ldc, batch, sA_bytes, sC, alpha, bias, gelu, pre, resid, ldr,
                      accumulate, amax_slot, planes_out, planes_slot, stream, 0);
}

// colsum_part (optional): [ceil(M / 64)][N] - row p = the column sums of the stored value over rows [64 p, 64 p + 64), the
// partials of a bias gradient (finish with eav_reduce_partials): with planes_out the producer of a gradient tensor
// leaves the planes AND the bias gradient behind, no conversion pass (fc2's data gradient -> the planes of dact).
// eav_gemm_sp_planes with options: EAV_GEMM_ONE_TERM (the hi.hi term alone, see eav_gemm_sp_x1), EAV_GEMM_SHARED_GPU (a
// second persistent GEMM runs beside this one: prefer the one-workgroup-per-CU form, see dispatch)
extern "C" int eav_gemm_sp_ex(const void* A, const void* B, float* C, const float* slotA, const float* slotB, int M, int N,
                              int K, int ldc, int batch, int64_t sA_bytes, int64_t sC, float alpha, const float* bias,
                              int gelu, float* pre, const float* resid, int ldr, int accumulate, float* amax_slot,
                              void* planes_out, const float* planes_slot, float* colsum_part, int flags,
                              void* stream) {
  return gemm_sp_impl(A, B, C, slotA, slotB, M, N, K, ldc, batch, sA_bytes, sC, alpha, bias, gelu, pre, resid, ldr,
                      accumulate, amax_slot, planes_out, planes_slot, stream, flags, colsum_part);
}

// eav_gemm_sp with the hi.hi term only: the product of the operands rounded to fp16 (11-bit mantissas under the planes'
// scales, fp32 accumulation) at a third of the matrix work.  Same planes, same epilogues.
extern "C" int eav_gemm_sp_x1(const void* A, const void* B, float* C, const float* slotA, const float* slotB, int M, int N,
                              int K, int ldc, int batch, int64_t sA_bytes, int64_t sC, float alpha, const float* bias,
                              int gelu, float* pre, const float* resid, int ldr, int accumulate, float* amax_slot,
                              void* stream) {
  EAV_REQUIRE(C, "eav_gemm_sp_x1: bad arguments");
  return gemm_sp_impl(A, B, C, slotA, slotB, M, N, K, ldc, batch, sA_bytes, sC, alpha, bias, gelu, pre, resid, ldr,
                      accumulate, amax_slot, nullptr, nullptr, stream, EAV_GEMM_ONE_TERM);
}

extern "C" int eav_gemm_sp(const void* A, const void* B, float* C, const float* slotA, const float* slotB, int M, int N,
                           int K, int ldc, int batch, int64_t sA_bytes, int64_t sC, float alpha, const float* bias,
                           int gelu, float* pre, const float* resid, int ldr, int accumulate, float* amax_slot,
                           void* stream) {
  EAV_REQUIRE(C, "eav_gemm_sp: bad arguments");
  return eav_gemm_sp_planes(A, B, C, slotA, slotB, M, N, K, ldc, batch, sA_bytes, sC, alpha, bias, gelu, pre, resid, ldr,
                            accumulate, amax_slot, nullptr, nullptr, stream);
}

// split-K plan for the weight-gradient shapes (small M x N output, long contraction).  All workgroups of a launch have
// the same length L = K-tiles per slice (+ ~15 K-tiles' worth of start-up and epilogue); a CU works through
// k = ceil(workgroups / 256) of them, two at a time at ~1.25 us per K-tile each, an odd last one alone at ~0.7 us; then the
// slabs are reduced (64 KB per tile and slice at ~3.5 TB/s).  The slice count minimises that estimate - fitted to
// tools/splitk_sweep.py on the AST / ViT shapes, where it lands within 3 % of the best measured count (ViT fc1, 144 tiles:
// 7 slices = 1008 workgroups = 4 per CU, 0.34 ms; the previous "~512 workgroups" rule took 4 slices = 3 per CU of almost
// twice the length, 0.41-0.48 ms).  At least 8 K-tiles per slice, at most 32 slices.
extern "C" int eav_gemm_sp_splitk_plan(int M, int N, int K) {
  const int tiles = cdiv(M, 128) * cdiv(N, 128), nkt = eav_sp_kpad(K) / 32;
  const int maxs = std::min(32, std::max(1, nkt / 8));
  int best = 1;
  double best_cost = 1e30;
  for (int ns = 1; ns <= maxs; ++ns) {
    const int k = cdiv(tiles * ns, 256);
    const double cost = (cdiv(nkt, ns) + 15) * (1.25 * (k / 2) + 0.7 * (k & 1)) + (ns > 1 ? 0.0187 * ns * tiles : 0.0);
    if (cost < best_cost) { best_cost = cost; best = ns; }
  }
  return best;
}

// Weight-gradient product C[M,N] = sum_t A[t,m] B[t,n] over ROW planes (contraction over the rows = tokens): A planes
// [Tp][Mp/8][2][8], B planes [Tp][Np/8][2][8] with Tp = T rounded up to 32 and the rows >= T ZERO (the conversion never
// writes them; allocate zero-filled).  Split-K over the token tiles with a fixed-order fp64 reduction of the slabs.
static int gemm_sp_splitk_impl(const void* A, const void* B, float* C, float* ws, const float* slotA,
                               const float* slotB, int M, int N, int T, int accumulate, void* stream, int terms) {
  EAV_REQUIRE(A && B && C && ws && slotA && slotB && M > 0 && N > 0 && T > 0, "eav_gemm_sp_splitk: bad arguments");
  EAV_REQUIRE((N & 3) == 0 && (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)ws) & 15) == 0,
              "eav_gemm_sp_splitk: N must be a multiple of 4, buffers 16-byte aligned");
  EAV_REQUIRE(g_loshift != 0, "eav_gemm_sp_splitk: the single-accumulator tuning mode has no token-contracting kernel");
  const int nsplit = g_splitk_force > 0 ? g_splitk_force : eav_gemm_sp_splitk_plan(M, N, T);
  SpArgs g;
  // a slice's tiles are spread over the 8 XCDs (tiles / 8 each): walk them so that an XCD's share is the squarer block
  g.order = g_order ? g_order - 1 : (cdiv(N, 128) <= 8 && cdiv(M, 128) > cdiv(N, 128) ? 1 : 0);
  g.A = (const unsigned char*)A; g.B = (const unsigned char*)B; g.slotA = slotA; g.slotB = slotB;
  g.bias = nullptr; g.resid = nullptr; g.pre = nullptr; g.amax = nullptr; g.planes = nullptr; g.slotP = nullptr;
  g.colsum = nullptr; g.noblk = 0;
  g.ldp = 0; g.lomul = 2048.f;
  g.M = M; g.N = N; g.nkt = cdiv(T, 32); g.ldA = (int64_t)eav_sp_kpad(M) * 4; g.ldB = (int64_t)eav_sp_kpad(N) * 4;
  g.ldc = N; g.ldr = 0; g.sA = 0; g.sC = 0; g.alpha = 1.f; g.gelu = 0;
  hipStream_t st = (hipStream_t)stream;
  if (nsplit <= 1) {
    g.C = C; g.accumulate = accumulate; g.kt_per_split = 0;
    EAV_REQUIRE(dispatch_tr(g, 1, st, terms), "eav_gemm_sp_splitk: no kernel for this mode");
    EAV_CHECK_LAUNCH("eav_gemm_sp_splitk");
    return EAV_OK;
  }
  g.C = ws; g.accumulate = 0;
  g.kt_per_split = cdiv(g.nkt, nsplit);
  const int nz = cdiv(g.nkt, g.kt_per_split);
  EAV_REQUIRE(dispatch_tr(g, nz, st, terms), "eav_gemm_sp_splitk: no kernel for this mode");
  EAV_CHECK_LAUNCH("eav_gemm_sp_splitk");
  const int64_t n = (int64_t)M * N;
  hipLaunchKernelGGL(sp_splitk_reduce_kernel, dim3((unsigned)cdiv64(n, 1024)), dim3(256), 0, st, ws, nz, n, C,
                     accumulate);
  EAV_CHECK_LAUNCH("eav_gemm_sp_splitk(reduce)");
  return EAV_OK;
}

extern "C" int eav_gemm_sp_splitk(const void* A, const void* B, float* C, float* ws, const float* slotA,
                                  const float* slotB, int M, int N, int T, int accumulate, void* stream) {
  return gemm_sp_splitk_impl(A, B, C, ws, slotA, slotB, M, N, T, accumulate, stream, 3);
}

// eav_gemm_sp_splitk on two terms, hi_A.hi_B + lo_A.hi_B: B (the activation operand of a weight gradient) rounded to fp16, A
// (the gradient operand) at full split precision - two thirds of the matrix work of eav_gemm_sp_splitk
extern "C" int eav_gemm_sp_splitk_x2(const void* A, const void* B, float* C, float* ws, const float* slotA,
                                     const float* slotB, int M, int N, int T, int accumulate, void* stream) {
  return gemm_sp_splitk_impl(A, B, C, ws, slotA, slotB, M, N, T, accumulate, stream, 2);
}

// eav_gemm_sp_splitk with the hi.hi term only (see eav_gemm_sp_x1)
extern "C" int eav_gemm_sp_splitk_x1(const void* A, const void* B, float* C, float* ws, const float* slotA,
                                     const float* slotB, int M, int N, int T, int accumulate, void* stream) {
  return gemm_sp_splitk_impl(A, B, C, ws, slotA, slotB, M, N, T, accumulate, stream, 1);
}
