// Shared helpers for the gfx950 kernels of libeav_hip.so (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define EAV_OK 0
#define EAV_EINVAL (-1)
#define EAV_ELAUNCH (-2)
#define EAV_EUNSUPPORTED (-3)

int eav_set_error(int code, const char* fmt, ...);

#define EAV_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) return eav_set_error(EAV_EINVAL, __VA_ARGS__);  \
  } while (0)

#define EAV_CHECK_LAUNCH(name)                                                        \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess)                                                            \
      return eav_set_error(EAV_ELAUNCH, "%s: %s", name, hipGetErrorString(e__));      \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ELU(alpha=1) as torch computes it (expm1 for v <= 0) and its derivative (EEGNet_tor.py:53,56,61: nn.ELU()).
#ifdef EAV_ABL_ELU      // timing-only ablation: the passes without their expm1 (results garbage)
__device__ __forceinline__ float elu_f(float v) { return v > 0.f ? v : 0.5f * v; }
#elif defined(EAV_ELU_LIBM)
// round 5: libm expm1f of the clamped argument, selected (22 instructions per call + clamp + select; `v > 0 ? v :
// expm1f(v)` compiles to an exec-masked branch around every call - the lanes of a wave diverge on the sign, so nothing is
// skipped, and the four calls per channel of the depthwise passes sit in four basic blocks that cannot be interleaved)
__device__ __forceinline__ float elu_f(float v) {
  const float e = expm1f(fminf(v, 0.f));
  return v <= 0.f ? e : v;
}
#else
// expm1 on [-17.5, 0] (below, expm1 rounds to -1 in fp32) in 13 branch-free VALU instructions:
//   t = x log2(e) + 1.5 2^23        the sum's low mantissa bits hold n = rint(x log2 e), no conversion instruction
//   r = x - n ln2_hi - n ln2_lo     |r| <= ln2 / 2, two fmas (ln2_hi has 15 mantissa bits: n ln2_hi is exact)
//   p = r + r^2 q(r)                q = degree-4 fit of (expm1(r) - r) / r^2 (tools/probes/expm1_fit.py)
//   s = 2^n                         (bits(t) << 23) + bits(1.0f): one v_lshl_add_u32
//   expm1(x) = fma(p, s, s - 1)     s - 1 is exact; n = 0 returns p itself
// Max error 0.97 ulp against float64 expm1 over 8 M arguments (the probe emulates this sequence in fp32 arithmetic) - the
// accuracy class of libm's expm1f, which torch's CPU ELU calls.  elu_f: NaN in, NaN out; -inf gives -1.
__device__ __forceinline__ float expm1_neg(float x) {       // x in [-17.5, 0]
#pragma clang fp contract(off)
  const float t = __builtin_fmaf(x, 1.4426950408889634f, 12582912.f);
  const float n = t - 12582912.f;
  float r = __builtin_fmaf(n, -0.693145751953125f, x);
  r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
  float q = 1.394644962e-03f;
  q = __builtin_fmaf(q, r, 8.366583847e-03f);
  q = __builtin_fmaf(q, r, 4.166628048e-02f);
  q = __builtin_fmaf(q, r, 1.666654348e-01f);
  q = __builtin_fmaf(q, r, 0.5f);
  const float p = __builtin_fmaf(r * r, q, r);
  const float s = __uint_as_float((__float_as_uint(t) << 23) + 0x3f800000u);
  return __builtin_fmaf(p, s, s - 1.f);
}
__device__ __forceinline__ float elu_f(float v) {
  // v_med3_f32 clamps in one instruction (and maps a NaN to a bound: the select below hands the NaN itself through,
  // as torch's F.elu does - `v <= 0` is false for it)
  const float e = expm1_neg(__builtin_amdgcn_fmed3f(v, -17.5f, 0.f));
  return v <= 0.f ? e : v;
}
#endif
__device__ __forceinline__ float elu_grad_from_out(float v, float a) { return v > 0.f ? 1.f : a + 1.f; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 32 lanes of each wave half (lanes 0-31 / 32-63 reduce separately)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum of NV values per thread (256 threads = 4 waves).  red: LDS, >= 4*NV floats.
// Result valid in thread 0..NV-1 order: returns total of value k in out[k] for threadIdx.x == 0 only
template <int NV>
__device__ __forceinline__ void block_sum_256(float (&v)[NV], float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float s = wave_sum(v[k]);
    if (lane == 0) red[wave * NV + k] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    v[0] = red[threadIdx.x] + red[NV + threadIdx.x] + red[2 * NV + threadIdx.x] + red[3 * NV + threadIdx.x];
  }
  __syncthreads();
}

// Operand-scale slot of the split-operand kernels (EAV_SP_SLOT floats, include/eav_hip.h): 64 shards of the bits of
// max|x|, one per 128-byte line (word 32*i) so that the producers' atomicMax traffic spreads over 64 L2 lines / channels
// instead of serialising on two; sigma at word EAV_SLOT_SIGMA, 1/sigma at EAV_SLOT_ISIGMA.
#define EAV_SLOT_SHARD(i) (32 * ((i) & 63))
#define EAV_SLOT_SIGMA 2048
#define EAV_SLOT_ISIGMA 2049
// Per-row-block refinement of the tensor-wide scale (rows = output rows of the products that read the planes as their A
// operand, e.g. the tokens of a gradient tensor): word EAV_SLOT_BMAX + (b & 1023) holds the bits of max|x| over the rows
// [32 b, 32 b + 32) (EAV_BLK_ROWS = one MFMA tile of rows / one token tile of the weight-gradient products; producers
// atomicMax it next to the tensor-wide shards; blocks beyond 1024 - tensors of more than 32768 rows - alias, which only makes
// the entry an over-estimate shared by the aliasing blocks), word EAV_SLOT_BEXP + (b & 1023) the boost exponent k_b >= 0 the conversion chose: the
// planes of block b hold sigma 2^k_b x.  k_b = 0 unless the block's maximum is >= 2^8 below the tensor's - then the block
// gets its own power of two, so a row keeps fp32-grade relative precision however small it is next to the largest row.
#define EAV_SLOT_BMAX 2080
#define EAV_SLOT_BEXP 3104
#define EAV_SLOT_NBLK 1024
#define EAV_BLK_SHIFT 5
#define EAV_BLK_ROWS (1 << EAV_BLK_SHIFT)
#define EAV_BOOST_MIN 8
__device__ __forceinline__ void eav_slot_blockmax(unsigned* slot, int row, float vmax) {
  if (vmax == vmax) atomicMax(slot + EAV_SLOT_BMAX + ((row >> EAV_BLK_SHIFT) & (EAV_SLOT_NBLK - 1)), __float_as_uint(vmax));
}
// boost exponent of row block b given the tensor-wide maximum bits (0: unknown / not small enough)
__device__ __forceinline__ int eav_slot_boost(const float* slot, int b, unsigned gbits) {
  const unsigned bb = __float_as_uint(slot[EAV_SLOT_BMAX + (b & (EAV_SLOT_NBLK - 1))]);
  if (bb == 0u || gbits == 0u) return 0;
  const int k = (int)((gbits >> 23) & 0xff) - (int)((bb >> 23) & 0xff);
  return k >= EAV_BOOST_MIN ? (k > 60 ? 60 : k) : 0;
}
__device__ __forceinline__ unsigned eav_slot_bits(const float* slot) {
  unsigned bits = 0u;
#pragma unroll 8
  for (int i = 0; i < 64; ++i) bits = max(bits, __float_as_uint(slot[32 * i]));
  return bits;
}

// counter-based dropout keep decision: pure function of (seed, element index)
__device__ __forceinline__ uint32_t eav_hash32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 40);  // 24 random bits
}
// returns the multiplier applied to the pooled value: 0 or 1/(1-p); 1 when dropout is off
// seed_dev (optional): device-resident step counter folded into the seed, so that a captured hipGraph
// draws a fresh mask on every replay without any host-side argument changing.
__device__ __forceinline__ uint64_t dropout_seed(uint64_t seed, const uint64_t* seed_dev) {
  return seed_dev ? seed + 2ull * (*seed_dev) : seed;
}
__device__ __forceinline__ float dropout_mult(float drop_p, uint64_t seed, const uint8_t* mask, uint64_t idx) {
  if (drop_p <= 0.f) return 1.f;
  bool keep = mask ? (mask[idx] != 0) : ((float)eav_hash32(seed, idx) * (1.0f / 16777216.0f) >= drop_p);
  return keep ? 1.f / (1.f - drop_p) : 0.f;
}
// drop_p < 0 selects nn.Dropout2d semantics with probability -drop_p: ONE draw per (sample, channel) row (the keep
// decision hashes the row index instead of the element index; an explicit mask stays per element)
__device__ __forceinline__ float dropout_mult_row(float drop_p, uint64_t seed, const uint8_t* mask, uint64_t idx,
                                                  uint64_t row) {
  if (drop_p >= 0.f) return dropout_mult(drop_p, seed, mask, idx);
  const float p = -drop_p;
  const bool keep = mask ? (mask[idx] != 0) : ((float)eav_hash32(seed, row) * (1.0f / 16777216.0f) >= p);
  return keep ? 1.f / (1.f - p) : 0.f;
}
