// EEGNet temporal FIR (firstConv, up to 321 taps, 1 -> 8 filters) and its weight gradient by FFT, exact fp32 arithmetic.
//
// Reference op: nn.Conv2d(1, F1=8, (1, kernLength=300), padding='same', bias=False) (CNN_torch/EEGNet_tor.py:24,51) and
// the weight gradient autograd derives for it (:109).  The direct form costs 2 x 300 flops per output and filter - 92 GFLOP
// per pass at [64,1,30,10000], 0.59 ms at the fp32 matrix peak (eegnet_fir.hip reaches 0.73 / 0.83 ms).  A convolution
// with a 300-tap filter is the same linear map evaluated by overlap-save blocks of 1024-point FFTs in ~350 flops per
// output sample for all 8 filters together: 13 x fewer flops, so the two kernels here are bound by the HBM traffic of
// y1 / g1 (614 MB each) instead of by arithmetic.
//
// Forward (per block of LB = 704 outputs of a pair of electrodes c, c+1):
//   s[n] = x[c][t0 - padl + n] + i x[c+1][t0 - padl + n], n = 0 .. 1023 (zero outside the recording)
//   y_f[j] = sum_k w_f[k] s[j + k]  (j < 704, k < klen: j + k <= 1023, no wrap)  = IFFT(FFT(s) . conj(FFT(w_f)))[j]
//   Re y_f = the output of electrode c, Im y_f = that of electrode c+1 (a real filter maps real to real, imaginary to
//   imaginary): one complex transform serves two electrodes with no unpacking.
// Weight gradient (dy formed from g1, y1 and the BatchNorm-backward coefficients while loading, as in eegnet_fir.hip):
//   dW_f[k] = sum_{b,c,t} dy[b,f,c,t] x[b,c,t + k - padl] = Re IFFT( sum_{b, pairs, blocks} conj(FFT(D_f)) . FFT(s) )[k]
//   with D_f = dy_c + i dy_{c+1} of a block (704 samples, zero-padded): Re(conj(D) s') = d_c s'_c + d_{c+1} s'_{c+1}.
//   The spectra are accumulated per workgroup in registers (wave f owns filter f), written once, summed over the
//   workgroups in a fixed order and transformed back by a finishing kernel - bit-reproducible run to run.
//
// The 1024-point complex FFT runs inside ONE wave, 16 points per lane (element j of lane l = index l + 64 j on input and
// output), as radix 16 x 16 x 4 with two exchanges through a wave-private LDS buffer:
//   n = 64 n1 + 4 n2 + n3,  k = k1 + 16 k2 + 256 k3
//   (1) radix-16 DFT over n1 in registers (lane = 4 n2 + n3), twiddle W^(lane k1)
//   (2) exchange: slot 68 k1 + lane  ->  lane' = 4 k1 + n3 reads slots 68 k1 + 4 n2 + n3
//   (3) radix-16 DFT over n2, twiddle W64^(n3 k2)
//   (4) exchange: slot k1 + 16 k2 + 260 n3  ->  lane'' reads slots lane'' + 64 m + 260 n3 (m < 4)
//   (5) radix-4 DFT over n3: X[lane'' + 64 (m + 4 k3)]
// Every ds_write_b64 / ds_read_b64 of both exchanges is bank-conflict free (pitches 68 and 260 slots - 264 has a 2-way conflict in the 16-lane groups of ds_write_b64; checked offline
// against the lane groups of MI355X_MICROARCH.md section LDS).  No workgroup barrier inside the transform.
#include <algorithm>

#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

constexpr int F1 = 8;
constexpr int NF = 1024;        // transform length
constexpr int LB = 704;         // outputs per block: 11 rows of 64 (2816 B = 22 cache lines, so every block is line aligned)
constexpr int NROW = LB / 64;   // 11
#ifdef FFTV_LDSX2
constexpr int WBUF = 1088;      // float2 slots of a wave's exchange buffer: max(68 x 16, 260 x 3 + 256)
#else
constexpr int XP = 66;          // pitch of the exchange rows: reads 66 k1 + n3 (+ 4 n2) cover 32 distinct 8-byte banks
constexpr int WBUF = 1056;      // float2 slots of a wave's exchange buffer: 66 x 15 + 64, rounded to 128 bytes
#endif
constexpr int MAXK = NF - LB + 1;   // 321 taps

typedef float v2f __attribute__((ext_vector_type(2)));

// y1 / g1 are streamed exactly once by these kernels (614 MB each): non-temporal accesses keep them from evicting the
// input segments that ARE re-used out of the L2 (-DFFTV_NO_NT: plain accesses, measured 2-4 % slower).
// FFTV_ABL_NOLOAD / FFTV_ABL_NOFFT: timing-only ablations (results are garbage): the arithmetic side alone / the memory side
// alone - forward 0.26 / 0.15 ms of 0.28, weight gradient 0.31 / 0.25 ms of 0.39: the transforms, not HBM, are the longer leg.
#if defined(FFTV_ABL_NOLOAD)
#define EAV_LDG(p) (1.0f)
#define EAV_STG(p, v) asm volatile("" ::"v"(v))
#elif !defined(FFTV_NO_NT)
#define EAV_LDG(p) __builtin_nontemporal_load(p)
#define EAV_STG(p, v) __builtin_nontemporal_store((v), (p))
#else
#define EAV_LDG(p) (*(p))
#define EAV_STG(p, v) (*(p) = (v))
#endif

__device__ __forceinline__ v2f swp(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }      // (y, x): an op_sel, no move
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// Complex products in forms hipcc maps to v_pk_mul_f32 / v_pk_fma_f32 with op_sel operands only - written as
// a.xx * b + a.yy * (-b.y, b.x) it builds the swapped / negated pair with a v_xor + v_mov per product (124 of the 454 VALU
// instructions of one inverse transform + filter product).
#ifdef FFTV_OLD_ARITH
__device__ __forceinline__ v2f cmul(v2f a, v2f b) { return a.xx * b + a.yy * (v2f){-b.y, b.x}; }
__device__ __forceinline__ v2f cmulc(v2f a, v2f b) { return a.xx * (v2f){b.x, -b.y} + a.yy * (v2f){b.y, b.x}; }
#else
__device__ __forceinline__ v2f cmul(v2f a, v2f b) {       // a b
  return fma2(a.yy * (v2f){-1.f, 1.f}, swp(b), a.xx * b);
}
__device__ __forceinline__ v2f cmulc(v2f a, v2f b) {      // a conj(b)
  return fma2(a.yy, swp(b), (a.xx * (v2f){1.f, -1.f}) * b);
}
#endif
#ifndef FFTV_NO_ASM_CMUL
// Two complex products a0 b0, a1 b1 (CONJ: a conj(b)) in FOUR packed instructions: what hipcc cannot select is one
// v_pk_fma_f32 with a swapped AND half-negated operand (it builds the pair with a third instruction).  The two products are
// interleaved so that no packed result is consumed by the next instruction (gfx950 needs one wait state there); the
// leading s_nop covers a packed producer of an input directly in front of the block, the trailing `s_nop 1` a consumer
// hipcc schedules directly behind it (the hazard recogniser does not see into inline asm: a v_permlane*_swap of the last
// result needs two wait states after the VALU write, a packed consumer one).
#ifdef EAV_FIR_NO_TRAILING_NOP      // (A/B only: round 5's blocks, correct by scheduling luck)
#define EAV_ASM_TAIL "s_nop 0"
#else
#define EAV_ASM_TAIL "s_nop 1"
#endif
template <bool CONJ>
__device__ __forceinline__ void cmul2(v2f a0, v2f b0, v2f a1, v2f b1, v2f& r0, v2f& r1) {
  if (CONJ)
    asm("s_nop 0\n\t"
        "v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %4, %5 op_sel_hi:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %1, %4, %5, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
        EAV_ASM_TAIL
        : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
  else
    asm("s_nop 0\n\t"
        "v_pk_mul_f32 %0, %2, %3 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %4, %5 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %4, %5, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        EAV_ASM_TAIL
        : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}
// acc0 += a0 conj(b0), acc1 += a1 conj(b1): four packed fmas, the accumulators interleaved for the same reason
__device__ __forceinline__ void cmacc2_conj(v2f a0, v2f b0, v2f a1, v2f b1, v2f& acc0, v2f& acc1) {
  asm("s_nop 0\n\t"
      "v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
      "v_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1] neg_hi:[0,1,0]\n\t"
      "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %1, %4, %5, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n\t"
      EAV_ASM_TAIL
      : "+v"(acc0), "+v"(acc1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}
#endif

// a w (forward) / a conj(w) (inverse) for a twiddle w: the w-only factors are loop invariants the compiler keeps in
// registers, which leaves two instructions per product
template <bool INV>
__device__ __forceinline__ v2f twmul(v2f a, v2f w) {
#ifdef FFTV_OLD_ARITH
  return INV ? cmulc(a, w) : cmul(a, w);
#else
  return fma2(w.yy * (INV ? (v2f){1.f, -1.f} : (v2f){-1.f, 1.f}), swp(a), w.xx * a);
#endif
}
// multiply by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ v2f rot(v2f a) {
#ifdef FFTV_OLD_ARITH
  return INV ? (v2f){-a.y, a.x} : (v2f){a.y, -a.x};
#else
  return swp(a) * (INV ? (v2f){-1.f, 1.f} : (v2f){1.f, -1.f});
#endif
}

template <bool INV>
__device__ __forceinline__ void dft4(v2f& a0, v2f& a1, v2f& a2, v2f& a3) {
#ifdef FFTV_OLD_ARITH
  const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = rot<INV>(a1 - a3);
  a0 = t0 + t2;
  a1 = t1 + t3;
  a2 = t0 - t2;
  a3 = t1 - t3;
#else
  // t1 +- rot(d) = fma(swap(d), (1, -1), t1): the rotation rides on the add (exact: the factors are +-1)
  const v2f t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = swp(a1 - a3);
  const v2f c = INV ? (v2f){-1.f, 1.f} : (v2f){1.f, -1.f};
  a0 = t0 + t2;
  a1 = fma2(d, c, t1);
  a2 = t0 - t2;
  a3 = fma2(d, -c, t1);
#endif
}

// 16-point DFT in registers: n = 4 a + b, k = c + 4 d; X[c + 4 d] = sum_b W16^(b c) W4^(b d) sum_a v[4 a + b] W4^(a c)
template <bool INV>
__device__ __forceinline__ void dft16(v2f (&v)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
  for (int b = 0; b < 4; ++b) dft4<INV>(v[b], v[4 + b], v[8 + b], v[12 + b]);      // v[4 c + b] = y[b][c]
  // y[b][c] *= W16^(b c), W16^m = (cos, -sin)(2 pi m / 16)
  v[4 * 1 + 1] = twmul<INV>(v[4 * 1 + 1], (v2f){C1, -S1});      // b c = 1
  v[4 * 2 + 1] = twmul<INV>(v[4 * 2 + 1], (v2f){R2, -R2});      // 2
  v[4 * 3 + 1] = twmul<INV>(v[4 * 3 + 1], (v2f){S1, -C1});      // 3
  v[4 * 1 + 2] = twmul<INV>(v[4 * 1 + 2], (v2f){R2, -R2});      // 2
  v[4 * 2 + 2] = rot<INV>(v[4 * 2 + 2]);                        // 4: -i
  v[4 * 3 + 2] = twmul<INV>(v[4 * 3 + 2], (v2f){-R2, -R2});     // 6
  v[4 * 1 + 3] = twmul<INV>(v[4 * 1 + 3], (v2f){S1, -C1});      // 3
  v[4 * 2 + 3] = twmul<INV>(v[4 * 2 + 3], (v2f){-R2, -R2});     // 6
  v[4 * 3 + 3] = twmul<INV>(v[4 * 3 + 3], (v2f){-C1, S1});      // 9
  // for each c: DFT over b of y[b][c] = v[4 c + b] -> X[c + 4 d]
  v2f o[16];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    v2f y0 = v[4 * c], y1 = v[4 * c + 1], y2 = v[4 * c + 2], y3 = v[4 * c + 3];
    dft4<INV>(y0, y1, y2, y3);
    o[c] = y0; o[c + 4] = y1; o[c + 8] = y2; o[c + 12] = y3;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = o[k];
}

// after dft4 over the a index the results sit at v[4 c + b]: the first loop above writes y[b][c] into (v[b], v[4 + b],
// v[8 + b], v[12 + b]) = positions 4 c + b.  (kept as a comment: the index bookkeeping is the easy thing to get wrong)

// Twiddle tables of a workgroup in LDS (16 KB): t1[k][lane] = W1024^(lane k), t2[k][lane] = W64^((lane & 3) k) - one
// conflict-free ds_read_b64 per use instead of 64 resident VGPRs (with them the forward kernel spilled 68 registers).
#ifdef FFTV_LDSX2
constexpr int TWSZ = 2 * 16 * 64;
#else
constexpr int TWSZ = 16 * 64 + 16 * 4;
#endif

__device__ __forceinline__ void make_twiddles(v2f* __restrict__ twl) {      // 512 threads: one entry each, twice
  for (int i = threadIdx.x; i < TWSZ; i += blockDim.x) {
    float s, c;
#ifdef FFTV_LDSX2
    const int which = i >> 10, k = (i >> 6) & 15, lane = i & 63;
    if (which == 0) sincospif((float)(lane * k) * (1.0f / 512.0f), &s, &c);      // 2 pi m / 1024 = pi (m / 512), m exact
    else sincospif((float)((lane & 3) * k) * (1.0f / 32.0f), &s, &c);            // 2 pi m / 64
#else
    if (i < 1024) sincospif((float)((i & 63) * (i >> 6)) * (1.0f / 512.0f), &s, &c);      // t1[k][lane] = W1024^(lane k)
    else sincospif((float)(((i - 1024) & 3) * ((i - 1024) >> 2)) * (1.0f / 32.0f), &s, &c);      // t2[k][n3] = W64^(n3 k)
#endif
    twl[i] = (v2f){c, -s};
  }
  __syncthreads();
}

// LDS reads as single ds_read_b64 (2 LDS cycles per wave-instruction): left alone, hipcc pairs them into ds_read2_b64 /
// ds_read2st64_b64, which the LDS serves as two 4 x 16-lane accesses = 8 cycles for the same 16 bytes per lane
// (MI355X_MICROARCH.md, LDS table).  A volatile access is the one form its load-store optimiser leaves alone.
#ifdef FFTV_READ2
__device__ __forceinline__ v2f lds_ld(const v2f* p) { return *p; }
#else
typedef const volatile __attribute__((address_space(3))) v2f* lds_cvp;
__device__ __forceinline__ v2f lds_ld(const v2f* p) { return *(lds_cvp)p; }
#endif

// The exchanges need no hardware wait between a wave's writes and its own reads: the LDS executes one wave's DS
// instructions in issue order, and the transform is private to the wave.  What is needed is that the COMPILER keeps the
// order (a "memory" clobber); it then places counted lgkmcnt waits in front of the first use of each read by itself, so the
// butterflies start on the first rows while the last ones are still in flight (-DFFTV_HWWAIT: the round-4 form, a full
// lgkmcnt(0) drain after the writes and again after the reads).
#ifdef FFTV_HWWAIT
#define FFT_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define FFT_ORDER() asm volatile("" ::: "memory")
#endif

// In-wave 1024-point FFT: v[j] = x[lane + 64 j] -> v[j] = X[lane + 64 j].  INV: conjugate twiddles, no 1/N.
#ifdef FFTV_LDSX2
template <bool INV>
__device__ __forceinline__ void fft1024(v2f (&v)[16], v2f* __restrict__ xb, int lane, const v2f* __restrict__ twl) {
#ifdef FFTV_ABL_NOFFT
  return;
#endif
  const v2f* t1 = twl + lane;
  const v2f* t2 = twl + 1024 + lane;
  dft16<INV>(v);
#pragma unroll
  for (int k = 1; k < 16; ++k) v[k] = twmul<INV>(v[k], lds_ld(t1 + 64 * k));
#pragma unroll
  for (int k = 0; k < 16; ++k) xb[68 * k + lane] = v[k];
  FFT_ORDER();
  {
    const v2f* rp = xb + 68 * (lane >> 2) + (lane & 3);
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2] = lds_ld(rp + 4 * n2);
  }
  FFT_ORDER();
  dft16<INV>(v);
#pragma unroll
  for (int k = 1; k < 16; ++k) v[k] = twmul<INV>(v[k], lds_ld(t2 + 64 * k));
  {
    v2f* wp = xb + (lane >> 2) + 260 * (lane & 3);
#pragma unroll
    for (int k = 0; k < 16; ++k) wp[16 * k] = v[k];
  }
  FFT_ORDER();
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    v2f u0 = lds_ld(xb + lane + 64 * m), u1 = lds_ld(xb + lane + 64 * m + 260), u2 = lds_ld(xb + lane + 64 * m + 520),
        u3 = lds_ld(xb + lane + 64 * m + 780);
    dft4<INV>(u0, u1, u2, u3);
    v[m] = u0; v[m + 4] = u1; v[m + 8] = u2; v[m + 12] = u3;
  }
  FFT_ORDER();
}
#else
// exchange a[lanes 32-63] with b[lanes 0-31] / a[lanes 16-31, 48-63] with b[lanes 0-15, 32-47]: one step of a transpose
// between a lane bit and a register-index bit, both directions in ONE instruction, no LDS
__device__ __forceinline__ void swap32(v2f& a, v2f& b) {
  const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
  a = (v2f){__uint_as_float(rx[0]), __uint_as_float(ry[0])};
  b = (v2f){__uint_as_float(rx[1]), __uint_as_float(ry[1])};
}
__device__ __forceinline__ void swap16(v2f& a, v2f& b) {
  const auto rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  const auto ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
  a = (v2f){__uint_as_float(rx[0]), __uint_as_float(ry[0])};
  b = (v2f){__uint_as_float(rx[1]), __uint_as_float(ry[1])};
}

template <bool INV>
__device__ __forceinline__ void fft1024(v2f (&v)[16], v2f* __restrict__ xb, int lane, const v2f* __restrict__ twl) {
#ifdef FFTV_ABL_NOFFT
  return;
#endif
  const v2f* t1 = twl + lane;
  const v2f* t2 = twl + 1024 + (lane >> 4);
  dft16<INV>(v);
#ifndef FFTV_NO_ASM_CMUL
  v[1] = twmul<INV>(v[1], lds_ld(t1 + 64));
#pragma unroll
  for (int k = 2; k < 16; k += 2) cmul2<INV>(v[k], lds_ld(t1 + 64 * k), v[k + 1], lds_ld(t1 + 64 * k + 64), v[k], v[k + 1]);
#else
#pragma unroll
  for (int k = 1; k < 16; ++k) v[k] = twmul<INV>(v[k], lds_ld(t1 + 64 * k));
#endif
#pragma unroll
  for (int k = 0; k < 16; ++k) xb[XP * k + lane] = v[k];
  FFT_ORDER();
  {
    const v2f* rp = xb + XP * (lane & 15) + (lane >> 4);
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) v[n2] = lds_ld(rp + 4 * n2);
  }
  FFT_ORDER();
  dft16<INV>(v);
#ifndef FFTV_NO_ASM_CMUL
  v[1] = twmul<INV>(v[1], lds_ld(t2 + 4));
#pragma unroll
  for (int k = 2; k < 16; k += 2) cmul2<INV>(v[k], lds_ld(t2 + 4 * k), v[k + 1], lds_ld(t2 + 4 * k + 4), v[k], v[k + 1]);
#else
#pragma unroll
  for (int k = 1; k < 16; ++k) v[k] = twmul<INV>(v[k], lds_ld(t2 + 4 * k));
#endif
  // lane 16 n3 + k1 holds k2 = 4 m + r in register 4 m + r; the 4 x 4 transposes put n3 into the register index:
  // lane 16 r + k1, register 4 m + n3
  v2f o[16];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    swap32(v[4 * m], v[4 * m + 2]);
    swap32(v[4 * m + 1], v[4 * m + 3]);
    swap16(v[4 * m], v[4 * m + 1]);
    swap16(v[4 * m + 2], v[4 * m + 3]);
    dft4<INV>(v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3]);
    o[m] = v[4 * m]; o[m + 4] = v[4 * m + 1]; o[m + 8] = v[4 * m + 2]; o[m + 12] = v[4 * m + 3];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) v[k] = o[k];
}
#endif

// Sum of a over the wave as a wave-uniform value: in-row inclusive scan (row_shr 1, 2, 4, 8), row totals carried by the
// two row broadcasts, total read from lane 63 - DPP operands only.  (__shfl_xor butterflies compile to six DEPENDENT
// ds_bpermute round trips, ~600 cycles of LDS latency at the end of every filter iteration of the forward kernel.)
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_mov(float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), CTRL, ROW_MASK, 0xf, BOUND));
}
__device__ __forceinline__ float wave_total(float a) {
  a += dpp_mov<0x111, 0xf, true>(a);      // row_shr:1
  a += dpp_mov<0x112, 0xf, true>(a);      // row_shr:2
  a += dpp_mov<0x114, 0xf, true>(a);      // row_shr:4
  a += dpp_mov<0x118, 0xf, true>(a);      // row_shr:8
  a += dpp_mov<0x142, 0xa, false>(a);     // row_bcast:15 into rows 1 and 3
  a += dpp_mov<0x143, 0xc, false>(a);     // row_bcast:31 into rows 2 and 3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
}

// segment of one electrode pair: v[j] = x[c0][t0 - padl + lane + 64 j] + i x[c0 + 1][...].  Interior segments of a full
// pair (12 of the 15 blocks of a 10000-sample row) take 32 unpredicated loads; the others one exec-masked load per element.
__device__ __forceinline__ void load_segment(v2f (&v)[16], const float* __restrict__ xrow, bool has1, int S, int tbase,
                                             int lane) {
  if (has1 && tbase >= 0 && tbase + NF <= S) {
    const float* p = xrow + tbase + lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = (v2f){p[64 * j], p[S + 64 * j]};
    return;
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int t = tbase + lane + 64 * j;
    const bool ok = t >= 0 && t < S;
    v[j].x = ok ? xrow[t] : 0.f;
    v[j].y = (ok && has1) ? xrow[S + t] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------------------------ fwd
// Work unit = (sample b, electrode pair, block of 704 outputs), one wave each; the NWF = 12 waves of a workgroup (3 per
// SIMD) share the 8 filter spectra H_f = conj(FFT(w_f)) / 1024, computed by the workgroup itself (wave f < 8 transforms
// filter f).  The filters are real, so H_f[1024 - k] = conj(H_f[k]): only bins 0 .. 512 are kept (33 KB instead of 64 KB -
// what makes room for the exchange buffers of 12 waves), rows j >= 8 of a lane read the mirrored bin and conjugate.
#ifdef FFTV_W8
constexpr int NWF = 8;
#else
constexpr int NWF = 12;
#endif
constexpr int HB = 520;         // float2 slots per filter: bins 0 .. 512, padded to a multiple of 64 bytes

template <bool STATS>
__global__ __launch_bounds__(64 * NWF, 1) void fir_fft_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ xidx,
                                                                  const float* __restrict__ w1, float* __restrict__ y1,
                                                                  float* __restrict__ part, int C, int S, int klen, int padl,
                                                                  int npair, int nblk, int nunits) {
  __shared__ v2f smem[F1 * HB + NWF * WBUF + TWSZ];
  v2f* Hs = smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave id as a scalar
  v2f* xb = smem + F1 * HB + wave * WBUF;
  const v2f* tw = smem + F1 * HB + NWF * WBUF;
  make_twiddles(smem + F1 * HB + NWF * WBUF);
  v2f v[16];
  if (wave < F1) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int n = lane + 64 * j;
      v[j] = (v2f){n < klen ? w1[wave * klen + n] : 0.f, 0.f};
    }
    fft1024<false>(v, xb, lane, tw);
#pragma unroll
    for (int j = 0; j < 8; ++j) Hs[wave * HB + j * 64 + lane] = (v2f){v[j].x, -v[j].y} * (1.0f / NF);
    if (lane == 0) Hs[wave * HB + 512] = (v2f){v[8].x, -v[8].y} * (1.0f / NF);
  }
  __syncthreads();
  float sacc = 0.f;      // lane f: sum of filter f's outputs, lane 8 + f: sum of their squares (this wave's share)
  const int wgid = blockIdx.x * NWF + wave, nwaves = gridDim.x * NWF;
  // the NEXT unit's input segment is fetched into registers before the eight inverse transforms of the current one
  v2f nx[16];
  auto fetch = [&](int u, v2f (&dst)[16]) {
    const int blk = u % nblk, pr = (u / nblk) % npair, b = u / (nblk * npair);
    const int c0 = 2 * pr;
    const float* xrow = x + ((xidx ? xidx[b] : (int64_t)b) * C + c0) * S;
    load_segment(dst, xrow, c0 + 1 < C, S, blk * LB - padl, lane);
  };
  if (wgid < nunits) fetch(wgid, nx);
  for (int u = wgid; u < nunits; u += nwaves) {
    const int blk = u % nblk, pr = (u / nblk) % npair, b = u / (nblk * npair);
    const int c0 = 2 * pr, t0 = blk * LB;
    const bool has1 = c0 + 1 < C;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = nx[j];
    if (u + nwaves < nunits) fetch(u + nwaves, nx);
    fft1024<false>(v, xb, lane, tw);
    v2f z[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) z[j] = v[j];
#pragma unroll 1
    for (int f = 0; f < F1; ++f) {
      const v2f* hp = Hs + f * HB + lane;       // bins lane + 64 j, j < 8
      const v2f* hm = Hs + f * HB - lane;       // bins 1024 - (lane + 64 j), j >= 8: the conjugates
#ifndef FFTV_NO_ASM_CMUL
#pragma unroll
      for (int j = 0; j < 8; j += 2) cmul2<false>(z[j], lds_ld(hp + 64 * j), z[j + 1], lds_ld(hp + 64 * j + 64), v[j], v[j + 1]);
#pragma unroll
      for (int j = 8; j < 16; j += 2)
        cmul2<true>(z[j], lds_ld(hm + (NF - 64 * j)), z[j + 1], lds_ld(hm + (NF - 64 * j - 64)), v[j], v[j + 1]);
#else
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = cmul(z[j], lds_ld(hp + 64 * j));
#pragma unroll
      for (int j = 8; j < 16; ++j) v[j] = cmulc(z[j], lds_ld(hm + (NF - 64 * j)));
#endif
      fft1024<true>(v, xb, lane, tw);
      float* dst = y1 + (((int64_t)b * F1 + f) * C + c0) * S + t0 + lane;
      float a1, a2;
      if (!STATS) {      // firstBN in eval mode: no batch statistics to collect (STATS = false: stat_part is not written)
        a1 = a2 = 0.f;
        if (has1 && t0 + LB <= S) {
#pragma unroll
          for (int j = 0; j < NROW; ++j) {
            EAV_STG(dst + 64 * j, v[j].x);
            EAV_STG(dst + S + 64 * j, v[j].y);
          }
        } else {
#pragma unroll
          for (int j = 0; j < NROW; ++j) {
            if (t0 + lane + 64 * j < S) {
              EAV_STG(dst + 64 * j, v[j].x);
              if (has1) EAV_STG(dst + S + 64 * j, v[j].y);
            }
          }
        }
        continue;
      }
      if (has1 && t0 + LB <= S) {      // a whole block of a full pair (14 of 15): no predicates
        v2f s1 = (v2f){0.f, 0.f}, s2 = (v2f){0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NROW; ++j) {
          EAV_STG(dst + 64 * j, v[j].x);
          EAV_STG(dst + S + 64 * j, v[j].y);
          s1 += v[j];
          s2 = fma2(v[j], v[j], s2);
        }
        a1 = s1.x + s1.y;
        a2 = s2.x + s2.y;
      } else {
        a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int j = 0; j < NROW; ++j) {
          if (t0 + lane + 64 * j < S) {
            EAV_STG(dst + 64 * j, v[j].x);
            a1 += v[j].x;
            a2 += v[j].x * v[j].x;
            if (has1) {
              EAV_STG(dst + S + 64 * j, v[j].y);
              a1 += v[j].y;
              a2 += v[j].y * v[j].y;
            }
          }
        }
      }
      a1 = wave_total(a1);
      a2 = wave_total(a2);
      if (lane == f) sacc += a1;
      if (lane == 8 + f) sacc += a2;
    }
  }
  if (STATS && lane < 16) part[wgid * 16 + lane] = sacc;      // (a wave without units writes its zeros)
}

// ---------------------------------------------------------------------------------------------------------------- wgrad
// Workgroup = 8 waves, wave f owns filter f's spectrum accumulator.  Per group of 8 units: wave w transforms the input
// segment of unit w into LDS (register layout), then every wave walks the 8 units, forms its filter's dy block, transforms
// it and accumulates Z conj(D).  PLAIN: BatchNorm in eval mode (dy = scale g, y1 is not read).
template <bool PLAIN>
__global__ __launch_bounds__(512, 1) void fir_fft_wgrad_kernel(const float* __restrict__ x, const int64_t* __restrict__ xidx,
                                                               const float* __restrict__ y1, const float* __restrict__ g1,
                                                               const float* __restrict__ bnp, float* __restrict__ spec,
                                                               int C, int S, int padl, int npair, int nblk, int nunits) {
  __shared__ v2f smem[8 * NF + 8 * WBUF + TWSZ];
  v2f* zb = smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), f = wave;
  v2f* xb = smem + 8 * NF + wave * WBUF;
  const v2f* tw = smem + 8 * NF + 8 * WBUF;
  make_twiddles(smem + 8 * NF + 8 * WBUF);
  const float mean = bnp[f], invstd = bnp[8 + f], sc = bnp[16 + f], m1 = bnp[32 + f], m2 = bnp[40 + f];
  v2f acc[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = (v2f){0.f, 0.f};
  v2f v[16];
  const int ngroups = (nunits + 7) >> 3;
  // raw g1 (and y1) rows of a unit for this wave's filter: .x = electrode c0, .y = electrode c0 + 1 (0 outside)
  v2f cg[NROW], cy[NROW];
  // (block, pair, sample) of a unit: one set of integer divisions per GROUP (they compile to ~60 scalar instructions each);
  // inside a group the position is advanced
  struct Pos { int blk, pr, b; };
  auto decode = [&](int u) { return Pos{u % nblk, (u / nblk) % npair, u / (nblk * npair)}; };
  auto advance = [&](Pos& q) {
    if (++q.blk == nblk) {
      q.blk = 0;
      if (++q.pr == npair) q.pr = 0, ++q.b;
    }
  };
  auto fetch_raw = [&](const Pos& q, v2f (&gr)[NROW], v2f (&yr)[NROW]) {
    const int blk = q.blk, pr = q.pr, b = q.b;
    const int c0 = 2 * pr, t0 = blk * LB;
    const bool has1 = c0 + 1 < C;
    const int64_t off = (((int64_t)b * F1 + f) * C + c0) * S + t0 + lane;
    if (has1 && t0 + LB <= S) {      // a whole block of a full pair: no predicates
#pragma unroll
      for (int j = 0; j < NROW; ++j) {
        gr[j] = (v2f){EAV_LDG(g1 + off + 64 * j), EAV_LDG(g1 + off + S + 64 * j)};
        if (!PLAIN) yr[j] = (v2f){EAV_LDG(y1 + off + 64 * j), EAV_LDG(y1 + off + S + 64 * j)};
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NROW; ++j) {
      const bool ok = t0 + lane + 64 * j < S;
      gr[j].x = ok ? EAV_LDG(g1 + off + 64 * j) : 0.f;
      gr[j].y = (ok && has1) ? EAV_LDG(g1 + off + S + 64 * j) : 0.f;
      if (!PLAIN) {
        yr[j].x = ok ? EAV_LDG(y1 + off + 64 * j) : 0.f;
        yr[j].y = (ok && has1) ? EAV_LDG(y1 + off + S + 64 * j) : 0.f;
      }
    }
  };
  Pos cur = decode(blockIdx.x * 8);      // the unit whose rows are in cg / cy
  if (blockIdx.x * 8 < nunits) fetch_raw(cur, cg, cy);
  // input segment of this wave's unit of a group: fetched one group ahead, so that the Z transforms that open a group
  // find their data in registers (the loads travel under the previous group's eight dy transforms)
  v2f nx[16];
  auto fetch_x = [&](int u, v2f (&dst)[16]) {
    const int blk = u % nblk, pr = (u / nblk) % npair, b = u / (nblk * npair);
    const int c0 = 2 * pr;
    const float* xrow = x + ((xidx ? xidx[b] : (int64_t)b) * C + c0) * S;
    load_segment(dst, xrow, c0 + 1 < C, S, blk * LB - padl, lane);
  };
  if (blockIdx.x * 8 + wave < nunits) fetch_x(blockIdx.x * 8 + wave, nx);
  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    if (g * 8 + wave < nunits) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = nx[j];
      const int un = (g + (int)gridDim.x) * 8 + wave;
      if (un < nunits) fetch_x(un, nx);
      fft1024<false>(v, xb, lane, tw);
#pragma unroll
      for (int j = 0; j < 16; ++j) zb[wave * NF + j * 64 + lane] = v[j];
    }
    __syncthreads();
#pragma unroll 1
    for (int uu = 0; uu < 8; ++uu) {
      const int u = g * 8 + uu;
      if (u >= nunits) break;
      const int t0 = cur.blk * LB;
      const bool has1 = 2 * cur.pr + 1 < C;
      // the next unit's rows (of this group, or the first of this workgroup's next group - they then travel under that
      // group's Z transforms too) are in flight under this unit's transform; they are consumed one iteration later
      v2f ng[NROW], ny[NROW];
      const int nu = (uu + 1 < 8) ? u + 1 : (g + (int)gridDim.x) * 8;
      if (uu + 1 < 8) advance(cur);
      else cur = decode(nu);
      if (nu < nunits) fetch_raw(cur, ng, ny);
#pragma unroll
      for (int j = 0; j < NROW; ++j) {
        if (PLAIN) {
          v[j] = sc * cg[j];
        } else {
          v[j] = sc * (cg[j] - m1 - (cy[j] - mean) * invstd * m2);
        }
      }
      if (!(has1 && t0 + LB <= S)) {      // ragged last block / odd electrode count: zero what lies outside
#pragma unroll
        for (int j = 0; j < NROW; ++j) {
          if (t0 + lane + 64 * j >= S) v[j] = (v2f){0.f, 0.f};
          if (!has1) v[j].y = 0.f;
        }
      }
#pragma unroll
      for (int j = NROW; j < 16; ++j) v[j] = (v2f){0.f, 0.f};
      fft1024<false>(v, xb, lane, tw);
      const v2f* zp = zb + uu * NF + lane;
#ifndef FFTV_NO_ASM_CMUL
#pragma unroll
      for (int j = 0; j < 16; j += 2) cmacc2_conj(lds_ld(zp + 64 * j), v[j], lds_ld(zp + 64 * j + 64), v[j + 1], acc[j], acc[j + 1]);
#else
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] += cmulc(lds_ld(zp + 64 * j), v[j]);
#endif
#pragma unroll
      for (int j = 0; j < NROW; ++j) {       // (after the transform: the copy is what waits for the loads)
        cg[j] = ng[j];
        if (!PLAIN) cy[j] = ny[j];
      }
    }
    __syncthreads();
  }
  float* out = spec + ((int64_t)blockIdx.x * F1 + f) * (2 * NF);
#pragma unroll
  for (int j = 0; j < 16; ++j) *reinterpret_cast<v2f*>(out + 2 * (j * 64 + lane)) = acc[j];
}

// Finish in two small launches.  (1) 256 workgroups = (filter, slice of 32 bins): 16 thread groups each sum every sixteenth
// per-workgroup spectrum of their bins in order (at 225 partials: 14 or 15 loads per thread, all in flight together), the
// 16 group sums are added as a fixed tree - bit-reproducible - and the total goes to row `nparts` of the workspace.
// (2) One wave per filter transforms the total back and keeps the real parts of lags 0 .. klen - 1.  (One workgroup per
// filter doing both took 30 us at 225 partials; 64 workgroups of 4 groups 16.6 us: a chain of 56 loads per thread.)
__global__ __launch_bounds__(512) void fir_fft_wgrad_sum_kernel(float* __restrict__ spec, int nparts) {
  __shared__ v2f red[512];
  const int f = blockIdx.x >> 5, bin = (blockIdx.x & 31) * 32 + (threadIdx.x & 31), grp = threadIdx.x >> 5;
  const v2f* src = reinterpret_cast<const v2f*>(spec) + (int64_t)f * NF + bin;
  v2f a = (v2f){0.f, 0.f};
#pragma unroll 4
  for (int p = grp; p < nparts; p += 16) a += src[(int64_t)p * F1 * NF];
  red[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < 128) {                 // groups g, g + 4, g + 8, g + 12
    a = (red[threadIdx.x] + red[threadIdx.x + 128]) + (red[threadIdx.x + 256] + red[threadIdx.x + 384]);
    red[threadIdx.x] = a;
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    a = (red[threadIdx.x] + red[threadIdx.x + 32]) + (red[threadIdx.x + 64] + red[threadIdx.x + 96]);
    reinterpret_cast<v2f*>(spec)[((int64_t)nparts * F1 + f) * NF + bin] = a;
  }
}

__global__ __launch_bounds__(64) void fir_fft_wgrad_finish_kernel(const float* __restrict__ spec, int nparts,
                                                                  float* __restrict__ dW, int klen) {
  __shared__ v2f smem[WBUF + TWSZ];
  const int lane = threadIdx.x, f = blockIdx.x;
  make_twiddles(smem + WBUF);
  v2f v[16];
  const v2f* src = reinterpret_cast<const v2f*>(spec) + ((int64_t)nparts * F1 + f) * NF + lane;
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = src[64 * j];
  fft1024<true>(v, smem, lane, smem + WBUF);
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int n = lane + 64 * j;
    if (n < klen) dW[f * klen + n] = v[j].x * (1.0f / NF);
  }
}

int fft_units(int B, int C, int S, int* npair, int* nblk) {
  *npair = (C + 1) / 2;
  *nblk = cdiv(S, LB);
  return B * *npair * *nblk;
}

// workgroups of 8 waves, at most one per CU; `items` work items of `per_wg` per round: the smallest grid that needs no
// more rounds than 256 workgroups would (14400 units = 1800 rounds-of-8: 225 workgroups x 8 rounds, no partial round)
int fft_grid(int nunits, int per_wg) {
  const int items = cdiv(nunits, per_wg);
#ifdef FFTV_G256
  return std::max(1, std::min(256, items));
#endif
  if (items <= 256) return std::max(1, items);
  const int rounds = cdiv(items, 256);
  return cdiv(items, rounds);
}

}  // namespace

extern "C" int eav_eegnet_fir_fft_max_taps(void) { return MAXK; }

// stat_part rows (16 floats each: 8 sums, 8 sums of squares) the forward writes - one per wave
extern "C" int eav_eegnet_fir_fwd_fft_nparts(int B, int C, int S) {
  int npair, nblk;
  const int nunits = fft_units(B, C, S, &npair, &nblk);
  return fft_grid(nunits, NWF) * NWF;
}

// y1 [B,8,C,S] = firstConv(x) for the batch x[xidx[0..B)] (xidx NULL: x itself), 'same' padding, klen <= 321 taps;
// stat_part [eav_eegnet_fir_fwd_fft_nparts][16]: per-wave sums / sums of squares per filter (input of eav_bn_finalize); NULL:
// none (firstBN in eval mode takes its running statistics - the sums, their wave totals and the partial rows are skipped).
extern "C" int eav_eegnet_fir_fwd_fft(const float* x, const int64_t* xidx, const float* w1, float* y1, float* stat_part,
                                      int B, int C, int S, int klen, void* stream) {
  EAV_REQUIRE(x && w1 && y1 && B > 0 && C > 0 && S > 0, "eav_eegnet_fir_fwd_fft: bad arguments");
  EAV_REQUIRE(klen >= 1 && klen <= MAXK, "eav_eegnet_fir_fwd_fft: kernLength %d outside [1,%d]", klen, MAXK);
  int npair, nblk;
  const int nunits = fft_units(B, C, S, &npair, &nblk);
  if (stat_part)
    hipLaunchKernelGGL(fir_fft_fwd_kernel<true>, dim3(fft_grid(nunits, NWF)), dim3(64 * NWF), 0, (hipStream_t)stream, x, xidx,
                       w1, y1, stat_part, C, S, klen, (klen - 1) / 2, npair, nblk, nunits);
  else      // firstBN in eval mode: running statistics, nothing to collect
    hipLaunchKernelGGL(fir_fft_fwd_kernel<false>, dim3(fft_grid(nunits, NWF)), dim3(64 * NWF), 0, (hipStream_t)stream, x, xidx,
                       w1, y1, stat_part, C, S, klen, (klen - 1) / 2, npair, nblk, nunits);
  EAV_CHECK_LAUNCH("eav_eegnet_fir_fwd_fft");
  return EAV_OK;
}

// floats of the spectrum workspace of eav_eegnet_fir_wgrad_fft
extern "C" int64_t eav_eegnet_fir_wgrad_fft_ws_floats(int B, int C, int S) {
  int npair, nblk;
  const int nunits = fft_units(B, C, S, &npair, &nblk);
  return ((int64_t)fft_grid(cdiv(nunits, 8), 1) + 1) * F1 * 2 * NF;      // + one row: the sum of the others
}

// dW [8, klen] = d loss / d firstConv.weight (written, not accumulated).  g1 = d loss / d(firstBN output) [B,8,C,S];
// y1 = the saved FIR output, or NULL for BatchNorm in eval mode (dy = scale g); bn_params as eav_bn_finalize /
// eav_bn_bwd_finalize leave them (mean, invstd, scale at 0 / 8 / 16, the backward means m1 / m2 at 32 / 40).
extern "C" int eav_eegnet_fir_wgrad_fft(const float* x, const int64_t* xidx, const float* y1, const float* g1,
                                        const float* bn_params, float* ws, float* dW, int B, int C, int S, int klen,
                                        void* stream) {
  EAV_REQUIRE(x && g1 && bn_params && ws && dW && B > 0 && C > 0 && S > 0, "eav_eegnet_fir_wgrad_fft: bad arguments");
  EAV_REQUIRE(klen >= 1 && klen <= MAXK, "eav_eegnet_fir_wgrad_fft: kernLength %d outside [1,%d]", klen, MAXK);
  int npair, nblk;
  const int nunits = fft_units(B, C, S, &npair, &nblk);
  const int grid = fft_grid(cdiv(nunits, 8), 1);
  hipStream_t st = (hipStream_t)stream;
  if (y1)
    hipLaunchKernelGGL(fir_fft_wgrad_kernel<false>, dim3(grid), dim3(512), 0, st, x, xidx, y1, g1, bn_params, ws, C, S,
                       (klen - 1) / 2, npair, nblk, nunits);
  else
    hipLaunchKernelGGL(fir_fft_wgrad_kernel<true>, dim3(grid), dim3(512), 0, st, x, xidx, y1, g1, bn_params, ws, C, S,
                       (klen - 1) / 2, npair, nblk, nunits);
  EAV_CHECK_LAUNCH("eav_eegnet_fir_wgrad_fft");
  hipLaunchKernelGGL(fir_fft_wgrad_sum_kernel, dim3(F1 * 32), dim3(512), 0, st, ws, grid);
  EAV_CHECK_LAUNCH("eav_eegnet_fir_wgrad_fft(sum)");
  hipLaunchKernelGGL(fir_fft_wgrad_finish_kernel, dim3(F1), dim3(64), 0, st, ws, grid, dW, klen);
  EAV_CHECK_LAUNCH("eav_eegnet_fir_wgrad_fft(finish)");
  return EAV_OK;
}
