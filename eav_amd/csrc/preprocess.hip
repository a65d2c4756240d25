// Frame pre-processing for the ViT path: what the reference does per frame on the
// host through the Hugging Face image processor (Transformer_Vision.py:52-59) - PIL bilinear resize of
// the uint8 HWC frame, x * (1/255), (x - mean) / std - as one kernel per batch of frames.
// The resize restates Pillow's 8-bit resampler exactly (ImagingResampleHorizontal/Vertical_8bpc:
// 22-bit fixed-point coefficients, horizontal pass rounded to uint8, then vertical pass); the
// coefficient tables are computed on the host the way Pillow's precompute_coeffs does.
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// one block per frame; tmp (horizontally resampled, [C][H][OW] uint8) lives in LDS
__global__ __launch_bounds__(256) void resize_norm_kernel(const uint8_t* __restrict__ in, const int* __restrict__ kx,
                                                          const int* __restrict__ bx, const int* __restrict__ ky,
                                                          const int* __restrict__ by, float* __restrict__ out, int H,
                                                          int W, int C, int OH, int OW, int ksx, int ksy,
                                                          double rescale, float m0, float m1, float m2, float s0,
                                                          float s1, float s2) {
  extern __shared__ uint8_t tmp[];
  const int img = blockIdx.x;
  const uint8_t* src = in + (int64_t)img * H * W * C;
  for (int idx = threadIdx.x; idx < C * H * OW; idx += 256) {
    const int xx = idx % OW, y = (idx / OW) % H, c = idx / (OW * H);
    const int xmin = bx[2 * xx], xcnt = bx[2 * xx + 1];
    int ss = 1 << (PRECISION_BITS - 1);
    for (int x = 0; x < xcnt; ++x) ss += (int)src[((int64_t)y * W + (x + xmin)) * C + c] * kx[xx * ksx + x];
    tmp[idx] = (uint8_t)clip8(ss);
  }
  __syncthreads();
  float* dst = out + (int64_t)img * C * OH * OW;
  for (int idx = threadIdx.x; idx < C * OH * OW; idx += 256) {
    const int xx = idx % OW, yy = (idx / OW) % OH, c = idx / (OW * OH);
    const int ymin = by[2 * yy], ycnt = by[2 * yy + 1];
    int ss = 1 << (PRECISION_BITS - 1);
    for (int y = 0; y < ycnt; ++y) ss += (int)tmp[(c * H + (y + ymin)) * OW + xx] * ky[yy * ksy + y];
    const float v = (float)((double)clip8(ss) * rescale);       // np_rescale: float64 product cast to float32
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), std = c == 0 ? s0 : (c == 1 ? s1 : s2);
    dst[idx] = (v - mean) / std;
  }
}


// ------------------------------------------------------------------------------------------------
// AST log-mel front-end: what AudioModelTrainer._feature_extract does per clip on
// the host (Transformer_Audio.py:38-42 -> HF ASTFeatureExtractor, numpy path): frames of 400 samples every
// 160, remove DC, pre-emphasis 0.97, Hann (symmetric) window, 512-point real FFT, power, 128 kaldi-mel
// filters (triangular in mel space, 20 Hz - 8 kHz), floor 1.19e-7, log, pad/truncate to max_len frames,
// (x - mean) / (2 std).  Everything up to the log is float64 like the numpy reference (incl. its rounding
// of the spectrum to complex64).  One block per (clip, frame).
__global__ __launch_bounds__(256) void ast_fbank_kernel(const float* __restrict__ wav, const double* __restrict__ window,
                                                        const double* __restrict__ tw /*[256][2] cos,-sin*/,
                                                        const double* __restrict__ melT /*[nmel][257]*/,
                                                        float* __restrict__ out, int L, int nframes, int max_len,
                                                        int nmel, double preemph, double mel_floor, float mean,
                                                        float std2) {
  __shared__ double re[512], im[512], pw[257], red[4];
  const int clip = blockIdx.y, frame = blockIdx.x;
  float* dst = out + ((int64_t)clip * max_len + frame) * nmel;
  if (frame >= nframes) {  // zero padding of the feature matrix, then normalised like every other value
    if ((int)threadIdx.x < nmel) dst[threadIdx.x] = (0.0f - mean) / std2;
    return;
  }
  const float* src = wav + (int64_t)clip * L + (int64_t)frame * 160;
  const int t = threadIdx.x;
  double a = (double)src[t], b = (t + 256 < 400) ? (double)src[t + 256] : 0.0;
  // mean over the 400 samples (fixed-order tree)
  double s = a + b;
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  const double mu = ((red[0] + red[1]) + (red[2] + red[3])) / 400.0;
  re[t] = a - mu;
  re[t + 256] = (t + 256 < 400) ? b - mu : 0.0;
  __syncthreads();
  // pre-emphasis (on the DC-removed frame) and window, written in bit-reversed order for the DIT FFT
  auto emph = [&](int i) { return i == 0 ? re[0] * (1.0 - preemph) : re[i] - preemph * re[i - 1]; };
  const double v0 = emph(t) * window[t];
  const double v1 = (t + 256 < 400) ? emph(t + 256) * window[t + 256] : 0.0;
  __syncthreads();
  auto brev9 = [](int i) { return (int)(__brev((unsigned)i) >> 23); };
  re[brev9(t)] = v0; im[brev9(t)] = 0.0;
  re[brev9(t + 256)] = v1; im[brev9(t + 256)] = 0.0;
  __syncthreads();
  for (int half = 1; half < 512; half <<= 1) {      // 9 radix-2 stages, one butterfly per thread
    const int j = t & (half - 1), base = ((t - j) << 1) + j;
    const int k = j * (256 / half);                  // twiddle index: W_512^(j * 512/(2*half))
    const double wr = tw[2 * k], wi = tw[2 * k + 1];
    const double xr = re[base + half], xi = im[base + half];
    const double tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
    const double ur = re[base], ui = im[base];
    re[base] = ur + tr; im[base] = ui + ti;
    re[base + half] = ur - tr; im[base + half] = ui - ti;
    __syncthreads();
  }
  for (int f = t; f < 257; f += 256) {              // complex64 rounding of the spectrum, |.| in float64, ^2
    const double r = (double)(float)re[f], i = (double)(float)im[f];
    const double mag = hypot(r, i);
    pw[f] = mag * mag;
  }
  __syncthreads();
  if (t < nmel) {
    const double* m = melT + (int64_t)t * 257;
    double acc = 0.0;
    for (int f = 0; f < 257; ++f) acc += m[f] * pw[f];
    const float v = (float)log(fmax(mel_floor, acc));
    dst[t] = (v - mean) / std2;
  }
}

}  // namespace

extern "C" int eav_resize_normalize_u8(const uint8_t* frames, const int* kx, const int* boundsx, const int* ky,
                                       const int* boundsy, float* out, int n, int H, int W, int C, int OH, int OW,
                                       int ksize_x, int ksize_y, double rescale, const float* mean3,
                                       const float* std3, void* stream) {
  EAV_REQUIRE(frames && kx && boundsx && ky && boundsy && out && mean3 && std3 && n > 0 && H > 0 && W > 0 && OH > 0 &&
                  OW > 0 && C >= 1 && C <= 3 && ksize_x > 0 && ksize_y > 0,
              "eav_resize_normalize_u8: bad arguments (mean3/std3 are HOST pointers to 3 floats)");
  const size_t lds = (size_t)C * H * OW;
  EAV_REQUIRE(lds <= 64 * 1024, "eav_resize_normalize_u8: C*H*OW = %zu exceeds the 64 KiB LDS tile", lds);
  hipLaunchKernelGGL(resize_norm_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, frames, kx, boundsx, ky,
                     boundsy, out, H, W, C, OH, OW, ksize_x, ksize_y, rescale, mean3[0], mean3[C > 1 ? 1 : 0],
                     mean3[C > 2 ? 2 : 0], std3[0], std3[C > 1 ? 1 : 0], std3[C > 2 ? 2 : 0]);
  EAV_CHECK_LAUNCH("eav_resize_normalize_u8");
  return EAV_OK;
}

extern "C" int eav_ast_fbank(const float* wav, const double* window400, const double* twiddle256,
                             const double* melT, float* out, int n, int L, int max_len, int nmel, double preemph,
                             double mel_floor, float mean, float std2, void* stream) {
  EAV_REQUIRE(wav && window400 && twiddle256 && melT && out && n > 0 && L >= 400 && max_len > 0 && nmel > 0 &&
                  nmel <= 256, "eav_ast_fbank: bad arguments (clips must hold at least one 400-sample frame)");
  const int nframes = 1 + (L - 400) / 160;
  dim3 grid(max_len, n);
  hipLaunchKernelGGL(ast_fbank_kernel, grid, dim3(256), 0, (hipStream_t)stream, wav, window400, twiddle256, melT, out,
                     L, nframes < max_len ? nframes : max_len, max_len, nmel, preemph, mel_floor, mean, std2);
  EAV_CHECK_LAUNCH("eav_ast_fbank");
  return EAV_OK;
}

// ------------------------------------------------------------------------------------------------
// EEG pre-processing (Dataload_eeg.py:85-121), float64 like scipy:
//  * downsampling(): scipy.signal.resample_poly(x, 1, down) = a zero-phase decimating FIR
//        y[c][m] = sum_j h[j] * x[c][m*down + center - j]           (zeros outside the record)
//  * bandpass_filter(): scipy.signal.sosfilt(sos, x) per channel - a cascade of biquads in direct form II
//    transposed, i.e. a linear recurrence over ~2e6 samples.  Parallelised exactly (no truncation):
//    (1) every chunk of Lc samples runs the cascade from a ZERO state (one thread per chunk) and records its
//        final state, (2) one thread per channel chains the true chunk-start states
//        z[c+1] = A^Lc z[c] + zend[c], (3) every sample adds the homogeneous response H[k] . z[chunk].
//    A^Lc and H come from the host (the same cascade run on unit initial states).
namespace {

__global__ __launch_bounds__(256) void decimate_fir_kernel(const double* __restrict__ x, const double* __restrict__ h,
                                                           double* __restrict__ y, int64_t n_in, int64_t n_out,
                                                           int down, int ntaps, int center) {
  const int c = blockIdx.y;
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= n_out) return;
  const double* xr = x + (int64_t)c * n_in;
  const int64_t base = m * down + center;
  double acc = 0.0;
  for (int j = 0; j < ntaps; ++j) {
    const int64_t i = base - j;
    if (i >= 0 && i < n_in) acc += h[j] * xr[i];
  }
  y[(int64_t)c * n_out + m] = acc;
}

constexpr int MAXSEC = 8;

// (1) zero-state cascade over one chunk; thread = (channel, chunk)
__global__ __launch_bounds__(64) void sos_chunk_kernel(const double* __restrict__ x, double* __restrict__ y,
                                                       const double* __restrict__ sos, double* __restrict__ zend,
                                                       int nch, int64_t n, int nsec, int Lc, int nchunk) {
  const int id = blockIdx.x * 64 + threadIdx.x;
  if (id >= nch * nchunk) return;
  const int c = id / nchunk, q = id - c * nchunk;
  double b0[MAXSEC], b1[MAXSEC], b2[MAXSEC], a1[MAXSEC], a2[MAXSEC], z0[MAXSEC], z1[MAXSEC];
#pragma unroll
  for (int s = 0; s < MAXSEC; ++s) {
    if (s < nsec) {
      b0[s] = sos[s * 6 + 0]; b1[s] = sos[s * 6 + 1]; b2[s] = sos[s * 6 + 2];
      a1[s] = sos[s * 6 + 4]; a2[s] = sos[s * 6 + 5];
    }
    z0[s] = z1[s] = 0.0;
  }
  const int64_t n0 = (int64_t)q * Lc, n1 = min(n, n0 + Lc);
  const double* xr = x + (int64_t)c * n;
  double* yr = y + (int64_t)c * n;
  for (int64_t i = n0; i < n1; ++i) {
    double v = xr[i];
#pragma unroll
    for (int s = 0; s < MAXSEC; ++s)
      if (s < nsec) {   // scipy _sosfilt: direct form II transposed
        const double o = b0[s] * v + z0[s];
        z0[s] = b1[s] * v - a1[s] * o + z1[s];
        z1[s] = b2[s] * v - a2[s] * o;
        v = o;
      }
    yr[i] = v;
  }
  double* ze = zend + (int64_t)id * 2 * nsec;
  for (int s = 0; s < nsec; ++s) { ze[2 * s] = z0[s]; ze[2 * s + 1] = z1[s]; }
}

// (2) chain the chunk-start states of one channel: zs[q+1] = AL . zs[q] + zend[q]  (chunks are full here)
__global__ void sos_scan_kernel(const double* __restrict__ zend, const double* __restrict__ AL,
                                double* __restrict__ zstart, int nch, int nsec, int nchunk) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nch) return;
  const int ns = 2 * nsec;
  double z[2 * MAXSEC], t[2 * MAXSEC];
  for (int j = 0; j < ns; ++j) z[j] = 0.0;
  for (int q = 0; q < nchunk; ++q) {
    double* dst = zstart + ((int64_t)c * nchunk + q) * ns;
    for (int j = 0; j < ns; ++j) dst[j] = z[j];
    const double* ze = zend + ((int64_t)c * nchunk + q) * ns;
    for (int i = 0; i < ns; ++i) {
      double a = ze[i];
      for (int j = 0; j < ns; ++j) a += AL[i * ns + j] * z[j];
      t[i] = a;
    }
    for (int j = 0; j < ns; ++j) z[j] = t[j];
  }
}

// (3) y[n] += H[n - n0] . zstart[chunk]
__global__ __launch_bounds__(256) void sos_fixup_kernel(double* __restrict__ y, const double* __restrict__ H,
                                                        const double* __restrict__ zstart, int64_t n, int nsec,
                                                        int Lc, int nchunk) {
  const int c = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int q = (int)(i / Lc), k = (int)(i - (int64_t)q * Lc);
  if (q == 0) return;                                   // the first chunk starts from the true zero state
  const int ns = 2 * nsec;
  const double* z = zstart + ((int64_t)c * nchunk + q) * ns;
  const double* hk = H + (int64_t)k * ns;
  double a = 0.0;
  for (int j = 0; j < ns; ++j) a += hk[j] * z[j];
  y[(int64_t)c * n + i] += a;
}

}  // namespace

extern "C" int eav_decimate_fir_f64(const double* x, const double* h, double* y, int nch, int64_t n_in, int64_t n_out,
                                    int down, int ntaps, int center, void* stream) {
  EAV_REQUIRE(x && h && y && nch > 0 && n_in > 0 && n_out > 0 && down > 0 && ntaps > 0, "eav_decimate_fir_f64: bad arguments");
  dim3 grid((unsigned)cdiv64(n_out, 256), nch);
  hipLaunchKernelGGL(decimate_fir_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, h, y, n_in, n_out, down, ntaps, center);
  EAV_CHECK_LAUNCH("eav_decimate_fir_f64");
  return EAV_OK;
}

extern "C" int eav_sosfilt_f64(const double* x, double* y, const double* sos, const double* H, const double* AL,
                               double* zend, double* zstart, int nch, int64_t n, int nsec, int Lc, void* stream) {
  EAV_REQUIRE(x && y && sos && H && AL && zend && zstart && nch > 0 && n > 0 && nsec > 0 && nsec <= MAXSEC && Lc > 0,
              "eav_sosfilt_f64: bad arguments (at most %d sections)", MAXSEC);
  const int nchunk = (int)cdiv64(n, Lc);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(sos_chunk_kernel, dim3(cdiv(nch * nchunk, 64)), dim3(64), 0, st, x, y, sos, zend, nch, n, nsec, Lc, nchunk);
  EAV_CHECK_LAUNCH("eav_sosfilt_f64(chunks)");
  hipLaunchKernelGGL(sos_scan_kernel, dim3(cdiv(nch, 64)), dim3(64), 0, st, zend, AL, zstart, nch, nsec, nchunk);
  EAV_CHECK_LAUNCH("eav_sosfilt_f64(scan)");
  dim3 grid((unsigned)cdiv64(n, 256), nch);
  hipLaunchKernelGGL(sos_fixup_kernel, grid, dim3(256), 0, st, y, H, zstart, n, nsec, Lc, nchunk);
  EAV_CHECK_LAUNCH("eav_sosfilt_f64(fixup)");
  return EAV_OK;
}
