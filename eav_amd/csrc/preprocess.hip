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
