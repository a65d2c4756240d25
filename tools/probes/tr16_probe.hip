// Probe of ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64) on gfx950: which (source lane, element) lands in
// which (destination lane, element).  LDS holds u16 value == its own index; lane l supplies the byte address 8 l (so its
// four source elements are 4l .. 4l+3).  Output line: "lane L: a b c d" where each value v = 4 * source_lane + source_elem.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/probes/tr16_probe.hip -o /tmp/tr16 && /tmp/tr16
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
__global__ void k(unsigned short* out) {
  __shared__ unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + 4 * threadIdx.x));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
  unsigned short* d;
  if (hipMalloc(&d, 64 * 4 * 2) != hipSuccess) return 1;
  k<<<1, 64>>>(d);
  unsigned short h[256];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int l = 0; l < 64; ++l)
    printf("lane %2d: %3d %3d %3d %3d   (src lane.elem: %d.%d %d.%d %d.%d %d.%d)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2],
           h[l * 4 + 3], h[l * 4] / 4, h[l * 4] % 4, h[l * 4 + 1] / 4, h[l * 4 + 1] % 4, h[l * 4 + 2] / 4, h[l * 4 + 2] % 4,
           h[l * 4 + 3] / 4, h[l * 4 + 3] % 4);
  return 0;
}
