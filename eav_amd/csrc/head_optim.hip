// Classifier head, loss and optimiser kernels shared by the three encoders.
//   dense + softmax (EEGNet_tor.py:64-66), cross-entropy with mean reduction
//   (nn.CrossEntropyLoss: EEGNet_tor.py:81,105; Transformer_Audio.py:31,73; HF ViT loss),
//   Adam / AdamW single-tensor update (EEGNet_tor.py:82; Transformer_Audio.py:30; Transformer_Vision.py:36).
#include "eav_common.h"
#include "../../include/eav_hip.h"

namespace {

constexpr int NCMAX = 16;

// one block (1024 threads) per sample: logits[j] = bias[j] + sum_i W[j,i]*in[b,i];  probs = softmax
__global__ __launch_bounds__(1024) void dense_softmax_fwd_kernel(const float* __restrict__ in,
                                                                 const float* __restrict__ w,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ logits, float* __restrict__ probs,
                                                                 int NF, int NC) {
  __shared__ float red[16 * NCMAX];
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* src = in + (int64_t)b * NF;
  float acc[NCMAX];
#pragma unroll
  for (int j = 0; j < NCMAX; ++j) acc[j] = 0.f;
  if ((NF & 3) == 0) {      // 16 B per lane: the kernel is pure load latency (one block per sample)
    const int nq = NF >> 2;
    if (NC <= 5) {
      // the reference's 5 classes: four strides of the feature vector at a time, all 4 + 20 loads in flight before the first
      // product (one stride per trip left a 64-block launch waiting on memory five times over: 19 us at NF = 19 968)
      for (int i0 = threadIdx.x; i0 < nq; i0 += 4096) {
        float4 v[4], wv[4][5];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + 1024 * u;
          const bool ok = i < nq;
          v[u] = ok ? reinterpret_cast<const float4*>(src)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int j = 0; j < 5; ++j)
            wv[u][j] = (ok && j < NC) ? reinterpret_cast<const float4*>(w + (int64_t)j * NF)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < 5; ++j)
            acc[j] += (wv[u][j].x * v[u].x + wv[u][j].y * v[u].y) + (wv[u][j].z * v[u].z + wv[u][j].w * v[u].w);
      }
    } else {
      for (int i = threadIdx.x; i < nq; i += 1024) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
#pragma unroll
        for (int j = 0; j < NCMAX; ++j)
          if (j < NC) {
            const float4 wv = reinterpret_cast<const float4*>(w + (int64_t)j * NF)[i];
            acc[j] += (wv.x * v.x + wv.y * v.y) + (wv.z * v.z + wv.w * v.w);
          }
      }
    }
  } else {
    for (int i = threadIdx.x; i < NF; i += 1024) {
      const float v = src[i];
#pragma unroll
      for (int j = 0; j < NCMAX; ++j)
        if (j < NC) acc[j] += w[(int64_t)j * NF + i] * v;
    }
  }
#pragma unroll
  for (int j = 0; j < NCMAX; ++j)
    if (j < NC) {
      float s = wave_sum(acc[j]);
      if (lane == 0) red[wave * NCMAX + j] = s;
    }
  __syncthreads();
  __shared__ float lg[NCMAX];
  if (threadIdx.x < NC) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k * NCMAX + threadIdx.x];
    lg[threadIdx.x] = s + bias[threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float mx = lg[0];
    for (int j = 1; j < NC; ++j) mx = fmaxf(mx, lg[j]);
    float e[NCMAX], s = 0.f;
    for (int j = 0; j < NC; ++j) { e[j] = expf(lg[j] - mx); s += e[j]; }
    for (int j = 0; j < NC; ++j) {
      if (logits) logits[b * NC + j] = lg[j];
      if (probs) probs[b * NC + j] = e[j] / s;
    }
  }
}

// dlogits from dprobs (softmax backward) or passed through;
//   dW[j,i] = sum_b dl[b,j]*in[b,i];  din[b,i] = sum_j W[j,i]*dl[b,j];  dbias[j] = sum_b dl[b,j]
// A block owns 64 input features i; its 256 threads = 64 features x 4 slices of the batch (rows b = slice, slice + 4, ...:
// the chain of dependent load latencies is a quarter as long as with one thread per feature - the kernel is pure load
// latency: 36 -> ~14 us at B = 64, NF = 4992); the four partial dW rows are added through LDS in slice order.
__global__ __launch_bounds__(256) void dense_softmax_bwd_kernel(const float* __restrict__ dout,
                                                                const float* __restrict__ probs,
                                                                const float* __restrict__ in,
                                                                const float* __restrict__ w, float* __restrict__ dw,
                                                                float* __restrict__ dbias, float* __restrict__ din,
                                                                int B, int NF, int NC) {
  extern __shared__ float dl[];  // [B][NC], then [4][NCMAX][64] partial dW
  float* pw = dl + B * NC;
  for (int idx = threadIdx.x; idx < B * NC; idx += blockDim.x) {
    const int b = idx / NC;
    float v = dout[idx];
    if (probs) {  // dl = p * (dp - sum_j dp_j p_j)
      float dot = 0.f;
      for (int j = 0; j < NC; ++j) dot += dout[b * NC + j] * probs[b * NC + j];
      v = probs[idx] * (v - dot);
    }
    dl[idx] = v;
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < NC) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dl[b * NC + threadIdx.x];
    dbias[threadIdx.x] = s;
  }
  const int f = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + f;
  const bool live = i < NF;
  float wv[NCMAX], acc[NCMAX];
#pragma unroll
  for (int j = 0; j < NCMAX; ++j) {
    wv[j] = (live && j < NC) ? w[(int64_t)j * NF + i] : 0.f;
    acc[j] = 0.f;
  }
  // eight rows per trip: the eight input loads go out together (one latency per trip, not per row)
  for (int b0 = slice; b0 < B; b0 += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (live && b0 + 4 * u < B) ? in[(int64_t)(b0 + 4 * u) * NF + i] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 4 * u;
      if (b < B) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < NCMAX; ++j)
          if (j < NC) {
            const float g = dl[b * NC + j];
            acc[j] += g * v[u];
            d += wv[j] * g;
          }
        if (din && live) din[(int64_t)b * NF + i] = d;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NCMAX; ++j) pw[(slice * NCMAX + j) * 64 + f] = acc[j];
  __syncthreads();
  if (slice == 0 && live) {
#pragma unroll
    for (int j = 0; j < NCMAX; ++j)
      if (j < NC)
        dw[(int64_t)j * NF + i] = (pw[j * 64 + f] + pw[(NCMAX + j) * 64 + f]) + (pw[(2 * NCMAX + j) * 64 + f] + pw[(3 * NCMAX + j) * 64 + f]);
  }
}

// mean cross-entropy over B rows of `in` (treated as logits) + gradient (softmax - onehot)/B
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ in, const int64_t* __restrict__ y,
                                                 float* __restrict__ loss, float* __restrict__ din,
                                                 int* __restrict__ ncorrect, int* __restrict__ bad_label, int B,
                                                 int NC) {
  // nn.CrossEntropyLoss semantics: targets equal to -100 (torch's default ignore_index) are excluded from the mean and
  // receive a zero gradient; any other index outside [0, NC) is an error - reported through `bad_label`, and the row is
  // treated like an ignored one (zero gradient: a bad label never pushes a wrong update into the weights).
  __shared__ float red[12];
  float nv[1] = {0.f};
  for (int b = threadIdx.x; b < B; b += 256) {
    const int64_t yl = y[b];
    nv[0] += (yl >= 0 && yl < NC) ? 1.f : 0.f;
  }
  block_sum_256<1>(nv, red);
  if (threadIdx.x == 0) red[8] = nv[0];
  __syncthreads();
  const float nvalid = red[8];
  float st[2] = {0.f, 0.f};
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* r = in + (int64_t)b * NC;
    float mx = r[0];
    int am = 0;
    for (int j = 1; j < NC; ++j)
      if (r[j] > mx) { mx = r[j]; am = j; }
    float s = 0.f;
    for (int j = 0; j < NC; ++j) s += expf(r[j] - mx);
    const float lse = mx + logf(s);
    const int64_t yl = y[b];
    const bool ok = yl >= 0 && yl < NC;        // torch asserts on class indices outside [0, NC); never index with one
    if (!ok && yl != -100 && bad_label)
      *bad_label = yl >= 0 ? (int)min(yl, (int64_t)0x7ffffffe) + 1 : (int)max(yl, (int64_t)-0x7fffffff);
    const int yy = ok ? (int)yl : -1;
    st[0] += ok ? lse - r[yy] : 0.f;
    st[1] += (am == yy) ? 1.f : 0.f;
    if (din)
      for (int j = 0; j < NC; ++j)
        din[(int64_t)b * NC + j] = ok ? (expf(r[j] - lse) - (j == yy ? 1.f : 0.f)) / nvalid : 0.f;
  }
  block_sum_256<2>(st, red);
  // all targets ignored: torch returns nan (0 / 0)
  if (threadIdx.x == 0 && loss) *loss = nvalid > 0.f ? st[0] / nvalid : __builtin_nanf("");
  if (threadIdx.x == 1 && ncorrect) *ncorrect += (int)(st[0] + 0.5f);
}

// torch.optim.Adam / AdamW (foreach/fused semantics, fp32 state):
//   adamw: p *= 1 - lr*wd;  adam: g += wd*p
//   m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float wd, float bc1,
                                                   float bc2_sqrt, int decoupled,
                                                   const int64_t* __restrict__ step_dev) {
  if (step_dev) {  // capturable: bias corrections from a device-resident step count
    const double st = (double)*step_dev;
    bc1 = (float)(1.0 - pow((double)b1, st));
    bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, st));
  }
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float pv = p[i], gv = g[i];
    if (wd != 0.f) {
      if (decoupled) pv *= 1.f - lr * wd; else gv += wd * pv;
    }
    const float mv = b1 * m[i] + (1.f - b1) * gv;
    const float vv = b2 * v[i] + (1.f - b2) * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = pv - (lr / bc1) * (mv / denom);
  }
}

}  // namespace

extern "C" int eav_dense_softmax_fwd(const float* in, const float* w, const float* bias, float* logits, float* probs,
                                     int B, int NF, int NC, void* stream) {
  EAV_REQUIRE(in && w && bias && (logits || probs) && B > 0 && NF > 0 && NC > 0 && NC <= NCMAX,
              "eav_dense_softmax_fwd: bad arguments (classes <= %d)", NCMAX);
  hipLaunchKernelGGL(dense_softmax_fwd_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, in, w, bias, logits, probs,
                     NF, NC);
  EAV_CHECK_LAUNCH("eav_dense_softmax_fwd");
  return EAV_OK;
}

extern "C" int eav_dense_softmax_bwd(const float* dout, const float* probs, const float* in, const float* w,
                                     float* dw, float* dbias, float* din, int B, int NF, int NC, void* stream) {
  EAV_REQUIRE(dout && in && w && dw && dbias && B > 0 && NF > 0 && NC > 0 && NC <= NCMAX,
              "eav_dense_softmax_bwd: bad arguments (classes <= %d)", NCMAX);
  EAV_REQUIRE((size_t)B * NC * sizeof(float) <= 40 * 1024, "eav_dense_softmax_bwd: batch %d too large", B);
  hipLaunchKernelGGL(dense_softmax_bwd_kernel, dim3(cdiv(NF, 64)), dim3(256),
                     (B * NC + 4 * NCMAX * 64) * sizeof(float), (hipStream_t)stream, dout, probs, in, w, dw, dbias, din, B,
                     NF, NC);
  EAV_CHECK_LAUNCH("eav_dense_softmax_bwd");
  return EAV_OK;
}

extern "C" int eav_ce_fwd_bwd(const float* in, const int64_t* y, float* loss, float* din, int* ncorrect,
                              int* bad_label, int B, int NC, void* stream) {
  EAV_REQUIRE(in && y && B > 0 && NC > 0, "eav_ce_fwd_bwd: bad arguments");
  hipLaunchKernelGGL(ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, in, y, loss, din, ncorrect, bad_label, B,
                     NC);
  EAV_CHECK_LAUNCH("eav_ce_fwd_bwd");
  return EAV_OK;
}

// v[i] *= *scalar (device scalar): applies the upstream gradient of the loss to its stored input gradient
__global__ __launch_bounds__(256) void scale_by_scalar_kernel(float* __restrict__ v, const float* __restrict__ sc,
                                                              int64_t n) {
  const float s = *sc;
  if (s == 1.f) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) v[i] *= s;
}

extern "C" int eav_scale_by_scalar(float* v, const float* scalar, int64_t n, void* stream) {
  EAV_REQUIRE(v && scalar && n > 0, "eav_scale_by_scalar: bad arguments");
  int64_t blocks = cdiv64(n, 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(scale_by_scalar_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, v, scalar, n);
  EAV_CHECK_LAUNCH("eav_scale_by_scalar");
  return EAV_OK;
}

extern "C" int eav_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int64_t step, int decoupled,
                             const int64_t* step_dev, void* stream) {
  EAV_REQUIRE(p && g && m && v && n > 0 && (step >= 1 || step_dev), "eav_adam_step: bad arguments");
  if (step < 1) step = 1;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  int64_t blocks = cdiv64(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), decoupled, step_dev);
  EAV_CHECK_LAUNCH("eav_adam_step");
  return EAV_OK;
}

__global__ void counter_inc_kernel(int64_t* p) { *p += 1; }

__global__ void counter_inc4_kernel(int64_t* a, int64_t* b, int64_t* c, int64_t* d) {
  int64_t* p = threadIdx.x == 0 ? a : threadIdx.x == 1 ? b : threadIdx.x == 2 ? c : d;
  if (p) *p += 1;
}

// up to four distinct counters in one launch (NULL entries are skipped)
extern "C" int eav_counter_inc4(int64_t* c0, int64_t* c1, int64_t* c2, int64_t* c3, void* stream) {
  EAV_REQUIRE(c0 || c1 || c2 || c3, "eav_counter_inc4: no counter");
  hipLaunchKernelGGL(counter_inc4_kernel, dim3(1), dim3(4), 0, (hipStream_t)stream, c0, c1, c2, c3);
  EAV_CHECK_LAUNCH("eav_counter_inc4");
  return EAV_OK;
}

extern "C" int eav_counter_inc(int64_t* counter, void* stream) {
  EAV_REQUIRE(counter, "eav_counter_inc: null counter");
  hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter);
  EAV_CHECK_LAUNCH("eav_counter_inc");
  return EAV_OK;
}

// Batch assembly: out[i, :] = src[idx[i], :] for rows of `row_elems` floats (16-B vectors when possible).
// Replaces the per-batch host copy of the reference loops (EEGNet_tor.py:100-101, Transformer_Audio.py:70):
// the split lives in HBM and a batch is one gather.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx,
                                                          float* __restrict__ out, int64_t row_elems, int vec) {
  const int64_t r = idx[blockIdx.y];
  if (vec) {
    const float4* s = reinterpret_cast<const float4*>(src + r * row_elems);
    float4* d = reinterpret_cast<float4*>(out + (int64_t)blockIdx.y * row_elems);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < row_elems / 4; i += (int64_t)gridDim.x * 256) d[i] = s[i];
  } else {
    const float* s = src + r * row_elems;
    float* d = out + (int64_t)blockIdx.y * row_elems;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < row_elems; i += (int64_t)gridDim.x * 256) d[i] = s[i];
  }
}

__global__ void gather_i64_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ idx,
                                  int64_t* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

// The start of a captured training step in ONE launch (EEGNet_tor.py:100-104's batch assembly + the counters nn.BatchNorm2d /
// the dropout stream / optim.Adam keep): up to five distinct step counters incremented by one thread, and the labels of the
// batch gathered (out[i] = labels[idx[i]]).  Replaces eav_counter_inc4 + eav_counter_inc + eav_gather_i64: three graph
// nodes at the ~4.7 us floor of a dependent node.
__global__ void step_begin_kernel(int64_t* c0, int64_t* c1, int64_t* c2, int64_t* c3, int64_t* c4,
                                  const int64_t* __restrict__ labels, const int64_t* __restrict__ idx,
                                  int64_t* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    if (c0) *c0 += 1;
    if (c1) *c1 += 1;
    if (c2) *c2 += 1;
    if (c3) *c3 += 1;
    if (c4) *c4 += 1;
  }
  if (i < n) out[i] = labels[idx[i]];
}

extern "C" int eav_step_begin(int64_t* c0, int64_t* c1, int64_t* c2, int64_t* c3, int64_t* c4, const int64_t* labels,
                              const int64_t* idx, int64_t* out, int n, void* stream) {
  EAV_REQUIRE(n >= 0 && (n == 0 || (labels && idx && out)), "eav_step_begin: bad arguments");
  EAV_REQUIRE(c0 || c1 || c2 || c3 || c4 || n > 0, "eav_step_begin: nothing to do");
  hipLaunchKernelGGL(step_begin_kernel, dim3(n > 0 ? cdiv(n, 256) : 1), dim3(256), 0, (hipStream_t)stream, c0, c1, c2, c3, c4,
                     labels, idx, out, n);
  EAV_CHECK_LAUNCH("eav_step_begin");
  return EAV_OK;
}

extern "C" int eav_gather_rows(const float* src, const int64_t* idx, float* out, int nrows, int64_t row_elems,
                               void* stream) {
  EAV_REQUIRE(src && idx && out && nrows > 0 && row_elems > 0, "eav_gather_rows: bad arguments");
  const int vec = (row_elems % 4 == 0) && ((((uintptr_t)src | (uintptr_t)out) & 15) == 0);
  int64_t per = vec ? row_elems / 4 : row_elems;
  int gx = (int)(cdiv64(per, 256) > 64 ? 64 : cdiv64(per, 256));
  hipLaunchKernelGGL(gather_rows_kernel, dim3(gx, nrows), dim3(256), 0, (hipStream_t)stream, src, idx, out, row_elems, vec);
  EAV_CHECK_LAUNCH("eav_gather_rows");
  return EAV_OK;
}

extern "C" int eav_gather_i64(const int64_t* src, const int64_t* idx, int64_t* out, int n, void* stream) {
  EAV_REQUIRE(src && idx && out && n > 0, "eav_gather_i64: bad arguments");
  hipLaunchKernelGGL(gather_i64_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, src, idx, out, n);
  EAV_CHECK_LAUNCH("eav_gather_i64");
  return EAV_OK;
}
