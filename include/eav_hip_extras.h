/* Comparison-only kernels of bench.py (`make -C eav_amd/csrc BENCH_EXTRAS=1` -> eav_amd/libeav_extras.so).  NOT part of the
 * product ABI (include/eav_hip.h): the default build, the trainers and `pytest -m gpu` do not need them.
 * Conventions as in eav_hip.h: int status, eav_last_error() of THIS library, device pointers, explicit stream. */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
const char* eav_last_error(void);
/* eav_gemm_f32 / eav_gemm_f32_splitk (eav_hip.h) with bf16 MFMA operands (fp32 in memory, rounded to bf16 while staging; fp32 accumulate and
 * output).  Opt-in fast mode: outside north_star's 1e-3 logit bound. */
int eav_gemm_bf16(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                  int transA, int transB, int batch, int heads, int64_t sAb, int64_t sAh, int64_t sBb, int64_t sBh,
                  int64_t sCb, int64_t sCh, float alpha, const float* bias, int gelu, float* pre, const float* resid,
                  int ldr, int accumulate, void* stream);
int eav_gemm_bf16_splitk(const float* A, const float* B, float* C, float* ws, int M, int N, int K, int lda, int ldb,
                         int transA, int transB, void* stream);
#ifdef __cplusplus
}
#endif
