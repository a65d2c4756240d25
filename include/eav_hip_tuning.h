/* TEST / TUNING ONLY - process-global overrides exported by libeav_hip.so beside the product ABI (include/eav_hip.h).
 * The trainers and the package never call them; the kernel benchmarks under tools/ and tests/test_split_kernels_gpu.py use
 * them to A/B tile shapes inside one process.  Each returns 0; the value is a plain global read at launch time. */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
int eav_gemm_sp_set_tile(int which);   /* 0 heuristic, 1 = 128x128 tiles, 2 = 256x128, +4 single accumulator, +8 non-persistent, +64 the 64x128 form wherever it applies, +128 never */
int eav_gemm_sp_set_splitk(int slices);  /* eav_gemm_sp_splitk: forced slice count (0 = the plan; ws must hold it) */
int eav_sp_set_convert_blocks(int n);  /* resident-block cap of eav_sp_convert (default 512; 0 = one block per tile) */
int eav_attn_sp_set_nw4_above(int n);   /* 128-row (4-wave) attention workgroups for N > n (default 128); n < 0: the software-pipelined forward from N >= -n (default 512) */
#ifdef __cplusplus
}
#endif
