/* Diagnostic entry points of libclipenc_hip_diag.so -- NOT part of the product ABI.
 *
 * Built only by `make -C clip_assisted_data_labeling_amd/csrc diag` (every source compiled with -DCLIPENC_DIAG); used by the
 * developer tools under tools/ (gemm_stamps.py).  The product library exports none of these and its kernels carry no
 * stamp hooks. */
#ifndef CLIPENC_DIAG_H
#define CLIPENC_DIAG_H
#include "clipenc.h"
#ifdef __cplusplus
extern "C" {
#endif
/* bf16-store GEMM that also writes, per tile, in-kernel stamps into stamps_dev[tiles][8]:
 * {workgroup, main loop start, main loop end, epilogue stores issued} in 100 MHz ticks (s_memrealtime) and the shader
 * cycle counter (s_memtime) at main loop start / end. */
int clipenc_op_gemm_nt_stamps(const void* a_dev, const void* w_dev, int m, int n, int k, void* out_dev,
                              unsigned long long* stamps_dev, void* stream);
/* The LayerNorm-folded GEMM on its own (QKV / FC1 of a block): out = act(rstd_m (A.W'^T - mean_m colsum_n) + bias_n), bf16;
 * stats_dev = [parts][stats_ld][2] raw (sum, sum of squares) of the rows of A; act -1 / 0 (QuickGELU) / 1 (erf GELU);
 * stamps_dev may be NULL. */
int clipenc_op_gemm_lnfold(const void* a_dev, const void* w_dev, int m, int n, int k, const float* colsum_dev,
                           const float* bias_dev, const float* stats_dev, int parts, int stats_ld, int act, void* out_dev,
                           unsigned long long* stamps_dev, void* stream);
/* The residual GEMM on its own (out-projection / FC2 of a block): x += A.W^T + bias in place (bf16), row statistics of the
 * rounded rows to stats_out_dev [n / 256][stats_ld][2]; stamps_dev may be NULL. */
int clipenc_op_gemm_resid(const void* a_dev, const void* w_dev, int m, int n, int k, const float* bias_dev, void* x_inout_dev,
                          float* stats_out_dev, int stats_ld, unsigned long long* stamps_dev, void* stream);
/* The bf16-store GEMM with explicit leading dimensions in elements (multiples of 8, >= k / n): row-pitch experiments. */
int clipenc_op_gemm_nt_ld(const void* a_dev, int lda, const void* w_dev, int ldw, int m, int n, int k, void* out_dev, int ldo, void* stream);
/* The fp8 GEMM ops of clipenc.h (clipenc_op_gemm_fp8, _q, _lnf, _resid_q) write the same [tiles][8] stamps into stamps_dev
 * from the next call on (NULL: off again; tools/gemm_fp8_stamps.py). */
int clipenc_diag_fp8_stamps(unsigned long long* stamps_dev);
#ifdef __cplusplus
}
#endif
#endif /* CLIPENC_DIAG_H */
