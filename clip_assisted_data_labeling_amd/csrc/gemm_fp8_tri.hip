// Near-duplicate SCREEN on the e4m3 MFMA (dedup.hip: ce_dedup_pairs_screened): the persistent fp8 pipeline of gemm_fp8.hip over
// the upper-triangular tile list of gemm_tri.hip, candidate-append epilogue.  A translation unit of its own.
#define GEMM_FP8_TRI_TU 1
#include "gemm_fp8.hip"
