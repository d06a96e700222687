// Internal launcher interface of the tiled MFMA GEMM (gemm_bf16.hip).
#pragma once
#include <hip/hip_runtime.h>

enum { CE_DT_BF16 = 0, CE_DT_F16 = 1 };
enum { EPI_STORE_F32 = 0, EPI_STORE_BF16 = 1, EPI_LNFOLD = 2, EPI_RESID = 3, EPI_THRESH = 4, EPI_STORE_FP8 = 5 };

struct GemmParams {
  const void* A; int lda;        // [M][lda] 16-bit elements, K-contiguous
  const void* W; int ldw;        // [N][ldw] 16-bit elements, K-contiguous
  int M, N, K;
  void* out; int ldo;            // [M][ldo]
  const float* bias;             // [N]
  const float* colsum;           // [N]   EPI_LNFOLD: sum_k W'[n][k]
  const float* stats_in;         // [parts][M][2] (sum, sumsq) of the rows of A   (EPI_LNFOLD)
  int stats_in_parts;
  int stats_ld;                  // row stride (in rows) between the parts of stats_in / stats_out (>= M, multiple of 256)
  float inv_width, eps;
  int act;                       // EPI_LNFOLD: CE_ACT_* or -1
  const void* resid;             // [M][ldo] bf16, may alias out                  (EPI_RESID)
  float* stats_out;              // [N/256][M][2]                                 (EPI_RESID)
  // EPI_THRESH (near-duplicate search): A == W == normalised embeddings, only tiles tn >= tm are
  // launched (tri != 0) and every (i < j < n_valid) with value > thr is appended to pairs/vals.
  int tri, n_valid, fp16_compare;
  float thr;
  long long* pairs;              // [cap][2]
  float* vals;                   // [cap]
  unsigned long long cap;
  unsigned long long* count;
  const unsigned* tile_list;     // [tt (tt + 1) / 2] tm | tn << 16: execution order of the upper-triangular tiles (set by ce_gemm_tri_persist)
  // fp8 path (gemm_fp8.hip): out = acc * scale_a[m] * scale_w[n] + bias[n]
  const float* scale_a;          // [M] per-token activation scale (NULL: 1)
  const float* scale_w;          // [N] per-output-channel weight scale
  const float* out_inv_scale;    // [N] EPI_STORE_FP8: out8[m][n] = e4m3(value * out_inv_scale[n])  (static per-column scale)
#ifdef CLIPENC_DIAG                // diagnostic build only (make diag -> libclipenc_hip_diag.so, used by tools/): the product
  unsigned long long* dbg;       // kernels carry no stamp hooks.  Optional [tiles][8] timing stamps.
#endif
};

hipError_t ce_gemm_nt(const GemmParams& p, int dtype, int epi, hipStream_t stream);
hipError_t ce_gemm_fp8(const GemmParams& p, int epi, hipStream_t stream);           // fp8 e4m3 operands; EPI_STORE_BF16 / EPI_RESID / EPI_STORE_FP8
hipError_t ce_gemm_nt_persist(const GemmParams& p, int epi, hipStream_t stream);   // bf16; EPI_STORE_BF16 / LNFOLD / RESID
hipError_t ce_gemm_tri_persist(const GemmParams& p, hipStream_t stream);            // f16 E.E^T, upper triangle, EPI_THRESH (gemm_tri.hip)
#include <vector>
std::vector<unsigned> tri_tile_order(int tt, int grid);                            // gemm_tri.hip: execution order of the triangular tile list
