// Internal launcher interface of the tiled MFMA GEMM (gemm_bf16.hip).
#pragma once
#include <hip/hip_runtime.h>

enum { CE_DT_BF16 = 0, CE_DT_F16 = 1 };
enum { EPI_STORE_F32 = 0, EPI_STORE_BF16 = 1, EPI_LNFOLD = 2, EPI_RESID = 3, EPI_THRESH = 4, EPI_STORE_FP8 = 5, EPI_RESID_Q = 6 };

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
  unsigned* ticket;              // optional (bf16 persistent kernel, EPI_STORE_BF16 / LNFOLD / RESID): a zeroed device word.  The tiles of the launch's last
                                 // one-to-two rounds are then handed out from it in the order the workgroups get there instead of by stride, so that
                                 // a workgroup that runs ahead takes more of them (gemm_persist.hip); the results do not depend on who runs a tile
  const unsigned* tile_list;     // [tt (tt + 1) / 2] tm | tn << 16: execution order of the upper-triangular tiles (set by ce_gemm_tri_persist)
  const unsigned long long* run_if_over;   // EPI_THRESH, optional: the launch does nothing unless *run_if_over > run_if_limit (read on the
  unsigned long long run_if_limit;         // device when the launch starts: the exact search as the fall-back of the screened one, dedup.hip)
  // fp8 path (gemm_fp8.hip): out = acc * scale_a[m] * scale_w[n] + bias[n]
  const float* scale_a;          // [M] per-token activation scale (NULL: 1)
  const float* scale_w;          // [N] per-output-channel weight scale
  const unsigned char* w_exp;    // [N] block-exponent consumers / producer (a_exp != NULL or EPI_RESID_Q): scale_w[n] = 2^(w_exp[n] - 127), a
                                 //     power of two, applied by the MFMA as the weight rows' block scale (no multiply in the epilogue)
  const float* out_inv_scale;    // [N] EPI_STORE_FP8: out8[m][n] = e4m3(value * out_inv_scale[n])  (static per-column scale)
  // fp8 path, block-exponent rows (the residual stream of the fp8 tower): an e4m3 row carries one E8M0 exponent byte per 256
  // columns, x[m][k] ~ a8[m][k] * 2^(exp[m][k / 256] - 127), applied by the scaled MFMA itself.
  //   consumer (a_exp != NULL, EPI_STORE_BF16 / EPI_STORE_FP8; K <= 1024): the LayerNorm of the rows is folded in,
  //     out = row_r[m] * scale_w[n] * acc + row_d[m] * colsum[n] + bias[n]     (row_r = rstd, row_d = -mean * rstd)
  //   producer (EPI_RESID_Q; N <= 1024): besides the bf16 residual rows it writes their e4m3 copy, exponents and row statistics
  const unsigned char* a_exp;    // [M][ld_aexp] bytes; the dword at row m holds the exponents of column blocks 0..3
  int ld_aexp;                   // bytes between rows of a_exp (multiple of 4)
  const float* row_r;            // [M * ld_row]
  const float* row_d;            // [M * ld_row]
  int ld_row;                    // floats between rows of row_r / row_d
  void* out8; int ld8;           // EPI_RESID_Q: e4m3 copy of the new rows [M][ld8] bytes
  unsigned char* out_exp;        //              exponent byte of (row m, block n0 / 256) at out_exp[m * ld_oexp + n0 / 256]
  int ld_oexp;
                                 //              stats_out [N / 256][stats_ld][2]: (sum, sum of squares) of the stored bf16 row over 256 columns (the tile's width)
#ifdef CLIPENC_DIAG                // diagnostic build only (make diag -> libclipenc_hip_diag.so, used by tools/): the product
  unsigned long long* dbg;       // kernels carry no stamp hooks.  Optional [tiles][8] timing stamps.
#endif
};

hipError_t ce_gemm_nt(const GemmParams& p, int dtype, int epi, hipStream_t stream);
hipError_t ce_gemm_fp8(const GemmParams& p, int epi, hipStream_t stream);           // fp8 e4m3 operands; EPI_STORE_BF16 / EPI_RESID / EPI_STORE_FP8 / EPI_RESID_Q
hipError_t ce_gemm_nt_persist(const GemmParams& p, int epi, hipStream_t stream);   // bf16; EPI_STORE_BF16 / LNFOLD / RESID
hipError_t ce_gemm_tri_persist(const GemmParams& p, hipStream_t stream);            // f16 E.E^T, upper triangle, EPI_THRESH (gemm_tri.hip)
hipError_t ce_gemm_fp8_tri(const GemmParams& p, hipStream_t stream);                // e4m3 Q.Q^T, upper triangle, candidate screen (gemm_fp8_tri.hip)
hipError_t ce_tri_tile_list(int tt, int grid, const unsigned** dev_list);           // gemm_tri.hip: the device copy of tri_tile_order(tt, grid), cached
#include <vector>
std::vector<unsigned> tri_tile_order(int tt, int grid);                            // gemm_tri.hip: execution order of the triangular tile list
