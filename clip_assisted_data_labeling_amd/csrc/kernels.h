// Internal launcher interface of the non-GEMM kernels (attention.hip, elementwise.hip, fcreg.hip, dedup.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// attention.hip
// q_blocks > 0: only the first q_blocks 32-query blocks of every (crop, head) are computed and stored (the CLS-only last block)
hipError_t ce_attention(const void* qkv, void* out, int n_crops, int n_tok, int width, int heads, const float* out_inv,
                        int q_blocks, hipStream_t stream);   // out_inv != NULL: out is e4m3 [T][width] = fp8(O * out_inv[c])

// cls_attention.hip: the last block's attention for the class-token query, without K and V
size_t ce_cls_attn_scratch_elems(int n_crops, int D, int H);
bool ce_cls_attn_supported(int n_crops, int n_tok, int D, int H);   // the one shape predicate of the class-token shortcut
hipError_t ce_cls_qmask(const void* q, size_t q_stride, void* Qm, int n_crops, int D, int H, hipStream_t stream);
hipError_t ce_cls_attn(const void* x, const float* stats, int parts, int stats_ld, const void* q, size_t q_stride, const float* colsum_k,
                       const float* bias_k, const void* R, void* Zp, float* mz, int n_crops, int n_tok, int D, int H, float eps,
                       hipStream_t stream);
hipError_t ce_cls_finish(const void* Of, const float* mz, const float* colsum_v, const float* bias_v, void* out, size_t out_stride,
                         const float* out_inv, int n_crops, int D, int H, hipStream_t stream);

// elementwise.hip
hipError_t ce_patchify(const void* crops, int in_dtype, void* a_patch, int n_crops, int image, int patch, int kpad,
                       const float* mean3, const float* std3, hipStream_t stream);
hipError_t ce_embed_ln_pre(const void* patch_emb, const float* cls, const float* pos, const float* gamma,
                           const float* beta, void* x, float* stats, int n_crops, int n_tok, int width, int ln_width,
                           float eps, hipStream_t stream);   // ln_width <= width: the columns the LayerNorm is over (the rest are zero padding)
// out[part][i] = in[part][i * row_stride] for i < n (float2 statistics of every row_stride-th row), parts x n
hipError_t ce_gather_row_stats(const float* in, int in_ld, float* out, int out_ld, int parts, int n, int row_stride, hipStream_t stream);
// stats[0][i] += stats[1][i] + ... + stats[parts - 1][i] (in that order), i < n: towers wider than 1024 hand the LayerNorm-folded GEMM ONE part
hipError_t ce_combine_row_stats(float* stats, int ld, int parts, int n, hipStream_t stream);
hipError_t ce_clock_probe(unsigned long long* out2, int spin_ticks, hipStream_t stream);   // {shader cycles, 100 MHz ticks}
// n_cu workgroups x 8 waves x iters x (16 bf16 16x16x32 | 8 fp8 32x32x64) MFMAs on operands read once from a 32-KiB buffer
hipError_t ce_mfma_stream(const void* operands_32k, int fp8, float* sink, long long iters, int n_cu, hipStream_t stream);
hipError_t ce_head(const void* x, const float* gamma, const float* beta, const float* proj, float* emb, int n_crops,
                   int n_tok, int width, int ln_width, int embed, float eps, int normalize, hipStream_t stream);

// quant_fp8.hip
hipError_t ce_quant_rows_fp8(const void* in, int in_f32, size_t ld_in, void* out8, size_t ld_out, float* scale, int n_rows,
                             int K, int ln, float eps, hipStream_t stream, int pow2 = 0, int ln_k = 0);   // ln_k: LayerNorm over the first ln_k columns (0: all K)
// block-exponent rows (gemm.h): the standalone quantiser (the tower's first block; later blocks are quantised by the GEMM that
// produces them), the per-row constants of the folded LayerNorm, the column sums of dequantised fp8 weight rows
hipError_t ce_quant_block_fp8(const void* in, size_t ld_in, void* out8, size_t ld_out, void* exps, size_t ld_exp, float* stats,
                              int n_rows, int K, hipStream_t stream);
hipError_t ce_row_norm_consts(const float* stats, int parts, size_t ld, int n_rows, int width, float eps, float* row_r, float* row_d,
                              int ld_row, hipStream_t stream);
hipError_t ce_scale_exponents(const float* scale, unsigned char* exp_out, int n, unsigned* bad, hipStream_t stream);
hipError_t ce_colsum_fp8(const void* W8, const float* scale, int N, int K, float* colsum, hipStream_t stream);
hipError_t ce_static_scale(const void* W_bf16, const float* bias, int N, int K, float* s, float* inv_s, hipStream_t stream);
hipError_t ce_scale_cols(const void* W_bf16, const float* s, float* out_f32, int N, int K, hipStream_t stream);

// jpeg_decode.hip (+ jpeg_host.cpp): baseline JPEG files -> RGB uint8 on the device, bit-identical to Pillow
struct JpegDecState;
JpegDecState* ce_jpegdec_create();
void ce_jpegdec_destroy(JpegDecState* s);
void ce_jpegdec_plan(JpegDecState* s, const void* const* files, const size_t* sizes, int n, int* status, int* widths, int* heights,
                     unsigned long long* rgb_offsets, unsigned long long* rgb_bytes);
hipError_t ce_jpegdec_run(JpegDecState* s, void* rgb_dev, int* status, hipStream_t stream);
hipError_t ce_jpegdec_reserve(JpegDecState* s, size_t arena_bytes, size_t stage_bytes);

// fctrain.hip
struct FcTrainState;
FcTrainState* ce_fctrain_create(int n_layers, const int* sizes, const float* const* W, const float* const* b, float slope, hipError_t* err);
void ce_fctrain_destroy(FcTrainState* s);
hipError_t ce_fctrain_epoch(FcTrainState* s, const float* X, const float* T, const long long* order, long n_order, int batch_size,
                            float lr, float wd, float p_drop, uint32_t seed, float* losses, hipStream_t st);
hipError_t ce_fctrain_predict(FcTrainState* s, const float* X, long n, float* y, hipStream_t st);
hipError_t ce_fctrain_get_params(FcTrainState* s, int layer, float* W_host, float* b_host);

// simsearch.hip
hipError_t ce_simsearch_distances(const void* emb, int emb_f16, long n, int d, long row_stride, const float* query, int measure,
                                  float* out, hipStream_t stream);
size_t ce_topn_workspace_bytes(long n, int top_n);
hipError_t ce_topn_smallest(const float* dist, long n, int top_n, long long* out_i, float* out_v, void* ws, size_t ws_bytes,
                            hipStream_t stream);

// diversity.hip
size_t ce_diversity_workspace_bytes(long n);
hipError_t ce_diversity_order(const float* emb, long n, int d, long ld, int first, const int* samples, int steps, int sample_size,
                              int* order, void* ws, size_t ws_bytes, hipStream_t stream);

// fcreg.hip
#define CE_FC_MAX_LAYERS 8
#define CE_FC_MAX_SEG 16
#define CE_FC_MAX_WIDTH 5120            // widest layer (input included): 2 x FC_ROWS x width fp32 must fit the 160 KiB LDS
struct FcRegParams {
  int n_layers;
  int sizes[CE_FC_MAX_LAYERS + 1];
  const float* Wt[CE_FC_MAX_LAYERS];   // [in][out] (transposed nn.Linear weight)
  const float* b[CE_FC_MAX_LAYERS];
  float negative_slope;
  const float* x; long row_stride;     // input row i, segment s: x + i*row_stride + seg_off[s], seg_len floats
  int n_seg, seg_len; int seg_off[CE_FC_MAX_SEG];
  float* y;                            // [n_rows][sizes[n_layers]]
  int n_rows;
  // store-scale path (fcreg_mfma_kernel): nn.Linear-layout copies, zero-padded to whole 32 x 32 tiles:
  // Wr[0] = [tiles(sizes[1]) * 32][sizes[0]]; Wr[l >= 1] = [tiles(sizes[l+1]) * 32][tiles(sizes[l]) * 32]; NULL: not available
  const float* Wr[CE_FC_MAX_LAYERS];
};
#define CE_FC_MFMA_MIN_ROWS 4096        // below this the 4-rows-per-workgroup kernel has more workgroups to offer
hipError_t ce_fcreg_forward(const FcRegParams& p, hipStream_t stream);

// preproc.hip
#include <vector>
struct PreprocState;
PreprocState* ce_preproc_create();
void ce_preproc_destroy(PreprocState* s);
int ce_preproc_axis_tables(int in_size, int out_size, int out0, int n_out, std::vector<int>& bounds, std::vector<int>& kk);
hipError_t ce_preproc_crops_u8_batch(PreprocState* s, int n_images, const uint8_t* const* imgs, const int* Hs, const int* Ws,
                                     const int* pitches, const int* crops_per_image, const int* boxes, int R, uint8_t* out,
                                     hipStream_t stream);
hipError_t ce_preproc_crops_u8(PreprocState* s, const uint8_t* img, int H, int W, int pitch, int n_crops, const int* boxes,
                               int R, uint8_t* out, hipStream_t stream);

// dedup.hip
hipError_t ce_dedup_normalize_f16(const void* emb_f16, void* out_f16, int n, int d, int ld_out, hipStream_t stream);
hipError_t ce_dedup_pairs(const void* ehat_f16, int n, int d, int ld, float threshold, int fp16_compare,
                          long long* pairs, float* vals, unsigned long long capacity, unsigned long long* count,
                          hipStream_t stream);
hipError_t ce_dedup_normalize_quant(const void* emb_f16, void* out_f16, int n, int d, int ld, void* q8_ws, int ld8, float* margin_ws,
                                    unsigned long long* cand_count, unsigned long long cand_cap, hipStream_t stream);
hipError_t ce_dedup_pairs_screened(const void* ehat_f16, int n, int ld, float threshold, int fp16_compare, void* q8_ws, float* margin_ws,
                                   void* cand_ws, unsigned long long cand_cap, unsigned long long* cand_count, long long* pairs,
                                   float* vals, unsigned long long capacity, unsigned long long* count, hipStream_t stream);
