/* C ABI of the MI355X-native CLIP-embed / label-score library (libclipenc_hip.so).
 *
 * Drop-in boundary for the hot path of aiXander/CLIP_assisted_data_labeling.  The reference is pure
 * Python with no FFI of its own; each entry point below names the reference call it replaces, and
 * INTEGRATION.md shows the ctypes stub a maintainer would put behind that call.
 *
 * Conventions: every function returns 0 on success or a non-zero status (clipenc_last_error()
 * gives the message for the calling thread); nothing aborts the process.  `*_dev` pointers are
 * device (HBM) addresses owned by the CALLER (e.g. torch tensors); the library owns only the weights
 * and workspace inside its handles.  `stream` is a hipStream_t (NULL = default stream); calls are
 * asynchronous on it and never synchronise the device: all device memory a handle needs is allocated by its
 * create / set_chunk / set_precision call, never by an encode / forward call.  One caller thread per handle at a time.
 */
#ifndef CLIPENC_H
#define CLIPENC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct clipenc_s* clipenc_t;   /* ViT image tower: weights + workspace on one device */
typedef struct fcreg_s* fcreg_t;       /* SimpleFC regressor weights on one device */
typedef struct preproc_s* preproc_t;   /* scratch of the crop / resize front end on one device */

#define CLIPENC_ACT_QUICK_GELU 0       /* "<arch>/openai" checkpoints */
#define CLIPENC_ACT_GELU_ERF 1

#define CLIPENC_IN_F32 0               /* crops as float32 NCHW (what _1_embed_with_CLIP.py:114-115 hands over) */
#define CLIPENC_IN_F16 1               /* crops as float16 NCHW (the reference's cuda path, utils/embedder.py:96-97) */
#define CLIPENC_IN_U8 2                /* raw uint8 NCHW pixels after Resize+CenterCrop: ToTensor + Normalize run on the device */

typedef struct clipenc_config {
  int image_size, patch, width, layers, heads, mlp_dim, embed_dim;
  int act;                             /* CLIPENC_ACT_* */
  float ln_eps;
} clipenc_config;

/* Host float32 tensors in the OpenAI / open_clip `visual.` state-dict layout.
 * Per-layer members are arrays of `layers` pointers.  Everything is copied/converted at create time. */
typedef struct clipenc_weights {
  const float* conv1_weight;           /* [width][3][patch][patch], no bias */
  const float* class_embedding;        /* [width] */
  const float* positional_embedding;   /* [tokens][width] */
  const float* ln_pre_w;  const float* ln_pre_b;
  const float* const* ln_1_w;          const float* const* ln_1_b;
  const float* const* in_proj_w;       const float* const* in_proj_b;    /* [3*width][width], [3*width] : q|k|v */
  const float* const* out_proj_w;      const float* const* out_proj_b;   /* [width][width] */
  const float* const* ln_2_w;          const float* const* ln_2_b;
  const float* const* c_fc_w;          const float* const* c_fc_b;       /* [mlp][width] */
  const float* const* c_proj_w;        const float* const* c_proj_b;     /* [width][mlp] */
  const float* ln_post_w; const float* ln_post_b;
  const float* proj;                   /* [width][embed_dim], used as x @ proj */
} clipenc_weights;

const char* clipenc_last_error(void);
int clipenc_device_count(int* count);

/* Replaces the model construction of CLIP_Encoder.__init__
 * (/root/reference/utils/embedder.py:66-74: open_clip.create_model_and_transforms + .to(device).eval()).
 * Shapes: width % heads == 0 with heads of at most 128 columns, at most 640 tokens (288 when the heads are not 64 wide),
 * embed_dim <= 1280.  The kernels are built for heads of 64 / 80 / 96 / 112 / 128 columns and widths / MLP widths that are
 * multiples of 256; any other tower (ViT-g-14: 16 heads of 88; ViT-bigG-14: 16 heads of 104) is run as the next such shape with zero weights in the added places and
 * LayerNorms over its true width -- the same arithmetic; at most 2048 columns after padding.  Weights are always handed over
 * and token rows (clipenc_forward_tokens) returned at the tower's OWN widths. */
int clipenc_create(const clipenc_config* cfg, const clipenc_weights* weights, int device, clipenc_t* out);
int clipenc_destroy(clipenc_t enc);

/* Mean / std of the Normalize step applied to CLIPENC_IN_U8 input (defaults: the OpenAI CLIP constants). */
int clipenc_set_pixel_norm(clipenc_t enc, const float* mean3, const float* std3);

/* Arithmetic of the four per-block GEMMs (QKV, attention out, FC1, FC2) — BASELINE.json configs[1] vs configs[3].
 * BF16 (default): bf16 MFMA, LayerNorm folded into the GEMM epilogue.
 * FP8: OCP e4m3 operands on the block-scaled MFMA (unit hardware scales) with per-token activation scales and
 *      per-output-channel weight scales applied in the epilogue, fp32 accumulation; attention, LayerNorm statistics,
 *      residual stream, patch embedding and head stay as in BF16.  The first switch quantises the handle's weights on
 *      the device.  open_clip has no such mode (the reference runs fp32/fp16 autocast, utils/embedder.py:94-97); the
 *      measured deviation from the fp32 oracle is stated in DESIGN.md and bounded by tests/test_gpu_fp8.py. */
#define CLIPENC_PREC_BF16 0
#define CLIPENC_PREC_FP8 1
int clipenc_set_precision(clipenc_t enc, int precision);

/* Crops pushed through the layer chain per pass (default 2048).  The workspace for one pass is (re)allocated HERE and in
 * clipenc_create / clipenc_set_precision -- these three calls may synchronise the device -- so that clipenc_encode never does. */
int clipenc_set_chunk(clipenc_t enc, int chunk_crops);
int clipenc_get_info(clipenc_t enc, int* tokens, int* embed_dim, int* chunk_crops, size_t* workspace_bytes);

/* Replaces CLIP_Encoder.encode_image (/root/reference/utils/embedder.py:94-100):
 *   crops_dev  [n_crops][3][R][R], contiguous NCHW, dtype `in_dtype`, row = image*4 + crop
 *   emb_dev    float32 [n_crops][embed_dim]; L2-normalised rows when `normalize` != 0 (:99). */
int clipenc_encode(clipenc_t enc, const void* crops_dev, int n_crops, int in_dtype, float* emb_dev,
                   int normalize, void* stream);

/* Token-level features: the residual stream [n_crops][tokens][width] (bf16) after `n_layers` transformer blocks
 * (0 = after ln_pre, cfg.layers = what ln_post + proj pool in clipenc_encode) -- open_clip's "output tokens" of the same
 * tower (/root/reference/utils/embedder.py:98 only ever takes the pooled output).  n_crops <= chunk_crops. */
int clipenc_forward_tokens(clipenc_t enc, const void* crops_dev, int n_crops, int in_dtype, int n_layers,
                           void* x_out_bf16_dev, void* stream);

/* Replaces `torch.load(model_file)` of a pickled SimpleFC as far as its arithmetic goes
 * (/root/reference/_5_predict_labels.py:107, utils/embedder.py:290; layer list utils/nn_model.py:21-33).
 * sizes[0..n_layers] = in, hidden..., out; W[l] host float32 [sizes[l+1]][sizes[l]] (nn.Linear layout). */
int fcreg_create(int n_layers, const int* sizes, const float* const* W, const float* const* b,
                 float negative_slope, int device, fcreg_t* out);
int fcreg_destroy(fcreg_t reg);

/* Replaces `model(features.float())` (/root/reference/_5_predict_labels.py:135, utils/nn_model.py:38-41).
 * Row i of the input is the concatenation of n_seg segments of seg_len floats found at
 * x_dev + i*row_stride + seg_off[s] (n_seg*seg_len == sizes[0]); a plain [n_rows][in] matrix is
 * n_seg = 1, seg_off = {0}, row_stride = in.   y_dev: float32 [n_rows][sizes[n_layers]].
 * Reproducibility: a call is bitwise deterministic, but the score BITS of a row depend on how many rows the call scores: fewer
 * than 4096 rows run a kernel that sums each neuron's products in input order, 4096 or more the store-scale kernel on the fp32
 * matrix pipe with another (fixed) summation order.  Both are fp32 throughout and agree within 2e-6 absolute on sigmoid outputs
 * (tests/test_gpu_fcreg_store.py asserts it); a store scored in one pass and re-scored in small batches matches to that
 * tolerance, not bit for bit.  The north_star tolerance against the reference is 1e-4. */
int fcreg_forward(fcreg_t reg, const float* x_dev, int n_rows, long row_stride, int n_seg, int seg_len,
                  const int* seg_off, float* y_dev, void* stream);

/* The fused composition AestheticRegressor.predict_score intends
 * (/root/reference/utils/embedder.py:298-311; _5_predict_labels.py:79 for the crop selection):
 * encode n_images*crops_per_image crops, then score each image on the embeddings of the crops
 * listed in crop_select (crop-major, model.crop_names order) without leaving the device.
 *   emb_dev    float32 [n_images][crops_per_image][embed_dim]  (normalised)
 *   score_dev  float32 [n_images][out] */
int clipenc_encode_score(clipenc_t enc, fcreg_t reg, const void* crops_dev, int n_images, int crops_per_image,
                         int in_dtype, const int* crop_select, int n_select, float* emb_dev, float* score_dev,
                         void* stream);

/* Regressor training on the device.  Replaces the optimisation loop of
 * /root/reference/_4_train_model.py:199-207 for the SimpleFC of utils/nn_model.py (created with the initial parameters,
 * same layout as fcreg_create; the last layer must have one output): per batch forward (LeakyReLU + Dropout hidden
 * layers, Sigmoid output), MSELoss, backward and one torch.optim.Adam step (betas 0.9 / 0.999, eps 1e-8, weight_decay
 * added to the gradient).
 *   x_dev / labels_dev   float32 [n][sizes[0]] / [n], the whole training set in HBM
 *   order_dev            int64 [n_order] row indices (one epoch's shuffled order), or NULL for rows 0..n_order-1
 *   lr                   the epoch's learning rate (CosineAnnealingWarmRestarts is evaluated by the caller, :210)
 *   dropout_prob, seed   masks are a counter-based hash of (seed, optimisation step, layer, row, column) -- reproducible,
 *                        but not torch's generator
 *   batch_losses_dev     float32 [ceil(n_order / batch_size)] out: each batch's MSE before its update (may be NULL)
 * The handle counts optimisation steps across calls (Adam's bias correction).  fctrain_predict is the eval-mode forward. */
typedef struct fctrain_s* fctrain_t;
int fctrain_create(int n_layers, const int* sizes, const float* const* W, const float* const* b, float negative_slope,
                   int device, fctrain_t* out);
int fctrain_destroy(fctrain_t t);
int fctrain_epoch(fctrain_t t, const float* x_dev, const float* labels_dev, const long long* order_dev, long n_order,
                  int batch_size, float lr, float weight_decay, float dropout_prob, unsigned seed, float* batch_losses_dev,
                  void* stream);
int fctrain_predict(fctrain_t t, const float* x_dev, long n, float* y_dev, void* stream);
int fctrain_get_params(fctrain_t t, int layer, float* W_host, float* b_host);

/* Replaces the similarity search of find_near_duplicates (/root/reference/_2_remove_duplicates.py:63-80):
 * rows of emb_f16_dev [n][d] (float16, as :38 casts them) are normalised (:67) and every pair i < j with
 * cosine > threshold (:69-74) is appended, unordered, to pairs_dev (int64 [capacity][2]) / vals_dev
 * (float32 [capacity]); *count_dev (uint64, zeroed by the call) receives the number found, which may
 * exceed capacity (only `capacity` are stored).  ehat_ws_dev: scratch of n_pad*d_pad float16 where
 * n_pad = n rounded up to 256 and d_pad = d rounded up to 128.  If fp16_compare != 0 the similarity is rounded to float16 before the
 * comparison, as the reference's float16 matmul output is. */
int dedup_find_pairs(const void* emb_f16_dev, int n, int d, float threshold, int fp16_compare,
                     void* ehat_ws_dev, long long* pairs_dev, float* vals_dev, unsigned long long capacity,
                     unsigned long long* count_dev, void* stream);
/* The same search -- the same pairs with the same values, in whatever order -- at about twice the rate: an e4m3 MFMA pass over the
 * upper triangle first SCREENS the pairs (unit rows quantised as e4m3(256 x); a pair stays a candidate unless its e4m3 product plus
 * the two rows' own measured quantisation-error norms, a Cauchy-Schwarz bound, is below what the exact rule could accept), then
 * only the candidates get their exact float16-MFMA value, accumulated in the order of dedup_find_pairs, and its rule.  No
 * reported pair can be lost to the screen; when the screen finds more candidates than `candidate_capacity` (a store of near-identical
 * rows) the exact search of dedup_find_pairs runs instead, decided on the device without a host round trip.
 *   screen_ws_dev  256-byte aligned scratch of dedup_screen_ws_bytes(n, d, candidate_capacity) bytes (8 bytes per candidate slot
 *                  + 1 byte per padded element + 4 per row); the other arguments as for dedup_find_pairs */
size_t dedup_screen_ws_bytes(int n, int d, unsigned long long candidate_capacity);
int dedup_find_pairs_screened(const void* emb_f16_dev, int n, int d, float threshold, int fp16_compare, void* ehat_ws_dev,
                              void* screen_ws_dev, size_t screen_ws_bytes, unsigned long long candidate_capacity, long long* pairs_dev,
                              float* vals_dev, unsigned long long capacity, unsigned long long* count_dev, void* stream);
/* Host-only view of HOW dedup_find_pairs walks the similarity matrix (no device work; the CPU tests check it): the
 * execution order of the 256 x 256 tiles of the upper triangle (tn >= tm) of a tiles_per_side x tiles_per_side tile grid for a
 * launch of `grid` persistent workgroups.  order_out[i] = tm | tn << 16 is the tile that workgroup i % grid runs in its round
 * i / grid; `capacity` must be >= tiles_per_side (tiles_per_side + 1) / 2 entries, tiles_per_side <= 65535.  The tiles that are
 * resident on one XCD at a time (workgroups b, b + 8, ... of a round) form 8 x 4 blocks of the grid: 12 operand panels per
 * 32 tiles instead of the 33 of a row-major walk. */
int dedup_tile_order(int tiles_per_side, int grid, unsigned* order_out, long capacity);

/* Replaces the per-file loop of find_similar_imgs (/root/reference/tools/find_similar_imgs.py:96-137): the distance of
 * every stored embedding row to one query (the mean context embedding, :62) and the top_n closest.
 *   emb_dev     [n] rows of d elements, float32 (emb_f16 == 0) or float16, consecutive rows row_stride elements apart
 *               (pass the [n][crops][E] store with row_stride = crops*E and the pointer advanced to the crop)
 *   query_dev   float32 [d]
 *   measure     SIMSEARCH_L2: || q - e + 1e-6 ||_2 (torch pairwise_distance, :92); SIMSEARCH_COSINE: (1 - cos)/2 with the
 *               1e-8 clamp of torch cosine_similarity on each norm (:90)
 *   dist_dev    float32 [n] out
 * simsearch_topn: indices / values of the top_n smallest distances, ascending, equal distances by lower index (the
 * reference keeps the earlier file on ties, :83), NaN = +inf; when n < top_n the tail is (-1, +inf).
 * ws_dev: scratch of simsearch_topn_workspace(n, top_n) bytes. */
#define SIMSEARCH_L2 0
#define SIMSEARCH_COSINE 1
int simsearch_distances(const void* emb_dev, int emb_f16, long n, int d, long row_stride, const float* query_dev,
                        int measure, float* dist_dev, void* stream);
size_t simsearch_topn_workspace(long n, int top_n);
int simsearch_topn(const float* dist_dev, long n, int top_n, long long* idx_out_dev, float* val_out_dev, void* ws_dev,
                   size_t ws_bytes, void* stream);

/* Replaces the function diversity_ordered_image_files of /root/reference/_3_label_images.py:135-177, the "diversity" ordering of a fresh
 * labeling session: starting from image `first`, `steps` times take the step's `sample_size` sampled candidates, and append
 * the one whose largest cosine similarity to the images chosen so far is smallest (torch.argmin: the first minimum).
 *   emb_dev      float32 [n] rows of d elements, row_stride elements apart (one crop of the packed [n][crops][E] store)
 *   samples_dev  int32 [steps][sample_size] candidate indices (the host draws them as the reference does, random.sample)
 *   order_dev    int32 [steps] out: the index appended at each step
 *   ws_dev       scratch of diversity_workspace(n) bytes
 * Asynchronous on `stream`; the walk's state (running maximum per image, last chosen index) never leaves the device. */
size_t diversity_workspace(long n);
int diversity_order(const float* emb_dev, long n, int d, long row_stride, int first, const int* samples_dev, int steps,
                    int sample_size, int* order_dev, void* ws_dev, size_t ws_bytes, void* stream);


/* GPU front end.  Replaces, for a decoded image that is already in HBM, the crop
 * extraction and the Resize + CenterCrop of the validation transform that the reference runs in its DataLoader
 * workers (/root/reference/utils/embedder.py:184-251 and :90-92; Pillow's bicubic resampler underneath):
 *   image_dev  uint8 RGB, HWC, `pitch_bytes` per row
 *   boxes      host int[n_crops][5] = {kind, a, b, c, d}: kind 0 = crop box (left, top, right, bottom);
 *              kind 1 = black square canvas (side, paste_x, paste_y, -) with the image pasted on it
 *   out_dev    uint8 [n_crops][3][out_size][out_size], bit-exact with the Pillow path; feed it to
 *              clipenc_encode as CLIPENC_IN_U8.
 * n_crops <= 32.  Asynchronous: three launches on `stream` (coefficient tables, horizontal pass, vertical pass), no host
 * synchronisation; a handle's scratch is reused by its next call, so use one stream at a time per handle. */
int preproc_create(int device, preproc_t* out);
int preproc_destroy(preproc_t p);
int preproc_crops_u8(preproc_t p, const uint8_t* image_dev, int height, int width, int pitch_bytes, int n_crops,
                     const int* boxes, int out_size, uint8_t* out_dev, void* stream);
/* The same for a whole batch in three launches (the per-image call costs three launches per image, which on a GPU busy
 * with the encoder's persistent kernels means three waits for free CUs per image): image i = images_dev[i]
 * (host array of device pointers) of heights[i] x widths[i], contributes crops_per_image[i] consecutive rows of `boxes`
 * and of out_dev; at most 65535 crops per call. */
int preproc_crops_u8_batch(preproc_t p, int n_images, const uint8_t* const* images_dev, const int* heights, const int* widths,
                           const int* pitches_bytes, const int* crops_per_image, const int* boxes, int out_size,
                           uint8_t* out_dev, void* stream);
/* Host-only: the fixed-point resampling tables of one axis (window start/length and 22-bit weights per output
 * coordinate out0 .. out0+n_out-1), exposed so that they can be pinned against Pillow without a GPU. */
int preproc_axis_tables(int in_size, int out_size, int out0, int n_out, int* bounds, int* kk, int kk_capacity, int* ksize);

/* Per-kernel timing of the chain behind clipenc_encode / clipenc_encode_score, taken with HIP events on
 * the caller's stream (used by bench.py for the roofline line).  While enabled, every kernel launch is
 * bracketed by two events; clipenc_profile_read synchronises them and returns, for kernel kind
 * 0 <= kind < clipenc_profile_kinds(): its name, summed duration, launch count and the ALGORITHMIC
 * FLOPs (unpadded tokens / K) of those launches. */
int clipenc_profile_enable(clipenc_t enc, int on);
int clipenc_profile_kinds(void);
int clipenc_profile_read(clipenc_t enc, int kind, const char** name, double* total_ms, long long* launches,
                         double* algorithmic_flops, int reset);

/* Clock the chip holds right now: a one-wave kernel that spins for spin_us microseconds and writes
 * out2_dev[0] = shader cycles (s_memtime), out2_dev[1] = 100 MHz ticks (s_memrealtime) elapsed; MHz = 100 * [0] / [1].
 * Launched by bench.py on a side stream beside the encoder to report the clock the board sustains under that load
 * (roofline.frac_at_sustained_clock); touches no handle and no product buffer. */
int clipenc_clock_probe(int device, unsigned long long* out2_dev, int spin_us, void* stream);

/* Matrix-pipe stream for bench.py's power ceiling: all CUs (8 waves each) issue `iters` x 16 v_mfma_f32_16x16x32_bf16
 * (fp8 = 0) or `iters` x 8 v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 = 1) per wave on operand registers read once from
 * operands_dev (32 KiB of bf16 / e4m3 bit patterns of the caller's choice); no LDS or memory traffic.  *flop_out = the
 * floating-point operations the launch performs.  Timed by the caller with events on `stream`: the rate is what the
 * board's power management grants the matrix pipes alone on this box.  sink_dev: one float, never written for finite
 * operands.  Touches no handle and no product buffer. */
int clipenc_mfma_stream_probe(int device, int fp8, const void* operands_dev, float* sink_dev, long long iters,
                              double* flop_out, void* stream);

/* ---- JPEG files -> RGB on the device: the decode step of the reference's image loader
 * (/root/reference/utils/embedder.py:167, PIL.Image.open(path).convert('RGB')), bit-identical to Pillow's libjpeg-turbo defaults
 * (integer "islow" inverse DCT, triangle-filter chroma upsampling, JFIF colour conversion).  Decodable here: baseline and
 * extended-sequential Huffman JPEG (one interleaved scan, or the components in several scans), and progressive Huffman JPEG
 * whose scans form a complete, orderly progression; 8 bits, greyscale or YCbCr with any sampling whose factors divide the largest ones (4:4:4, 4:2:2, 4:2:0, 4:4:0,
 * 4:1:1, ...), restart intervals.  Anything else gets a reason code
 * from jpegdec_plan and is left to the caller (the embed driver gives such files to Pillow).  One batch at a time per handle: plan (host only), then run. */
typedef struct jpegdec_s* jpegdec_t;
int jpegdec_create(int device, jpegdec_t* out);
int jpegdec_destroy(jpegdec_t d);
/* files[i] / sizes[i]: the file bytes in host memory (they must stay valid until jpegdec_run returns); at most 65535 files per
 * call.  status[i]: 0 = will be
 * decoded, 1..12 = why not (jpegdec_reason); widths / heights: for every file whose header could be read; rgb_offsets[i]: where
 * image i's uint8 [height][width][3] starts in an output buffer of *rgb_bytes bytes (images 256-byte aligned). */
int jpegdec_plan(jpegdec_t d, const void* const* files, const size_t* sizes, int n, int* status, int* widths, int* heights,
                 unsigned long long* rgb_offsets, unsigned long long* rgb_bytes);
/* Decodes the planned batch into rgb_dev (device memory, >= *rgb_bytes of the plan) and waits for it.  status[] (the array the
 * plan filled, or a copy): entries of decoded images become 0, or 100 + code if their entropy-coded data was invalid or short
 * (their pixels are then undefined). */
int jpegdec_run(jpegdec_t d, void* rgb_dev, int* status, void* stream);
/* Returned by jpegdec_run / jpegdec_reserve when the device scratch or the page-locked staging buffer cannot be allocated (the
 * caller decodes that batch by other means -- the embed driver: Pillow -- and goes on). */
#define CLIPENC_JPEGDEC_NO_MEMORY 77
/* Sets aside device scratch (about 4.5 bytes per decoded pixel + the files' bytes) and page-locked staging (about the files'
 * bytes) for the batches to come; never shrinks.  jpegdec_run grows them on demand, but hipFree / hipMalloc synchronise the whole
 * device -- a caller that runs other work beside the decoder (the embed driver: the encoder) reserves once, up front. */
int jpegdec_reserve(jpegdec_t d, unsigned long long scratch_bytes, unsigned long long staging_bytes);
const char* jpegdec_reason(int code);
/* Host only, no handle, thread-safe: would jpegdec_plan take this file?  Returns 0 or the reason code; *width / *height (may be
 * NULL) whenever the header could be read; *n_scans (may be NULL): 1 for a sequential file, the number of scans of a progressive
 * one.  (The embed driver's reader threads use it: files the device does not take, and by default progressive files -- whose
 * scans one lane per image walks serially, long enough to keep the encoder's persistent kernels off the CUs meanwhile -- are
 * decoded with Pillow right there, in parallel.) */
int jpegdec_probe(const void* file, size_t size, int* width, int* height, int* n_scans);

/* Operator-level entry points (used by the parity tests to pin each kernel on its own). */
#define CLIPENC_DT_BF16 0
#define CLIPENC_DT_F16 1
#define CLIPENC_EPI_STORE_F32 0
#define CLIPENC_EPI_STORE_BF16 1
/* out[M][N] = A[M][K] . W[N][K]^T (+ bias[N]); A, W 16-bit row-major; N % 256 == 0, K % 128 == 0 */
int clipenc_op_gemm_nt(const void* a_dev, const void* w_dev, int m, int n, int k, int dtype, int epi,
                       const float* bias_dev, void* out_dev, void* stream);
/* Row quantisation to e4m3: out8[r][k] = fp8(f(in[r][k]) * 448 / absmax_r), scale[r] = absmax_r / 448, f = identity or
 * (ln == 1) the LayerNorm normalisation without affine; ln == 2: f = identity and the scale rounded UP to a power of two (what the
 * block-exponent GEMMs below want of their weight scales).  in: bf16 (in_f32 == 0) or fp32 [n_rows][k]; k % 8 == 0, k <= 4096 */
int clipenc_op_quant_rows_fp8(const void* in_dev, int in_f32, int n_rows, int k, int ln, float eps, void* out8_dev,
                              float* scale_dev, void* stream);
/* out_bf16[M][N] = act((A8[M][K] . W8[N][K]^T) * scale_a[M] * scale_w[N] + bias[N]) (+ resid_bf16[M][N] when given,
 * act must be -1 then; resid may alias out);  N % 256 == 0, K % 256 == 0;  act: -1 none, 0 QuickGELU, 1 erf-GELU */
int clipenc_op_gemm_fp8(const void* a8_dev, const void* w8_dev, int m, int n, int k, const float* scale_a_dev,
                        const float* scale_w_dev, const float* bias_dev, int act, const void* resid_dev, void* out_dev,
                        void* stream);
/* Same product, stored as e4m3: out8[M][N] = fp8(act(... + bias[N]) * out_inv_scale[N]) — how FC1 hands the MLP hidden
 * activations to FC2 in CLIPENC_PREC_FP8 (static per-column scale, folded into FC2's weight columns).  scale_a_dev may be
 * NULL (= 1) here and in clipenc_op_gemm_fp8. */
int clipenc_op_gemm_fp8_q(const void* a8_dev, const void* w8_dev, int m, int n, int k, const float* scale_a_dev,
                          const float* scale_w_dev, const float* bias_dev, int act, const float* out_inv_scale_dev,
                          void* out8_dev, void* stream);
/* Block-exponent e4m3 rows, the residual stream's operand form in the fused CLIPENC_PREC_FP8 tower (width <= 1024):
 *   x[r][k] ~ e4m3(out8[r][k]) * 2^(exp[r][k / 256] - 127),   exp byte = max(ex - 7, 0) with ex the biased fp32 exponent of the
 *   block's max |x| (so |x * 2^-e| < 256), four bytes per row (exp_dev [n_rows][4], bytes past k / 256 are 0);
 * stats_dev (may be NULL) receives the row's (sum, sum of squares) [n_rows][2].  in: bf16 [n_rows][k], k = 256 .. 1024, k % 256 == 0 */
int clipenc_op_quant_block_fp8(const void* in_dev, int n_rows, int k, void* out8_dev, void* exp_dev, float* stats_dev,
                               void* stream);
/* (row_r, row_d)[r] = (rstd, -mean * rstd) of row r from `parts` partial (sum, sum of squares): stats_dev [parts][ld][2] */
int clipenc_op_row_norm_consts(const float* stats_dev, int parts, int ld, int n_rows, int width, float eps, float* row_r_dev,
                               float* row_d_dev, void* stream);
/* The LayerNorm-folded fp8 GEMM on block-exponent rows (QKV / FC1 of the fused tower):
 *   v = act(row_r[m] * scale_w[n] * (A8 (2^exp) . W8^T)[m][n] + row_d[m] * colsum[n] + bias[n])
 * stored as bf16 (out_inv_scale_dev == NULL) or as e4m3(v * out_inv_scale[n]);  N % 256 == 0, K % 256 == 0, K <= 1024.
 * scale_w must hold POWERS OF TWO (clipenc_op_quant_rows_fp8 with ln = 2): only their exponent is used -- the MFMA applies it as the
 * block scale of the weight rows, there is no multiply by scale_w in the epilogue.  (This op and the next keep the exponent bytes in
 * one scratch per process: developer / test entry points, one caller at a time.) */
int clipenc_op_gemm_fp8_lnf(const void* a8_dev, const void* exp_dev, const void* w8_dev, int m, int n, int k,
                            const float* row_r_dev, const float* row_d_dev, const float* scale_w_dev, const float* colsum_dev,
                            const float* bias_dev, int act, const float* out_inv_scale_dev, void* out_dev, void* stream);
/* The residual fp8 GEMM that also quantises what it produces (out-projection / FC2 of the fused tower):
 *   x[m][n] = bf16(x[m][n] + (A8 . W8^T)[m][n] * scale_w[n] + bias[n])   in place, and for the new rows their block-exponent
 *   copy (out8_dev [m][n], exp_dev [m][4]: the fp32 value before the bf16 rounding is what gets quantised) and
 *   stats_dev [n / 256][stats_ld][2]: (sum, sum of squares) of the stored bf16 row over each 256 columns.  N = 256 .. 1024;
 *   scale_w: powers of two, as for clipenc_op_gemm_fp8_lnf */
int clipenc_op_gemm_fp8_resid_q(const void* a8_dev, const void* w8_dev, int m, int n, int k, const float* scale_w_dev,
                                const float* bias_dev, void* x_inout_dev, void* out8_dev, void* exp_dev, float* stats_dev,
                                int stats_ld, void* stream);
/* qkv bf16 [n_crops*n_tok][3*width] -> out bf16 [n_crops*n_tok][width]; head dim 64, n_tok <= 640 */
int clipenc_op_attention(const void* qkv_dev, void* out_dev, int n_crops, int n_tok, int width, int heads,
                         void* stream);
/* Same, stored as e4m3 with a static per-channel scale: out8[t][c] = fp8(O[t][c] * out_inv_scale[c]) (CLIPENC_PREC_FP8) */
int clipenc_op_attention_q(const void* qkv_dev, void* out8_dev, int n_crops, int n_tok, int width, int heads,
                           const float* out_inv_scale_dev, void* stream);
#ifdef __cplusplus
}
#endif
#endif /* CLIPENC_H */
