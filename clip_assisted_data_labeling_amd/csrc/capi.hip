// C ABI of libclipenc_hip.so (see include/clipenc.h): handle management, weight preparation
// (bf16 conversion + LayerNorm folding), workspace, and the kernel chain of the ViT tower.
#include "../../include/clipenc.h"
#ifdef CLIPENC_DIAG
#include "../../include/clipenc_diag.h"
#endif

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "gemm.h"
#include "jpeg_host.h"
#include "kernels.h"

namespace {

thread_local std::string g_err;

int fail(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return 1;
}
#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  hipError_t alloc(size_t n) {
    release();
    hipError_t e = hipMalloc(&p, n ? n : 16);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
};

template <typename F>
void parallel_for(int n, F f) {
  int nt = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  nt = std::min(nt, n);
  if (nt <= 1) { for (int i = 0; i < n; ++i) f(i); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) th.emplace_back([=] { for (int i = t; i < n; i += nt) f(i); });
  for (auto& t : th) t.join();
}

struct LayerDev {
  bf16_t *w_qkv, *w_out, *w_fc, *w_proj;                 // bf16 [N][K]
  float *cs_qkv, *b_qkv, *b_out, *cs_fc, *b_fc, *b_proj;
};
struct LayerDev8 {                                       // CLIPENC_PREC_FP8: e4m3 [N][K] + per-output-channel scale [N]
  uint8_t *w_qkv, *w_out, *w_fc, *w_proj;
  float *s_qkv, *s_out, *s_fc, *s_proj;
  uint8_t* e_all;                                        // fused tower (widths <= 1024): the scales are powers of two; their E8M0 bytes, [3 width | width | mlp_dim | width]
  const uint8_t* wexp(const float* sw, size_t D, size_t M) const {   // the exponent bytes that belong to scale pointer sw (may point into an array)
    const float* base[4] = {s_qkv, s_out, s_fc, s_proj};
    const size_t n[4] = {3 * D, D, M, D};
    size_t off = 0;
    for (int i = 0; i < 4; ++i) {
      if (sw >= base[i] && sw < base[i] + n[i]) return e_all + off + (sw - base[i]);
      off += n[i];
    }
    return nullptr;
  }
  float* is_hid;                                         // [mlp_dim] 1 / static scale of the MLP hidden columns (folded into w_proj)
  float* is_attn;                                        // [width]   1 / static scale of the attention output (folded into w_out)
  float *cs_qkv, *cs_fc;                                 // [3 width], [mlp_dim] column sums of the DEQUANTISED LN-folded rows (the folded mean term)
};

}  // namespace

// one kind per device kernel, named exactly as rocprofv3 --kernel-trace prints it (template arguments included)
enum { PK_PATCHIFY = 0, PK_GEMM_PATCH, PK_EMBED_LN_PRE, PK_GEMM_QKV, PK_ATTENTION, PK_GEMM_RESID, PK_GEMM_FC1, PK_HEAD, PK_FCREG,
       PK_SUB_OUT, PK_SUB_FC2,                // PK_SUB_*: the EPI_RESID launches split by shape
       PK_QUANT_LN, PK_QUANT, PK_GEMM8_QKV, PK_GEMM8_FC1, PK_GEMM8_RESID, PK_SUB8_OUT, PK_SUB8_FC2, PK_QUANT_BLOCK, PK_ROW_CONSTS,
       PK_CLS_ATTENTION,                      // the last block's class-token attention: qmask + r_h GEMM + cls_attn_kernel + o_h GEMM + finish
       PK_COUNT };
static const char* const kProfileNames[PK_COUNT] = {
    "patchify_kernel<float, 14>", "gemm_persist_kernel<1, -1>", "embed_ln_pre_kernel<2>", "gemm_persist_kernel<2, -1>",
    "attn_stream_kernel<9, 7, true>", "gemm_persist_kernel<3, -1>", "gemm_persist_kernel<2, 0>", "head_kernel", "fcreg_kernel",
    "shape:out_proj(gemm_persist_kernel<3, -1>)", "shape:fc2(gemm_persist_kernel<3, -1>)",
    "quant_ln16_kernel<8>", "quant_rows_kernel<unsigned short, false, 2, 4>", "gemm_fp8_kernel<0, -1, false>",
    "gemm_fp8_kernel<2, 0, false>", "gemm_fp8_kernel<1, -1, false>", "shape:out_proj(gemm_fp8_kernel<1, -1, false>)",
    "shape:fc2(gemm_fp8_kernel<1, -1, false>)", "quant_block_kernel<8>", "row_norm_consts_kernel",
    "cls_attn_kernel"};
// (template arguments: <EPI, ACT> and, fp8, <EPI, ACT, LNF>; FC1 is <2, 0> with QuickGELU and <2, 1> with erf-GELU; the fp8
//  names above are the unfused tower's (widths over 1024), clipenc_profile_read substitutes the fused tower's; the attention
//  name is the ViT-L/14 instantiation, other token counts use attn_stream_kernel<NKT, 7, false> / attn_kernel<NKT> / attn_long_stream_kernel<11>; quant_ln16_kernel<width / 128>)

struct ProfRec { int kind, sub; hipEvent_t a, b; double flops; };

struct Profiler {
  bool on = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  double ms[PK_COUNT] = {0}, flops[PK_COUNT] = {0};
  long long launches[PK_COUNT] = {0};
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  void begin(int kind, double fl, hipStream_t st, int sub = -1) {
    if (!on) return;
    ProfRec r{kind, sub, get(), get(), fl};
    (void)hipEventRecord(r.a, st);
    recs.push_back(r);
  }
  void end(hipStream_t st) {
    if (!on) return;
    (void)hipEventRecord(recs.back().b, st);
  }
  void collect() {
    for (auto& r : recs) {
      (void)hipEventSynchronize(r.b);
      float t = 0.f;
      if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
        ms[r.kind] += t; flops[r.kind] += r.flops; launches[r.kind]++;
        if (r.sub >= 0) { ms[r.sub] += t; flops[r.sub] += r.flops; launches[r.sub]++; }
      }
      pool.push_back(r.a); pool.push_back(r.b);
    }
    recs.clear();
  }
  void release() {
    collect();
    for (auto e : pool) (void)hipEventDestroy(e);
    pool.clear();
  }
};

constexpr int CE_TICKET_WORDS = 2048;                   // 256 launches x 8 words per pass (a 24-block tower issues 146)
struct clipenc_s {
  Profiler prof;
  clipenc_config cfg;                                    // the tower as it runs on the device: width, heads, mlp_dim after zero padding (clipenc_create)
  clipenc_config user;                                   // the tower as the caller described it
  int ln_width = 0;                                      // = user.width: the columns a LayerNorm is over
  bool padded = false;
  int device = 0;
  int tokens = 0, kpad = 0;
  float pix_mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};      // OpenAI CLIP constants (utils/embedder.py:121-124)
  float pix_std[3] = {0.26862954f, 0.26130258f, 0.27577711f};
  DevBuf weights;                                        // one slab
  bf16_t* w_conv = nullptr;                              // [width][kpad]
  float *cls = nullptr, *pos = nullptr, *ln_pre_w = nullptr, *ln_pre_b = nullptr;
  float *ln_post_w = nullptr, *ln_post_b = nullptr, *proj = nullptr;
  std::vector<LayerDev> layers;
  int precision = CLIPENC_PREC_BF16;
  bool cls_only_last = true;                             // last block on the class-token rows only (the diagnostic build reads CLIPENC_FULL_LAST_BLOCK=1 to disable it)
  DevBuf weights8;                                       // fp8 copies of the block weights (made by clipenc_set_precision)
  std::vector<LayerDev8> layers8;
  uint8_t* a8 = nullptr;                                 // workspace: quantised GEMM operand [T][width] (the e4m3 MLP hidden
                                                         // [T][mlp_dim] lives in the bf16 `hid` buffer)
  float* sa8 = nullptr;                                  //            its per-token scales [T]
  // fused fp8 tower (width <= 1024): the residual stream's e4m3 block-exponent copy, written by the GEMM that produces the rows
  uint8_t* x8 = nullptr;                                 // [T][width]
  uint8_t* xe8 = nullptr;                                // [T][4] exponent bytes
  float* st8 = nullptr;                                  // [width / 256][Tp][2] partial row statistics (one pair per row and 256 columns)
  float *rr8 = nullptr, *rd8 = nullptr;                  // [Tp] rstd, -mean * rstd
  int ws_precision = -1;
  bool fp8_unfused = false;                              // diagnostic build: CLIPENC_FP8_UNFUSED=1 runs the separate LayerNorm-quantise pass at every width
  // workspace (sized for `chunk` crops)
  int chunk = 2048, ws_chunk = 0;
  DevBuf ws;
  bf16_t *a_patch = nullptr, *pe = nullptr, *x = nullptr, *qkv = nullptr, *attn = nullptr, *hid = nullptr;
  float *stats0 = nullptr, *stats_a = nullptr, *stats_b = nullptr, *stats_c = nullptr;   // stats_c: [parts][chunk] of the CLS rows
  bf16_t* w_k_t = nullptr;                               // last layer: (gamma-folded W_k)^T [D (k)][D (n)], the operand of the r_h GEMM (cls_attention.hip)
  bool cls_shortcut = true;                              // last block's attention without K and V (the diagnostic build reads CLIPENC_CLS_KV=1 to switch it off)
  unsigned* tickets = nullptr;                           // [CE_TICKET_WORDS] zeroed per pass: eight ticket words per persistent GEMM launch (gemm.h)
  bool dynamic_tail = true;                              // (the diagnostic build reads CLIPENC_STATIC_TILES=1 to switch the tickets off)
};

struct preproc_s {
  int device = 0;
  PreprocState* st = nullptr;
};

struct jpegdec_s {
  int device = 0;
  JpegDecState* st = nullptr;
};

struct fctrain_s {
  int device = 0;
  FcTrainState* st = nullptr;
};

struct fcreg_s {
  int device = 0;
  int n_layers = 0;
  int sizes[CE_FC_MAX_LAYERS + 1];
  float negative_slope = 0.01f;
  DevBuf slab;
  const float* Wt[CE_FC_MAX_LAYERS];
  const float* b[CE_FC_MAX_LAYERS];
  const float* Wr[CE_FC_MAX_LAYERS];                     // zero-padded nn.Linear-layout copies for the store-scale kernel
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// The fp8 tower quantises the residual stream inside the GEMM that produces it (block-exponent rows, gemm.h) when a row's
// exponents fit one dword: width a multiple of 256, at most 1024.  Wider towers run the separate LayerNorm-quantise pass.
bool fp8_fused(const clipenc_config& g) { return g.width % 256 == 0 && g.width <= 1024; }

// (Re)allocates the workspace of one pass of `chunk` crops at the handle's precision.  Called by clipenc_create,
// clipenc_set_chunk and clipenc_set_precision only (hipMalloc / hipFree synchronise the device): the encode calls never
// allocate.
int ensure_workspace(clipenc_s* e) {
  const int c = e->chunk;
  if (c == e->ws_chunk && e->ws_precision == e->precision) return 0;
  const clipenc_config& g = e->cfg;
  const size_t T = (size_t)c * e->tokens, P = (size_t)c * (e->tokens - 1);
  const size_t parts = g.width / 256;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  const size_t o_ap = take(P * e->kpad * 2), o_pe = take(P * g.width * 2), o_x = take(T * g.width * 2);
  const size_t o_qkv = take(T * 3 * g.width * 2), o_at = take(T * g.width * 2), o_h = take(T * g.mlp_dim * 2);
  const size_t Tp = align_up(T, 256);
  const size_t o_s0 = take(Tp * 8), o_sa = take(parts * Tp * 8), o_sb = take(parts * Tp * 8);
  const size_t o_sc = take(parts * align_up((size_t)c, 256) * 8);
  const size_t o_tk = take(CE_TICKET_WORDS * sizeof(unsigned));
  const bool f8 = e->precision == CLIPENC_PREC_FP8;
  const size_t o_a8 = f8 ? take(T * (size_t)g.width) : 0, o_sa8 = f8 ? take(T * 4) : 0;
  const bool f8f = f8 && fp8_fused(g);
  const size_t o_x8 = f8f ? take(T * (size_t)g.width) : 0, o_xe8 = f8f ? take(Tp * 4) : 0;
  const size_t o_st8 = f8f ? take((size_t)(g.width / 256) * Tp * 8) : 0, o_rr8 = f8f ? take(Tp * 4) : 0, o_rd8 = f8f ? take(Tp * 4) : 0;
  HIP_TRY(hipSetDevice(e->device));
  // allocate the new slab FIRST and swap it in on success: on failure the handle keeps its old, still valid workspace
  // (and the chunk / precision it was sized for), so no pointer ever refers to freed memory
  DevBuf fresh;
  if (hipError_t err = fresh.alloc(off); err != hipSuccess)
    return fail("workspace of %zu bytes for a chunk of %d crops: %s", off, c, hipGetErrorString(err));
  e->ws.release();
  e->ws = fresh;
  char* b = (char*)e->ws.p;
  e->a_patch = (bf16_t*)(b + o_ap); e->pe = (bf16_t*)(b + o_pe); e->x = (bf16_t*)(b + o_x);
  e->qkv = (bf16_t*)(b + o_qkv); e->attn = (bf16_t*)(b + o_at); e->hid = (bf16_t*)(b + o_h);
  e->stats0 = (float*)(b + o_s0); e->stats_a = (float*)(b + o_sa); e->stats_b = (float*)(b + o_sb); e->stats_c = (float*)(b + o_sc);
  e->tickets = (unsigned*)(b + o_tk);
  e->a8 = f8 ? (uint8_t*)(b + o_a8) : nullptr; e->sa8 = f8 ? (float*)(b + o_sa8) : nullptr;
  e->x8 = f8f ? (uint8_t*)(b + o_x8) : nullptr; e->xe8 = f8f ? (uint8_t*)(b + o_xe8) : nullptr;
  e->st8 = f8f ? (float*)(b + o_st8) : nullptr; e->rr8 = f8f ? (float*)(b + o_rr8) : nullptr; e->rd8 = f8f ? (float*)(b + o_rd8) : nullptr;
  e->ws_chunk = c; e->ws_precision = e->precision;
  return 0;
}

// runs patch-embed + ln_pre + `n_layers` blocks on `c` crops; leaves the residual stream in e->x
// cls_only_last: the caller only needs token 0 of the last block (clipenc_encode); false keeps every token (forward_tokens)
int run_tower(clipenc_s* e, const void* crops, int c, int in_dtype, int n_layers, hipStream_t st, bool cls_only_last = false) {
  const clipenc_config& g = e->cfg;
  const int T = c * e->tokens, P = c * (e->tokens - 1);
  const int parts = g.width / 256;
  // what a LayerNorm-folded consumer is handed: the producer's parts as they are (its LDS layout holds four), or -- wider towers -- ONE part,
  // the producer's added up by a small pass behind it (fold_stats)
  const int cparts = parts > 4 ? 1 : parts;
  const int Tp = (int)align_up((size_t)T, 256);
  Profiler& pf = e->prof;
  // profiler flops are the TOWER's (the caller's widths): a zero-padded tower's extra columns are work, not result
  const double dT = (double)T, dD = (double)e->user.width, dM = (double)e->user.mlp_dim;
  auto own = [&](int n) -> double {          // a GEMM dimension on the device -> the tower's own
    return n == g.mlp_dim ? (double)n * e->user.mlp_dim / g.mlp_dim : (double)n * e->user.width / g.width;
  };
  // ticket counters of this pass's persistent GEMM launches (at most 6 per block + the patch GEMM): zeroed by one fill
  int tk = 0;
  if (e->dynamic_tail) HIP_TRY(hipMemsetAsync(e->tickets, 0, CE_TICKET_WORDS * sizeof(unsigned), st));
  auto ticket = [&]() -> unsigned* { return (e->dynamic_tail && (tk + 1) * 8 <= CE_TICKET_WORDS) ? e->tickets + 8 * tk++ : nullptr; };
  // The LAST block's attention for the class-token query without K and V (cls_attention.hip): r_h = (Q_cls masked to head h) . W'_k
  // and o_h = z_h . W'_v^T as two small launches of the persistent GEMM (16 x redundant -- every head's row against every column --
  // and still ~65 us each), the scores / softmax / weighted row sums in between as one pass-twice-over-x kernel.  Scratch: four
  // [c H][D] bf16 buffers + [c H] floats in `hid`, which is idle until the block's FC1.  `out`: the class-token rows of attn (bf16,
  // row stride tokens * D) or of a8 (e4m3 with out_inv).  Returns false when the scratch does not fit (tiny test towers).
  const LayerDev& LL = e->layers[g.layers - 1];
  auto cls_attention = [&](const float* stats, int stats_parts_, int stats_ld_, void* out, const float* out_inv, bool* done) -> hipError_t {
    const int Dw = g.width, H = g.heads;
    const size_t n = ce_cls_attn_scratch_elems(c, Dw, H);
    *done = false;
    // one predicate for "can this shape take the shortcut" (ce_cls_attn_supported, cls_attention.hip), tested BEFORE anything is
    // launched: an unsupported shape returns done = false and the caller runs the projected K | V path
    if (!e->cls_shortcut || !ce_cls_attn_supported(c, e->tokens, Dw, H) ||
        (size_t)T * g.mlp_dim * 2 < 5 * n * 2 + (size_t)c * H * 4 + 1024) return hipSuccess;
    bf16_t *Qm = e->hid, *R = e->hid + n, *Zp = e->hid + 2 * n;
    float* Of = (float*)(e->hid + 3 * n);                       // fp32: rounded once, after the mean term is subtracted (cls_finish)
    float* mzv = (float*)(e->hid + 5 * n);
    const size_t q_stride = (size_t)e->tokens * 3 * Dw;
    pf.begin(PK_CLS_ATTENTION, 4.0 * c * (double)e->tokens * dD, st);   // its own kind: PK_ATTENTION is the streaming kernel's launches alone
    hipError_t err = ce_cls_qmask(e->qkv, q_stride, Qm, c, Dw, H, st);
    GemmParams a{};
    a.A = Qm; a.lda = Dw; a.W = e->w_k_t; a.ldw = Dw; a.M = c * H; a.N = Dw; a.K = Dw; a.out = R; a.ldo = Dw; a.ticket = ticket();
    if (err == hipSuccess) err = ce_gemm_nt(a, CE_DT_BF16, EPI_STORE_BF16, st);
    if (err == hipSuccess)
      err = ce_cls_attn(e->x, stats, stats_parts_, stats_ld_, e->qkv, q_stride, LL.cs_qkv + Dw, LL.b_qkv + Dw, R, Zp, mzv, c, e->tokens, Dw, H,
                        g.ln_eps, st);
    GemmParams b{};
    b.A = Zp; b.lda = Dw; b.W = LL.w_qkv + (size_t)2 * Dw * Dw; b.ldw = Dw; b.M = c * H; b.N = Dw; b.K = Dw; b.out = Of; b.ldo = Dw; b.ticket = ticket();
    if (err == hipSuccess) err = ce_gemm_nt(b, CE_DT_BF16, EPI_STORE_F32, st);
    if (err == hipSuccess)
      err = ce_cls_finish(Of, mzv, LL.cs_qkv + 2 * Dw, LL.b_qkv + 2 * Dw, out, (size_t)e->tokens * Dw, out_inv, c, Dw, H, st);
    pf.end(st);
    *done = err == hipSuccess;
    return err;
  };
  auto fold_stats = [&](float* stats, int ld, int n) -> hipError_t {
    return parts > 4 ? ce_combine_row_stats(stats, ld, parts, n, st) : hipSuccess;
  };
  pf.begin(PK_PATCHIFY, 0.0, st);
  HIP_TRY(ce_patchify(crops, in_dtype, e->a_patch, c, g.image_size, g.patch, e->kpad, e->pix_mean, e->pix_std, st));
  pf.end(st);
  GemmParams p{};
  p.A = e->a_patch; p.lda = e->kpad; p.W = e->w_conv; p.ldw = e->kpad; p.M = P; p.N = g.width; p.K = e->kpad;
  p.out = e->pe; p.ldo = g.width; p.bias = nullptr; p.ticket = ticket();
  pf.begin(PK_GEMM_PATCH, 2.0 * P * dD * (3.0 * g.patch * g.patch), st);
  HIP_TRY(ce_gemm_nt(p, CE_DT_BF16, EPI_STORE_BF16, st));
  pf.end(st);
  pf.begin(PK_EMBED_LN_PRE, 0.0, st);
  HIP_TRY(ce_embed_ln_pre(e->pe, e->cls, e->pos, e->ln_pre_w, e->ln_pre_b, e->x, e->stats0, c, e->tokens, g.width,
                          e->ln_width, g.ln_eps, st));
  pf.end(st);
  const float* stats_in = e->stats0;
  int stats_parts = 1;
  if (e->precision == CLIPENC_PREC_FP8 && fp8_fused(g) && !e->fp8_unfused) {
    // Fused fp8 tower.  The residual stream keeps an e4m3 block-exponent copy x8 (+ exponent dwords xe8) next to its bf16 rows:
    // the out-projection and FC2 GEMMs write it for the rows they produce (EPI_RESID_Q) together with partial row statistics,
    // and the QKV / FC1 GEMMs read it with the LayerNorm folded into their epilogue (gemm.h: row_r, row_d, colsum) -- gamma sits
    // in the fp8 weights and beta in the bias, as for bf16.  No pass over the residual stream between the GEMMs; only the
    // tower's first block is quantised by a kernel of its own, and a one-thread-per-row kernel turns the statistics into
    // (rstd, -mean * rstd) in front of each consumer.
    const int Dw = g.width, Mh = g.mlp_dim, sparts = g.width / 256;
    uint8_t* h8 = (uint8_t*)e->hid;
    auto run8 = [&](GemmParams& q, int epi, int kind, int sub) -> hipError_t {
      q.ticket = ticket();
      pf.begin(kind, 2.0 * (double)q.M * own(q.N) * own(q.K), st, sub);
      hipError_t err = ce_gemm_fp8(q, epi, st);
      pf.end(st);
      return err;
    };
    // LayerNorm-folded consumer on M rows of x8, `rstride` rows of the stream apart; row constants `ld_row` floats apart
    const LayerDev8* curQ = nullptr;                        // (the layer whose weights the lambdas below are handed)
    auto lnf = [&](int M, int rstride, const uint8_t* W8, const float* sw, const float* cs, const float* bias, int N, int act, void* out,
                   int ldo, int epi, const float* out_inv, int ld_row, int kind) -> hipError_t {
      GemmParams q{};
      q.A = e->x8; q.lda = rstride * Dw; q.W = W8; q.ldw = Dw; q.M = M; q.N = N; q.K = Dw; q.out = out; q.ldo = ldo; q.bias = bias;
      q.scale_w = sw; q.w_exp = curQ->wexp(sw, (size_t)Dw, (size_t)Mh); q.colsum = cs; q.act = act; q.out_inv_scale = out_inv;
      q.a_exp = e->xe8; q.ld_aexp = rstride * 4; q.row_r = e->rr8; q.row_d = e->rd8; q.ld_row = ld_row;
      return run8(q, epi, kind, -1);
    };
    // x += A8 . W8^T + bias on M rows (`rstride` apart); quant: also their x8 / xe8 / statistics
    auto resid = [&](const uint8_t* A8, int M, int lda, const uint8_t* W8, const float* sw, const float* bias, int K, int rstride,
                     bool quant, int stats_ld, int sub) -> hipError_t {
      GemmParams q{};
      q.A = A8; q.lda = lda; q.W = W8; q.ldw = K; q.M = M; q.N = Dw; q.K = K; q.out = e->x; q.ldo = rstride * Dw; q.bias = bias;
      q.scale_w = sw; q.w_exp = curQ->wexp(sw, (size_t)Dw, (size_t)Mh); q.act = -1; q.resid = e->x;
      q.out8 = e->x8; q.ld8 = rstride * Dw; q.out_exp = e->xe8; q.ld_oexp = rstride * 4; q.stats_out = e->st8; q.stats_ld = stats_ld;
      return run8(q, quant ? EPI_RESID_Q : EPI_RESID, PK_GEMM8_RESID, sub);
    };
    auto consts = [&](const float* stats, int parts, int ld, int n) -> hipError_t {
      pf.begin(PK_ROW_CONSTS, 0.0, st);
      hipError_t err = ce_row_norm_consts(stats, parts, (size_t)ld, n, e->ln_width, g.ln_eps, e->rr8, e->rd8, 1, st);
      pf.end(st);
      return err;
    };
    if (n_layers > 0) {
      pf.begin(PK_QUANT_BLOCK, 0.0, st);
      HIP_TRY(ce_quant_block_fp8(e->x, (size_t)Dw, e->x8, (size_t)Dw, e->xe8, 4, nullptr, T, Dw, st));
      pf.end(st);
    }
    const float* st_in = e->stats0;
    int parts_in = 1;
    for (int l = 0; l < n_layers; ++l) {
      const LayerDev& L = e->layers[l];
      const LayerDev8& Q = e->layers8[l];
      curQ = &Q;
      const bool last = l == n_layers - 1;
      HIP_TRY(consts(st_in, parts_in, Tp, T));
      if (cls_only_last && last) {
        // LAST block on the class-token rows only (see the bf16 branch below for the argument): K | V for every token, then Q,
        // attention, out-proj and the MLP on the c CLS rows, which stay where they are (row stride = tokens rows) in x, x8 and xe8
        const int stride = e->tokens;
        const int Tpc = (int)align_up((size_t)c, 256);
        HIP_TRY(lnf(c, stride, Q.w_qkv, Q.s_qkv, Q.cs_qkv, L.b_qkv, Dw, -1, e->qkv, stride * 3 * Dw, EPI_STORE_BF16, nullptr, stride,
                    PK_GEMM8_QKV));
        // the class-token attention without K and V, on the bf16 residual rows and bf16 weights (cls_attention.hip), O as e4m3
        bool short_done = false;
        HIP_TRY(cls_attention(st_in, parts_in, Tp, e->a8, Q.is_attn, &short_done));
        if (!short_done) {
          HIP_TRY(lnf(T, 1, Q.w_qkv + (size_t)Dw * Dw, Q.s_qkv + Dw, Q.cs_qkv + Dw, L.b_qkv + Dw, 2 * Dw, -1, e->qkv + Dw, 3 * Dw,
                      EPI_STORE_BF16, nullptr, 1, PK_GEMM8_QKV));
          pf.begin(PK_ATTENTION, 4.0 * c * (double)e->tokens * dD, st);
          HIP_TRY(ce_attention(e->qkv, e->a8, c, e->tokens, g.width, g.heads, Q.is_attn, 1, st));   // O of rows 0..31 of every crop, e4m3
          pf.end(st);
        }
        HIP_TRY(resid(e->a8, c, stride * Dw, Q.w_out, Q.s_out, L.b_out, Dw, stride, true, Tpc, PK_SUB8_OUT));
        HIP_TRY(consts(e->st8, sparts, Tpc, c));               // compact: constants of CLS row i at [i]
        HIP_TRY(lnf(c, stride, Q.w_fc, Q.s_fc, Q.cs_fc, L.b_fc, Mh, g.act, h8, Mh, EPI_STORE_FP8, Q.is_hid, 1, PK_GEMM8_FC1));
        HIP_TRY(resid(h8, c, Mh, Q.w_proj, Q.s_proj, L.b_proj, Mh, stride, false, Tpc, PK_SUB8_FC2));
        break;
      }
      HIP_TRY(lnf(T, 1, Q.w_qkv, Q.s_qkv, Q.cs_qkv, L.b_qkv, 3 * Dw, -1, e->qkv, 3 * Dw, EPI_STORE_BF16, nullptr, 1, PK_GEMM8_QKV));
      pf.begin(PK_ATTENTION, 4.0 * c * (double)e->tokens * e->tokens * dD, st);
      // attention writes O as e4m3 directly (static per-channel scale from the V rows of w_qkv, folded into w_out)
      HIP_TRY(ce_attention(e->qkv, e->a8, c, e->tokens, g.width, g.heads, Q.is_attn, 0, st));
      pf.end(st);
      HIP_TRY(resid(e->a8, T, Dw, Q.w_out, Q.s_out, L.b_out, Dw, 1, true, Tp, PK_SUB8_OUT));
      HIP_TRY(consts(e->st8, sparts, Tp, T));
      // FC1 writes the hidden activations as e4m3 directly (static per-column scale, folded into w_proj): no bf16 round trip
      HIP_TRY(lnf(T, 1, Q.w_fc, Q.s_fc, Q.cs_fc, L.b_fc, Mh, g.act, h8, Mh, EPI_STORE_FP8, Q.is_hid, 1, PK_GEMM8_FC1));
      HIP_TRY(resid(h8, T, Mh, Q.w_proj, Q.s_proj, L.b_proj, Mh, 1, !last, Tp, PK_SUB8_FC2));   // (nothing reads x8 after the last block)
      st_in = e->st8; parts_in = sparts;
    }
    return 0;
  }
  if (e->precision == CLIPENC_PREC_FP8) {
    // (widths over 1024)
    // per block: quantise the (normalised) GEMM operand row by row, run the e4m3 GEMM, scales + bias (+ act / residual)
    // in its epilogue.  LayerNorm: gamma sits in the fp8 weights, beta in the bias (as for bf16); the statistics are
    // computed by the quantiser itself.
    const size_t D = g.width;
    uint8_t* h8 = (uint8_t*)e->hid;
    // A8: operand rows (per-token scales sa, or NULL when its columns carry static scales folded into W8)
    // general form: M rows of A8 (row stride lda bytes), output rows ldo elements apart
    auto gemm8x = [&](const uint8_t* A8, int M, int lda, const float* sa, const uint8_t* W8, const float* sw, const float* bias, int N,
                      int K, int act, void* out, int ldo, int epi, const float* out_inv, int kind, int sub) -> hipError_t {
      GemmParams q{};
      q.A = A8; q.lda = lda; q.W = W8; q.ldw = K; q.M = M; q.N = N; q.K = K; q.out = out; q.ldo = ldo; q.bias = bias;
      q.scale_a = sa; q.scale_w = sw; q.act = act; q.resid = epi == EPI_RESID ? out : nullptr; q.out_inv_scale = out_inv;
      pf.begin(kind, 2.0 * (double)M * own(N) * own(K), st, sub);
      hipError_t err = ce_gemm_fp8(q, epi, st);
      pf.end(st);
      return err;
    };
    auto gemm8 = [&](const uint8_t* A8, const float* sa, const uint8_t* W8, const float* sw, const float* bias, int N, int K,
                     int act, void* out, int epi, const float* out_inv, int kind, int sub) -> hipError_t {
      return gemm8x(A8, T, K, sa, W8, sw, bias, N, K, act, out, N, epi, out_inv, kind, sub);
    };
    auto quant = [&](const bf16_t* in, size_t K, int ln) -> hipError_t {
      pf.begin(ln ? PK_QUANT_LN : PK_QUANT, 0.0, st);
      hipError_t err = ce_quant_rows_fp8(in, 0, K, e->a8, K, e->sa8, T, (int)K, ln, g.ln_eps, st, 0, ln ? e->ln_width : 0);
      pf.end(st);
      return err;
    };
    for (int l = 0; l < n_layers; ++l) {
      const LayerDev& L = e->layers[l];
      const LayerDev8& Q = e->layers8[l];
      if (cls_only_last && l == n_layers - 1) {
        // LAST block on the class-token rows only (see the bf16 branch below for the argument): K | V for every token, then
        // Q, attention, out-proj and the MLP on the c CLS rows (row stride = tokens rows).  The compact quantised CLS rows and
        // their scales reuse the head of a8 / sa8 once the K|V GEMM has consumed the full operand (stream order).
        const size_t Dw = D;
        const int stride = e->tokens;
        HIP_TRY(quant(e->x, D, 1));
        HIP_TRY(gemm8x(e->a8, T, (int)Dw, e->sa8, Q.w_qkv + Dw * Dw, Q.s_qkv + Dw, L.b_qkv + Dw, 2 * g.width, g.width, -1,
                       e->qkv + Dw, 3 * g.width, EPI_STORE_BF16, nullptr, PK_GEMM8_QKV, -1));
        auto quant_cls = [&]() -> hipError_t {                 // LN + quantise the CLS rows of x -> a8[0..c), sa8[0..c)
          pf.begin(PK_QUANT_LN, 0.0, st);
          hipError_t err = ce_quant_rows_fp8(e->x, 0, (size_t)stride * Dw, e->a8, Dw, e->sa8, c, (int)Dw, 1, g.ln_eps, st, 0, e->ln_width);
          pf.end(st);
          return err;
        };
        HIP_TRY(quant_cls());
        HIP_TRY(gemm8x(e->a8, c, (int)Dw, e->sa8, Q.w_qkv, Q.s_qkv, L.b_qkv, g.width, g.width, -1, e->qkv, stride * 3 * g.width,
                       EPI_STORE_BF16, nullptr, PK_GEMM8_QKV, -1));
        pf.begin(PK_ATTENTION, 4.0 * c * (double)e->tokens * dD, st);
        HIP_TRY(ce_attention(e->qkv, e->a8, c, e->tokens, g.width, g.heads, Q.is_attn, 1, st));   // O of rows 0..31 of every crop, e4m3
        pf.end(st);
        HIP_TRY(gemm8x(e->a8, c, stride * g.width, nullptr, Q.w_out, Q.s_out, L.b_out, g.width, g.width, -1, e->x, stride * g.width,
                       EPI_RESID, nullptr, PK_GEMM8_RESID, PK_SUB8_OUT));
        HIP_TRY(quant_cls());
        HIP_TRY(gemm8x(e->a8, c, (int)Dw, e->sa8, Q.w_fc, Q.s_fc, L.b_fc, g.mlp_dim, g.width, g.act, h8, g.mlp_dim, EPI_STORE_FP8,
                       Q.is_hid, PK_GEMM8_FC1, -1));
        HIP_TRY(gemm8x(h8, c, g.mlp_dim, nullptr, Q.w_proj, Q.s_proj, L.b_proj, g.width, g.mlp_dim, -1, e->x, stride * g.width,
                       EPI_RESID, nullptr, PK_GEMM8_RESID, PK_SUB8_FC2));
        break;
      }
      HIP_TRY(quant(e->x, D, 1));
      HIP_TRY(gemm8(e->a8, e->sa8, Q.w_qkv, Q.s_qkv, L.b_qkv, 3 * g.width, g.width, -1, e->qkv, EPI_STORE_BF16, nullptr, PK_GEMM8_QKV, -1));
      pf.begin(PK_ATTENTION, 4.0 * c * (double)e->tokens * e->tokens * dD, st);
      // attention writes O as e4m3 directly (static per-channel scale from the V rows of w_qkv, folded into w_out)
      HIP_TRY(ce_attention(e->qkv, e->a8, c, e->tokens, g.width, g.heads, Q.is_attn, 0, st));
      pf.end(st);
      HIP_TRY(gemm8(e->a8, nullptr, Q.w_out, Q.s_out, L.b_out, g.width, g.width, -1, e->x, EPI_RESID, nullptr, PK_GEMM8_RESID, PK_SUB8_OUT));
      HIP_TRY(quant(e->x, D, 1));
      // FC1 writes the hidden activations as e4m3 directly (static per-column scale, folded into w_proj): no bf16 round trip
      HIP_TRY(gemm8(e->a8, e->sa8, Q.w_fc, Q.s_fc, L.b_fc, g.mlp_dim, g.width, g.act, h8, EPI_STORE_FP8, Q.is_hid, PK_GEMM8_FC1, -1));
      HIP_TRY(gemm8(h8, nullptr, Q.w_proj, Q.s_proj, L.b_proj, g.width, g.mlp_dim, -1, e->x, EPI_RESID, nullptr, PK_GEMM8_RESID, PK_SUB8_FC2));
    }
    return 0;
  }
  for (int l = 0; l < n_layers; ++l) {
    const LayerDev& L = e->layers[l];
    if (cls_only_last && l == n_layers - 1) {
      // ---- LAST block, class-token rows only.  The embedding is ln_post + proj of token 0 (SURVEY.md Appendix A.2 step 5,
      // /root/reference/utils/embedder.py:98 takes the pooled output), and nothing after this block reads another token, so
      // of this block only K and V are needed for every token; Q, attention, out-proj and the MLP run on the c CLS rows
      // (row stride = tokens rows).  Same arithmetic per row as the full block -- the rows it leaves out are dead.
      const int Dw = g.width, stride = e->tokens;
      const int Tpc = (int)align_up((size_t)c, 256);
      // statistics of the CLS rows, compact (row i = crop i)
      float* stats_c = e->stats_c;
      pf.begin(PK_EMBED_LN_PRE, 0.0, st);
      HIP_TRY(ce_gather_row_stats(stats_in, Tp, stats_c, Tpc, stats_parts, c, stride, st));
      pf.end(st);
      // Q of the CLS rows -> their rows of qkv (columns 0..D)
      GemmParams qc{};
      qc.A = e->x; qc.lda = stride * Dw; qc.W = L.w_qkv; qc.ldw = Dw; qc.M = c; qc.N = Dw; qc.K = Dw;
      qc.out = e->qkv; qc.ldo = stride * 3 * Dw; qc.bias = L.b_qkv; qc.colsum = L.cs_qkv;
      qc.stats_in = stats_c; qc.stats_in_parts = stats_parts; qc.stats_ld = Tpc; qc.inv_width = 1.0f / e->ln_width; qc.eps = g.ln_eps; qc.act = -1;
      qc.ticket = ticket();
      pf.begin(PK_GEMM_QKV, 2.0 * c * dD * dD, st);
      HIP_TRY(ce_gemm_nt(qc, CE_DT_BF16, EPI_LNFOLD, st));
      pf.end(st);
      // the class-token attention: without K and V (cls_attention.hip) -- or, when its scratch does not fit, the standard way
      bool short_done = false;
      HIP_TRY(cls_attention(stats_in, stats_parts, Tp, e->attn, nullptr, &short_done));
      if (!short_done) {
        // K | V = LN1(x) . W[D:3D]^T + b, every token  -> columns D..3D of qkv
        GemmParams kv{};
        kv.A = e->x; kv.lda = Dw; kv.W = L.w_qkv + (size_t)Dw * Dw; kv.ldw = Dw; kv.M = T; kv.N = 2 * Dw; kv.K = Dw;
        kv.out = e->qkv + Dw; kv.ldo = 3 * Dw; kv.bias = L.b_qkv + Dw; kv.colsum = L.cs_qkv + Dw;
        kv.stats_in = stats_in; kv.stats_in_parts = stats_parts; kv.stats_ld = Tp; kv.inv_width = 1.0f / e->ln_width; kv.eps = g.ln_eps; kv.act = -1;
        kv.ticket = ticket();
        pf.begin(PK_GEMM_QKV, 2.0 * dT * 2.0 * dD * dD, st);
        HIP_TRY(ce_gemm_nt(kv, CE_DT_BF16, EPI_LNFOLD, st));
        pf.end(st);
        // attention of the first 32-query block of every (crop, head); only its row 0 is a real query here
        pf.begin(PK_ATTENTION, 4.0 * c * (double)e->tokens * dD, st);
        HIP_TRY(ce_attention(e->qkv, e->attn, c, e->tokens, Dw, g.heads, nullptr, 1, st));
        pf.end(st);
      }
      // x[cls] += attn[cls] . Wo^T + bo
      GemmParams o{};
      o.A = e->attn; o.lda = stride * Dw; o.W = L.w_out; o.ldw = Dw; o.M = c; o.N = Dw; o.K = Dw;
      o.out = e->x; o.ldo = stride * Dw; o.bias = L.b_out; o.resid = e->x; o.stats_out = e->stats_a; o.stats_ld = Tpc;
      o.ticket = ticket();
      pf.begin(PK_GEMM_RESID, 2.0 * c * dD * dD, st, PK_SUB_OUT);
      HIP_TRY(ce_gemm_nt(o, CE_DT_BF16, EPI_RESID, st));
      HIP_TRY(fold_stats(e->stats_a, Tpc, c));
      pf.end(st);
      // h[cls] = act(LN2(x[cls]) . Wfc^T + b)   (compact [c][mlp])
      GemmParams f{};
      f.A = e->x; f.lda = stride * Dw; f.W = L.w_fc; f.ldw = Dw; f.M = c; f.N = g.mlp_dim; f.K = Dw;
      f.out = e->hid; f.ldo = g.mlp_dim; f.bias = L.b_fc; f.colsum = L.cs_fc;
      f.stats_in = e->stats_a; f.stats_in_parts = cparts; f.stats_ld = Tpc; f.inv_width = 1.0f / e->ln_width; f.eps = g.ln_eps; f.act = g.act;
      f.ticket = ticket();
      pf.begin(PK_GEMM_FC1, 2.0 * c * dD * dM, st);
      HIP_TRY(ce_gemm_nt(f, CE_DT_BF16, EPI_LNFOLD, st));
      pf.end(st);
      // x[cls] += h . Wproj^T + b
      GemmParams r{};
      r.A = e->hid; r.lda = g.mlp_dim; r.W = L.w_proj; r.ldw = g.mlp_dim; r.M = c; r.N = Dw; r.K = g.mlp_dim;
      r.out = e->x; r.ldo = stride * Dw; r.bias = L.b_proj; r.resid = e->x; r.stats_out = e->stats_b; r.stats_ld = Tpc;
      r.ticket = ticket();
      pf.begin(PK_GEMM_RESID, 2.0 * c * dD * dM, st, PK_SUB_FC2);
      HIP_TRY(ce_gemm_nt(r, CE_DT_BF16, EPI_RESID, st));
      pf.end(st);
      break;
    }
    // K3: qkv = LN1(x) . Wqkv^T + b   (LayerNorm folded into the GEMM epilogue)
    GemmParams q{};
    q.A = e->x; q.lda = g.width; q.W = L.w_qkv; q.ldw = g.width; q.M = T; q.N = 3 * g.width; q.K = g.width;
    q.out = e->qkv; q.ldo = 3 * g.width; q.bias = L.b_qkv; q.colsum = L.cs_qkv;
    q.stats_in = stats_in; q.stats_in_parts = stats_parts; q.stats_ld = Tp; q.inv_width = 1.0f / e->ln_width; q.eps = g.ln_eps; q.act = -1;
      q.ticket = ticket();
    pf.begin(PK_GEMM_QKV, 2.0 * dT * 3.0 * dD * dD, st);
    HIP_TRY(ce_gemm_nt(q, CE_DT_BF16, EPI_LNFOLD, st));
    pf.end(st);
    // K4
    pf.begin(PK_ATTENTION, 4.0 * c * (double)e->tokens * e->tokens * dD, st);
    HIP_TRY(ce_attention(e->qkv, e->attn, c, e->tokens, g.width, g.heads, nullptr, 0, st));
    pf.end(st);
    // K5: x += attn . Wo^T + bo
    GemmParams o{};
    o.A = e->attn; o.lda = g.width; o.W = L.w_out; o.ldw = g.width; o.M = T; o.N = g.width; o.K = g.width;
    o.out = e->x; o.ldo = g.width; o.bias = L.b_out; o.resid = e->x; o.stats_out = e->stats_a; o.stats_ld = Tp;
      o.ticket = ticket();
    pf.begin(PK_GEMM_RESID, 2.0 * dT * dD * dD, st, PK_SUB_OUT);
    HIP_TRY(ce_gemm_nt(o, CE_DT_BF16, EPI_RESID, st));
    HIP_TRY(fold_stats(e->stats_a, Tp, T));
    pf.end(st);
    // K6: h = act(LN2(x) . Wfc^T + b)
    GemmParams f{};
    f.A = e->x; f.lda = g.width; f.W = L.w_fc; f.ldw = g.width; f.M = T; f.N = g.mlp_dim; f.K = g.width;
    f.out = e->hid; f.ldo = g.mlp_dim; f.bias = L.b_fc; f.colsum = L.cs_fc;
    f.stats_in = e->stats_a; f.stats_in_parts = cparts; f.stats_ld = Tp; f.inv_width = 1.0f / e->ln_width; f.eps = g.ln_eps; f.act = g.act;
      f.ticket = ticket();
    pf.begin(PK_GEMM_FC1, 2.0 * dT * dD * dM, st);
    HIP_TRY(ce_gemm_nt(f, CE_DT_BF16, EPI_LNFOLD, st));
    pf.end(st);
    // K7: x += h . Wproj^T + b
    GemmParams r{};
    r.A = e->hid; r.lda = g.mlp_dim; r.W = L.w_proj; r.ldw = g.mlp_dim; r.M = T; r.N = g.width; r.K = g.mlp_dim;
    r.out = e->x; r.ldo = g.width; r.bias = L.b_proj; r.resid = e->x; r.stats_out = e->stats_b; r.stats_ld = Tp;
      r.ticket = ticket();
    pf.begin(PK_GEMM_RESID, 2.0 * dT * dD * dM, st, PK_SUB_FC2);
    HIP_TRY(ce_gemm_nt(r, CE_DT_BF16, EPI_RESID, st));
    HIP_TRY(fold_stats(e->stats_b, Tp, T));
    pf.end(st);
    stats_in = e->stats_b; stats_parts = cparts;
  }
  return 0;
}

size_t crop_bytes(const clipenc_config& g, int in_dtype) {
  return (size_t)3 * g.image_size * g.image_size * (in_dtype == CLIPENC_IN_F32 ? 4 : (in_dtype == CLIPENC_IN_F16 ? 2 : 1));
}

}  // namespace

extern "C" {

const char* clipenc_last_error(void) { return g_err.c_str(); }

int clipenc_device_count(int* count) {
  if (!count) return fail("count is NULL");
  HIP_TRY(hipGetDeviceCount(count));
  return 0;
}

int clipenc_create(const clipenc_config* cfg, const clipenc_weights* w, int device, clipenc_t* out) {
  if (!cfg || !w || !out) return fail("clipenc_create: NULL argument");
  const clipenc_config u = *cfg;                            // the caller's tower; `g` below is the one the kernels run
  if (u.width <= 0 || u.width % 8 != 0) return fail("width %d must be a positive multiple of 8", u.width);
  if (u.mlp_dim <= 0) return fail("mlp_dim %d must be positive", u.mlp_dim);
  if (u.heads < 1 || u.width % u.heads != 0) return fail("width %d is not a multiple of heads %d", u.width, u.heads);
  // Shapes the kernels are built for: heads of 64, 80, 96, 112 or 128 columns, width = heads x that, width and mlp_dim multiples of 256.  Any other
  // tower is run as the next such shape with ZERO weights in the added places -- exact arithmetic, not an approximation:
  //   * a head of hd_r < hd columns gets hd - hd_r zero rows in W_q, W_k, W_v (and zero bias): the scores and the output do not see them;
  //     the attention kernels scale by hd^-1/2, so the q rows (weights and bias) carry (hd / hd_r)^1/2;
  //   * whole zero heads until heads x hd is a multiple of 256 (uniform softmax over zero values: zero output, zero out-projection columns);
  //   * residual-stream columns beyond the true width stay zero through every block (zero conv / embedding / out-proj / FC2 rows and bias);
  //     the LayerNorms divide by the TRUE width (ln_width) and their gamma / beta are zero there;
  //   * zero FC1 rows / FC2 columns up to a multiple of 256 (both activations map 0 to 0).
  const int hd_r = u.width / u.heads;
  const int hd = hd_r <= 64 ? 64 : (hd_r <= 80 ? 80 : (hd_r + 15) / 16 * 16);
  if (hd_r > 128) return fail("head dim %d > 128 not built (width %d, heads %d)", hd_r, u.width, u.heads);
  int heads_d = u.heads;
  while ((heads_d * hd) % 256 != 0) ++heads_d;
  clipenc_config g = u;
  g.heads = heads_d; g.width = heads_d * hd; g.mlp_dim = (int)align_up((size_t)u.mlp_dim, 256);
  const bool padded = g.width != u.width || g.heads != u.heads || g.mlp_dim != u.mlp_dim;
  if (g.patch <= 0 || g.image_size % g.patch != 0) return fail("image_size %d not divisible by patch %d", g.image_size, g.patch);
  if (g.embed_dim <= 0 || g.embed_dim > 1280) return fail("embed_dim %d out of range (1..1280)", g.embed_dim);
  if (g.width > 2048) return fail("width %d%s > 2048 not built (LayerNorm / head kernels)", g.width, padded ? " (after padding)" : "");
  if (g.layers < 1) return fail("layers %d < 1", g.layers);
  if (g.act != CLIPENC_ACT_QUICK_GELU && g.act != CLIPENC_ACT_GELU_ERF) return fail("unknown activation %d", g.act);
  const int grid = g.image_size / g.patch, tokens = grid * grid + 1;
  if (tokens > 640) return fail("%d tokens > 640: K and V of one head no longer fit the 160 KiB LDS", tokens);
  if (hd != 64 && tokens > 288) return fail("%d tokens > 288 at head dim %d: only the one-pass attention kernel is built for it", tokens, hd_r);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail("device %d out of range (%d visible)", device, ndev);
  HIP_TRY(hipSetDevice(device));

  clipenc_s* e = new clipenc_s();
  e->cfg = g; e->user = u; e->ln_width = u.width; e->padded = padded; e->device = device; e->tokens = tokens;
#ifdef CLIPENC_DIAG                         // diagnostic library only: run the last block on every token (tests/test_gpu_cls_only.py)
  e->cls_only_last = getenv("CLIPENC_FULL_LAST_BLOCK") == nullptr;
  e->fp8_unfused = getenv("CLIPENC_FP8_UNFUSED") != nullptr;
  e->dynamic_tail = getenv("CLIPENC_STATIC_TILES") == nullptr;
  e->cls_shortcut = getenv("CLIPENC_CLS_KV") == nullptr;
#endif
  if (padded) e->cls_shortcut = false;                     // (its kernels take 64^-1/2 and 1 / width as constants)
  const int kreal = 3 * g.patch * g.patch;
  e->kpad = (int)align_up(kreal, 128);
  const int D = g.width, M = g.mlp_dim, L = g.layers, E = g.embed_dim;

  // ---- host staging slab (bf16 weights with LayerNorm gamma folded in; fp32 vectors) ----
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
  const size_t o_conv = take((size_t)D * e->kpad * 2);
  const size_t o_cls = take(D * 4), o_pos = take((size_t)tokens * D * 4);
  const size_t o_lpw = take(D * 4), o_lpb = take(D * 4), o_low = take(D * 4), o_lob = take(D * 4);
  const size_t o_proj = take((size_t)D * E * 4);
  struct LOff { size_t w_qkv, w_out, w_fc, w_proj, cs_qkv, b_qkv, b_out, cs_fc, b_fc, b_proj; };
  std::vector<LOff> lo(L);
  for (int l = 0; l < L; ++l) {
    lo[l].w_qkv = take((size_t)3 * D * D * 2); lo[l].w_out = take((size_t)D * D * 2);
    lo[l].w_fc = take((size_t)M * D * 2); lo[l].w_proj = take((size_t)D * M * 2);
    lo[l].cs_qkv = take(3 * D * 4); lo[l].b_qkv = take(3 * D * 4); lo[l].b_out = take(D * 4);
    lo[l].cs_fc = take(M * 4); lo[l].b_fc = take(M * 4); lo[l].b_proj = take(D * 4);
  }
  const size_t o_wkt = take((size_t)D * D * 2);
  std::vector<char> host(off);
  char* hb = host.data();
  // device index -> index in the caller's tensor, or -1 (zero): residual columns, head-major columns (q / k / v rows, out-proj columns), MLP
  const int Du = u.width, Mu = u.mlp_dim;
  std::vector<int> res_map(D), head_map(D), mlp_map(M), qkv_map(3 * (size_t)D);
  for (int j = 0; j < D; ++j) {
    res_map[j] = j < Du ? j : -1;
    head_map[j] = (j / hd < u.heads && j % hd < hd_r) ? (j / hd) * hd_r + j % hd : -1;
  }
  for (int j = 0; j < M; ++j) mlp_map[j] = j < Mu ? j : -1;
  for (int j = 0; j < 3 * D; ++j) qkv_map[j] = head_map[j % D] < 0 ? -1 : (j / D) * Du + head_map[j % D];
  // zero-padded fp32 copy [Nd][Kd] of the caller's [..][Ku] matrix (rows / columns through the maps); vectors likewise
  auto pad_mat = [&](const float* W, int Ku, const std::vector<int>& nmap, int Nd, const std::vector<int>& kmap, int Kd) {
    std::vector<float> o((size_t)Nd * Kd, 0.f);
    float* op = o.data();
    const int *nm = nmap.data(), *km = kmap.data();
    parallel_for(Nd, [=](int n) {
      if (nm[n] < 0) return;
      for (int k = 0; k < Kd; ++k)
        if (km[k] >= 0) op[(size_t)n * Kd + k] = W[(size_t)nm[n] * Ku + km[k]];
    });
    return o;
  };
  auto pad_vec = [&](const float* v, const std::vector<int>& map, int n) {
    std::vector<float> o(n, 0.f);
    for (int i = 0; i < n; ++i) if (map[i] >= 0) o[i] = v[map[i]];
    return o;
  };
  {
    bf16_t* wc = (bf16_t*)(hb + o_conv);
    for (int n = 0; n < D; ++n)
      for (int k = 0; k < e->kpad; ++k)
        wc[(size_t)n * e->kpad + k] = (k < kreal && res_map[n] >= 0) ? host_f32_to_bf16(w->conv1_weight[(size_t)res_map[n] * kreal + k]) : 0;
    if (!padded) {
      memcpy(hb + o_cls, w->class_embedding, D * 4);
      memcpy(hb + o_pos, w->positional_embedding, (size_t)tokens * D * 4);
      memcpy(hb + o_lpw, w->ln_pre_w, D * 4); memcpy(hb + o_lpb, w->ln_pre_b, D * 4);
      memcpy(hb + o_low, w->ln_post_w, D * 4); memcpy(hb + o_lob, w->ln_post_b, D * 4);
      memcpy(hb + o_proj, w->proj, (size_t)D * E * 4);
    } else {
      std::vector<int> all_tok(tokens), all_e(E);
      for (int i = 0; i < tokens; ++i) all_tok[i] = i;
      for (int i = 0; i < E; ++i) all_e[i] = i;
      memcpy(hb + o_cls, pad_vec(w->class_embedding, res_map, D).data(), D * 4);
      memcpy(hb + o_pos, pad_mat(w->positional_embedding, Du, all_tok, tokens, res_map, D).data(), (size_t)tokens * D * 4);
      memcpy(hb + o_lpw, pad_vec(w->ln_pre_w, res_map, D).data(), D * 4); memcpy(hb + o_lpb, pad_vec(w->ln_pre_b, res_map, D).data(), D * 4);
      memcpy(hb + o_low, pad_vec(w->ln_post_w, res_map, D).data(), D * 4); memcpy(hb + o_lob, pad_vec(w->ln_post_b, res_map, D).data(), D * 4);
      memcpy(hb + o_proj, pad_mat(w->proj, E, res_map, D, all_e, E).data(), (size_t)D * E * 4);
    }
  }
  // LayerNorm folding:  LN(x).W^T + b = rstd*(x.(g*W)^T - mean*colsum) + (b + W.beta)
  auto fold = [&](const float* W, const float* b, const float* gamma, const float* beta, int N, int K, bf16_t* Wq,
                  float* colsum, float* bias) {
    parallel_for(N, [=](int n) {
      double cs = 0.0, bb = 0.0;
      for (int k = 0; k < K; ++k) {
        const float wv = W[(size_t)n * K + k];
        const bf16_t q = host_f32_to_bf16(wv * gamma[k]);
        Wq[(size_t)n * K + k] = q;
        cs += (double)host_bf16_to_f32(q);
        bb += (double)wv * (double)beta[k];
      }
      colsum[n] = (float)cs;
      bias[n] = (float)((double)b[n] + bb);
    });
  };
  auto plain = [&](const float* W, int N, int K, bf16_t* Wq) {
    parallel_for(N, [=](int n) {
      for (int k = 0; k < K; ++k) Wq[(size_t)n * K + k] = host_f32_to_bf16(W[(size_t)n * K + k]);
    });
  };
  for (int l = 0; l < L; ++l) {
    const float *ipw = w->in_proj_w[l], *ipb = w->in_proj_b[l], *opw = w->out_proj_w[l], *opb = w->out_proj_b[l];
    const float *g1 = w->ln_1_w[l], *b1 = w->ln_1_b[l], *g2 = w->ln_2_w[l], *b2 = w->ln_2_b[l];
    const float *fcw = w->c_fc_w[l], *fcb = w->c_fc_b[l], *pjw = w->c_proj_w[l], *pjb = w->c_proj_b[l];
    std::vector<float> t[12];                              // the padded copies of this block (released with the iteration)
    if (padded) {
      t[0] = pad_mat(ipw, Du, qkv_map, 3 * D, res_map, D); t[1] = pad_vec(ipb, qkv_map, 3 * D);
      const float qs = sqrtf((float)hd / (float)hd_r);     // the kernels' hd^-1/2 -> the tower's hd_r^-1/2
      if (hd != hd_r) {
        for (size_t i = 0; i < (size_t)D * D; ++i) t[0][i] *= qs;
        for (int i = 0; i < D; ++i) t[1][i] *= qs;
      }
      t[2] = pad_mat(opw, Du, res_map, D, head_map, D); t[3] = pad_vec(opb, res_map, D);
      t[4] = pad_vec(g1, res_map, D); t[5] = pad_vec(b1, res_map, D); t[6] = pad_vec(g2, res_map, D); t[7] = pad_vec(b2, res_map, D);
      t[8] = pad_mat(fcw, Du, mlp_map, M, res_map, D); t[9] = pad_vec(fcb, mlp_map, M);
      t[10] = pad_mat(pjw, Mu, res_map, D, mlp_map, M); t[11] = pad_vec(pjb, res_map, D);
      ipw = t[0].data(); ipb = t[1].data(); opw = t[2].data(); opb = t[3].data();
      g1 = t[4].data(); b1 = t[5].data(); g2 = t[6].data(); b2 = t[7].data();
      fcw = t[8].data(); fcb = t[9].data(); pjw = t[10].data(); pjb = t[11].data();
    }
    fold(ipw, ipb, g1, b1, 3 * D, D, (bf16_t*)(hb + lo[l].w_qkv), (float*)(hb + lo[l].cs_qkv), (float*)(hb + lo[l].b_qkv));
    plain(opw, D, D, (bf16_t*)(hb + lo[l].w_out));
    memcpy(hb + lo[l].b_out, opb, D * 4);
    fold(fcw, fcb, g2, b2, M, D, (bf16_t*)(hb + lo[l].w_fc), (float*)(hb + lo[l].cs_fc), (float*)(hb + lo[l].b_fc));
    plain(pjw, D, M, (bf16_t*)(hb + lo[l].w_proj));
    memcpy(hb + lo[l].b_proj, pjb, D * 4);
  }
  {
    // (gamma-folded W_k of the LAST layer)^T: Wt[k][n] = W'_k[n][k]  (cls_attention.hip: r_h = sum_{n in head} q_n W'_k[n, :])
    const bf16_t* wk = (const bf16_t*)(hb + lo[L - 1].w_qkv) + (size_t)D * D;
    bf16_t* wt = (bf16_t*)(hb + o_wkt);
    parallel_for(D, [=](int k) { for (int n = 0; n < D; ++n) wt[(size_t)k * D + n] = wk[(size_t)n * D + k]; });
  }
  hipError_t err = e->weights.alloc(off);
  if (err == hipSuccess) err = hipMemcpy(e->weights.p, hb, off, hipMemcpyHostToDevice);
  if (err != hipSuccess) { e->weights.release(); delete e; return fail("weight upload failed: %s", hipGetErrorString(err)); }
  char* db = (char*)e->weights.p;
  e->w_conv = (bf16_t*)(db + o_conv); e->cls = (float*)(db + o_cls); e->pos = (float*)(db + o_pos);
  e->ln_pre_w = (float*)(db + o_lpw); e->ln_pre_b = (float*)(db + o_lpb);
  e->ln_post_w = (float*)(db + o_low); e->ln_post_b = (float*)(db + o_lob); e->proj = (float*)(db + o_proj);
  e->w_k_t = (bf16_t*)(db + o_wkt);
  e->layers.resize(L);
  for (int l = 0; l < L; ++l) {
    LayerDev& d = e->layers[l];
    d.w_qkv = (bf16_t*)(db + lo[l].w_qkv); d.w_out = (bf16_t*)(db + lo[l].w_out);
    d.w_fc = (bf16_t*)(db + lo[l].w_fc); d.w_proj = (bf16_t*)(db + lo[l].w_proj);
    d.cs_qkv = (float*)(db + lo[l].cs_qkv); d.b_qkv = (float*)(db + lo[l].b_qkv); d.b_out = (float*)(db + lo[l].b_out);
    d.cs_fc = (float*)(db + lo[l].cs_fc); d.b_fc = (float*)(db + lo[l].b_fc); d.b_proj = (float*)(db + lo[l].b_proj);
  }
  if (int rc = ensure_workspace(e)) { clipenc_destroy(e); return rc; }
  *out = e;
  return 0;
}

int clipenc_destroy(clipenc_t e) {
  if (!e) return 0;
  (void)hipSetDevice(e->device);
  e->prof.release();
  e->weights.release();
  e->weights8.release();
  e->ws.release();
  delete e;
  return 0;
}

int clipenc_set_pixel_norm(clipenc_t e, const float* mean3, const float* std3) {
  if (!e || !mean3 || !std3) return fail("NULL argument");
  for (int i = 0; i < 3; ++i) {
    if (!(std3[i] > 0.f)) return fail("std[%d] = %g must be positive", i, std3[i]);
    e->pix_mean[i] = mean3[i]; e->pix_std[i] = std3[i];
  }
  return 0;
}

int clipenc_set_precision(clipenc_t e, int precision) {
  if (!e) return fail("NULL handle");
  if (precision != CLIPENC_PREC_BF16 && precision != CLIPENC_PREC_FP8) return fail("unknown precision %d", precision);
  if (precision == CLIPENC_PREC_FP8 && e->layers8.empty()) {
    const clipenc_config& g = e->cfg;
    if (g.mlp_dim > 8192 || g.width > 4096) return fail("fp8: width %d over 4096 (the row quantiser) or mlp_dim %d over 8192 not built", g.width, g.mlp_dim);
    HIP_TRY(hipSetDevice(e->device));
    const size_t D = g.width, M = g.mlp_dim;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    struct LOff { size_t w[4], s[4], is_hid, is_attn, cs_qkv, cs_fc, e_all; };
    const int pow2 = fp8_fused(g) ? 1 : 0;                  // the fused tower's GEMMs take the weight scale as an MFMA block exponent
    std::vector<LOff> lo(g.layers);
    const size_t rows[4] = {3 * D, D, M, D}, cols[4] = {D, D, D, M};
    for (auto& o : lo) {
      for (int i = 0; i < 4; ++i) { o.w[i] = take(rows[i] * cols[i]); o.s[i] = take(rows[i] * 4); }
      o.is_hid = take(M * 4); o.is_attn = take(D * 4);
      o.cs_qkv = take(3 * D * 4); o.cs_fc = take(M * 4);
      o.e_all = take(5 * D + M);
    }
    HIP_TRY(e->weights8.alloc(off));
    DevBuf tmp;                                            // [mlp_dim] static scales + fp32 [width][mlp_dim] folded w_proj
    const size_t o_fold = align_up(std::max(M, D) * 4, 256);
    struct Guard { DevBuf& b; ~Guard() { b.release(); } } tmp_guard{tmp};
    HIP_TRY(tmp.alloc(o_fold + D * std::max(M, D) * 4));
    float* s_hid = (float*)tmp.p;
    float* folded = (float*)((char*)tmp.p + o_fold);
    char* db = (char*)e->weights8.p;
    std::vector<LayerDev8> l8(g.layers);
    for (int l = 0; l < g.layers; ++l) {
      const LayerDev& L = e->layers[l];
      LayerDev8& Q = l8[l];
      Q.w_qkv = (uint8_t*)(db + lo[l].w[0]); Q.w_out = (uint8_t*)(db + lo[l].w[1]);
      Q.w_fc = (uint8_t*)(db + lo[l].w[2]); Q.w_proj = (uint8_t*)(db + lo[l].w[3]);
      Q.s_qkv = (float*)(db + lo[l].s[0]); Q.s_out = (float*)(db + lo[l].s[1]);
      Q.s_fc = (float*)(db + lo[l].s[2]); Q.s_proj = (float*)(db + lo[l].s[3]);
      const bf16_t* src[4] = {L.w_qkv, L.w_out, L.w_fc, L.w_proj};   // (gamma already folded into w_qkv / w_fc)
      uint8_t* dst[4] = {Q.w_qkv, Q.w_out, Q.w_fc, Q.w_proj};
      float* sc[4] = {Q.s_qkv, Q.s_out, Q.s_fc, Q.s_proj};
      for (int i = 0; i < 3; i += 2)
        HIP_TRY(ce_quant_rows_fp8(src[i], 0, cols[i], dst[i], cols[i], sc[i], (int)rows[i], (int)cols[i], 0, 0.f, nullptr, pow2));
      // the folded mean term multiplies the column sums of the rows the GEMM actually multiplies by: the dequantised ones
      Q.cs_qkv = (float*)(db + lo[l].cs_qkv); Q.cs_fc = (float*)(db + lo[l].cs_fc);
      HIP_TRY(ce_colsum_fp8(Q.w_qkv, Q.s_qkv, (int)(3 * D), (int)D, Q.cs_qkv, nullptr));
      HIP_TRY(ce_colsum_fp8(Q.w_fc, Q.s_fc, (int)M, (int)D, Q.cs_fc, nullptr));
      // attention output: a softmax-weighted mean of V rows, so the bound of the V third of the LN-folded w_qkv holds for it
      Q.is_attn = (float*)(db + lo[l].is_attn);
      HIP_TRY(ce_static_scale(L.w_qkv + 2 * D * D, L.b_qkv + 2 * D, (int)D, (int)D, s_hid, Q.is_attn, nullptr));
      HIP_TRY(ce_scale_cols(L.w_out, s_hid, folded, (int)D, (int)D, nullptr));
      HIP_TRY(ce_quant_rows_fp8(folded, 1, D, Q.w_out, D, Q.s_out, (int)D, (int)D, 0, 0.f, nullptr, pow2));
      // MLP hidden: static column scales from the LN-folded FC1 rows; their product with w_proj's columns is what gets quantised
      Q.is_hid = (float*)(db + lo[l].is_hid);
      HIP_TRY(ce_static_scale(L.w_fc, L.b_fc, (int)M, (int)D, s_hid, Q.is_hid, nullptr));
      HIP_TRY(ce_scale_cols(L.w_proj, s_hid, folded, (int)D, (int)M, nullptr));
      HIP_TRY(ce_quant_rows_fp8(folded, 1, M, Q.w_proj, M, Q.s_proj, (int)D, (int)M, 0, 0.f, nullptr, pow2));
      Q.e_all = (uint8_t*)(db + lo[l].e_all);
      {
        size_t eo = 0;
        for (int i = 0; i < 4; ++i) {
          HIP_TRY(ce_scale_exponents(sc[i], Q.e_all + eo, (int)rows[i], nullptr, nullptr));
          eo += rows[i];
        }
      }
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    e->layers8.swap(l8);
  }
  const int old_precision = e->precision;
  e->precision = precision;                                // (ensure_workspace sizes for e->precision)
  if (int rc = ensure_workspace(e)) { e->precision = old_precision; return rc; }   // the fp8 operand buffers belong to the workspace
  return 0;
}

int clipenc_set_chunk(clipenc_t e, int chunk_crops) {
  if (!e) return fail("NULL handle");
  if (chunk_crops < 1) return fail("chunk_crops %d < 1", chunk_crops);
  HIP_TRY(hipSetDevice(e->device));
  const int old_chunk = e->chunk;
  e->chunk = chunk_crops;
  if (int rc = ensure_workspace(e)) { e->chunk = old_chunk; return rc; }
  return 0;
}

int clipenc_get_info(clipenc_t e, int* tokens, int* embed_dim, int* chunk_crops, size_t* workspace_bytes) {
  if (!e) return fail("NULL handle");
  if (tokens) *tokens = e->tokens;
  if (embed_dim) *embed_dim = e->cfg.embed_dim;
  if (chunk_crops) *chunk_crops = e->chunk;
  if (workspace_bytes) *workspace_bytes = e->ws.bytes;
  return 0;
}

int clipenc_encode(clipenc_t e, const void* crops_dev, int n_crops, int in_dtype, float* emb_dev, int normalize,
                   void* stream) {
  if (!e) return fail("NULL handle");
  if (n_crops < 0) return fail("n_crops %d < 0", n_crops);
  if (n_crops == 0) return 0;
  if (!crops_dev || !emb_dev) return fail("NULL device pointer");
  if (in_dtype != CLIPENC_IN_F32 && in_dtype != CLIPENC_IN_F16 && in_dtype != CLIPENC_IN_U8) return fail("unknown in_dtype %d", in_dtype);
  if ((in_dtype != CLIPENC_IN_U8 && ((uintptr_t)crops_dev & 1)) || ((uintptr_t)emb_dev & 3)) return fail("misaligned device pointer");
  HIP_TRY(hipSetDevice(e->device));
  if (e->ws_chunk != e->chunk || e->ws_precision != e->precision) return fail("workspace not allocated (internal error)");
  hipStream_t st = (hipStream_t)stream;
  const clipenc_config& g = e->cfg;
  const size_t cb = crop_bytes(g, in_dtype);
  for (int c0 = 0; c0 < n_crops; c0 += e->chunk) {
    const int c = std::min(e->chunk, n_crops - c0);
    if (int rc = run_tower(e, (const char*)crops_dev + (size_t)c0 * cb, c, in_dtype, g.layers, st, e->cls_only_last)) return rc;
    e->prof.begin(PK_HEAD, 2.0 * c * (double)e->user.width * g.embed_dim, st);
    HIP_TRY(ce_head(e->x, e->ln_post_w, e->ln_post_b, e->proj, emb_dev + (size_t)c0 * g.embed_dim, c, e->tokens,
                    g.width, e->ln_width, g.embed_dim, g.ln_eps, normalize, st));
    e->prof.end(st);
  }
  return 0;
}

int clipenc_profile_enable(clipenc_t e, int on) {
  if (!e) return fail("NULL handle");
  HIP_TRY(hipSetDevice(e->device));
  e->prof.collect();
  e->prof.on = on != 0;
  return 0;
}

int clipenc_clock_probe(int device, unsigned long long* out2_dev, int spin_us, void* stream) {
  if (!out2_dev) return fail("clock_probe: NULL device pointer");
  if (spin_us < 1 || spin_us > 10000) return fail("clock_probe: spin_us %d outside 1..10000", spin_us);
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(ce_clock_probe(out2_dev, spin_us * 100, (hipStream_t)stream));
  return 0;
}

int clipenc_mfma_stream_probe(int device, int fp8, const void* operands_dev, float* sink_dev, long long iters, double* flop_out, void* stream) {
  if (!operands_dev || !sink_dev) return fail("mfma_stream_probe: NULL device pointer");
  if (iters < 1 || iters > (1ll << 31)) return fail("mfma_stream_probe: iters %lld outside 1..2^31", iters);
  HIP_TRY(hipSetDevice(device));
  int n_cu = 0;
  HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device));
  HIP_TRY(ce_mfma_stream(operands_dev, fp8 != 0, sink_dev, iters, n_cu, (hipStream_t)stream));
  // per wave and iteration: 16 x 16x16x32 (16 384 flop) = 8 x 32x32x64 (131 072 flop) ... both 2 x M x N x K per instruction
  if (flop_out) *flop_out = (double)n_cu * 8.0 * (double)iters * (fp8 ? 8.0 * 131072.0 : 16.0 * 16384.0);
  return 0;
}

int clipenc_profile_kinds(void) { return PK_COUNT; }

int clipenc_profile_read(clipenc_t e, int kind, const char** name, double* total_ms, long long* launches,
                         double* algorithmic_flops, int reset) {
  if (!e) return fail("NULL handle");
  if (kind < 0 || kind >= PK_COUNT) return fail("profile kind %d out of range", kind);
  HIP_TRY(hipSetDevice(e->device));
  e->prof.collect();
  if (name) {
    *name = kProfileNames[kind];
    if (e->cfg.act == CLIPENC_ACT_GELU_ERF && kind == PK_GEMM_FC1) *name = "gemm_persist_kernel<2, 1>";
    if (e->cfg.act == CLIPENC_ACT_GELU_ERF && kind == PK_GEMM8_FC1) *name = "gemm_fp8_kernel<2, 1, false>";
    if (fp8_fused(e->cfg) && !e->fp8_unfused) {              // the fused fp8 tower's instantiations (run_tower)
      if (kind == PK_GEMM8_QKV) *name = "gemm_fp8_kernel<0, -1, true>";
      if (kind == PK_GEMM8_FC1) *name = e->cfg.act == CLIPENC_ACT_GELU_ERF ? "gemm_fp8_kernel<2, 1, true>" : "gemm_fp8_kernel<2, 0, true>";
      if (kind == PK_GEMM8_RESID) *name = "gemm_fp8_kernel<3, -1, false>";
      if (kind == PK_SUB8_OUT) *name = "shape:out_proj(gemm_fp8_kernel<3, -1, false>)";
      if (kind == PK_SUB8_FC2) *name = "shape:fc2(gemm_fp8_kernel<3, -1, false>)";
      if (kind == PK_QUANT_BLOCK) { static thread_local char qb[32]; snprintf(qb, sizeof qb, "quant_block_kernel<%d>", e->cfg.width / 128); *name = qb; }
    }
    if (kind == PK_PATCHIFY || kind == PK_EMBED_LN_PRE) {   // (patchify: fp32 crops; uint8 / f16 inputs run the <unsigned char, P> / <_Float16, P> twins)
      static thread_local char nb[2][40];
      snprintf(nb[0], sizeof nb[0], "patchify_kernel<float, %d>", e->cfg.patch);
      snprintf(nb[1], sizeof nb[1], "embed_ln_pre_kernel<%d>", (e->cfg.width + 511) / 512);
      *name = nb[kind == PK_EMBED_LN_PRE];
    }
    if (kind == PK_ATTENTION) {                      // the instantiation ce_attention picks for this token count
      const int nkt = (e->tokens + 31) / 32;
      if (e->cfg.heads * 64 != e->cfg.width) { static thread_local char hb[32]; const int hdv = e->cfg.width / e->cfg.heads; snprintf(hb, sizeof hb, "attn_hd_kernel<%d, %d, 8>", nkt, hdv); *name = hb; }
      else if (nkt > 19) *name = "attn_long_kernel<12>";
      else if (nkt > 9) *name = "attn_long_stream_kernel<11>";     // (launches of fewer than 64 tasks take attn_long_kernel<12>)
      else if (nkt == 9) *name = (e->tokens & 31) == 1 ? "attn_stream_kernel<9, 7, true>" : "attn_stream_kernel<9, 7, false>";
      else if (nkt >= 3) { static thread_local char sb[40]; snprintf(sb, sizeof sb, "attn_stream_kernel<%d, 7, false>", nkt); *name = sb; }   // (launches of fewer than 64 tasks: attn_kernel<NKT>)
      else { static thread_local char buf[32]; snprintf(buf, sizeof buf, "attn_kernel<%d>", nkt); *name = buf; }
    }
  }
  if (total_ms) *total_ms = e->prof.ms[kind];
  if (launches) *launches = e->prof.launches[kind];
  if (algorithmic_flops) *algorithmic_flops = e->prof.flops[kind];
  if (reset) { e->prof.ms[kind] = 0; e->prof.launches[kind] = 0; e->prof.flops[kind] = 0; }
  return 0;
}

int clipenc_forward_tokens(clipenc_t e, const void* crops_dev, int n_crops, int in_dtype, int n_layers,
                             void* x_out, void* stream) {
  if (!e || !crops_dev || !x_out) return fail("NULL argument");
  if (n_crops < 1 || n_crops > e->chunk) return fail("n_crops %d outside 1..chunk(%d)", n_crops, e->chunk);
  if (n_layers < 0 || n_layers > e->cfg.layers) return fail("n_layers %d out of range", n_layers);
  HIP_TRY(hipSetDevice(e->device));
  if (e->ws_chunk != e->chunk || e->ws_precision != e->precision) return fail("workspace not allocated (internal error)");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = run_tower(e, crops_dev, n_crops, in_dtype, n_layers, st)) return rc;
  if (!e->padded) {
    HIP_TRY(hipMemcpyAsync(x_out, e->x, (size_t)n_crops * e->tokens * e->cfg.width * 2, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  // [rows][user width] to the caller: a padded tower's rows are wider on the device, their extra columns are zeros
  HIP_TRY(hipMemcpy2DAsync(x_out, (size_t)e->user.width * 2, e->x, (size_t)e->cfg.width * 2, (size_t)e->user.width * 2,
                           (size_t)n_crops * e->tokens, hipMemcpyDeviceToDevice, st));
  return 0;
}

int fcreg_create(int n_layers, const int* sizes, const float* const* W, const float* const* b, float negative_slope,
                 int device, fcreg_t* out) {
  if (!sizes || !W || !b || !out) return fail("fcreg_create: NULL argument");
  if (n_layers < 1 || n_layers > CE_FC_MAX_LAYERS) return fail("n_layers %d outside 1..%d", n_layers, CE_FC_MAX_LAYERS);
  for (int l = 0; l <= n_layers; ++l) {
    if (sizes[l] < 1) return fail("layer size %d at %d", sizes[l], l);
    if (sizes[l] > CE_FC_MAX_WIDTH)
      return fail("layer %d is %d wide; the fused regressor kernel keeps a row block of every layer in LDS and takes at most %d "
                  "(e.g. 6 crops x 768 = 4608)", l, sizes[l], CE_FC_MAX_WIDTH);
  }
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail("device %d out of range (%d visible)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  fcreg_s* r = new fcreg_s();
  r->device = device; r->n_layers = n_layers; r->negative_slope = negative_slope;
  size_t off = 0;
  std::vector<size_t> ow(n_layers), ob(n_layers), orr(n_layers);
  for (int l = 0; l <= n_layers; ++l) r->sizes[l] = sizes[l];
  auto tiles32 = [](int n) { return (size_t)((n + 31) / 32) * 32; };
  for (int l = 0; l < n_layers; ++l) {
    ow[l] = off; off += align_up((size_t)sizes[l] * sizes[l + 1] * 4, 256);
    ob[l] = off; off += align_up((size_t)sizes[l + 1] * 4, 256);
    // store-scale kernel: [out rounded up to 32][in (layer 0) | in rounded up to 32 (later layers)], zero padded
    orr[l] = off; off += align_up(tiles32(sizes[l + 1]) * (l == 0 ? (size_t)sizes[0] : tiles32(sizes[l])) * 4, 256);
  }
  std::vector<char> host(off, 0);
  for (int l = 0; l < n_layers; ++l) {
    const int in = sizes[l], on = sizes[l + 1];
    float* wt = (float*)(host.data() + ow[l]);
    for (int j = 0; j < on; ++j)
      for (int k = 0; k < in; ++k) wt[(size_t)k * on + j] = W[l][(size_t)j * in + k];     // transpose to [in][out]
    memcpy(host.data() + ob[l], b[l], (size_t)on * 4);
    float* wr = (float*)(host.data() + orr[l]);
    const size_t ldr = l == 0 ? (size_t)in : tiles32(in);
    for (int j = 0; j < on; ++j) memcpy(wr + (size_t)j * ldr, W[l] + (size_t)j * in, (size_t)in * 4);
  }
  hipError_t err = r->slab.alloc(off);
  if (err == hipSuccess) err = hipMemcpy(r->slab.p, host.data(), off, hipMemcpyHostToDevice);
  if (err != hipSuccess) { r->slab.release(); delete r; return fail("regressor upload failed: %s", hipGetErrorString(err)); }
  for (int l = 0; l < n_layers; ++l) {
    r->Wt[l] = (const float*)((char*)r->slab.p + ow[l]);
    r->b[l] = (const float*)((char*)r->slab.p + ob[l]);
    r->Wr[l] = (const float*)((char*)r->slab.p + orr[l]);
  }
  *out = r;
  return 0;
}

int fcreg_destroy(fcreg_t r) {
  if (!r) return 0;
  (void)hipSetDevice(r->device);
  r->slab.release();
  delete r;
  return 0;
}

int fcreg_forward(fcreg_t r, const float* x_dev, int n_rows, long row_stride, int n_seg, int seg_len,
                  const int* seg_off, float* y_dev, void* stream) {
  if (!r) return fail("NULL handle");
  if (n_rows < 0) return fail("n_rows %d < 0", n_rows);
  if (n_rows == 0) return 0;
  if (!x_dev || !y_dev || !seg_off) return fail("NULL pointer");
  if (n_seg < 1 || n_seg > CE_FC_MAX_SEG) return fail("n_seg %d outside 1..%d", n_seg, CE_FC_MAX_SEG);
  if ((long)n_seg * seg_len != r->sizes[0])
    return fail("input width mismatch: %d segments x %d != regressor input %d", n_seg, seg_len, r->sizes[0]);
  HIP_TRY(hipSetDevice(r->device));
  FcRegParams p{};
  p.n_layers = r->n_layers;
  for (int l = 0; l <= r->n_layers; ++l) p.sizes[l] = r->sizes[l];
  for (int l = 0; l < r->n_layers; ++l) { p.Wt[l] = r->Wt[l]; p.b[l] = r->b[l]; p.Wr[l] = r->Wr[l]; }
  p.negative_slope = r->negative_slope;
  p.x = x_dev; p.row_stride = row_stride; p.n_seg = n_seg; p.seg_len = seg_len;
  for (int s = 0; s < n_seg; ++s) p.seg_off[s] = seg_off[s];
  p.y = y_dev; p.n_rows = n_rows;
  HIP_TRY(ce_fcreg_forward(p, (hipStream_t)stream));
  return 0;
}

int clipenc_encode_score(clipenc_t e, fcreg_t r, const void* crops_dev, int n_images, int crops_per_image, int in_dtype,
                         const int* crop_select, int n_select, float* emb_dev, float* score_dev, void* stream) {
  if (!e || !r) return fail("NULL handle");
  if (e->device != r->device) return fail("encoder on device %d, regressor on device %d", e->device, r->device);
  if (n_images < 0 || crops_per_image < 1) return fail("bad shape: %d images x %d crops", n_images, crops_per_image);
  if (!crop_select || n_select < 1 || n_select > CE_FC_MAX_SEG) return fail("bad crop selection");
  const int E = e->cfg.embed_dim;
  int seg_off[CE_FC_MAX_SEG];
  for (int s = 0; s < n_select; ++s) {
    if (crop_select[s] < 0 || crop_select[s] >= crops_per_image) return fail("crop_select[%d] = %d out of range", s, crop_select[s]);
    seg_off[s] = crop_select[s] * E;
  }
  if (n_images == 0) return 0;
  if (int rc = clipenc_encode(e, crops_dev, n_images * crops_per_image, in_dtype, emb_dev, 1, stream)) return rc;
  double fl = 0.0;
  for (int l = 0; l < r->n_layers; ++l) fl += 2.0 * n_images * (double)r->sizes[l] * r->sizes[l + 1];
  e->prof.begin(PK_FCREG, fl, (hipStream_t)stream);
  const int rc = fcreg_forward(r, emb_dev, n_images, (long)crops_per_image * E, n_select, E, seg_off, score_dev, stream);
  e->prof.end((hipStream_t)stream);
  return rc;
}

int dedup_find_pairs(const void* emb_f16_dev, int n, int d, float threshold, int fp16_compare, void* ehat_ws_dev,
                     long long* pairs_dev, float* vals_dev, unsigned long long capacity,
                     unsigned long long* count_dev, void* stream) {
  if (n < 0 || d < 1) return fail("dedup_find_pairs: bad shape n=%d d=%d", n, d);
  if (!count_dev) return fail("dedup_find_pairs: count_dev is NULL");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(unsigned long long), st));
  if (n < 2) return 0;
  if (!emb_f16_dev || !ehat_ws_dev || (capacity && (!pairs_dev || !vals_dev))) return fail("dedup_find_pairs: NULL device pointer");
  const int ld = (d + 127) / 128 * 128;
  HIP_TRY(ce_dedup_normalize_f16(emb_f16_dev, ehat_ws_dev, n, d, ld, st));
  HIP_TRY(ce_dedup_pairs(ehat_ws_dev, n, d, ld, threshold, fp16_compare, pairs_dev, vals_dev, capacity, count_dev, st));
  return 0;
}

namespace {
struct ScreenWs { size_t q8, margin, cand, total; int ld8; unsigned long long cand_cap; };
// layout of the screened search's scratch: [0, 256) the candidate counter | e4m3 rows | margins | candidate slots
ScreenWs screen_ws_layout(int n, int d, unsigned long long cand_cap) {
  const size_t n_pad = ((size_t)n + 255) / 256 * 256;
  const int ld = (d + 127) / 128 * 128, ld8 = std::max(512, (ld + 255) / 256 * 256);   // (the fp8 pipeline wants two stage pairs)
  ScreenWs w{};
  w.ld8 = ld8; w.cand_cap = cand_cap;
  w.q8 = 256;
  w.margin = w.q8 + n_pad * (size_t)ld8;
  w.cand = (w.margin + n_pad * sizeof(float) + 255) / 256 * 256;
  w.total = w.cand + (size_t)cand_cap * 8;
  return w;
}
}  // namespace

size_t dedup_screen_ws_bytes(int n, int d, unsigned long long candidate_capacity) {
  if (n < 0 || d < 1) return 0;
  return screen_ws_layout(n, d, candidate_capacity).total;
}

int dedup_find_pairs_screened(const void* emb_f16_dev, int n, int d, float threshold, int fp16_compare, void* ehat_ws_dev,
                              void* screen_ws_dev, size_t screen_ws_bytes, unsigned long long candidate_capacity, long long* pairs_dev,
                              float* vals_dev, unsigned long long capacity, unsigned long long* count_dev, void* stream) {
  if (n < 0 || d < 1) return fail("dedup_find_pairs_screened: bad shape n=%d d=%d", n, d);
  if (!count_dev) return fail("dedup_find_pairs_screened: count_dev is NULL");
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(unsigned long long), st));
  if (n < 2) return 0;
  if (!emb_f16_dev || !ehat_ws_dev || !screen_ws_dev || (capacity && (!pairs_dev || !vals_dev))) return fail("dedup_find_pairs_screened: NULL device pointer");
  if (candidate_capacity < 1) return fail("dedup_find_pairs_screened: candidate_capacity must be at least 1");
  const ScreenWs w = screen_ws_layout(n, d, candidate_capacity);
  if (screen_ws_bytes < w.total) return fail("dedup_find_pairs_screened: screen_ws_bytes %zu < %zu (dedup_screen_ws_bytes)", screen_ws_bytes, w.total);
  if ((uintptr_t)screen_ws_dev & 255) return fail("dedup_find_pairs_screened: screen_ws_dev must be 256-byte aligned");
  const int ld = (d + 127) / 128 * 128;
  char* ws = (char*)screen_ws_dev;
  HIP_TRY(ce_dedup_normalize_quant(emb_f16_dev, ehat_ws_dev, n, d, ld, ws + w.q8, w.ld8, (float*)(ws + w.margin), (unsigned long long*)ws,
                                   candidate_capacity, st));
  HIP_TRY(ce_dedup_pairs_screened(ehat_ws_dev, n, ld, threshold, fp16_compare, ws + w.q8, (float*)(ws + w.margin), ws + w.cand,
                                  candidate_capacity, (unsigned long long*)ws, pairs_dev, vals_dev, capacity, count_dev, st));
  return 0;
}

int dedup_tile_order(int tiles_per_side, int grid, unsigned* order_out, long capacity) {
  if (tiles_per_side < 1 || tiles_per_side > 0xffff || grid < 1 || !order_out) return fail("dedup_tile_order: bad argument");
  const std::vector<unsigned> order = tri_tile_order(tiles_per_side, grid);
  if ((long)order.size() > capacity) return fail("dedup_tile_order: capacity %ld < %zu tiles", capacity, order.size());
  memcpy(order_out, order.data(), order.size() * sizeof(unsigned));
  return 0;
}

int fctrain_create(int n_layers, const int* sizes, const float* const* W, const float* const* b, float negative_slope,
                   int device, fctrain_t* out) {
  if (!sizes || !W || !b || !out) return fail("fctrain_create: NULL argument");
  if (n_layers < 1 || n_layers > CE_FC_MAX_LAYERS) return fail("fctrain_create: %d layers (1..%d supported)", n_layers, CE_FC_MAX_LAYERS);
  if (sizes[n_layers] != 1) return fail("fctrain_create: the last layer must have one output (MSE on a scalar label), got %d", sizes[n_layers]);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail("device %d out of range (%d visible)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  hipError_t err = hipSuccess;
  FcTrainState* st = ce_fctrain_create(n_layers, sizes, W, b, negative_slope, &err);
  if (!st) return fail("fctrain_create failed: %s", hipGetErrorString(err));
  fctrain_s* t = new fctrain_s();
  t->device = device; t->st = st;
  *out = t;
  return 0;
}

int fctrain_destroy(fctrain_t t) {
  if (!t) return 0;
  (void)hipSetDevice(t->device);
  ce_fctrain_destroy(t->st);
  delete t;
  return 0;
}

int fctrain_epoch(fctrain_t t, const float* x_dev, const float* labels_dev, const long long* order_dev, long n_order,
                  int batch_size, float lr, float weight_decay, float dropout_prob, unsigned seed, float* batch_losses_dev,
                  void* stream) {
  if (!t || !t->st) return fail("NULL handle");
  if (!x_dev || !labels_dev) return fail("fctrain_epoch: NULL device pointer");
  if (n_order == 0) return 0;
  HIP_TRY(hipSetDevice(t->device));
  hipError_t err = ce_fctrain_epoch(t->st, x_dev, labels_dev, order_dev, n_order, batch_size, lr, weight_decay, dropout_prob, seed,
                                    batch_losses_dev, (hipStream_t)stream);
  if (err != hipSuccess) return fail("fctrain_epoch(%ld rows, batch %d) failed: %s", n_order, batch_size, hipGetErrorString(err));
  return 0;
}

int fctrain_predict(fctrain_t t, const float* x_dev, long n, float* y_dev, void* stream) {
  if (!t || !t->st) return fail("NULL handle");
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(t->device));
  hipError_t err = ce_fctrain_predict(t->st, x_dev, n, y_dev, (hipStream_t)stream);
  if (err != hipSuccess) return fail("fctrain_predict(%ld rows) failed: %s", n, hipGetErrorString(err));
  return 0;
}

int fctrain_get_params(fctrain_t t, int layer, float* W_host, float* b_host) {
  if (!t || !t->st) return fail("NULL handle");
  HIP_TRY(hipSetDevice(t->device));
  hipError_t err = ce_fctrain_get_params(t->st, layer, W_host, b_host);
  if (err != hipSuccess) return fail("fctrain_get_params(layer %d) failed: %s", layer, hipGetErrorString(err));
  return 0;
}

int simsearch_distances(const void* emb_dev, int emb_f16, long n, int d, long row_stride, const float* query_dev,
                        int measure, float* dist_dev, void* stream) {
  if (!emb_dev || !query_dev || !dist_dev) return fail("simsearch_distances: NULL device pointer");
  if (n < 0) return fail("simsearch_distances: n %ld < 0", n);
  if (n == 0) return 0;
  if (measure != SIMSEARCH_L2 && measure != SIMSEARCH_COSINE) return fail("Similarity measure %d not implemented!", measure);
  hipError_t err = ce_simsearch_distances(emb_dev, emb_f16, n, d, row_stride, query_dev, measure, dist_dev, (hipStream_t)stream);
  if (err != hipSuccess) return fail("simsearch_distances(%ld x %d) failed: %s", n, d, hipGetErrorString(err));
  return 0;
}

size_t simsearch_topn_workspace(long n, int top_n) { return (n < 1 || top_n < 1) ? 256 : ce_topn_workspace_bytes(n, top_n); }

int simsearch_topn(const float* dist_dev, long n, int top_n, long long* idx_out_dev, float* val_out_dev, void* ws_dev,
                   size_t ws_bytes, void* stream) {
  if (!dist_dev || !idx_out_dev || !val_out_dev || !ws_dev) return fail("simsearch_topn: NULL device pointer");
  hipError_t err = ce_topn_smallest(dist_dev, n, top_n, idx_out_dev, val_out_dev, ws_dev, ws_bytes, (hipStream_t)stream);
  if (err != hipSuccess) return fail("simsearch_topn(n=%ld, top_n=%d) failed: %s", n, top_n, hipGetErrorString(err));
  return 0;
}

size_t diversity_workspace(long n) { return n < 1 ? 512 : ce_diversity_workspace_bytes(n); }

int diversity_order(const float* emb_dev, long n, int d, long row_stride, int first, const int* samples_dev, int steps,
                    int sample_size, int* order_dev, void* ws_dev, size_t ws_bytes, void* stream) {
  if (!emb_dev || !order_dev || !ws_dev || (steps > 0 && !samples_dev)) return fail("diversity_order: NULL device pointer");
  if (n < 1 || first < 0 || first >= n) return fail("diversity_order: first index %d outside [0, %ld)", first, n);
  if (steps < 0 || sample_size < 1) return fail("diversity_order: steps %d / sample_size %d", steps, sample_size);
  hipError_t err = ce_diversity_order(emb_dev, n, d, row_stride, first, samples_dev, steps, sample_size, order_dev, ws_dev, ws_bytes,
                                      (hipStream_t)stream);
  if (err != hipSuccess) return fail("diversity_order(n=%ld, d=%d, steps=%d) failed: %s", n, d, steps, hipGetErrorString(err));
  return 0;
}

int clipenc_op_gemm_nt(const void* a_dev, const void* w_dev, int m, int n, int k, int dtype, int epi,
                       const float* bias_dev, void* out_dev, void* stream) {
  if (epi != CLIPENC_EPI_STORE_F32 && epi != CLIPENC_EPI_STORE_BF16) return fail("epi %d not exposed", epi);
  GemmParams p{};
  p.A = a_dev; p.lda = k; p.W = w_dev; p.ldw = k; p.M = m; p.N = n; p.K = k; p.out = out_dev; p.ldo = n; p.bias = bias_dev;
  hipError_t err = ce_gemm_nt(p, dtype, epi, (hipStream_t)stream);
  if (err != hipSuccess) return fail("gemm_nt(%d,%d,%d) failed: %s", m, n, k, hipGetErrorString(err));
  return 0;
}

int clipenc_op_quant_rows_fp8(const void* in_dev, int in_f32, int n_rows, int k, int ln, float eps, void* out8_dev,
                              float* scale_dev, void* stream) {
  if (!in_dev || !out8_dev || !scale_dev) return fail("NULL device pointer");
  if (ln < 0 || ln > 2) return fail("quant_rows_fp8: ln = %d (0 plain, 1 LayerNorm first, 2 plain with power-of-two scales)", ln);
  hipError_t err = ce_quant_rows_fp8(in_dev, in_f32, (size_t)k, out8_dev, (size_t)k, scale_dev, n_rows, k, ln == 1, eps, (hipStream_t)stream,
                                     ln == 2);
  if (err != hipSuccess) return fail("quant_rows_fp8(%d,%d) failed: %s", n_rows, k, hipGetErrorString(err));
  return 0;
}

#ifdef CLIPENC_DIAG
static unsigned long long* g_fp8_stamps = nullptr;   // diagnostic library: [tiles][8] stamps of the next fp8 GEMM ops (tools/gemm_fp8_stamps.py)
int clipenc_diag_fp8_stamps(unsigned long long* stamps_dev) { g_fp8_stamps = stamps_dev; return 0; }
#define FP8_DBG(p) (p).dbg = g_fp8_stamps
#else
#define FP8_DBG(p) ((void)0)
#endif

int clipenc_op_gemm_fp8(const void* a8_dev, const void* w8_dev, int m, int n, int k, const float* scale_a_dev,
                        const float* scale_w_dev, const float* bias_dev, int act, const void* resid_dev, void* out_dev,
                        void* stream) {
  if (resid_dev && act != -1) return fail("gemm_fp8: activation and residual are exclusive");
  GemmParams p{};
  p.A = a8_dev; p.lda = k; p.W = w8_dev; p.ldw = k; p.M = m; p.N = n; p.K = k; p.out = out_dev; p.ldo = n; p.bias = bias_dev;
  p.scale_a = scale_a_dev; p.scale_w = scale_w_dev; p.act = act; p.resid = resid_dev;
  FP8_DBG(p);
  hipError_t err = ce_gemm_fp8(p, resid_dev ? EPI_RESID : EPI_STORE_BF16, (hipStream_t)stream);
  if (err != hipSuccess) return fail("gemm_fp8(%d,%d,%d) failed: %s", m, n, k, hipGetErrorString(err));
  return 0;
}

int clipenc_op_gemm_fp8_q(const void* a8_dev, const void* w8_dev, int m, int n, int k, const float* scale_a_dev,
                          const float* scale_w_dev, const float* bias_dev, int act, const float* out_inv_scale_dev,
                          void* out8_dev, void* stream) {
  GemmParams p{};
  p.A = a8_dev; p.lda = k; p.W = w8_dev; p.ldw = k; p.M = m; p.N = n; p.K = k; p.out = out8_dev; p.ldo = n; p.bias = bias_dev;
  p.scale_a = scale_a_dev; p.scale_w = scale_w_dev; p.act = act; p.out_inv_scale = out_inv_scale_dev;
  FP8_DBG(p);
  hipError_t err = ce_gemm_fp8(p, EPI_STORE_FP8, (hipStream_t)stream);
  if (err != hipSuccess) return fail("gemm_fp8_q(%d,%d,%d) failed: %s", m, n, k, hipGetErrorString(err));
  return 0;
}

int clipenc_op_quant_block_fp8(const void* in_dev, int n_rows, int k, void* out8_dev, void* exp_dev, float* stats_dev,
                               void* stream) {
  if (!in_dev || !out8_dev || !exp_dev) return fail("NULL device pointer");
  hipError_t err = ce_quant_block_fp8(in_dev, (size_t)k, out8_dev, (size_t)k, exp_dev, 4, stats_dev, n_rows, k, (hipStream_t)stream);
  if (err != hipSuccess) return fail("quant_block_fp8(%d,%d) failed: %s", n_rows, k, hipGetErrorString(err));
  return 0;
}

int clipenc_op_row_norm_consts(const float* stats_dev, int parts, int ld, int n_rows, int width, float eps, float* row_r_dev,
                               float* row_d_dev, void* stream) {
  if (!stats_dev || !row_r_dev || !row_d_dev) return fail("NULL device pointer");
  hipError_t err = ce_row_norm_consts(stats_dev, parts, (size_t)ld, n_rows, width, eps, row_r_dev, row_d_dev, 1, (hipStream_t)stream);
  if (err != hipSuccess) return fail("row_norm_consts(%d,%d) failed: %s", parts, n_rows, hipGetErrorString(err));
  return 0;
}

namespace {
// The stand-alone block-exponent GEMM ops take float weight scales like the other fp8 ops; the kernels want their E8M0 bytes
// (gemm.h: w_exp).  Derived here into one scratch per process (grown as needed, never shrunk): these ops are developer / test
// entry points, one caller at a time.
std::mutex g_opexp_mu;
unsigned char* g_opexp = nullptr; size_t g_opexp_cap = 0; int g_opexp_dev = -1;
hipError_t op_weight_exponents(const float* scale_w_dev, int n, hipStream_t st, const unsigned char** out) {
  std::lock_guard<std::mutex> lock(g_opexp_mu);
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
  if (dev != g_opexp_dev || g_opexp_cap < (size_t)n) {
    if (g_opexp) { (void)hipDeviceSynchronize(); (void)hipFree(g_opexp); g_opexp = nullptr; g_opexp_cap = 0; }
    const size_t want = std::max<size_t>((size_t)n, 65536);
    if (hipError_t e = hipMalloc((void**)&g_opexp, want); e != hipSuccess) return e;
    g_opexp_cap = want; g_opexp_dev = dev;
  }
  *out = g_opexp;
  return ce_scale_exponents(scale_w_dev, g_opexp, n, nullptr, st);
}
}  // namespace

int clipenc_op_gemm_fp8_lnf(const void* a8_dev, const void* exp_dev, const void* w8_dev, int m, int n, int k,
                            const float* row_r_dev, const float* row_d_dev, const float* scale_w_dev, const float* colsum_dev,
                            const float* bias_dev, int act, const float* out_inv_scale_dev, void* out_dev, void* stream) {
  if (!exp_dev) return fail("gemm_fp8_lnf: NULL exponent rows");
  GemmParams p{};
  p.A = a8_dev; p.lda = k; p.W = w8_dev; p.ldw = k; p.M = m; p.N = n; p.K = k; p.out = out_dev; p.ldo = n; p.bias = bias_dev;
  p.scale_w = scale_w_dev; p.colsum = colsum_dev; p.act = act; p.out_inv_scale = out_inv_scale_dev;
  if (!scale_w_dev || n < 1) return fail("gemm_fp8_lnf: NULL weight scales");
  HIP_TRY(op_weight_exponents(scale_w_dev, n, (hipStream_t)stream, &p.w_exp));
  p.a_exp = (const unsigned char*)exp_dev; p.ld_aexp = 4; p.row_r = row_r_dev; p.row_d = row_d_dev; p.ld_row = 1;
  FP8_DBG(p);
  hipError_t err = ce_gemm_fp8(p, out_inv_scale_dev ? EPI_STORE_FP8 : EPI_STORE_BF16, (hipStream_t)stream);
  if (err != hipSuccess) return fail("gemm_fp8_lnf(%d,%d,%d) failed: %s", m, n, k, hipGetErrorString(err));
  return 0;
}

int clipenc_op_gemm_fp8_resid_q(const void* a8_dev, const void* w8_dev, int m, int n, int k, const float* scale_w_dev,
                                const float* bias_dev, void* x_inout_dev, void* out8_dev, void* exp_dev, float* stats_dev,
                                int stats_ld, void* stream) {
  GemmParams p{};
  p.A = a8_dev; p.lda = k; p.W = w8_dev; p.ldw = k; p.M = m; p.N = n; p.K = k; p.out = x_inout_dev; p.ldo = n; p.bias = bias_dev;
  p.scale_w = scale_w_dev; p.act = -1; p.resid = x_inout_dev;
  if (!scale_w_dev || n < 1) return fail("gemm_fp8_resid_q: NULL weight scales");
  HIP_TRY(op_weight_exponents(scale_w_dev, n, (hipStream_t)stream, &p.w_exp));
  p.out8 = out8_dev; p.ld8 = n; p.out_exp = (unsigned char*)exp_dev; p.ld_oexp = 4; p.stats_out = stats_dev; p.stats_ld = stats_ld;
  FP8_DBG(p);
  hipError_t err = ce_gemm_fp8(p, EPI_RESID_Q, (hipStream_t)stream);
  if (err != hipSuccess) return fail("gemm_fp8_resid_q(%d,%d,%d) failed: %s", m, n, k, hipGetErrorString(err));
  return 0;
}

int jpegdec_create(int device, jpegdec_t* out) {
  if (!out) return fail("jpegdec_create: NULL argument");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail("device %d out of range (%d visible)", device, ndev);
  jpegdec_s* d = new jpegdec_s();
  d->device = device;
  d->st = ce_jpegdec_create();
  *out = d;
  return 0;
}

int jpegdec_destroy(jpegdec_t d) {
  if (!d) return 0;
  (void)hipSetDevice(d->device);
  ce_jpegdec_destroy(d->st);
  delete d;
  return 0;
}

int jpegdec_plan(jpegdec_t d, const void* const* files, const size_t* sizes, int n, int* status, int* widths, int* heights,
                 unsigned long long* rgb_offsets, unsigned long long* rgb_bytes) {
  if (!d || !files || !sizes || !status || !widths || !heights || !rgb_offsets || !rgb_bytes) return fail("jpegdec_plan: NULL argument");
  if (n < 0 || n > 65535) return fail("jpegdec_plan: %d files (0 .. 65535 per call: one grid row per image)", n);
  ce_jpegdec_plan(d->st, files, sizes, n, status, widths, heights, rgb_offsets, rgb_bytes);
  return 0;
}

int jpegdec_run(jpegdec_t d, void* rgb_dev, int* status, void* stream) {
  if (!d || !status) return fail("jpegdec_run: NULL argument");
  HIP_TRY(hipSetDevice(d->device));
  hipError_t err = ce_jpegdec_run(d->st, rgb_dev, status, (hipStream_t)stream);
  if (err == hipErrorOutOfMemory) { fail("jpegdec_run: no device scratch / page-locked staging for this batch"); return CLIPENC_JPEGDEC_NO_MEMORY; }
  if (err != hipSuccess) return fail("jpegdec_run failed: %s", hipGetErrorString(err));
  return 0;
}

int jpegdec_reserve(jpegdec_t d, unsigned long long scratch_bytes, unsigned long long staging_bytes) {
  if (!d) return fail("jpegdec_reserve: NULL argument");
  HIP_TRY(hipSetDevice(d->device));
  hipError_t err = ce_jpegdec_reserve(d->st, (size_t)scratch_bytes, (size_t)staging_bytes);
  if (err == hipErrorOutOfMemory) { fail("jpegdec_reserve: %llu + %llu bytes not available", scratch_bytes, staging_bytes); return CLIPENC_JPEGDEC_NO_MEMORY; }
  if (err != hipSuccess) return fail("jpegdec_reserve failed: %s", hipGetErrorString(err));
  return 0;
}

int jpegdec_probe(const void* file, size_t size, int* width, int* height, int* n_scans) {
  if (!file) return jpg::JPG_NOT_JPEG;
  static thread_local jpg::ImageDesc d;                       // (7 KiB of tables: not on the caller's stack)
  size_t so = 0, sl = 0;
  jpg::ProgInfo prog;
  const int rc = jpg::parse_jpeg((const uint8_t*)file, size, &d, &so, &sl, &prog);
  if (width) *width = d.width;
  if (height) *height = d.height;
  if (n_scans) *n_scans = prog.scans.empty() ? 1 : (int)prog.scans.size();
  return rc;
}

const char* jpegdec_reason(int code) {
  if (code >= 100) return "invalid or truncated entropy-coded data";
  return jpg::reason_text(code);
}

int preproc_create(int device, preproc_t* out) {
  if (!out) return fail("preproc_create: NULL argument");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail("device %d out of range (%d visible)", device, ndev);
  preproc_s* p = new preproc_s();
  p->device = device;
  p->st = ce_preproc_create();
  *out = p;
  return 0;
}

int preproc_destroy(preproc_t p) {
  if (!p) return 0;
  (void)hipSetDevice(p->device);
  ce_preproc_destroy(p->st);
  delete p;
  return 0;
}

int preproc_crops_u8(preproc_t p, const uint8_t* image_dev, int height, int width, int pitch_bytes, int n_crops,
                     const int* boxes, int out_size, uint8_t* out_dev, void* stream) {
  if (!p) return fail("NULL handle");
  HIP_TRY(hipSetDevice(p->device));
  hipError_t err = ce_preproc_crops_u8(p->st, image_dev, height, width, pitch_bytes, n_crops, boxes, out_size, out_dev,
                                       (hipStream_t)stream);
  if (err != hipSuccess) return fail("preproc_crops_u8(%dx%d, %d crops -> %d) failed: %s", width, height, n_crops, out_size,
                                     hipGetErrorString(err));
  return 0;
}

int preproc_crops_u8_batch(preproc_t p, int n_images, const uint8_t* const* images_dev, const int* heights, const int* widths,
                           const int* pitches_bytes, const int* crops_per_image, const int* boxes, int out_size, uint8_t* out_dev,
                           void* stream) {
  if (!p || !p->st) return fail("NULL handle");
  HIP_TRY(hipSetDevice(p->device));
  hipError_t err = ce_preproc_crops_u8_batch(p->st, n_images, images_dev, heights, widths, pitches_bytes, crops_per_image, boxes,
                                             out_size, out_dev, (hipStream_t)stream);
  if (err != hipSuccess) return fail("preproc_crops_u8_batch(%d images -> %d) failed: %s", n_images, out_size, hipGetErrorString(err));
  return 0;
}

int preproc_axis_tables(int in_size, int out_size, int out0, int n_out, int* bounds, int* kk, int kk_capacity, int* ksize) {
  if (in_size < 1 || out_size < 1 || out0 < 0 || n_out < 1 || out0 + n_out > out_size || !bounds || !kk || !ksize)
    return fail("preproc_axis_tables: bad argument");
  std::vector<int> b, k;
  *ksize = ce_preproc_axis_tables(in_size, out_size, out0, n_out, b, k);
  if ((long)k.size() > kk_capacity) return fail("kk_capacity %d < %ld", kk_capacity, (long)k.size());
  memcpy(bounds, b.data(), b.size() * sizeof(int));
  memcpy(kk, k.data(), k.size() * sizeof(int));
  return 0;
}

#ifdef CLIPENC_DIAG
int clipenc_op_gemm_lnfold(const void* a_dev, const void* w_dev, int m, int n, int k, const float* colsum_dev,
                           const float* bias_dev, const float* stats_dev, int parts, int stats_ld, int act, void* out_dev,
                           unsigned long long* stamps_dev, void* stream) {
  GemmParams p{};
  p.A = a_dev; p.lda = k; p.W = w_dev; p.ldw = k; p.M = m; p.N = n; p.K = k; p.out = out_dev; p.ldo = n;
  p.bias = bias_dev; p.colsum = colsum_dev; p.stats_in = stats_dev; p.stats_in_parts = parts; p.stats_ld = stats_ld;
  p.inv_width = 1.0f / k; p.eps = 1e-5f; p.act = act; p.dbg = stamps_dev;
  hipError_t err = ce_gemm_nt(p, CE_DT_BF16, EPI_LNFOLD, (hipStream_t)stream);
  if (err != hipSuccess) return fail("gemm_lnfold failed: %s", hipGetErrorString(err));
  return 0;
}

int clipenc_op_gemm_resid(const void* a_dev, const void* w_dev, int m, int n, int k, const float* bias_dev, void* x_inout_dev,
                          float* stats_out_dev, int stats_ld, unsigned long long* stamps_dev, void* stream) {
  GemmParams p{};
  p.A = a_dev; p.lda = k; p.W = w_dev; p.ldw = k; p.M = m; p.N = n; p.K = k; p.out = x_inout_dev; p.ldo = n;
  p.bias = bias_dev; p.resid = x_inout_dev; p.stats_out = stats_out_dev; p.stats_ld = stats_ld; p.dbg = stamps_dev;
  hipError_t err = ce_gemm_nt(p, CE_DT_BF16, EPI_RESID, (hipStream_t)stream);
  if (err != hipSuccess) return fail("gemm_resid failed: %s", hipGetErrorString(err));
  return 0;
}

// the bf16-store GEMM with explicit leading dimensions (elements): the row-pitch experiment of tools/gemm_pitch_probe.py
int clipenc_op_gemm_nt_ld(const void* a_dev, int lda, const void* w_dev, int ldw, int m, int n, int k, void* out_dev, int ldo, void* stream) {
  GemmParams p{};
  p.A = a_dev; p.lda = lda; p.W = w_dev; p.ldw = ldw; p.M = m; p.N = n; p.K = k; p.out = out_dev; p.ldo = ldo;
  hipError_t err = ce_gemm_nt(p, CE_DT_BF16, EPI_STORE_BF16, (hipStream_t)stream);
  if (err != hipSuccess) return fail("gemm_nt_ld failed: %s", hipGetErrorString(err));
  return 0;
}

int clipenc_op_gemm_nt_stamps(const void* a_dev, const void* w_dev, int m, int n, int k, void* out_dev,
                              unsigned long long* stamps_dev, void* stream) {
  GemmParams p{};
  p.A = a_dev; p.lda = k; p.W = w_dev; p.ldw = k; p.M = m; p.N = n; p.K = k; p.out = out_dev; p.ldo = n;
  p.dbg = stamps_dev;
  hipError_t err = ce_gemm_nt(p, CE_DT_BF16, EPI_STORE_BF16, (hipStream_t)stream);
  if (err != hipSuccess) return fail("gemm_nt_stamps failed: %s", hipGetErrorString(err));
  return 0;
}
#endif

int clipenc_op_attention(const void* qkv_dev, void* out_dev, int n_crops, int n_tok, int width, int heads, void* stream) {
  hipError_t err = ce_attention(qkv_dev, out_dev, n_crops, n_tok, width, heads, nullptr, 0, (hipStream_t)stream);
  if (err != hipSuccess) return fail("attention(%d crops, %d tok) failed: %s", n_crops, n_tok, hipGetErrorString(err));
  return 0;
}

int clipenc_op_attention_q(const void* qkv_dev, void* out8_dev, int n_crops, int n_tok, int width, int heads,
                           const float* out_inv_scale_dev, void* stream) {
  if (!out_inv_scale_dev) return fail("attention_q: NULL out_inv_scale");
  hipError_t err = ce_attention(qkv_dev, out8_dev, n_crops, n_tok, width, heads, out_inv_scale_dev, 0, (hipStream_t)stream);
  if (err != hipSuccess) return fail("attention_q(%d crops, %d tok) failed: %s", n_crops, n_tok, hipGetErrorString(err));
  return 0;
}

}  // extern "C"
