// fp8 (OCP e4m3) "NT" GEMM for the ViT tower, BASELINE.json configs[3]:  C[M,N] = A8[M,K] . W8[N,K]^T  on the
// block-scaled MFMA v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 MFMA rate) with all hardware block scales set to
// 1.0 (E8M0 0x7f) and the real scales applied in the epilogue:  out = acc * sa[m] * sw[n] + bias[n]
// (per-token activation scale, per-output-channel weight scale).  Operand / result lane maps were established on
// hardware by tools/probes/fp8probe2.hip: lane l holds row l&31 and 32 of its 64 k bytes; D as the 32x32 bf16 form.
// With unit scales any assignment of k bytes to (lane half, register half) is correct as long as both operands use the
// same one (this kernel gives lane half h the bytes [32h, 32h+32)).  The hardware's own order matters only for non-unit
// block scales (tools/probes/fp8probe3-5.hip): bytes 0-15 of lanes 0-31 are k 0..15, bytes 0-15 of lanes 32-63 are
// k 16..31, bytes 16-31 of lanes 0-31 are k 32..47, bytes 16-31 of lanes 32-63 are k 48..63; byte `opsel` of a lane's
// scale register scales block l>>5 (k 32*(l>>5) .. +31) of row l&31 of that operand.
//
// Same structure as gemm_persist.hip (read its header for the pipeline and its ordering argument): persistent 256x256
// tiles, 8 waves as 2(M) x 4(N), two LDS buffers of 128-BYTE row stages (K = 128 fp8 elements per stage) filled by
// LDS-DMA pieces of 8 rows x 128 B = 8 whole cache lines (the earlier 64-byte-row ring issued twice the L1->L2 read
// requests for the same bytes), two phases per stage with the half-phase stagger of the two wave rows, stages 0-1 of
// the next tile in the pipeline before the epilogue's stores, bf16 / e4m3 output through a wave-private LDS image.
// Per stage and wave: 2 phases x 8 MFMAs of 64 cycles.  Buffer image: 1-KiB blocks of 8 rows x 128 B, 16-B chunk c of
// row r at position c ^ ((r >> 1) & 7); a lane reads chunks (4kh + 2h, 4kh + 2h + 1) of its row: conflict-free for the
// 32-row fragments (checked against the ds_read_b128 lane groups).  Every wave stages blocks of ONE parity (block index
// = wave + 8i), so the swizzle term of the block parity is a per-wave constant of the DMA source offset.
//
// Block-exponent rows (LNF = true consumers, EPI 3 producer): the residual stream's e4m3 copy carries one E8M0 exponent byte per
// (row, 256 columns), x ~ a8 * 2^(e - 127), and the consumer hands that byte to the MFMA as the A operand's block scale (both
// hardware K blocks of an MFMA lie inside one 256-column block, so every lane of a row supplies the same byte and the k
// assignment above stays free).  That is what lets the PRODUCING GEMM quantise its own output tile: the exponent comes from the
// tile's own 256 columns, no row-wide reduction across tiles, and the LayerNorm of the row is folded into the consumer's
// epilogue from row statistics (gemm.h).  It replaces the separate LayerNorm-quantise pass over the residual stream
// (quant_fp8.hip: 1.6 GB of HBM traffic per call, twice per block).
// In those instantiations the WEIGHT scales are powers of two (capi.hip quantises the fused tower's weights that way) and ride in
// the same MFMA as the block scale of the weight operand (p.w_exp: one E8M0 byte per output channel, 64 per wave and tile from one
// scalar load): the accumulator comes out scaled by scale_w[n] and the epilogue has no multiply by it.
#include "common.h"
#include "gemm.h"

#ifndef F8_NT_STORES             // 1: the QKV / FC1 outputs (written once, read by another kernel) leave with the streaming cache policy, so that the
                                 // 2-4 MB an XCD round writes stop evicting the e4m3 A panels (2 MB) its next round re-reads: QKV 1.703 -> 1.665 ms, FC1
                                 // 2.242 -> 2.207, e4m3 step +0.9 % (three interleaved pairs; 0: plain stores)
#define F8_NT_STORES 1
#endif
#ifndef F8_MMA_ORDER
#define F8_MMA_ORDER 1
#endif
// developer switches (tools/gemm_variants.sh): F8_LNF_ZERO_C = peeled zero-C first stage pair in the LNF variants too;
// F8_EPI3_PARTS = how much of the quantising epilogue runs (3 all, 2 no e4m3 rows, 1 no exchange either, 0 = EPI 1)
#ifndef F8_LNF_ZERO_C
#define F8_LNF_ZERO_C 1
#endif
#ifndef F8_EPI3_PARTS
#define F8_EPI3_PARTS 3
#endif
#ifndef F8_EPI3_SCALECVT         // 1: the quantising pass of EPI 3 converts with v_cvt_scalef32_pk_fp8_f32 (no multiply per element)
#define F8_EPI3_SCALECVT 1
#endif
#ifndef F8_LNF_DBG               // timing bisection only (results wrong): 1 unit scale operand, 2 no exponent reads either,
#define F8_LNF_DBG 0             // 4 no exponent DMA
#endif
// instantiations with the register room for the peeled zero-C first stage pair (the others spill with it)
#define ZERO_C_SET(EPI, ACT) (((EPI) == 2 && (ACT) <= 0) || (EPI) == 4)

namespace {

constexpr int BM = 256, BN = 256;
constexpr int BUF = 65536, WREG = 32768;    // one K=128 stage: A region | W region
constexpr int RING = 2 * BUF;               // 131072
constexpr int TR_OFF = RING;                // 8 x 4 KiB wave-private images
constexpr int LDS_BYTES = RING + 32768;     // 163840

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) short s16x2_t;

#define LDS_PTR(off) ((__attribute__((address_space(3))) void*)(smem + (off)))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int ACT>
__device__ __forceinline__ float act_apply_t(float u) {
  if constexpr (ACT == CE_ACT_QUICK_GELU) return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * u));
  else if constexpr (ACT == CE_ACT_GELU_ERF) return ce_gelu_erf(u);
  else return u;
}

// One LDS-DMA piece: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset) to 1 KiB of LDS at lds_addr.
// Inline asm on purpose: behind the builtin, LLVM's waitcnt pass treats every later ds_read as possibly aliasing the
// pieces in flight and puts `s_waitcnt vmcnt(0)` in front of the fragment reads of EVERY phase -- a full drain of the
// prefetch pipeline (it did so in this kernel; the pipeline's correctness is the hand-placed counted waits).
// (m0 is written; nothing else in this file uses it.)
__device__ __forceinline__ void glds16_at(const char* base, unsigned off, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_addr) : "memory");
}

// the 4-byte form: 64 lanes x 4 B -> 256 B of LDS at lds_addr (per-row scales, every lane its own clamped row)
__device__ __forceinline__ void glds4_at(const char* base, unsigned off, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(off), "s"(base), "s"(lds_addr) : "memory");
}

struct TileId { int m0, n0; };

__device__ __forceinline__ TileId decode_tile(int idx, int tiles_m, int tiles_n) {
  const int nwg = tiles_m * tiles_n;
  const int q = nwg >> 3, r = nwg & 7, xcd = idx & 7, pos = idx >> 3;
  const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;   // XCD-aware, bijective
  constexpr int GM = 8;
  const int group = bid / (GM * tiles_n);
  const int first_m = group * GM;
  const int gsz = min(tiles_m - first_m, GM);
  const int tm = first_m + (bid % (GM * tiles_n)) % gsz;
  const int tn = (bid % (GM * tiles_n)) / gsz;
  return TileId{tm * BM, tn * BN};
}

// EPI 4 (gemm_fp8_tri.hip): the tiles of the upper triangle in the host-made order of gemm_tri.hip, one scalar load per tile
// (scalar memory counts in lgkmcnt, not in the vmcnt of the DMA pipeline's counted waits)
__device__ __forceinline__ TileId tri_tile(const unsigned* list, int i) {
  unsigned v;
  asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(list), "s"(i * 4) : "memory");
  return TileId{(int)(v & 0xffffu) * BM, (int)(v >> 16) * BN};
}

// EPI: 0 = scale + bias (+ activation) -> bf16;  1 = scale + bias + residual (bf16, in place) -> bf16;
//      2 = scale + bias (+ activation), then * out_inv_scale[n] -> e4m3 (the next GEMM's operand, no bf16 round trip)
//      3 = as 1, and the e4m3 block-exponent copy of the new rows + their (sum, sum of squares) per 64 columns
//      4 = near-duplicate SCREEN (A == W == e4m3 rows of unit vectors x 256, upper-triangular tile list): no output matrix; the
//          (i < j) whose e4m3 product cannot rule them out are appended as candidates (ce_gemm_fp8_tri below)
// LNF: A rows are block-exponent rows (a_exp) and the epilogue applies the folded LayerNorm (row_r, row_d, colsum)
template <int EPI, int ACT, bool LNF>
__global__ __launch_bounds__(512, 2) void gemm_fp8_kernel(const GemmParams p) {
  constexpr bool ZERO_C_OK = ZERO_C_SET(EPI, ACT) && (!LNF || F8_LNF_ZERO_C);
  constexpr bool RES = EPI == 1 || EPI == 3;
  // the fused tower's instantiations: the weight scales are powers of two and ride in the MFMA as the weight rows' block scale
  // (p.w_exp, one E8M0 byte per output channel), so the epilogue has no multiply by scale_w
  constexpr bool WEXP = LNF || EPI == 3;
  // wave image, LNF: [0,256) weight scales, [256,512) column sums, [512,768) biases, [1024,1536) row_r, [1536,2048) row_d,
  // [2560,2816) inverse output scales (EPI 2), [3072,3584) the NEXT tile's exponent dwords (row-major, landed by DMA during this
  // tile's last two stages), [3584,4096) this tile's exponent dwords (read by every phase)
  constexpr int E_NEXT = 3072, E_CUR = 3584;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // EPI 2 writes e4m3 with a static per-column scale.  v_cvt_pk_fp8_f32 turns a finite value beyond +-448 into NaN by default; with
  // MODE.FP16_OVFL set it SATURATES to +-448 instead (tools/probes/fp8_ovfl_probe.hip: 470, 1000, 1e9 -> 0x7e; NaN stays NaN,
  // +-inf -> NaN), which is what the v_med3_f32 in front of every conversion was for -- one VALU instruction per element of an
  // epilogue that is bound by VALU issue (F8_OVFL_MODE 0: the explicit clamp).  The MODE register is per wave and set up anew for
  // every launch; nothing else in this kernel converts to a 16-bit float type that the bit would touch.
#ifndef F8_OVFL_MODE
#define F8_OVFL_MODE 1
#endif
#if F8_OVFL_MODE
  if constexpr (EPI == 2) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
#define F8_SAT(x) (x)
#else
#define F8_SAT(x) __builtin_amdgcn_fmed3f((x), -448.0f, 448.0f)
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int r32 = lane & 31, h = lane >> 5;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  const int nwg = EPI == 4 ? tiles_n * (tiles_n + 1) / 2 : tiles_m * tiles_n;
#define DECODE_TILE(i_) (EPI == 4 ? tri_tile(p.tile_list, (i_)) : decode_tile((i_), tiles_m, tiles_n))
  const int G = gridDim.x;
  // Dynamic tail (p.ticket; gemm_persist.hip has the argument and the measurements): the tiles from S on -- the launch's last
  // one-to-two rounds -- are taken from a global ticket counter (a scalar atomic: lgkmcnt, not the DMA pipeline's vmcnt) in the
  // order the workgroups get there.  The ticket waits in word 512 of wave 4's image: [2048, 2560) of an image is written by no
  // DMA piece of any instantiation, and the epilogue, which owns the whole image, starts after the ticket has been read.
#ifndef F8_DYN_ROUNDS
#define F8_DYN_ROUNDS 1
#endif
  // (outputs wider than 4 tiles: one counter per XCD + stealing, two rounds; else one counter, one round -- gemm_persist.hip)
  const bool by_xcd = tiles_n > 4 && (G & 7) == 0;
  const int dyn_rounds = F8_DYN_ROUNDS == 0 ? 0 : (by_xcd ? 2 : 1);
  const int S = (dyn_rounds > 0 && EPI != 4 && p.ticket != nullptr && nwg / G >= dyn_rounds + 2) ? (nwg / G - dyn_rounds) * G : 0x7fffffff;
  constexpr int TICKET_SLOT = TR_OFF + 4 * 4096 + 2048;
  const size_t lda_b = (size_t)p.lda, ldw_b = (size_t)p.ldw;      // fp8: 1 byte per element
  const int kend = p.K;                      // bytes along K; one stage = 128 B; K % 256 == 0 (stages come in pairs)

  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(0));   // LDS address of smem[0]
#define glds16(base, off, smem_, lds_off) glds16_at((base), (off), lds0 + (unsigned)(lds_off))
  // LDS-DMA: lane L fetches logical chunk (L&7) ^ f of row L>>3 of an 8-row block, f = (block parity << 2) | (row >> 1);
  // wave w stages the A blocks i*16 + h*8 + w (tile rows i*128 + h*64 + 8w, i = 0,1 per row half h) and the W blocks
  // w + 8i (rows 8w + 64i, i = 0..3): all of parity w & 1
  const int dg = lane >> 3;
  const unsigned dchunk16 = (unsigned)(((lane & 7) ^ (((w & 1) << 2) | (dg >> 1))) * 16);
  const int arow0 = 8 * w + dg;                          // + i*128 + h*64
  const int a_dma = w * 1024;                            // + (i*16 + h*8)*1024 (+ buffer)
  const unsigned woff = (unsigned)((8 * w + dg) * ldw_b) + dchunk16;   // + i*64*ldw_b through the scalar base
  const int w_dma = WREG + w * 1024;                     // + i*8192 (+ buffer)
  // fragment of a 32-row tile: lane (r32, h) reads the 16-B chunks 4kh + 2h and 4kh + 2h + 1 of row r32
  const int rd0 = (r32 >> 3) * 1024 + (r32 & 7) * 128 + ((((2 * h) ^ ((r32 >> 1) & 7))) << 4);   // kh = 0; second chunk ^16, kh = 1 ^64
  const int a_base = wr * 16 * 1024;                     // + buffer + mt*4096
  const int w_base = WREG + wc * 8 * 1024;               // + buffer + nt*4096

  // epilogue image: [32 rows][128 B] per wave, 16-B chunk index XOR row&7 (its per-lane addresses are derived in the epilogue)
  char* tr = smem + TR_OFF + w * 4096;
#define TW_ADDR(c) (tr + tw_base + ((((c)) ^ tw_sw) << 4))

  int idx = blockIdx.x;
  TileId cur = DECODE_TILE(idx);
  const char* Ablk = (const char*)p.A + (size_t)cur.m0 * lda_b;
  const char* Wblk = (const char*)p.W + (size_t)cur.n0 * ldw_b;
#define AOFF(m0v, r) ((unsigned)((min((m0v) + (r), p.M - 1) - (m0v)) * lda_b) + dchunk16)
  unsigned aoff00 = AOFF(cur.m0, arow0), aoff01 = AOFF(cur.m0, arow0 + 128);          // half 0, i = 0,1
  unsigned aoff10 = AOFF(cur.m0, arow0 + 64), aoff11 = AOFF(cur.m0, arow0 + 192);     // half 1

#define ISSUE_AH0(b, blk, o0, o1, kbyte)                                                    \
  do {                                                                                      \
    glds16((blk) + (kbyte), (o0), smem, (b) * BUF + a_dma);                                 \
    glds16((blk) + (kbyte), (o1), smem, (b) * BUF + a_dma + 16384);                         \
  } while (0)
#define ISSUE_AH1(b, blk, o0, o1, kbyte)                                                    \
  do {                                                                                      \
    glds16((blk) + (kbyte), (o0), smem, (b) * BUF + a_dma + 8192);                          \
    glds16((blk) + (kbyte), (o1), smem, (b) * BUF + a_dma + 24576);                         \
  } while (0)
#define ISSUE_W(b, blk, kbyte)                                                              \
  do {                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                        \
      glds16((blk) + (kbyte) + (size_t)i_ * 64 * ldw_b, woff, smem, (b) * BUF + w_dma + i_ * 8192); \
  } while (0)
#define LD_FRAG(dst, off, kh)                                                               \
  do {                                                                                      \
    const uint4 lo_ = *(const uint4*)(smem + (off) + (rd0 ^ ((kh) * 64)));                  \
    const uint4 hi_ = *(const uint4*)(smem + (off) + (rd0 ^ ((kh) * 64) ^ 16));             \
    dst = i32x8_t{(int)lo_.x, (int)lo_.y, (int)lo_.z, (int)lo_.w, (int)hi_.x, (int)hi_.y, (int)hi_.z, (int)hi_.w}; \
  } while (0)
  // fragments of one phase (gemm_persist.hip): W for both k halves (fb[2 kh + j], kept for both phases of the stage) and
  // one A row half for both k halves (fa[2 kh + i])
#define LD_W2(b) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) _Pragma("unroll") for (int j = 0; j < 2; ++j) LD_FRAG(fb[kh * 2 + j], (b) * BUF + w_base + j * 4096, kh);
#define LD_A2(b, half) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) _Pragma("unroll") for (int i = 0; i < 2; ++i) LD_FRAG(fa[kh * 2 + i], (b) * BUF + a_base + ((half) * 2 + i) * 4096, kh);
  // LNF: the exponent byte of this stage pair's 256-column block for the two 32-row tiles of the phase (byte 0 of the register is
  // what opsel 0 selects; sh = 8 * block index)
#define LD_SC(half, sh)                                                                     \
  do {                                                                                      \
    if constexpr (LNF && !(F8_LNF_DBG & 2)) {                                               \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                         \
        sc[i] = (int)(*(const unsigned*)(tr + E_CUR + (((half) * 2 + i) * 32 + r32) * 4) >> (sh)); \
    }                                                                                       \
  } while (0)
  // ZC: the first MFMA of every accumulator of a tile takes the constant 0 as C (gemm_persist.hip)
#define MMA(half, ZC)                                                                       \
  do {                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                          \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                      \
      const int j = (F8_MMA_ORDER && (i & 1)) ? 1 - j_ : j_;      /* serpentine: one operand register set changes per MFMA (gemm_persist.hip) */ \
      acc[(half) * 2 + i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[kh * 2 + j], fa[kh * 2 + i],      \
                                   ((ZC) && kh == 0) ? zero16 : acc[(half) * 2 + i][j], 0, 0,                        \
                                                                  0, WEXP ? wsc[j] : 0x7f7f7f7f, 0, (LNF && !(F8_LNF_DBG & 1)) ? sc[i] : 0x7f7f7f7f); \
    }                                                                                       \
    __builtin_amdgcn_s_setprio(0);                                                          \
  } while (0)
#define BARRIER() asm volatile("s_barrier" ::: "memory")
#define WAIT_LDS()                                                                          \
  do {                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
#define SYNC_MMA(half, ZC)                                                                  \
  do {                                                                                      \
    WAIT_LDS(); BARRIER(); __builtin_amdgcn_sched_barrier(0); MMA(half, ZC); BARRIER();     \
  } while (0)
#define ISSUE_WAH0(b, ablk, wblk, o00, o01, kbyte)                                          \
  do { ISSUE_W(b, wblk, kbyte); ISSUE_AH0(b, ablk, o00, o01, kbyte); } while (0)
  // one K=128 stage on buffer b = two phases of 8 MFMAs (512 cycles) per wave (gemm_persist.hip).  PA: W (both k halves)
  // and A(half 0), PA_ISSUE = the A(half 1) rows of stage s+1 into the other buffer; PB: A(half 1), PB_ISSUE = W and
  // A(half 0) of stage s+2 into this one; each phase's counted wait sits behind its own pieces (the 8 newest stay in flight)
  // (sh: 8 * index of the stage's 256-column block, used by the LNF variants only)
#define STAGE(b, sh, VMWAIT, PA_ISSUE, PB_ISSUE) STAGE_Z(b, sh, 0, VMWAIT, PA_ISSUE, PB_ISSUE)
#define STAGE_Z(b, sh, ZC, VMWAIT, PA_ISSUE, PB_ISSUE)                                      \
  do {                                                                                      \
    LD_W2(b) __builtin_amdgcn_sched_barrier(0); LD_A2(b, 0) LD_SC(0, sh);                   \
    PA_ISSUE;                                                                               \
    VMWAIT;                                                                                 \
    SYNC_MMA(0, ZC);                                                                        \
    LD_A2(b, 1) LD_SC(1, sh);                                                               \
    PB_ISSUE;                                                                               \
    VMWAIT;                                                                                 \
    SYNC_MMA(1, ZC);                                                                        \
  } while (0)
#define VM8 asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
  // First two waits after an epilogue: what they need was issued before the epilogue's stores and vmcnt retires in order,
  // so they may leave the NST row stores of a complete tile outstanding as well (see gemm_persist.hip); one opaque
  // instruction for the compiler.
  // stores per wave and tile.  EPI 3: 16 bf16 rows + 8 e4m3 rows in every wave; the 4 row-statistics stores (one (sum, sum of squares)
  // per row and 256 columns, reduced over the four wave columns through LDS) and the 4 exponent-byte stores are issued by wave
  // column 0 alone, so its count is NST + NST_WC0 (the relaxed wait must name the count of THIS wave: one too many lets a DMA piece
  // of the new tile be read before it has landed)
  constexpr int NST = EPI == 2 ? 8 : EPI == 3 ? (F8_EPI3_PARTS >= 3 ? 24 : 16) : 16;
  constexpr int NST_WC0 = (EPI == 3 && F8_EPI3_PARTS >= 2) ? 8 : 0;
#define VM_RELAX                                                                            \
  do {                                                                                      \
    const int sel_ = __builtin_amdgcn_readfirstlane(relax > 0 ? ((NST_WC0 > 0 && wc == 0) ? 2 : 1) : 0);   \
    relax = relax > 0 ? relax - 1 : 0;                                                      \
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lf8vm8_%=\n\ts_cmp_eq_u32 %0, 1\n\ts_cbranch_scc1 .Lf8vma_%=\n\t"       \
                 "s_waitcnt vmcnt(%2)\n\ts_branch .Lf8end_%=\n.Lf8vma_%=:\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Lf8end_%=\n"       \
                 ".Lf8vm8_%=:\n\ts_waitcnt vmcnt(8)\n.Lf8end_%=:" : : "s"(sel_), "n"(8 + NST + CB_PIECES), "n"(8 + NST + NST_WC0 + CB_PIECES) : "memory", "scc");    \
  } while (0)
  // the column constants (weight scales, biases, EPI 2: inverse output scales) of the tile are staged into the wave's image at
  // the top of the tile (below): those pieces sit between the previous tile's stores and the first waits (the per-row
  // scale pieces, issued only with per-token scales, are not counted: the wait is then two pieces stricter than need be)
  constexpr int CB_PIECES = EPI == 4 ? 0 : LNF ? (EPI == 2 ? 7 : 6) : EPI == 3 ? 1 : (EPI == 2 ? 3 : 2);

  // ---- cold prologue of the first tile ----
  // E_ROWS(m0v, dst): the exponent dwords of the wave's 128 rows of the tile at m0v, two DMA pieces (lane = row)
#define E_ROWS(m0v, dst, lane_v)                                                                                       \
  do {                                                                                                                 \
    const int r0_ = (m0v) + wr * 128 + (lane_v);                                                                       \
    glds4_at((const char*)p.a_exp, (unsigned)min(r0_, p.M - 1) * (unsigned)p.ld_aexp, lds0 + (unsigned)(TR_OFF + w * 4096 + (dst))); \
    glds4_at((const char*)p.a_exp, (unsigned)min(r0_ + 64, p.M - 1) * (unsigned)p.ld_aexp, lds0 + (unsigned)(TR_OFF + w * 4096 + (dst) + 256)); \
  } while (0)
  if constexpr (LNF) E_ROWS(cur.m0, E_CUR, lane);             // oldest pieces: landed with the first counted wait below
  ISSUE_WAH0(0, Ablk, Wblk, aoff00, aoff01, 0); ISSUE_AH1(0, Ablk, aoff10, aoff11, 0);
  ISSUE_WAH0(1, Ablk, Wblk, aoff00, aoff01, 128);           // A(half 1) of stage 1 follows in PA of stage 0
  VM8;                                                      // W and A(half 0) of stage 0 have landed
  int relax = 0;                             // waits of the coming tile that may leave the previous tile's stores in flight
  BARRIER();

  for (;;) {
    f32x16_t acc[4][2];
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    i32x8_t fa[4], fb[4];
    [[maybe_unused]] int sc[2];
    [[maybe_unused]] int wsc[2] = {0x7f7f7f7f, 0x7f7f7f7f};
    if constexpr (WEXP) {
      // the exponent bytes of the wave's 64 weight rows: 64 consecutive bytes at a wave-uniform address, fetched as scalar loads
      // (lgkmcnt, not the vmcnt of the DMA pipeline) and picked per lane: lane (r, h) supplies row r of its 32-row fragment,
      // byte 0 of the register is what opsel 0 selects
      typedef __attribute__((address_space(4))) const u32x4_t cu32x4_t;
      const cu32x4_t* ep = (const cu32x4_t*)(uintptr_t)(p.w_exp + cur.n0 + wc * 64);
      const u32x4_t q0 = ep[0], q1 = ep[1], q2 = ep[2], q3 = ep[3];
      int lane_w;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_w));
      const int rw = lane_w & 31, bsh = (rw & 3) * 8;
      // dword rw >> 2 of the eight: branch-free selects (the plain ?: form compiled into exec-masked branches)
      const unsigned m1 = 0u - ((unsigned)(rw >> 2) & 1u), m2 = 0u - ((unsigned)(rw >> 3) & 1u), m4 = 0u - ((unsigned)(rw >> 4) & 1u);
#define SEL_(x, y, m) ((x) ^ (((x) ^ (y)) & (m)))
#define PICK_(a, b) ((int)(SEL_(SEL_(SEL_((a)[0], (a)[1], m1), SEL_((a)[2], (a)[3], m1), m2),                                   \
                                SEL_(SEL_((b)[0], (b)[1], m1), SEL_((b)[2], (b)[3], m1), m2), m4) >> bsh))
      wsc[0] = PICK_(q0, q1); wsc[1] = PICK_(q2, q3);
#undef PICK_
#undef SEL_
    }

    // Column constants of this tile's 64 columns per wave and the per-token scales of its 128 rows, by LDS-DMA into the wave's
    // (idle until the epilogue) 4-KiB image: [0, 256) weight scales, [1024, 1280) biases, [2560, 2816) inverse output scales
    // (EPI 2), [3584, 4096) per-row scales.  As global loads at the head of the epilogue they made the first block wait for
    // every older DMA piece of the next tile (vmcnt retires in order) -- see gemm_persist.hip, same change, same argument:
    // extra pieces only make the counted waits stricter, and they have retired before the epilogue.
    if constexpr (EPI != 4) {
      int lane_t;                                              // the lane id from the hardware (not kept across the main loop)
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
      const unsigned ti = lds0 + (unsigned)(TR_OFF + w * 4096);
      if constexpr (LNF) {
        // one dword per lane = the wave's 64 columns / 64 of its 128 rows per piece
        const unsigned coff4 = (unsigned)(cur.n0 + wc * 64 + lane_t) * 4u;
        glds4_at((const char*)p.colsum, coff4, ti + 256);
        glds4_at((const char*)p.bias, coff4, ti + 512);
        if constexpr (EPI == 2) glds4_at((const char*)p.out_inv_scale, coff4, ti + 2560);
        const int r0 = cur.m0 + wr * 128 + lane_t;
        const unsigned ro0 = (unsigned)min(r0, p.M - 1) * (unsigned)p.ld_row * 4u, ro1 = (unsigned)min(r0 + 64, p.M - 1) * (unsigned)p.ld_row * 4u;
        glds4_at((const char*)p.row_r, ro0, ti + 1024);
        glds4_at((const char*)p.row_r, ro1, ti + 1280);
        glds4_at((const char*)p.row_d, ro0, ti + 1536);
        glds4_at((const char*)p.row_d, ro1, ti + 1792);
      } else {
        const unsigned coff = (unsigned)(cur.n0 + wc * 64) * 4u + (unsigned)(lane_t & 15) * 16u;
        if constexpr (!WEXP) glds16_at((const char*)p.scale_w, coff, ti);
        glds16_at((const char*)p.bias, coff, ti + 1024);
        if constexpr (EPI == 2) glds16_at((const char*)p.out_inv_scale, coff, ti + 2560);
        if (p.scale_a) {
          const int r0 = cur.m0 + wr * 128 + lane_t;
          glds4_at((const char*)p.scale_a, (unsigned)min(r0, p.M - 1) * 4u, ti + 3584);
          glds4_at((const char*)p.scale_a, (unsigned)min(r0 + 64, p.M - 1) * 4u, ti + 3840);
        }
      }
    }
#ifdef CLIPENC_DIAG
    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 1] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 4] = __builtin_amdgcn_s_memtime(); }
#endif
    if (wr == 1) {
      if (EPI != 4 && w == 4 && idx + G >= S) {             // the tile after this one comes from the ticket counter
        int cand = -1;
        if (by_xcd) {
          for (int j = 0; j < 8 && cand < 0; ++j) {
            const int x = (blockIdx.x + j) & 7;
            unsigned tk = 1;
            asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(tk) : "s"(p.ticket + x) : "memory");
            const int id = S + (int)tk * 8 + x;
            if ((unsigned)id < (unsigned)nwg) cand = id;
          }
        } else {
          unsigned tk = 1;
          asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(tk) : "s"(p.ticket) : "memory");
          cand = S + (int)tk;
        }
        if (lane == 0) *(volatile __attribute__((address_space(3))) int*)LDS_PTR(TICKET_SLOT) = cand;
      }
      BARRIER();                             // second wave row runs half a phase behind
    }

    constexpr bool ZERO_C = ZERO_C_OK;
    if (ZERO_C && kend > 256) {
      // first stage pair of the tile: every accumulator starts from the constant 0 in its first MFMA
      STAGE_Z(0, 0, 1, VM_RELAX, ISSUE_AH1(1, Ablk, aoff10, aoff11, 128), ISSUE_WAH0(0, Ablk, Wblk, aoff00, aoff01, 256));
      STAGE(1, 0, VM8, ISSUE_AH1(0, Ablk, aoff10, aoff11, 256), ISSUE_WAH0(1, Ablk, Wblk, aoff00, aoff01, 384));
      for (int kb = 256; kb < kend - 256; kb += 256) {
        STAGE(0, kb >> 5, VM8, ISSUE_AH1(1, Ablk, aoff10, aoff11, kb + 128), ISSUE_WAH0(0, Ablk, Wblk, aoff00, aoff01, kb + 256));
        STAGE(1, kb >> 5, VM8, ISSUE_AH1(0, Ablk, aoff10, aoff11, kb + 256), ISSUE_WAH0(1, Ablk, Wblk, aoff00, aoff01, kb + 384));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16;
      if constexpr (!ZERO_C) {
        for (int kb = 0; kb < kend - 256; kb += 256) {
          STAGE(0, kb >> 5, VM_RELAX, ISSUE_AH1(1, Ablk, aoff10, aoff11, kb + 128), ISSUE_WAH0(0, Ablk, Wblk, aoff00, aoff01, kb + 256));
          STAGE(1, kb >> 5, VM8, ISSUE_AH1(0, Ablk, aoff10, aoff11, kb + 256), ISSUE_WAH0(1, Ablk, Wblk, aoff00, aoff01, kb + 384));
        }
      }
    }
    // ---- last two stages: the DMA crosses into the next tile (or re-fetches this one into dead buffers) ----
    int nidx = idx + G;
    if (EPI != 4 && nidx >= S)               // (wave-uniform) the ticket taken at the top of this tile
      nidx = __builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) const int*)LDS_PTR(TICKET_SLOT));   // (-1: nothing left)
    const bool has_next = (unsigned)nidx < (unsigned)nwg;
    TileId nxt = cur;
    const char *Anext = Ablk, *Wnext = Wblk;
    if (has_next) {
      nxt = DECODE_TILE(nidx);
      Anext = (const char*)p.A + (size_t)nxt.m0 * lda_b;
      Wnext = (const char*)p.W + (size_t)nxt.n0 * ldw_b;
    }
    // the next tile's A offsets take over this tile's registers as they die: half 0 here, half 1 between the last two
    // stages (without a next tile nxt == cur and they are recomputed to the same values)
    // (re-derived from the hardware's lane id: kept across the main loop these constants -- and the lane id itself -- are spilled, and a spill
    // reload between two stages carries a compiler-counted vmcnt wait that drains the DMA pipeline)
    int lane_b;                                             // the lane id, from the hardware (an asm statement is not hoisted)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_b));
    const int dg_b = lane_b >> 3;
    const unsigned dch_b = (unsigned)(((lane_b & 7) ^ (((w & 1) << 2) | (dg_b >> 1))) * 16);
    const int arow_b = 8 * w + dg_b;
#define AOFF_B(r) ((unsigned)((min(nxt.m0 + (r), p.M - 1) - nxt.m0) * lda_b) + dch_b)
    aoff00 = AOFF_B(arow_b); aoff01 = AOFF_B(arow_b + 128);
    // LNF: the next tile's exponent dwords -> E_NEXT.  Older than every piece of the last two stages, so the final counted wait
    // of the main loop (all but the 8 newest pieces) covers them: the epilogue reads them without a wait of its own.
    if constexpr (LNF && !(F8_LNF_DBG & 4)) E_ROWS(nxt.m0, E_NEXT, lane_b);
    {
      const int kb = kend - 256;
      STAGE(0, kb >> 5, VM_RELAX, ISSUE_AH1(1, Ablk, aoff10, aoff11, kb + 128), ISSUE_WAH0(0, Anext, Wnext, aoff00, aoff01, 0));
      aoff10 = AOFF_B(arow_b + 64); aoff11 = AOFF_B(arow_b + 192);
      STAGE(1, kb >> 5, VM8, ISSUE_AH1(0, Anext, aoff10, aoff11, 0), ISSUE_WAH0(1, Anext, Wnext, aoff00, aoff01, 128));
    }
#undef AOFF_B
    // pin the accumulators here: without a use in this block LLVM sinks the tail's 32 MFMAs below the conditional
    // barrier (all fragments live at once -> hundreds of spilled VGPRs)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][j]));
    if (wr == 0) BARRIER();                  // re-align the two wave rows for the epilogue
#ifdef CLIPENC_DIAG
    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 2] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 5] = __builtin_amdgcn_s_memtime(); }
#endif

    // ------------------------------- epilogue of tile `cur` -------------------------------
    if constexpr (EPI == 4) {
      // Screen (dedup.hip has the derivation): acc = 65536 * (q_i . q_j) with q = e4m3(256 e_hat); the true cosine differs from
      // acc / 65536 by at most 1.1 (err_i + err_j), err = the rows' own quantisation error norms.  p.scale_a[r] holds row r's
      // share of that margin in accumulator units, p.thr the smallest cosine the exact rule could still accept.  First one
      // running maximum against the worst-case margin (2 x 1.1 x 0.0665: rows with a larger error divert the call, dedup.hip), then -- rarely -- the test per pair.
      int lane_e;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
      const int r32e = lane_e & 31, he = lane_e >> 5;
      float vmax = -1e30f;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int q = 0; q < 8; ++q) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(vmax) : "v"(acc[mt][nt][2 * q]), "v"(acc[mt][nt][2 * q + 1]));
      const float thr_acc = p.thr * 65536.0f;
      if (__builtin_amdgcn_ballot_w64(vmax > thr_acc - 0.147f * 65536.0f) != 0ull) {
        const float* marg = p.scale_a;
        uint2* cand = (uint2*)p.pairs;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int i = cur.m0 + wr * 128 + mt * 32 + r32e;
          const float mi = marg[i];
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int j0 = cur.n0 + wc * 64 + nt * 32 + 8 * g + 4 * he;
              const f32x4_t mj = *(const f32x4_t*)(marg + j0);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int j = j0 + e;
                if (j > i && j < p.n_valid && acc[mt][nt][4 * g + e] + mi + mj[e] > thr_acc) {
                  const unsigned long long slot = atomicAdd(p.count, 1ull);
                  if (slot < p.cap) cand[slot] = uint2{(unsigned)i, (unsigned)j};
                }
              }
            }
        }
      }
    } else {
    // lane (r32, h) of m-tile mt holds row mw0 + mt*32 + r32, columns nb + nt*32 + 8g + 4h + (0..3) in acc[mt][nt][4g..4g+3]
    // (every per-lane constant of the epilogue comes from the hardware's lane id HERE: derived at the top of the kernel they are
    // loop invariants that hipcc keeps across the main loop -- in scratch, reloaded below behind a vmcnt(0) that drains the DMA
    // pipeline and the tile's stores)
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int r32e = lane_e & 31, he = lane_e >> 5;
    const int tw_base = r32e * 128 + he * 8;
    const int tw_sw = r32e & 7;
    const int tr_base = (lane_e >> 3) * 128 + (((lane_e & 7) ^ (lane_e >> 3)) << 4);   // + k*1024: rows 8k + (lane>>3)
    const int row_l = lane_e >> 3;
    const int mw0 = cur.m0 + wr * 128;
    const int nb = cur.n0 + wc * 64;
    const size_t gcol = (size_t)nb + (lane_e & 7) * 8;
    uint4 rres[4];                           // residual rows: 4 x 16 B per 32-row block, one block in flight (the first block's
                                             // loads are issued here: their latency passes under pass 1)
#define LOAD_RES(k)                                                                           \
  do {                                                                                        \
    const int m_ = mw0 + (k) * 8 + row_l;                                                     \
    rres[(k) & 3] = uint4{0, 0, 0, 0};                                                        \
    if (m_ < p.M) rres[(k) & 3] = *(const uint4*)((const bf16_t*)p.resid + (size_t)m_ * p.ldo + gcol); \
  } while (0)
    if constexpr (RES) {
#pragma unroll
      for (int k = 0; k < 4; ++k) LOAD_RES(k);
    }
    // The bias and the folded mean term are rank-1 updates of the accumulator and go to the matrix pipe, idle in the epilogue:
    //   out = sa[m] sw[n] (acc + rowA[m] colA[n] + rowB[m] colB[n])
    //   LNF:   sa = row_r, rowA = row_d / row_r (= -mean), colA = colsum / sw, rowB = 1 / row_r, colB = bias / sw
    //   else:  sa = scale_a (or 1), rowA = 0, rowB = 1 / sa, colB = bias / sw
    // as ONE bf16 MFMA of 32 cycles per accumulator: every factor is split into bf16 (hi, lo), the k slots 0-3 hold the four
    // products of the first term and 4-7 those of the second (relative error 2^-16 of each term; fp32 accumulation).  As VALU
    // work the same update is 2 instructions per element and wave -- 256 per lane and tile at ~5 cycles each, two waves per
    // SIMD: the largest single item of the epilogue.  What remains of pass 1 is acc *= sw[n].
    float sa[4];
    [[maybe_unused]] unsigned e_hop[4];                          // LNF: the next tile's exponent dwords, E_NEXT -> registers -> E_CUR
    {
      constexpr int SW_OFF = 0, CS_OFF = 256, BS_OFF = LNF ? 512 : 1024;
      auto split = [](float x) -> unsigned {                     // bf16 hi | lo << 16 with hi + lo = x to 2^-16
        const float hi = __uint_as_float(pack_bf16x2(x, 0.f) << 16);
        return pack_bf16x2(x, x - hi);
      };
      u32x4_t rowf[4], colf[2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float ra = 0.f, rb;
        if constexpr (LNF) {
          sa[mt] = *(const float*)(tr + 1024 + (mt * 32 + r32e) * 4);
          rb = __builtin_amdgcn_rcpf(sa[mt]);
          ra = *(const float*)(tr + 1536 + (mt * 32 + r32e) * 4) * rb;
          e_hop[mt] = *(const unsigned*)(tr + E_NEXT + (mt * 32 + r32e) * 4);
        } else {
          sa[mt] = p.scale_a ? *(const float*)(tr + 3584 + (mt * 32 + r32e) * 4) : 1.0f;
          rb = __builtin_amdgcn_rcpf(sa[mt]);
        }
        const unsigned wa = he == 0 ? split(ra) : 0u, wb = he == 0 ? split(rb) : 0u;   // k slots 8-15 (lanes 32-63): zeros
        rowf[mt] = u32x4_t{wa, wa, wb, wb};
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int n = nt * 32 + r32e;                             // the lane's column of this 32-column tile
        float isw = 1.0f;                                        // WEXP: the accumulator already carries the weight scale
        if constexpr (!WEXP) {
          const float swn = *(const float*)(tr + SW_OFF + n * 4);
          isw = swn != 0.f ? __builtin_amdgcn_rcpf(swn) : 0.f;   // (a zero weight scale zeroes the column, bias included)
        }
        const float cb = *(const float*)(tr + BS_OFF + n * 4) * isw;
        float ca = 0.f;
        if constexpr (LNF) ca = *(const float*)(tr + CS_OFF + n * 4) * isw;
        const unsigned sca = split(ca), scb = split(cb);
        // [hi, hi, lo, lo] against the rows' [hi, lo, hi, lo]
        const unsigned z = he == 0 ? 0xffffffffu : 0u;
        colf[nt] = u32x4_t{((sca & 0xffffu) * 0x10001u) & z, ((sca >> 16) * 0x10001u) & z, ((scb & 0xffffu) * 0x10001u) & z, ((scb >> 16) * 0x10001u) & z};
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, colf[nt]), __builtin_bit_cast(bf16x8_t, rowf[mt]),
                                                                acc[mt][nt], 0, 0, 0);
      // pass 1 (keeps registers low): acc <- acc * sw[n], so that pass 2 only multiplies by sa[m]; the scales of column group
      // c + 1 are read before group c is computed (one LDS latency per tile, not eight)
#define NL(c) (((c) >> 2) * 32 + ((c) & 3) * 8 + he * 4)          /* column within the wave's 64 */
      if constexpr (!WEXP) {
      f32x4_t sw_n = *(const f32x4_t*)(tr + SW_OFF + NL(0) * 4);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4_t sw = sw_n;
        if (c + 1 < 8) sw_n = *(const f32x4_t*)(tr + SW_OFF + NL(c + 1) * 4);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mt][c >> 2][(c & 3) * 4 + e] *= sw[e];
        }
        // the results exist HERE: without a use hipcc sinks this arithmetic into pass 2 and keeps the constants live until then
        if ((c & 3) == 3) asm volatile("" : "+v"(acc[0][c >> 2]), "+v"(acc[1][c >> 2]), "+v"(acc[2][c >> 2]), "+v"(acc[3][c >> 2]));
      }
      }
#undef NL
    }
    // EPI 3: per 32-row block the row's max |x| over the wave's 64 columns (before the bf16 rounding), and the (sum, sum of squares)
    // of the row's 64 ROUNDED values -- kept until the four wave columns exchange them below
    [[maybe_unused]] float amax[4] = {0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] float rs4[4] = {0.f, 0.f, 0.f, 0.f}, rss4[4] = {0.f, 0.f, 0.f, 0.f};

    if constexpr (EPI == 2) {
      // fp8 image: [32 rows][80 B pitch] per wave (64 data bytes; the pitch keeps the dword writes 2-way conflicted at most),
      // the wave's 64 inverse output scales behind it
      float* isc = (float*)(tr + 2560);                         // (landed by the DMA of the tile's top)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // pass 1's reads of the image are done before it is rewritten
      // QuickGELU, u * sigmoid(1.702 u) * is[n], with two multiplies per element less: a = k u comes straight from the row scale
      // (k = -1.702 log2 e, folded into sa), and  u is / (1 + 2^a)  =  a / (c + c 2^a)  with c[n] = k / is[n] -- one fma in front of
      // the reciprocal instead of an add in front and two multiplies behind it.  c replaces is in the wave's image, one value per lane.
      constexpr float QK = -2.4554669595930156f;
      if constexpr (ACT == CE_ACT_QUICK_GELU) {
        const float is1 = isc[lane_e];
        isc[lane_e] = is1 != 0.f ? QK * __builtin_amdgcn_rcpf(is1) : 1e30f;      // (a zero output scale: the column becomes zero)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) sa[mt] *= QK;
      }
      // the lane's 32 column constants, once per tile (the same for its four 32-row blocks; the fragment registers are dead here)
      f32x4_t isr[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) isr[c] = *(const f32x4_t*)(isc + (c >> 2) * 32 + (c & 3) * 8 + he * 4);   // (the LDS serves one wave's accesses in order: the values written above)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int nt = c >> 2, g = c & 3;
          const int col = nt * 32 + g * 8 + he * 4;
          const f32x4_t is = isr[c];
          float v[4];
          if constexpr (ACT == CE_ACT_QUICK_GELU) {
            // the four elements stage by stage (4 x exp2, 4 x fma, 4 x rcp, 4 x mul): written element by element hipcc issues
            // the four dependent chains one after the other, each transcendental followed by the wait state its consumer needs
            // (s_nop 0) -- 38 issue cycles per element instead of 30, in an epilogue that both waves of a SIMD spend in VALU issue
            float a[4], ex[4], den[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = acc[mt][nt][g * 4 + e] * sa[mt];
#pragma unroll
            for (int e = 0; e < 4; ++e) ex[e] = __builtin_amdgcn_exp2f(a[e]);
            asm volatile("" : "+v"(ex[0]), "+v"(ex[1]), "+v"(ex[2]), "+v"(ex[3]));
#pragma unroll
            for (int e = 0; e < 4; ++e) den[e] = __builtin_fmaf(ex[e], is[e], is[e]);
#pragma unroll
            for (int e = 0; e < 4; ++e) den[e] = __builtin_amdgcn_rcpf(den[e]);
            asm volatile("" : "+v"(den[0]), "+v"(den[1]), "+v"(den[2]), "+v"(den[3]));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = F8_SAT(a[e] * den[e]);
          } else if constexpr (ACT == CE_ACT_GELU_ERF) {
            f32x4_t u4;
#pragma unroll
            for (int e = 0; e < 4; ++e) u4[e] = acc[mt][nt][g * 4 + e] * sa[mt];
            u4 = ce_gelu_erf4(u4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = F8_SAT(u4[e] * is[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = F8_SAT(act_apply_t<ACT>(acc[mt][nt][g * 4 + e] * sa[mt]) * is[e]);
          }
          int wd = 0;
          wd = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], wd, false);
          wd = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], wd, true);
          *(int*)(tr + r32e * 80 + col) = wd;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int row = k * 16 + (lane_e >> 2);
          const uint4 v = *(const uint4*)(tr + row * 80 + (lane_e & 3) * 16);
          const int m = mw0 + mt * 32 + row;
#if F8_NT_STORES
          if (m < p.M) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), (u32x4_t*)((char*)p.out + (size_t)m * p.ldo + nb + (lane_e & 3) * 16));
#else
          if (m < p.M) *(uint4*)((char*)p.out + (size_t)m * p.ldo + nb + (lane_e & 3) * 16) = v;
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if constexpr (RES) {
        // residual rows of this 32-row block: row-major image -> fragment layout
#pragma unroll
        for (int k = 0; k < 4; ++k) *(uint4*)(tr + k * 1024 + tr_base) = rres[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // cross-lane_e hand-off through the image
        if (mt + 1 < 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) LOAD_RES(mt * 4 + 4 + k);
        }
      }
      uint2 pk[8];
      [[maybe_unused]] float rs = 0.f, rss = 0.f;               // EPI 3: sum and sum of squares of the ROUNDED values, as stored
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int nt = c >> 2, g = c & 3;
        f32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = EPI == 3 ? acc[mt][nt][g * 4 + e] : acc[mt][nt][g * 4 + e] * sa[mt];   // (EPI 3: no row scale)
        if constexpr (RES) {
          const uint2 rr = *(const uint2*)TW_ADDR(c);
          v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
          v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = act_apply_t<ACT>(v[e]);
        }
        pk[c] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        if constexpr (EPI == 3 && F8_EPI3_PARTS >= 1) {
          amax[mt] = __builtin_fmaxf(amax[mt], __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])));
          amax[mt] = __builtin_fmaxf(amax[mt], __builtin_fmaxf(__builtin_fabsf(v[2]), __builtin_fabsf(v[3])));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[mt][nt][g * 4 + e] = v[e];      // kept for the quantising pass below
          const bf16x2_t ones2 = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
          const bf16x2_t p0 = __builtin_bit_cast(bf16x2_t, pk[c].x), p1 = __builtin_bit_cast(bf16x2_t, pk[c].y);
          rs = __builtin_amdgcn_fdot2_f32_bf16(p0, ones2, rs, false);
          rs = __builtin_amdgcn_fdot2_f32_bf16(p1, ones2, rs, false);
          rss = __builtin_amdgcn_fdot2_f32_bf16(p0, p0, rss, false);
          rss = __builtin_amdgcn_fdot2_f32_bf16(p1, p1, rss, false);
        }
      }
      if constexpr (EPI == 3) {
        // the other 32 of the wave's 64 columns sit in lane_e ^ 32
        rs4[mt] = rs + __shfl_xor(rs, 32);
        rss4[mt] = rss + __shfl_xor(rss, 32);
        amax[mt] = __builtin_fmaxf(amax[mt], __shfl_xor(amax[mt], 32));
      }
      if constexpr (RES) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads done before the image is rewritten
#pragma unroll
      for (int c = 0; c < 8; ++c) *(uint2*)TW_ADDR(c) = pk[c];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint4 v = *(const uint4*)(tr + k * 1024 + tr_base);
        const int m = mw0 + mt * 32 + k * 8 + row_l;
#if F8_NT_STORES
        if (m < p.M) { if constexpr (EPI == 0) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), (u32x4_t*)((bf16_t*)p.out + (size_t)m * p.ldo + gcol)); else *(uint4*)((bf16_t*)p.out + (size_t)m * p.ldo + gcol) = v; }
#else
        if (m < p.M) *(uint4*)((bf16_t*)p.out + (size_t)m * p.ldo + gcol) = v;
#endif
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // image reads done before the next block's writes
    }
    }
#undef LOAD_RES

    if constexpr (EPI == 3 && F8_EPI3_PARTS >= 2) {
      // ---- the e4m3 copy of the tile's new rows.  One exponent per (row, this tile's 256 columns): the four waves of a wave
      // row exchange their row maxima through their images ([3584, 4096): untouched by the tile-top DMA of this variant), then
      // x * 2^-e with |x * 2^-e| < 256 goes through the image as e4m3, 64 B per row and wave.
      // (the row statistics travel with the maxima: [2560, 3584) of the image holds (sum, sum of squares) of the wave's 64 columns
      //  per row; wave column 0 adds the four in a fixed order and stores ONE pair per row and 256 columns -- a quarter of the
      //  bytes the consumers' row-constant pass reads, and no partial store per 64 columns)
      if (he == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          *(float*)(tr + 3584 + (mt * 32 + r32e) * 4) = amax[mt];
          *(float2*)(tr + 2560 + (mt * 32 + r32e) * 8) = float2{rs4[mt], rss4[mt]};
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (not __syncthreads: its vmcnt(0) would drain the tile's stores)
      if (wc == 0 && he == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          float s_ = 0.f, ss_ = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float2 t = *(const float2*)(smem + TR_OFF + (wr * 4 + q) * 4096 + 2560 + (mt * 32 + r32e) * 8);
            s_ += t.x; ss_ += t.y;
          }
          const int m = mw0 + mt * 32 + r32e;
          if (m < p.M) *(float2*)(p.stats_out + ((size_t)(cur.n0 >> 8) * p.stats_ld + m) * 2) = float2{s_, ss_};
        }
      }
      float mul[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) a = __builtin_fmaxf(a, *(const float*)(smem + TR_OFF + (wr * 4 + q) * 4096 + 3584 + (mt * 32 + r32e) * 4));
        // |x| < 2^(ex - 126) with ex the biased exponent of the row maximum: e = ex - 134 puts it below 2^8 (e4m3 reaches 448)
        const int ex = (int)((__float_as_uint(a) >> 23) & 0xffu);
        const int eb = max(ex - 7, 0);                            // the E8M0 byte, e + 127
#if F8_EPI3_SCALECVT
        mul[mt] = __uint_as_float((unsigned)eb << 23);            // 2^e: the conversion's scale operand
#else
        mul[mt] = __uint_as_float((unsigned)(254 - eb) << 23);    // 2^-e
#endif
        const int m = mw0 + mt * 32 + r32e;
        if (wc == 0 && he == 0 && m < p.M) p.out_exp[(size_t)m * p.ld_oexp + (cur.n0 >> 8)] = (unsigned char)eb;
      }
      if constexpr (F8_EPI3_PARTS >= 3) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int nt = c >> 2, g = c & 3;
          const int col = nt * 32 + g * 8 + he * 4;
#if F8_EPI3_SCALECVT
          // the conversion divides by the power of two of its scale operand itself (tools/probes/cvt_scalef32_probe.hip)
          s16x2_t wq = {0, 0};
          wq = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wq, acc[mt][nt][g * 4 + 0], acc[mt][nt][g * 4 + 1], mul[mt], false);
          wq = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wq, acc[mt][nt][g * 4 + 2], acc[mt][nt][g * 4 + 3], mul[mt], true);
          const int wd = __builtin_bit_cast(int, wq);
#else
          int wd = 0;
          wd = __builtin_amdgcn_cvt_pk_fp8_f32(acc[mt][nt][g * 4 + 0] * mul[mt], acc[mt][nt][g * 4 + 1] * mul[mt], wd, false);
          wd = __builtin_amdgcn_cvt_pk_fp8_f32(acc[mt][nt][g * 4 + 2] * mul[mt], acc[mt][nt][g * 4 + 3] * mul[mt], wd, true);
#endif
          *(int*)(tr + r32e * 80 + col) = wd;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int row = k * 16 + (lane_e >> 2);
          const uint4 v = *(const uint4*)(tr + row * 80 + (lane_e & 3) * 16);
          const int m = mw0 + mt * 32 + row;
          if (m < p.M) *(uint4*)((char*)p.out8 + (size_t)m * p.ld8 + nb + (lane_e & 3) * 16) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      }
    }
    if constexpr (LNF) {
      // the next tile's exponent dwords take their place for its main loop (the image above is done with)
      if (he == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *(unsigned*)(tr + E_CUR + (mt * 32 + r32e) * 4) = e_hop[mt];
      }
    }
    }   // EPI != 4

#ifdef CLIPENC_DIAG
    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 3] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 0] = blockIdx.x; }
#endif
    if (!has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the redundant DMA before the LDS is released
      break;
    }
    relax = (EPI != 4 && cur.m0 + BM <= p.M) ? 2 : 0;    // (EPI 4 issues a data-dependent number of stores: never relaxed)  all 256 rows valid: every guarded row store above was issued; the first two waits
#ifdef CLIPENC_DIAG
    if (p.dbg) relax = 0;                    // (the stamps add stores: no relaxed waits then)
#endif
    idx = nidx; cur = nxt; Ablk = Anext; Wblk = Wnext;
  }
}

template <int EPI, int ACT, bool LNF = false>
hipError_t launch_fp8(const GemmParams& p, hipStream_t stream) {
  static DeviceKernelSetup setup;             // per device: LDS opt-in + CU count (common.h)
  int n_cu = 0;
  if (hipError_t e = setup.ensure((const void*)gemm_fp8_kernel<EPI, ACT, LNF>, LDS_BYTES, &n_cu); e != hipSuccess) return e;
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  int grid = n_cu > 0 ? n_cu : 256;
  grid -= grid % 8;
  if (grid < 8) grid = 8;
  if (tiles < grid) grid = tiles;
  hipLaunchKernelGGL((gemm_fp8_kernel<EPI, ACT, LNF>), dim3(grid), dim3(512), LDS_BYTES, stream, p);
  return hipGetLastError();
}

}  // namespace

#ifdef GEMM_FP8_TRI_TU
// gemm_fp8_tri.hip: the screen instantiation in a translation unit of its own (it cannot perturb the code generation of the
// tower's instantiations).  A == W: e4m3 [n_pad][lda] bytes (n_pad a multiple of 256, K = lda a multiple of 256, zero padded);
// scale_a = per-row margins [n_pad]; pairs = candidate slots (8 bytes each), cap, count; tile_list as ce_gemm_tri_persist uses.
hipError_t ce_gemm_fp8_tri(const GemmParams& p_in, hipStream_t stream) {
  GemmParams p = p_in;
  if (p.N < BN || p.N % BN != 0 || p.M != p.N || p.K < 512 || p.K % 256 != 0 || p.lda != p.ldw || p.lda < p.K || p.A != p.W) return hipErrorInvalidValue;
  if (!p.scale_a || !p.pairs || !p.count || ((uintptr_t)p.A & 15) || ((uintptr_t)p.scale_a & 15) || ((uintptr_t)p.pairs & 7)) return hipErrorInvalidValue;
  if ((size_t)255 * p.lda + 64 >= 0x7fffffffull) return hipErrorInvalidValue;
  const int tt = p.N / BN;
  if (tt > 0xffff) return hipErrorInvalidValue;
  static DeviceKernelSetup setup;
  int n_cu = 0;
  if (hipError_t e = setup.ensure((const void*)gemm_fp8_kernel<4, -1, false>, LDS_BYTES, &n_cu); e != hipSuccess) return e;
  const int tiles = tt * (tt + 1) / 2;
  int grid = n_cu > 0 ? n_cu : 256;
  grid -= grid % 8;
  if (grid < 8) grid = 8;
  if (tiles < grid) grid = tiles;
  if (hipError_t e = ce_tri_tile_list(tt, grid, &p.tile_list); e != hipSuccess) return e;
  hipLaunchKernelGGL((gemm_fp8_kernel<4, -1, false>), dim3(grid), dim3(512), LDS_BYTES, stream, p);
  return hipGetLastError();
}
#else
// A: fp8 [M][lda] (bytes), W: fp8 [N][ldw]; out bf16 [M][ldo]; scale_a [M], scale_w [N], bias [N] fp32.
// epi: EPI_STORE_BF16 (scale + bias + p.act), EPI_RESID (scale + bias + bf16 residual, may alias out) or
// EPI_STORE_FP8 (as STORE_BF16, then e4m3(value * out_inv_scale[n]) into out [M][ldo] BYTES).  scale_a NULL = 1.
hipError_t ce_gemm_fp8(const GemmParams& p, int epi, hipStream_t stream) {
  if (p.M < 1 || p.N < BN || p.N % BN != 0 || p.K < 256 || p.K % 256 != 0) return hipErrorInvalidValue;
  if (p.lda % 16 != 0 || p.ldw % 16 != 0 || p.lda < p.K || p.ldw < p.K) return hipErrorInvalidValue;
  if (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out) & 15) return hipErrorInvalidValue;
  if (!p.out || !p.scale_w || !p.bias) return hipErrorInvalidValue;
  if ((size_t)255 * p.lda + 64 >= 0x7fffffffull || (size_t)255 * p.ldw + 64 >= 0x7fffffffull) return hipErrorInvalidValue;
  if (epi == EPI_RESID) {
    if (!p.resid) return hipErrorInvalidValue;
    return launch_fp8<1, -1>(p, stream);
  }
  if (epi == EPI_RESID_Q) {
    // one exponent byte per (row, 256-column tile), four of them in a dword: N <= 1024
    if (!p.resid || p.scale_a || !p.w_exp || !p.out8 || !p.out_exp || !p.stats_out || p.N > 1024 || p.ld8 < p.N || p.ld8 % 16 != 0 || p.ld_oexp < 4 ||
        p.stats_ld < p.M || ((uintptr_t)p.out8 & 15) || ((uintptr_t)p.stats_out & 7))
      return hipErrorInvalidValue;
    return launch_fp8<3, -1>(p, stream);
  }
  if (p.a_exp) {
    // block-exponent A rows with the folded LayerNorm; per-token scales do not combine with it
    if (p.scale_a || !p.w_exp || !p.row_r || !p.row_d || !p.colsum || p.K > 1024 || p.ld_aexp < 4 || p.ld_aexp % 4 != 0 || p.ld_row < 1 ||
        ((uintptr_t)p.a_exp & 3) || (size_t)(p.M - 1) * p.ld_aexp >= 0x7fffffffull || (size_t)(p.M - 1) * p.ld_row * 4 >= 0x7fffffffull)
      return hipErrorInvalidValue;
    if (epi == EPI_STORE_FP8) {
      if (!p.out_inv_scale) return hipErrorInvalidValue;
      if (p.act == CE_ACT_QUICK_GELU) return launch_fp8<2, CE_ACT_QUICK_GELU, true>(p, stream);
      if (p.act == CE_ACT_GELU_ERF) return launch_fp8<2, CE_ACT_GELU_ERF, true>(p, stream);
      return launch_fp8<2, -1, true>(p, stream);
    }
    if (epi != EPI_STORE_BF16) return hipErrorInvalidValue;
    if (p.act == CE_ACT_QUICK_GELU) return launch_fp8<0, CE_ACT_QUICK_GELU, true>(p, stream);
    if (p.act == CE_ACT_GELU_ERF) return launch_fp8<0, CE_ACT_GELU_ERF, true>(p, stream);
    return launch_fp8<0, -1, true>(p, stream);
  }
  if (epi == EPI_STORE_FP8) {
    if (!p.out_inv_scale) return hipErrorInvalidValue;
    if (p.act == CE_ACT_QUICK_GELU) return launch_fp8<2, CE_ACT_QUICK_GELU>(p, stream);
    if (p.act == CE_ACT_GELU_ERF) return launch_fp8<2, CE_ACT_GELU_ERF>(p, stream);
    return launch_fp8<2, -1>(p, stream);
  }
  if (epi != EPI_STORE_BF16) return hipErrorInvalidValue;
  if (p.act == CE_ACT_QUICK_GELU) return launch_fp8<0, CE_ACT_QUICK_GELU>(p, stream);
  if (p.act == CE_ACT_GELU_ERF) return launch_fp8<0, CE_ACT_GELU_ERF>(p, stream);
  return launch_fp8<0, -1>(p, stream);
}
#endif
