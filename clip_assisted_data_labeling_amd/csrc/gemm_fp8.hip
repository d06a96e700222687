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
#include "common.h"
#include "gemm.h"

#ifndef F8_MMA_ORDER
#define F8_MMA_ORDER 1
#endif
// instantiations with the register room for the peeled zero-C first stage pair (the others spill with it)
#define ZERO_C_SET(EPI, ACT) ((EPI) == 2 && (ACT) <= 0)

namespace {

constexpr int BM = 256, BN = 256;
constexpr int BUF = 65536, WREG = 32768;    // one K=128 stage: A region | W region
constexpr int RING = 2 * BUF;               // 131072
constexpr int TR_OFF = RING;                // 8 x 4 KiB wave-private images
constexpr int LDS_BYTES = RING + 32768;     // 163840

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

#define LDS_PTR(off) ((__attribute__((address_space(3))) void*)(smem + (off)))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int ACT>
__device__ __forceinline__ float act_apply_t(float u) {
  if constexpr (ACT == CE_ACT_QUICK_GELU) return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * u));
  else if constexpr (ACT == CE_ACT_GELU_ERF) return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
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

// EPI: 0 = scale + bias (+ activation) -> bf16;  1 = scale + bias + residual (bf16, in place) -> bf16;
//      2 = scale + bias (+ activation), then * out_inv_scale[n] -> e4m3 (the next GEMM's operand, no bf16 round trip)
template <int EPI, int ACT>
__global__ __launch_bounds__(512, 2) void gemm_fp8_kernel(const GemmParams p) {
  constexpr bool ZERO_C_OK = ZERO_C_SET(EPI, ACT);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int r32 = lane & 31, h = lane >> 5;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  const int nwg = tiles_m * tiles_n;
  const int G = gridDim.x;
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

  // epilogue image: [32 rows][128 B] per wave, 16-B chunk index XOR row&7
  char* tr = smem + TR_OFF + w * 4096;
  const int tw_base = r32 * 128 + h * 8;
  const int tw_sw = r32 & 7;
  const int tr_base = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);   // + k*1024: rows 8k + (lane>>3)
  const int row_l = lane >> 3;
#define TW_ADDR(c) (tr + tw_base + ((((c)) ^ tw_sw) << 4))

  int idx = blockIdx.x;
  TileId cur = decode_tile(idx, tiles_m, tiles_n);
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
                                                                              0, 0x7f7f7f7f, 0, 0x7f7f7f7f);        \
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
#define STAGE(b, VMWAIT, PA_ISSUE, PB_ISSUE) STAGE_Z(b, 0, VMWAIT, PA_ISSUE, PB_ISSUE)
#define STAGE_Z(b, ZC, VMWAIT, PA_ISSUE, PB_ISSUE)                                          \
  do {                                                                                      \
    LD_W2(b) __builtin_amdgcn_sched_barrier(0); LD_A2(b, 0)                                 \
    PA_ISSUE;                                                                               \
    VMWAIT;                                                                                 \
    SYNC_MMA(0, ZC);                                                                        \
    LD_A2(b, 1)                                                                             \
    PB_ISSUE;                                                                               \
    VMWAIT;                                                                                 \
    SYNC_MMA(1, ZC);                                                                        \
  } while (0)
#define VM8 asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
  // First two waits after an epilogue: what they need was issued before the epilogue's stores and vmcnt retires in order,
  // so they may leave the NST row stores of a complete tile outstanding as well (see gemm_persist.hip); one opaque
  // instruction for the compiler.
  constexpr int NST = EPI == 2 ? 8 : 16;     // row stores per wave and tile
#define VM_RELAX                                                                            \
  do {                                                                                      \
    const int sel_ = __builtin_amdgcn_readfirstlane(relax > 0 ? 1 : 0);                     \
    relax = relax > 0 ? relax - 1 : 0;                                                      \
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lf8vm8_%=\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Lf8end_%=\n"               \
                 ".Lf8vm8_%=:\n\ts_waitcnt vmcnt(8)\n.Lf8end_%=:" : : "s"(sel_), "n"(8 + NST + CB_PIECES) : "memory", "scc");    \
  } while (0)
  // the column constants (weight scales, biases, EPI 2: inverse output scales) of the tile are staged into the wave's image at
  // the top of the tile (below): those pieces sit between the previous tile's stores and the first waits (the per-row
  // scale pieces, issued only with per-token scales, are not counted: the wait is then two pieces stricter than need be)
  constexpr int CB_PIECES = EPI == 2 ? 3 : 2;

  // ---- cold prologue of the first tile ----
  ISSUE_WAH0(0, Ablk, Wblk, aoff00, aoff01, 0); ISSUE_AH1(0, Ablk, aoff10, aoff11, 0);
  ISSUE_WAH0(1, Ablk, Wblk, aoff00, aoff01, 128);           // A(half 1) of stage 1 follows in PA of stage 0
  VM8;                                                      // W and A(half 0) of stage 0 have landed
  int relax = 0;                             // waits of the coming tile that may leave the previous tile's stores in flight
  BARRIER();

  for (;;) {
    f32x16_t acc[4][2];
    const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    i32x8_t fa[4], fb[4];

    // Column constants of this tile's 64 columns per wave and the per-token scales of its 128 rows, by LDS-DMA into the wave's
    // (idle until the epilogue) 4-KiB image: [0, 256) weight scales, [1024, 1280) biases, [2560, 2816) inverse output scales
    // (EPI 2), [3584, 4096) per-row scales.  As global loads at the head of the epilogue they made the first block wait for
    // every older DMA piece of the next tile (vmcnt retires in order) -- see gemm_persist.hip, same change, same argument:
    // extra pieces only make the counted waits stricter, and they have retired before the epilogue.
    {
      int lane_t;                                              // the lane id from the hardware (not kept across the main loop)
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
      const unsigned coff = (unsigned)(cur.n0 + wc * 64) * 4u + (unsigned)(lane_t & 15) * 16u;
      const unsigned ti = lds0 + (unsigned)(TR_OFF + w * 4096);
      glds16_at((const char*)p.scale_w, coff, ti);
      glds16_at((const char*)p.bias, coff, ti + 1024);
      if constexpr (EPI == 2) glds16_at((const char*)p.out_inv_scale, coff, ti + 2560);
      if (p.scale_a) {
        const int r0 = cur.m0 + wr * 128 + lane_t;
        glds4_at((const char*)p.scale_a, (unsigned)min(r0, p.M - 1) * 4u, ti + 3584);
        glds4_at((const char*)p.scale_a, (unsigned)min(r0 + 64, p.M - 1) * 4u, ti + 3840);
      }
    }
    if (wr == 1) BARRIER();                  // second wave row runs half a phase behind

    constexpr bool ZERO_C = ZERO_C_OK;
    if (ZERO_C && kend > 256) {
      // first stage pair of the tile: every accumulator starts from the constant 0 in its first MFMA
      STAGE_Z(0, 1, VM_RELAX, ISSUE_AH1(1, Ablk, aoff10, aoff11, 128), ISSUE_WAH0(0, Ablk, Wblk, aoff00, aoff01, 256));
      STAGE(1, VM8, ISSUE_AH1(0, Ablk, aoff10, aoff11, 256), ISSUE_WAH0(1, Ablk, Wblk, aoff00, aoff01, 384));
      for (int kb = 256; kb < kend - 256; kb += 256) {
        STAGE(0, VM8, ISSUE_AH1(1, Ablk, aoff10, aoff11, kb + 128), ISSUE_WAH0(0, Ablk, Wblk, aoff00, aoff01, kb + 256));
        STAGE(1, VM8, ISSUE_AH1(0, Ablk, aoff10, aoff11, kb + 256), ISSUE_WAH0(1, Ablk, Wblk, aoff00, aoff01, kb + 384));
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16;
      if constexpr (!ZERO_C) {
        for (int kb = 0; kb < kend - 256; kb += 256) {
          STAGE(0, VM_RELAX, ISSUE_AH1(1, Ablk, aoff10, aoff11, kb + 128), ISSUE_WAH0(0, Ablk, Wblk, aoff00, aoff01, kb + 256));
          STAGE(1, VM8, ISSUE_AH1(0, Ablk, aoff10, aoff11, kb + 256), ISSUE_WAH0(1, Ablk, Wblk, aoff00, aoff01, kb + 384));
        }
      }
    }
    // ---- last two stages: the DMA crosses into the next tile (or re-fetches this one into dead buffers) ----
    const int nidx = idx + G;
    const bool has_next = nidx < nwg;
    TileId nxt = cur;
    const char *Anext = Ablk, *Wnext = Wblk;
    if (has_next) {
      nxt = decode_tile(nidx, tiles_m, tiles_n);
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
    {
      const int kb = kend - 256;
      STAGE(0, VM_RELAX, ISSUE_AH1(1, Ablk, aoff10, aoff11, kb + 128), ISSUE_WAH0(0, Anext, Wnext, aoff00, aoff01, 0));
      aoff10 = AOFF_B(arow_b + 64); aoff11 = AOFF_B(arow_b + 192);
      STAGE(1, VM8, ISSUE_AH1(0, Anext, aoff10, aoff11, 0), ISSUE_WAH0(1, Anext, Wnext, aoff00, aoff01, 128));
    }
#undef AOFF_B
    // pin the accumulators here: without a use in this block LLVM sinks the tail's 32 MFMAs below the conditional
    // barrier (all fragments live at once -> hundreds of spilled VGPRs)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][j]));
    if (wr == 0) BARRIER();                  // re-align the two wave rows for the epilogue

    // ------------------------------- epilogue of tile `cur` -------------------------------
    // lane (r32, h) of m-tile mt holds row mw0 + mt*32 + r32, columns nb + nt*32 + 8g + 4h + (0..3) in acc[mt][nt][4g..4g+3]
    const int mw0 = cur.m0 + wr * 128;
    const int nb = cur.n0 + wc * 64;
    const size_t gcol = (size_t)nb + (lane & 7) * 8;
    // pass 1 (keeps registers low): acc <- acc * sw[n] + bias[n] / sa[m], so that pass 2 only multiplies by sa[m]
    float sa[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) sa[mt] = p.scale_a ? *(const float*)(tr + 3584 + (mt * 32 + r32) * 4) : 1.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int nl = (c >> 2) * 32 + (c & 3) * 8 + h * 4;          // column within the wave's 64
      const f32x4_t sw = *(const f32x4_t*)(tr + nl * 4);
      const f32x4_t bs = *(const f32x4_t*)(tr + 1024 + nl * 4);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const float rsa = __builtin_amdgcn_rcpf(sa[mt]);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[mt][c >> 2][(c & 3) * 4 + e] = fmaf(acc[mt][c >> 2][(c & 3) * 4 + e], sw[e], bs[e] * rsa);
      }
    }
    uint4 rres[4];                           // residual rows: 4 x 16 B per 32-row block, one block in flight
#define LOAD_RES(k)                                                                           \
  do {                                                                                        \
    const int m_ = mw0 + (k) * 8 + row_l;                                                     \
    rres[(k) & 3] = uint4{0, 0, 0, 0};                                                        \
    if (m_ < p.M) rres[(k) & 3] = *(const uint4*)((const bf16_t*)p.resid + (size_t)m_ * p.ldo + gcol); \
  } while (0)
    if constexpr (EPI == 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) LOAD_RES(k);
    }

    if constexpr (EPI == 2) {
      // fp8 image: [32 rows][80 B pitch] per wave (64 data bytes; the pitch keeps the dword writes 2-way conflicted at most),
      // the wave's 64 inverse output scales behind it
      const float* isc = (const float*)(tr + 2560);             // (landed by the DMA of the tile's top)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // pass 1's reads of the image are done before it is rewritten
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int nt = c >> 2, g = c & 3;
          const int col = nt * 32 + g * 8 + h * 4;
          const f32x4_t is = *(const f32x4_t*)(isc + col);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = __builtin_amdgcn_fmed3f(act_apply_t<ACT>(acc[mt][nt][g * 4 + e] * sa[mt]) * is[e], -448.0f, 448.0f);
          int wd = 0;
          wd = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], wd, false);
          wd = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], wd, true);
          *(int*)(tr + r32 * 80 + col) = wd;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int row = k * 16 + (lane >> 2);
          const uint4 v = *(const uint4*)(tr + row * 80 + (lane & 3) * 16);
          const int m = mw0 + mt * 32 + row;
          if (m < p.M) *(uint4*)((char*)p.out + (size_t)m * p.ldo + nb + (lane & 3) * 16) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if constexpr (EPI == 1) {
        // residual rows of this 32-row block: row-major image -> fragment layout
#pragma unroll
        for (int k = 0; k < 4; ++k) *(uint4*)(tr + k * 1024 + tr_base) = rres[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // cross-lane hand-off through the image
        if (mt + 1 < 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) LOAD_RES(mt * 4 + 4 + k);
        }
      }
      uint2 pk[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int nt = c >> 2, g = c & 3;
        f32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mt][nt][g * 4 + e] * sa[mt];
        if constexpr (EPI == 1) {
          const uint2 rr = *(const uint2*)TW_ADDR(c);
          v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
          v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = act_apply_t<ACT>(v[e]);
        }
        pk[c] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
      if constexpr (EPI == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragment reads done before the image is rewritten
#pragma unroll
      for (int c = 0; c < 8; ++c) *(uint2*)TW_ADDR(c) = pk[c];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint4 v = *(const uint4*)(tr + k * 1024 + tr_base);
        const int m = mw0 + mt * 32 + k * 8 + row_l;
        if (m < p.M) *(uint4*)((bf16_t*)p.out + (size_t)m * p.ldo + gcol) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // image reads done before the next block's writes
    }
    }
#undef LOAD_RES

    if (!has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the redundant DMA before the LDS is released
      break;
    }
    relax = (cur.m0 + BM <= p.M) ? 2 : 0;    // all 256 rows valid: every guarded row store above was issued; the first two waits
    idx = nidx; cur = nxt; Ablk = Anext; Wblk = Wnext;
  }
}

template <int EPI, int ACT>
hipError_t launch_fp8(const GemmParams& p, hipStream_t stream) {
  static DeviceKernelSetup setup;             // per device: LDS opt-in + CU count (common.h)
  int n_cu = 0;
  if (hipError_t e = setup.ensure((const void*)gemm_fp8_kernel<EPI, ACT>, LDS_BYTES, &n_cu); e != hipSuccess) return e;
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  int grid = n_cu > 0 ? n_cu : 256;
  grid -= grid % 8;
  if (grid < 8) grid = 8;
  if (tiles < grid) grid = tiles;
  hipLaunchKernelGGL((gemm_fp8_kernel<EPI, ACT>), dim3(grid), dim3(512), LDS_BYTES, stream, p);
  return hipGetLastError();
}

}  // namespace

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
