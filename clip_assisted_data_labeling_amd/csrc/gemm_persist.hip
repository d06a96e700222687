// Persistent bf16 "NT" GEMM (tile, MFMA mapping and epilogues as in gemm_bf16.hip; this file changes WHO runs
// the tiles and HOW the operands are staged): one workgroup per CU walks a strided list of 256x256 tiles and keeps a
// double-buffered K=64 stage pipeline running ACROSS tile boundaries (the last two stages of a tile already fetch
// stages 0 and 1 of the next tile), so a tile costs main loop + epilogue only.
//
// Staging: every LDS-DMA instruction fetches 8 rows x 128 B = 8 WHOLE cache lines (8 consecutive lanes per line).
// The earlier K=32 ring fetched 16 rows x 64 B per instruction, i.e. half lines: same bytes, but twice the L1->L2 read
// requests (rocprofv3 TCP_TCC_READ_REQ: 100.8 M vs 50.3 M per launch for the vendor kernel on the same shapes), and at
// ~87 requests per clock the L2's request throughput, not the MFMA issue stream, set the pace of every variant of the
// main loop that was tried.
//
// LDS (160 KiB): 2 buffers x (A 256 rows x 128 B | W 256 rows x 128 B) = 128 KiB | [128K,144K) AUX: EPI_LNFOLD raw row
// statistics, 2 x [4 parts][256 rows][sum,sumsq] landed by LDS-DMA one tile ahead; EPI_RESID per-wave row partial sums
// | [144K,160K) 8 wave-private 2 KiB images for the epilogue's layout change.
// Buffer image: 1-KiB blocks of 8 rows x 128 B; 16-B chunk c of row r is stored at position c ^ (r & 6) of its row
// (applied to the per-lane SOURCE address of the DMA and to the read address), which makes every ds_read_b128 of a
// 16-row x 4-chunk fragment conflict-free (checked exhaustively against the lane-group table).
//
// One K=64 stage = 2 phases, each {fragment reads, DMA issue, counted wait} lgkmcnt(0) s_barrier {32 MFMA} s_barrier, the
// two wave rows (the two waves of every SIMD) half a phase apart (one extra barrier) so that one issues MFMAs while the
// other reads LDS:
//   PA: W of both k halves (8 fragments, kept in registers for the whole stage) + A(row half 0) of both k halves
//   PB: A(row half 1) of both k halves
// (the first version ran four phases of 16 MFMAs: twice the barrier hand-overs, each an idle matrix pipe for a barrier
// round trip).  W and A(half 0) of a buffer are last read in PA, A(half 1) in PB, so stage s+2's W + A(half 0) are
// fetched into the buffer of stage s in PB of stage s (6 pieces per wave) and A(half 1) of stage s+1 in PA of stage s
// (2 pieces).  Each phase ends its read section with ONE counted wait, vmcnt(8), behind its own pieces: the 8 newest
// stay in flight and what retires is what the NEXT phase reads, so every piece has two phases to land.
// Ordering: a phase's reads are retired by lgkmcnt(0) BEFORE its first barrier, so rows are re-staged one phase after
// their last read: when a wave issues DMA in phase p it has passed the second barrier of p-1, which the other wave row
// only reaches after the first barrier of its own p-1, i.e. after its reads of p-1 have returned.  A stage is read one
// phase after the wait that retired it (the later wave row waits one barrier later and reads one barrier later).
// Across a tile boundary stage 0 and W/A(half 0) of stage 1 of the next tile are in the pipeline BEFORE the epilogue's
// stores are issued, so the first two counted waits of the new tile may leave those 16-17 stores outstanding (vmcnt
// retires in order).
#include <stdlib.h>

#include "common.h"
#include "gemm.h"

namespace {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr int BM = 256, BN = 256;
constexpr int BUF = 65536, WREG = 32768;    // one K=64 stage: A region | W region
constexpr int RING = 2 * BUF;               // 131072
constexpr int AUX_OFF = RING;               // 16 KiB
constexpr int TR_OFF = RING + 16384;        // 8 x 2 KiB
constexpr int LDS_BYTES = RING + 32768;     // 163840

#define LDS_PTR(off) ((__attribute__((address_space(3))) void*)(smem + (off)))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int ACT>
__device__ __forceinline__ float act_apply_t(float u) {
  // compile-time activation: a run-time `act` makes hipcc evaluate BOTH activations per element and select
  if constexpr (ACT == CE_ACT_QUICK_GELU) return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * u));
  else if constexpr (ACT == CE_ACT_GELU_ERF) return ce_gelu_erf(u);
  else return u;
}


// One LDS-DMA piece: 64 lanes x 16 B from (uniform base in an SGPR pair + per-lane 32-bit offset) to 1 KiB of LDS at
// lds_off.  Inline asm: the builtin makes a 64-bit VGPR address per piece (two VGPRs per offset plus temporaries, which
// this kernel does not have), and every wait on these pieces is hand-placed anyway.
__device__ __forceinline__ void glds16_at(const char* base, unsigned off, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_addr) : "memory");
}

struct TileId { int m0, n0, tn; };

// operand type per epilogue: the near-duplicate search (EPI_THRESH, gemm_tri.hip) runs on f16 operands, everything else on bf16
template <int EPI> struct OperandOf { typedef bf16x8_t frag; };
template <> struct OperandOf<EPI_THRESH> { typedef f16x8_t frag; };
__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4_t mfma16(f16x8_t a, f16x8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// Upper-triangular tile list of a TT x TT tile grid (EPI_THRESH): the ORDER is made on the host (tri_tile_order below) and read
// here with one scalar load per tile -- entry i = tm | tn << 16 is the tile that workgroup i % G runs in its round i / G.
// (Scalar memory counts in lgkmcnt, not in the vmcnt that the DMA pipeline's counted waits rely on.)
__device__ __forceinline__ TileId tri_tile(const unsigned* list, int i) {
  unsigned v;
  asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(list), "s"(i * 4) : "memory");
  const int tm = (int)(v & 0xffffu), tn = (int)(v >> 16);
  return TileId{tm * BM, tn * BN, tn};
}

// GEMM_WALK_SG > 0 (experiment, round 4): the XCD sweeps GEMM_WALK_SG tiles along M for a block of GEMM_WALK_NB tiles along N,
// then the next N block of the same super-group: the NB W panels of a block (NB x 512 KiB at K = 1024) stay in the XCD's L2 for
// SG * NB / 32 rounds while the A panels stream through once per N block (re-read from the Infinity Cache tiles_n / NB times).
#ifndef GEMM_WALK_SG
#define GEMM_WALK_SG 0
#endif
#ifndef GEMM_WALK_NB
#define GEMM_WALK_NB 4
#endif
__device__ __forceinline__ TileId decode_tile(int idx, int tiles_m, int tiles_n, bool deep_narrow) {
  const int nwg = tiles_m * tiles_n;
  const int q = nwg >> 3, r = nwg & 7, xcd = idx & 7, pos = idx >> 3;
  const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;   // XCD-aware, bijective
#if GEMM_WALK_SG > 0
  if (!deep_narrow && tiles_n > GEMM_WALK_NB && tiles_n % GEMM_WALK_NB == 0) {
    const int per_sg = GEMM_WALK_SG * tiles_n;
    const int sg = bid / per_sg, rr = bid - sg * per_sg;
    const int first = sg * GEMM_WALK_SG;
    const int gsz_ = min(tiles_m - first, GEMM_WALK_SG);
    const int per_blk = gsz_ * GEMM_WALK_NB;
    const int nb = rr / per_blk, rem = rr - nb * per_blk;
    const int tm_ = first + rem / GEMM_WALK_NB, tn_ = nb * GEMM_WALK_NB + rem % GEMM_WALK_NB;
    return TileId{tm_ * BM, tn_ * BN, tn_};
  }
#endif
  const int GM = deep_narrow ? 2 : 8;         // tiles along M per group; FC2's shape (4 N-tiles, K = 4096) measured 2.3 % faster with 2
  const int group = bid / (GM * tiles_n);
  const int first_m = group * GM;
  const int gsz = min(tiles_m - first_m, GM);
  const int tm = first_m + (bid % (GM * tiles_n)) % gsz;
  const int tn = (bid % (GM * tiles_n)) / gsz;
  return TileId{tm * BM, tn * BN, tn};
}

template <int EPI, int ACT>
__global__ __launch_bounds__(512, 2) void gemm_persist_kernel(const GemmParams p) {
  typedef typename OperandOf<EPI>::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (EPI == EPI_THRESH) {           // the exact search as a fall-back (gemm.h): nothing to do unless the counter ran over
    if (p.run_if_over && *(const volatile unsigned long long*)p.run_if_over <= p.run_if_limit) return;
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int frow = lane & 15;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  const int nwg = EPI == EPI_THRESH ? tiles_n * (tiles_n + 1) / 2 : tiles_m * tiles_n;
  const bool deep_narrow = tiles_n <= 4 && p.K >= 2048;
#define DECODE_TILE(i_) (EPI == EPI_THRESH ? tri_tile(p.tile_list, (i_)) : decode_tile((i_), tiles_m, tiles_n, deep_narrow))
  const int G = gridDim.x;
  // Dynamic tail (p.ticket): workgroup b walks b, b + G, ... by stride through all rounds but the last full one; the tiles from
  // S on -- between one and two rounds -- are taken from a global ticket counter in the order the workgroups arrive.  The
  // workgroups of a launch drift apart (tools/gemm_stamps.py: finish times spread over more than a tile time; mean idle at the end
  // of a launch 2-3.4 %, of which the static partial round is 0.4-2.7 %), and with tickets the ones that run ahead take the extra
  // tiles.  The ticket is a SCALAR atomic (s_atomic_add, counted in lgkmcnt): a vector atomic would join the in-order vmcnt
  // queue that the DMA pipeline's hand-counted waits are written against.  One wave takes it at the top of a tile, a tile
  // before it is needed, and parks it in a free word of its epilogue image; which workgroup runs a tile changes no result bit.
  // Two forms (measured, tools/experiments/README.md): outputs of 4 tiles along N (out-proj, FC2) -- ONE counter, one full round
  // by ticket; wider outputs (QKV, FC1: 12 / 16 tiles along N, whose XCD-aware order is worth 0.5 % on two ticketed rounds) -- one
  // counter PER XCD (p.ticket[0..7]), two full rounds: the workgroups of XCD x take the positions of ITS strided sequence (tile
  // index mod 8 = x) in the order they get there, so the tiles resident on an XCD stay neighbours; an XCD that has run out takes
  // from the next one's counter (at most seven more atomics, at the very end of a launch).
#ifndef GEMM_DYN_ROUNDS
#define GEMM_DYN_ROUNDS 1                    // 0: every tile by stride (rounds 1-3)
#endif
  const bool by_xcd = tiles_n > 4 && (G & 7) == 0;
  const int dyn_rounds = GEMM_DYN_ROUNDS == 0 ? 0 : (by_xcd ? 2 : 1);
  // (K = 128: the tile is ONE stage pair and no STAGE barrier lies between wave 4's write of the ticket slot and the first wave
  //  row's read of it -- such launches walk by stride)
  const int S = (dyn_rounds > 0 && EPI != EPI_THRESH && p.ticket != nullptr && nwg / G >= dyn_rounds + 2 && p.K > 128)
                    ? (nwg / G - dyn_rounds) * G : 0x7fffffff;
  // where the ticket waits in LDS: EPI_LNFOLD -- word 0 of part 1 of the CURRENT raw-statistics buffer (dead once this tile's
  // (mean, rstd) are converted, which the ticket-taking wave has done itself by then; the next tile's statistics land in the other
  // buffer); otherwise the unused fourth quarter of AUX.  (The waves' epilogue images are NOT free during the main loop: the
  // column-sum / bias pieces are whole 1-KiB DMA writes.)
#define TICKET_SLOT (EPI == EPI_LNFOLD ? AUX_OFF + (tile_iter & 1) * 8192 + 2048 : AUX_OFF + 12288)
  const size_t lda_b = (size_t)p.lda * 2, ldw_b = (size_t)p.ldw * 2;
  const int kend = p.K * 2;                  // bytes along K; one stage = 128 B; K % 128 == 0 (stages come in pairs)

  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(0));   // LDS address of smem[0]
  // LDS-DMA lane mapping: lane L fetches logical 16-B chunk (L&7) ^ ((L>>3)&6) of row L>>3 of an 8-row block
  const int dg = lane >> 3;
  const int dchunk16 = ((lane & 7) ^ (dg & 6)) * 16;
  // A blocks of wave w (per row half h): q = 2w+i, i = 0,1 -> tile rows (w>>2)*128 + h*64 + (2(w&3)+i)*8 + dg
  const int arow0 = (w >> 2) * 128 + (2 * (w & 3)) * 8 + dg;          // + h*64 + i*8
  const int a_dma = ((w >> 2) * 16 + 2 * (w & 3)) * 1024;             // + (h*8 + i)*1024 (+ buffer)
  // W blocks of wave w: rows 32w + 8i + dg, i = 0..3
  const unsigned woff = (unsigned)((32 * w + dg) * ldw_b) + dchunk16;   // + i*8*ldw_b through the scalar base
  const int w_dma = WREG + 4 * w * 1024;                               // + i*1024 (+ buffer)
  // fragment reads: row frow, k-chunk kq of a 16-row block; the second k half (k 32..63) is the address ^ 64
  const int rdl = (frow >> 3) * 1024 + (frow & 7) * 128 + ((((lane >> 4) ^ (frow & 6))) << 4);
  const int a_rd0 = wr * 16 * 1024 + rdl;
  const int w_rd0 = WREG + wc * 8 * 1024 + rdl;

#define TW_ADDR(nt) (tr + tw_base + ((((nt) * 2 + (qd >> 1)) ^ tw_sw) << 4))

  int idx = blockIdx.x;
  TileId cur = DECODE_TILE(idx);
  const char* Ablk = (const char*)p.A + (size_t)cur.m0 * lda_b;
  const char* Wblk = (const char*)p.W + (size_t)cur.n0 * ldw_b;
#define AOFF(m0v, r) ((unsigned)((min((m0v) + (r), p.M - 1) - (m0v)) * lda_b) + dchunk16)
  unsigned aoff00 = AOFF(cur.m0, arow0), aoff01 = AOFF(cur.m0, arow0 + 8);            // half 0, i = 0,1
  unsigned aoff10 = AOFF(cur.m0, arow0 + 64), aoff11 = AOFF(cur.m0, arow0 + 72);      // half 1

  // DMA pieces of one stage (buffer b, operand block pointers, byte offset along K)
#define ISSUE_AH0(b, blk, o0, o1, kbyte)                                                    \
  do {                                                                                      \
    glds16((blk) + (kbyte), (o0), smem, (b) * BUF + a_dma);                                 \
    glds16((blk) + (kbyte), (o1), smem, (b) * BUF + a_dma + 1024);                          \
  } while (0)
#define ISSUE_AH1(b, blk, o0, o1, kbyte)                                                    \
  do {                                                                                      \
    glds16((blk) + (kbyte), (o0), smem, (b) * BUF + a_dma + 8192);                          \
    glds16((blk) + (kbyte), (o1), smem, (b) * BUF + a_dma + 9216);                          \
  } while (0)
#define ISSUE_W(b, blk, kbyte)                                                              \
  do {                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                        \
      glds16((blk) + (kbyte) + (size_t)i_ * 8 * ldw_b, woff, smem, (b) * BUF + w_dma + i_ * 1024); \
  } while (0)
  // raw row statistics of a tile's 256 rows: parts x 2 KiB, fetched by waves 0 and 1 (EPI_LNFOLD)
#define glds16(base, off, smem_, lds_off) glds16_at((base), (off), lds0 + (unsigned)(lds_off))
#define STAGE_STATS(buf, m0v)                                                               \
  do {                                                                                      \
    if (EPI == EPI_LNFOLD && w < 2) {                                                        \
      for (int part = 0; part < p.stats_in_parts; ++part)                                    \
        glds16((const char*)p.stats_in + ((size_t)part * p.stats_ld + (m0v)) * 8, (unsigned)((w * 64 + lane) * 16), smem, \
               AUX_OFF + (buf) * 8192 + part * 2048 + w * 1024);                             \
    }                                                                                       \
  } while (0)
  // fragments of one phase: W for both k halves (fb[4 kh + j], kept for both phases of the stage) and one A row half for
  // both k halves (fa[4 kh + i]); the k 0..31 fragments are read first, they feed the first MFMAs
  // (the address of the second k half is re-derived where it is used -- one v_xor in an asm statement that hipcc cannot
  // hoist: kept across the whole kernel these two addresses were what the register-tight instantiations spilled)
#define XOR64(dst, src) asm volatile("v_xor_b32 %0, 64, %1" : "=v"(dst) : "v"(src))
#define LD_W2(b) { int w_rd1_; XOR64(w_rd1_, w_rd0);                                                                                \
                   _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[j] = *(const frag_t*)(smem + (b) * BUF + w_rd0 + j * 2048);      \
                   _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[4 + j] = *(const frag_t*)(smem + (b) * BUF + w_rd1_ + j * 2048); }
#define LD_A2(b, half) { int a_rd1_; XOR64(a_rd1_, a_rd0);                                                                          \
                         _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[i] = *(const frag_t*)(smem + (b) * BUF + a_rd0 + ((half) * 8 + i * 2) * 1024);  \
                         _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[4 + i] = *(const frag_t*)(smem + (b) * BUF + a_rd1_ + ((half) * 8 + i * 2) * 1024); }
  // ZC: the first MFMA of every accumulator of a tile takes the constant 0 as its C operand instead of a zeroed register
  // (128 v_mov per wave and tile with the matrix pipe idle otherwise) -- or, for EPI_RESID, the BIAS of its four columns:
  // the sum starts from the bias and the epilogue has no bias add
#define CINIT(j) (EPI == EPI_RESID ? binit[(j)] : f32x4_t{0.f, 0.f, 0.f, 0.f})
// Issue order of a phase's 32 MFMAs (16 accumulators x 2 k halves).  What the board pays for is switching: a pure stream of this
// block on all CUs (tools/probes/power_probe.py) runs 2.0 % faster when exactly ONE operand register set changes between
// consecutive MFMAs (serpentine: j runs 0..3, 3..0, ...; PROBE_ONLY=order) -- and 5.0 % faster still when the two k halves of an
// accumulator are issued BACK TO BACK (PROBE_ONLY=acc, round 4: 126.3 -> 132.6 G MFMA/s at the same board power), although both
// operands then change at every instruction (one operand pair and ONE accumulator for all 16: +16 %: the accumulator port costs
// more than both operand ports together).  In THIS kernel that order (GEMM_MMA_ORDER 2, bit-identical: every accumulator still
// sees k half 0, then k half 1 of every stage) measured equal for QKV / FC1 and 1 % slower for the residual GEMMs, whose
// instantiation then spills one accumulator in the last stage (three interleaved same-box pairs, tools/experiments/README.md):
// the pure stream runs into a ~1.33 kW limit of the matrix pipes alone, the GEMM into the 1.4 kW board cap with the clock
// already at 1.75 GHz, where a few per cent of MFMA energy move the clock by less than the noise.  Order 1 stays.
//   GEMM_MMA_ORDER 1: k half outer, serpentine (shipped) | 2: i, j serpentine, k half inner | 0: k half outer, j restarts at 0
#ifndef GEMM_MMA_ORDER
#define GEMM_MMA_ORDER 1
#endif
#if GEMM_MMA_ORDER == 2
#define MMA2(half, ZC)                                                                      \
  do {                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                        \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh) {                                      \
      const int j = (i & 1) ? 3 - j_ : j_;                                                  \
      acc[(half) * 4 + i][j] = mfma16(fb[kh * 4 + j], fa[kh * 4 + i],                                               \
                                   ((ZC) && kh == 0) ? CINIT(j) : acc[(half) * 4 + i][j]);                            \
    }                                                                                       \
  } while (0)
#else
#define MMA2(half, ZC)                                                                      \
  do {                                                                                      \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                      \
      const int j = (GEMM_MMA_ORDER && (i & 1)) ? 3 - j_ : j_;                              \
      acc[(half) * 4 + i][j] = mfma16(fb[kh * 4 + j], fa[kh * 4 + i],                                               \
                                   ((ZC) && kh == 0) ? CINIT(j) : acc[(half) * 4 + i][j]);                            \
    }                                                                                       \
  } while (0)
#endif
#define BARRIER() asm volatile("s_barrier" ::: "memory")
#define WAIT_LDS()                                                                          \
  do {                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
  // {reads retired} barrier {32 MFMA} barrier; the sched_barrier keeps the register-only MFMAs behind the s_barrier.
  // (Handing over early -- the second barrier in front of the last four MFMAs, so that the other wave row starts while
  // these drain -- measured 7-8 % SLOWER end to end, with 16- and with 32-MFMA phases.)
#define SYNC_MMA(half, ZC)                                                                  \
  do {                                                                                      \
    WAIT_LDS(); BARRIER(); __builtin_amdgcn_sched_barrier(0);                               \
    __builtin_amdgcn_s_setprio(1); MMA2(half, ZC); __builtin_amdgcn_s_setprio(0);           \
    BARRIER();                                                                              \
  } while (0)
  // The counted wait of a phase, issued behind the phase's own pieces: PA has just issued 2, PB 6, and the pieces that
  // must have landed are older than the 8 newest (vmcnt retires in order).  First two waits after an epilogue: what they
  // need was issued BEFORE the epilogue's stores, so when every wave issued exactly its 16 row stores (17 for the waves
  // that also store EPI_RESID statistics) those may stay outstanding as well and drain under the MFMAs instead of
  // stalling the pipeline at the head of every tile.  One opaque instruction for the compiler (a real branch here splits
  // the stage into basic blocks and costs ~20 spilled VGPRs): sel 0 -> vmcnt(8), 1 -> vmcnt(24), 2 -> vmcnt(25).
#define VM8 asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define VM_RELAX                                                                            \
  do {                                                                                      \
    const int sel_ = __builtin_amdgcn_readfirstlane(relax > 0 ? relax_sel : 0);             \
    relax = relax > 0 ? relax - 1 : 0;                                                      \
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lvm8_%=\n\ts_cmp_eq_u32 %0, 1\n\ts_cbranch_scc1 .Lvma_%=\n\t"          \
                 "s_waitcnt vmcnt(%2)\n\ts_branch .Lvmend_%=\n.Lvma_%=:\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Lvmend_%=\n"       \
                 ".Lvm8_%=:\n\ts_waitcnt vmcnt(8)\n.Lvmend_%=:" : : "s"(sel_), "n"(RELAX_A), "n"(RELAX_B) : "memory", "scc");    \
  } while (0)
#define ISSUE_WAH0(b, ablk, wblk, o00, o01, kbyte)                                          \
  do { ISSUE_W(b, wblk, kbyte); ISSUE_AH0(b, ablk, o00, o01, kbyte); } while (0)
  // one K=64 stage on buffer b = two phases of 32 MFMAs per wave.  PA: W (both k halves) and A(half 0), PA_ISSUE = the
  // A(half 1) rows of stage s+1 into the other buffer; PB: A(half 1), PB_ISSUE = W and A(half 0) of stage s+2 into this one
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
#define STAGE(b, VMWAIT, PA_ISSUE, PB_ISSUE) STAGE_Z(b, 0, VMWAIT, PA_ISSUE, PB_ISSUE)

  // ---- cold prologue of the first tile ----
  int tile_iter = 0;
  STAGE_STATS(0, cur.m0);
  // EPI_RESID: the bias of a tile's 256 columns is ONE piece (1 KiB), fetched by wave 0 into the free half of AUX a tile ahead
#define STAGE_BIAS(n0v)                                                                     \
  do {                                                                                      \
    if (EPI == EPI_RESID && w == 0) glds16_at((const char*)p.bias, (unsigned)(n0v) * 4u + (unsigned)lane * 16u, lds0 + (unsigned)(AUX_OFF + 8192)); \
  } while (0)
  STAGE_BIAS(cur.n0);
  ISSUE_WAH0(0, Ablk, Wblk, aoff00, aoff01, 0); ISSUE_AH1(0, Ablk, aoff10, aoff11, 0);
  ISSUE_WAH0(1, Ablk, Wblk, aoff00, aoff01, 128);           // A(half 1) of stage 1 follows in PA of stage 0
  VM8;                                                      // W and A(half 0) of stage 0 have landed
  BARRIER();
  int relax = 0;                             // waits of the coming tile that may leave the previous tile's stores in flight
  const int relax_sel = (EPI == EPI_RESID && w < 4) ? 2 : 1;   // those waves also store the row statistics: 17 stores, not 16
  // what may stay outstanding at the first two waits of a tile: the 8 newest pieces + the previous tile's 16 (17) row stores
  // + the column-sum / bias pieces issued at the top of the tile (2 for EPI_LNFOLD; EPI_STORE_BF16 issues one only with a
  // bias: counted as none, i.e. that wait is one piece stricter than it need be; EPI_RESID stages its bias a tile ahead)
  constexpr int CB_PIECES = EPI == EPI_LNFOLD ? 2 : 0;
  constexpr int RELAX_A = 24 + CB_PIECES, RELAX_B = 25 + CB_PIECES;

  for (;;) {
    f32x4_t acc[8][4];
    frag_t fa[8], fb[8];

#ifdef CLIPENC_DIAG
    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 1] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 4] = __builtin_amdgcn_s_memtime(); }
#endif
    // Column sums and biases of this tile's 64 columns per wave, staged into the wave's (idle until the epilogue) 2-KiB image
    // by LDS-DMA: [0, 256) colsum, [1024, 1280) bias (lanes 16..63 fetch duplicates).  As global loads at the head of the
    // epilogue they were the only VMEM loads there, and waiting for a load means waiting for every older DMA piece of the
    // next tile as well (vmcnt retires in order): 0.5-0.6 us per tile with the pipeline drained (tools/gemm_tile_boundary.py).
    // An extra piece only makes the counted waits below stricter (one more piece must have landed), never weaker; it has
    // retired before the epilogue because every path issues >= 8 younger pieces and passes a vmcnt(8) behind them.
    f32x4_t binit[4];                        // EPI_RESID: bias of the lane's columns = the first MFMAs' C operand
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) binit[nt] = *(const f32x4_t*)(smem + AUX_OFF + 8192 + (wc * 64 + nt * 16 + (lane >> 4) * 4) * 4);
    } else if (EPI == EPI_LNFOLD || p.bias) {
      const unsigned coff = (unsigned)(cur.n0 + wc * 64) * 4u + (unsigned)(lane & 15) * 16u;
      if (EPI == EPI_LNFOLD) glds16_at((const char*)p.colsum, coff, lds0 + (unsigned)(TR_OFF + w * 2048));
      glds16_at((const char*)p.bias, coff, lds0 + (unsigned)(TR_OFF + w * 2048 + 1024));
    }
    if (wr == 1) {
      if constexpr (EPI == EPI_LNFOLD) {
        // (mean, rstd) of this tile's 256 rows from the raw partial sums that the DMA left in AUX[buf] a tile ago (retired and
        // behind a barrier since) -- by the second wave row, which would otherwise only wait here for the first row's read
        // section; at the head of the epilogue this cost a workgroup barrier and 0.4 us per tile.  The tile's barriers order
        // these writes before the epilogue's reads.
        char* raw = smem + AUX_OFF + (tile_iter & 1) * 8192;
        const int row = tid - 256;
        float s_ = 0.f, ss_ = 0.f;
        for (int part = 0; part < p.stats_in_parts; ++part) {
          const float2 t = *(const float2*)(raw + part * 2048 + row * 8);
          s_ += t.x; ss_ += t.y;
        }
        const float mean = s_ * p.inv_width;
        const float var = fmaxf(ss_ * p.inv_width - mean * mean, 0.f);
        *(float2*)(raw + row * 8) = float2{mean, rsqrtf(var + p.eps)};
      }
      if (EPI != EPI_THRESH && w == 4 && idx + G >= S) {
        // the tile after this one comes from the ticket counters: take the ticket now (~1 us; this wave row waits here anyway)
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
        if (lane == 0) *(volatile int*)(smem + TICKET_SLOT) = cand;
      }
      BARRIER();                             // second wave row runs half a phase behind
    }

    if (kend > 256) {
      // first stage pair of the tile: every accumulator starts from the constant 0 in its first MFMA
      STAGE_Z(0, 1, VM_RELAX, ISSUE_AH1(1, Ablk, aoff10, aoff11, 128), ISSUE_WAH0(0, Ablk, Wblk, aoff00, aoff01, 256));
      STAGE(1, VM8, ISSUE_AH1(0, Ablk, aoff10, aoff11, 256), ISSUE_WAH0(1, Ablk, Wblk, aoff00, aoff01, 384));
      for (int kb = 256; kb < kend - 256; kb += 256) {
        STAGE(0, VM8, ISSUE_AH1(1, Ablk, aoff10, aoff11, kb + 128), ISSUE_WAH0(0, Ablk, Wblk, aoff00, aoff01, kb + 256));
        STAGE(1, VM8, ISSUE_AH1(0, Ablk, aoff10, aoff11, kb + 256), ISSUE_WAH0(1, Ablk, Wblk, aoff00, aoff01, kb + 384));
      }
    } else {                                 // (K = 128: the tile is its last stage pair)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = CINIT(j);
    }
    // ---- last two stages: the DMA crosses into the next tile ----
    int nidx = idx + G;
    if (EPI != EPI_THRESH && nidx >= S)      // (wave-uniform) written at the top of this tile, many barriers ago; read before the epilogue reuses the image
      nidx = __builtin_amdgcn_readfirstlane(*(volatile const int*)(smem + TICKET_SLOT));      // (-1: nothing left)
    const bool has_next = (unsigned)nidx < (unsigned)nwg;
    TileId nxt = cur;
    const char *Anext = Ablk, *Wnext = Wblk;
    if (has_next) {
      nxt = DECODE_TILE(nidx);
      Anext = (const char*)p.A + (size_t)nxt.m0 * lda_b;
      Wnext = (const char*)p.W + (size_t)nxt.n0 * ldw_b;
    }
    // The next tile's A offsets take over this tile's registers as they die (half 0 here, half 1 between the last two
    // stages; without a next tile nxt == cur and they are recomputed to the same values).  They are re-derived from an
    // opaque copy of the lane id: as loop invariants the constants would be kept (spilled) across the main loop, and a
    // spill reload here carries a compiler-counted vmcnt wait that drains the DMA pipeline.
    int lane_b = lane;
    asm volatile("" : "+v"(lane_b));
    const int dg_b = lane_b >> 3;
    const unsigned dch_b = (unsigned)(((lane_b & 7) ^ (dg_b & 6)) * 16);
    const int arow_b = (w >> 2) * 128 + (2 * (w & 3)) * 8 + dg_b;
#define AOFF_B(r) ((unsigned)((min(nxt.m0 + (r), p.M - 1) - nxt.m0) * lda_b) + dch_b)
    aoff00 = AOFF_B(arow_b); aoff01 = AOFF_B(arow_b + 8);
    {
      const int kb = kend - 256;
      // ONE code path: without a next tile the DMA harmlessly re-fetches this tile's first stages into dead
      // buffers (two variants of this block made hipcc spill ~270 VGPRs)
      if (has_next) { STAGE_STATS((tile_iter + 1) & 1, nxt.m0); STAGE_BIAS(nxt.n0); }
      STAGE(0, VM_RELAX, ISSUE_AH1(1, Ablk, aoff10, aoff11, kb + 128), ISSUE_WAH0(0, Anext, Wnext, aoff00, aoff01, 0));
      aoff10 = AOFF_B(arow_b + 64); aoff11 = AOFF_B(arow_b + 72);
      STAGE(1, VM8, ISSUE_AH1(0, Anext, aoff10, aoff11, 0), ISSUE_WAH0(1, Anext, Wnext, aoff00, aoff01, 128));
    }
#undef AOFF_B
    // (EPI_THRESH re-aligns the two wave rows here; the other epilogues first issue their residual loads and read their
    //  column constants -- wave-private work -- and re-align in front of the first 16-row block)
    if (EPI == EPI_THRESH && wr == 0) BARRIER();
#ifdef CLIPENC_DIAG
    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 2] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 5] = __builtin_amdgcn_s_memtime(); }
#endif

    // ------------------------------- epilogue of tile `cur` -------------------------------
    if constexpr (EPI == EPI_THRESH) {
      // /root/reference/_2_remove_duplicates.py:74: where(triu(S, 1) > threshold) -> (i, j, S[i][j]), appended through one
      // atomic counter; lane (row = lane & 15, quad = lane >> 4) holds columns 4 quad + 0..3 of each 16 x 16 block
      const float thr = p.fp16_compare ? (float)(_Float16)p.thr : p.thr;
      const int frow_t = lane & 15, q4t = (lane >> 4) * 4;
      // Screen first: almost every tile holds no value anywhere near the threshold (random unit vectors have cosines of a
      // few hundredths), and the exact test below costs ~7 VALU instructions per value (fp16 rounding, index tests): one
      // running maximum per lane and a wave vote skip it.  The margin covers the fp16 rounding of the value (half an ulp of
      // fp16 at 1.0 = 4.9e-4), so no qualifying pair can hide behind the screen.
      float vmax = -1e30f;
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
        {   // two v_max3_f32 per accumulator (fmaxf puts a canonicalising v_max in front of every MFMA output: 3x the instructions)
          asm("v_max3_f32 %0, %0, %1, %2" : "+v"(vmax) : "v"(acc[mt][nt][0]), "v"(acc[mt][nt][1]));
          asm("v_max3_f32 %0, %0, %1, %2" : "+v"(vmax) : "v"(acc[mt][nt][2]), "v"(acc[mt][nt][3]));
        }
      if (__builtin_amdgcn_ballot_w64(vmax > thr - 2.0e-3f * fmaxf(1.0f, fabsf(thr))) != 0ull) {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const int i = cur.m0 + wr * 128 + mt * 16 + frow_t;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = cur.n0 + wc * 64 + nt * 16 + q4t + e;
            float v = acc[mt][nt][e];
            if (p.fp16_compare) v = (float)(_Float16)v;
            if (j > i && j < p.n_valid && v > thr) {
              const unsigned long long slot = atomicAdd(p.count, 1ull);
              if (slot < p.cap) {
                p.pairs[slot * 2 + 0] = i;
                p.pairs[slot * 2 + 1] = j;
                p.vals[slot] = v;
              }
            }
          }
        }
      }
      }
    } else {
    // epilogue lane mapping: 16-row x 128-B image per wave, 16-B chunk index XOR row&7 -- derived here from an opaque copy
    // of the lane id, so that hipcc does not carry these constants through the main loop
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    char* tr = smem + TR_OFF + w * 2048;
    const int frow_e = lane_e & 15, qd = lane_e >> 4;
    const int tw_base = frow_e * 128 + (qd & 1) * 8;
    const int tw_sw = frow_e & 7;
    const int tr_base = (lane_e >> 3) * 128 + (((lane_e & 7) ^ (lane_e >> 3)) << 4);   // + 1024 for rows 8..15
    const int row_l = lane_e >> 3;
    // Row stores and residual loads go through buffer descriptors that cover exactly the tile's existing rows: the hardware
    // drops (stores) / zero-fills (loads) the rows of a ragged last tile, so the blocks below are straight-line code
    // without per-row exec-mask branches.  Descriptors are built from wave-uniform values only (SGPRs, no waterfall loop).
    const unsigned row_bytes = (unsigned)p.ldo * 2u;
    const unsigned tile_bytes = (unsigned)min(p.M - cur.m0, BM) * row_bytes;                  // <= 256 rows x 8 KiB
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((char*)p.out + (size_t)cur.m0 * row_bytes, 0, (int)tile_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((char*)p.resid + (size_t)cur.m0 * row_bytes, 0, (int)tile_bytes, 0x00020000);
    const unsigned lcol_b = (unsigned)(cur.n0 + wc * 64 + (lane_e & 7) * 8) * 2u;              // byte column of the lane's 16-B piece

    // column sums / biases of the wave's columns: from the image the DMA filled at the top of the tile (no VMEM load here)
    f32x4_t cs[4], bs[4];
    if constexpr (EPI == EPI_LNFOLD) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) cs[nt] = *(const f32x4_t*)(tr + nt * 64 + qd * 16);
    }
    if (EPI == EPI_LNFOLD || (EPI == EPI_STORE_BF16 && p.bias)) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bs[nt] = *(const f32x4_t*)(tr + 1024 + nt * 64 + qd * 16);
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bs[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#ifdef CLIPENC_DIAG
    if (p.dbg && tid == 0) p.dbg[(size_t)idx * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif

    // residual rows: ALL 16 x 16 B per lane of the wave's 128 rows are requested at once, as soon as the last MFMA phase has
    // released the fragment registers and before the wave rows re-align.  They come from HBM (the residual stream is 1 GB,
    // written a whole block of kernels ago) and, behind the 8 DMA pieces of the next tile in the in-order vmcnt queue, take
    // ~2 us; with 8 in flight and the rest requested four blocks ahead the epilogue paid that latency more than twice
    // (tools/gemm_tile_boundary.py: 7.4 us per tile for out-proj / FC2 against 2.0 us for QKV).
    uint4 rres[16];
#define LOAD_RES(k)                                                                           \
  do {                                                                                        \
    const unsigned off_ = (unsigned)(wr * 128 + (k) * 8 + row_l) * row_bytes + lcol_b;        \
    rres[(k)] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, off_, 0, 0)); \
  } while (0)
    // EPI_RESID: the 0/1 column selectors of the residual MFMA as A operands -- lane (c = lane & 15, k group qd) holds
    // k = 8 qd .. 8 qd + 7 of selector row c: e_lo[c][k] = (k == c), e_hi[c][k] = (k == 16 + c) -- and the all-ones operand
    frag_t e_lo, e_hi, ones8;
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
      for (int k = 0; k < 16; ++k) LOAD_RES(k);
      u32x4_t lo, hi;
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int k0 = 8 * qd + 2 * pp;
        lo[pp] = (k0 == frow_e ? 0x3f80u : 0u) | (k0 + 1 == frow_e ? 0x3f800000u : 0u);
        hi[pp] = (k0 == 16 + frow_e ? 0x3f80u : 0u) | (k0 + 1 == 16 + frow_e ? 0x3f800000u : 0u);
      }
      e_lo = __builtin_bit_cast(frag_t, lo); e_hi = __builtin_bit_cast(frag_t, hi);
      ones8 = __builtin_bit_cast(frag_t, u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
    }
    if (wr == 0) BARRIER();                  // re-align the two wave rows for the epilogue

#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      uint2 pk[4];
      if constexpr (EPI == EPI_STORE_BF16) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const f32x4_t v = acc[mt][nt] + bs[nt];
          pk[nt] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
      } else if constexpr (EPI == EPI_LNFOLD) {
        const float2 t = *(const float2*)(smem + AUX_OFF + (tile_iter & 1) * 8192 + (wr * 128 + mt * 16 + frow_e) * 8);
        const float mean = t.x, rstd = t.y;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          // whole-vector arithmetic so that hipcc emits v_pk_fma / v_pk_mul / v_pk_add (two elements per instruction);
          // only exp2 and rcp stay per element
          f32x4_t v = rstd * (acc[mt][nt] - mean * cs[nt]) + bs[nt];
          if constexpr (ACT == CE_ACT_QUICK_GELU) {
            f32x4_t t = v * -2.4554669595930156f;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_amdgcn_exp2f(t[e]);
            t = t + 1.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = __builtin_amdgcn_rcpf(t[e]);
            v = v * t;
          } else if constexpr (ACT == CE_ACT_GELU_ERF) {
            v = ce_gelu_erf4(v);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_apply_t<ACT>(v[e]);
          }
          pk[nt] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
      } else {
        // Residual add and row statistics on the MATRIX pipe, which idles in the epilogue (as VALU work -- unpack, add, two
        // dot products per packed pair, cross-lane sums -- this block was 75 instructions per wave and the epilogue of
        // out-proj / FC2 took 7.4 us per tile against 2.0 us for QKV: tools/gemm_tile_boundary.py).
        // (1) the block's residual rows go through the row-major image as before, but are read back as an MFMA B operand:
        //     lane (row frow, k group qd) takes the whole 16-B piece 4 h + qd of its row (8 consecutive columns), and
        //     D[c][row] += sum_k E[c][k] R[row][32 h + k] with the 0/1 selector E (k == c, or k == 16 + c for the odd column
        //     block) adds exactly R[row][col] to the accumulator that already holds bias + A.W^T.
        *(uint4*)(tr + tr_base) = rres[mt * 2];
        *(uint4*)(tr + 1024 + tr_base) = rres[mt * 2 + 1];
        // (no wait: the LDS serves one wave's accesses in order)
        const frag_t r0 = *(const frag_t*)(tr + frow_e * 128 + ((qd ^ tw_sw) << 4));
        const frag_t r1 = *(const frag_t*)(tr + frow_e * 128 + (((4 + qd) ^ tw_sw) << 4));
        acc[mt][0] = mfma16(e_lo, r0, acc[mt][0]); acc[mt][1] = mfma16(e_hi, r0, acc[mt][1]);
        acc[mt][2] = mfma16(e_lo, r1, acc[mt][2]); acc[mt][3] = mfma16(e_hi, r1, acc[mt][3]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) pk[nt] = uint2{pack_bf16x2(acc[mt][nt][0], acc[mt][nt][1]), pack_bf16x2(acc[mt][nt][2], acc[mt][nt][3])};
        // (2) sum and sum of squares of the ROUNDED row: the packed pairs of two column blocks are 8 distinct columns per lane,
        //     i.e. a valid operand (any column order sums the same): ones . P^T gives the row sum in every register of the
        //     row's lanes, P . P^T has the row's sum of squares on its diagonal (products of bf16 are exact in fp32)
        const frag_t p01 = __builtin_bit_cast(frag_t, u32x4_t{pk[0].x, pk[0].y, pk[1].x, pk[1].y});
        const frag_t p23 = __builtin_bit_cast(frag_t, u32x4_t{pk[2].x, pk[2].y, pk[3].x, pk[3].y});
        f32x4_t d1 = mfma16(ones8, p01, f32x4_t{0.f, 0.f, 0.f, 0.f});
        d1 = mfma16(ones8, p23, d1);
        f32x4_t d2 = mfma16(p01, p01, f32x4_t{0.f, 0.f, 0.f, 0.f});
        d2 = mfma16(p23, p23, d2);
        // D[m][row]: lane (row, qd) register i is m = 4 qd + i, so the diagonal sits in lane qd == row / 4, register row % 4
        const int di = frow_e & 3;
        const float ssq = di == 0 ? d2[0] : (di == 1 ? d2[1] : (di == 2 ? d2[2] : d2[3]));
        if (qd == (frow_e >> 2))
          *(float2*)(smem + AUX_OFF + ((size_t)wc * 256 + wr * 128 + mt * 16 + frow_e) * 8) = float2{d1[0], ssq};
      }
      // fragment layout -> row-major image -> two 16-B-per-lane stores of 8 full rows each
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) *(uint2*)TW_ADDR(nt) = pk[nt];
      // (no wait between the image writes and the row-major reads, nor before the next block's writes: one wave, in order)
      const uint4 v0 = *(const uint4*)(tr + tr_base);
      const uint4 v1 = *(const uint4*)(tr + 1024 + tr_base);
      const unsigned oa = (unsigned)(wr * 128 + mt * 16 + row_l) * row_bytes + lcol_b;
      // The QKV / FC1 outputs (EPI_LNFOLD) are written once and read by another kernel: they leave with the streaming policy, so that an XCD
      // round's 4 MB of output rows do not push operand panels and the attention kernel's rows out of the L2 / Infinity Cache on their way.
      // Six interleaved pairs, same box: 1 915-1 926 -> 1 929-1 935 images/s (+0.6 %, every pair), QKV 60.2 -> 59.5 ms, FC1 86.2 -> 85.5, attention
      // 22.9 -> 22.7 (round 2 had measured three pairs "equal"; GEMM_NT_STORES=0: plain stores).
#ifndef GEMM_NT_STORES
#define GEMM_NT_STORES 1
#endif
      constexpr int ST_AUX = (GEMM_NT_STORES && EPI == EPI_LNFOLD) ? 2 : 0;     // 2 = nt (streaming)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v0), orsrc, oa, 0, ST_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v1), orsrc, oa + 8 * row_bytes, 0, ST_AUX);
#ifdef CLIPENC_DIAG
      if (mt == 0 && p.dbg && tid == 0) p.dbg[(size_t)idx * 8 + 7] = __builtin_amdgcn_s_memrealtime();   // first 16-row block out: column sums / bias / residual have arrived
#endif
    }

    if constexpr (EPI == EPI_RESID) {
      __syncthreads();
      if (tid < 256 && cur.m0 + tid < p.M) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float2 t = *(const float2*)(smem + AUX_OFF + ((size_t)c * 256 + tid) * 8);
          s += t.x; ss += t.y;
        }
        *(float2*)(p.stats_out + ((size_t)cur.tn * p.stats_ld + cur.m0 + tid) * 2) = float2{s, ss};
      }
      __syncthreads();                       // AUX is rewritten by the next tile's epilogue
    }

    }

#ifdef CLIPENC_DIAG
    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 3] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 0] = blockIdx.x; }
#endif
    if (!has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the redundant DMA before the LDS is released
      break;
    }
    // all 256 rows valid: every guarded store above was issued (the waves of EPI_RESID that store statistics issued 17)
#ifdef CLIPENC_DIAG
    relax = (EPI != EPI_THRESH && cur.m0 + BM <= p.M && p.dbg == nullptr) ? 2 : 0;   // (the stamps add stores: no relaxed waits then)
#else
    relax = (EPI != EPI_THRESH && cur.m0 + BM <= p.M) ? 2 : 0;      // the first two waits of the coming tile (EPI_THRESH issues a data-dependent number of stores: never relaxed)
#endif
    idx = nidx; cur = nxt; Ablk = Anext; Wblk = Wnext;
    ++tile_iter;
  }
}

// grid of the persistent launch: one workgroup per CU, a multiple of 8 so that tile index mod 8 == workgroup index mod 8
// (workgroups b and b + 8 share an XCD under the round-robin dispatch: speed only, never correctness)
int persist_grid(int n_cu, int tiles) {
  int grid = n_cu > 0 ? n_cu : 256;
  grid -= grid % 8;
  if (grid < 8) grid = 8;
  if (tiles < grid) grid = tiles;
  return grid;
}

template <int EPI, int ACT>
hipError_t launch_persist(const GemmParams& p, hipStream_t stream) {
  static DeviceKernelSetup setup;             // per device: LDS opt-in + CU count (common.h)
  int n_cu = 0;
  if (hipError_t e = setup.ensure((const void*)gemm_persist_kernel<EPI, ACT>, LDS_BYTES, &n_cu); e != hipSuccess) return e;
  int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  if (EPI == EPI_THRESH) { const int tt = p.N / BN; tiles = tt * (tt + 1) / 2; }
  const int grid = persist_grid(n_cu, tiles);
  hipLaunchKernelGGL((gemm_persist_kernel<EPI, ACT>), dim3(grid), dim3(512), LDS_BYTES, stream, p);
  return hipGetLastError();
}

}  // namespace

#ifdef GEMM_PERSIST_TRI_TU
// gemm_tri.hip: the same kernel template instantiated in a translation unit of its own for the near-duplicate search (f16
// operands, upper-triangular tile list, threshold + append epilogue), so that it cannot perturb the code generation of the
// encoder's instantiations (guide rule 19).
#include <mutex>
#include <vector>

// Order of the upper-triangular tile list (TT x TT tiles, tn >= tm) for a grid of G workgroups, G / 8 per XCD.
// Entry i = the tile of workgroup i % G in its round i / G.  What the order is for: the G / 8 tiles that are resident on ONE
// XCD at a time should touch as few 256-row operand panels as possible, because every panel a tile needs and its neighbours
// do not is a read that misses the XCD's 4 MiB L2.  A row-major walk (round 2) gave an XCD 32 tiles of one tile row:
// 1 shared A panel + 32 different W panels = 33 panels per 32 tiles (rocprofv3: 35 GB of L2-miss reads per launch for a
// 0.15 GB operand).  Here the grid is cut into super-blocks of 8 (tm) x 4 (tn) tiles = 12 panels per 32 tiles; the FULL
// super-blocks come first, super-row-major, one super-block per XCD and round, so the 8 XCDs of a round hold 8 neighbouring
// super-blocks of one super-row (the same 8 A panels, hot in the Infinity Cache) and an XCD's consecutive rounds stay in that
// super-row (its A panels may still be in its L2); the tiles of the partial super-blocks along the diagonal and the grid
// edge follow, compacted.  Every valid tile appears exactly once; the rounds are full except the last.
std::vector<unsigned> tri_tile_order(int TT, int G) {
  const long long total = (long long)TT * (TT + 1) / 2;
  std::vector<unsigned> seq;
  seq.reserve((size_t)total);
  constexpr int SBM = 8, SBN = 4;
  const int SR = (TT + SBM - 1) / SBM, SC = (TT + SBN - 1) / SBN;
  auto valid = [&](int tm, int tn) { return tm < TT && tn < TT && tn >= tm; };
  for (int pass = 0; pass < 2; ++pass)          // pass 0: full super-blocks, pass 1: the tiles of the partial ones
    for (int R = 0; R < SR; ++R)
      for (int C = 0; C < SC; ++C) {
        if (C * SBN + SBN - 1 < R * SBM) continue;                           // wholly below the diagonal
        const bool full = valid(R * SBM + SBM - 1, C * SBN) && C * SBN + SBN - 1 < TT;
        if (full != (pass == 0)) continue;
        for (int j = 0; j < SBN; ++j)
          for (int i = 0; i < SBM; ++i)
            if (valid(R * SBM + i, C * SBN + j)) seq.push_back((unsigned)(R * SBM + i) | ((unsigned)(C * SBN + j) << 16));
      }
  std::vector<unsigned> order(seq.size());
  const int lpx = G / 8;                          // workgroups per XCD
  const size_t grouped = (G >= 8 && G % 8 == 0) ? seq.size() / (size_t)G * (size_t)G : 0;   // whole rounds
  for (size_t e = 0; e < seq.size(); ++e) {
    size_t idx = e;
    if (e < grouped) {                            // element e of a round: XCD slot q takes lpx consecutive tiles
      const size_t round = e / (size_t)G, w = e % (size_t)G, q = w / (size_t)lpx, l = w % (size_t)lpx;
      idx = round * (size_t)G + l * 8 + q;
    }
    order[idx] = seq[e];
  }
  return order;
}

namespace {
struct TriOrderEntry { int device, tt, grid; unsigned* dev_list; };
std::mutex g_tri_mu;
std::vector<TriOrderEntry> g_tri_cache;         // a handful of (device, problem size) pairs per process; oldest evicted
}  // namespace

// The device copy of tri_tile_order(tt, grid) for the current device (made at the first use of a problem size, cached).
hipError_t ce_tri_tile_list(int tt, int grid, const unsigned** dev_list) {
  int device = 0;
  if (hipError_t e = hipGetDevice(&device); e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(g_tri_mu);
  for (const auto& c : g_tri_cache)
    if (c.device == device && c.tt == tt && c.grid == grid) { *dev_list = c.dev_list; return hipSuccess; }
  const std::vector<unsigned> order = tri_tile_order(tt, grid);
  unsigned* d = nullptr;
  if (hipError_t e = hipMalloc((void**)&d, order.size() * sizeof(unsigned)); e != hipSuccess) return e;
  // (first use of a problem size only; a blocking copy, so the list is complete before any launch can read it)
  if (hipError_t e = hipMemcpy(d, order.data(), order.size() * sizeof(unsigned), hipMemcpyHostToDevice); e != hipSuccess) {
    (void)hipFree(d);
    return e;
  }
  if (g_tri_cache.size() >= 8) {              // the evicted list may still be read by a launch in flight: wait for the device
    (void)hipDeviceSynchronize();
    (void)hipFree(g_tri_cache.front().dev_list);
    g_tri_cache.erase(g_tri_cache.begin());
  }
  g_tri_cache.push_back(TriOrderEntry{device, tt, grid, d});
  *dev_list = d;
  return hipSuccess;
}

hipError_t ce_gemm_tri_persist(const GemmParams& p_in, hipStream_t stream) {
  GemmParams p = p_in;
  const int tt = p.N / BN;
  if (tt > 0xffff) return hipErrorInvalidValue;                              // tile coordinates are packed 16 + 16 bits
  static DeviceKernelSetup setup;
  int n_cu = 0;
  if (hipError_t e = setup.ensure((const void*)gemm_persist_kernel<EPI_THRESH, -1>, LDS_BYTES, &n_cu); e != hipSuccess) return e;
  const int grid = persist_grid(n_cu, tt * (tt + 1) / 2);
  if (hipError_t e = ce_tri_tile_list(tt, grid, &p.tile_list); e != hipSuccess) return e;
  return launch_persist<EPI_THRESH, -1>(p, stream);
}
#else
hipError_t ce_gemm_nt_persist(const GemmParams& p, int epi, hipStream_t stream) {
  switch (epi) {
    case EPI_STORE_BF16: return launch_persist<EPI_STORE_BF16, -1>(p, stream);
    case EPI_LNFOLD:     // the activation is a template parameter: a run-time switch made hipcc evaluate both GELUs per element
      if (p.act == CE_ACT_QUICK_GELU) return launch_persist<EPI_LNFOLD, CE_ACT_QUICK_GELU>(p, stream);
      if (p.act == CE_ACT_GELU_ERF) return launch_persist<EPI_LNFOLD, CE_ACT_GELU_ERF>(p, stream);
      return launch_persist<EPI_LNFOLD, -1>(p, stream);
    case EPI_RESID: return launch_persist<EPI_RESID, -1>(p, stream);
    default: return hipErrorInvalidValue;
  }
}
#endif
