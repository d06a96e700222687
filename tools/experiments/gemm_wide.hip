// bf16 "NT" GEMM, wide-wave variant of gemm_persist.hip: the same persistent 256x256 tiles, the same two LDS buffers of
// K=64 stages (8-row x 128-B blocks, every LDS-DMA piece = 8 whole cache lines, chunk swizzle c ^ (r & 6)) fed across
// tile boundaries and the same three epilogues, but FOUR waves per workgroup (one per SIMD, 2 x 2), each owning a
// 128 x 128 corner of the tile with its 256 accumulator registers in the upper half of the 512-entry register file.
// Against the eight-wave kernel (128 x 64 per wave) that is 1/3 less LDS fragment traffic per MFMA, half the waves and
// ONE barrier per K=64 stage.  One wave per SIMD has nobody to hide behind, so a stage is software-pipelined in
// registers with single-buffered fragments; per stage s (buffer b) and wave, two K=32 halves of two passes each:
//   H0 pass 0: 32 MFMAs (A rows 0..7 x W cols 0..3)   | reads fb[4..7](s, k 0..31)
//   H0 pass 1: 32 MFMAs (cols 4..7)                   | reads fa[i](s, k 32..63) behind row i, fb[0..3](s, k 32..63)
//   H1 pass 0: 32 MFMAs (cols 0..3)                   | reads fb[4..7](s, k 32..63)
//   wait for the wave's 16 pieces of stage s+1, lgkmcnt(0), barrier      (buffer b is dead, buffer b^1 has landed)
//   H1 pass 1: 32 MFMAs (cols 4..7)                   | reads fa[i], fb[0..3](s+1, k 0..31) | the 16 pieces of stage s+2 -> b
#include <stdlib.h>
#include <stdlib.h>

#include "common.h"
#include "gemm.h"

namespace {

constexpr int BM = 256, BN = 256;
constexpr int BUF = 65536, WREG = 32768;    // one K=64 stage: A region | W region
constexpr int RING = 2 * BUF;               // 131072
constexpr int AUX_OFF = RING;               // 16 KiB: LNFOLD raw row statistics (2 buffers) / RESID per-wave-column row sums
constexpr int TR_OFF = RING + 16384;        // 4 x 4 KiB wave-private images: 16 rows x 256 B
constexpr int LDS_BYTES = RING + 32768;     // 163840

#define LDS_PTR(off) ((__attribute__((address_space(3))) void*)(smem + (off)))

template <int ACT>
__device__ __forceinline__ float act_apply_t(float u) {
  if constexpr (ACT == CE_ACT_QUICK_GELU) return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * u));
  else if constexpr (ACT == CE_ACT_GELU_ERF) return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
  else return u;
}

// One LDS-DMA piece: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset) to 1 KiB of LDS at lds_addr.
// Inline asm on purpose: behind the builtin LLVM books every LDS-DMA as a FLAT access pending on BOTH counters and, as
// this kernel's vmcnt waits are hand-placed, never sees it retire -- every later LDS dependency then becomes
// `s_waitcnt lgkmcnt(0)` instead of a counted wait, which serialises the register-pipelined fragment reads.
// (m0 is written; nothing else in this file uses it.)
__device__ __forceinline__ void glds16_at(const char* base, unsigned off, unsigned lds_addr_) {
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane(lds_addr_);   // uniform by construction; tells the compiler so
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_addr) : "memory");
}

struct TileId { int m0, n0, tn; };

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
  return TileId{tm * BM, tn * BN, tn};
}

template <int EPI, int ACT>
__global__ __launch_bounds__(256) void gemm_wide_kernel(const GemmParams p) {
  typedef bf16x8_t frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wc = w & 1;
  const int frow = lane & 15, qd = lane >> 4;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  const int nwg = tiles_m * tiles_n;
  const int G = gridDim.x;
  const size_t lda_b = (size_t)p.lda * 2, ldw_b = (size_t)p.ldw * 2;
  const int kend = p.K * 2;                  // bytes along K; one stage = 128 B; K % 128 == 0 (stages come in pairs)

  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(0));   // LDS address of smem[0]
  // LDS-DMA: lane L fetches logical 16-B chunk (L&7) ^ ((L>>3)&6) of row L>>3 of an 8-row block; wave w fills the blocks
  // 8w..8w+7 (rows 64w + 8q + dg) of both operands: 16 pieces per stage
  const int dg = lane >> 3;
  const unsigned dchunk16 = (unsigned)(((lane & 7) ^ (dg & 6)) * 16);
  const int drow0 = 64 * w + dg;                         // + 8q
  const int dma_lds = 8 * w * 1024;                      // + buffer (+ WREG) + q*1024
  const int rdl = (frow >> 3) * 1024 + (frow & 7) * 128 + (((qd ^ (frow & 6))) << 4);   // k 0..31; k 32..63 is ^64
  const int a_rd = wr * 16 * 1024 + rdl;                 // + buffer + i*2048
  const int w_rd = WREG + wc * 16 * 1024 + rdl;          // + buffer + j*2048
  const unsigned woff = (unsigned)(drow0 * ldw_b) + dchunk16;   // + q*8*ldw_b through the scalar base

  // epilogue image: 16 rows x 256 B per wave, 16-B chunk index XOR row
  char* tr = smem + TR_OFF + w * 4096;
  const int tw_base = frow * 256 + (qd & 1) * 8;
#define TW_ADDR(nt) (tr + tw_base + ((((nt) * 2 + (qd >> 1)) ^ frow) << 4))
  const int rrow = lane >> 4, rchunk = lane & 15;       // row-major side: rows rrow + 4k, logical chunk rchunk
#define TR_ADDR(k) (tr + (rrow + 4 * (k)) * 256 + ((rchunk ^ (rrow + 4 * (k))) << 4))

  int idx = blockIdx.x;
  TileId cur = decode_tile(idx, tiles_m, tiles_n);
  const char* Ablk = (const char*)p.A + (size_t)cur.m0 * lda_b;
  const char* Wblk = (const char*)p.W + (size_t)cur.n0 * ldw_b;
  unsigned aoff[8], naoff[8];                 // this tile's / the next tile's clamped A row offsets
#define SET_AOFF(o, m0v) _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) o[q_] = (unsigned)((min((m0v) + drow0 + 8 * q_, p.M - 1) - (m0v)) * lda_b) + dchunk16
  SET_AOFF(aoff, cur.m0);

  // piece q of a stage: q < 8 = A block 8w+q, else W block 8w+q-8
#define PIECE(b, ao, ablk, wblk, kbyte, q)                                                  \
  do {                                                                                      \
    if ((q) < 8) glds16_at((ablk) + (kbyte), ao[(q) & 7], lds0 + (unsigned)((b) * BUF + dma_lds + ((q) & 7) * 1024)); \
    else glds16_at((wblk) + (kbyte) + (size_t)((q) & 7) * 8 * ldw_b, woff, lds0 + (unsigned)((b) * BUF + WREG + dma_lds + ((q) & 7) * 1024)); \
  } while (0)
#define STAGE_ALL(b, ablk, wblk, kbyte)                                                     \
  do {                                                                                      \
    _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) PIECE(b, aoff, ablk, wblk, kbyte, q_); \
  } while (0)
#define glds16(base, off, smem_, lds_off) glds16_at((base), (off), lds0 + (unsigned)(lds_off))
  // raw row statistics of a tile's 256 rows: parts x 2 KiB, fetched by waves 0 and 1 (EPI_LNFOLD); always issued BEFORE
  // the stage's own pieces, so that the counted waits (newest 16 outstanding) only ever leave stage pieces in flight
#define STAGE_STATS(buf, m0v)                                                               \
  do {                                                                                      \
    if (EPI == EPI_LNFOLD && w < 2) {                                                        \
      for (int part = 0; part < p.stats_in_parts; ++part)                                    \
        glds16((const char*)p.stats_in + ((size_t)part * p.stats_ld + (m0v)) * 8, (unsigned)((w * 64 + lane) * 16), smem, \
               AUX_OFF + (buf) * 8192 + part * 2048 + w * 1024);                             \
    }                                                                                       \
  } while (0)
#define RD_A(b, kh, i) fa[i] = *(const frag_t*)(smem + (b) * BUF + (a_rd ^ ((kh) * 64)) + (i) * 2048)
#define RD_W(b, kh, j) fb[j] = *(const frag_t*)(smem + (b) * BUF + (w_rd ^ ((kh) * 64)) + (j) * 2048)
#define BARRIER() asm volatile("s_barrier" ::: "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)
  // The MFMA as an asm statement with the accumulator tied to an AGPR quad: with the builtin the register allocator
  // rotated accumulators through arch VGPRs in this loop (hundreds of v_accvgpr_read/write/mov per iteration).
#define MMA_ROW(i, j0)                                                                      \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][(j0) + j]) : "v"(fb[(j0) + j]), "v"(fa[i]))
  // stage s+1 retired: its 16 pieces are the wave's only outstanding ones (vmcnt(0)) -- except right after an epilogue,
  // whose 32 row stores (33 with the EPI_RESID statistics) were issued BEHIND them and may stay in flight
#define VM0 asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define VM_FIRST                                                                            \
  do {                                                                                      \
    const int sel_ = __builtin_amdgcn_readfirstlane(relax);                                 \
    relax = 0;                                                                              \
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lwv0_%=\n\ts_cmp_eq_u32 %0, 1\n\ts_cbranch_scc1 .Lwv32_%=\n\t"          \
                 "s_waitcnt vmcnt(33)\n\ts_branch .Lwvend_%=\n.Lwv32_%=:\n\ts_waitcnt vmcnt(32)\n\ts_branch .Lwvend_%=\n"       \
                 ".Lwv0_%=:\n\ts_waitcnt vmcnt(0)\n.Lwvend_%=:" : : "s"(sel_) : "memory", "scc");                                  \
  } while (0)
  // one K=64 stage on buffer b; the 16 pieces of stage s+2 (operand block pointers nab/nwb, byte offset nkb) go to buffer
  // b behind the barrier.  PREF = 0: last stage of a tile (the next tile's first fragments are read at its top).
#define WSTAGE(b, VMWAIT, nao, nab, nwb, nkb, PREF)                                              \
  do {                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                         \
      MMA_ROW(i, 0); SB();                                                                  \
      if (i < 4) RD_W(b, 0, 4 + i);                                                         \
      SB();                                                                                 \
    }                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                         \
      MMA_ROW(i, 4); SB();                                                                  \
      RD_A(b, 1, i);                                                                        \
      if (i < 4) RD_W(b, 1, i);                                                             \
      SB();                                                                                 \
    }                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                         \
      MMA_ROW(i, 0); SB();                                                                  \
      if (i < 4) RD_W(b, 1, 4 + i);                                                         \
      SB();                                                                                 \
    }                                                                                       \
    VMWAIT;                                                                                 \
    BARRIER();                                                                              \
    SB();                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                         \
      MMA_ROW(i, 4); SB();                                                                  \
      if (PREF) RD_A((b) ^ 1, 0, i);                                                        \
      if (PREF && i < 4) RD_W((b) ^ 1, 0, i);                                               \
      PIECE(b, nao, nab, nwb, nkb, 2 * i); PIECE(b, nao, nab, nwb, nkb, 2 * i + 1);                 \
      SB();                                                                                 \
    }                                                                                       \
  } while (0)

  // ---- cold prologue of the first tile: stages 0 and 1 ----
  int tile_iter = 0;
  STAGE_STATS(0, cur.m0);
  STAGE_ALL(0, Ablk, Wblk, 0);
  STAGE_ALL(1, Ablk, Wblk, 128);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  BARRIER();
  int relax = 0;                             // 1 / 2: the coming tile's first wait may leave the previous tile's 32 / 33 stores in flight

  for (;;) {
    f32x4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    frag_t fa[8], fb[8];

    const int nidx = idx + G;
    const bool has_next = nidx < nwg;
    TileId nxt = cur;
    const char *Anext = Ablk, *Wnext = Wblk;
    if (has_next) {
      nxt = decode_tile(nidx, tiles_m, tiles_n);
      Anext = (const char*)p.A + (size_t)nxt.m0 * lda_b;
      Wnext = (const char*)p.W + (size_t)nxt.n0 * ldw_b;
      STAGE_STATS((tile_iter + 1) & 1, nxt.m0);
      SET_AOFF(naoff, nxt.m0);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) naoff[q] = aoff[q];
    }
    // fragments of the tile's first half stage (buffer 0 landed at least one barrier ago); fb[4..7] follow in its pass 0
#pragma unroll
    for (int j = 0; j < 4; ++j) RD_W(0, 0, j);
#pragma unroll
    for (int i = 0; i < 8; ++i) RD_A(0, 0, i);
    if (p.dbg && tid == 0) p.dbg[(size_t)idx * 8 + 1] = __builtin_amdgcn_s_memrealtime();

    for (int kb = 0; kb < kend - 256; kb += 256) {
      WSTAGE(0, VM_FIRST, aoff, Ablk, Wblk, kb + 256, 1);
      WSTAGE(1, VM0, aoff, Ablk, Wblk, kb + 384, 1);
    }
    {
      // last two stages: the DMA crosses into the next tile (without one it re-fetches this tile's first stages into
      // buffers that nobody reads again -- one code path, see gemm_persist.hip)
      WSTAGE(0, VM_FIRST, naoff, Anext, Wnext, 0, 1);
      WSTAGE(1, VM0, naoff, Anext, Wnext, 128, 0);
    }
    // the MFMAs are asm statements: hipcc pads nothing between the last one and the epilogue's accumulator reads
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (p.dbg && tid == 0) p.dbg[(size_t)idx * 8 + 2] = __builtin_amdgcn_s_memrealtime();

    // ------------------------------- epilogue of tile `cur` -------------------------------
    // acc[mt][nt][e]: row mw0 + 16*mt + frow, column nw0 + 16*nt + 4*qd + e
    const int mw0 = cur.m0 + wr * 128;
    const int nw0 = cur.n0 + wc * 128;
    const size_t gcol = (size_t)nw0 + rchunk * 8;

    if constexpr (EPI == EPI_LNFOLD) {
      // (mean, rstd) of the tile's rows from the raw partial sums that the DMA left in AUX[buf]
      char* raw = smem + AUX_OFF + (tile_iter & 1) * 8192;
      {
        float s = 0.f, ss = 0.f;
        for (int part = 0; part < p.stats_in_parts; ++part) {
          const float2 t = *(const float2*)(raw + part * 2048 + tid * 8);
          s += t.x; ss += t.y;
        }
        const float mean = s * p.inv_width;
        const float var = fmaxf(ss * p.inv_width - mean * mean, 0.f);
        *(float2*)(raw + tid * 8) = float2{mean, rsqrtf(var + p.eps)};
      }
      __syncthreads();
    }

    f32x4_t cs[8], bs[8];
    if constexpr (EPI == EPI_LNFOLD) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) cs[nt] = *(const f32x4_t*)(p.colsum + nw0 + nt * 16 + qd * 4);
    }
    if (EPI != EPI_STORE_BF16 || p.bias) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) bs[nt] = *(const f32x4_t*)(p.bias + nw0 + nt * 16 + qd * 4);
    } else {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) bs[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // residual rows: 4 x 16 B per lane and 16-row block, two blocks in flight
    uint4 rres[8];
#define LOAD_RES(mt_, k)                                                                      \
  do {                                                                                        \
    const int m_ = mw0 + (mt_) * 16 + rrow + 4 * (k);                                         \
    rres[((mt_) & 1) * 4 + (k)] = uint4{0, 0, 0, 0};                                          \
    if (m_ < p.M) rres[((mt_) & 1) * 4 + (k)] = *(const uint4*)((const bf16_t*)p.resid + (size_t)m_ * p.ldo + gcol); \
  } while (0)
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { LOAD_RES(0, k); LOAD_RES(1, k); }
    }

#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      uint2 pk[8];
      if constexpr (EPI == EPI_STORE_BF16) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          const f32x4_t v = acc[mt][nt] + bs[nt];
          pk[nt] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
      } else if constexpr (EPI == EPI_LNFOLD) {
        const float2 t = *(const float2*)(smem + AUX_OFF + (tile_iter & 1) * 8192 + (wr * 128 + mt * 16 + frow) * 8);
        const float mean = t.x, rstd = t.y;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          f32x4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = act_apply_t<ACT>(rstd * (acc[mt][nt][e] - mean * cs[nt][e]) + bs[nt][e]);
          pk[nt] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
      } else {
        // residual rows of this 16-row block: row-major image -> fragment layout
#pragma unroll
        for (int k = 0; k < 4; ++k) *(uint4*)TR_ADDR(k) = rres[(mt & 1) * 4 + k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // cross-lane hand-off through the image
        if (mt + 2 < 8) {
#pragma unroll
          for (int k = 0; k < 4; ++k) LOAD_RES(mt + 2, k);
        }
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
          const uint2 rr = *(const uint2*)TW_ADDR(nt);
          f32x4_t v = acc[mt][nt] + bs[nt];
          v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
          v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
          pk[nt] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          const float r0 = __uint_as_float(pk[nt].x << 16), r1 = __uint_as_float(pk[nt].x & 0xffff0000u);
          const float r2 = __uint_as_float(pk[nt].y << 16), r3 = __uint_as_float(pk[nt].y & 0xffff0000u);
          s += (r0 + r1) + (r2 + r3);
          ss += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
        }
        if (mw0 + mt * 16 + frow >= p.M) { s = 0.f; ss = 0.f; }
        s += __shfl_xor(s, 16); ss += __shfl_xor(ss, 16);
        s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
        if (lane < 16)
          *(float2*)(smem + AUX_OFF + ((size_t)wc * 256 + wr * 128 + mt * 16 + lane) * 8) = float2{s, ss};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // fragment reads done before the image is rewritten
      }
      // fragment layout -> row-major image -> four 16-B-per-lane stores of 4 full 256-B rows each
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) *(uint2*)TW_ADDR(nt) = pk[nt];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint4 v = *(const uint4*)TR_ADDR(k);
        const int m = mw0 + mt * 16 + rrow + 4 * k;
        if (m < p.M) *(uint4*)((bf16_t*)p.out + (size_t)m * p.ldo + gcol) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // image reads done before the next block's writes
    }
#undef LOAD_RES

    if constexpr (EPI == EPI_RESID) {
      __syncthreads();
      if (cur.m0 + tid < p.M) {
        const float2 t0 = *(const float2*)(smem + AUX_OFF + (size_t)tid * 8);
        const float2 t1 = *(const float2*)(smem + AUX_OFF + ((size_t)256 + tid) * 8);
        *(float2*)(p.stats_out + ((size_t)cur.tn * p.stats_ld + cur.m0 + tid) * 2) = float2{t0.x + t1.x, t0.y + t1.y};
      }
      __syncthreads();                       // AUX is rewritten by the next tile's epilogue
    }

    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 3] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 0] = blockIdx.x; }
    if (!has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the redundant DMA before the LDS is released
      break;
    }
    // all 256 rows valid: every guarded store above was issued (32 row stores per wave, 33 with EPI_RESID statistics)
    relax = (cur.m0 + BM <= p.M && p.dbg == nullptr) ? (EPI == EPI_RESID ? 2 : 1) : 0;
    idx = nidx; cur = nxt; Ablk = Anext; Wblk = Wnext;
#pragma unroll
    for (int q = 0; q < 8; ++q) aoff[q] = naoff[q];
    ++tile_iter;
  }
}

template <int EPI, int ACT>
hipError_t launch_wide(const GemmParams& p, hipStream_t stream) {
  static int n_cu = 0;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_wide_kernel<EPI, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    int dev = 0;
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    n_cu = prop.multiProcessorCount;
    attr_set = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  int grid = n_cu > 0 ? n_cu : 256;
  grid -= grid % 8;                           // keep tile index mod 8 == workgroup index mod 8 (XCD affinity)
  if (grid < 8) grid = 8;
  if (tiles < grid) grid = tiles;
  hipLaunchKernelGGL((gemm_wide_kernel<EPI, ACT>), dim3(grid), dim3(256), LDS_BYTES, stream, p);
  return hipGetLastError();
}

}  // namespace

hipError_t ce_gemm_nt_wide(const GemmParams& p, int epi, hipStream_t stream) {
  switch (epi) {
    case EPI_STORE_BF16: return launch_wide<EPI_STORE_BF16, -1>(p, stream);
    case EPI_LNFOLD:     // the activation is a template parameter: a run-time switch made hipcc evaluate both GELUs per element
      if (p.act == CE_ACT_QUICK_GELU) return launch_wide<EPI_LNFOLD, CE_ACT_QUICK_GELU>(p, stream);
      if (p.act == CE_ACT_GELU_ERF) return launch_wide<EPI_LNFOLD, CE_ACT_GELU_ERF>(p, stream);
      return launch_wide<EPI_LNFOLD, -1>(p, stream);
    case EPI_RESID: return launch_wide<EPI_RESID, -1>(p, stream);
    default: return hipErrorInvalidValue;
  }
}
