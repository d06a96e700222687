// bf16 "NT" GEMM, wide-wave variant of gemm_persist.hip: the same persistent 256x256 tiles, the same 4-slot K=32 LDS
// ring fed by LDS-DMA across tile boundaries and the same three epilogues, but FOUR waves per workgroup (one per SIMD,
// 2 x 2), each owning a 128 x 128 corner of the tile with its 256 accumulator registers in the upper half of the 512-entry
// register file.  Against the eight-wave kernel (128 x 64 per wave) that is 1/3 less LDS fragment traffic per MFMA and
// half the barrier participants.  One wave per SIMD has nobody to hide behind, so the stage is software-pipelined in
// registers with single-buffered fragments and a half-stage skew:
//   pass 0 of stage s:  32 MFMAs (A rows 0..7 x W cols 0..3) | reads fb[4..7](s) | the wave's 8 DMA pieces of stage s+3
//   wait for the wave's pieces of stage s+1, barrier
//   pass 1 of stage s:  32 MFMAs (A rows 0..7 x W cols 4..7) | fa[i](s+1) after row i, fb[0..3](s+1)
// A ring slot is read during pass 1 of the stage before and pass 0 of its own stage, so the slot refilled in pass 0 of
// stage s (stage s+3 -> slot (s-1)&3) was released by the barrier of stage s-1.
#include <stdlib.h>

#include "common.h"
#include "gemm.h"

namespace {

constexpr int BM = 256, BN = 256;
constexpr int STG = 32768, WPART = 16384;
constexpr int RING = 4 * STG;               // 131072
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

// One LDS-DMA piece: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset) to 1 KiB of LDS at lds_off.
// Inline asm on purpose: behind the builtin LLVM books every LDS-DMA as a FLAT access pending on BOTH counters and, as
// this kernel's vmcnt waits are hand-placed, never sees it retire -- every later LDS dependency then becomes
// `s_waitcnt lgkmcnt(0)` instead of a counted wait, which serialises the register-pipelined fragment reads.
// (m0 is written; nothing else in this file uses it.)
__device__ __forceinline__ void glds16(const char* base, unsigned off, char* smem, int lds_off) {
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(lds_off));
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
  const int kend = p.K * 2;                  // bytes along K; one stage = 64 B; K % 128 == 0 (stages % 4 == 0)

  // LDS-DMA: wave w fills the 1-KiB subtiles 4w..4w+3 (16 rows x 64 B) of both operands; LDS chunk lane&3 of row lane>>2
  // holds logical 16-B chunk (lane&3) ^ (2*(row>>3))
  const int lchunk16 = ((lane & 3) ^ (((lane >> 5) & 1) << 1)) * 16;
  const int srow = 64 * w + (lane >> 2);                 // + 16*j
  const int dma_lds = w * 4096;                          // + slot*STG (+ WPART) + j*1024
  const int rd = frow * 64 + (((lane >> 4) ^ ((frow >> 3) << 1)) << 4);
  const int a_rd = wr * 8 * 1024 + rd;                   // + slot*STG + i*1024
  const int w_rd = WPART + wc * 8 * 1024 + rd;           // + slot*STG + j*1024
  unsigned woff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) woff[j] = (unsigned)((srow + 16 * j) * ldw_b) + lchunk16;

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
  unsigned aoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) aoff[j] = (unsigned)((min(cur.m0 + srow + 16 * j, p.M - 1) - cur.m0) * lda_b) + lchunk16;

#define PIECE_A(slot, blk, o, kbyte, j) glds16((blk) + (kbyte), (o)[j], smem, (slot) * STG + dma_lds + (j) * 1024)
#define PIECE_W(slot, blk, kbyte, j) glds16((blk) + (kbyte), woff[j], smem, (slot) * STG + WPART + dma_lds + (j) * 1024)
#define STAGE_ALL(slot, ab, ao, wb, kbyte)                                                  \
  do {                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) { PIECE_A(slot, ab, ao, kbyte, j); PIECE_W(slot, wb, kbyte, j); } \
  } while (0)
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
#define RD_A(slot, i) fa[i] = *(const frag_t*)(smem + (slot) * STG + a_rd + (i) * 1024)
#define RD_W(slot, j) fb[j] = *(const frag_t*)(smem + (slot) * STG + w_rd + (j) * 1024)
#define BARRIER() asm volatile("s_barrier" ::: "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MMA_ROW(i, j0)                                                                      \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
    acc[i][(j0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[(j0) + j], fa[i], acc[i][(j0) + j], 0, 0, 0)
  // stage on ring slot S; NS = slot of the next stage; the 8 DMA pieces of stage +3 go to slot (S+3)&3: piece index 0..3
  // A, 4..7 W, written as the statement list DMA(q).  PREF = 0: last stage of a tile.
#define PSTAGE(S, NS, DMA, PREF)                                                            \
  do {                                                                                      \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                         \
      MMA_ROW(i, 0);                                                                        \
      SB();                                                                                 \
      if (i < 4) RD_W(S, 4 + i);                                                            \
      DMA(i);                                                                               \
      SB();                                                                                 \
    }                                                                                       \
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   /* stages +2, +3 may be in flight: stage +1 has landed */ \
    BARRIER();                                                                              \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                         \
      MMA_ROW(i, 4);                                                                        \
      SB();                                                                                 \
      if (PREF) RD_A(NS, i);                                                                \
      if (PREF && i < 4) RD_W(NS, i);                                                       \
      SB();                                                                                 \
    }                                                                                       \
  } while (0)

  // ---- cold prologue of the first tile: stages 0, 1, 2 ----
  int tile_iter = 0;
  STAGE_STATS(0, cur.m0);
  STAGE_ALL(0, Ablk, aoff, Wblk, 0);
  STAGE_ALL(1, Ablk, aoff, Wblk, 64);
  STAGE_ALL(2, Ablk, aoff, Wblk, 128);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  BARRIER();

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
    unsigned naoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) naoff[j] = aoff[j];
    if (has_next) {
      nxt = decode_tile(nidx, tiles_m, tiles_n);
      Anext = (const char*)p.A + (size_t)nxt.m0 * lda_b;
      Wnext = (const char*)p.W + (size_t)nxt.n0 * ldw_b;
#pragma unroll
      for (int j = 0; j < 4; ++j) naoff[j] = (unsigned)((min(nxt.m0 + srow + 16 * j, p.M - 1) - nxt.m0) * lda_b) + lchunk16;
      STAGE_STATS((tile_iter + 1) & 1, nxt.m0);
    }
    // fragments of the tile's first stage (slot 0 landed at least one barrier ago); fb[4..7] follow in its pass 0
#pragma unroll
    for (int j = 0; j < 4; ++j) RD_W(0, j);
#pragma unroll
    for (int i = 0; i < 8; ++i) RD_A(0, i);
    if (p.dbg && tid == 0) p.dbg[(size_t)idx * 8 + 1] = __builtin_amdgcn_s_memrealtime();

#define DMA_CUR(slot, kbyte) if (q < 4) PIECE_A(slot, Ablk, aoff, kbyte, q & 3); else PIECE_W(slot, Wblk, kbyte, q & 3)
#define DMA_NXT(slot, kbyte) if (q < 4) PIECE_A(slot, Anext, naoff, kbyte, q & 3); else PIECE_W(slot, Wnext, kbyte, q & 3)
#define D0(q_) { const int q = q_; DMA_CUR(3, kb + 192); }
#define D1(q_) { const int q = q_; DMA_CUR(0, kb + 256); }
#define D2(q_) { const int q = q_; DMA_CUR(1, kb + 320); }
#define D3(q_) { const int q = q_; DMA_CUR(2, kb + 384); }
#define E1(q_) { const int q = q_; DMA_NXT(0, 0); }
#define E2(q_) { const int q = q_; DMA_NXT(1, 64); }
#define E3(q_) { const int q = q_; DMA_NXT(2, 128); }
    for (int kb = 0; kb < kend - 256; kb += 256) {
      PSTAGE(0, 1, D0, 1);
      PSTAGE(1, 2, D1, 1);
      PSTAGE(2, 3, D2, 1);
      PSTAGE(3, 0, D3, 1);
    }
    {
      // last four stages: the DMA crosses into the next tile (without one it re-fetches this tile's first stages into
      // slots that nobody reads again -- one code path, see gemm_persist.hip)
      const int kb = kend - 256;
      PSTAGE(0, 1, D0, 1);
      PSTAGE(1, 2, E1, 1);
      PSTAGE(2, 3, E2, 1);
      PSTAGE(3, 0, E3, 0);
    }
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
    idx = nidx; cur = nxt; Ablk = Anext; Wblk = Wnext;
#pragma unroll
    for (int j = 0; j < 4; ++j) aoff[j] = naoff[j];
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
