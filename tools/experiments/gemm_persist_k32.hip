// Persistent variant of the bf16 "NT" GEMM (see gemm_bf16.hip for the tile, LDS image and phase
// schedule; this file only changes WHO runs the tiles): one workgroup per CU walks a strided list of
// 256x256 tiles and keeps the 4-slot K=32 stage ring running ACROSS tile boundaries.  During the last
// stages of a tile the LDS-DMA already fetches stages 0,1,2 of the next tile, so a tile costs main loop +
// epilogue only: the per-workgroup launch gap (~3 us) and the cold prologue (~1.8 us) of the
// one-tile-per-workgroup kernel (31 us per K=1024 tile, measured with in-kernel stamps) disappear.
//
// LDS (160 KiB): [0,128K) stage ring | [128K,144K) AUX: EPI_LNFOLD raw row statistics, 2 buffers x
// [4 parts][256 rows][sum,sumsq] landed by LDS-DMA one tile ahead; EPI_RESID per-wave row partial sums |
// [144K,160K) 8 wave-private 2 KiB images used to turn the MFMA fragment layout into whole 16-B row
// chunks (and the residual the other way) one 16-row block at a time.
#include <stdlib.h>

#include "common.h"
#include "gemm.h"

namespace {

constexpr int BM = 256, BN = 256;
constexpr int STG = 32768, WPART = 16384;
constexpr int RING = 4 * STG;               // 131072
constexpr int AUX_OFF = RING;               // 16 KiB
constexpr int TR_OFF = RING + 16384;        // 8 x 2 KiB
constexpr int LDS_BYTES = RING + 32768;     // 163840

#define LDS_PTR(off) ((__attribute__((address_space(3))) void*)(smem + (off)))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int ACT>
__device__ __forceinline__ float act_apply_t(float u) {
  // compile-time activation: a run-time `act` makes hipcc evaluate BOTH activations per element and select
  if constexpr (ACT == CE_ACT_QUICK_GELU) return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * u));
  else if constexpr (ACT == CE_ACT_GELU_ERF) return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
  else return u;
}


// uniform base (SGPR pair) + per-lane 32-bit unsigned offset: selects the saddr form of the DMA, no 64-bit VGPR address
__device__ __forceinline__ void glds16(const char* base, unsigned off, char* smem, int lds_off) {
  __builtin_amdgcn_global_load_lds(GLOBAL_PTR(base + off), LDS_PTR(lds_off), 16, 0, 0);
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
__global__ __launch_bounds__(512, 2) void gemm_persist_kernel(const GemmParams p) {
  typedef bf16x8_t frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int frow = lane & 15;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  const int nwg = tiles_m * tiles_n;
  const int G = gridDim.x;
  const size_t lda_b = (size_t)p.lda * 2, ldw_b = (size_t)p.ldw * 2;
  const int kend = p.K * 2;                  // bytes along K; one stage = 64 B; K % 128 == 0

  // LDS-DMA lane mapping (see gemm_bf16.hip IMPL 2)
  const int lrow = 16 * w + (lane >> 2);
  const int lchunk16 = ((lane & 3) ^ (((lane >> 5) & 1) << 1)) * 16;
  const int dma_lds = w * 1024;
  const int rd = frow * 64 + (((lane >> 4) ^ ((frow >> 3) << 1)) << 4);
  const int a_rd = wr * 8 * 1024 + rd;
  const int w_rd = WPART + wc * 4 * 1024 + rd;
  const unsigned woff0 = (unsigned)(lrow * ldw_b) + lchunk16, woff1 = (unsigned)((128 + lrow) * ldw_b) + lchunk16;

  // epilogue lane mapping: 16-row x 128-B image per wave, 16-B chunk index XOR row&7
  char* tr = smem + TR_OFF + w * 2048;
  const int qd = lane >> 4;
  const int tw_base = frow * 128 + (qd & 1) * 8;
  const int tw_sw = frow & 7;
  const int tr_base = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);   // + 1024 for rows 8..15
  const int row_l = lane >> 3;
#define TW_ADDR(nt) (tr + tw_base + ((((nt) * 2 + (qd >> 1)) ^ tw_sw) << 4))

  int idx = blockIdx.x;
  TileId cur = decode_tile(idx, tiles_m, tiles_n);
  const char* Ablk = (const char*)p.A + (size_t)cur.m0 * lda_b;
  const char* Wblk = (const char*)p.W + (size_t)cur.n0 * ldw_b;
  unsigned aoff0 = (unsigned)((min(cur.m0 + lrow, p.M - 1) - cur.m0) * lda_b) + lchunk16;
  unsigned aoff1 = (unsigned)((min(cur.m0 + 128 + lrow, p.M - 1) - cur.m0) * lda_b) + lchunk16;

#define STAGE_A(slot, blk, o0, o1, kbyte)                                                   \
  do {                                                                                      \
    glds16((blk) + (kbyte), (o0), smem, (slot) * STG + dma_lds);                            \
    glds16((blk) + (kbyte), (o1), smem, (slot) * STG + 8192 + dma_lds);                     \
  } while (0)
#define STAGE_W(slot, blk, kbyte)                                                           \
  do {                                                                                      \
    glds16((blk) + (kbyte), woff0, smem, (slot) * STG + WPART + dma_lds);                   \
    glds16((blk) + (kbyte), woff1, smem, (slot) * STG + WPART + 8192 + dma_lds);            \
  } while (0)
  // raw row statistics of a tile's 256 rows: parts x 2 KiB, fetched by waves 0 and 1 (EPI_LNFOLD)
#define STAGE_STATS(buf, m0v)                                                               \
  do {                                                                                      \
    if (EPI == EPI_LNFOLD && w < 2) {                                                        \
      for (int part = 0; part < p.stats_in_parts; ++part)                                    \
        glds16((const char*)p.stats_in + ((size_t)part * p.stats_ld + (m0v)) * 8, (unsigned)((w * 64 + lane) * 16), smem, \
               AUX_OFF + (buf) * 8192 + part * 2048 + w * 1024);                             \
    }                                                                                       \
  } while (0)
#define LD_W(slot) _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[j] = *(const frag_t*)(smem + (slot) * STG + w_rd + j * 1024);
#define LD_A(slot, half) _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[i] = *(const frag_t*)(smem + (slot) * STG + a_rd + ((half) * 4 + i) * 1024);
#define MMA(half)                                                                           \
  do {                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
      acc[(half) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[(half) * 4 + i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                          \
  } while (0)
#define BARRIER() asm volatile("s_barrier" ::: "memory")
#define WAIT_LDS()                                                                          \
  do {                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
  // one K=32 stage = two phases; ISSUE_W / ISSUE_A are the DMA statements of the phases, VM the counted wait
#define STAGE(slot, ISSUE_W, ISSUE_A, WAIT_VM)                                              \
  do {                                                                                      \
    LD_W(slot) __builtin_amdgcn_sched_barrier(0); LD_A(slot, 0)                             \
    ISSUE_W;                                                                                \
    BARRIER(); WAIT_LDS(); MMA(0); BARRIER();                                               \
    LD_A(slot, 1)                                                                           \
    ISSUE_A;                                                                                \
    WAIT_VM;                                                                                \
    BARRIER(); WAIT_LDS(); MMA(1); BARRIER();                                               \
  } while (0)
  // Both parts of stage t+3 are issued in stage t (W in phase a, A in phase b); the counted wait leaves stages t+2 and t+3
  // in flight (4 x 2 pieces) and retires stage t+1.
#define VM8 asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
  // First two stages after an epilogue: the pieces they need (stages 1 and 2 of the new tile) were issued BEFORE the
  // epilogue's stores, and vmcnt retires in order -- so when every wave issued exactly its 16 row stores (17 for the
  // waves that also write EPI_RESID statistics) the wait may leave those outstanding as well, and the stores get two
  // and a half stage times to drain instead of stalling the ring (1.9 us per tile, tools/gemm_stamps.py).
#define VM_AFTER_EPILOGUE                                                                   \
  do {                                                                                      \
    /* one opaque instruction for the compiler (a real branch here splits every stage into basic blocks and costs   */ \
    /* ~20 spilled VGPRs): sel 0 -> vmcnt(8), 1 -> vmcnt(24), 2 -> vmcnt(25); only in the first two stages of a tile */ \
    const int sel_ = __builtin_amdgcn_readfirstlane(relax > 0 ? relax_sel : 0);                                          \
    relax = relax > 0 ? relax - 1 : 0;                                                      \
    asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lvm8_%=\n\ts_cmp_eq_u32 %0, 1\n\ts_cbranch_scc1 .Lvm24_%=\n\t"          \
                 "s_waitcnt vmcnt(25)\n\ts_branch .Lvmend_%=\n.Lvm24_%=:\n\ts_waitcnt vmcnt(24)\n\ts_branch .Lvmend_%=\n"       \
                 ".Lvm8_%=:\n\ts_waitcnt vmcnt(8)\n.Lvmend_%=:" : : "s"(sel_) : "memory", "scc");                                  \
  } while (0)

  // ---- cold prologue of the first tile ----
  int tile_iter = 0;
  STAGE_STATS(0, cur.m0);
  STAGE_A(0, Ablk, aoff0, aoff1, 0); STAGE_W(0, Wblk, 0);
  STAGE_A(1, Ablk, aoff0, aoff1, 64); STAGE_W(1, Wblk, 64);
  STAGE_A(2, Ablk, aoff0, aoff1, 128); STAGE_W(2, Wblk, 128);
  VM8;
  BARRIER();
  int relax = 0;                             // stages of the coming tile that may leave the previous tile's stores in flight
  const int relax_sel = (EPI == EPI_RESID && w < 4) ? 2 : 1;   // those waves also write the row statistics: 17 stores, not 16

  for (;;) {
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    frag_t fa[4], fb[4];

    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 1] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 4] = __builtin_amdgcn_s_memtime(); }
    if (wr == 1) BARRIER();                  // second wave row runs half a phase behind

    for (int kb = 0; kb < kend - 256; kb += 256) {
      STAGE(0, STAGE_W(3, Wblk, kb + 192), STAGE_A(3, Ablk, aoff0, aoff1, kb + 192), VM_AFTER_EPILOGUE);
      STAGE(1, STAGE_W(0, Wblk, kb + 256), STAGE_A(0, Ablk, aoff0, aoff1, kb + 256), VM_AFTER_EPILOGUE);
      STAGE(2, STAGE_W(1, Wblk, kb + 320), STAGE_A(1, Ablk, aoff0, aoff1, kb + 320), VM8);
      STAGE(3, STAGE_W(2, Wblk, kb + 384), STAGE_A(2, Ablk, aoff0, aoff1, kb + 384), VM8);
    }
    // ---- last four stages: the DMA crosses into the next tile ----
    const int nidx = idx + G;
    const bool has_next = nidx < nwg;
    TileId nxt = cur;
    const char *Anext = Ablk, *Wnext = Wblk;
    unsigned naoff0 = aoff0, naoff1 = aoff1;
    if (has_next) {
      nxt = decode_tile(nidx, tiles_m, tiles_n);
      Anext = (const char*)p.A + (size_t)nxt.m0 * lda_b;
      Wnext = (const char*)p.W + (size_t)nxt.n0 * ldw_b;
      naoff0 = (unsigned)((min(nxt.m0 + lrow, p.M - 1) - nxt.m0) * lda_b) + lchunk16;
      naoff1 = (unsigned)((min(nxt.m0 + 128 + lrow, p.M - 1) - nxt.m0) * lda_b) + lchunk16;
    }
    {
      const int kb = kend - 256;
      // ONE code path: without a next tile the DMA harmlessly re-fetches this tile's first stages into dead
      // slots (two variants of this block made hipcc spill ~270 VGPRs)
      if (has_next) STAGE_STATS((tile_iter + 1) & 1, nxt.m0);
      STAGE(0, STAGE_W(3, Wblk, kb + 192), STAGE_A(3, Ablk, aoff0, aoff1, kb + 192), VM_AFTER_EPILOGUE);
      STAGE(1, STAGE_W(0, Wnext, 0), STAGE_A(0, Anext, naoff0, naoff1, 0), VM_AFTER_EPILOGUE);
      STAGE(2, STAGE_W(1, Wnext, 64), STAGE_A(1, Anext, naoff0, naoff1, 64), VM8);
      STAGE(3, STAGE_W(2, Wnext, 128), STAGE_A(2, Anext, naoff0, naoff1, 128), VM8);
    }
    if (wr == 0) BARRIER();                  // re-align the two wave rows for the epilogue
    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 2] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 5] = __builtin_amdgcn_s_memtime(); }

    // ------------------------------- epilogue of tile `cur` -------------------------------
    const int q4 = qd * 4;
    const int ncol0 = cur.n0 + wc * 64 + q4;         // + nt*16
    const int mw0 = cur.m0 + wr * 128;               // first row of the wave tile
    const size_t gcol = (size_t)cur.n0 + wc * 64 + (lane & 7) * 8;

    if constexpr (EPI == EPI_LNFOLD) {
      // (mean, rstd) of the tile's rows from the raw partial sums that the DMA left in AUX[buf]
      char* raw = smem + AUX_OFF + (tile_iter & 1) * 8192;
      if (tid < 256) {
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

    f32x4_t cs[4], bs[4];
    if constexpr (EPI == EPI_LNFOLD) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) cs[nt] = *(const f32x4_t*)(p.colsum + ncol0 + nt * 16);
    }
    if (EPI != EPI_STORE_BF16 || p.bias) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bs[nt] = *(const f32x4_t*)(p.bias + ncol0 + nt * 16);
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bs[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // residual rows, 8 x 16 B per lane in flight (four 16-row blocks ahead of their use)
    uint4 rres[8];
#define LOAD_RES(k)                                                                           \
  do {                                                                                        \
    const int m_ = mw0 + (k) * 8 + row_l;                                                     \
    rres[(k) & 7] = uint4{0, 0, 0, 0};                                                        \
    if (m_ < p.M) rres[(k) & 7] = *(const uint4*)((const bf16_t*)p.resid + (size_t)m_ * p.ldo + gcol); \
  } while (0)
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
      for (int k = 0; k < 8; ++k) LOAD_RES(k);
    }

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
        const float2 t = *(const float2*)(smem + AUX_OFF + (tile_iter & 1) * 8192 + (wr * 128 + mt * 16 + frow) * 8);
        const float mean = t.x, rstd = t.y;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          f32x4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = act_apply_t<ACT>(rstd * (acc[mt][nt][e] - mean * cs[nt][e]) + bs[nt][e]);
          pk[nt] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
      } else {
        // residual rows of this 16-row block: row-major image -> fragment layout
        *(uint4*)(tr + tr_base) = rres[(mt * 2) & 7];
        *(uint4*)(tr + 1024 + tr_base) = rres[(mt * 2 + 1) & 7];
        // (no wait: the LDS serves one wave's accesses in order, so the fragment-layout reads below see these writes)
        if (mt + 4 < 8) { LOAD_RES(mt * 2 + 8); LOAD_RES(mt * 2 + 9); }
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
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
        // (in order again: the image may be rewritten right behind the fragment reads)
      }
      // fragment layout -> row-major image -> two 16-B-per-lane stores of 8 full rows each
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) *(uint2*)TW_ADDR(nt) = pk[nt];
      // (no wait between the image writes and the row-major reads, nor before the next block's writes: one wave, in order)
      const uint4 v0 = *(const uint4*)(tr + tr_base);
      const uint4 v1 = *(const uint4*)(tr + 1024 + tr_base);
      const int ma = mw0 + mt * 16 + row_l, mb = ma + 8;
      if (ma < p.M) *(uint4*)((bf16_t*)p.out + (size_t)ma * p.ldo + gcol) = v0;
      if (mb < p.M) *(uint4*)((bf16_t*)p.out + (size_t)mb * p.ldo + gcol) = v1;
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

    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 3] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 0] = blockIdx.x; }
    if (!has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the redundant DMA before the LDS is released
      break;
    }
    relax = (cur.m0 + BM <= p.M && p.dbg == nullptr) ? 2 : 0;   // all 256 rows valid: every guarded store above was issued
    idx = nidx; cur = nxt; Ablk = Anext; Wblk = Wnext; aoff0 = naoff0; aoff1 = naoff1;
    ++tile_iter;
  }
}

template <int EPI, int ACT>
hipError_t launch_persist(const GemmParams& p, hipStream_t stream) {
  static int n_cu = 0;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_persist_kernel<EPI, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
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
  hipLaunchKernelGGL((gemm_persist_kernel<EPI, ACT>), dim3(grid), dim3(512), LDS_BYTES, stream, p);
  return hipGetLastError();
}

}  // namespace

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
