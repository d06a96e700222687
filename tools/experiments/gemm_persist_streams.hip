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
constexpr int SLOT = 16384;                 // one operand's K=32 stage: 256 rows x 64 B
constexpr int A_SLOTS = 5, W_SLOTS = 3;
constexpr int RING = (A_SLOTS + W_SLOTS) * SLOT;   // 131072
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
// Issued as inline asm on purpose: behind the builtin LLVM books every LDS-DMA as a FLAT access that is pending on BOTH
// counters and, since this kernel's vmcnt waits are hand-placed, never sees it retire -- every later LDS dependency then
// becomes `s_waitcnt lgkmcnt(0)` instead of a counted wait, which serialises the register-pipelined fragment reads.
__device__ __forceinline__ void glds16(const char* base, unsigned off, char* smem, int lds_off) {
  const unsigned lds_addr = (unsigned)(size_t)LDS_PTR(lds_off);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_addr) : "memory");   // (m0 is written; nothing else in this file uses it)
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

  // Operand streams.  LDS ring = 5 A slots + 3 W slots of 16 KiB (256 rows x 64 B, sixteen 1-KiB subtiles of 16 rows,
  // 16-B chunk index XOR 2*(row>>3)).  Waves 0-3 feed the A ring, waves 4-7 the W ring: each wave owns subtiles
  // 4*(w&3)..+3 of its operand, one LDS-DMA piece each per stage, and runs AHEAD of the MFMAs by 5 (A, streamed from HBM)
  // or 3 (W, L2-resident) stages, across tile boundaries.  Per-wave in-order vmcnt then counts one stream only, so the
  // long A latency is covered by 3-4 stage times instead of 2 (both operands through one wave, one ring depth).
  const bool is_a = w < 4;
  const int ring_off = is_a ? 0 : A_SLOTS * SLOT;
  const int n_slots = is_a ? A_SLOTS : W_SLOTS;
  const char* const s_mat = is_a ? (const char*)p.A : (const char*)p.W;
  const size_t s_ld = is_a ? lda_b : ldw_b;
  const int s_rows = is_a ? p.M : p.N;                       // rows past the end are clamped (ragged last M tile)
  const int lchunk16 = ((lane & 3) ^ (((lane >> 5) & 1) << 1)) * 16;
  const int srow = 64 * (w & 3) + (lane >> 2);               // + 16*j: row of piece j within the 256-row block
  const int dma_lds = ring_off + (w & 3) * 4096;             // + slot*SLOT + j*1024
  const int rd = frow * 64 + (((lane >> 4) ^ ((frow >> 3) << 1)) << 4);
  const int a_rd = wr * 8 * 1024 + rd;                       // + a_slot*SLOT + i*1024
  const int w_rd = A_SLOTS * SLOT + wc * 4 * 1024 + rd;      // + w_slot*SLOT + j*1024

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

  // raw row statistics of a tile's 256 rows: parts x 2 KiB, fetched by waves 0 and 1 (EPI_LNFOLD).  Always issued BEFORE
  // the stage's stream pieces: the counted waits below leave the newest N operations outstanding, and those must be
  // stream pieces only.
#define STAGE_STATS(buf, m0v)                                                               \
  do {                                                                                      \
    if (EPI == EPI_LNFOLD && w < 2) {                                                        \
      for (int part = 0; part < p.stats_in_parts; ++part)                                    \
        glds16((const char*)p.stats_in + ((size_t)part * p.stats_ld + (m0v)) * 8, (unsigned)((w * 64 + lane) * 16), smem, \
               AUX_OFF + (buf) * 8192 + part * 2048 + w * 1024);                             \
    }                                                                                       \
  } while (0)

  // ---- this wave's stream cursor: (tile, byte offset along K, ring slot) of the next stage to fetch ----
  int s_idx = idx;                                           // tile the stream is in
  int s_k = 0;
  int s_slot = 0;
  const char* s_blk;
  unsigned s_off[4];
#define STREAM_TILE(tid_)                                                                   \
  do {                                                                                      \
    const TileId t_ = decode_tile(tid_, tiles_m, tiles_n);                                  \
    const int r0_ = is_a ? t_.m0 : t_.n0;                                                   \
    s_blk = s_mat + (size_t)r0_ * s_ld;                                                     \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
      s_off[j] = (unsigned)((min(r0_ + srow + 16 * j, s_rows - 1) - r0_) * s_ld) + lchunk16; \
  } while (0)
  // one piece of the stream's next stage; after the fourth the cursor moves on (next K step, or the next tile of this
  // workgroup; past the last tile it re-fetches the last tile's stages into slots that nobody reads)
#define STREAM_PIECE(j) glds16(s_blk + s_k, s_off[j], smem, dma_lds + s_slot * SLOT + (j) * 1024)
#define STREAM_ADVANCE()                                                                    \
  do {                                                                                      \
    s_slot = (s_slot + 1 == n_slots) ? 0 : s_slot + 1;                                       \
    s_k += 64;                                                                              \
    if (s_k == kend) {                                                                      \
      s_k = 0;                                                                              \
      if (s_idx + G < nwg) { s_idx += G; STREAM_TILE(s_idx); }                              \
    }                                                                                       \
  } while (0)
#define STREAM_WAIT()                                                                       \
  do {                                                                                      \
    if (is_a) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  /* A: stages +3..+5 may be in flight */ \
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");        /* W: stage +3 */             \
  } while (0)

#define RD_W(buf, off, j) fb[buf][j] = *(const frag_t*)(smem + (off) + w_rd + (j) * 1024)
#define RD_A(off, i) fa[i] = *(const frag_t*)(smem + (off) + a_rd + (i) * 1024)
#define BARRIER() asm volatile("s_barrier" ::: "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)
  // One K=32 stage, software-pipelined in registers: its fragments (fa, fb[B]) were read during the previous stage; while
  // its 32 MFMAs issue, the next stage's fragments replace each fa[i] right after its last use and fill the other fb
  // buffer, and the wave's four stream pieces go out after the first MFMA rows.  One counted wait + one barrier per
  // stage: on leaving, every wave's pieces of stage +2 have landed and every wave is done with this stage's slots.
  // PREF = 0: last stage of a tile (its successor's fragments are read at the top of the next tile instead).
#define PSTAGE(B, PREF)                                                                     \
  do {                                                                                      \
    const int ao_ = a_rs * SLOT, wo_ = w_rs * SLOT;                                         \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                         \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                         \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[B][j], fa[i], acc[i][j], 0, 0, 0); \
      SB();                                                                                 \
      if (PREF) RD_A(ao_, i);                                                               \
      if (PREF && i < 4) RD_W((B) ^ 1, wo_, i);                                             \
      if (i < 4) STREAM_PIECE(i);                                                           \
      SB();                                                                                 \
    }                                                                                       \
    STREAM_ADVANCE();                                                                       \
    if (PREF) { a_rs = (a_rs + 1 == A_SLOTS) ? 0 : a_rs + 1; w_rs = (w_rs + 1 == W_SLOTS) ? 0 : w_rs + 1; } \
    STREAM_WAIT();                                                                          \
    BARRIER();                                                                              \
  } while (0)

  // ---- cold prologue: the stream's first 5 (A) or 3 (W) stages ----
  int tile_iter = 0;
  STAGE_STATS(0, cur.m0);
  STREAM_TILE(s_idx);
  for (int st = 0; st < n_slots; ++st) {
#pragma unroll
    for (int j = 0; j < 4; ++j) STREAM_PIECE(j);
    STREAM_ADVANCE();
  }
  STREAM_WAIT();                                           // stages 0 and 1 have landed
  BARRIER();
  int a_rs = 0, w_rs = 0;                                  // ring slots of the next stage whose fragments get read

  for (;;) {
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    frag_t fa[8], fb[2][4];
    const int nidx = idx + G;
    const bool has_next = nidx < nwg;
    TileId nxt = cur;
    if (has_next) {
      nxt = decode_tile(nidx, tiles_m, tiles_n);
      STAGE_STATS((tile_iter + 1) & 1, nxt.m0);
    }
    // the fragments of the tile's first stage (landed one barrier ago)
    {
      const int ao_ = a_rs * SLOT, wo_ = w_rs * SLOT;
#pragma unroll
      for (int j = 0; j < 4; ++j) RD_W(0, wo_, j);
#pragma unroll
      for (int i = 0; i < 8; ++i) RD_A(ao_, i);
      a_rs = (a_rs + 1 == A_SLOTS) ? 0 : a_rs + 1; w_rs = (w_rs + 1 == W_SLOTS) ? 0 : w_rs + 1;
      // The first stage's stream pieces overwrite exactly these slots (the rings have no spare slot), and after an
      // epilogue the waves are no longer in step: nobody may issue them before everybody's reads have returned.
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      BARRIER();
    }
    if (p.dbg && tid == 0) p.dbg[(size_t)idx * 8 + 1] = __builtin_amdgcn_s_memrealtime();

    for (int kb = 0; kb < kend - 128; kb += 128) {
      PSTAGE(0, 1);
      PSTAGE(1, 1);
    }
    PSTAGE(0, 1);
    PSTAGE(1, 0);
    if (p.dbg && tid == 0) p.dbg[(size_t)idx * 8 + 2] = __builtin_amdgcn_s_memrealtime();

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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // cross-lane hand-off through the image
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // fragment reads done before the image is rewritten
      }
      // fragment layout -> row-major image -> two 16-B-per-lane stores of 8 full rows each
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) *(uint2*)TW_ADDR(nt) = pk[nt];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint4 v0 = *(const uint4*)(tr + tr_base);
      const uint4 v1 = *(const uint4*)(tr + 1024 + tr_base);
      const int ma = mw0 + mt * 16 + row_l, mb = ma + 8;
      if (ma < p.M) *(uint4*)((bf16_t*)p.out + (size_t)ma * p.ldo + gcol) = v0;
      if (mb < p.M) *(uint4*)((bf16_t*)p.out + (size_t)mb * p.ldo + gcol) = v1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // image reads done before the next block's writes
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
    idx = nidx; cur = nxt;
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
