// bf16 "NT" GEMM, two-workgroups-per-CU variant of gemm_persist.hip for the epilogues that do not write row statistics
// (EPI_STORE_BF16, EPI_LNFOLD): persistent 256 x 128 tiles, four waves (2 x 2, 128 x 64 per wave as in the eight-wave
// kernel), 80 KiB of LDS per workgroup, so that TWO independent workgroups share a CU.  Per K = 1024 tile the
// eight-wave kernel spends 27 us in the main loop and 4.6 us in the epilogue and the store drain behind it, with the
// MFMA pipe idle (tools/gemm_stamps.py; without any epilogue the same kernel is 17-21 % faster on K = 1024 shapes);
// here the other workgroup's main loop runs underneath.  Price: 1.5x the L2 -> LDS operand traffic per MFMA.
//
// LDS: ring of 3 slots x (A 256 rows x 64 B | W 128 rows x 64 B) = 72 KiB + 8 KiB raw row statistics (EPI_LNFOLD).
// Stage g: issue the DMA of stage g+2 into the slot freed by the previous barrier, read the 12 fragments of stage g,
// 32 MFMAs, one counted wait (stage g+1 landed) + one barrier.  The ring runs across tiles; the epilogue's wave images
// live in the slot the last stage just released (the next tile's first DMA targets it: one barrier ends the epilogue).
#include <stdlib.h>

#include "common.h"
#include "gemm.h"

namespace {

constexpr int BM = 256, BN = 128;
constexpr int A_BYTES = 16384, W_BYTES = 8192, SLOT = A_BYTES + W_BYTES;   // 24 KiB
constexpr int RING = 3 * SLOT;              // 73728
constexpr int STATS_OFF = RING;             // 8 KiB: [4 parts][256 rows][sum, sumsq]
constexpr int LDS_BYTES = RING + 8192;      // 81920

#define LDS_PTR(off) ((__attribute__((address_space(3))) void*)(smem + (off)))

template <int ACT>
__device__ __forceinline__ float act_apply_t(float u) {
  if constexpr (ACT == CE_ACT_QUICK_GELU) return u * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.4554669595930156f * u));
  else if constexpr (ACT == CE_ACT_GELU_ERF) return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
  else return u;
}

// One LDS-DMA piece (inline asm: see gemm_fp8.hip for why the builtin is not used); m0 is written, nothing else here uses it
__device__ __forceinline__ void glds16(const char* base, unsigned off, char* smem, int lds_off) {
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(lds_off));
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_addr) : "memory");
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

template <int EPI, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_dual_kernel(const GemmParams p) {
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
  const int kend = p.K * 2;                  // bytes along K; one stage = 64 B; K % 128 == 0

  // DMA: wave w fills A subtiles 4w..4w+3 and W subtiles 2w, 2w+1 (1 KiB = 16 rows x 64 B, chunk c of row r at c ^ 2*(r>>3))
  const int lchunk16 = ((lane & 3) ^ (((lane >> 5) & 1) << 1)) * 16;
  const int arow = 64 * w + (lane >> 2);     // + 16*j
  const int wrow = 32 * w + (lane >> 2);     // + 16*j
  const int dma_a = w * 4096, dma_w = A_BYTES + w * 2048;
  const int rd = frow * 64 + (((lane >> 4) ^ ((frow >> 3) << 1)) << 4);
  const int a_rd = wr * 8 * 1024 + rd;                   // + slot + i*1024
  const int w_rd = A_BYTES + wc * 4 * 1024 + rd;         // + slot + j*1024
  unsigned woff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) woff[j] = (unsigned)((wrow + 16 * j) * ldw_b) + lchunk16;

  // epilogue lane mapping: 16-row x 128-B image per wave, 16-B chunk index XOR row&7
  const int tw_base = frow * 128 + (qd & 1) * 8;
  const int tw_sw = frow & 7;
  const int tr_base = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);   // + 1024 for rows 8..15
  const int row_l = lane >> 3;
#define TW_ADDR(nt) (tr + tw_base + ((((nt) * 2 + (qd >> 1)) ^ tw_sw) << 4))

  int idx = blockIdx.x;
  TileId cur = decode_tile(idx, tiles_m, tiles_n);

  // DMA cursor: (A block, W block, per-lane A offsets, byte offset along K) of the next stage to fetch
  const char* d_ablk = (const char*)p.A + (size_t)cur.m0 * lda_b;
  const char* d_wblk = (const char*)p.W + (size_t)cur.n0 * ldw_b;
  unsigned d_aoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) d_aoff[j] = (unsigned)((min(cur.m0 + arow + 16 * j, p.M - 1) - cur.m0) * lda_b) + lchunk16;
  int d_k = 0;
  // the tile the cursor moves to when it reaches the end of K (set at the top of every tile)
  const char *n_ablk = d_ablk, *n_wblk = d_wblk;
  unsigned n_aoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) n_aoff[j] = d_aoff[j];

#define ISSUE_STAGE(slot_off)                                                               \
  do {                                                                                      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) glds16(d_ablk + d_k, d_aoff[j], smem, (slot_off) + dma_a + j * 1024);   \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) glds16(d_wblk + d_k, woff[j], smem, (slot_off) + dma_w + j * 1024);     \
    d_k += 64;                                                                              \
    if (d_k == kend) {                                                                      \
      d_k = 0; d_ablk = n_ablk; d_wblk = n_wblk;                                            \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) d_aoff[j] = n_aoff[j];                  \
    }                                                                                       \
  } while (0)
  // raw row statistics of a tile's 256 rows: parts x 2 KiB, fetched by waves 0 and 1 (EPI_LNFOLD); issued before the
  // stage pieces that follow, so the counted waits never leave them behind
#define STAGE_STATS(m0v)                                                                    \
  do {                                                                                      \
    if (EPI == EPI_LNFOLD && w < 2) {                                                        \
      for (int part = 0; part < p.stats_in_parts; ++part)                                    \
        glds16((const char*)p.stats_in + ((size_t)part * p.stats_ld + (m0v)) * 8, (unsigned)((w * 64 + lane) * 16), smem, \
               STATS_OFF + part * 2048 + w * 1024);                                          \
    }                                                                                       \
  } while (0)
#define BARRIER() asm volatile("s_barrier" ::: "memory")

  // ---- cold prologue: stages 0 and 1 ----
  int s0 = 0, s1 = SLOT, s2 = 2 * SLOT;      // ring slot (byte offset) of the current stage, the next, and the DMA target
  STAGE_STATS(cur.m0);
  ISSUE_STAGE(s0);
  ISSUE_STAGE(s1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  BARRIER();

  const int nst = kend / 64;
  for (;;) {
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nidx = idx + G;
    const bool has_next = nidx < nwg;
    TileId nxt = cur;
    if (has_next) {
      nxt = decode_tile(nidx, tiles_m, tiles_n);
      n_ablk = (const char*)p.A + (size_t)nxt.m0 * lda_b;
      n_wblk = (const char*)p.W + (size_t)nxt.n0 * ldw_b;
#pragma unroll
      for (int j = 0; j < 4; ++j) n_aoff[j] = (unsigned)((min(nxt.m0 + arow + 16 * j, p.M - 1) - nxt.m0) * lda_b) + lchunk16;
    }
    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 1] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 4] = __builtin_amdgcn_s_memtime(); }

    for (int st = 0; st < nst; ++st) {
      ISSUE_STAGE(s2);                       // stage +2 (crosses into the next tile during the last two stages)
      frag_t fb[4], fa[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = *(const frag_t*)(smem + s0 + w_rd + j * 1024);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = *(const frag_t*)(smem + s0 + a_rd + i * 1024);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // stage +1 has landed (stage +2 may be in flight)
      BARRIER();
      const int t_ = s0; s0 = s1; s1 = s2; s2 = t_;
    }
    // pin the accumulators: keeps LLVM from sinking MFMAs out of the loop into the epilogue
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 2] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 5] = __builtin_amdgcn_s_memtime(); }

    // ------------------------------- epilogue of tile `cur` -------------------------------
    // scratch: the slot the last stage released (= s2, the target of the next stage's DMA)
    char* tr = smem + s2 + w * 2048;
    const int q4 = qd * 4;
    const int ncol0 = cur.n0 + wc * 64 + q4;         // + nt*16
    const int mw0 = cur.m0 + wr * 128;               // first row of the wave tile
    const size_t gcol = (size_t)cur.n0 + wc * 64 + (lane & 7) * 8;

    if constexpr (EPI == EPI_LNFOLD) {
      // (mean, rstd) of the tile's rows from the raw partial sums that the DMA left in STATS
      char* raw = smem + STATS_OFF;
      float s = 0.f, ss = 0.f;
      for (int part = 0; part < p.stats_in_parts; ++part) {
        const float2 t = *(const float2*)(raw + part * 2048 + tid * 8);
        s += t.x; ss += t.y;
      }
      const float mean = s * p.inv_width;
      const float var = fmaxf(ss * p.inv_width - mean * mean, 0.f);
      __syncthreads();                               // every thread has read its raw sums
      *(float2*)(raw + tid * 8) = float2{mean, rsqrtf(var + p.eps)};
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

#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      uint2 pk[4];
      if constexpr (EPI == EPI_STORE_BF16) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const f32x4_t v = acc[mt][nt] + bs[nt];
          pk[nt] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
      } else {
        const float2 t = *(const float2*)(smem + STATS_OFF + (wr * 128 + mt * 16 + frow) * 8);
        const float mean = t.x, rstd = t.y;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          f32x4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = act_apply_t<ACT>(rstd * (acc[mt][nt][e] - mean * cs[nt][e]) + bs[nt][e]);
          pk[nt] = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        }
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

    if (p.dbg && tid == 0) { p.dbg[(size_t)idx * 8 + 3] = __builtin_amdgcn_s_memrealtime(); p.dbg[(size_t)idx * 8 + 0] = blockIdx.x; }
    if (!has_next) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the redundant DMA before the LDS is released
      break;
    }
    __syncthreads();                         // the scratch slot and STATS are rewritten from here on
    STAGE_STATS(nxt.m0);
    idx = nidx; cur = nxt;
  }
}

template <int EPI, int ACT>
hipError_t launch_dual(const GemmParams& p, hipStream_t stream) {
  static int n_cu = 0;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_dual_kernel<EPI, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
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
  int grid = 2 * (n_cu > 0 ? n_cu : 256);     // two workgroups per CU
  grid -= grid % 8;                           // keep tile index mod 8 == workgroup index mod 8 (XCD affinity)
  if (grid < 8) grid = 8;
  if (tiles < grid) grid = tiles;
  hipLaunchKernelGGL((gemm_dual_kernel<EPI, ACT>), dim3(grid), dim3(256), LDS_BYTES, stream, p);
  return hipGetLastError();
}

}  // namespace

// EPI_STORE_BF16 / EPI_LNFOLD only (EPI_RESID writes per-256-column row statistics: it stays on gemm_persist.hip)
hipError_t ce_gemm_nt_dual(const GemmParams& p, int epi, hipStream_t stream) {
  if (p.N % BN != 0) return hipErrorInvalidValue;
  switch (epi) {
    case EPI_STORE_BF16: return launch_dual<EPI_STORE_BF16, -1>(p, stream);
    case EPI_LNFOLD:
      if (p.act == CE_ACT_QUICK_GELU) return launch_dual<EPI_LNFOLD, CE_ACT_QUICK_GELU>(p, stream);
      if (p.act == CE_ACT_GELU_ERF) return launch_dual<EPI_LNFOLD, CE_ACT_GELU_ERF>(p, stream);
      return launch_dual<EPI_LNFOLD, -1>(p, stream);
    default: return hipErrorInvalidValue;
  }
}
