// bf16/f16 "NT" GEMM:  C[M,N] = A[M,K] . W[N,K]^T  with fp32 accumulation.  gfx950 only: MFMA 16x16x32,
// LDS-DMA staging (global_load_lds_dwordx4), 8 waves.
//
// This file holds (a) the host entry ce_gemm_nt, which validates the shape contract and routes the bf16
// epilogues of the ViT tower (K1, K3, K5, K6, K7 of SURVEY.md §2.2) to the persistent kernel in
// gemm_persist.hip, and (b) the one-tile-per-workgroup kernel that carries the two remaining epilogues:
// plain fp32 output (operator-level parity tests) and, on f16 operands, the thresholded upper-triangular
// E.E^T of the near-duplicate search (K11; /root/reference/_2_remove_duplicates.py:69-80).
//
// Tile: 256 x 256 per workgroup, 8 waves as 2 (M) x 4 (N), each wave owns 128 x 64 of C held as
// acc[8 m-tiles][4 n-tiles] of 16x16 (128 fp32 VGPRs).  The MFMA is issued as D = Wfrag . Afrag^T, so a lane
// ends up with 4 CONSECUTIVE output columns of one row (row = lane&15, cols = 4*(lane>>4)+reg).
//
// LDS (128 KiB): two buffers of K=64 stages, (A 256 rows x 128 B | W 256 rows x 128 B) each, staged exactly as in
// gemm_persist.hip (read its header): every LDS-DMA piece = 8 whole cache lines, 16-B chunk c of row r at c ^ (r & 6),
// two phases of 32 MFMAs per stage and wave with the two wave rows half a phase apart, A(half 1) of stage s+1 issued in
// PA of stage s, W + A(half 0) of stage s+2 in PB, one vmcnt(8) per phase.  This kernel runs ONE tile per workgroup: past
// the last stage the same pieces are issued again for the last stage's k range (rows that nobody reads any more), so the
// counted waits stay uniform, and the pipeline is drained before the epilogue.
#include <stdlib.h>

#include "common.h"
#include "gemm.h"

namespace {

constexpr int BM = 256, BN = 256;
constexpr int BUF = 65536, WREG = 32768;    // one K=64 stage: A region | W region
constexpr int LDS_BYTES = 2 * BUF;          // 131072 B

#define LDS_PTR(off) ((__attribute__((address_space(3))) void*)(smem + (off)))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <typename T> struct Mfma;
template <> struct Mfma<__bf16> {
  typedef bf16x8_t frag;
  static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma<_Float16> {
  typedef f16x8_t frag;
  static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// one LDS-DMA instruction: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset) -> 1 KiB of LDS at lds_addr
// (inline asm: the builtin builds a 64-bit VGPR address per piece; every wait on these pieces is hand-placed)
__device__ __forceinline__ void glds16_at(const char* base, unsigned off, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_addr) : "memory");
}

template <typename T, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel(const GemmParams p) {
  typedef typename Mfma<T>::frag frag_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  int tm, tn;
  if (EPI == EPI_THRESH) {
    // upper-triangular tile list, row-major: row tm holds tiles tn = tm .. TT-1
    const int TT = tiles_n;
    const int b = blockIdx.x;
    int t = (int)(((2.0 * TT + 1.0) - sqrt((2.0 * TT + 1.0) * (2.0 * TT + 1.0) - 8.0 * (double)b)) * 0.5);
    t = max(0, min(t, TT - 1));
    while (t > 0 && t * TT - t * (t - 1) / 2 > b) --t;
    while ((t + 1) * TT - (t + 1) * t / 2 <= b) ++t;
    tm = t;
    tn = t + (b - (t * TT - t * (t - 1) / 2));
  } else {
    // XCD-aware (blocks b, b+8, ... share an L2; bijective remap, guide §5) + groups of 8 tiles along M
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    constexpr int GM = 8;
    const int group = bid / (GM * tiles_n);
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    tm = first_m + (bid % (GM * tiles_n)) % gsz;
    tn = (bid % (GM * tiles_n)) / gsz;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  f32x4_t acc[8][4];
  const int frow = lane & 15;

  {
    const size_t lda_b = (size_t)p.lda * 2, ldw_b = (size_t)p.ldw * 2;
    const char* Ablk = (const char*)p.A + (size_t)m0 * lda_b;
    const char* Wblk = (const char*)p.W + (size_t)n0 * ldw_b;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(0));
    // LDS-DMA lane mapping and fragment addresses: gemm_persist.hip
    const int dg = lane >> 3;
    const unsigned dchunk16 = (unsigned)(((lane & 7) ^ (dg & 6)) * 16);
    const int arow0 = (w >> 2) * 128 + (2 * (w & 3)) * 8 + dg;
    const int a_dma = ((w >> 2) * 16 + 2 * (w & 3)) * 1024;
#define AOFF(r) ((unsigned)((min(m0 + (r), p.M - 1) - m0) * lda_b) + dchunk16)     /* M edge: re-read the last valid row */
    const unsigned aoff00 = AOFF(arow0), aoff01 = AOFF(arow0 + 8), aoff10 = AOFF(arow0 + 64), aoff11 = AOFF(arow0 + 72);
    const unsigned woff = (unsigned)((32 * w + dg) * ldw_b) + dchunk16;
    const int w_dma = WREG + 4 * w * 1024;
    const int rdl = (frow >> 3) * 1024 + (frow & 7) * 128 + ((((lane >> 4) ^ (frow & 6))) << 4);
    const int a_rd0 = wr * 16 * 1024 + rdl, a_rd1 = a_rd0 ^ 64;
    const int w_rd0 = WREG + wc * 8 * 1024 + rdl, w_rd1 = w_rd0 ^ 64;
    frag_t fa[8], fb[8];
    const int kend = p.K * 2;                // bytes along K; one stage = 128 B; K % 128 == 0
    const int klast = kend - 128;

#define ISSUE_AH0(b, kbyte)                                                                 \
  do {                                                                                      \
    glds16_at(Ablk + (kbyte), aoff00, lds0 + (unsigned)((b) * BUF + a_dma));                \
    glds16_at(Ablk + (kbyte), aoff01, lds0 + (unsigned)((b) * BUF + a_dma + 1024));         \
  } while (0)
#define ISSUE_AH1(b, kbyte)                                                                 \
  do {                                                                                      \
    glds16_at(Ablk + (kbyte), aoff10, lds0 + (unsigned)((b) * BUF + a_dma + 8192));         \
    glds16_at(Ablk + (kbyte), aoff11, lds0 + (unsigned)((b) * BUF + a_dma + 9216));         \
  } while (0)
#define ISSUE_W(b, kbyte)                                                                   \
  do {                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                        \
      glds16_at(Wblk + (kbyte) + (size_t)i_ * 8 * ldw_b, woff, lds0 + (unsigned)((b) * BUF + w_dma + i_ * 1024)); \
  } while (0)
#define LD_W2(b) _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[j] = *(const frag_t*)(smem + (b) * BUF + w_rd0 + j * 2048);  \
                 _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[4 + j] = *(const frag_t*)(smem + (b) * BUF + w_rd1 + j * 2048);
#define LD_A2(b, half) _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[i] = *(const frag_t*)(smem + (b) * BUF + a_rd0 + ((half) * 8 + i * 2) * 1024);  \
                       _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[4 + i] = *(const frag_t*)(smem + (b) * BUF + a_rd1 + ((half) * 8 + i * 2) * 1024);
  // ZC: the first MFMA of every accumulator takes the constant 0 as C (no zeroed registers)
#define MMA2(half, ZC)                                                                      \
  do {                                                                                      \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
      acc[(half) * 4 + i][j] = Mfma<T>::run(fb[kh * 4 + j], fa[kh * 4 + i],                 \
                                            ((ZC) && kh == 0) ? f32x4_t{0.f, 0.f, 0.f, 0.f} : acc[(half) * 4 + i][j]); \
  } while (0)
#define BARRIER() asm volatile("s_barrier" ::: "memory")
#define WAIT_LDS()                                                                          \
  do {                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
#define SYNC_MMA(half, ZC)                                                                  \
  do {                                                                                      \
    WAIT_LDS(); BARRIER(); __builtin_amdgcn_sched_barrier(0);                               \
    __builtin_amdgcn_s_setprio(1); MMA2(half, ZC); __builtin_amdgcn_s_setprio(0);           \
    BARRIER();                                                                              \
  } while (0)
#define VM8 asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
  // stage at byte offset kb on buffer b: PA issues A(half 1) of the next stage, PB issues W + A(half 0) of the one after
  // (both clamped to the last stage's k range once the tile runs out: harmless re-fetches into rows that are dead)
#define STAGE(b, kb) STAGE_Z(b, kb, 0)
#define STAGE_Z(b, kb, ZC)                                                                  \
  do {                                                                                      \
    LD_W2(b) __builtin_amdgcn_sched_barrier(0); LD_A2(b, 0)                                 \
    ISSUE_AH1((b) ^ 1, min((kb) + 128, klast));                                             \
    VM8;                                                                                    \
    SYNC_MMA(0, ZC);                                                                        \
    LD_A2(b, 1)                                                                             \
    { const int k2_ = min((kb) + 256, klast); ISSUE_W(b, k2_); ISSUE_AH0(b, k2_); }         \
    VM8;                                                                                    \
    SYNC_MMA(1, ZC);                                                                        \
  } while (0)

    ISSUE_W(0, 0); ISSUE_AH0(0, 0); ISSUE_AH1(0, 0);
    { const int k1_ = min(128, klast); ISSUE_W(1, k1_); ISSUE_AH0(1, k1_); }
    VM8;                                     // W and A(half 0) of stage 0 have landed
    BARRIER();
    if (wr == 1) BARRIER();                  // second wave row runs half a phase behind
    STAGE_Z(0, 0, 1);                        // first stage: the accumulators start from the constant 0
    STAGE(1, 128);                           // K % 128 == 0: stages come in pairs
    for (int kb = 256; kb < kend; kb += 256) {
      STAGE(0, kb);
      STAGE(1, kb + 128);
    }
    if (wr == 0) BARRIER();                  // re-align the two wave rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the redundant tail pieces land before the LDS is released
  }

  // ---------------------------------- epilogue ----------------------------------
  const int ncol0 = n0 + wc * 64 + (lane >> 4) * 4;   // + nt*16
  const int mrow0 = m0 + wr * 128 + frow;             // + mt*16

  if constexpr (EPI == EPI_STORE_F32) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = mrow0 + mt * 16;
      if (m < p.M) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          float* o = (float*)p.out + (size_t)m * p.ldo + ncol0 + nt * 16;
          *(f32x4_t*)o = acc[mt][nt];
        }
      }
    }
  } else if constexpr (EPI == EPI_THRESH) {
    // /root/reference/_2_remove_duplicates.py:74: where(triu(S, 1) > threshold) -> (i, j, S[i][j])
    const float thr = p.fp16_compare ? (float)(_Float16)p.thr : p.thr;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int i = mrow0 + mt * 16;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = ncol0 + nt * 16 + e;
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
}

template <typename T, int EPI>
hipError_t launch_t(const GemmParams& p, hipStream_t stream) {
  int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  if (EPI == EPI_THRESH) { const int nt = p.N / BN; tiles = nt * (nt + 1) / 2; }
  static DeviceKernelSetup setup;             // per device: LDS opt-in (common.h)
  if (hipError_t e = setup.ensure((const void*)gemm_nt_kernel<T, EPI>, LDS_BYTES, nullptr); e != hipSuccess) return e;
  hipLaunchKernelGGL((gemm_nt_kernel<T, EPI>), dim3(tiles), dim3(512), LDS_BYTES, stream, p);
  return hipGetLastError();
}

}  // namespace

// Host launcher.  Shape contract (checked): N % 256 == 0, K % 128 == 0, M >= 1, 16-B aligned rows.
hipError_t ce_gemm_nt(const GemmParams& p, int dtype, int epi, hipStream_t stream) {
  if (p.M < 1 || p.N < BN || p.N % BN != 0 || p.K < 128 || p.K % 128 != 0) return hipErrorInvalidValue;
  if (p.lda % 8 != 0 || p.ldw % 8 != 0 || p.lda < p.K || p.ldw < p.K) return hipErrorInvalidValue;
  if (((uintptr_t)p.A | (uintptr_t)p.W | (uintptr_t)p.out) & 15) return hipErrorInvalidValue;
  if (epi != EPI_THRESH && !p.out) return hipErrorInvalidValue;
  if ((size_t)255 * p.lda * 2 + 128 >= 0x7fffffffull || (size_t)255 * p.ldw * 2 + 128 >= 0x7fffffffull)
    return hipErrorInvalidValue;
  if (epi == EPI_STORE_BF16 || epi == EPI_LNFOLD || epi == EPI_RESID) {
    if (dtype != CE_DT_BF16) return hipErrorInvalidValue;
    if (epi == EPI_LNFOLD && (!p.colsum || !p.bias || !p.stats_in || p.stats_in_parts < 1 || p.stats_in_parts > 4))
      return hipErrorInvalidValue;
    if (epi == EPI_RESID && (!p.bias || !p.resid || !p.stats_out)) return hipErrorInvalidValue;
    if (epi != EPI_STORE_BF16 && (p.stats_ld < p.M || p.stats_ld % 256 != 0)) return hipErrorInvalidValue;
    return ce_gemm_nt_persist(p, epi, stream);
  }
  if (epi == EPI_STORE_F32) {
    if (dtype == CE_DT_BF16) return launch_t<__bf16, EPI_STORE_F32>(p, stream);
    if (dtype == CE_DT_F16) return launch_t<_Float16, EPI_STORE_F32>(p, stream);
    return hipErrorInvalidValue;
  }
  if (epi == EPI_THRESH) {
    if (!p.tri || p.M != p.N || p.A != p.W || !p.pairs || !p.vals || !p.count || p.n_valid > p.M) return hipErrorInvalidValue;
    if (dtype == CE_DT_F16) return ce_gemm_tri_persist(p, stream);             // persistent pipeline (gemm_tri.hip)
    if (dtype == CE_DT_BF16) return launch_t<__bf16, EPI_THRESH>(p, stream);
  }
  return hipErrorInvalidValue;
}
