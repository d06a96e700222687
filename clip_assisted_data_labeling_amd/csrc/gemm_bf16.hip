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
// LDS (128 KiB): 4-slot ring of K=32 stages, each A part 16 KiB | W part 16 KiB.  LDS subtile = 16 rows x
// 64 B (1 KiB, what one LDS-DMA instruction writes); 16-B chunk c of row r sits at chunk c ^ (2*(r>>3)):
// every ds_read_b128 fragment read is conflict-free.  The LDS-DMA writes LDS linearly (lane*16), so the
// swizzle is applied to the per-lane SOURCE address and to the read address (guide rule 21).
//
// Schedule: two phases per stage, each {ds_read fragments, issue 2 DMA} s_barrier {16 MFMA} s_barrier:
//   (a) reads W frags + A frags of m-tiles 0-3, issues the W part of stage t+2;
//   (b) reads A frags of m-tiles 4-7, issues the A part of stage t+3, and retires stage t+1 with ONE counted
//       vmcnt that leaves the 3 youngest parts (6 DMA) in flight.
// The two wave rows (wr = 0 / 1: the two waves of every SIMD) run half a phase apart, so one issues MFMAs
// while the other issues LDS reads and DMA.  A slot is re-staged >= 2 phases after its last read; a stage is
// read >= 1 phase after the wait that retired it.
#include <stdlib.h>

#include "common.h"
#include "gemm.h"

namespace {

constexpr int BM = 256, BN = 256;
constexpr int STG = 32768, WPART = 16384;
constexpr int LDS_BYTES = 4 * STG;          // 131072 B

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

// one LDS-DMA instruction: 64 lanes x 16 B -> 1 KiB of LDS at `lds_off` (wave-uniform) + lane*16
__device__ __forceinline__ void glds16(const char* g, char* smem, int lds_off) {
  __builtin_amdgcn_global_load_lds(GLOBAL_PTR(g), LDS_PTR(lds_off), 16, 0, 0);
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
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;

  {
    const size_t lda_b = (size_t)p.lda * 2, ldw_b = (size_t)p.ldw * 2;
    const char* Ablk = (const char*)p.A + (size_t)m0 * lda_b;
    const char* Wblk = (const char*)p.W + (size_t)n0 * ldw_b;
    // DMA instruction j of a part fills subtile 8j + w: rows 16(8j+w) + (lane>>2); LDS chunk lane&3 holds
    // logical chunk (lane&3) ^ (2*((lane>>5)&1))
    const int lrow = 16 * w + (lane >> 2);
    const int lchunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
    int aoff[2], woff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 128 * j + lrow;
      const int ra = min(m0 + r, p.M - 1) - m0;          // M edge: re-read the last valid row (masked at store)
      aoff[j] = (int)(ra * lda_b) + lchunk * 16;
      woff[j] = (int)(r * ldw_b) + lchunk * 16;
    }
    const int dma_lds = w * 1024;
    const int rd = frow * 64 + (((lane >> 4) ^ ((frow >> 3) << 1)) << 4);
    const int a_rd = wr * 8 * 1024 + rd;                 // + slot*STG + mt*1024
    const int w_rd = WPART + wc * 4 * 1024 + rd;         // + slot*STG + nt*1024
    frag_t fa[4], fb[4];

#define STAGE_A(slot, kbyte)                                                                \
  do {                                                                                      \
    glds16(Ablk + (kbyte) + aoff[0], smem, (slot) * STG + dma_lds);                          \
    glds16(Ablk + (kbyte) + aoff[1], smem, (slot) * STG + 8192 + dma_lds);                   \
  } while (0)
#define STAGE_W(slot, kbyte)                                                                \
  do {                                                                                      \
    glds16(Wblk + (kbyte) + woff[0], smem, (slot) * STG + WPART + dma_lds);                  \
    glds16(Wblk + (kbyte) + woff[1], smem, (slot) * STG + WPART + 8192 + dma_lds);           \
  } while (0)
#define LD_W(slot) _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[j] = *(const frag_t*)(smem + (slot) * STG + w_rd + j * 1024);
#define LD_A(slot, half) _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[i] = *(const frag_t*)(smem + (slot) * STG + a_rd + ((half) * 4 + i) * 1024);
#define MMA(half)                                                                           \
  do {                                                                                      \
    __builtin_amdgcn_s_setprio(1);                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                           \
      acc[(half) * 4 + i][j] = Mfma<T>::run(fb[j], fa[i], acc[(half) * 4 + i][j]);          \
    __builtin_amdgcn_s_setprio(0);                                                          \
  } while (0)
#define BARRIER() asm volatile("s_barrier" ::: "memory")
#define WAIT_LDS()                                                                          \
  do {                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
#define STAGE(slot, kb)                                                                     \
  do {                                                                                      \
    /* phase a */                                                                           \
    LD_W(slot) __builtin_amdgcn_sched_barrier(0); LD_A(slot, 0)                             \
    if ((kb) + 128 < kend) STAGE_W(((slot) + 2) & 3, (kb) + 128);                           \
    BARRIER(); WAIT_LDS(); MMA(0); BARRIER();                                               \
    /* phase b */                                                                           \
    LD_A(slot, 1)                                                                           \
    if ((kb) + 192 < kend) { STAGE_A(((slot) + 3) & 3, (kb) + 192); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); } \
    else if ((kb) + 128 < kend) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");            \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
    BARRIER(); WAIT_LDS(); MMA(1); BARRIER();                                               \
  } while (0)

    const int kend = p.K * 2;                // bytes along K; one stage = 64 B; K % 128 == 0 -> >= 4 stages
    STAGE_A(0, 0); STAGE_W(0, 0); STAGE_A(1, 64); STAGE_W(1, 64); STAGE_A(2, 128);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    BARRIER();
    if (wr == 1) BARRIER();                  // second wave row runs half a phase behind
    for (int kb = 0; kb < kend; kb += 256) {
      STAGE(0, kb);
      STAGE(1, kb + 64);
      STAGE(2, kb + 128);
      STAGE(3, kb + 192);
    }
    if (wr == 0) BARRIER();                  // re-align the two wave rows
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
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<T, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
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
    if (dtype == CE_DT_F16) return launch_t<_Float16, EPI_THRESH>(p, stream);
    if (dtype == CE_DT_BF16) return launch_t<__bf16, EPI_THRESH>(p, stream);
  }
  return hipErrorInvalidValue;
}
