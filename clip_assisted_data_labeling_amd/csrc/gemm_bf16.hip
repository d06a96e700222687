// bf16/f16 "NT" GEMM for the ViT tower:  C[M,N] = A[M,K] . W[N,K]^T  with fp32 accumulation and a
// fused epilogue.  gfx950 only: MFMA 16x16x32, LDS-DMA staging (global_load_lds_dwordx4), 8 waves.
//
// This kernel carries K1, K3, K5, K6, K7 of SURVEY.md §2.2 (the arithmetic open_clip performs for
// /root/reference/utils/embedder.py:98) and, with f16 operands, K11 (the E.E^T of
// /root/reference/_2_remove_duplicates.py:69).
//
// Tile: 256 x 256 x 64 per workgroup, 8 waves as 2 (M) x 4 (N), each wave owns 128 x 64 of C held
// as acc[8 m-tiles][4 n-tiles] of 16x16 (128 fp32 VGPRs).  The MFMA is issued as
// D = Wfrag . Afrag^T, so a lane ends up with 4 CONSECUTIVE output columns of one row
// (row = lane&15, cols = 4*(lane>>4)+reg): epilogue loads/stores are 8 B (bf16) / 16 B (f32) per lane.
//
// LDS (128 KiB): 2 buffers x { A rows 0-127 | A rows 128-255 | W rows 0-127 | W rows 128-255 },
// each half-tile 128 rows x 64 k = 16 KiB.  Rows are 128 B; the 16-B chunk index is XOR-ed with
// (row>>1)&7, which makes every ds_read_b128 of a fragment conflict-free (guide T2).  The LDS-DMA
// writes LDS linearly (lane*16), so the swizzle is applied to the per-lane SOURCE address and to the
// read address (guide rule 21).
//
// Schedule: 4 phases per K-tile, each phase = ds_read one register sub-tile + stage one half-tile,
// s_barrier, 16 MFMAs (one 64x32 quadrant of the wave tile x K=64), s_barrier.  The two wave rows
// (wr=0 / wr=1; they are the two waves of every SIMD) run half a phase apart, so one issues MFMAs
// while the other issues LDS reads and DMA.  DMA stays in flight across barriers; one counted
// s_waitcnt vmcnt per K-tile (phase 4) retires the next tile.  Ordering proof is in DESIGN.md §GEMM.
#include "common.h"
#include "gemm.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 128 * BK * 2;          // 16384 B
constexpr int BUF = 4 * HALF;               // 65536 B: A0 A1 W0 W1
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

__device__ __forceinline__ float act_apply(float u, int act) {
  if (act == CE_ACT_QUICK_GELU) return u / (1.0f + __expf(-1.702f * u));      // u * sigmoid(1.702 u)
  if (act == CE_ACT_GELU_ERF) return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
  return u;
}

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

  // ---- tile id: XCD-aware (blocks b, b+8, ... share an L2) + grouped along M ----
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  const int nwg = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;      // bijective (guide §5)
  }
  constexpr int GM = 8;
  int tm, tn;
  if (EPI == EPI_THRESH) {
    // upper-triangular tile list, row-major: row tm holds tiles tn = tm .. T-1
    const int TT = tiles_n;
    const int b = blockIdx.x;
    int t = (int)(((2.0 * TT + 1.0) - sqrt((2.0 * TT + 1.0) * (2.0 * TT + 1.0) - 8.0 * (double)b)) * 0.5);
    t = max(0, min(t, TT - 1));
    while (t > 0 && t * TT - t * (t - 1) / 2 > b) --t;
    while ((t + 1) * TT - (t + 1) * t / 2 <= b) ++t;
    tm = t;
    tn = t + (b - (t * TT - t * (t - 1) / 2));
  } else {
    const int group = bid / (GM * tiles_n);
    const int first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM);
    tm = first_m + (bid % (GM * tiles_n)) % gsz;
    tn = (bid % (GM * tiles_n)) / gsz;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- LDS-DMA source offsets (per lane, relative to the tile's first row) ----
  // instruction j of a half-tile fills LDS rows 64j + 8w + (lane>>3); LDS chunk lane&7 holds logical
  // chunk (lane&7) ^ ((row>>1)&7), and (row>>1)&7 == 4*(w&1) + (lane>>4) for both j.
  const int lrow = 8 * w + (lane >> 3);
  const int lchunk = (lane & 7) ^ (4 * (w & 1) + (lane >> 4));
  const size_t lda_b = (size_t)p.lda * 2, ldw_b = (size_t)p.ldw * 2;
  const char* Ablk = (const char*)p.A + (size_t)m0 * lda_b;
  const char* Wblk = (const char*)p.W + (size_t)n0 * ldw_b;
  int aoff[4], woff[4];                      // [half*2 + j]
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = 64 * i + lrow;                   // row inside the 256-row tile
    int ra = min(m0 + r, p.M - 1) - m0;      // M edge: re-read the last valid row (masked at store)
    aoff[i] = (int)(ra * lda_b) + lchunk * 16;
    woff[i] = (int)(r * ldw_b) + lchunk * 16;
  }
  const int dma_lds = w * 1024;              // + 8192 for j = 1

  // ---- fragment read offsets ----
  const int frow = lane & 15;
  const int rd0 = frow * 128 + (((lane >> 4) ^ ((frow >> 1) & 7)) << 4);      // k-step 0
  const int rd1 = rd0 ^ 64;                                                   // k-step 1 (chunk + 4)
  const int a_base = wr * HALF;                                               // + buf*BUF + mt*2048
  const int w_base = 2 * HALF + (wc >> 1) * HALF + (wc & 1) * 64 * 128;       // + buf*BUF + nt*2048

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  frag_t fa[8], fb0[4], fb1[4];

#define STAGE_A(buf, half, kbyte)                                                            \
  do {                                                                                       \
    glds16(Ablk + (kbyte) + aoff[(half) * 2 + 0], smem, (buf) * BUF + (half) * HALF + dma_lds);          \
    glds16(Ablk + (kbyte) + aoff[(half) * 2 + 1], smem, (buf) * BUF + (half) * HALF + 8192 + dma_lds);   \
  } while (0)
#define STAGE_W(buf, half, kbyte)                                                            \
  do {                                                                                       \
    glds16(Wblk + (kbyte) + woff[(half) * 2 + 0], smem, (buf) * BUF + 2 * HALF + (half) * HALF + dma_lds);        \
    glds16(Wblk + (kbyte) + woff[(half) * 2 + 1], smem, (buf) * BUF + 2 * HALF + (half) * HALF + 8192 + dma_lds); \
  } while (0)
#define LD_A(buf, mi)                                                                        \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
    fa[i * 2 + 0] = *(const frag_t*)(smem + (buf) * BUF + a_base + ((mi) * 4 + i) * 2048 + rd0); \
    fa[i * 2 + 1] = *(const frag_t*)(smem + (buf) * BUF + a_base + ((mi) * 4 + i) * 2048 + rd1); \
  }
#define LD_W(dst, buf, ni)                                                                   \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                            \
    dst[i * 2 + 0] = *(const frag_t*)(smem + (buf) * BUF + w_base + ((ni) * 2 + i) * 2048 + rd0); \
    dst[i * 2 + 1] = *(const frag_t*)(smem + (buf) * BUF + w_base + ((ni) * 2 + i) * 2048 + rd1); \
  }
#define MMA(mi, ni, fbx)                                                                     \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                              \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                            \
    acc[(mi) * 4 + i][(ni) * 2 + j] = Mfma<T>::run(fbx[j * 2 + 0], fa[i * 2 + 0], acc[(mi) * 4 + i][(ni) * 2 + j]); \
    acc[(mi) * 4 + i][(ni) * 2 + j] = Mfma<T>::run(fbx[j * 2 + 1], fa[i * 2 + 1], acc[(mi) * 4 + i][(ni) * 2 + j]); \
  }
#define BARRIER() asm volatile("s_barrier" ::: "memory")
#define WAIT_LDS()                                                                           \
  do {                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  } while (0)
#define COMPUTE(mi, ni, fbx)                                                                 \
  do {                                                                                       \
    __builtin_amdgcn_s_setprio(1);                                                           \
    MMA(mi, ni, fbx)                                                                         \
    __builtin_amdgcn_s_setprio(0);                                                           \
  } while (0)

  const int nk = p.K / BK;                   // even (host-checked)

  // ---- prologue: tile 0 (4 half-tiles) + W0 of tile 1 ----
  STAGE_A(0, 0, 0); STAGE_A(0, 1, 0); STAGE_W(0, 0, 0); STAGE_W(0, 1, 0);
  if (nk > 1) { STAGE_W(1, 0, 128); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  BARRIER();
  if (wr == 1) BARRIER();                    // second wave row runs half a phase behind

  // One K-tile = 4 phases.  `cur` is the buffer being computed; staging targets:
  //   ph1: W1 of tile t+1 -> buf cur^1      ph2: A0 of tile t+1 -> buf cur^1
  //   ph3: A1 of tile t+1 -> buf cur^1      ph4: W0 of tile t+2 -> buf cur   (+ counted vmcnt)
#define KTILE(cur, kb)                                                                       \
  do {                                                                                       \
    const bool has1 = (kb) + 128 < kend, has2 = (kb) + 256 < kend;                           \
    /* phase 1 */                                                                            \
    LD_W(fb0, cur, 0) __builtin_amdgcn_sched_barrier(0); LD_A(cur, 0)                        \
    if (has1) STAGE_W((cur) ^ 1, 1, (kb) + 128);                                             \
    BARRIER(); WAIT_LDS(); COMPUTE(0, 0, fb0); BARRIER();                                    \
    /* phase 2 */                                                                            \
    LD_W(fb1, cur, 1)                                                                        \
    if (has1) STAGE_A((cur) ^ 1, 0, (kb) + 128);                                             \
    BARRIER(); WAIT_LDS(); COMPUTE(0, 1, fb1); BARRIER();                                    \
    /* phase 3 */                                                                            \
    LD_A(cur, 1)                                                                             \
    if (has1) STAGE_A((cur) ^ 1, 1, (kb) + 128);                                             \
    BARRIER(); WAIT_LDS(); COMPUTE(1, 1, fb1); BARRIER();                                    \
    /* phase 4 */                                                                            \
    if (has2) { STAGE_W(cur, 0, (kb) + 256); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); } \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    \
    BARRIER(); COMPUTE(1, 0, fb0); BARRIER();                                                \
  } while (0)

  const int kend = p.K * 2;                  // bytes along K
  for (int kb = 0; kb < kend; kb += 256) {
    KTILE(0, kb);
    KTILE(1, kb + 128);
  }
  if (wr == 0) BARRIER();                    // re-align the two wave rows

  // ---------------------------------- epilogue ----------------------------------
  const int q4 = (lane >> 4) * 4;
  const int ncol0 = n0 + wc * 64 + q4;       // + nt*16
  const int mrow0 = m0 + wr * 128 + frow;    // + mt*16

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
  } else if constexpr (EPI == EPI_STORE_BF16) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = mrow0 + mt * 16;
      if (m < p.M) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          f32x4_t v = acc[mt][nt];
          if (p.bias) { f32x4_t b = *(const f32x4_t*)(p.bias + ncol0 + nt * 16); v += b; }
          uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *(uint2*)((bf16_t*)p.out + (size_t)m * p.ldo + ncol0 + nt * 16) = pk;
        }
      }
    }
  } else if constexpr (EPI == EPI_LNFOLD) {
    // out = act( rstd_m * (acc - mean_m * colsum_n) + bias_n ): LayerNorm folded into the GEMM.
    // mean/rstd come from per-row partial sums (sum, sumsq) left by the producer of A.
    f32x4_t cs[4], bs[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      cs[nt] = *(const f32x4_t*)(p.colsum + ncol0 + nt * 16);
      bs[nt] = *(const f32x4_t*)(p.bias + ncol0 + nt * 16);
    }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = mrow0 + mt * 16;
      if (m < p.M) {
        float s = 0.f, ss = 0.f;
        for (int part = 0; part < p.stats_in_parts; ++part) {
          float2 t = *(const float2*)(p.stats_in + ((size_t)part * p.M + m) * 2);
          s += t.x; ss += t.y;
        }
        const float mean = s * p.inv_width;
        const float var = fmaxf(ss * p.inv_width - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          f32x4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = act_apply(rstd * (acc[mt][nt][e] - mean * cs[nt][e]) + bs[nt][e], p.act);
          uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *(uint2*)((bf16_t*)p.out + (size_t)m * p.ldo + ncol0 + nt * 16) = pk;
        }
      }
    }
  } else if constexpr (EPI == EPI_RESID) {
    // x_new = acc + bias + resid (bf16, may alias out); also per-row (sum, sumsq) of the ROUNDED x_new
    // over this tile's 256 columns -> stats_out[tn][m][2] for the next LayerNorm-folded GEMM.
    float* red = (float*)smem;               // [4 wc][256 rows][2]; LDS is idle after the K loop
    f32x4_t bs[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bs[nt] = *(const f32x4_t*)(p.bias + ncol0 + nt * 16);
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = mrow0 + mt * 16;
      float s = 0.f, ss = 0.f;
      if (m < p.M) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const size_t off = (size_t)m * p.ldo + ncol0 + nt * 16;
          uint2 rr = *(const uint2*)((const bf16_t*)p.resid + off);
          f32x4_t v = acc[mt][nt] + bs[nt];
          v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
          v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
          uint2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *(uint2*)((bf16_t*)p.out + off) = pk;
          float r0 = __uint_as_float(pk.x << 16), r1 = __uint_as_float(pk.x & 0xffff0000u);
          float r2 = __uint_as_float(pk.y << 16), r3 = __uint_as_float(pk.y & 0xffff0000u);
          s += (r0 + r1) + (r2 + r3);
          ss += (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3);
        }
      }
      s += __shfl_xor(s, 16); ss += __shfl_xor(ss, 16);
      s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
      if (lane < 16) {
        const int r = wr * 128 + mt * 16 + lane;
        *(float2*)(red + ((size_t)wc * 256 + r) * 2) = float2{s, ss};
      }
    }
    __syncthreads();
    if (tid < 256 && m0 + tid < p.M) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float2 t = *(const float2*)(red + ((size_t)c * 256 + tid) * 2);
        s += t.x; ss += t.y;
      }
      *(float2*)(p.stats_out + ((size_t)tn * p.M + m0 + tid) * 2) = float2{s, ss};
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

template <typename T>
hipError_t launch_t(const GemmParams& p, int epi, hipStream_t stream) {
  int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  if (epi == EPI_THRESH) { const int nt = p.N / BN; tiles = nt * (nt + 1) / 2; }
  dim3 grid(tiles), block(512);
#define CE_LAUNCH(E)                                                                                     \
  case E: {                                                                                              \
    static bool attr_set = false;                                                                        \
    if (!attr_set) {                                                                                     \
      hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<T, E>,                              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);         \
      if (e != hipSuccess) return e;                                                                     \
      attr_set = true;                                                                                   \
    }                                                                                                    \
    hipLaunchKernelGGL((gemm_nt_kernel<T, E>), grid, block, LDS_BYTES, stream, p);                       \
    break;                                                                                               \
  }
  switch (epi) {
    CE_LAUNCH(EPI_STORE_F32)
    CE_LAUNCH(EPI_STORE_BF16)
    CE_LAUNCH(EPI_LNFOLD)
    CE_LAUNCH(EPI_RESID)
    CE_LAUNCH(EPI_THRESH)
    default: return hipErrorInvalidValue;
  }
#undef CE_LAUNCH
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
  if (epi == EPI_LNFOLD && (!p.colsum || !p.bias || !p.stats_in || p.stats_in_parts < 1)) return hipErrorInvalidValue;
  if (epi == EPI_RESID && (!p.bias || !p.resid || !p.stats_out)) return hipErrorInvalidValue;
  if (epi == EPI_THRESH && (!p.tri || p.M != p.N || p.A != p.W || !p.pairs || !p.vals || !p.count || p.n_valid > p.M))
    return hipErrorInvalidValue;
  if (dtype == CE_DT_BF16) return launch_t<__bf16>(p, epi, stream);
  if (dtype == CE_DT_F16) return launch_t<_Float16>(p, epi, stream);
  return hipErrorInvalidValue;
}
