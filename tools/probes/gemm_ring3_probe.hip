// Probe (round 3): does a DEEPER A ring speed up the bf16 GEMM main loop?
// DESIGN.md section 3.1 (round 3) finds all GEMM shapes of the encoder at 42-47 GB/s of tile traffic per CU, which is what the
// 64 KiB a two-stage ring keeps in flight sustains at the loaded latency.  This file holds the one-tile-per-workgroup kernel of
// csrc/gemm_bf16.hip twice: RING = 2 is that kernel's pipeline as it is (two buffers of A | W, every piece has two phases to
// land); RING = 3 gives A THREE 32-KiB slots next to W's two (160 KiB: the space the product kernel spends on AUX and the
// epilogue images), so that A(half 0) of stage s+3 is issued in PB of stage s and A(half 1) of stage s+2 in PA of stage s:
// 80 KiB in flight per CU at the counted waits and three to four phases for an A piece to land.  Both store bf16 straight from
// the accumulators (8-B pieces, no LDS image: the epilogue is not what is compared).  Driven by tools/gemm_ring3_probe.py.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -shared -fPIC -o tools/probes/libgemm_ring3_probe.so tools/probes/gemm_ring3_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../clip_assisted_data_labeling_amd/csrc/common.h"

namespace {

constexpr int BM = 256, BN = 256;
constexpr int SLOT = 32768;                 // one K=64 stage of ONE operand: 256 rows x 128 B

struct Params { const void* A; const void* W; void* out; int M, N, K; unsigned long long* stamps; };

__device__ __forceinline__ void glds16_at(const char* base, unsigned off, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_addr) : "memory");
}

template <int RING>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int A_OFF = 0, W_OFF = RING * SLOT;          // A slots | W slots (always two)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 2, wc = w & 3;
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  // XCD-aware order, groups of 8 tiles along M (csrc/gemm_bf16.hip)
  const int nwg = tiles_m * tiles_n;
  const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  constexpr int GM = 8;
  const int group = bid / (GM * tiles_n), first_m = group * GM, gsz = min(tiles_m - first_m, GM);
  const int tm = first_m + (bid % (GM * tiles_n)) % gsz, tn = (bid % (GM * tiles_n)) / gsz;
  const int m0 = tm * BM, n0 = tn * BN;

  f32x4_t acc[8][4];
  const int frow = lane & 15;
  const size_t lda_b = (size_t)p.K * 2, ldw_b = (size_t)p.K * 2;
  const char* Ablk = (const char*)p.A + (size_t)m0 * lda_b;
  const char* Wblk = (const char*)p.W + (size_t)n0 * ldw_b;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) void*)smem));
  const int dg = lane >> 3;
  const unsigned dchunk16 = (unsigned)(((lane & 7) ^ (dg & 6)) * 16);
  const int arow0 = (w >> 2) * 128 + (2 * (w & 3)) * 8 + dg;
  const int a_dma = ((w >> 2) * 16 + 2 * (w & 3)) * 1024;
#define AOFF(rr) ((unsigned)((min(m0 + (rr), p.M - 1) - m0) * lda_b) + dchunk16)
  const unsigned aoff00 = AOFF(arow0), aoff01 = AOFF(arow0 + 8), aoff10 = AOFF(arow0 + 64), aoff11 = AOFF(arow0 + 72);
  const unsigned woff = (unsigned)((32 * w + dg) * ldw_b) + dchunk16;
  const int w_dma = 4 * w * 1024;
  const int rdl = (frow >> 3) * 1024 + (frow & 7) * 128 + ((((lane >> 4) ^ (frow & 6))) << 4);
  const int a_rd0 = wr * 16 * 1024 + rdl, a_rd1 = a_rd0 ^ 64;
  const int w_rd0 = wc * 8 * 1024 + rdl, w_rd1 = w_rd0 ^ 64;
  bf16x8_t fa[8], fb[8];
  const int kend = p.K * 2, klast = kend - 128;
  const int S = kend / 128;                                   // stages

  // slot = byte offset of the operand slot inside smem
#define ISSUE_AH0(slot, kbyte)                                                              \
  do {                                                                                      \
    glds16_at(Ablk + (kbyte), aoff00, lds0 + (unsigned)((slot) + a_dma));                   \
    glds16_at(Ablk + (kbyte), aoff01, lds0 + (unsigned)((slot) + a_dma + 1024));            \
  } while (0)
#define ISSUE_AH1(slot, kbyte)                                                              \
  do {                                                                                      \
    glds16_at(Ablk + (kbyte), aoff10, lds0 + (unsigned)((slot) + a_dma + 8192));            \
    glds16_at(Ablk + (kbyte), aoff11, lds0 + (unsigned)((slot) + a_dma + 9216));            \
  } while (0)
#define ISSUE_W(slot, kbyte)                                                                \
  do {                                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                        \
      glds16_at(Wblk + (kbyte) + (size_t)i_ * 8 * ldw_b, woff, lds0 + (unsigned)((slot) + w_dma + i_ * 1024)); \
  } while (0)
#define LD_W2(slot) _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8_t*)(smem + (slot) + w_rd0 + j * 2048);  \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) fb[4 + j] = *(const bf16x8_t*)(smem + (slot) + w_rd1 + j * 2048);
#define LD_A2(slot, half) _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[i] = *(const bf16x8_t*)(smem + (slot) + a_rd0 + ((half) * 8 + i * 2) * 1024);  \
                          _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[4 + i] = *(const bf16x8_t*)(smem + (slot) + a_rd1 + ((half) * 8 + i * 2) * 1024);
#define MMA2(half)                                                                          \
  do {                                                                                      \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                        \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                           \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                      \
      const int j = (i & 1) ? 3 - j_ : j_;                                                  \
      acc[(half) * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kh * 4 + j], fa[kh * 4 + i], acc[(half) * 4 + i][j], 0, 0, 0); \
    }                                                                                       \
  } while (0)
#define BARRIER() asm volatile("s_barrier" ::: "memory")
#define SYNC_MMA(half)                                                                      \
  do {                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);   \
    BARRIER(); __builtin_amdgcn_sched_barrier(0);                                           \
    __builtin_amdgcn_s_setprio(1); MMA2(half); __builtin_amdgcn_s_setprio(0);               \
    BARRIER();                                                                              \
  } while (0)
#define KB(s_) min((s_) * 128, klast)

#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  unsigned long long t0 = 0, c0 = 0;
  if constexpr (RING == 2) {
    // ---- the shipped pipeline: PA(s) issues A(half 1)(s+1), PB(s) issues W(s+2) + A(half 0)(s+2); vmcnt(8) per phase ----
    ISSUE_W(W_OFF, 0); ISSUE_AH0(A_OFF, 0); ISSUE_AH1(A_OFF, 0);
    ISSUE_W(W_OFF + SLOT, KB(1)); ISSUE_AH0(A_OFF + SLOT, KB(1));
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    BARRIER();
    if (p.stamps && tid == 0) { t0 = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_amdgcn_s_memtime(); }
    if (wr == 1) BARRIER();
    for (int s = 0; s < S; ++s) {
      const int b = (s & 1) * SLOT;
      LD_W2(W_OFF + b) __builtin_amdgcn_sched_barrier(0); LD_A2(A_OFF + b, 0)
      ISSUE_AH1(A_OFF + (SLOT - b), KB(s + 1));
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      SYNC_MMA(0);
      LD_A2(A_OFF + b, 1)
      ISSUE_W(W_OFF + b, KB(s + 2)); ISSUE_AH0(A_OFF + b, KB(s + 2));
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      SYNC_MMA(1);
    }
  } else {
    // ---- three A slots: PA(s) issues A(half 1)(s+2) into slot (s+2) % 3, PB(s) issues W(s+2) into W slot s % 2 and
    //      A(half 0)(s+3) into A slot s % 3.  Queue order in steady state: PB(s-1): W(s+1) x4, Ah0(s+2) x2 | PA(s): Ah1(s+2) x2 |
    //      PB(s): W(s+2) x4, Ah0(s+3) x2.  The wait at the end of PB(s) leaves the 10 youngest in flight (Ah0(s+3), W(s+2),
    //      Ah1(s+2), Ah0(s+2)) and so retires all of stage s+1; PA's wait (16) never binds.  Re-staging happens one phase after a
    //      region's last read, as in the two-slot pipeline (same ordering argument). ----
    ISSUE_AH0(A_OFF, 0); ISSUE_AH1(A_OFF, 0);
    ISSUE_W(W_OFF, 0); ISSUE_AH0(A_OFF + SLOT, KB(1));
    ISSUE_AH1(A_OFF + SLOT, KB(1));
    ISSUE_W(W_OFF + SLOT, KB(1)); ISSUE_AH0(A_OFF + 2 * SLOT, KB(2));
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");            // W(0), A(0) have landed
    BARRIER();
    if (p.stamps && tid == 0) { t0 = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_amdgcn_s_memtime(); }
    if (wr == 1) BARRIER();
    int sa = 0;                                                   // A slot of stage s (s % 3), as a byte offset
    for (int s = 0; s < S; ++s) {
      const int wb = W_OFF + (s & 1) * SLOT;
      const int sa2 = sa >= SLOT ? sa - SLOT : sa + 2 * SLOT;     // (s + 2) % 3
      LD_W2(wb) __builtin_amdgcn_sched_barrier(0); LD_A2(A_OFF + sa, 0)
      ISSUE_AH1(A_OFF + sa2, KB(s + 2));
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      SYNC_MMA(0);
      LD_A2(A_OFF + sa, 1)
      ISSUE_W(wb, KB(s + 2)); ISSUE_AH0(A_OFF + sa, KB(s + 3));
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      SYNC_MMA(1);
      sa = sa == 2 * SLOT ? 0 : sa + SLOT;
    }
  }
  if (wr == 0) BARRIER();
  if (p.stamps && tid == 0) {
    p.stamps[(size_t)blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime() - t0;
    p.stamps[(size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime() - c0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: bf16, 8-B pieces straight from the MFMA layout ----
  const int ncol0 = n0 + wc * 64 + (lane >> 4) * 4;
  const int mrow0 = m0 + wr * 128 + frow;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int m = mrow0 + mt * 16;
    if (m < p.M) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bf16_t* o = (bf16_t*)p.out + (size_t)m * p.N + ncol0 + nt * 16;
        *(uint2*)o = uint2{pack_bf16x2(acc[mt][nt][0], acc[mt][nt][1]), pack_bf16x2(acc[mt][nt][2], acc[mt][nt][3])};
      }
    }
  }
}

template <int RING>
int launch(const Params& p, hipStream_t st) {
  const int lds = (RING + 2) * SLOT;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute((const void*)gemm_ring_kernel<RING>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -2;
    done = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  hipLaunchKernelGGL((gemm_ring_kernel<RING>), dim3(tiles), dim3(512), lds, st, p);
  return (int)hipGetLastError();
}

}  // namespace

// out bf16 [M][N] = A[M][K] . W[N][K]^T; N % 256 == 0, K % 128 == 0; stamps (optional) [tiles][4]: main loop 100 MHz ticks, cycles
extern "C" int gemm_ring_probe(int ring, const void* a, const void* w, void* out, int m, int n, int k, void* stamps, void* stream) {
  if (n % 256 != 0 || k % 128 != 0 || k < 384 || m < 1) return -1;
  Params p{a, w, out, m, n, k, (unsigned long long*)stamps};
  return ring == 3 ? launch<3>(p, (hipStream_t)stream) : launch<2>(p, (hipStream_t)stream);
}
