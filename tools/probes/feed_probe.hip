// Probe (round 6): is the 58 GB/s-per-CU ceiling of the GEMM's L2 -> LDS feed a property of LDS-DMA?
// One workgroup of 8 waves per CU streams the operand stages of the persistent GEMMs (per stage 256 rows x 128 B of an A panel +
// 256 rows x 128 B of a W panel = 64 KB, pieces of 8 rows x 128 B = 8 whole cache lines per wave instruction) into a 128-KiB LDS
// ring through
//   FEED 0  LDS-DMA only (global_load_lds_dwordx4: what every GEMM of the tree does)
//   FEED 1  global_load_dwordx4 -> VGPR -> ds_write_b128 only
//   FEED 2  both at once: A by LDS-DMA, W through registers
//   FEED 3  both at once the other way round: W by LDS-DMA, A through registers
//   FEED 4  global_load_dwordx4 -> VGPR, never written to LDS (the load path without the LDS write)
// in two forms:
//   BARE  nothing else runs; DEPTH stages stay in flight per wave (8 pieces each)
//   LOOP  the GEMM's own schedule around it: two phases per stage with 16 + 8 ds_read_b128 of the ring, the stage's MFMAs
//         (MMA 1: 64 x v_mfma_f32_16x16x32_bf16 per wave, MMA 2: 16 x v_mfma_scale_f32_32x32x64_f8f6f4), the four barriers, the two
//         wave rows half a phase apart, pieces issued where gemm_persist.hip / gemm_fp8.hip issue them, their counted waits.
//         Results are garbage on purpose (nothing orders a read behind its piece beyond what the real kernel does; the register
//         feeds write their stage one stage after the loads were issued).
// Panel placement (SHARE):
//   0  L2-resident: the 32 workgroups of an XCD (blockIdx & 7) read 8 A panels and 4 W panels, the same ones every round
//   1  private: every workgroup its own A and W panel, the same ones every round (128 MB in all: Infinity-Cache resident, no L2 sharing)
//   2  GEMM-like: 8 new A panels per XCD and round (streamed from HBM / Infinity Cache), W panels shared by the XCDs
// Data: zeros or random (bf16 normal variates / e4m3 bytes).  Printed per case: ms, GB/s per CU, bytes per shader cycle per CU, and the
// in-kernel clock (s_memtime / s_memrealtime of thread 0, median over workgroups).
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/feed_probe.bin tools/probes/feed_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define LDS_PTR(off) ((__attribute__((address_space(3))) void*)(smem + (off)))
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void glds16_at(const char* base, unsigned off, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void gld16(u32x4_t& r, const char* base, unsigned off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void dsw16(unsigned lds_addr, u32x4_t& r) {
  asm volatile("ds_write_b128 %0, %1" : : "v"(lds_addr), "v"(r) : "memory");
}
template <int N> __device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

struct Params {
  const char* A;
  const char* W;
  size_t ld;           // bytes per row
  int kbytes;          // bytes along K per tile (multiple of 512)
  int rounds;          // tiles per workgroup
  int share;
  int pf_dist;        // PF kernels: stages ahead that the XCD's workgroups touch their share of the panels' lines into the L2
  unsigned long long* stamps;   // [grid][4]: memtime start, end, memrealtime start, end
  float* sink;
};

__device__ __forceinline__ void panels(const Params& p, int round, const char*& Ablk, const char*& Wblk) {
  const int b = blockIdx.x, xcd = b & 7, pos = b >> 3, tm = pos & 7, tn = (pos >> 3) & 3;
  const size_t panel = 256 * p.ld;
  if (p.share == 0) { Ablk = p.A + (size_t)(xcd * 8 + tm) * panel; Wblk = p.W + (size_t)tn * panel; }
  else if (p.share == 1) { Ablk = p.A + (size_t)b * panel; Wblk = p.W + (size_t)b * panel; }
  else { Ablk = p.A + (size_t)((round * 8 + xcd) * 8 + tm) * panel; Wblk = p.W + (size_t)((round % 3) * 4 + tn) * panel; }
}

// PF: three waves of every workgroup touch (4 bytes per 64-byte half line, LDS-DMA into a dummy KiB: no register, counted in vmcnt) the
// workgroup's 8 rows of each of the 12 panels its XCD reads in stage (round, kb): a line is then requested from beyond the L2 once,
// pf_dist stages before its 4-8 readers arrive.  (blockIdx & 7 labels the XCD under the observed round-robin placement: speed only.)
__device__ __forceinline__ void touch(const Params& p, int w, int lane, int round, int kb, unsigned lds_dummy) {
  const int b = blockIdx.x, xcd = b & 7, pos = b >> 3;
  const size_t panel = 256 * p.ld;
  const int row = pos * 8 + ((lane >> 1) & 7), half = lane & 1, pi = lane >> 4;           // 4 panels per wave
  const char* base;
  if (w < 2) {
    const int tm = pi + 4 * w;
    base = p.share == 0 ? p.A + (size_t)(xcd * 8 + tm) * panel : p.A + (size_t)((round * 8 + xcd) * 8 + tm) * panel;
  } else {
    base = p.share == 0 ? p.W + (size_t)pi * panel : p.W + (size_t)((round % 3) * 4 + pi) * panel;
  }
  const char* src = base + (size_t)row * p.ld + kb + half * 64;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(src), "s"(lds_dummy) : "memory");
}

// ------------------------------------------------------------------------------------------------------------------------------
// BARE: stream only.  Wave w stages the A blocks (rows 8w + 64i) and the W blocks (rows 8w + 64i), i = 0..3.
template <int FEED, int DEPTH, bool PF = false>
__global__ __launch_bounds__(512) void bare_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(0));
  const int dg = lane >> 3;
  const unsigned off = (unsigned)((8 * w + dg) * p.ld) + (unsigned)(((lane & 7) ^ (((w & 1) << 2) | (dg >> 1))) * 16);
  constexpr bool A_REG = FEED == 1 || FEED == 3 || FEED == 4, W_REG = FEED == 1 || FEED == 2 || FEED == 4;
  constexpr bool WRITE = FEED != 4;
  u32x4_t ra[DEPTH][4], rw[DEPTH][4];
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  int stage = 0;
  for (int round = 0; round < p.rounds; ++round) {
    const char *Ablk, *Wblk;
    panels(p, round, Ablk, Wblk);
    for (int kb = 0; kb < p.kbytes; kb += 128 * DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int buf = (stage & 1) * 65536;
        // write out what this register set holds (the stage issued DEPTH stages ago: every younger stage stays in flight)
        if (A_REG || W_REG) {
          if (stage >= DEPTH) {
            vmwait<8 * (DEPTH - 1)>();
            if (WRITE) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                if (A_REG) dsw16(lds0 + buf + (w + 8 * i) * 1024 + lane * 16, ra[d][i]);
                if (W_REG) dsw16(lds0 + buf + 32768 + (w + 8 * i) * 1024 + lane * 16, rw[d][i]);
              }
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i) asm volatile("" : : "v"(ra[d][i]), "v"(rw[d][i]));
            }
          }
        } else if (PF && w < 3) {
          vmwait<9 * (DEPTH - 1)>();
        } else {
          vmwait<8 * (DEPTH - 1)>();
        }
        if (PF && w < 3) {
          const int ts = stage + p.pf_dist, spt = p.kbytes / 128;
          const int tr = ts / spt < p.rounds ? ts / spt : p.rounds - 1;
          touch(p, w, lane, tr, (ts % spt) * 128, lds0 + 131072);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (A_REG) gld16(ra[d][i], Ablk + kb + d * 128 + (size_t)i * 64 * p.ld, off);
          else glds16_at(Ablk + kb + d * 128 + (size_t)i * 64 * p.ld, off, lds0 + buf + (w + 8 * i) * 1024);
          if (W_REG) gld16(rw[d][i], Wblk + kb + d * 128 + (size_t)i * 64 * p.ld, off);
          else glds16_at(Wblk + kb + d * 128 + (size_t)i * 64 * p.ld, off, lds0 + buf + 32768 + (w + 8 * i) * 1024);
        }
        ++stage;
      }
    }
  }
  vmwait<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    p.stamps[blockIdx.x * 4 + 0] = t0; p.stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
    p.stamps[blockIdx.x * 4 + 2] = r0; p.stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
  if (p.sink && threadIdx.x == 9999) p.sink[0] = *(float*)smem;
}

// ------------------------------------------------------------------------------------------------------------------------------
// LOOP: the persistent GEMM's phase structure.  Per stage s on buffer b = s & 1:
//   PA: 16 ds_read_b128 (W both k halves, A row half 0) | issue A(half 1) of stage s+1 (2 pieces) | wait | barrier | half the MFMAs | barrier
//   PB:  8 ds_read_b128 (A row half 1)                  | issue W (4) + A(half 0) (2) of stage s+2 | wait | barrier | other half   | barrier
// Register feeds: a group's loads are written to LDS at the same point of the NEXT stage, right before the group is issued again.
template <int FEED, int MMA, bool PF = false>
__global__ __launch_bounds__(512, 2) void loop_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = w >> 2, wc = w & 3;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(0));
  const int dg = lane >> 3;
  const unsigned off = (unsigned)((8 * w + dg) * p.ld) + (unsigned)(((lane & 7) ^ (((w & 1) << 2) | (dg >> 1))) * 16);
  constexpr bool A_REG = FEED == 1 || FEED == 3, W_REG = FEED == 1 || FEED == 2;
  constexpr bool ANY_DMA = !(A_REG && W_REG);
  // fragment read address (the fp8 kernel's: 32-row tiles, two chunks of the lane's row)
  const int r32 = lane & 31, h = lane >> 5;
  const int rd0 = (r32 >> 3) * 1024 + (r32 & 7) * 128 + ((((2 * h) ^ ((r32 >> 1) & 7))) << 4);
  const int a_base = wr * 16 * 1024, w_base = 32768 + wc * 8 * 1024;

  u32x4_t rw[4], rah0[2], rah1[2];
  u32x4_t fa[8], fb[8];                       // 8 + 8 ds_read_b128 results (one phase's fragments; W kept for both phases)
  f32x4_t acc4[16];
  f32x16_t acc16[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc4[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc16[i][e] = 0.f;

  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  if (wr == 1) asm volatile("s_barrier" ::: "memory");     // second wave row half a phase behind

#define RD(dst, offs) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"((unsigned)(lds0 + (offs))) : "memory")
#define MMA_HALF(half)                                                                                                  \
  do {                                                                                                                  \
    __builtin_amdgcn_s_setprio(1);                                                                                      \
    if constexpr (MMA == 1) {                                                                                           \
      _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                                  \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                     \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                     \
        acc4[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb[kh * 4 + j]),         \
                              __builtin_bit_cast(bf16x8_t, fa[kh * 4 + i]), acc4[i * 4 + j], 0, 0, 0);                  \
    } else if constexpr (MMA == 2) {                                                                                    \
      _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                                  \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                     \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                   \
        const i32x8_t b8 = {(int)fb[(kh * 2 + j) * 2][0], (int)fb[(kh * 2 + j) * 2][1], (int)fb[(kh * 2 + j) * 2][2], (int)fb[(kh * 2 + j) * 2][3], \
                            (int)fb[(kh * 2 + j) * 2 + 1][0], (int)fb[(kh * 2 + j) * 2 + 1][1], (int)fb[(kh * 2 + j) * 2 + 1][2], (int)fb[(kh * 2 + j) * 2 + 1][3]}; \
        const i32x8_t a8 = {(int)fa[(kh * 2 + i) * 2][0], (int)fa[(kh * 2 + i) * 2][1], (int)fa[(kh * 2 + i) * 2][2], (int)fa[(kh * 2 + i) * 2][3], \
                            (int)fa[(kh * 2 + i) * 2 + 1][0], (int)fa[(kh * 2 + i) * 2 + 1][1], (int)fa[(kh * 2 + i) * 2 + 1][2], (int)fa[(kh * 2 + i) * 2 + 1][3]}; \
        acc16[((half) * 2 + i) * 2 + j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, acc16[((half) * 2 + i) * 2 + j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
      }                                                                                                                 \
    }                                                                                                                   \
    __builtin_amdgcn_s_setprio(0);                                                                                      \
  } while (0)
#define SYNC_MMA(half)                                                                                                  \
  do {                                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);                               \
    asm volatile("s_barrier" ::: "memory"); __builtin_amdgcn_sched_barrier(0);                                          \
    MMA_HALF(half);                                                                                                     \
    asm volatile("s_barrier" ::: "memory");                                                                             \
  } while (0)

  int stage = 0;
  for (int round = 0; round < p.rounds; ++round) {
    const char *Ablk, *Wblk;
    panels(p, round, Ablk, Wblk);
    for (int kb = 0; kb < p.kbytes; kb += 128, ++stage) {
      const int b = (stage & 1) * 65536, nb = 65536 - b;
      // ---- PA ----
#pragma unroll
      for (int q = 0; q < 8; ++q) RD(fb[q], b + w_base + ((q >> 1) & 1) * 4096 + (rd0 ^ ((q >> 2) * 64) ^ ((q & 1) * 16)));     // q = 4 kh + 2 j + chunk
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 8; ++q) RD(fa[q], b + a_base + ((q >> 1) & 1) * 4096 + (rd0 ^ ((q >> 2) * 64) ^ ((q & 1) * 16)));
      // A(half 1) of stage s+1 -> other buffer (rows 8w + 64 + 128 i)
      if (A_REG) {
        if (stage > 0) {
          vmwait<6>();
#pragma unroll
          for (int i = 0; i < 2; ++i) dsw16(lds0 + nb + (w + 8 + 16 * i) * 1024 + lane * 16, rah1[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) gld16(rah1[i], Ablk + kb + (size_t)(64 + 128 * i) * p.ld, off);
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_at(Ablk + kb + (size_t)(64 + 128 * i) * p.ld, off, lds0 + nb + (w + 8 + 16 * i) * 1024);
      }
      if (PF && w < 3) {                      // (one more piece per stage in these waves' queues: their counted waits leave 9)
        const int ts = stage + p.pf_dist, spt = p.kbytes / 128;
        const int tr = ts / spt < p.rounds ? ts / spt : p.rounds - 1;
        touch(p, w, lane, tr, (ts % spt) * 128, lds0 + 131072);
        vmwait<9>();
      } else if (ANY_DMA) vmwait<8>();
      SYNC_MMA(0);
      // ---- PB ----
#pragma unroll
      for (int q = 0; q < 8; ++q) RD(fa[q], b + a_base + 8192 + ((q >> 1) & 1) * 4096 + (rd0 ^ ((q >> 2) * 64) ^ ((q & 1) * 16)));
      if (W_REG || A_REG) {
        if (stage > 0) {
          if (W_REG && A_REG) vmwait<2>(); else if (W_REG) vmwait<4>(); else vmwait<2>();
          if (W_REG) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dsw16(lds0 + b + 32768 + (w + 8 * i) * 1024 + lane * 16, rw[i]);
          }
          if (A_REG) {
#pragma unroll
            for (int i = 0; i < 2; ++i) dsw16(lds0 + b + (w + 16 * i) * 1024 + lane * 16, rah0[i]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (W_REG) gld16(rw[i], Wblk + kb + (size_t)i * 64 * p.ld, off);
        else glds16_at(Wblk + kb + (size_t)i * 64 * p.ld, off, lds0 + b + 32768 + (w + 8 * i) * 1024);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (A_REG) gld16(rah0[i], Ablk + kb + (size_t)(128 * i) * p.ld, off);
        else glds16_at(Ablk + kb + (size_t)(128 * i) * p.ld, off, lds0 + b + (w + 16 * i) * 1024);
      }
      if (PF && w < 3) vmwait<9>(); else if (ANY_DMA) vmwait<8>();
      SYNC_MMA(1);
    }
  }
  if (wr == 0) asm volatile("s_barrier" ::: "memory");
  vmwait<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (A_REG || W_REG) asm volatile("" : : "v"(rw[0]), "v"(rw[1]), "v"(rw[2]), "v"(rw[3]), "v"(rah0[0]), "v"(rah0[1]), "v"(rah1[0]), "v"(rah1[1]));
  if (threadIdx.x == 0) {
    p.stamps[blockIdx.x * 4 + 0] = t0; p.stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
    p.stamps[blockIdx.x * 4 + 2] = r0; p.stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc16[i][e];
  if (p.sink) p.sink[blockIdx.x * 512 + threadIdx.x] = s;
}

// ------------------------------------------------------------------------------------------------------------------------------
struct Result { double ms, clock_ghz; };

template <typename K>
Result time_kernel(K kern, const Params& p, int reps) {
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 132096));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 132096, 0, p);
  std::vector<double> ts;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 132096, 0, p);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
  }
  CK(hipGetLastError());
  std::sort(ts.begin(), ts.end());
  std::vector<unsigned long long> st(256 * 4);
  CK(hipMemcpy(st.data(), p.stamps, st.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> clk;
  for (int i = 0; i < 256; ++i) {
    const double cyc = (double)(st[i * 4 + 1] - st[i * 4 + 0]), rt = (double)(st[i * 4 + 3] - st[i * 4 + 2]);
    if (rt > 0) clk.push_back(cyc / rt * 0.1);     // memrealtime ticks at 100 MHz
  }
  std::sort(clk.begin(), clk.end());
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return Result{ts[ts.size() / 2], clk.empty() ? 0.0 : clk[clk.size() / 2]};
}

static void fill(char* dev, size_t bytes, int kind) {
  // kind 0 zeros, 1 bf16 normal variates, 2 e4m3 bytes (no NaN codes)
  if (kind == 0) { CK(hipMemset(dev, 0, bytes)); return; }
  const size_t chunk = 64u << 20;
  std::vector<uint8_t> hbuf(std::min(bytes, chunk));
  std::mt19937_64 rng(1234 + kind);
  if (kind == 1) {
    std::normal_distribution<float> nd(0.f, 1.f);
    uint16_t* q = (uint16_t*)hbuf.data();
    for (size_t i = 0; i < hbuf.size() / 2; ++i) { float f = nd(rng); uint32_t u; memcpy(&u, &f, 4); q[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1)) >> 16); }
  } else {
    for (size_t i = 0; i < hbuf.size(); i += 8) { uint64_t v = rng(); memcpy(&hbuf[i], &v, 8); }
    for (size_t i = 0; i < hbuf.size(); ++i) if ((hbuf[i] & 0x7f) == 0x7f) hbuf[i] ^= 0x40;
  }
  for (size_t o = 0; o < bytes; o += hbuf.size()) CK(hipMemcpy(dev + o, hbuf.data(), std::min(hbuf.size(), bytes - o), hipMemcpyHostToDevice));
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 96;
  const int reps = argc > 2 ? atoi(argv[2]) : 7;
  const bool quick = argc > 3 && atoi(argv[3]) == 1;
  const bool pf_only = argc > 3 && atoi(argv[3]) == 2;
  const int kbytes = 1024;                     // the fp8 K = 1024 shapes (8 stages per tile); the bf16 K = 1024 shapes are 2048
  const size_t ld = (size_t)kbytes, panel = 256 * ld;
  const size_t a_bytes = (size_t)rounds * 64 * panel, w_bytes = 256 * panel;
  char *A, *W; CK(hipMalloc(&A, a_bytes)); CK(hipMalloc(&W, w_bytes));
  unsigned long long* stamps; CK(hipMalloc(&stamps, 256 * 4 * 8));
  float* sink; CK(hipMalloc(&sink, 256 * 512 * 4));
  const double bytes_per_cu = (double)rounds * (kbytes / 128) * 65536.0;
  printf("# feed_probe: 256 workgroups x 512 threads, %d tiles of %d stages (64 KB each) per workgroup = %.1f MB per CU and launch; median of %d launches\n",
         rounds, kbytes / 128, bytes_per_cu / 1e6, reps);
  printf("# form  feed                         share      data    mma    ms     GB/s/CU  B/clk/CU  clock GHz  TB/s chip\n");
  const char* feed_name[5] = {"LDS-DMA only", "registers + ds_write only", "A DMA + W registers", "W DMA + A registers", "registers, no ds_write"};
  const char* share_name[3] = {"L2-hot", "private", "gemm-like"};
  const char* data_name[3] = {"zeros", "bf16", "e4m3"};
  auto report = [&](const char* form, int feed, int depth, int share, int data, int mma, Result r) {
    const double gbs = bytes_per_cu / (r.ms * 1e-3) / 1e9;
    char f[64]; if (depth > 0) snprintf(f, sizeof f, "%s d%d", feed_name[feed], depth); else snprintf(f, sizeof f, "%s", feed_name[feed]);
    printf("%-5s %-28s %-10s %-6s %-5s %7.3f %8.1f %8.1f %9.3f %9.2f\n", form, f, share_name[share], data_name[data], mma == 0 ? "-" : mma == 1 ? "bf16" : "fp8",
           r.ms, gbs, gbs / r.clock_ghz, r.clock_ghz, gbs * 256 / 1e3);
    fflush(stdout);
  };
  int last_data = -1;
  if (pf_only) {
    // touch-ahead: does requesting a line from beyond the L2 ONCE, some stages before its 4-8 readers arrive, lift the gemm-like rate?
    for (int data : {0, 2, 1}) {
      fill(A, a_bytes, data); fill(W, w_bytes, data);
      for (int share : {0, 2}) {
        Params p{A, W, ld, kbytes, rounds, share, 0, stamps, sink};
        auto rep = [&](const char* form, const char* what, int dist, int mma, Result r) {
          const double gbs = bytes_per_cu / (r.ms * 1e-3) / 1e9;
          char f[64]; if (dist > 0) snprintf(f, sizeof f, "%s, touch %d ahead", what, dist); else snprintf(f, sizeof f, "%s", what);
          printf("%-5s %-28s %-10s %-6s %-5s %7.3f %8.1f %8.1f %9.3f %9.2f\n", form, f, share_name[share], data_name[data], mma == 0 ? "-" : mma == 1 ? "bf16" : "fp8",
                 r.ms, gbs, gbs / r.clock_ghz, r.clock_ghz, gbs * 256 / 1e3);
          fflush(stdout);
        };
        if (data != 1) {
          rep("bare", "LDS-DMA d2", 0, 0, time_kernel(bare_kernel<0, 2>, p, reps));
          for (int dist : {2, 4, 8, 16}) { p.pf_dist = dist; rep("bare", "LDS-DMA d2", dist, 0, time_kernel(bare_kernel<0, 2, true>, p, reps)); }
          rep("loop", "LDS-DMA", 0, 2, time_kernel(loop_kernel<0, 2>, p, reps));
          for (int dist : {2, 4, 8}) { p.pf_dist = dist; rep("loop", "LDS-DMA", dist, 2, time_kernel(loop_kernel<0, 2, true>, p, reps)); }
        }
        if (data != 2) {
          rep("loop", "LDS-DMA", 0, 1, time_kernel(loop_kernel<0, 1>, p, reps));
          for (int dist : {2, 4, 8}) { p.pf_dist = dist; rep("loop", "LDS-DMA", dist, 1, time_kernel(loop_kernel<0, 1, true>, p, reps)); }
        }
      }
    }
    return 0;
  }
  for (int data : {0, 2, 1}) {
    if (quick && data == 1) continue;
    for (int share : {0, 1, 2}) {
      if (data != last_data) { fill(A, a_bytes, data); fill(W, w_bytes, data); last_data = data; }
      Params p{A, W, ld, kbytes, rounds, share, 0, stamps, sink};
      if (data != 1) {     // the bare stream does not look at the data: zeros and one random fill
        report("bare", 0, 1, share, data, 0, time_kernel(bare_kernel<0, 1>, p, reps));
        report("bare", 0, 2, share, data, 0, time_kernel(bare_kernel<0, 2>, p, reps));
        report("bare", 0, 4, share, data, 0, time_kernel(bare_kernel<0, 4>, p, reps));
        report("bare", 1, 1, share, data, 0, time_kernel(bare_kernel<1, 1>, p, reps));
        report("bare", 1, 2, share, data, 0, time_kernel(bare_kernel<1, 2>, p, reps));
        report("bare", 1, 4, share, data, 0, time_kernel(bare_kernel<1, 4>, p, reps));
        report("bare", 2, 2, share, data, 0, time_kernel(bare_kernel<2, 2>, p, reps));
        report("bare", 2, 4, share, data, 0, time_kernel(bare_kernel<2, 4>, p, reps));
        report("bare", 3, 2, share, data, 0, time_kernel(bare_kernel<3, 2>, p, reps));
        report("bare", 4, 2, share, data, 0, time_kernel(bare_kernel<4, 2>, p, reps));
        report("bare", 4, 4, share, data, 0, time_kernel(bare_kernel<4, 4>, p, reps));
        report("loop", 0, 0, share, data, 0, time_kernel(loop_kernel<0, 0>, p, reps));
        report("loop", 1, 0, share, data, 0, time_kernel(loop_kernel<1, 0>, p, reps));
        report("loop", 2, 0, share, data, 0, time_kernel(loop_kernel<2, 0>, p, reps));
        report("loop", 3, 0, share, data, 0, time_kernel(loop_kernel<3, 0>, p, reps));
      }
      if (data != 1) {
        report("loop", 0, 0, share, data, 2, time_kernel(loop_kernel<0, 2>, p, reps));
        // (loop_kernel<1, 2>: all-register feed beside 128 accumulator registers spills -- not a candidate, not run)
        report("loop", 2, 0, share, data, 2, time_kernel(loop_kernel<2, 2>, p, reps));
        report("loop", 3, 0, share, data, 2, time_kernel(loop_kernel<3, 2>, p, reps));
      }
      if (data != 2) {
        report("loop", 0, 0, share, data, 1, time_kernel(loop_kernel<0, 1>, p, reps));
        report("loop", 1, 0, share, data, 1, time_kernel(loop_kernel<1, 1>, p, reps));
        report("loop", 2, 0, share, data, 1, time_kernel(loop_kernel<2, 1>, p, reps));
        report("loop", 3, 0, share, data, 1, time_kernel(loop_kernel<3, 1>, p, reps));
      }
    }
  }
  return 0;
}
