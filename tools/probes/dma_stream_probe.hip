// Probe: how fast can all CUs stream a row-major [M][K] bf16 matrix into LDS with LDS-DMA when a piece takes
// W bytes per row (W = 64: what the GEMM ring does today; 128 / 256: whole L2 lines per row)?  No MFMA, no LDS reads.
// hipcc --offload-arch=gfx950 -O3 dma_stream_probe.hip -o dma_stream_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define LDS_PTR(off) ((__attribute__((address_space(3))) void*)(smem + (off)))
__device__ __forceinline__ void glds16(const char* base, unsigned off, char* smem, int lds_off) {
  const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)LDS_PTR(lds_off));
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_addr) : "memory");
}

// one "stage" = 256 rows x 64 B = 16 KiB = 16 pieces = 2 per wave, whatever W is (W/64 stages share a row segment)
template <int W>
__global__ __launch_bounds__(512) void stream_kernel(const char* A, size_t ld_bytes, int m_tiles, int kbytes, unsigned long long* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPR = W / 16;                 // lanes per row
  constexpr int RPP = 64 / LPR;               // rows per piece
  // a group of W/64 stages covers 256 rows x W bytes = (256 / RPP) pieces; wave w takes pieces w, w+8, ...
  constexpr int PIECES = 256 / RPP;
  constexpr int PPW = PIECES / 8;             // pieces per wave per group
  unsigned off[PPW];
  for (int j = 0; j < PPW; ++j) off[j] = (unsigned)(((w + 8 * j) * RPP + lane / LPR) * ld_bytes + (lane % LPR) * 16);
  int slot = 0;
  for (int t = blockIdx.x; t < m_tiles; t += gridDim.x) {
    const char* blk = A + (size_t)t * 256 * ld_bytes;
    for (int kb = 0; kb < kbytes; kb += W) {
#pragma unroll
      for (int j = 0; j < PPW; ++j) glds16(blk + kb, off[j], smem, slot * 16384 * (W / 64) % 131072 + (w + 8 * j) * 1024);
      slot = (slot + 1) & 3;
      // keep (about) three 16-KiB stages per wave-share in flight, as the GEMM ring does
      if (W == 64) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (W == 128) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0 && sink) sink[blockIdx.x] = *(unsigned long long*)smem;
}

template <int W>
double run(const char* A, size_t ld, int m_tiles, int kbytes, unsigned long long* sink, int grid) {
  hipFuncSetAttribute((const void*)stream_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(stream_kernel<W>, dim3(grid), dim3(512), 131072, 0, A, ld, m_tiles, kbytes, sink);
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(stream_kernel<W>, dim3(grid), dim3(512), 131072, 0, A, ld, m_tiles, kbytes, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 10;
}

int main() {
  const int M = 526336;
  for (int K : {1024, 4096}) {
    const size_t ld = (size_t)K * 2;
    char* A; hipMalloc(&A, (size_t)M * ld);
    hipMemset(A, 1, (size_t)M * ld);
    unsigned long long* sink; hipMalloc(&sink, 256 * 8);
    const int m_tiles = M / 256;
    const double gb = (double)M * ld / 1e9;
    double t64 = run<64>(A, ld, m_tiles, (int)ld, sink, 256);
    double t128 = run<128>(A, ld, m_tiles, (int)ld, sink, 256);
    double t256 = run<256>(A, ld, m_tiles, (int)ld, sink, 256);
    printf("K=%d (%.2f GB, each byte once): 64 B/row pieces %.3f ms = %.2f TB/s | 128 B/row %.3f ms = %.2f TB/s | 256 B/row %.3f ms = %.2f TB/s\n",
           K, gb, t64, gb / t64, t128, gb / t128, t256, gb / t256);
    hipFree(A); hipFree(sink);
  }
  return 0;
}
