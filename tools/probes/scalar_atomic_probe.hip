// Probe: does gfx950 execute scalar memory atomics (s_atomic_add with return, counted in lgkmcnt)?  A persistent kernel that takes
// its next tile from a global ticket counter needs the ticket WITHOUT a vector-memory operation (the GEMM's DMA pipeline is paced by
// hand-counted vmcnt waits).  2048 workgroups x 4 waves x 4 tickets (scalar instructions execute once per WAVE): every ticket
// 0 .. 32767 must come back exactly once, from workgroups on all eight XCDs (each XCD has its own L2: the read-modify-write must
// happen at one point of coherence).
//   hipcc -O2 --offload-arch=gfx950 -o tools/probes/scalar_atomic_probe.bin tools/probes/scalar_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(unsigned* c, unsigned* out) {
  for (int i = 0; i < 4; ++i) {
    unsigned v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(c) : "memory");
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + i] = v;
  }
}
int main() {
  unsigned *c, *out;
  const int n = 2048 * 4 * 4;
  hipMalloc(&c, 4); hipMalloc(&out, n * 4);
  hipMemset(c, 0, 4);
  hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, c, out);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
  std::vector<unsigned> h(n); unsigned fin = 0;
  hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost); hipMemcpy(&fin, c, 4, hipMemcpyDeviceToHost);
  std::vector<int> seen(n, 0); int bad = 0;
  for (unsigned v : h) { if (v >= (unsigned)n) ++bad; else ++seen[v]; }
  for (int s : seen) if (s != 1) ++bad;
  printf("final counter %u (want %d), tickets not seen exactly once: %d\n", fin, n, bad);
  return bad || fin != (unsigned)n;
}
