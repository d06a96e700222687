// Probe: how much VALU work fits beside MFMAs on one gfx950 SIMD -- inside one wave (interleaved stream) and across the
// two waves that share a SIMD.  One workgroup of 8 waves per CU (wave w runs on SIMD w % 4); waves 0-3 run role A, waves 4-7
// role B; every wave times its own loop with s_memtime (shader cycles).  Printed: cycles per loop body, per role.
//   hipcc -O3 --offload-arch=gfx950 -o tools/probes/issue_overlap_probe.bin tools/probes/issue_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

enum Role { IDLE = 0, M32 = 1, EXP = 2, FMA = 3, M32_EXP = 4, M32_FMA = 5, M16 = 6, M16_EXP = 7, M16_FMA = 8, M32DEP = 9, M32DEP_EXP = 10,
            PKFMA = 11, M32_PKFMA = 12, CVT = 13, MAX3 = 14, M32_EXP4 = 15 };

#define MFMA32(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA16(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VEXP(d, s) asm volatile("v_exp_f32 %0, %1" : "=v"(d) : "v"(s))
#define VFMA(d, s) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(d) : "v"(s))
#define VPKFMA(d, s) asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(d) : "v"(s))
#define VCVT(d, s) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(d) : "v"(s))
#define VMAX3(d, s) asm volatile("v_max3_f32 %0, %1, %1, %1" : "=v"(d) : "v"(s))

template <int ROLE>
__device__ __forceinline__ void body(f32x16_t (&acc)[8], f32x4_t (&acc4)[8], bf16x8_t a, bf16x8_t b, float (&x)[16], double& px) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  f32x2_t p2 = {x[0], x[1]}, q2;
  if constexpr (ROLE == M32) {                         // 8 independent 32x32x16: 8 x 32 cycles of matrix pipe
#pragma unroll
    for (int i = 0; i < 8; ++i) MFMA32(acc[i]);
  } else if constexpr (ROLE == M32DEP) {               // one accumulator chain
#pragma unroll
    for (int i = 0; i < 8; ++i) MFMA32(acc[0]);
  } else if constexpr (ROLE == M16) {                  // 16 independent-ish 16x16x32: 16 x 16 cycles
#pragma unroll
    for (int i = 0; i < 16; ++i) MFMA16(acc4[i & 7]);
  } else if constexpr (ROLE == EXP) {                  // 16 x 16 cycles
#pragma unroll
    for (int i = 0; i < 16; ++i) VEXP(x[i], x[i]);
  } else if constexpr (ROLE == FMA) {                  // 64 x 4 cycles
#pragma unroll
    for (int i = 0; i < 64; ++i) VFMA(x[i & 15], x[i & 15]);
  } else if constexpr (ROLE == PKFMA) {                // 64 packed
#pragma unroll
    for (int i = 0; i < 64; ++i) { VPKFMA(q2, p2); }
    x[0] = q2[0];
  } else if constexpr (ROLE == CVT) {
#pragma unroll
    for (int i = 0; i < 64; ++i) VCVT(x[i & 15], x[i & 15]);
  } else if constexpr (ROLE == MAX3) {
#pragma unroll
    for (int i = 0; i < 64; ++i) VMAX3(x[i & 15], x[i & 15]);
  } else if constexpr (ROLE == M32_EXP) {              // 1 MFMA + 2 exp, x 8
#pragma unroll
    for (int i = 0; i < 8; ++i) { MFMA32(acc[i]); VEXP(x[2 * i], x[2 * i]); VEXP(x[2 * i + 1], x[2 * i + 1]); }
  } else if constexpr (ROLE == M32_EXP4) {             // 1 MFMA + 4 exp, x 8: exp-bound if they overlap (512), 768 if not
#pragma unroll
    for (int i = 0; i < 8; ++i) { MFMA32(acc[i]); VEXP(x[2 * i], x[2 * i]); VEXP(x[2 * i + 1], x[2 * i + 1]); VEXP(x[2 * i], x[2 * i]); VEXP(x[2 * i + 1], x[2 * i + 1]); }
  } else if constexpr (ROLE == M32DEP_EXP) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { MFMA32(acc[0]); VEXP(x[2 * i], x[2 * i]); VEXP(x[2 * i + 1], x[2 * i + 1]); }
  } else if constexpr (ROLE == M32_FMA) {              // 1 MFMA + 8 fma, x 8
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      MFMA32(acc[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) VFMA(x[(8 * i + j) & 15], x[(8 * i + j) & 15]);
    }
  } else if constexpr (ROLE == M32_PKFMA) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      MFMA32(acc[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) VPKFMA(q2, p2);
    }
    x[0] = q2[0];
  } else if constexpr (ROLE == M16_EXP) {              // 1 MFMA16 + 1 exp, x 16
#pragma unroll
    for (int i = 0; i < 16; ++i) { MFMA16(acc4[i & 7]); VEXP(x[i], x[i]); }
  } else if constexpr (ROLE == M16_FMA) {              // 1 MFMA16 + 4 fma, x 16
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      MFMA16(acc4[i & 7]);
#pragma unroll
      for (int j = 0; j < 4; ++j) VFMA(x[(4 * i + j) & 15], x[(4 * i + j) & 15]);
    }
  }
}

template <int RA, int RB>
__global__ __launch_bounds__(512, 2) void probe(unsigned long long* out, int iters) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x16_t acc[8]; f32x4_t acc4[8];
  for (int i = 0; i < 8; ++i) { for (int e = 0; e < 16; ++e) acc[i][e] = 0.f; for (int e = 0; e < 4; ++e) acc4[i][e] = 0.f; }
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * lane); b[e] = (__bf16)(0.002f * e); }
  float x[16]; for (int i = 0; i < 16; ++i) x[i] = -0.01f * (lane + i);
  double px = 0;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) { if (RA != IDLE) for (int it = 0; it < iters; ++it) body<RA>(acc, acc4, a, b, x, px); }
  else          { if (RB != IDLE) for (int it = 0; it < iters; ++it) body<RB>(acc, acc4, a, b, x, px); }
  asm volatile("s_nop 7\n s_nop 7" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sink = x[0];
  for (int i = 0; i < 8; ++i) sink += acc[i][0] + acc4[i][0];
  if (lane == 0) out[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
  if (sink == 12345.678f) out[0] = 0;
}

template <int RA, int RB>
void run(const char* name, unsigned long long* d_out, int iters, double expectA, double expectB) {
  const int G = 256;
  hipLaunchKernelGGL((probe<RA, RB>), dim3(G), dim3(512), 0, 0, d_out, iters);
  hipLaunchKernelGGL((probe<RA, RB>), dim3(G), dim3(512), 0, 0, d_out, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(G * 8);
  hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
  double sa = 0, sb = 0;
  for (int g = 0; g < G; ++g) for (int w = 0; w < 8; ++w) (w < 4 ? sa : sb) += (double)h[g * 8 + w];
  sa /= G * 4.0 * iters; sb /= G * 4.0 * iters;
  printf("%-34s  A %8.1f cycles/body (alone %5.0f)   B %8.1f (alone %5.0f)\n", name, sa, expectA, sb, expectB);
}

int main() {
  unsigned long long* d_out; hipMalloc(&d_out, 256 * 8 * 8);
  const int it = 2000;
  run<M32, IDLE>("A: 8 mfma32 indep", d_out, it, 256, 0);
  run<M32DEP, IDLE>("A: 8 mfma32 one chain", d_out, it, 256, 0);
  run<M16, IDLE>("A: 16 mfma16", d_out, it, 256, 0);
  run<EXP, IDLE>("A: 16 exp", d_out, it, 256, 0);
  run<FMA, IDLE>("A: 64 fma", d_out, it, 256, 0);
  run<PKFMA, IDLE>("A: 64 pk_fma", d_out, it, 256, 0);
  run<CVT, IDLE>("A: 64 cvt_pk_bf16", d_out, it, 256, 0);
  run<MAX3, IDLE>("A: 64 max3", d_out, it, 256, 0);
  run<M32_EXP, IDLE>("A: 8 x (mfma32 + 2 exp) one wave", d_out, it, 256, 0);
  run<M32_EXP4, IDLE>("A: 8 x (mfma32 + 4 exp) one wave", d_out, it, 512, 0);
  run<M32DEP_EXP, IDLE>("A: 8 x (mfma32 chain + 2 exp)", d_out, it, 256, 0);
  run<M32_FMA, IDLE>("A: 8 x (mfma32 + 8 fma) one wave", d_out, it, 256, 0);
  run<M32_PKFMA, IDLE>("A: 8 x (mfma32 + 8 pk_fma) one wave", d_out, it, 256, 0);
  run<M16_EXP, IDLE>("A: 16 x (mfma16 + exp) one wave", d_out, it, 256, 0);
  run<M16_FMA, IDLE>("A: 16 x (mfma16 + 4 fma) one wave", d_out, it, 256, 0);
  run<M32, EXP>("A mfma32 | B exp (two waves)", d_out, it, 256, 256);
  run<M32, FMA>("A mfma32 | B fma (two waves)", d_out, it, 256, 256);
  run<M32, PKFMA>("A mfma32 | B pk_fma (two waves)", d_out, it, 256, 256);
  run<M16, EXP>("A mfma16 | B exp (two waves)", d_out, it, 256, 256);
  run<M16, FMA>("A mfma16 | B fma (two waves)", d_out, it, 256, 256);
  run<M32, M32>("A mfma32 | B mfma32", d_out, it, 256, 256);
  run<EXP, EXP>("A exp | B exp", d_out, it, 256, 256);
  run<EXP, FMA>("A exp | B fma", d_out, it, 256, 256);
  run<M32_EXP, M32_EXP>("A,B: 8 x (mfma32 + 2 exp)", d_out, it, 256, 256);
  run<M32_EXP4, M32_EXP4>("A,B: 8 x (mfma32 + 4 exp)", d_out, it, 512, 512);
  hipFree(d_out);
  return 0;
}
