// Probe 4 (exploratory): which output entries does lane L0's scale register touch, and which of its bytes is used?
// All data = 1.0, all scales = 1.0 except lane L0 of ONE operand, whose bytes are 2^1, 2^2, 2^3, 2^4.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int OPSEL>
__global__ void k(const uint32_t* SA, const uint32_t* SB, float* C) {
  int l = threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPSEL, (int)SA[l], OPSEL, (int)SB[l]);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
int main() {
  uint32_t hS[64], hU[64]; float hC[1024];
  uint32_t *dS, *dU; float* dC;
  (void)hipMalloc(&dS, 256); (void)hipMalloc(&dU, 256); (void)hipMalloc(&dC, 4096);
  for (int l = 0; l < 64; ++l) hU[l] = 0x7f7f7f7f;
  (void)hipMemcpy(dU, hU, 256, hipMemcpyHostToDevice);
  for (int which = 0; which < 2; ++which)
    for (int L0 : {0, 5, 37}) {
      for (int l = 0; l < 64; ++l) hS[l] = 0x7f7f7f7f;
      hS[L0] = 0x83828180u;      // byte0 = 2^1, byte1 = 2^2, byte2 = 2^3, byte3 = 2^4
      (void)hipMemcpy(dS, hS, 256, hipMemcpyHostToDevice);
      for (int opsel = 0; opsel < 4; ++opsel) {
        const uint32_t* sa = which == 0 ? dS : dU; const uint32_t* sb = which == 0 ? dU : dS;
        if (opsel == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, sa, sb, dC);
        if (opsel == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, sa, sb, dC);
        if (opsel == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, sa, sb, dC);
        if (opsel == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, sa, sb, dC);
        (void)hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
        int nrow = 0, ncol = 0, r0 = -1, c0 = -1; float val = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (hC[i * 32 + j] != 64.f) { val = hC[i * 32 + j]; if (r0 < 0) { r0 = i; c0 = j; } }
        for (int i = 0; i < 32; ++i) { int any = 0; for (int j = 0; j < 32; ++j) any |= hC[i * 32 + j] != 64.f; nrow += any; }
        for (int j = 0; j < 32; ++j) { int any = 0; for (int i = 0; i < 32; ++i) any |= hC[i * 32 + j] != 64.f; ncol += any; }
        printf("%s scale, lane %2d, opsel %d: %2d rows x %2d cols differ from 64, first at (%d,%d) = %g  -> factor on one k-block %g\n",
               which == 0 ? "first-operand " : "second-operand", L0, opsel, nrow, ncol, r0, c0, val, (val - 32.f) / 32.f);
      }
    }
  return 0;
}
