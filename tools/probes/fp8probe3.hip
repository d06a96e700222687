// Probe 3: per-lane E8M0 block scales of v_mfma_scale_f32_32x32x64_f8f6f4.
// Hypothesis: lane l supplies the scale of ITS OWN 32 k-elements (row l&31, k-block l>>5) in byte `opsel` of the scale
// VGPR, for the first operand (scale_a) and the second (scale_b) independently, and the product term is
// a * 2^(sa-127) * b * 2^(sb-127).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int OPSEL>
__global__ void k(const uint8_t* A, const uint8_t* B, const uint32_t* SA, const uint32_t* SB, float* C) {
  int l = threadIdx.x;
  i32x8 a = *(const i32x8*)(A + (l & 31) * 64 + (l >> 5) * 32);
  i32x8 b = *(const i32x8*)(B + (l & 31) * 64 + (l >> 5) * 32);
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPSEL, (int)SA[l], OPSEL, (int)SB[l]);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
static float e4m3_to_f(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1 + m / 8.0f, e - 7);
  return s ? -f : f;
}
int main() {
  uint8_t hA[32 * 64], hB[32 * 64];
  uint32_t hSA[64], hSB[64];
  srand(3);
  for (int i = 0; i < 32 * 64; ++i) { hA[i] = rand() % 0x70 | ((rand() & 1) << 7); hB[i] = rand() % 0x70 | ((rand() & 1) << 7); }
  for (int l = 0; l < 64; ++l) {   // 4 candidate bytes per lane; byte `opsel` is the one that should be used
    hSA[l] = 0; hSB[l] = 0;
    for (int b = 0; b < 4; ++b) { hSA[l] |= (uint32_t)(120 + rand() % 14) << (8 * b); hSB[l] |= (uint32_t)(122 + rand() % 10) << (8 * b); }
  }
  uint8_t *dA, *dB; uint32_t *dSA, *dSB; float* dC; float hC[1024];
  (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dSA, sizeof hSA); (void)hipMalloc(&dSB, sizeof hSB); (void)hipMalloc(&dC, sizeof hC);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  (void)hipMemcpy(dSA, hSA, sizeof hSA, hipMemcpyHostToDevice); (void)hipMemcpy(dSB, hSB, sizeof hSB, hipMemcpyHostToDevice);
  for (int opsel = 0; opsel < 4; ++opsel) {
    if (opsel == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
    if (opsel == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
    if (opsel == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
    if (opsel == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dC);
    (void)hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    // hypotheses about which byte is used: [h0] byte opsel of the lane's own register
    double err[4] = {0, 0, 0, 0}, maxref = 0;
    for (int hyp = 0; hyp < 4; ++hyp)
      for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int kb = 0; kb < 2; ++kb) {
          const int la = i + 32 * kb, lb = j + 32 * kb;       // lanes that hold row i / j, k-block kb
          const int sa = (hSA[la] >> (8 * ((opsel + hyp) & 3))) & 255, sb = (hSB[lb] >> (8 * ((opsel + hyp) & 3))) & 255;
          double part = 0;
          for (int kk = 0; kk < 32; ++kk) part += (double)e4m3_to_f(hA[i * 64 + kb * 32 + kk]) * e4m3_to_f(hB[j * 64 + kb * 32 + kk]);
          ref += part * ldexp(1.0, sa - 127) * ldexp(1.0, sb - 127);
        }
        err[hyp] = fmax(err[hyp], fabs(hC[i * 32 + j] - ref)); maxref = fmax(maxref, fabs(ref));
      }
    printf("opsel %d: max |err| if the byte used is opsel+0/+1/+2/+3: %.4g %.4g %.4g %.4g   (max |ref| %.4g)\n", opsel, err[0], err[1], err[2], err[3], maxref);
  }
  return 0;
}
