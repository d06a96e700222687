// Probe 2: v_mfma_scale_f32_32x32x64_f8f6f4 operand / result maps with unit scales, and v_cvt_pk_fp8_f32.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(const uint8_t* A, const uint8_t* B, float* C, const float* f, uint32_t* q) {
  int l = threadIdx.x;
  // assumed: lane l holds row l&31, bytes [32*(l>>5), +32) of a 64-byte (K = 64) row
  i32x8 a = *(const i32x8*)(A + (l & 31) * 64 + (l >> 5) * 32);
  i32x8 b = *(const i32x8*)(B + (l & 31) * 64 + (l >> 5) * 32);
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  // assumed D map: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
  // fp8 conversion of 2 floats per call into the low / high half-word
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(f[l * 4 + 0], f[l * 4 + 1], w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(f[l * 4 + 2], f[l * 4 + 3], w, true);
  q[l] = (uint32_t)w;
}
static float e4m3_to_f(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  if (e == 15 && m == 7) return NAN;
  float f = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1 + m / 8.0f, e - 7);
  return s ? -f : f;
}
int main() {
  uint8_t hA[32 * 64], hB[32 * 64];
  srand(2);
  for (int i = 0; i < 32 * 64; ++i) { hA[i] = rand() % 0x78 | ((rand() & 1) << 7); hB[i] = rand() % 0x78 | ((rand() & 1) << 7); }
  float hf[256]; for (int i = 0; i < 256; ++i) hf[i] = (i % 2 ? -1.f : 1.f) * ldexpf(1.0f + (i % 37) / 37.0f, (i % 17) - 8);
  hf[0] = 448.f; hf[1] = 449.f; hf[2] = 1000.f; hf[3] = 0.001f; hf[4] = 464.f; hf[5] = -480.f;
  uint8_t *dA, *dB; float *dC, *df; uint32_t* dq; float hC[1024]; uint32_t hq[64];
  (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dC, sizeof hC); (void)hipMalloc(&df, sizeof hf); (void)hipMalloc(&dq, sizeof hq);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  (void)hipMemcpy(df, hf, sizeof hf, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, df, dq);
  (void)hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost); (void)hipMemcpy(hq, dq, sizeof hq, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double ref = 0; for (int kk = 0; kk < 64; ++kk) ref += (double)e4m3_to_f(hA[i * 64 + kk]) * e4m3_to_f(hB[j * 64 + kk]);
    maxerr = fmax(maxerr, fabs(hC[i * 32 + j] - ref)); maxref = fmax(maxref, fabs(ref));
  }
  printf("32x32x64 scaled mfma: max |err| %.4g  max |ref| %.4g\n", maxerr, maxref);
  printf("cvt_pk_fp8: 448 -> %02x (%.1f)  449 -> %02x (%.1f)  1000 -> %02x (%.1f)  0.001 -> %02x (%g)  464 -> %02x (%.1f)  -480 -> %02x (%.1f)\n",
         hq[0] & 255, e4m3_to_f(hq[0] & 255), (hq[0] >> 8) & 255, e4m3_to_f((hq[0] >> 8) & 255), (hq[0] >> 16) & 255, e4m3_to_f((hq[0] >> 16) & 255),
         hq[0] >> 24, e4m3_to_f(hq[0] >> 24), hq[1] & 255, e4m3_to_f(hq[1] & 255), (hq[1] >> 8) & 255, e4m3_to_f((hq[1] >> 8) & 255));
  double worst = 0;
  for (int l = 2; l < 64; ++l) for (int j = 0; j < 4; ++j) {
    float x = hf[l * 4 + j], y = e4m3_to_f((hq[l] >> (8 * j)) & 255);
    if (fabsf(x) >= 0.015625f && fabsf(x) <= 448.f) worst = fmax(worst, fabs(y - x) / fabs(x));
  }
  printf("cvt_pk_fp8 worst relative rounding error on normal-range inputs: %.4f (RNE e4m3 bound 0.0625)\n", worst);
  return 0;
}
