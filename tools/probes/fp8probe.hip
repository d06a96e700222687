#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// Probe: v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit (2^0) block scales.
// Assumed operand map (to be verified): lane l holds A[row l&15][k = 32*(l>>4) + 0..31] as 32 consecutive bytes.
__global__ void k(const uint8_t* A, const uint8_t* B, float* C) {
  int l = threadIdx.x;
  i32x8 a = *(const i32x8*)(A + (l & 15) * 128 + (l >> 4) * 32);
  i32x8 b = *(const i32x8*)(B + (l & 15) * 128 + (l >> 4) * 32);
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
static float e4m3_to_f(uint8_t v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf(m / 8.0f, -6) : ldexpf(1 + m / 8.0f, e - 7);
  return s ? -f : f;
}
int main() {
  uint8_t hA[16 * 128], hB[16 * 128];
  srand(1);
  for (int i = 0; i < 16 * 128; ++i) { hA[i] = rand() % 0x78; if (rand() & 1) hA[i] |= 0x80; hB[i] = rand() % 0x78; if (rand() & 1) hB[i] |= 0x80; }
  uint8_t *dA, *dB; float* dC; float hC[256];
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double ref = 0; for (int kk = 0; kk < 128; ++kk) ref += (double)e4m3_to_f(hA[i * 128 + kk]) * e4m3_to_f(hB[j * 128 + kk]);
    // D[row][col]: row from A operand rows, col from B operand rows (B^T input)
    double got = hC[i * 16 + j];
    maxerr = fmax(maxerr, fabs(got - ref)); maxref = fmax(maxref, fabs(ref));
  }
  printf("fp8 scaled mfma probe: max |err| %.4g  max |ref| %.4g  C[0][1]=%.3f C[1][0]=%.3f\n", maxerr, maxref, hC[1], hC[16]);
  return 0;
}
