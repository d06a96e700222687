// Does MODE.FP16_OVFL (bit 23 of the MODE register) make v_cvt_pk_fp8_f32 SATURATE instead of producing NaN on gfx950?
// (By default 449 -> 448 but 1000 -> NaN: every static-scale quantiser of this repo clamps with v_med3_f32 first.)
//   hipcc --offload-arch=gfx950 -O2 fp8_ovfl_probe.hip -o fp8_ovfl_probe.bin && ./fp8_ovfl_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) short s16x2_t;
__global__ void probe(const float* in, unsigned* out, int n, int ovfl) {
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
  const int i = threadIdx.x;
  if (i < n) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], -in[i], w, false);
    s16x2_t q = {0, 0};
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(q, in[i], -in[i], 1.0f, false);
    out[i] = (unsigned)(w & 0xffff) | ((unsigned)(unsigned short)q[0] << 16);
  }
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
}
int main() {
  const float h[8] = {1.0f, 447.0f, 449.0f, 470.0f, 1000.0f, 1e9f, __builtin_inff(), __builtin_nanf("")};
  float* d; unsigned* o; unsigned r[8];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, 8, ovfl);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("FP16_OVFL=%d:", ovfl);
    for (int i = 0; i < 8; ++i) printf("  %g -> cvt_pk %02x %02x  scalef32 %02x %02x", h[i], r[i] & 0xff, (r[i] >> 8) & 0xff, (r[i] >> 16) & 0xff, (r[i] >> 24) & 0xff);
    printf("\n");
  }
  return 0;
}
