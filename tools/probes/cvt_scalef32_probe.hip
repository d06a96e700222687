// Probe: operand semantics of v_cvt_scalef32_pk_fp8_f32 on gfx950 (scale direction, what part of the scale operand counts,
// saturation) next to v_cvt_pk_fp8_f32.   hipcc --offload-arch=gfx950 -O2 cvt_scalef32_probe.hip -o cvt_scalef32_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(2))) short s16x2_t;

__global__ void probe(const float* x, const float* scale, int nx, int ns, unsigned* out_scaled, unsigned* out_plain, float* back) {
  const int i = threadIdx.x;
  if (i >= nx * ns) return;
  const float v = x[i % nx], s = scale[i / nx];
  s16x2_t old = {0, 0};
  const s16x2_t r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, v, -v, s, false);
  out_scaled[i] = (unsigned)(unsigned short)r[0];
  int wd = 0;
  wd = __builtin_amdgcn_cvt_pk_fp8_f32(v, -v, wd, false);
  out_plain[i] = (unsigned)wd & 0xffffu;
  back[i] = __builtin_amdgcn_cvt_f32_fp8((int)(unsigned short)r[0], 0);
}

int main() {
  const std::vector<float> xs = {0.f, 0.3f, 1.0f, 3.3f, 17.f, 100.f, 447.f, 449.f, 470.f, 1000.f, 1e6f, INFINITY, NAN, 0.001f, 0.0019f, 1e-5f};
  const std::vector<float> ss = {1.f, 2.f, 0.5f, 3.f, 1.5f, 4.f, 0.25f, 1024.f};
  const int nx = (int)xs.size(), ns = (int)ss.size();
  float *dx, *dscale, *dback; unsigned *d1, *d2;
  hipMalloc(&dx, nx * 4); hipMalloc(&dscale, ns * 4); hipMalloc(&d1, nx * ns * 4); hipMalloc(&d2, nx * ns * 4); hipMalloc(&dback, nx * ns * 4);
  hipMemcpy(dx, xs.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(dscale, ss.data(), ns * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, dx, dscale, nx, ns, d1, d2, dback);
  std::vector<unsigned> a(nx * ns), b(nx * ns); std::vector<float> f(nx * ns);
  hipMemcpy(a.data(), d1, nx * ns * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d2, nx * ns * 4, hipMemcpyDeviceToHost);
  hipMemcpy(f.data(), dback, nx * ns * 4, hipMemcpyDeviceToHost);
  for (int j = 0; j < ns; ++j) {
    printf("scale %g:\n", ss[j]);
    for (int i = 0; i < nx; ++i)
      printf("  x %-10g scaled bytes (x, -x) %02x %02x = %-8g   plain %02x %02x\n", xs[i], a[j * nx + i] & 0xff, (a[j * nx + i] >> 8) & 0xff, f[j * nx + i],
             b[j * nx + i] & 0xff, (b[j * nx + i] >> 8) & 0xff);
  }
  return 0;
}
