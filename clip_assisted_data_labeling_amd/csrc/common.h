// Shared device/host helpers for the CLIP-embed HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                             // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;          // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;            // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;          // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {              // round-to-nearest-even, NaN kept
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  // one v_cvt_pk_bf16_f32 (RNE, as the scalar cast): two scalar casts + shift/or compiled to five VALU instructions per pair
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// host-side bf16 rounding identical to the device cast (RNE)
static inline uint16_t host_f32_to_bf16(float f) {
  uint32_t u; __builtin_memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float host_bf16_to_f32(uint16_t h) {
  uint32_t u = ((uint32_t)h) << 16; float f; __builtin_memcpy(&f, &u, 4); return f;
}

#define CE_ACT_QUICK_GELU 0
#define CE_ACT_GELU_ERF 1

#ifdef __HIPCC__
// erf-GELU, u Phi(u) with Phi(u) = 1/2 erfc(-u / sqrt 2), for the GEMM epilogues (open_clip's laion / datacomp towers, ViT-H-14).
// erfc(z), z >= 0, by Abramowitz & Stegun 7.1.26:  t = 1 / (1 + p z),  erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2),
// |error| <= 1.5e-7 -- three decimal digits under the bf16 / e4m3 rounding of the result.  With q = erfc(|u| / sqrt 2):
// GELU(u) = u - u q / 2 for u >= 0 and u q / 2 for u < 0, i.e. max(u, 0) - |u q / 2|.  16 issue slots (two of them transcendental)
// where libdevice's erff is a branchy ~45: the erf-GELU FC1 epilogue was 22 % slower than the QuickGELU one.
__device__ __forceinline__ float ce_gelu_erf(float u) {
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_fabsf(u), 0.3275911f * 0.70710678118654752f, 1.0f));
  float pl = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  pl = __builtin_fmaf(pl, t, 0.5f * 1.421413741f);
  pl = __builtin_fmaf(pl, t, 0.5f * -0.284496736f);
  pl = __builtin_fmaf(pl, t, 0.5f * 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(u * u * (-0.5f * 1.44269504088896340736f));     // exp(-u^2 / 2)
  const float h = u * (pl * t * e);                                                       // u q / 2
  return __builtin_fmaxf(u, 0.0f) - __builtin_fabsf(h);
}
// four elements stage by stage (4 x rcp, 4 x exp2 apart from their consumers: element by element hipcc issues four dependent chains with a
// wait state behind every transcendental)
typedef __attribute__((ext_vector_type(4))) float ce_f32x4_t;
__device__ __forceinline__ ce_f32x4_t ce_gelu_erf4(ce_f32x4_t u) {
  ce_f32x4_t t, e, pl, r;
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = __builtin_fmaf(__builtin_fabsf(u[i]), 0.3275911f * 0.70710678118654752f, 1.0f);
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_rcpf(t[i]);
  e = u * u * (-0.5f * 1.44269504088896340736f);
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
  pl = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
  pl = pl * t + (0.5f * 1.421413741f);
  pl = pl * t + (0.5f * -0.284496736f);
  pl = pl * t + (0.5f * 0.254829592f);
  const ce_f32x4_t h = u * (pl * t * e);
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaxf(u[i], 0.0f) - __builtin_fabsf(h[i]);
  return r;
}
#endif

// One-time launch setup of a kernel PER DEVICE: opts it into `lds_bytes` of dynamic LDS on the current device and returns
// that device's CU count.  The C ABI takes a device per handle, so one process may launch the same kernel on several
// GPUs: a single function-local `static bool` would set the attribute on the first device only and reuse its grid size.
// Thread-safe: plain atomics on a small table, the (idempotent) setup may run twice under a race.
#include <atomic>
struct DeviceKernelSetup {
  static constexpr int kMaxDevices = 64;
  std::atomic<int> cu[kMaxDevices];
  hipError_t ensure(const void* func, int lds_bytes, int* n_cu) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    int c = cu[dev].load(std::memory_order_acquire);
    if (c == 0) {
      if (lds_bytes > 0 && (e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)) != hipSuccess) return e;
      if ((e = hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
      if (c <= 0) c = 256;
      cu[dev].store(c, std::memory_order_release);
    }
    if (n_cu) *n_cu = c;
    return hipSuccess;
  }
};
