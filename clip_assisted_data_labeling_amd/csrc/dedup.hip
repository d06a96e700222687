// Near-duplicate search (K11 of SURVEY.md §2.2): row normalisation of the float16 embeddings
// (/root/reference/_2_remove_duplicates.py:67) into a zero-padded [n_pad][k_pad] operand for the
// thresholded f16 MFMA GEMM (gemm_bf16.hip, EPI_THRESH; :69-80).  The N x N similarity matrix is
// never materialised.
#include "common.h"
#include "gemm.h"
#include "kernels.h"

namespace {

// one wave per row; float16 in, float16 out.  The reference normalises IN float16:
// norm = half(sqrt(sum x^2)), e_hat = half(x / norm)  — reproduced with fp32 intermediates.
__global__ __launch_bounds__(256) void dedup_normalize_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out,
                                                              int n, int d, int n_pad, int ld_out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_pad) return;
  _Float16* o = out + (size_t)row * ld_out;
  if (row >= n) {
    for (int k = lane; k < ld_out; k += 64) o[k] = (_Float16)0.f;
    return;
  }
  const _Float16* x = in + (size_t)row * d;
  float ss = 0.f;
  for (int k = lane; k < d; k += 64) { const float v = (float)x[k]; ss += v * v; }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) ss += __shfl_xor(ss, s);
  const float nrm = (float)(_Float16)sqrtf(ss);
  for (int k = lane; k < ld_out; k += 64) o[k] = k < d ? (_Float16)((float)x[k] / nrm) : (_Float16)0.f;
}

// ---- the screened search: e4m3 screen at twice the f16 MFMA rate, exact values only for the candidates ----
// q_i = e4m3(256 e_hat_i) (a power-of-two scale: |256 e_hat| <= 256.2 < 448, elements below 2^-6 / 256 fall into e4m3's subnormals:
// absolute error <= 2^-10 / 256 each).  With d_i = e_hat_i - q_i / 256 and err_i = |d_i|_2, measured here on the row itself,
//   | e_hat_i . e_hat_j  -  q_i . q_j / 65536 |  =  | (q_i/256) . d_j + d_i . (q_j/256) + d_i . d_j |
//                                               <=  |q_i/256| err_j + err_i |q_j/256| + err_i err_j  <=  1.1 (err_i + err_j)
// (Cauchy-Schwarz; |q/256| <= |e_hat| + err <= 1.07 and err <= 2^-4 |e_hat| because e4m3 rounds every element by at most 2^-4
// of itself).  The screen (gemm_fp8.hip, EPI 4) keeps (i, j) when  acc + m_i + m_j > 65536 thr_lo  with the per-row margin
//   m_i = 65536 (1.1 (1 + 2^-10) err_i + 1e-4)     [the 2^-10 covers this kernel's own fp32 rounding, the 1e-4 the MFMA's
//                                                     accumulation error -- 4e-6 of sum |a||w| <= 1, DESIGN section 3.6]
// and thr_lo = the smallest true cosine the exact rule can accept (its threshold minus the fp16 rounding of the value).  No pair
// that the exact search reports can fail that test: the screen only ever adds work.  err is ~0.026 on unit vectors, so the
// screen passes what lies above thr - 0.06.
__global__ __launch_bounds__(256) void dedup_quant_fp8_kernel(const _Float16* __restrict__ ehat, int ld, unsigned char* __restrict__ q8,
                                                              int ld8, float* __restrict__ margin, int n_pad,
                                                              unsigned long long* overflow, unsigned long long over_by) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_pad) return;
  const _Float16* x = ehat + (size_t)row * ld;
  unsigned* o = (unsigned*)(q8 + (size_t)row * ld8);
  float ee = 0.f, qq = 0.f;
  bool finite = true;
  for (int k4 = lane; k4 < ld8 / 4; k4 += 64) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = 4 * k4 + e < ld ? (float)x[4 * k4 + e] : 0.f;
      finite = finite && fabsf(v[e]) <= 65504.f;               // (false for NaN)
    }
    int wd = 0;
    wd = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[0] * 256.f, -448.f, 448.f), __builtin_amdgcn_fmed3f(v[1] * 256.f, -448.f, 448.f), wd, false);
    wd = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v[2] * 256.f, -448.f, 448.f), __builtin_amdgcn_fmed3f(v[3] * 256.f, -448.f, 448.f), wd, true);
    o[k4] = (unsigned)wd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float q = (e == 0 ? __builtin_amdgcn_cvt_f32_fp8(wd, 0) : e == 1 ? __builtin_amdgcn_cvt_f32_fp8(wd, 1)
                       : e == 2 ? __builtin_amdgcn_cvt_f32_fp8(wd, 2) : __builtin_amdgcn_cvt_f32_fp8(wd, 3)) * (1.f / 256.f);
      const float d = v[e] - q;
      ee += d * d; qq += q * q;
    }
  }
  finite = __builtin_amdgcn_ballot_w64(!finite) == 0ull;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) { ee += __shfl_xor(ee, s); qq += __shfl_xor(qq, s); }
  if (!finite) {
    // a row with a NaN (a zero embedding divided by its zero norm) or an infinity is no one's duplicate under the exact rule
    // (NaN > threshold is false): it enters the screen as a zero row, which is no one's candidate
    for (int k4 = lane; k4 < ld8 / 4; k4 += 64) o[k4] = 0u;
    if (lane == 0) margin[row] = 0.f;
    return;
  }
  const float err = sqrtf(ee) * (1.f + 0x1p-10f), nq = sqrtf(qq) * (1.f + 0x1p-10f);
  if (lane == 0) {
    margin[row] = 65536.f * (1.1f * err + 1e-4f);
    // the 1.1 above needs |q / 256| <= 1.065 and err <= 0.0665, which every normalised row meets (|e_hat| = 1 to 1e-3 and e4m3
    // rounds by at most 2^-4 of each element); a row that does not (a norm in fp16's subnormals ...) sends the whole call to the
    // exact search: the candidate counter is put over its limit
    if (!(nq <= 1.065f && err <= 0.0665f)) atomicAdd(overflow, over_by);
  }
}

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// The exact value of every candidate, by the MFMA the exact search uses, in its K order (16 candidates per wave: candidate t's row
// i feeds A row t, its row j feeds B column t, the value is D[t][t]), then the exact search's own rule and append.
// Does nothing when the screen ran out of candidate slots: the exact search runs then (gemm.h: run_if_over).
__global__ __launch_bounds__(256) void dedup_recheck_kernel(const _Float16* __restrict__ ehat, int ld, const uint2* __restrict__ cand,
                                                            const unsigned long long* __restrict__ cand_count, unsigned long long cand_cap,
                                                            float thr_in, int fp16_compare, int n_valid, long long* __restrict__ pairs,
                                                            float* __restrict__ vals, unsigned long long cap, unsigned long long* count) {
  const unsigned long long total = *cand_count;
  if (total > cand_cap) return;
  const float thr = fp16_compare ? (float)(_Float16)thr_in : thr_in;
  const int lane = threadIdx.x & 63, t = lane & 15, kq = lane >> 4;
  const unsigned long long nwaves = (unsigned long long)gridDim.x * 4;
  for (unsigned long long g = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6); g * 16 < total; g += nwaves) {
    const unsigned long long c = g * 16 + t;
    const uint2 ij = cand[c < total ? c : g * 16];
    const _Float16* a = ehat + (size_t)ij.x * ld + 8 * kq;
    const _Float16* b = ehat + (size_t)ij.y * ld + 8 * kq;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < ld; k += 32)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const f16x8_t*)(a + k), *(const f16x8_t*)(b + k), acc, 0, 0, 0);
    // D[r][col]: lane (col = lane & 15, quad = lane >> 4) holds rows 4 quad + 0..3
    if (c < total && kq == (t >> 2)) {
      float v = acc[t & 3];
      if (fp16_compare) v = (float)(_Float16)v;
      const int i = (int)ij.x, j = (int)ij.y;
      if (j > i && j < n_valid && v > thr) {
        const unsigned long long slot = atomicAdd(count, 1ull);
        if (slot < cap) {
          pairs[slot * 2 + 0] = i;
          pairs[slot * 2 + 1] = j;
          vals[slot] = v;
        }
      }
    }
  }
}

// Normalisation (dedup_normalize_kernel's arithmetic) and the screen's quantisation (dedup_quant_fp8_kernel's) in one pass over the
// rows, 16-byte accesses: one wave per row, a lane takes 8 columns at a time (d % 8 == 0; other widths use the two kernels above).
// q8 == nullptr: normalise only.
typedef __attribute__((ext_vector_type(8))) _Float16 h16x8_t;
__global__ __launch_bounds__(256) void dedup_normalize_quant_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out, int n, int d,
                                                                    int n_pad, int ld, unsigned char* __restrict__ q8, int ld8,
                                                                    float* __restrict__ margin, unsigned long long* overflow,
                                                                    unsigned long long over_by) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_pad) return;
  h16x8_t* o = (h16x8_t*)(out + (size_t)row * ld);
  uint2* o8 = q8 ? (uint2*)(q8 + (size_t)row * ld8) : nullptr;
  const h16x8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const int dc = d / 8, lc = ld / 8, l8c = ld8 / 8;
  if (row >= n) {
    for (int c = lane; c < lc; c += 64) o[c] = zero8;
    if (q8) {
      for (int c = lane; c < l8c; c += 64) o8[c] = uint2{0u, 0u};
      if (lane == 0) margin[row] = 0.f;
    }
    return;
  }
  const h16x8_t* x = (const h16x8_t*)(in + (size_t)row * d);
  float ss = 0.f;
  for (int c = lane; c < dc; c += 64) {
    const h16x8_t v = x[c];
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += (float)v[e] * (float)v[e];
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) ss += __shfl_xor(ss, s);
  const float nrm = (float)(_Float16)sqrtf(ss);
  float ee = 0.f, qq = 0.f;
  bool finite = true;
  for (int c = lane; c < max(lc, l8c); c += 64) {
    h16x8_t r = zero8;
    if (c < dc) {
      const h16x8_t v = x[c];
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = (_Float16)((float)v[e] / nrm);
    }
    if (c < lc) o[c] = r;
    if (q8 && c < l8c) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        f[e] = (float)r[e];
        finite = finite && fabsf(f[e]) <= 65504.f;
      }
      int w0 = 0, w1 = 0;
#define Q_(a) __builtin_amdgcn_fmed3f((a) * 256.f, -448.f, 448.f)
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(Q_(f[0]), Q_(f[1]), w0, false); w0 = __builtin_amdgcn_cvt_pk_fp8_f32(Q_(f[2]), Q_(f[3]), w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(Q_(f[4]), Q_(f[5]), w1, false); w1 = __builtin_amdgcn_cvt_pk_fp8_f32(Q_(f[6]), Q_(f[7]), w1, true);
#undef Q_
      o8[c] = uint2{(unsigned)w0, (unsigned)w1};
      const float q[8] = {__builtin_amdgcn_cvt_f32_fp8(w0, 0), __builtin_amdgcn_cvt_f32_fp8(w0, 1), __builtin_amdgcn_cvt_f32_fp8(w0, 2),
                          __builtin_amdgcn_cvt_f32_fp8(w0, 3), __builtin_amdgcn_cvt_f32_fp8(w1, 0), __builtin_amdgcn_cvt_f32_fp8(w1, 1),
                          __builtin_amdgcn_cvt_f32_fp8(w1, 2), __builtin_amdgcn_cvt_f32_fp8(w1, 3)};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float qv = q[e] * (1.f / 256.f), dv = f[e] - qv;
        ee += dv * dv; qq += qv * qv;
      }
    }
  }
  if (!q8) return;
  finite = __builtin_amdgcn_ballot_w64(!finite) == 0ull;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) { ee += __shfl_xor(ee, s); qq += __shfl_xor(qq, s); }
  if (!finite) {                               // as dedup_quant_fp8_kernel: a zero row in the screen
    for (int c = lane; c < l8c; c += 64) o8[c] = uint2{0u, 0u};
    if (lane == 0) margin[row] = 0.f;
    return;
  }
  const float err = sqrtf(ee) * (1.f + 0x1p-10f), nq = sqrtf(qq) * (1.f + 0x1p-10f);
  if (lane == 0) {
    margin[row] = 65536.f * (1.1f * err + 1e-4f);
    if (!(nq <= 1.065f && err <= 0.0665f)) atomicAdd(overflow, over_by);
  }
}

}  // namespace

hipError_t ce_dedup_normalize_f16(const void* emb_f16, void* out_f16, int n, int d, int ld_out, hipStream_t stream) {
  const int n_pad = (n + 255) / 256 * 256;
  if (d % 8 == 0 && ld_out % 8 == 0 && !(((uintptr_t)emb_f16 | (uintptr_t)out_f16) & 15))
    hipLaunchKernelGGL(dedup_normalize_quant_kernel, dim3((n_pad + 3) / 4), dim3(256), 0, stream, (const _Float16*)emb_f16, (_Float16*)out_f16, n, d,
                       n_pad, ld_out, (unsigned char*)nullptr, 0, (float*)nullptr, (unsigned long long*)nullptr, 0ull);
  else
    hipLaunchKernelGGL(dedup_normalize_kernel, dim3((n_pad + 3) / 4), dim3(256), 0, stream, (const _Float16*)emb_f16,
                       (_Float16*)out_f16, n, d, n_pad, ld_out);
  return hipGetLastError();
}

// normalise + quantise for the screened search (one pass when the width allows 16-byte accesses); zeroes *cand_count first
hipError_t ce_dedup_normalize_quant(const void* emb_f16, void* out_f16, int n, int d, int ld, void* q8_ws, int ld8, float* margin_ws,
                                    unsigned long long* cand_count, unsigned long long cand_cap, hipStream_t stream) {
  const int n_pad = (n + 255) / 256 * 256;
  if (hipError_t e = hipMemsetAsync(cand_count, 0, sizeof(unsigned long long), stream); e != hipSuccess) return e;
  if (d % 8 == 0 && ld % 8 == 0 && ld8 % 8 == 0 && !(((uintptr_t)emb_f16 | (uintptr_t)out_f16 | (uintptr_t)q8_ws) & 15)) {
    hipLaunchKernelGGL(dedup_normalize_quant_kernel, dim3((n_pad + 3) / 4), dim3(256), 0, stream, (const _Float16*)emb_f16, (_Float16*)out_f16, n, d,
                       n_pad, ld, (unsigned char*)q8_ws, ld8, margin_ws, cand_count, cand_cap + 1);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(dedup_normalize_kernel, dim3((n_pad + 3) / 4), dim3(256), 0, stream, (const _Float16*)emb_f16, (_Float16*)out_f16, n, d, n_pad, ld);
  hipLaunchKernelGGL(dedup_quant_fp8_kernel, dim3((n_pad + 3) / 4), dim3(256), 0, stream, (const _Float16*)out_f16, ld, (unsigned char*)q8_ws, ld8,
                     margin_ws, n_pad, cand_count, cand_cap + 1);
  return hipGetLastError();
}

hipError_t ce_dedup_pairs(const void* ehat_f16, int n, int d, int ld, float threshold, int fp16_compare,
                          long long* pairs, float* vals, unsigned long long capacity, unsigned long long* count,
                          hipStream_t stream) {
  const int n_pad = (n + 255) / 256 * 256;
  GemmParams p{};
  p.A = ehat_f16; p.lda = ld; p.W = ehat_f16; p.ldw = ld; p.M = n_pad; p.N = n_pad; p.K = ld;
  p.tri = 1; p.n_valid = n; p.fp16_compare = fp16_compare; p.thr = threshold;
  p.pairs = pairs; p.vals = vals; p.cap = capacity; p.count = count;
  (void)d;
  return ce_gemm_nt(p, CE_DT_F16, EPI_THRESH, stream);
}

// The same pairs and values as ce_dedup_pairs, found by the e4m3 screen + exact recheck (kernels above), on rows prepared by
// ce_dedup_normalize_quant.  q8_ws: n_pad * ld8 bytes
// (ld8 = ld rounded up to 256, at least 512), margin_ws: n_pad floats, cand_ws: cand_cap slots of 8 bytes, cand_count: the counter ce_dedup_normalize_quant zeroed.
// When the screen finds more candidates than slots, the exact search runs instead, decided on the device (no host round trip).
hipError_t ce_dedup_pairs_screened(const void* ehat_f16, int n, int ld, float threshold, int fp16_compare, void* q8_ws, float* margin_ws,
                                   void* cand_ws, unsigned long long cand_cap, unsigned long long* cand_count, long long* pairs,
                                   float* vals, unsigned long long capacity, unsigned long long* count, hipStream_t stream) {
  const int n_pad = (n + 255) / 256 * 256;
  const int ld8 = ld + 255 < 512 ? 512 : (ld + 255) / 256 * 256;    // zero padded; at least two stage pairs of the fp8 pipeline
  // the smallest true cosine the exact rule can accept: it compares the (fp16-rounded) value with the (fp16-rounded) threshold
  const float thr_x = fp16_compare ? (float)(_Float16)threshold : threshold;
  const float thr_lo = thr_x - 2.0e-3f * fmaxf(1.0f, fabsf(thr_x));
  GemmParams p{};
  p.A = q8_ws; p.lda = ld8; p.W = q8_ws; p.ldw = ld8; p.M = n_pad; p.N = n_pad; p.K = ld8;
  p.tri = 1; p.n_valid = n; p.thr = thr_lo; p.scale_a = margin_ws;
  p.pairs = (long long*)cand_ws; p.cap = cand_cap; p.count = cand_count;
  if (hipError_t e = ce_gemm_fp8_tri(p, stream); e != hipSuccess) return e;
  hipLaunchKernelGGL(dedup_recheck_kernel, dim3(256), dim3(256), 0, stream, (const _Float16*)ehat_f16, ld, (const uint2*)cand_ws, cand_count,
                     cand_cap, threshold, fp16_compare, n, pairs, vals, capacity, count);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  GemmParams x{};
  x.A = ehat_f16; x.lda = ld; x.W = ehat_f16; x.ldw = ld; x.M = n_pad; x.N = n_pad; x.K = ld;
  x.tri = 1; x.n_valid = n; x.fp16_compare = fp16_compare; x.thr = threshold;
  x.pairs = pairs; x.vals = vals; x.cap = capacity; x.count = count;
  x.run_if_over = cand_count; x.run_if_limit = cand_cap;
  return ce_gemm_nt(x, CE_DT_F16, EPI_THRESH, stream);
}
