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

}  // namespace

hipError_t ce_dedup_normalize_f16(const void* emb_f16, void* out_f16, int n, int d, int ld_out, hipStream_t stream) {
  const int n_pad = (n + 255) / 256 * 256;
  hipLaunchKernelGGL(dedup_normalize_kernel, dim3((n_pad + 3) / 4), dim3(256), 0, stream, (const _Float16*)emb_f16,
                     (_Float16*)out_f16, n, d, n_pad, ld_out);
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
