// Diversity ordering of a labeling session (/root/reference/_3_label_images.py:128-177): a sampled farthest-point walk in
// CLIP space.  The reference keeps the list of chosen embeddings and, per step, loads `sample_size` random `.pt` files,
// forms the [chosen x sample] cosine matrix, takes the column maxima and appends the sample whose maximum is smallest.
// Here every stored embedding keeps its running maximum cosine to the chosen set, so a step is
//   div_update_kernel: maxsim[i] = max(maxsim[i], cos(e_i, e_chosen))  for ALL rows -- one HBM-bound pass (one wave per
//                      row, the chosen row held in registers, two rows in flight per wave; the same scan as simsearch), and
//   div_pick_kernel:   argmin of maxsim over the step's sampled candidates (first minimum, as torch.argmin), appended to
//                      the order and published on the device for the next update -- no host round trip between steps.
// fp32 throughout like the reference; cos = (a . b) / (|a| |b|) with the norms computed once (the reference divides
// each vector by its norm first, :130-131: the same value up to rounding).
#include <stdint.h>

#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// inv_norm[i] = 1 / ||e_i||  (no epsilon: the reference divides by the plain norm, :130), maxsim[i] = -inf
__global__ __launch_bounds__(256) void div_init_kernel(const float* __restrict__ emb, long n, int d, long ld, float* __restrict__ inv_norm,
                                                       float* __restrict__ maxsim, int* __restrict__ cur, int first) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *cur = first;
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  for (long i = wave; i < n; i += nwaves) {
    const float* r = emb + i * ld;
    float ss = 0.f;
    for (int k = lane; k < d; k += 64) { const float x = r[k]; ss += x * x; }
    ss = wave_sum(ss);
    if (lane == 0) { inv_norm[i] = 1.0f / sqrtf(ss); maxsim[i] = -INFINITY; }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void div_update_kernel(const float* __restrict__ emb, long n, int d, long ld, const float* __restrict__ inv_norm,
                                                         const int* __restrict__ cur, float* __restrict__ maxsim) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  const long c = *cur;
  const float* q = emb + c * ld;
  const float qinv = inv_norm[c];
  if constexpr (VEC == 4) {
    // d % 4 == 0, 16-B aligned rows: the chosen row in registers (up to 8 float4 per lane = d <= 2048), two rows in flight
    float4 qr[8];
    const int chunks = (d / 4 + 63) / 64;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int k4 = t * 64 + lane;
      qr[t] = (t < chunks && k4 * 4 < d) ? *(const float4*)(q + k4 * 4) : float4{0.f, 0.f, 0.f, 0.f};
    }
    for (long i = wave * 2; i < n; i += nwaves * 2) {
      const float* r0 = emb + i * ld;
      const bool two = i + 1 < n;
      const float* r1 = two ? r0 + ld : r0;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int k4 = t * 64 + lane;
        if (t < chunks && k4 * 4 < d) {
          const float4 a = *(const float4*)(r0 + k4 * 4), b = *(const float4*)(r1 + k4 * 4);
          s0 += a.x * qr[t].x + a.y * qr[t].y + a.z * qr[t].z + a.w * qr[t].w;
          s1 += b.x * qr[t].x + b.y * qr[t].y + b.z * qr[t].z + b.w * qr[t].w;
        }
      }
      s0 = wave_sum(s0); s1 = wave_sum(s1);
      if (lane == 0) {
        maxsim[i] = fmaxf(maxsim[i], s0 * inv_norm[i] * qinv);
        if (two) maxsim[i + 1] = fmaxf(maxsim[i + 1], s1 * inv_norm[i + 1] * qinv);
      }
    }
  } else {
    for (long i = wave; i < n; i += nwaves) {
      const float* r = emb + i * ld;
      float s = 0.f;
      for (int k = lane; k < d; k += 64) s += r[k] * q[k];
      s = wave_sum(s);
      if (lane == 0) maxsim[i] = fmaxf(maxsim[i], s * inv_norm[i] * qinv);
    }
  }
}

// one workgroup: first minimum of maxsim over the step's candidates (NaN never wins, as it never compares smaller)
__global__ __launch_bounds__(256) void div_pick_kernel(const float* __restrict__ maxsim, const int* __restrict__ cand, int k, int* __restrict__ cur,
                                                       int* __restrict__ order_slot) {
  __shared__ float sv[256];
  __shared__ int sp[256];
  float bv = INFINITY; int bp = 0x7fffffff;
  for (int j = threadIdx.x; j < k; j += 256) {
    const float v = maxsim[cand[j]];
    if (v < bv || (v == bv && j < bp)) { bv = v; bp = j; }
  }
  sv[threadIdx.x] = bv; sp[threadIdx.x] = bp;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const float v = sv[threadIdx.x + s]; const int pidx = sp[threadIdx.x + s];
      if (v < sv[threadIdx.x] || (v == sv[threadIdx.x] && pidx < sp[threadIdx.x])) { sv[threadIdx.x] = v; sp[threadIdx.x] = pidx; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int pos = sp[0] == 0x7fffffff ? 0 : sp[0];       // every candidate NaN / +inf: torch.argmin returns 0
    const int chosen = cand[pos];
    *cur = chosen; *order_slot = chosen;
  }
}

}  // namespace

size_t ce_diversity_workspace_bytes(long n) { return ((size_t)n * 2 * sizeof(float) + 255) / 256 * 256 + 256; }

hipError_t ce_diversity_order(const float* emb, long n, int d, long ld, int first, const int* samples, int steps, int sample_size,
                              int* order, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n < 1 || d < 1 || ld < d || first < 0 || first >= n || steps < 0 || sample_size < 1) return hipErrorInvalidValue;
  if (ws_bytes < ce_diversity_workspace_bytes(n)) return hipErrorInvalidValue;
  float* inv_norm = (float*)ws;
  float* maxsim = inv_norm + n;
  int* cur = (int*)((char*)ws + ((size_t)n * 2 * sizeof(float) + 255) / 256 * 256);
  const unsigned grid = (unsigned)std::min<long>((n + 3) / 4, 2048);
  hipLaunchKernelGGL(div_init_kernel, dim3(grid), dim3(256), 0, stream, emb, n, d, ld, inv_norm, maxsim, cur, first);
  const bool wide = d % 4 == 0 && ld % 4 == 0 && ((uintptr_t)emb & 15) == 0 && d <= 2048;
  const unsigned ugrid = (unsigned)std::min<long>((n + 7) / 8, 2048);
  for (int t = 0; t < steps; ++t) {
    if (wide) hipLaunchKernelGGL((div_update_kernel<4>), dim3(ugrid), dim3(256), 0, stream, emb, n, d, ld, inv_norm, cur, maxsim);
    else hipLaunchKernelGGL((div_update_kernel<1>), dim3(grid), dim3(256), 0, stream, emb, n, d, ld, inv_norm, cur, maxsim);
    hipLaunchKernelGGL(div_pick_kernel, dim3(1), dim3(256), 0, stream, maxsim, samples + (size_t)t * sample_size, sample_size, cur, order + t);
  }
  return hipGetLastError();
}
