// Similarity search of /root/reference/tools/find_similar_imgs.py:88-137 as two kernels:
//   simsearch_dist_kernel  distance of every stored embedding row to ONE query (the mean context embedding), "l2"
//                          (|| q - e + 1e-6 ||_2, torch pairwise_distance) or "cosine" ((1 - cos) / 2, torch's 1e-8 clamp
//                          on each norm); one wave per row, fp32 accumulation.  HBM-bound: d * 4 (or 2) bytes per row.
//   topn_kernel            the n smallest (value, index) pairs, ascending, ties by lower index, NaN = +inf: every block
//                          extracts the top n of its slice by n lexicographic-successor argmin passes (no scratch copy,
//                          no atomics, deterministic), one block merges the candidates the same way.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<_Float16>(const _Float16* p) { return (float)*p; }

// VEC consecutive elements per lane and step (16 B when the layout allows it, else 1)
template <typename T, int VEC>
__device__ __forceinline__ void ldv(const T* p, float (&v)[VEC]) {
  if constexpr (VEC == 1) v[0] = ld1<T>(p);
  else if constexpr (sizeof(T) == 4) { const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  else {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    const h8 t = *(const h8*)p;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
  }
}

// One wave per row, rows strided over the grid.  With QCH > 0 the wave keeps its share of the query in registers
// (QCH chunks of VEC elements per lane, d <= 64 * VEC * QCH) and loads two rows per step; QCH == 0 is the generic path.
template <typename T, int VEC, int QCH>
__global__ __launch_bounds__(256) void simsearch_dist_kernel(const T* __restrict__ emb, long n, int d, long row_stride,
                                                             const float* __restrict__ query, int measure,
                                                             float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
  if constexpr (QCH > 0) {
    float q[QCH][VEC];
    float qq = 0.f;
#pragma unroll
    for (int c = 0; c < QCH; ++c) {
      const int k = (c * 64 + lane) * VEC;
#pragma unroll
      for (int j = 0; j < VEC; ++j) { q[c][j] = k < d ? query[k + j] : 0.f; qq = fmaf(q[c][j], q[c][j], qq); }
    }
    qq = wsum(qq);
    for (long row = wave; row < n; row += 2 * n_waves) {
      const long row2 = row + n_waves;
      const bool two = row2 < n;
      const T* e0 = emb + row * row_stride;
      const T* e1 = emb + (two ? row2 : row) * row_stride;
      float v0[QCH][VEC], v1[QCH][VEC];
#pragma unroll
      for (int c = 0; c < QCH; ++c) {
        const int k = (c * 64 + lane) * VEC;
        if (k < d) { ldv<T, VEC>(e0 + k, v0[c]); ldv<T, VEC>(e1 + k, v1[c]); }
        else {
#pragma unroll
          for (int j = 0; j < VEC; ++j) { v0[c][j] = 0.f; v1[c][j] = 0.f; }
        }
      }
      float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
#pragma unroll
      for (int c = 0; c < QCH; ++c) {
        const bool live = (c * 64 + lane) * VEC < d;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if (measure == 0) {
            const float t0 = (q[c][j] - v0[c][j]) + 1e-6f, t1 = (q[c][j] - v1[c][j]) + 1e-6f;
            if (live) { a0 = fmaf(t0, t0, a0); a1 = fmaf(t1, t1, a1); }
          } else {
            a0 = fmaf(q[c][j], v0[c][j], a0); b0 = fmaf(v0[c][j], v0[c][j], b0);
            a1 = fmaf(q[c][j], v1[c][j], a1); b1 = fmaf(v1[c][j], v1[c][j], b1);
          }
        }
      }
      a0 = wsum(a0); a1 = wsum(a1);
      if (measure == 0) {
        if (lane == 0) { out[row] = sqrtf(a0); if (two) out[row2] = sqrtf(a1); }
      } else {
        b0 = wsum(b0); b1 = wsum(b1);
        const float nq = fmaxf(sqrtf(qq), 1e-8f);
        if (lane == 0) {
          out[row] = (1.0f - a0 / (fmaxf(sqrtf(b0), 1e-8f) * nq)) * 0.5f;
          if (two) out[row2] = (1.0f - a1 / (fmaxf(sqrtf(b1), 1e-8f) * nq)) * 0.5f;
        }
      }
    }
  } else {
    for (long row = wave; row < n; row += n_waves) {
      const T* e = emb + row * row_stride;
      float a = 0.f, b = 0.f, c = 0.f;        // l2: a = sum (q - e + eps)^2;  cosine: a = q.e, b = e.e, c = q.q
      for (int k = lane * VEC; k < d; k += 64 * VEC) {
        float v[VEC];
        ldv<T, VEC>(e + k, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float qv = query[k + j];
          if (measure == 0) { const float t = (qv - v[j]) + 1e-6f; a = fmaf(t, t, a); }
          else { a = fmaf(qv, v[j], a); b = fmaf(v[j], v[j], b); c = fmaf(qv, qv, c); }
        }
      }
      a = wsum(a);
      if (measure == 0) {
        if (lane == 0) out[row] = sqrtf(a);
      } else {
        b = wsum(b); c = wsum(c);
        if (lane == 0) out[row] = (1.0f - a / (fmaxf(sqrtf(b), 1e-8f) * fmaxf(sqrtf(c), 1e-8f))) * 0.5f;
      }
    }
  }
}

struct Cand { float v; long long i; };
__device__ __forceinline__ bool less(float v0, long long i0, float v1, long long i1) { return v0 < v1 || (v0 == v1 && i0 < i1); }

// values come either from `dist` (index = position) or from a candidate list (cand_v / cand_i)
__global__ __launch_bounds__(256) void topn_kernel(const float* __restrict__ dist, const float* __restrict__ cand_v,
                                                   const long long* __restrict__ cand_i, long n, long slice, int top_n,
                                                   float* __restrict__ out_v, long long* __restrict__ out_i) {
  __shared__ float sv[256];
  __shared__ long long si[256];
  const long lo = (long)blockIdx.x * slice, hi = min(n, lo + slice);
  float pv = -INFINITY;
  long long pi = -1;                           // last extracted pair; (-inf, -1) precedes everything
  for (int t = 0; t < top_n; ++t) {
    float bv = INFINITY;
    long long bi = 0x7fffffffffffffffLL;
    for (long k = lo + threadIdx.x; k < hi; k += 256) {
      float v = dist ? dist[k] : cand_v[k];
      const long long i = dist ? (long long)k : cand_i[k];
      if (i < 0) continue;                     // padding of a short slice
      if (v != v) v = INFINITY;
      if (less(pv, pi, v, i) && less(v, i, bv, bi)) { bv = v; bi = i; }
    }
    sv[threadIdx.x] = bv; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s && less(sv[threadIdx.x + s], si[threadIdx.x + s], sv[threadIdx.x], si[threadIdx.x])) {
        sv[threadIdx.x] = sv[threadIdx.x + s]; si[threadIdx.x] = si[threadIdx.x + s];
      }
      __syncthreads();
    }
    pv = sv[0]; pi = si[0];
    __syncthreads();
    const bool none = pi == 0x7fffffffffffffffLL;
    if (threadIdx.x == 0) {
      out_v[(size_t)blockIdx.x * top_n + t] = none ? INFINITY : pv;
      out_i[(size_t)blockIdx.x * top_n + t] = none ? -1 : pi;
    }
    if (none) {                                // slice exhausted: pad the rest
      for (int u = t + 1 + threadIdx.x; u < top_n; u += 256) {
        out_v[(size_t)blockIdx.x * top_n + u] = INFINITY;
        out_i[(size_t)blockIdx.x * top_n + u] = -1;
      }
      return;
    }
  }
}

}  // namespace

hipError_t ce_simsearch_distances(const void* emb, int emb_f16, long n, int d, long row_stride, const float* query, int measure,
                                  float* out, hipStream_t stream) {
  if (n < 1 || d < 1 || row_stride < d || (measure != 0 && measure != 1)) return hipErrorInvalidValue;
  const dim3 grid((unsigned)std::min<long>((n + 3) / 4, 4096)), block(256);
  const int vec = emb_f16 ? 8 : 4;
  const bool wide = d % vec == 0 && row_stride % vec == 0 && ((uintptr_t)emb & 15) == 0;
  const int chunks = (d + 64 * vec - 1) / (64 * vec);          // register-resident query when it fits 4 chunks per lane
#define LAUNCH(T, V, Q) hipLaunchKernelGGL((simsearch_dist_kernel<T, V, Q>), grid, block, 0, stream, (const T*)emb, n, d, row_stride, query, measure, out)
  if (emb_f16) {
    if (wide && chunks <= 2) LAUNCH(_Float16, 8, 2);
    else if (wide && chunks <= 4) LAUNCH(_Float16, 8, 4);
    else if (wide) LAUNCH(_Float16, 8, 0);
    else LAUNCH(_Float16, 1, 0);
  } else {
    if (wide && chunks <= 2) LAUNCH(float, 4, 2);
    else if (wide && chunks <= 4) LAUNCH(float, 4, 4);
    else if (wide && chunks <= 8) LAUNCH(float, 4, 8);
    else if (wide) LAUNCH(float, 4, 0);
    else LAUNCH(float, 1, 0);
  }
#undef LAUNCH
  return hipGetLastError();
}

size_t ce_topn_workspace_bytes(long n, int top_n) {
  const long blocks = std::min<long>(256, (n + 1023) / 1024);
  return (size_t)blocks * top_n * (sizeof(float) + sizeof(long long)) + 256;
}

// n smallest of dist[0..n): out_i / out_v [top_n] ascending (padding: index -1, value +inf when n < top_n)
hipError_t ce_topn_smallest(const float* dist, long n, int top_n, long long* out_i, float* out_v, void* ws, size_t ws_bytes,
                            hipStream_t stream) {
  if (n < 1 || top_n < 1 || top_n > 4096) return hipErrorInvalidValue;
  if (ws_bytes < ce_topn_workspace_bytes(n, top_n)) return hipErrorInvalidValue;
  const long blocks = std::min<long>(256, (n + 1023) / 1024);
  const long slice = (n + blocks - 1) / blocks;
  long long* ci = (long long*)ws;                                  // [blocks][top_n]
  float* cv = (float*)((char*)ws + (((size_t)blocks * top_n * sizeof(long long) + 255) & ~(size_t)255));
  hipLaunchKernelGGL(topn_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dist, (const float*)nullptr, (const long long*)nullptr, n, slice,
                     top_n, cv, ci);
  const long m = blocks * top_n;
  hipLaunchKernelGGL(topn_kernel, dim3(1), dim3(256), 0, stream, (const float*)nullptr, cv, ci, m, m, top_n, out_v, out_i);
  return hipGetLastError();
}
