// Row quantisation to OCP fp8 e4m3 for the fp8 GEMM path (BASELINE.json configs[3]):
//   out8[row][k] = e4m3( f(in[row][k]) * 448 / absmax_row ),   scale[row] = absmax_row / 448
// with f = identity (weights, attention output, MLP hidden) or the LayerNorm normalisation (x - mean) * rstd WITHOUT
// the affine part: gamma is folded into the weight rows and beta into the bias at create time, exactly as for the
// bf16 path, so the quantised operand is the unit-variance row.  One wave per row, 16 B per lane and access;
// v_cvt_pk_fp8_f32 does not saturate (1000 -> NaN, tools/probes/fp8probe2.hip), so values are clamped to +-448 first.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int QMAXC = 16;                    // row length <= 8192 elements (ViT-H-14's 5 120-wide hidden rows: weight rows of FC2)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <typename TIN> __device__ __forceinline__ void load8(const TIN* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  const uint4 raw = *(const uint4*)p;
  v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
  v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
  v[4] = __uint_as_float(raw.z << 16); v[5] = __uint_as_float(raw.z & 0xffff0000u);
  v[6] = __uint_as_float(raw.w << 16); v[7] = __uint_as_float(raw.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// One wave per row, NR rows in flight per wave (their loads are issued together: a 2 KiB row alone leaves too few bytes
// in flight to approach the HBM rate), rows strided over the grid.
template <typename TIN, bool LN, int CH, int NR>
__global__ __launch_bounds__(256) void quant_rows_kernel(const TIN* __restrict__ in, size_t ld_in, uint8_t* __restrict__ out,
                                                         size_t ld_out, float* __restrict__ scale, int n_rows, int K, float eps, int pow2,
                                                         int ln_k) {
  // ln_k <= K (LN only): the columns the LayerNorm is over; the rest are a padded tower's zeros and stay zeros
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
  for (int row0 = wave * NR; row0 < n_rows; row0 += n_waves * NR) {
    float v[NR][CH][8];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int row = min(row0 + r, n_rows - 1);
      const TIN* src = in + (size_t)row * ld_in;
#pragma unroll
      for (int ci = 0; ci < CH; ++ci) {
        const int c = ci * 512 + lane * 8;
        if (c < K) load8<TIN>(src + c, v[r][ci]);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[r][ci][j] = 0.f;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int row = row0 + r;
      if (row >= n_rows) break;
      if constexpr (LN) {
        float s = 0.f;
#pragma unroll
        for (int ci = 0; ci < CH; ++ci)
#pragma unroll
          for (int j = 0; j < 8; ++j) s += v[r][ci][j];
        const float mean = wave_sum(s) / (float)ln_k;
        float ss = 0.f;
#pragma unroll
        for (int ci = 0; ci < CH; ++ci)
          if (ci * 512 + lane * 8 < ln_k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[r][ci][j] -= mean; ss += v[r][ci][j] * v[r][ci][j]; }
          }
        const float rstd = rsqrtf(wave_sum(ss) / (float)ln_k + eps);
#pragma unroll
        for (int ci = 0; ci < CH; ++ci)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[r][ci][j] *= rstd;
      }
      float amax = 0.f;
#pragma unroll
      for (int ci = 0; ci < CH; ++ci)
#pragma unroll
        for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[r][ci][j]));
      amax = wave_max(amax);
      float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
      float inv = amax > 0.f ? 448.0f / amax : 0.f;
      if (pow2 && amax > 0.f) {
        // the scale rounded UP to a power of two (the weights of the fused fp8 tower: the exponent rides in the MFMA's block
        // scale, gemm_fp8.hip); e4m3 keeps its relative precision, only its subnormal threshold moves by less than a bit
        const unsigned b = __float_as_uint(sc);
        const unsigned e = min(max((b >> 23) + ((b & 0x7fffffu) ? 1u : 0u), 1u), 254u);
        sc = __uint_as_float(e << 23);
        inv = __uint_as_float((254u - e) << 23);
      }
      uint8_t* dst = out + (size_t)row * ld_out;
#pragma unroll
      for (int ci = 0; ci < CH; ++ci) {
        const int c = ci * 512 + lane * 8;
        if (c < K) {
          float q[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) q[j] = fminf(fmaxf(v[r][ci][j] * inv, -448.0f), 448.0f);
          int w0 = 0, w1 = 0;
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w0, false);
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w0, true);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w1, false);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w1, true);
          *(uint2*)(dst + c) = uint2{(uint32_t)w0, (uint32_t)w1};
        }
      }
      if (lane == 0) scale[row] = sc;
    }
  }
}

// The LayerNorm-quantise pass of the fp8 tower (bf16 residual rows of 768 / 1024 / 1280 columns, twice per block: 18 of a 200 ms
// step with the kernel above, whose three 64-lane reductions per row are six dependent ds_bpermute steps each and whose VALU work
// per row, ~175 instructions, already costs what the HBM transfer does).  Here a row lives in SIXTEEN lanes (lane i of the
// group holds the 16-B pieces i, i + 16, ... of the row: every load and store instruction of the wave still covers whole 128-B
// lines, four rows at a time), so each reduction is four DPP adds inside a DPP row and serves four rows at once: ~100
// instructions per row, no LDS crossbar.  Same arithmetic as above (two-pass variance, y = (x - mean) * rstd, scale = max|y| /
// 448); no clamp: |y * (448 / max|y|)| <= 448 (1 + 2^-22), which v_cvt_pk_fp8_f32 rounds to 448.
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {          // all 16 lanes of a DPP row end with the row's sum
  v += dpp_row<0xB1>(v);                                        // quad_perm [1,0,3,2]
  v += dpp_row<0x4E>(v);                                        // quad_perm [2,3,0,1]
  v += dpp_row<0x141>(v);                                       // row_half_mirror
  v += dpp_row<0x140>(v);                                       // row_mirror
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_row<0xB1>(v));
  v = fmaxf(v, dpp_row<0x4E>(v));
  v = fmaxf(v, dpp_row<0x141>(v));
  v = fmaxf(v, dpp_row<0x140>(v));
  return v;
}

template <int NCH>                                              // K = 128 * NCH
__global__ __launch_bounds__(256) void quant_ln16_kernel(const bf16_t* __restrict__ in, size_t ld_in, uint8_t* __restrict__ out,
                                                         size_t ld_out, float* __restrict__ scale, int n_rows, float eps) {
  constexpr int K = NCH * 128;
  const int lane = threadIdx.x & 63, sub = lane & 15, rr = lane >> 4;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
  for (int row0 = wave * 4; row0 < n_rows; row0 += n_waves * 4) {
    const int row = min(row0 + rr, n_rows - 1);
    const bf16_t* src = in + (size_t)row * ld_in + sub * 8;
    float v[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) load8<bf16_t>(src + c * 128, v[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[c][j];
    const float mean = row16_sum(s) * (1.0f / (float)K);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[c][j] -= mean; ss += v[c][j] * v[c][j]; }
    const float rstd = rsqrtf(row16_sum(ss) * (1.0f / (float)K) + eps);
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[c][j] *= rstd; amax = fmaxf(amax, fabsf(v[c][j])); }
    amax = row16_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = amax > 0.f ? 448.0f / amax : 0.f;
    if (row0 + rr < n_rows) {
      uint8_t* dst = out + (size_t)row * ld_out + sub * 8;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * inv, v[c][1] * inv, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * inv, v[c][3] * inv, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] * inv, v[c][5] * inv, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] * inv, v[c][7] * inv, w1, true);
        *(uint2*)(dst + c * 128) = uint2{(uint32_t)w0, (uint32_t)w1};
      }
      if (sub == 0) scale[row] = sc;
    }
  }
}

// ---- block-exponent rows (gemm_fp8.hip, gemm.h): x[row][k] ~ e4m3 * 2^(exp[row][k / 256] - 127) ----
// The same row layout as above (16 lanes per row, lane i holds the 16-B pieces i, i + 16, ...): a 256-column block is the pieces
// 32 b .. 32 b + 31 = two values of c in all 16 lanes, so its maximum is one DPP-row reduction.  The exponent rule is the
// GEMM epilogue's (gemm_fp8.hip, EPI 3): biased exponent ex of the block's max |x|, byte = max(ex - 7, 0), |x * 2^-e| < 256.
// Also writes the row's (sum, sum of squares), the statistics the consuming GEMM's folded LayerNorm needs.
template <int NCH>                                              // K = 128 * NCH, NCH even
__global__ __launch_bounds__(256) void quant_block_kernel(const bf16_t* __restrict__ in, size_t ld_in, uint8_t* __restrict__ out,
                                                          size_t ld_out, uint8_t* __restrict__ exps, size_t ld_exp,
                                                          float* __restrict__ stats, int n_rows) {
  const int lane = threadIdx.x & 63, sub = lane & 15, rr = lane >> 4;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
  for (int row0 = wave * 4; row0 < n_rows; row0 += n_waves * 4) {
    const int row = min(row0 + rr, n_rows - 1);
    const bf16_t* src = in + (size_t)row * ld_in + sub * 8;
    float v[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c) load8<bf16_t>(src + c * 128, v[c]);
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) { s += v[c][j]; ss = fmaf(v[c][j], v[c][j], ss); }
    s = row16_sum(s);
    ss = row16_sum(ss);
    unsigned ebytes = 0;
    float mul[NCH / 2];
#pragma unroll
    for (int b = 0; b < NCH / 2; ++b) {
      float amax = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fmaxf(fabsf(v[2 * b][j]), fabsf(v[2 * b + 1][j])));
      amax = row16_max(amax);
      const int ex = (int)((__float_as_uint(amax) >> 23) & 0xffu);
      const int eb = max(ex - 7, 0);
      mul[b] = __uint_as_float((unsigned)(254 - eb) << 23);
      ebytes |= (unsigned)eb << (8 * b);
    }
    if (row0 + rr < n_rows) {
      uint8_t* dst = out + (size_t)row * ld_out + sub * 8;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float m = mul[c >> 1];
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][0] * m, v[c][1] * m, w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][2] * m, v[c][3] * m, w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][4] * m, v[c][5] * m, w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[c][6] * m, v[c][7] * m, w1, true);
        *(uint2*)(dst + c * 128) = uint2{(uint32_t)w0, (uint32_t)w1};
      }
      if (sub == 0) {
        *(unsigned*)(exps + (size_t)row * ld_exp) = ebytes;
        if (stats) *(float2*)(stats + (size_t)row * 2) = float2{s, ss};
      }
    }
  }
}

// (row_r, row_d) = (rstd, -mean * rstd) of every row from its `parts` partial (sum, sum of squares): the per-row constants of the
// LayerNorm folded into the fp8 GEMM's epilogue.  One thread per row; stats [parts][ld][2].
__global__ __launch_bounds__(256) void row_norm_consts_kernel(const float* __restrict__ stats, int parts, size_t ld, int n_rows,
                                                              float inv_width, float eps, float* __restrict__ row_r,
                                                              float* __restrict__ row_d, int ld_row) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= n_rows) return;
  float s = 0.f, ss = 0.f;
  for (int q = 0; q < parts; ++q) {
    const float2 t = *(const float2*)(stats + ((size_t)q * ld + row) * 2);
    s += t.x; ss += t.y;
  }
  const float mean = s * inv_width;
  const float var = fmaxf(ss * inv_width - mean * mean, 0.f);
  const float r = rsqrtf(var + eps);
  row_r[(size_t)row * ld_row] = r;
  row_d[(size_t)row * ld_row] = -mean * r;
}

// colsum[n] = scale[n] * sum_k e4m3(W8[n][k]): the column sums of the DEQUANTISED weight rows, what the folded mean term of
// the fp8 GEMM multiplies.  One wave per row.
__global__ __launch_bounds__(256) void colsum_fp8_kernel(const uint8_t* __restrict__ W8, const float* __restrict__ scale, int N, int K,
                                                         float* __restrict__ colsum) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float s = 0.f;
  for (int c = lane * 4; c < K; c += 256) {
    const int wd = *(const int*)(W8 + (size_t)n * K + c);
    const f32x2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(wd, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(wd, true);
    s += (lo[0] + lo[1]) + (hi[0] + hi[1]);
  }
  s = wave_sum(s);
  if (lane == 0) colsum[n] = s * scale[n];
}

// Static (data-free) output scale of a LayerNorm-fed linear layer, from the Cauchy-Schwarz bound
//   |LN(x) . w'_n + b'_n| <= sqrt(K) * ||w'_n||_2 + |b'_n|      (||LN(x) without affine||_2 <= sqrt(K))
// widened by 1.07 for the e4m3 rounding of both operands and by a 1.2 margin:  s[n] = bound / 448 * 1.2, inv_s = 1/s.
// |act(u)| <= |u| for both GELUs, and a softmax-weighted mean of such values obeys the same bound, so
// e4m3(value * inv_s[n]) can never overflow.  One wave per row.
__global__ __launch_bounds__(256) void static_scale_kernel(const bf16_t* __restrict__ W, const float* __restrict__ bias, int N,
                                                           int K, float* __restrict__ s, float* __restrict__ inv_s) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float ss = 0.f;
  for (int c = lane * 8; c < K; c += 512) {
    float v[8];
    load8<bf16_t>(W + (size_t)n * K + c, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
  }
  ss = wave_sum(ss);
  if (lane == 0) {
    const float bound = 1.2f * (1.07f * sqrtf((float)K) * sqrtf(ss) + fabsf(bias[n])) * (1.0f / 448.0f);
    const float sc = bound > 1e-30f ? bound : 1.0f;
    s[n] = sc;
    inv_s[n] = 1.0f / sc;
  }
}

// out[n][k] = W[n][k] * s[k]  (fp32): the static scale of the operand's columns folded into the consuming weight
__global__ __launch_bounds__(256) void scale_cols_kernel(const bf16_t* __restrict__ W, const float* __restrict__ s,
                                                         float* __restrict__ out, size_t total, int K) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= total) return;
  float v[8];
  load8<bf16_t>(W + i, v);
  const int k = (int)(i % (size_t)K);
  const float4 s0 = *(const float4*)(s + k), s1 = *(const float4*)(s + k + 4);
  *(float4*)(out + i) = float4{v[0] * s0.x, v[1] * s0.y, v[2] * s0.z, v[3] * s0.w};
  *(float4*)(out + i + 4) = float4{v[4] * s1.x, v[5] * s1.y, v[6] * s1.z, v[7] * s1.w};
}

}  // namespace

hipError_t ce_static_scale(const void* W_bf16, const float* bias, int N, int K, float* s, float* inv_s, hipStream_t stream) {
  if (N < 1 || K < 8 || K % 8 != 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(static_scale_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, (const bf16_t*)W_bf16, bias, N, K, s, inv_s);
  return hipGetLastError();
}

hipError_t ce_scale_cols(const void* W_bf16, const float* s, float* out_f32, int N, int K, hipStream_t stream) {
  if (N < 1 || K < 8 || K % 8 != 0) return hipErrorInvalidValue;
  const size_t total = (size_t)N * K;
  hipLaunchKernelGGL(scale_cols_kernel, dim3((unsigned)((total / 8 + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)W_bf16, s,
                     out_f32, total, K);
  return hipGetLastError();
}

// bf16 rows of K = 256 .. 1024 elements (K % 256 == 0) -> e4m3 rows + one exponent dword per row (bytes 0 .. K/256 - 1) and,
// if stats != NULL, the row's (sum, sum of squares) at stats[row][2]
hipError_t ce_quant_block_fp8(const void* in, size_t ld_in, void* out8, size_t ld_out, void* exps, size_t ld_exp, float* stats,
                              int n_rows, int K, hipStream_t stream) {
  if (n_rows < 1 || K < 256 || K > 1024 || K % 256 != 0 || ld_in < (size_t)K || ld_out < (size_t)K || ld_in % 8 != 0 ||
      ld_out % 8 != 0 || ld_exp < 4 || ld_exp % 4 != 0 || (((uintptr_t)in | (uintptr_t)out8) & 15) || ((uintptr_t)exps & 3) ||
      ((uintptr_t)stats & 7))
    return hipErrorInvalidValue;
  const int waves = (n_rows + 3) / 4;
  dim3 grid((unsigned)std::min((waves + 3) / 4, 16384)), block(256);
#define QB(NCH_)                                                                                                                  \
  case NCH_: hipLaunchKernelGGL((quant_block_kernel<NCH_>), grid, block, 0, stream, (const bf16_t*)in, ld_in, (uint8_t*)out8,     \
                                ld_out, (uint8_t*)exps, ld_exp, stats, n_rows);                                                   \
    return hipGetLastError();
  switch (K / 128) { QB(2) QB(4) QB(6) QB(8) default: break; }
#undef QB
  return hipErrorInvalidValue;
}

hipError_t ce_row_norm_consts(const float* stats, int parts, size_t ld, int n_rows, int width, float eps, float* row_r, float* row_d,
                              int ld_row, hipStream_t stream) {
  if (n_rows < 1 || parts < 1 || ld < (size_t)n_rows || width < 1 || ld_row < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(row_norm_consts_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, stream, stats, parts, ld, n_rows,
                     1.0f / (float)width, eps, row_r, row_d, ld_row);
  return hipGetLastError();
}

namespace {
// E8M0 byte (the biased exponent) of every scale: what the MFMA applies as the block scale of the weight rows when the scales are
// powers of two (ce_quant_rows_fp8 with pow2).  `bad` counts the scales that are not (mantissa bits set, zero, infinity, NaN).
__global__ void scale_exponents_kernel(const float* __restrict__ scale, unsigned char* __restrict__ out, int n, unsigned* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned b = __float_as_uint(scale[i]);
  const unsigned e = (b >> 23) & 0xffu;
  out[i] = (unsigned char)e;
  if (bad && ((b & 0x807fffffu) || e == 0u || e == 255u)) atomicAdd(bad, 1u);
}
}  // namespace

hipError_t ce_scale_exponents(const float* scale, unsigned char* exp_out, int n, unsigned* bad, hipStream_t stream) {
  hipLaunchKernelGGL(scale_exponents_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, scale, exp_out, n, bad);
  return hipGetLastError();
}

hipError_t ce_colsum_fp8(const void* W8, const float* scale, int N, int K, float* colsum, hipStream_t stream) {
  if (N < 1 || K < 4 || K % 4 != 0 || ((uintptr_t)W8 & 3)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(colsum_fp8_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, (const uint8_t*)W8, scale, N, K, colsum);
  return hipGetLastError();
}

// in: bf16 (in_f32 == 0) or fp32 rows of K elements (K % 8 == 0, K <= 8192); ln != 0 normalises each row first.
hipError_t ce_quant_rows_fp8(const void* in, int in_f32, size_t ld_in, void* out8, size_t ld_out, float* scale, int n_rows,
                             int K, int ln, float eps, hipStream_t stream, int pow2, int ln_k) {
  if (n_rows < 1 || K < 8 || K % 8 != 0 || K > QMAXC * 512 || ld_in < (size_t)K || ld_out < (size_t)K) return hipErrorInvalidValue;
  if (ln_k <= 0) ln_k = K;
  if (ln_k > K || ln_k % 8 != 0) return hipErrorInvalidValue;
  if (!pow2 && !in_f32 && ln && ln_k == K && K % 128 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && ((uintptr_t)in | (uintptr_t)out8) % 16 == 0) {
    // the tower's LayerNorm-quantise pass: 16 lanes per row (quant_ln16_kernel)
    const int waves = (n_rows + 3) / 4;
    dim3 grid((unsigned)std::min((waves + 3) / 4, 16384)), block(256);
#define Q16(NCH_)                                                                                                                  \
    case NCH_: hipLaunchKernelGGL((quant_ln16_kernel<NCH_>), grid, block, 0, stream, (const bf16_t*)in, ld_in, (uint8_t*)out8,     \
                                  ld_out, scale, n_rows, eps);                                                                     \
      return hipGetLastError();
    switch (K / 128) { Q16(6) Q16(8) Q16(10) default: break; }
#undef Q16
  }
  const int chunks = (K + 511) / 512;
  // rows per wave step: as many as keep the row registers (CH * 8 * NR floats) around 64
#define QLAUNCH(T, LNV, CH, NR)                                                             \
  do {                                                                                      \
    const int waves = (n_rows + (NR) - 1) / (NR);                                           \
    dim3 grid((unsigned)std::min((waves + 3) / 4, 8192)), block(256);                       \
    hipLaunchKernelGGL((quant_rows_kernel<T, LNV, CH, NR>), grid, block, 0, stream, (const T*)in, ld_in, (uint8_t*)out8, ld_out, scale, \
                       n_rows, K, eps, pow2, ln_k);                                               \
  } while (0)
#define QDISPATCH(T, LNV)                                                                   \
  do {                                                                                      \
    if (chunks <= 1) QLAUNCH(T, LNV, 1, 4);                                                 \
    else if (chunks <= 2) QLAUNCH(T, LNV, 2, 4);                                            \
    else if (chunks <= 4) QLAUNCH(T, LNV, 4, 2);                                            \
    else if (chunks <= 8) QLAUNCH(T, LNV, 8, 1);                                            \
    else QLAUNCH(T, LNV, 16, 1);                                                            \
  } while (0)
  if (in_f32) { if (ln) QDISPATCH(float, true); else QDISPATCH(float, false); }
  else { if (ln) QDISPATCH(bf16_t, true); else QDISPATCH(bf16_t, false); }
#undef QDISPATCH
#undef QLAUNCH
  return hipGetLastError();
}
