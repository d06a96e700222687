// Bandwidth-bound helpers around the GEMM chain of the ViT tower (K0, K2, K8, K9 of SURVEY.md §2.2;
// the open_clip steps behind /root/reference/utils/embedder.py:98 and the normalise of :99).
#include "common.h"
#include "kernels.h"

namespace {

// ---------------------------------------------------------------------------------------------
// K0: fp32 NCHW crops -> bf16 patch rows  A[(crop*g + gy)*g + gx][k],  k = (c*p + ky)*p + kx,
// zero-padded to kpad (the conv-as-GEMM operand; conv1 has stride = patch and no bias).
// One block per (crop, gy): the 3*p input rows it touches are read in contiguous runs of p floats.
// ---------------------------------------------------------------------------------------------
// uint8 input (CLIPENC_IN_U8): the ToTensor + Normalize tail of the validation transform
// (/root/reference/utils/embedder.py:90-92) is applied here, v = (u / 255 - mean_c) / std_c in fp32 with the same
// operation order as torchvision, so the host only ships 1 byte per pixel.
struct PixelNorm { float mean[3], std[3]; };

template <typename TIN>
__device__ __forceinline__ float load_pixel(const TIN* p, int c, const PixelNorm& nm) { return (float)*p; }
template <>
__device__ __forceinline__ float load_pixel<uint8_t>(const uint8_t* p, int c, const PixelNorm& nm) {
  return ((float)*p / 255.0f - nm.mean[c]) / nm.std[c];
}

// One workgroup per (crop, gy) = one row of g patches.  The 3 * patch image rows it needs are contiguous runs of `image`
// elements: they are fetched with 16-B loads per lane (whole 128-B lines per 8 lanes; the first version read one element
// per lane in runs of `patch` elements and wrote 2 B per lane: 2.8 TB/s), converted once and kept in LDS as bf16
// [c][ky][x]; the g * kpad output elements of the patch row are one contiguous block of dst and leave as 16-B pieces of 8
// consecutive k (gathered from the LDS image with 2-B reads: LDS bandwidth is not what this kernel is short of).
// Same arithmetic per element as before (bitwise-equal operand, checked by tests/test_gpu_parity.py).
template <typename TIN> struct Vec16 { static constexpr int N = 16 / sizeof(TIN); };

// PATCH is a template constant: the index arithmetic of the gather divides by patch, patch^2 and kpad, which must be
// multiplications (the run-time form costs more VALU than the block has memory time for).
template <typename TIN, int PATCH>
__global__ __launch_bounds__(256) void patchify_kernel(const TIN* __restrict__ in, bf16_t* __restrict__ out,
                                                       int image, PixelNorm nm) {
  constexpr int patch = PATCH, kpad = (3 * PATCH * PATCH + 127) / 128 * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* img = (bf16_t*)smem;                       // [3 * patch][image]
  constexpr int V = Vec16<TIN>::N;                   // elements per 16-B load: 4 fp32, 8 f16, 16 uint8
  const int g = image / patch;
  const int crop = blockIdx.x / g, gy = blockIdx.x % g;
  const int k_real = 3 * patch * patch;
  const TIN* src = in + (size_t)crop * 3 * image * image;
  bf16_t* dst = out + ((size_t)crop * g + gy) * g * kpad;
  const int rows = 3 * patch, per_row = image / V;   // host checks image % V == 0 for the vector path
  for (int idx = threadIdx.x; idx < rows * per_row; idx += 256) {
    const int r = idx / per_row, v = idx - r * per_row;
    const int c = r / patch, ky = r - c * patch;
    const TIN* p = src + ((size_t)c * image + gy * patch + ky) * image + v * V;
    const uint4 raw = *(const uint4*)p;
    const TIN* e = (const TIN*)&raw;
    bf16_t* o = img + (size_t)r * image + v * V;
#pragma unroll
    for (int j = 0; j < V; ++j) o[j] = f32_to_bf16(load_pixel<TIN>(e + j, c, nm));
  }
  __syncthreads();
  const int pp = patch * patch;
  const int pieces = g * kpad / 8;                   // kpad % 128 == 0
  for (int idx = threadIdx.x; idx < pieces; idx += 256) {
    const int gx = (idx * 8) / kpad, k0 = idx * 8 - gx * kpad;
    // (c, ky, kx) of the piece's first k, advanced element by element
    int c = k0 / pp, rem = k0 - c * pp;
    int ky = rem / patch, kx = rem - ky * patch;
    const bf16_t* rowp = img + (c * patch + ky) * image + gx * patch;
    bf16_t w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      w[j] = (k0 + j < k_real) ? rowp[kx] : (bf16_t)0;
      if (++kx == patch) { kx = 0; rowp += image; }          // next ky (c * patch + ky is one running row index)
    }
    *(uint4*)(dst + (size_t)idx * 8) = *(const uint4*)w;
  }
}

// element-per-lane fallback for image sizes whose rows are not a whole number of 16-B vectors
template <typename TIN>
__global__ __launch_bounds__(256) void patchify_scalar_kernel(const TIN* __restrict__ in, bf16_t* __restrict__ out,
                                                              int image, int patch, int kpad, PixelNorm nm) {
  const int g = image / patch;
  const int crop = blockIdx.x / g, gy = blockIdx.x % g;
  const int k_real = 3 * patch * patch;
  const TIN* src = in + (size_t)crop * 3 * image * image;
  bf16_t* dst = out + ((size_t)crop * g + gy) * g * kpad;
  const int total = g * kpad;
  for (int idx = threadIdx.x; idx < total; idx += 256) {
    const int gx = idx / kpad, k = idx - gx * kpad;
    float v = 0.f;
    if (k < k_real) {
      const int c = k / (patch * patch), rem = k - c * patch * patch;
      const int ky = rem / patch, kx = rem - ky * patch;
      v = load_pixel<TIN>(src + ((size_t)c * image + gy * patch + ky) * image + gx * patch + kx, c, nm);
    }
    dst[idx] = f32_to_bf16(v);
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---------------------------------------------------------------------------------------------
// K2: x[crop][tok] = LayerNorm_pre( (tok == 0 ? class_embedding : patch_emb[crop][tok-1]) + pos[tok] )
// written as bf16, plus the per-row (sum, sumsq) of the ROUNDED row for the LayerNorm folded into
// the first QKV GEMM.  One wave per token row; lanes own 8-element (16 B) column chunks.
// ---------------------------------------------------------------------------------------------
// Token-major: a wave owns ONE token position and walks `cpw` crops of it, so pos[tok], gamma and beta (12 KB per row
// from L2 against 2 KB of patch embedding and 2 KB of output from/to HBM in the row-major first version: 3.7 TB/s) are
// read once per wave and live in registers; per row only the patch embedding is loaded, two rows in flight per wave.
// Row arithmetic (per-lane accumulation order, wave reductions) is unchanged, so the output is bitwise the same.
constexpr int LN_MAX_CHUNKS = 4;   // width <= 2048

template <int NCH>
__global__ __launch_bounds__(256) void embed_ln_pre_kernel(const bf16_t* __restrict__ pe, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, bf16_t* __restrict__ x,
                                                           float* __restrict__ stats, int n_crops, int n_tok, int width,
                                                           int ln_width, float eps, int cpw, int waves_per_tok) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tok = gw / waves_per_tok, slot = gw - tok * waves_per_tok;
  if (tok >= n_tok) return;
  const int crop0 = slot * cpw, crop1 = min(n_crops, crop0 + cpw);
  float pz[NCH][8], gm[NCH][8], bt[NCH][8];
#pragma unroll
  for (int ci = 0; ci < NCH; ++ci) {
    const int c = ci * 512 + lane * 8;
    if (c < width) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float4 p = *(const float4*)(pos + (size_t)tok * width + c + 4 * h);
        const float4 g = *(const float4*)(gamma + c + 4 * h), b = *(const float4*)(beta + c + 4 * h);
        pz[ci][4 * h + 0] = p.x; pz[ci][4 * h + 1] = p.y; pz[ci][4 * h + 2] = p.z; pz[ci][4 * h + 3] = p.w;
        gm[ci][4 * h + 0] = g.x; gm[ci][4 * h + 1] = g.y; gm[ci][4 * h + 2] = g.z; gm[ci][4 * h + 3] = g.w;
        bt[ci][4 * h + 0] = b.x; bt[ci][4 * h + 1] = b.y; bt[ci][4 * h + 2] = b.z; bt[ci][4 * h + 3] = b.w;
      }
    }
  }
  if (tok == 0) {                                    // class token: the same embedding for every crop
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
      const int c = ci * 512 + lane * 8;
      if (c < width) {
        const float4 c0 = *(const float4*)(cls + c), c1 = *(const float4*)(cls + c + 4);
        pz[ci][0] += c0.x; pz[ci][1] += c0.y; pz[ci][2] += c0.z; pz[ci][3] += c0.w;
        pz[ci][4] += c1.x; pz[ci][5] += c1.y; pz[ci][6] += c1.z; pz[ci][7] += c1.w;
      }
    }
  }
  // one row: v = e + pos (e = 0 and pos already holds cls + pos for the class token), LayerNorm, store, statistics
  auto row_out = [&](int crop, const uint4 (&raw)[NCH]) {
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
      const int c = ci * 512 + lane * 8;
      if (c < width) {
        if (tok == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[ci][j] = pz[ci][j];
        } else {
          const uint32_t w4[4] = {raw[ci].x, raw[ci].y, raw[ci].z, raw[ci].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[ci][2 * j] = __uint_as_float(w4[j] << 16) + pz[ci][2 * j];
            v[ci][2 * j + 1] = __uint_as_float(w4[j] & 0xffff0000u) + pz[ci][2 * j + 1];
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[ci][j];
      }
    }
    // a zero-padded tower (capi.hip, clipenc_create): columns ln_width .. width - 1 hold zeros and zero gamma / beta; they add nothing to the
    // sum, mean^2 each to the squared deviations (taken off again: exactly nothing when ln_width == width), and come out as zeros
    const float mean = wave_sum(s) / (float)ln_width;
    float ss = 0.f;
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
      const int c = ci * 512 + lane * 8;
      if (c < width) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[ci][j] - mean; ss += d * d; }
      }
    }
    const float rstd = rsqrtf((wave_sum(ss) - (float)(width - ln_width) * mean * mean) / (float)ln_width + eps);
    float rs = 0.f, rss = 0.f;
    const size_t row = (size_t)crop * n_tok + tok;
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
      const int c = ci * 512 + lane * 8;
      if (c < width) {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = (v[ci][j] - mean) * rstd * gm[ci][j] + bt[ci][j];
        uint4 pk = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7])};
        *(uint4*)(x + row * width + c) = pk;
        const uint32_t w4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = __uint_as_float(w4[j] << 16), b = __uint_as_float(w4[j] & 0xffff0000u);
          rs += a + b; rss += a * a + b * b;
        }
      }
    }
    rs = wave_sum(rs); rss = wave_sum(rss);
    if (lane == 0) *(float2*)(stats + row * 2) = float2{rs, rss};
  };
  auto load_row = [&](int crop, uint4 (&raw)[NCH]) {
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
      const int c = ci * 512 + lane * 8;
      raw[ci] = uint4{0, 0, 0, 0};
      if (c < width && tok != 0) raw[ci] = *(const uint4*)(pe + ((size_t)crop * (n_tok - 1) + tok - 1) * width + c);
    }
  };
  int crop = crop0;
  for (; crop + 1 < crop1; crop += 2) {              // two rows in flight
    uint4 ra[NCH], rb[NCH];
    load_row(crop, ra); load_row(crop + 1, rb);
    row_out(crop, ra); row_out(crop + 1, rb);
  }
  if (crop < crop1) {
    uint4 ra[NCH];
    load_row(crop, ra);
    row_out(crop, ra);
  }
}

// ---------------------------------------------------------------------------------------------
// K8 + K9: emb[crop] = normalise( LayerNorm_post(x[crop][0]) . proj[width][embed] )  in fp32.
// HEAD_CROPS crops per block so that proj (3 MB at ViT-L/14) is streamed from L2 once per group.
// ---------------------------------------------------------------------------------------------
constexpr int HEAD_CROPS = 8;      // class-token rows per workgroup: every row of `proj` a thread streams serves 8 crops

__global__ __launch_bounds__(256) void head_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ proj,
                                                   float* __restrict__ emb, int n_crops, int n_tok, int width, int ln_width,
                                                   int embed, float eps, int normalize) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* c = (float*)smem;                         // [HEAD_CROPS][width]
  float* red = c + HEAD_CROPS * width;             // [HEAD_CROPS][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int crop0 = blockIdx.x * HEAD_CROPS;
  // LayerNorm of the CLS rows: wave w handles crops crop0 + w, crop0 + w + 4
  for (int cw = wave; cw < HEAD_CROPS; cw += 4) {
    const int crop = crop0 + cw;
    if (crop < n_crops) {
      const bf16_t* row = x + (size_t)crop * n_tok * width;
      float s = 0.f;
      for (int k = lane; k < width; k += 64) s += bf16_to_f32(row[k]);
      const float mean = wave_sum(s) / (float)ln_width;       // (zero-padded towers: as in embed_ln_pre_kernel)
      float ss = 0.f;
      for (int k = lane; k < width; k += 64) { const float d = bf16_to_f32(row[k]) - mean; ss += d * d; }
      const float rstd = rsqrtf((wave_sum(ss) - (float)(width - ln_width) * mean * mean) / (float)ln_width + eps);
      for (int k = lane; k < width; k += 64)
        c[cw * width + k] = (bf16_to_f32(row[k]) - mean) * rstd * gamma[k] + beta[k];
    } else {
      for (int k = lane; k < width; k += 64) c[cw * width + k] = 0.f;
    }
  }
  __syncthreads();
  float sq[HEAD_CROPS];
#pragma unroll
  for (int r = 0; r < HEAD_CROPS; ++r) sq[r] = 0.f;
  constexpr int MAX_E_PER_THREAD = 5;              // embed <= 1280
  float acc[MAX_E_PER_THREAD][HEAD_CROPS];
#pragma unroll
  for (int i = 0; i < MAX_E_PER_THREAD; ++i)
#pragma unroll
    for (int r = 0; r < HEAD_CROPS; ++r) acc[i][r] = 0.f;
  // k in steps of 4 (width % 4 == 0): the 4 x MAX_E proj loads of a step are independent and in flight together (one k per
  // iteration left this loop waiting on an L2 round trip per k: 1.1 ms for 2048 crops, now ~0.1 ms); every (e, crop)
  // accumulator still sums k in ascending order, so the results are bit-identical to the one-k loop
  for (int k = 0; k < width; k += 4) {
    float w4[MAX_E_PER_THREAD][4];
#pragma unroll
    for (int i = 0; i < MAX_E_PER_THREAD; ++i) {
      const int e = tid + i * 256;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) w4[i][kk] = e < embed ? proj[(size_t)(k + kk) * embed + e] : 0.f;
    }
    float4 cv[HEAD_CROPS];
#pragma unroll
    for (int r = 0; r < HEAD_CROPS; ++r) cv[r] = *(const float4*)(c + r * width + k);
#pragma unroll
    for (int i = 0; i < MAX_E_PER_THREAD; ++i) {
#pragma unroll
      for (int r = 0; r < HEAD_CROPS; ++r) {
        acc[i][r] = fmaf(cv[r].x, w4[i][0], acc[i][r]);
        acc[i][r] = fmaf(cv[r].y, w4[i][1], acc[i][r]);
        acc[i][r] = fmaf(cv[r].z, w4[i][2], acc[i][r]);
        acc[i][r] = fmaf(cv[r].w, w4[i][3], acc[i][r]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAX_E_PER_THREAD; ++i)
#pragma unroll
    for (int r = 0; r < HEAD_CROPS; ++r) sq[r] += acc[i][r] * acc[i][r];
#pragma unroll
  for (int r = 0; r < HEAD_CROPS; ++r) {
    const float t = wave_sum(sq[r]);
    if (lane == 0) red[r * 4 + wave] = t;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < HEAD_CROPS; ++r) {
    const int crop = crop0 + r;
    if (crop >= n_crops) continue;
    float scale = 1.0f;
    if (normalize) scale = 1.0f / sqrtf(red[r * 4 + 0] + red[r * 4 + 1] + red[r * 4 + 2] + red[r * 4 + 3]);
#pragma unroll
    for (int i = 0; i < MAX_E_PER_THREAD; ++i) {
      const int e = tid + i * 256;
      if (e < embed) emb[(size_t)crop * embed + e] = acc[i][r] * scale;
    }
  }
}

}  // namespace

hipError_t ce_patchify(const void* crops, int in_dtype, void* a_patch, int n_crops, int image, int patch, int kpad,
                       const float* mean3, const float* std3, hipStream_t stream) {
  if (image % patch != 0 || kpad < 3 * patch * patch || n_crops < 1) return hipErrorInvalidValue;
  const int g = image / patch;
  dim3 grid(n_crops * g), block(256);
  PixelNorm nm{{mean3[0], mean3[1], mean3[2]}, {std3[0], std3[1], std3[2]}};
  const size_t lds = (size_t)3 * patch * image * 2;                 // bf16 image rows of one patch row
  const int esz = in_dtype == 0 ? 4 : (in_dtype == 1 ? 2 : 1);
  // 16-B loads need every image row to start on a 16-B boundary: rows are `image` elements apart from an aligned base
  const bool vec = (image * esz) % 16 == 0 && ((uintptr_t)crops & 15) == 0 && lds <= 64 * 1024;
  const bool kp = kpad == (3 * patch * patch + 127) / 128 * 128;
#define LAUNCH_PATCHIFY(T)                                                                                                     \
  do {                                                                                                                         \
    if (vec && kp && patch == 14) hipLaunchKernelGGL((patchify_kernel<T, 14>), grid, block, lds, stream, (const T*)crops, (bf16_t*)a_patch, image, nm); \
    else if (vec && kp && patch == 16) hipLaunchKernelGGL((patchify_kernel<T, 16>), grid, block, lds, stream, (const T*)crops, (bf16_t*)a_patch, image, nm); \
    else if (vec && kp && patch == 32) hipLaunchKernelGGL((patchify_kernel<T, 32>), grid, block, lds, stream, (const T*)crops, (bf16_t*)a_patch, image, nm); \
    else hipLaunchKernelGGL(patchify_scalar_kernel<T>, grid, block, 0, stream, (const T*)crops, (bf16_t*)a_patch, image, patch, kpad, nm); \
  } while (0)
  if (in_dtype == 0) LAUNCH_PATCHIFY(float);
  else if (in_dtype == 1) LAUNCH_PATCHIFY(_Float16);
  else if (in_dtype == 2) LAUNCH_PATCHIFY(uint8_t);
  else return hipErrorInvalidValue;
#undef LAUNCH_PATCHIFY
  return hipGetLastError();
}

hipError_t ce_embed_ln_pre(const void* patch_emb, const float* cls, const float* pos, const float* gamma,
                           const float* beta, void* x, float* stats, int n_crops, int n_tok, int width, int ln_width,
                           float eps, hipStream_t stream) {
  if (width % 8 != 0 || width > LN_MAX_CHUNKS * 512 || ln_width < 1 || ln_width > width) return hipErrorInvalidValue;
  // crops per wave: enough rows to pay for the per-wave constants, enough waves to fill the chip
  const int cpw = n_crops >= 2048 ? 16 : (n_crops >= 256 ? 8 : (n_crops >= 32 ? 2 : 1));
  const int waves_per_tok = (n_crops + cpw - 1) / cpw;
  const long long waves = (long long)n_tok * waves_per_tok;
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  const int nch = (width + 511) / 512;
#define LAUNCH_LN_PRE(N)                                                                                                  \
  hipLaunchKernelGGL(embed_ln_pre_kernel<N>, grid, block, 0, stream, (const bf16_t*)patch_emb, cls, pos, gamma, beta, (bf16_t*)x, \
                     stats, n_crops, n_tok, width, ln_width, eps, cpw, waves_per_tok)
  switch (nch) {
    case 1: LAUNCH_LN_PRE(1); break;
    case 2: LAUNCH_LN_PRE(2); break;
    case 3: LAUNCH_LN_PRE(3); break;
    default: LAUNCH_LN_PRE(4); break;
  }
#undef LAUNCH_LN_PRE
  return hipGetLastError();
}

hipError_t ce_head(const void* x, const float* gamma, const float* beta, const float* proj, float* emb, int n_crops,
                   int n_tok, int width, int ln_width, int embed, float eps, int normalize, hipStream_t stream) {
  if (embed > 1280 || width > 2048 || width % 4 != 0 || ln_width < 1 || ln_width > width) return hipErrorInvalidValue;
  const size_t lds = (size_t)HEAD_CROPS * width * 4 + HEAD_CROPS * 4 * 4;
  hipLaunchKernelGGL(head_kernel, dim3((n_crops + HEAD_CROPS - 1) / HEAD_CROPS), dim3(256), lds, stream,
                     (const bf16_t*)x, gamma, beta, proj, emb, n_crops, n_tok, width, ln_width, embed, eps, normalize);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Clock probe (measurement support, not on the hot path): one wave reads the shader-cycle counter (s_memtime) and the
// constant 100 MHz counter (s_memrealtime) around a short spin; cycles / ticks x 100 = the shader clock in MHz that the
// chip is holding at that moment.  bench.py launches it on a second stream while the encoder runs, so that it lands in
// the gaps between the persistent kernels and reads the clock the board grants UNDER that load (DVFS moves on a
// millisecond scale; MI355X_MICROARCH.md 'DVFS give-back' item 6).  Its result goes to a buffer of its own.
__global__ void clock_probe_kernel(unsigned long long* out, int spin_ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  for (int guard = 0; guard < (1 << 20) && (long long)(r1 - r0) < spin_ticks; ++guard) {   // bounded spin
    __builtin_amdgcn_s_sleep(8);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  r1 = __builtin_amdgcn_s_memrealtime();
  out[0] = c1 - c0;
  out[1] = r1 - r0;
}

hipError_t ce_clock_probe(unsigned long long* out2, int spin_ticks, hipStream_t stream) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, out2, spin_ticks);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Matrix-pipe stream probe for bench.py: every SIMD of the chip (one workgroup of 8 waves per CU, two per SIMD) issues nothing
// but the GEMMs' MFMA -- v_mfma_f32_16x16x32_bf16, or v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales -- on operand
// registers loaded once from the caller's buffer (so the bit patterns, which set the switching power, are the caller's choice).
// No LDS, no memory traffic: the rate this reaches is what the board's power management grants the matrix pipes alone, the
// ceiling any GEMM on this box sits under ("power"); bench.py reports it beside roofline.peak.
template <bool FP8>
__global__ __launch_bounds__(512, 2) void mfma_stream_kernel(const uint32_t* __restrict__ operands, float* __restrict__ sink, long long iters) {
  typedef __attribute__((ext_vector_type(8))) int i32x8_t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t* src = operands + ((size_t)(wave & 3) * 64 + lane) * 32;       // 128 B per lane, 4 wave patterns: 32 KiB
  float total = 0.f;
  if constexpr (FP8) {
    f32x16_t acc[4];
    i32x8_t a[2], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[i][e] = (int)src[i * 8 + e]; b[i][e] = (int)src[16 + i * 8 + e]; }
    for (long long it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        acc[i & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i & 1], b[(i >> 1) & 1], acc[i & 3], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) total += acc[i][0];
  } else {
    f32x4_t acc[8];
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = *(const bf16x8_t*)(src + i * 4); b[i] = *(const bf16x8_t*)(src + 16 + i * 4); }
    for (long long it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i >> 2) & 3], acc[i & 7], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) total += acc[i][0];
  }
  if (total == 12345.678f) sink[0] = total;                    // keeps the accumulators alive; never true for finite operands of this size
}

hipError_t ce_mfma_stream(const void* operands_32k, int fp8, float* sink, long long iters, int n_cu, hipStream_t stream) {
  if (fp8) hipLaunchKernelGGL(mfma_stream_kernel<true>, dim3(n_cu), dim3(512), 0, stream, (const uint32_t*)operands_32k, sink, iters);
  else hipLaunchKernelGGL(mfma_stream_kernel<false>, dim3(n_cu), dim3(512), 0, stream, (const uint32_t*)operands_32k, sink, iters);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Row statistics of the CLS rows only: out[part][i] = in[part][i * row_stride] (float2 = sum, sum of squares).  Feeds the
// LayerNorm-folded GEMMs of the LAST transformer block, which run on the class-token rows alone (capi.hip, run_tower).
__global__ void gather_row_stats_kernel(const float2* __restrict__ in, int in_ld, float2* __restrict__ out, int out_ld, int n,
                                        int row_stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, part = blockIdx.y;
  if (i < n) out[(size_t)part * out_ld + i] = in[(size_t)part * in_ld + (size_t)i * row_stride];
}

// Towers wider than 1024 (ViT-H-14: five 256-column parts): the LayerNorm-folded GEMM's LDS layout holds four parts per row, so the
// producer's parts are added up here, in a fixed order, into part 0 and the consumer is handed one part.
__global__ void combine_row_stats_kernel(float2* __restrict__ stats, int ld, int parts, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float2 a = stats[i];
  for (int p = 1; p < parts; ++p) { const float2 b = stats[(size_t)p * ld + i]; a.x += b.x; a.y += b.y; }
  stats[i] = a;
}

hipError_t ce_combine_row_stats(float* stats, int ld, int parts, int n, hipStream_t stream) {
  if (n < 1 || parts < 1) return hipErrorInvalidValue;
  if (parts == 1) return hipSuccess;
  hipLaunchKernelGGL(combine_row_stats_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (float2*)stats, ld, parts, n);
  return hipGetLastError();
}

hipError_t ce_gather_row_stats(const float* in, int in_ld, float* out, int out_ld, int parts, int n, int row_stride, hipStream_t stream) {
  if (n < 1 || parts < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_row_stats_kernel, dim3((n + 255) / 256, parts), dim3(256), 0, stream, (const float2*)in, in_ld, (float2*)out,
                     out_ld, n, row_stride);
  return hipGetLastError();
}
