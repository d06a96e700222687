// Multi-head self-attention of the ViT tower (K4 of SURVEY.md §2.2): per (crop, head)
//   O = softmax(Q K^T / sqrt(64)) V,  no mask, head dim 64; n_tok <= 288 in one pass (attn_kernel),
//   up to 640 tokens with a single pass over the keys (attn_long_kernel).
// This is the nn.MultiheadAttention step of the open_clip forward the reference reaches through
// /root/reference/utils/embedder.py:98.
//
// One workgroup (4 waves) per (crop, head).  K and V of that head are staged once into LDS
// (rows of 128 B, 16-B chunks XOR-swizzled so that the row-wise K reads (ds_read_b128) and the
// transposed V reads (ds_read_b64_tr_b16) are bank-conflict free).  Each wave owns 32-query blocks.
// Scores are computed TRANSPOSED, S^T = K . Q^T with the 32x32x16 MFMA, so a lane holds one query
// column: the softmax row reductions are lane-local (plus one exchange with lane^32), and the S^T
// accumulator registers feed the P.V MFMA directly as its B operand (guide §3 "An accumulator tile
// as the next MFMA's operand"); V^T comes from the hardware-transposing LDS read.
// The whole key range fits the register file (<= 9 tiles x 16 fp32), so the softmax is exact
// two-pass (true row max), not an online rescale.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

// CLIPENC_PREC_FP8: O can be stored as e4m3 with a static per-channel scale (out_inv[c] = 1/scale, folded into the
// out-projection's weight columns): 4 values -> one dword
__device__ __forceinline__ int pack_fp8x4(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a, -448.0f, 448.0f), __builtin_amdgcn_fmed3f(b, -448.0f, 448.0f), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c, -448.0f, 448.0f), __builtin_amdgcn_fmed3f(d, -448.0f, 448.0f), w, true);
  return w;
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
// two fp32 -> one packed bf16 pair (a single v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}

constexpr unsigned SPIN_LIMIT = 1u << 24;                       // every spin of the persistent kernels below is bounded
__device__ __forceinline__ unsigned lds_load_u32(const char* p) {
  return __atomic_load_n((const unsigned*)p, __ATOMIC_RELAXED);
}

__device__ __forceinline__ int k_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_swz(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4); }

template <int NKT>
__global__ __launch_bounds__(256, 2) void attn_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                      int n_tok, int width, int heads, float scale_log2e,
                                                     const float* __restrict__ out_inv, int q_blocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = NKT * 32;
  char* Ks = smem;
  char* Vs = smem + ROWS * 128;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int crop = blockIdx.x / heads, head = blockIdx.x % heads;
  const size_t ld = (size_t)3 * width;                         // qkv row stride (elements)
  const bf16_t* base = qkv + (size_t)crop * n_tok * ld + head * 64;

  // ---- stage K, V (zero rows beyond n_tok) ----
  for (int idx = tid; idx < ROWS * 8; idx += 256) {
    const int row = idx >> 3, c = idx & 7;
    uint4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
    if (row < n_tok) {
      const bf16_t* g = base + (size_t)row * ld + c * 8;
      kv = *(const uint4*)(g + width);
      vv = *(const uint4*)(g + 2 * width);
    }
    *(uint4*)(Ks + k_swz(row, c)) = kv;
    *(uint4*)(Vs + v_swz(row, c)) = vv;
  }
  __syncthreads();

  const int r = lane & 31, h = lane >> 5;
  const int n_qb = min((n_tok + 31) >> 5, q_blocks);           // q_blocks: only the first 32-query blocks (CLS-only last layer)
  for (int qb = wave; qb < n_qb; qb += 4) {
    // ---- Q fragments straight from global: lane (r,h) holds Q[q0+r][16*step + 8h .. +7] ----
    const int q = qb * 32 + r;
    const bf16_t* qrow = base + (size_t)min(q, n_tok - 1) * ld;
    bf16x8_t qf[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = *(const bf16x8_t*)(qrow + st * 16 + h * 8);

    // ---- S^T tiles: s[kt][reg] = <K[32kt + krow(reg,h)], Q[q]> ----
    f32x16_t s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        bf16x8_t kf = *(const bf16x8_t*)(Ks + k_swz(kt * 32 + r, st * 2 + h));
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s[kt], 0, 0, 0);
      }
    }
    // ---- mask padded keys, row max ----
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (kt == NKT - 1) {                                   // only the last key tile can be partial
          const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= n_tok) s[kt][e] = -INFINITY;
        }
        mx = fmaxf(mx, s[kt][e]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float moff = mx * scale_log2e;

    // ---- P = exp2(s*c - m*c), row sum, and O^T += V^T . P^T ----
    f32x16_t o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
    f32x16_t lacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) lacc[e] = 0.f;
    const u32x4_t ones_w = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // eight bf16 1.0
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_w);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          if (kt < NKT - 1 || s2 == 0 || kt * 32 + 16 < n_tok) {   // wave-uniform; skips an all-padding k-step
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = __builtin_amdgcn_exp2f(fmaf(s[kt][s2 * 8 + j], scale_log2e, -moff));
            const u32x4_t pw = {cvt_pk_bf16(pv[0], pv[1]), cvt_pk_bf16(pv[2], pv[3]), cvt_pk_bf16(pv[4], pv[5]), cvt_pk_bf16(pv[6], pv[7])};
            const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
            lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lacc, 0, 0, 0);   // row sums of the rounded weights
            // element j of lane half h is key 16*s2 + 8*(j>>2) + 4h + (j&3) of the tile
            const int key0 = kt * 32 + s2 * 16 + 4 * h;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
              // transposed read: 16-lane group g reads keys key0.. (+8), d columns dt*32 + 16*(g&1) ..
              const int i = lane & 15, qq = i >> 2, pp = i & 3, g1 = (lane >> 4) & 1;
              const int dcol = dt * 32 + g1 * 16 + pp * 4;      // 4 contiguous d of one key row
              const int ra = key0 + qq, rb = key0 + 8 + qq;
              s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (__attribute__((address_space(3))) s16x4_t*)(Vs + v_swz(ra, dcol >> 3) + (dcol & 7) * 2));
              s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (__attribute__((address_space(3))) s16x4_t*)(Vs + v_swz(rb, dcol >> 3) + (dcol & 7) * 2));
              typedef __attribute__((ext_vector_type(8))) short s16x8_t;
              s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
              o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vv), pf, o[dt], 0, 0, 0);
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const float inv = 1.0f / lacc[0];

    // ---- store O[q][head*64 + d]: reg group g4 holds d = dt*32 + 8*g4 + 4h + (0..3) ----
    if (q < n_tok && out_inv) {
      uint8_t* orow = (uint8_t*)out + ((size_t)crop * n_tok + q) * width + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int col = dt * 32 + g4 * 8 + h * 4;
          const f32x4_t is = *(const f32x4_t*)(out_inv + head * 64 + col);
          *(int*)(orow + col) = pack_fp8x4(o[dt][g4 * 4 + 0] * inv * is[0], o[dt][g4 * 4 + 1] * inv * is[1],
                                           o[dt][g4 * 4 + 2] * inv * is[2], o[dt][g4 * 4 + 3] * inv * is[3]);
        }
    } else if (q < n_tok) {
      bf16_t* orow = out + ((size_t)crop * n_tok + q) * width + head * 64;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          uint2 pk = {pack_bf16x2(o[dt][g4 * 4 + 0] * inv, o[dt][g4 * 4 + 1] * inv),
                      pack_bf16x2(o[dt][g4 * 4 + 2] * inv, o[dt][g4 * 4 + 3] * inv)};
          *(uint2*)(orow + dt * 32 + g4 * 8 + h * 4) = pk;
        }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Head dim 80 (ViT-H-14: width 1280, 16 heads), up to 288 tokens: the attn_kernel scheme -- one workgroup of four waves per (crop, head),
// K and V of the head staged once, exact two-pass softmax with every score tile in registers -- on LDS rows of 256 B (160 B used:
// ten 16-B chunks; chunk ch of row r sits at ch ^ (((r & 3) << 2) | ((r >> 2) & 3)), the image that serves the row-wise K reads and the
// transposing V reads without bank conflicts: guide T10 (b)).  Five k steps of 16 for the scores; O^T in three 32-row tiles of d, the upper
// half of the third (d = 80 .. 95) reads zero columns and is not stored.  (The score tiles of 288 keys + three output
// tiles are ~230 registers), two waves per SIMD: a correct path for the wider tower, not a tuned one.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz256(int row, int chunk) { return row * 256 + ((chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }

// Heads of 97 .. 128 columns (ViT-bigG-14's 104, run as 112): a fourth output tile.  Eight waves of 256 registers spill there (~20 registers at nine key
// tiles) and are still 20 % faster than four waves of 512 that do not (2 048 crops x 257 tokens x 16 heads of 112: 3.12 against 3.90 ms, same bits).
template <int NKT, int HD, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void attn_hd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                         int n_tok, int width, int heads, float scale_log2e,
                                                         const float* __restrict__ out_inv, int q_blocks) {
  static_assert(HD % 16 == 0 && HD > 64 && HD <= 128, "five to eight k steps, three or four output tiles");
  constexpr int NDT = (HD + 31) / 32, NCHUNK = NDT * 4;         // output tiles of 32 columns; 16-B chunks per LDS row that are read
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = NKT * 32, KS = HD / 16, CH = HD / 8;     // k steps; 16-B chunks per row
  char* Ks = smem;
  char* Vs = smem + ROWS * 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int crop = blockIdx.x / heads, head = blockIdx.x % heads;
  const size_t ld = (size_t)3 * width;
  const bf16_t* base = qkv + (size_t)crop * n_tok * ld + head * HD;

  for (int idx = tid; idx < ROWS * NCHUNK; idx += WAVES * 64) {            // chunks 0 .. CH-1: data; CH .. NCHUNK-1: the zero columns the last d tile reads
    const int row = idx / NCHUNK, c = idx - row * NCHUNK;
    uint4 kv = {0, 0, 0, 0}, vv = {0, 0, 0, 0};
    if (row < n_tok && c < CH) {
      const bf16_t* g = base + (size_t)row * ld + c * 8;
      kv = *(const uint4*)(g + width);
      vv = *(const uint4*)(g + 2 * width);
    }
    *(uint4*)(Ks + swz256(row, c)) = kv;
    *(uint4*)(Vs + swz256(row, c)) = vv;
  }
  __syncthreads();

  const int r = lane & 31, h = lane >> 5;
  const int n_qb = min((n_tok + 31) >> 5, q_blocks);
  for (int qb = wave; qb < n_qb; qb += WAVES) {
    const int q = qb * 32 + r;
    const bf16_t* qrow = base + (size_t)min(q, n_tok - 1) * ld;
    bf16x8_t qf[KS];
#pragma unroll
    for (int st = 0; st < KS; ++st) qf[st] = *(const bf16x8_t*)(qrow + st * 16 + h * 8);
    f32x16_t s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        const bf16x8_t kf = *(const bf16x8_t*)(Ks + swz256(kt * 32 + r, st * 2 + h));
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s[kt], 0, 0, 0);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        if (kt == NKT - 1) {
          const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= n_tok) s[kt][e] = -INFINITY;
        }
        mx = fmaxf(mx, s[kt][e]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float moff = mx * scale_log2e;
    f32x16_t o[NDT], lacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) o[dt][e] = 0.f;
      lacc[e] = 0.f;
    }
    const u32x4_t ones_w = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_w);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (kt < NKT - 1 || s2 == 0 || kt * 32 + 16 < n_tok) {   // wave-uniform; skips an all-padding k step
          float pv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pv[j] = __builtin_amdgcn_exp2f(fmaf(s[kt][s2 * 8 + j], scale_log2e, -moff));
          const u32x4_t pw = {cvt_pk_bf16(pv[0], pv[1]), cvt_pk_bf16(pv[2], pv[3]), cvt_pk_bf16(pv[4], pv[5]), cvt_pk_bf16(pv[6], pv[7])};
          const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
          lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lacc, 0, 0, 0);
          const int key0 = kt * 32 + s2 * 16 + 4 * h;
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            const int i = lane & 15, qq = i >> 2, pp = i & 3, g1 = (lane >> 4) & 1;
            const int chunk = dt * 4 + g1 * 2 + (pp >> 1);        // 16-B chunk of d = 32 dt + 16 g1 + 4 pp .. + 3
            const int ra = key0 + qq, rb = key0 + 8 + qq;
            s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(Vs + swz256(ra, chunk) + (pp & 1) * 8));
            s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(Vs + swz256(rb, chunk) + (pp & 1) * 8));
            typedef __attribute__((ext_vector_type(8))) short s16x8_t;
            s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vv), pf, o[dt], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const float inv = 1.0f / lacc[0];
    if (q < n_tok && out_inv) {
      uint8_t* orow = (uint8_t*)out + ((size_t)crop * n_tok + q) * width + head * HD;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int col = dt * 32 + g4 * 8 + h * 4;
          if (col < HD) {
            const f32x4_t is = *(const f32x4_t*)(out_inv + head * HD + col);
            *(int*)(orow + col) = pack_fp8x4(o[dt][g4 * 4 + 0] * inv * is[0], o[dt][g4 * 4 + 1] * inv * is[1],
                                             o[dt][g4 * 4 + 2] * inv * is[2], o[dt][g4 * 4 + 3] * inv * is[3]);
          }
        }
    } else if (q < n_tok) {
      bf16_t* orow = out + ((size_t)crop * n_tok + q) * width + head * HD;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int col = dt * 32 + g4 * 8 + h * 4;
          if (col < HD)
            *(uint2*)(orow + col) = uint2{pack_bf16x2(o[dt][g4 * 4 + 0] * inv, o[dt][g4 * 4 + 1] * inv),
                                          pack_bf16x2(o[dt][g4 * 4 + 2] * inv, o[dt][g4 * 4 + 3] * inv)};
        }
    }
  }
}

#ifndef HD_WAVES_WIDE
#define HD_WAVES_WIDE 8            // heads of 97 .. 128 columns (4: one wave per SIMD, no spills, slower)
#endif
template <int NKT, int HD>
hipError_t launch_attn_hd(const bf16_t* qkv, bf16_t* out, int n_crops, int n_tok, int width, int heads,
                          const float* out_inv, int q_blocks, hipStream_t stream) {
  constexpr int WAVES = HD <= 96 ? 8 : HD_WAVES_WIDE;           // two waves per SIMD (the kernel needs ~230 registers at three output tiles)
  const int lds = NKT * 32 * 256 * 2;
  static DeviceKernelSetup setup;
  if (hipError_t e = setup.ensure((const void*)attn_hd_kernel<NKT, HD, WAVES>, lds, nullptr); e != hipSuccess) return e;
  const float scale_log2e = 1.44269504088896340736f / sqrtf((float)HD);
  hipLaunchKernelGGL((attn_hd_kernel<NKT, HD, WAVES>), dim3(n_crops * heads), dim3(WAVES * 64), lds, stream, qkv, out, n_tok, width, heads, scale_log2e, out_inv,
                     q_blocks);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Long-sequence kernel (289..640 tokens, e.g. ViT-L-14-336: 577 -- the reference's default model,
// /root/reference/_1_embed_with_CLIP.py:190).  One workgroup of NW waves per (crop, head); K and V of that head take up to 160 KiB of LDS
// (the same images, fragments and MFMA orientation as attn_kernel); 19 x 16 score registers do not fit the register file, so the keys are
// walked ONCE and in order with a single-pass softmax: a row's reference m is the maximum of its FIRST key tile, every weight is
// exp2((s - m) c) -- not <= 1 any more, but bf16 and fp32 have the exponent range -- and only a score more than 64 / c above m (a weight
// beyond 2^64, looked for among the weights themselves) rescales the row's state to the new maximum: a wave-uniform branch that multiplies in place (through asm "+v": as plain
// C++ the conditional writes to the accumulators became ~50 register copies per tile) and that ordinary inputs never take.  The softmax
// is the same function of the scores; sums and O stay in fp32.  (Rounds 1-4 walked the keys in chunks of 7 tiles with a two-pass softmax
// per chunk and a rescale of O between chunks: 112 score registers, 8 waves.  Round 5, 480 crops x 577 tokens, same box, interleaved:
// that kernel 1.108-1.146 ms per layer, this one with 8 waves 1.028-1.071, with 12 waves -- three per SIMD, 152 registers -- 0.999-1.035:
// -10 %; with 16 waves it spills.)  Row sums are fp32 adds of the weights (not an MFMA against ones: two of ten MFMAs per tile); the
// next block's Q is loaded a block ahead; the K fragments of the next tile are read behind the score MFMAs of this one.
// ---------------------------------------------------------------------------------------------
#ifndef LP_THR                   // the weight beyond which a row's state is moved to a new reference (developer builds lower it to force the path)
#define LP_THR 0x1p64f
#endif
template <int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void attn_long_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                  int n_tok, int width, int heads, float scale_log2e, int nkt,
                                                                  const float* __restrict__ out_inv, int q_blocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int rows = nkt * 32;
  char* Ks = smem;
  char* Vs = smem + rows * 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int crop = blockIdx.x / heads, head = blockIdx.x % heads;
  const size_t ld = (size_t)3 * width;
  const bf16_t* base = qkv + (size_t)crop * n_tok * ld + head * 64;
  const int r = lane & 31, h = lane >> 5;
  const int n_qb = min((n_tok + 31) >> 5, q_blocks);
  auto q_load = [&](int qb, bf16x8_t (&qf)[4]) {
    const bf16_t* qrow = base + (size_t)min(qb * 32 + r, n_tok - 1) * ld;
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = *(const bf16x8_t*)(qrow + st * 16 + h * 8);
  };
  bf16x8_t qn[4];
  if (wave < n_qb) q_load(wave, qn);                             // (in flight beside the K / V pieces)
  {
    const char* tb = (const char*)base;
    const unsigned ldb = (unsigned)(ld * 2);
    for (int j = wave; j < rows / 8; j += NW) {
      const int row = 8 * j + (lane >> 3);
      const unsigned rb = (unsigned)min(row, n_tok - 1) * ldb;
      const int ck = (lane & 7) ^ ((row >> 1) & 7);
      const int cv = (lane & 7) ^ (((row >> 1) & 1) << 2);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + (size_t)width * 2 + (rb + ck * 16)),
                                       (__attribute__((address_space(3))) void*)(Ks + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + (size_t)width * 4 + (rb + cv * 16)),
                                       (__attribute__((address_space(3))) void*)(Vs + j * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // K fragment of key tile c, k step st: row 32 c + r, chunk 2 st + h -- the swizzle term (row >> 1) & 7 does not depend on c
  const char* kb0 = Ks + k_swz(r, 0 + h); const char* kb1 = Ks + k_swz(r, 2 + h);
  const char* kb2 = Ks + k_swz(r, 4 + h); const char* kb3 = Ks + k_swz(r, 6 + h);
  // V^T fragment addresses (attn_stream_kernel): rows 16 j + 4 h + q and + 8 of key step j, a constant 2 KiB per step
  const int vi = lane & 15, vq = vi >> 2, vp = vi & 3, vg1 = (lane >> 4) & 1;
  const char* vbase0 = Vs + v_swz(4 * h + vq, (vg1 * 16 + vp * 4) >> 3) + ((vg1 * 16 + vp * 4) & 7) * 2;
  const char* vbase1 = Vs + v_swz(4 * h + vq, (32 + vg1 * 16 + vp * 4) >> 3) + ((vg1 * 16 + vp * 4) & 7) * 2;
  auto v_frag = [&](int j, const char* vb) -> bf16x8_t {
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + j * 2048));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + j * 2048 + 1024));
    s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, vv);
  };
  auto row_max = [&](const f32x16_t& s) -> float {              // compiler-visible (hipcc places the MFMA -> VALU wait states); tile 0 and the rare path only
    float a = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
    for (int e = 4; e < 16; e += 4) a = fmaxf(a, fmaxf(fmaxf(s[e], s[e + 1]), fmaxf(s[e + 2], s[e + 3])));
    return a;
  };
  // largest weight of a tile as an INTEGER maximum of the bit patterns (exp2 results are >= 0 or +inf, where the unsigned order is the float
  // order): compiler-visible v_max3_u32 -- fmaxf would put a canonicalising v_max in front of every value, and an asm statement on the
  // v_exp_f32 results would read them without the wait state hipcc places between a transcendental and its first reader
  auto weight_max_bits = [&](const float (&pv)[16]) -> unsigned {
    unsigned m = __float_as_uint(pv[0]);
#pragma unroll
    for (int j = 1; j < 16; ++j) m = max(m, __float_as_uint(pv[j]));
    return m;
  };
  const bool two_last = (nkt - 1) * 32 + 16 < n_tok;             // the last key tile's second 16-key step holds real keys

  for (int qb = wave; qb < n_qb; qb += NW) {
    bf16x8_t qf[4], kf[4], pf[2];
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = qn[st];
    if (qb + NW < n_qb) q_load(qb + NW, qn);                     // the next block's queries: in flight for the whole of this block
    f32x16_t o0 = zero16, o1 = zero16, s;
    float lsum = 0.f, m_ref = 0.f, moff = 0.f;
    kf[0] = *(const bf16x8_t*)kb0; kf[1] = *(const bf16x8_t*)kb1; kf[2] = *(const bf16x8_t*)kb2; kf[3] = *(const bf16x8_t*)kb3;
    for (int c = 0; c < nkt; ++c) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], zero16, 0, 0, 0);
#pragma unroll
      for (int st = 1; st < 4; ++st) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[st], qf[st], s, 0, 0, 0);
      // behind the MFMAs: V^T of this tile, K of the next one (past the end: a re-read of the last tile)
      const bf16x8_t va0 = v_frag(2 * c, vbase0), vb0 = v_frag(2 * c, vbase1);
      const bf16x8_t va1 = v_frag(2 * c + 1, vbase0), vb1 = v_frag(2 * c + 1, vbase1);
      {
        const int ko = min(c + 1, nkt - 1) * 4096;
        kf[0] = *(const bf16x8_t*)(kb0 + ko); kf[1] = *(const bf16x8_t*)(kb1 + ko);
        kf[2] = *(const bf16x8_t*)(kb2 + ko); kf[3] = *(const bf16x8_t*)(kb3 + ko);
      }
      if (c == nkt - 1) {                                        // the last key tile may be partial
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = c * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= n_tok) s[e] = -INFINITY;
        }
      }
      if (c == 0) {
        m_ref = row_max(s);                                      // tile 0 holds 32 real keys (n_tok > 288)
        m_ref = fmaxf(m_ref, __shfl_xor(m_ref, 32));
        moff = m_ref * scale_log2e;
      }
      float pv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) pv[j] = __builtin_amdgcn_exp2f(fmaf(s[j], scale_log2e, -moff));
      // The overflow guard looks at the WEIGHTS (VALU results; an asm statement on the score registers themselves would read them
      // without the wait states hipcc inserts between an MFMA and the first compiler-visible reader of its result).
      if (c > 0 && __builtin_amdgcn_ballot_w64(weight_max_bits(pv) > __float_as_uint(LP_THR)) != 0ull) {
        // (rare) a score of this tile towers over the reference: move the row's state to the new maximum, in place, and take this
        // tile's weights again.  Every weight at the old reference has been multiplied into o / lsum already.
        float mx = row_max(s);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_ref, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_ref - m_new) * scale_log2e);
        // (in place through asm; the s_nop: alpha comes out of v_exp_f32, and hipcc does not place the transcendental -> VALU wait
        //  state in front of an asm statement -- without it the first product read a stale alpha in some lanes)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %0" : "+v"(o0[e]) : "v"(alpha));
          asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %0" : "+v"(o1[e]) : "v"(alpha));
        }
        asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %0" : "+v"(lsum) : "v"(alpha));
        m_ref = m_new;
        moff = m_ref * scale_log2e;
#pragma unroll
        for (int j = 0; j < 16; ++j) pv[j] = __builtin_amdgcn_exp2f(fmaf(s[j], scale_log2e, -moff));
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) lsum += pv[j];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const u32x4_t pw = {cvt_pk_bf16(pv[s2 * 8 + 0], pv[s2 * 8 + 1]), cvt_pk_bf16(pv[s2 * 8 + 2], pv[s2 * 8 + 3]),
                            cvt_pk_bf16(pv[s2 * 8 + 4], pv[s2 * 8 + 5]), cvt_pk_bf16(pv[s2 * 8 + 6], pv[s2 * 8 + 7])};
        pf[s2] = __builtin_bit_cast(bf16x8_t, pw);
      }
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va0, pf[0], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb0, pf[0], o1, 0, 0, 0);
      if (c < nkt - 1 || two_last) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va1, pf[1], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb1, pf[1], o1, 0, 0, 0);
      }
    }
    lsum += __shfl_xor(lsum, 32);                                // the row's other 16 keys of every tile sit in lane ^ 32
    const float inv = __builtin_amdgcn_rcpf(lsum);
    const int q = qb * 32 + r;
    if (q < n_tok && out_inv) {
      uint8_t* orow = (uint8_t*)out + ((size_t)crop * n_tok + q) * width + head * 64;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int col = g4 * 8 + h * 4;
        const f32x4_t is0 = *(const f32x4_t*)(out_inv + head * 64 + col), is1 = *(const f32x4_t*)(out_inv + head * 64 + 32 + col);
        *(int*)(orow + col) = pack_fp8x4(o0[g4 * 4 + 0] * inv * is0[0], o0[g4 * 4 + 1] * inv * is0[1],
                                         o0[g4 * 4 + 2] * inv * is0[2], o0[g4 * 4 + 3] * inv * is0[3]);
        *(int*)(orow + 32 + col) = pack_fp8x4(o1[g4 * 4 + 0] * inv * is1[0], o1[g4 * 4 + 1] * inv * is1[1],
                                              o1[g4 * 4 + 2] * inv * is1[2], o1[g4 * 4 + 3] * inv * is1[3]);
      }
    } else if (q < n_tok) {
      bf16_t* orow = out + ((size_t)crop * n_tok + q) * width + head * 64;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        *(uint2*)(orow + g4 * 8 + h * 4) = uint2{pack_bf16x2(o0[g4 * 4 + 0] * inv, o0[g4 * 4 + 1] * inv),
                                                 pack_bf16x2(o0[g4 * 4 + 2] * inv, o0[g4 * 4 + 3] * inv)};
        *(uint2*)(orow + 32 + g4 * 8 + h * 4) = uint2{pack_bf16x2(o1[g4 * 4 + 0] * inv, o1[g4 * 4 + 1] * inv),
                                                      pack_bf16x2(o1[g4 * 4 + 2] * inv, o1[g4 * 4 + 3] * inv)};
      }
    }
  }
}

hipError_t launch_attn_long(const bf16_t* qkv, bf16_t* out, int n_crops, int n_tok, int width, int heads,
                            const float* out_inv, int q_blocks, hipStream_t stream) {
  constexpr int NW = 12;                                         // three waves per SIMD
  const int nkt = (n_tok + 31) / 32;
  const int lds = nkt * 32 * 128 * 2;
  if (lds > 160 * 1024 || n_tok <= 288) return hipErrorInvalidValue;   // (tile 0 must hold 32 real keys; shorter sequences have their own kernels)
  static DeviceKernelSetup setup;                                // per device: LDS opt-in (common.h)
  if (hipError_t e = setup.ensure((const void*)attn_long_kernel<NW>, 160 * 1024, nullptr); e != hipSuccess) return e;
  const float scale_log2e = 0.125f * 1.44269504088896340736f;
  hipLaunchKernelGGL((attn_long_kernel<NW>), dim3(n_crops * heads), dim3(NW * 64), lds, stream, qkv, out, n_tok, width, heads,
                     scale_log2e, nkt, out_inv, q_blocks);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Long sequences, streaming form (289..608 tokens, >= 64 tasks, every query block asked for: ViT-L-14-336 in the tower).
// attn_long_kernel above loads a head's K | V (148 KB at 577 tokens) and only then starts: a fifth of its time nothing is computed, and
// a second task's K | V do not fit beside the first's.  Here ONE persistent workgroup per CU (NCW compute waves + 1 loader wave) walks a
// contiguous range of (crop, head) tasks and the next task's key tiles REPLACE the current task's IN PLACE, tile by tile, as they die:
//   * the 32-query blocks of consecutive tasks form one stream handed out in order from an LDS counter (as in attn_stream_kernel); a
//     block sweeps the key tiles 0 .. nkt-1 once (the single-pass softmax of attn_long_kernel, the same arithmetic in the same order);
//   * done[c] counts the block sweeps that have passed tile c (monotonic over tasks; the count is taken after the sweep's last read of
//     the tile -- the LDS executes a wave's instructions in order, so the add lands behind the reads); when all n_qb blocks of task k
//     have passed tile c the loader fetches tile c of task k + 1 into the same 8 KiB by LDS-DMA, and publishes `landed`, the number of
//     tiles of the stream that are readable, behind a counted vmcnt wait (four tiles stay in flight);
//   * a block of task k + 1 waits for `landed` tile by tile: a wave that runs out of blocks of task k starts on task k + 1 behind the
//     sweeps still going on -- the load of a task hides behind the last sweeps of the task before, and HBM sees a steady stream.
//   Every spin is bounded; no s_barrier in the steady state.  Deadlock-free: blocks are grabbed in stream order, every grabbed block of
//   task k is being swept by a wave, and tile c of task k stays put until all of them have passed it.
// The overflow guard of the single-pass softmax (a weight beyond 2^64 moves the row's reference: attn_long_kernel looks for one in every
// tile, 6 % of its time) is taken ONCE per block here: a row sum that is not inside [2^-100, 2^100] -- a weight overflowed, or every weight
// underflowed -- flags the block in an LDS bitmap instead of storing it, and when the stream has drained the workgroup loads each flagged
// task's K | V whole and sweeps the flagged blocks again WITH the per-tile guard (the exact path, the code of attn_long_kernel).  Ordinary
// inputs never flag; tests/test_gpu_parity.py::test_long_attention_single_pass_softmax_rescales_where_it_must forces it.
// ---------------------------------------------------------------------------------------------
template <int NCW>
__global__ __launch_bounds__((NCW + 1) * 64, (NCW + 1) / 4) void attn_long_stream_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int n_tok, int width, int heads, float scale_log2e, int nkt, int n_tasks,
    const float* __restrict__ out_inv) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = NCW + 1;
#ifndef LS_DFLIGHT
#define LS_DFLIGHT 2
#endif
#ifndef LS_PRIO
#define LS_PRIO 3
#endif
#ifndef LS_EARLY_MARK
#define LS_EARLY_MARK 1
#endif
  constexpr int D_FLIGHT = LS_DFLIGHT;                           // key tiles (8 pieces each) the loader keeps in flight at most
  const int rows = nkt * 32;
  char* Ks = smem;
  char* Vs = smem + rows * 128;
  char* ctrl = smem + 2 * rows * 128;                            // [0] landed, [4] next block, [64 + 4 c] done[c], [256 + 4 k] redo bits of task k
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x, wg = blockIdx.x;
  const int t0 = (int)(((long long)n_tasks * wg) / G), t1 = (int)(((long long)n_tasks * (wg + 1)) / G);
  const int ntask = t1 - t0;                                     // host guarantees 1 <= ntask <= 496
  const int n_qb = nkt;                                          // every query block (the host routes other requests to attn_long_kernel)
  const int total_blocks = ntask * n_qb;
  const size_t ld = (size_t)3 * width;
  const unsigned ldb = (unsigned)(ld * 2);
  const int r = lane & 31, h = lane >> 5;

  for (int i = tid; i < 1024; i += NW * 64) ((unsigned*)ctrl)[i] = 0u;
  __syncthreads();

  // K / V pieces j0 .. j0 + n - 1 (8 rows x 128 B each) of the task at `tb` into their places (rows beyond n_tok: the last row again, masked)
  auto issue_pieces = [&](const char* tb, int j0, int n) {
    for (int j = j0; j < j0 + n; ++j) {
      const int row = 8 * j + (lane >> 3);
      const unsigned rb = (unsigned)min(row, n_tok - 1) * ldb;
      const int ck = (lane & 7) ^ ((row >> 1) & 7);
      const int cv = (lane & 7) ^ (((row >> 1) & 1) << 2);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + (size_t)width * 2 + (rb + ck * 16)),
                                       (__attribute__((address_space(3))) void*)(Ks + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + (size_t)width * 4 + (rb + cv * 16)),
                                       (__attribute__((address_space(3))) void*)(Vs + j * 1024), 16, 0, 0);
    }
  };

  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const char* kb0 = Ks + k_swz(r, 0 + h); const char* kb1 = Ks + k_swz(r, 2 + h);
  const char* kb2 = Ks + k_swz(r, 4 + h); const char* kb3 = Ks + k_swz(r, 6 + h);
  const int vi = lane & 15, vq = vi >> 2, vp = vi & 3, vg1 = (lane >> 4) & 1;
  const char* vbase0 = Vs + v_swz(4 * h + vq, (vg1 * 16 + vp * 4) >> 3) + ((vg1 * 16 + vp * 4) & 7) * 2;
  const char* vbase1 = Vs + v_swz(4 * h + vq, (32 + vg1 * 16 + vp * 4) >> 3) + ((vg1 * 16 + vp * 4) & 7) * 2;
  auto v_frag = [&](int j, const char* vb) -> bf16x8_t {
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + j * 2048));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + j * 2048 + 1024));
    s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, vv);
  };
  auto row_max = [&](const f32x16_t& s) -> float {
    float a = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
    for (int e = 4; e < 16; e += 4) a = fmaxf(a, fmaxf(fmaxf(s[e], s[e + 1]), fmaxf(s[e + 2], s[e + 3])));
    return a;
  };
  auto weight_max_bits = [&](const float (&pv)[16]) -> unsigned {
    unsigned m = __float_as_uint(pv[0]);
#pragma unroll
    for (int j = 1; j < 16; ++j) m = max(m, __float_as_uint(pv[j]));
    return m;
  };
  const bool two_last = (nkt - 1) * 32 + 16 < n_tok;
  unsigned seen = 0;                                             // the last value of `landed` this wave has read (wave-uniform)
#ifdef LS_COUNT
  unsigned long long cnt_wait = 0, cnt_w0 = 0, cnt_w2 = 0, cnt_t0 = __builtin_amdgcn_s_memtime();
#endif
  auto wait_landed = [&](unsigned need) {
#ifdef LS_DBG
    return;
#endif
    if (seen < need) {
#ifdef LS_COUNT
      const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
      unsigned spins = 0;
      do {
        seen = __builtin_amdgcn_readfirstlane(lds_load_u32(ctrl));
        if (seen >= need) break;
        __builtin_amdgcn_s_sleep(1);
      } while (++spins < SPIN_LIMIT);
#ifdef LS_COUNT
      { const unsigned long long dw = __builtin_amdgcn_s_memtime() - w0; cnt_wait += dw; const unsigned cc = (need - 1) % (unsigned)nkt; if (cc < 2) cnt_w0 += dw; else if (cc >= (unsigned)nkt - 4) cnt_w2 += dw; }
#endif
    }
    asm volatile("" ::: "memory");
  };

  // One 32-query block of (crop, head) over all key tiles.  STREAM: the tiles are waited for and released one by one, the overflow guard is the
  // caller's look at the returned row sum.  !STREAM: the task's K | V are resident, per-tile guard (the body of attn_long_kernel).
  auto sweep = [&](auto stream_tag, const bf16x8_t (&qf)[4], unsigned gi0, int crop, int head, int qb) -> float {
    constexpr bool STREAM = decltype(stream_tag)::value;
    bf16x8_t kf[4], pf[2];
    f32x16_t o0 = zero16, o1 = zero16, s;
    float lsum = 0.f, m_ref = 0.f, moff = 0.f;
    if constexpr (STREAM) wait_landed(gi0 + 1);
    kf[0] = *(const bf16x8_t*)kb0; kf[1] = *(const bf16x8_t*)kb1; kf[2] = *(const bf16x8_t*)kb2; kf[3] = *(const bf16x8_t*)kb3;
    for (int c = 0; c < nkt; ++c) {
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], zero16, 0, 0, 0);
#pragma unroll
      for (int st = 1; st < 4; ++st) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[st], qf[st], s, 0, 0, 0);
      const bf16x8_t va0 = v_frag(2 * c, vbase0), vb0 = v_frag(2 * c, vbase1);
      const bf16x8_t va1 = v_frag(2 * c + 1, vbase0), vb1 = v_frag(2 * c + 1, vbase1);
      if constexpr (STREAM && LS_EARLY_MARK) {
        // the sweep's last reads of tile c are in this wave's LDS queue: an add issued behind them executes behind them
        asm volatile("" ::: "memory");
        if (lane == 0) __atomic_fetch_add((unsigned*)(ctrl + 64 + 4 * c), 1u, __ATOMIC_RELAXED);
        asm volatile("" ::: "memory");
      }
      {
        const int cn = min(c + 1, nkt - 1);
        if constexpr (STREAM) wait_landed(gi0 + cn + 1);
        const int ko = cn * 4096;
        kf[0] = *(const bf16x8_t*)(kb0 + ko); kf[1] = *(const bf16x8_t*)(kb1 + ko);
        kf[2] = *(const bf16x8_t*)(kb2 + ko); kf[3] = *(const bf16x8_t*)(kb3 + ko);
      }
      if (c == nkt - 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = c * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= n_tok) s[e] = -INFINITY;
        }
      }
      if (c == 0) {
        m_ref = row_max(s);
        m_ref = fmaxf(m_ref, __shfl_xor(m_ref, 32));
        moff = m_ref * scale_log2e;
      }
      float pv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) pv[j] = __builtin_amdgcn_exp2f(fmaf(s[j], scale_log2e, -moff));
      if constexpr (!STREAM) {
        if (c > 0 && __builtin_amdgcn_ballot_w64(weight_max_bits(pv) > __float_as_uint(LP_THR)) != 0ull) {
          float mx = row_max(s);
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          const float m_new = fmaxf(m_ref, mx);
          const float alpha = __builtin_amdgcn_exp2f((m_ref - m_new) * scale_log2e);
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %0" : "+v"(o0[e]) : "v"(alpha));
            asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %0" : "+v"(o1[e]) : "v"(alpha));
          }
          asm volatile("s_nop 1\n\tv_mul_f32 %0, %1, %0" : "+v"(lsum) : "v"(alpha));
          m_ref = m_new;
          moff = m_ref * scale_log2e;
#pragma unroll
          for (int j = 0; j < 16; ++j) pv[j] = __builtin_amdgcn_exp2f(fmaf(s[j], scale_log2e, -moff));
        }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) lsum += pv[j];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const u32x4_t pw = {cvt_pk_bf16(pv[s2 * 8 + 0], pv[s2 * 8 + 1]), cvt_pk_bf16(pv[s2 * 8 + 2], pv[s2 * 8 + 3]),
                            cvt_pk_bf16(pv[s2 * 8 + 4], pv[s2 * 8 + 5]), cvt_pk_bf16(pv[s2 * 8 + 6], pv[s2 * 8 + 7])};
        pf[s2] = __builtin_bit_cast(bf16x8_t, pw);
      }
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va0, pf[0], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb0, pf[0], o1, 0, 0, 0);
      if (c < nkt - 1 || two_last) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va1, pf[1], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb1, pf[1], o1, 0, 0, 0);
      }
      if constexpr (STREAM && !LS_EARLY_MARK) {
        // this sweep's reads of tile c (its K fragments an iteration ago, its V fragments above) are in the LDS queue ahead of the add
#ifndef LS_DBG
        asm volatile("" ::: "memory");
        if (lane == 0) __atomic_fetch_add((unsigned*)(ctrl + 64 + 4 * c), 1u, __ATOMIC_RELAXED);
#endif
      }
    }
    lsum += __shfl_xor(lsum, 32);
    if constexpr (STREAM) {
      // the single-pass weights are trustworthy while no weight overflowed and not all of them underflowed; otherwise: the exact path later
      if (__builtin_amdgcn_ballot_w64(!(lsum >= 0x1p-100f && lsum <= 0x1p100f)) != 0ull) return -1.0f;
    }
    const float inv = __builtin_amdgcn_rcpf(lsum);
    const int q = qb * 32 + r;
    if (q < n_tok && out_inv) {
      uint8_t* orow = (uint8_t*)out + ((size_t)crop * n_tok + q) * width + head * 64;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int col = g4 * 8 + h * 4;
        const f32x4_t is0 = *(const f32x4_t*)(out_inv + head * 64 + col), is1 = *(const f32x4_t*)(out_inv + head * 64 + 32 + col);
        *(int*)(orow + col) = pack_fp8x4(o0[g4 * 4 + 0] * inv * is0[0], o0[g4 * 4 + 1] * inv * is0[1],
                                         o0[g4 * 4 + 2] * inv * is0[2], o0[g4 * 4 + 3] * inv * is0[3]);
        *(int*)(orow + 32 + col) = pack_fp8x4(o1[g4 * 4 + 0] * inv * is1[0], o1[g4 * 4 + 1] * inv * is1[1],
                                              o1[g4 * 4 + 2] * inv * is1[2], o1[g4 * 4 + 3] * inv * is1[3]);
      }
    } else if (q < n_tok) {
      bf16_t* orow = out + ((size_t)crop * n_tok + q) * width + head * 64;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        *(uint2*)(orow + g4 * 8 + h * 4) = uint2{pack_bf16x2(o0[g4 * 4 + 0] * inv, o0[g4 * 4 + 1] * inv),
                                                 pack_bf16x2(o0[g4 * 4 + 2] * inv, o0[g4 * 4 + 3] * inv)};
        *(uint2*)(orow + 32 + g4 * 8 + h * 4) = uint2{pack_bf16x2(o1[g4 * 4 + 0] * inv, o1[g4 * 4 + 1] * inv),
                                                      pack_bf16x2(o1[g4 * 4 + 2] * inv, o1[g4 * 4 + 3] * inv)};
      }
    }
    return 1.0f;
  };
  auto q_load = [&](int crop, int head, int qb, bf16x8_t (&qf)[4]) {
    const bf16_t* qrow = qkv + (size_t)crop * n_tok * ld + head * 64 + (size_t)min(qb * 32 + r, n_tok - 1) * ld;
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = *(const bf16x8_t*)(qrow + st * 16 + h * 8);
  };

  if (wave == NCW) {
    // ------------------------------------ loader wave ------------------------------------
    // Up to D_FLIGHT tiles in flight.  The wave never sits in a wait for its newest tile while an older one could be published or a
    // freed tile refilled: when the next tile of the stream is not free yet it retires the OLDEST tile in flight (vmcnt retires in
    // order: a counted wait for all but the younger ones), publishes it and looks again.  (Waiting for everything in flight before
    // looking at done[] again made the stream one tile per memory latency, slower than the sweeps free them.)
    if (LS_PRIO) __builtin_amdgcn_s_setprio(LS_PRIO);
    unsigned issued = 0, published = 0;
#ifdef LS_COUNT
    unsigned long long l_issue = 0, l_block = 0, l_idle = 0, l_t;
#define LS_T0() l_t = __builtin_amdgcn_s_memtime()
#define LS_ACC(x) x += __builtin_amdgcn_s_memtime() - l_t
#else
#define LS_T0()
#define LS_ACC(x)
#endif
    int k = 0, c = 0;                                            // the next tile of the stream: tile c of task k
    const char* tb = nullptr;
    unsigned idle = 0;
    while (k < ntask || published < issued) {
      bool can_issue = false;
      if (k < ntask && issued - published < (unsigned)D_FLIGHT) {
#ifdef LS_DBG
        can_issue = LS_DBG < 2 || k == 0;
        if (!can_issue) { if (++c == nkt) { c = 0; ++k; } continue; }
#else
        can_issue = k == 0 || lds_load_u32(ctrl + 64 + 4 * c) >= (unsigned)(n_qb * k);   // every block of task k - 1 has passed tile c
#endif
      }
      if (can_issue) {
        if (c == 0) {
          const int t = t0 + k;
          const int crop = t / heads, head = t - crop * heads;
          tb = (const char*)(qkv + (size_t)crop * n_tok * ld + head * 64);
        }
        asm volatile("" ::: "memory");
        LS_T0();
        issue_pieces(tb, 4 * c, 4);
        LS_ACC(l_issue);
        ++issued;
        if (++c == nkt) { c = 0; ++k; }
        idle = 0;
      } else if (published < issued) {
        LS_T0();
        switch (issued - published) {                            // all but the (in flight - 1) youngest tiles have landed
          case 1: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
          case 3: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
          case 4: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
          case 5: asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); break;
          case 6: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(48)" ::: "memory"); break;
        }
        LS_ACC(l_block);
        ++published;
        if (lane == 0) __atomic_store_n((unsigned*)ctrl, published, __ATOMIC_RELAXED);
      } else {
        if (++idle >= SPIN_LIMIT) break;                         // (bounded: a tile that is never freed ends the stream)
        LS_T0();
        __builtin_amdgcn_s_sleep(2);
        LS_ACC(l_idle);
      }
    }
#ifdef LS_COUNT
    if (lane == 0) { unsigned long long* so = (unsigned long long*)out + 8192 + (size_t)wg * 4; so[0] = l_issue; so[1] = l_block; so[2] = l_idle; }
#endif
  } else {
    // ------------------------------------ compute waves ------------------------------------
    auto grab = [&]() -> int {
      unsigned v = 0;
      if (lane == 0) v = __atomic_fetch_add((unsigned*)(ctrl + 4), 1u, __ATOMIC_RELAXED);
      return (int)__builtin_amdgcn_readfirstlane(v);
    };
    auto where = [&](int g, int& k, int& qb, int& crop, int& head) {
      k = g / n_qb; qb = g - k * n_qb;
      const int t = t0 + k;
      crop = t / heads; head = t - crop * heads;
    };
    int g = grab();
    bf16x8_t qn[4];
    {
      int k, qb, crop, head;
      where(min(g, total_blocks - 1), k, qb, crop, head);
      q_load(crop, head, qb, qn);
    }
    while (g < total_blocks) {
      const int g_next = grab();
      int k, qb, crop, head;
      where(g, k, qb, crop, head);
      bf16x8_t qf[4];
#pragma unroll
      for (int st = 0; st < 4; ++st) qf[st] = qn[st];
      {                                                          // the next block's queries: in flight for the whole of this sweep
        int k2, qb2, crop2, head2;
        where(min(g_next, total_blocks - 1), k2, qb2, crop2, head2);
        q_load(crop2, head2, qb2, qn);
      }
      const float ok = sweep(std::true_type{}, qf, (unsigned)(k * nkt), crop, head, qb);
      if (ok < 0.f && lane == 0) __atomic_fetch_or((unsigned*)(ctrl + 256 + 4 * k), 1u << qb, __ATOMIC_RELAXED);
      g = g_next;
    }
  }
#ifdef LS_COUNT
  if (lane == 0) { unsigned long long* so = (unsigned long long*)out + ((size_t)wg * NW + wave) * 2; so[0] = cnt_wait; so[1] = __builtin_amdgcn_s_memtime() - cnt_t0; unsigned long long* s2 = (unsigned long long*)out + 16384 + ((size_t)wg * NW + wave) * 2; s2[0] = cnt_w0; s2[1] = cnt_w2; }
#endif
  __syncthreads();                                               // the stream has drained: every tile is dead, every flag is set

  // ---- the exact path for flagged blocks (rare): the task's K | V whole, the flagged blocks dealt to the waves, per-tile guard ----
  for (int k = 0; k < ntask; ++k) {
    const unsigned bits = __builtin_amdgcn_readfirstlane(lds_load_u32(ctrl + 256 + 4 * k));
    if (bits == 0u) continue;                                    // (workgroup-uniform)
    const int t = t0 + k;
    const int crop = t / heads, head = t - crop * heads;
    const char* tb = (const char*)(qkv + (size_t)crop * n_tok * ld + head * 64);
    for (int j = wave; j < rows / 8; j += NW) issue_pieces(tb, j, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int nth = 0;
    for (int qb = 0; qb < n_qb; ++qb) {
      if (!((bits >> qb) & 1u)) continue;
      if (nth++ % NW != wave) continue;
      bf16x8_t qf[4];
      q_load(crop, head, qb, qf);
      sweep(std::false_type{}, qf, 0u, crop, head, qb);
    }
    __syncthreads();
  }
}

hipError_t launch_attn_long_stream(const bf16_t* qkv, bf16_t* out, int n_crops, int n_tok, int width, int heads,
                                   const float* out_inv, hipStream_t stream) {
  constexpr int NCW = 11;                                        // + the loader: three waves per SIMD
  const int nkt = (n_tok + 31) / 32;
  const int lds = nkt * 32 * 128 * 2 + 4096;
  if (lds > 160 * 1024 || n_tok <= 288) return hipErrorInvalidValue;
  static DeviceKernelSetup setup;
  int n_cu = 256;
  if (hipError_t e = setup.ensure((const void*)attn_long_stream_kernel<NCW>, 160 * 1024, &n_cu); e != hipSuccess) return e;
  const int n_tasks = n_crops * heads;
  int grid = n_tasks < n_cu ? n_tasks : n_cu;
  while ((n_tasks + grid - 1) / grid > 496) grid *= 2;           // the redo bitmap holds 496 tasks per workgroup
  const float scale_log2e = 0.125f * 1.44269504088896340736f;
  hipLaunchKernelGGL((attn_long_stream_kernel<NCW>), dim3(grid), dim3((NCW + 1) * 64), lds, stream, qkv, out, n_tok, width, heads,
                     scale_log2e, nkt, n_tasks, out_inv);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Streaming variant for 225..288 tokens (ViT-L/14: 257): the throughput kernel.
// One persistent workgroup per CU = 7 compute waves + 1 loader wave over a contiguous range of
// (crop, head) tasks.  K and V of two tasks live in LDS (2 x 72 KiB).  The loader wave fills the buffer of
// task k by LDS-DMA as soon as all 32-query blocks of task k-2 are done, so the HBM stream keeps
// running while the compute waves work on the resident tasks.  The blocks of consecutive tasks form ONE
// stream handed out in order from an LDS counter (no 4+4+1 tail per task; of the 7 compute waves on 4 SIMDs
// the one that shares its SIMD only with the loader takes ~50 % more blocks).  There is no s_barrier in the
// steady state: the loader publishes `landed` (tasks whose K/V are readable) in LDS and the compute
// waves count finished blocks per task in LDS; both sides poll with s_sleep and every spin is bounded.
// Compute waves issue no LDS-DMA, so hipcc keeps counted waits for their Q loads and O stores.
// O goes through a wave-private 2 KiB LDS image and leaves as whole 128-B rows.
// ---------------------------------------------------------------------------------------------
#ifdef ATTN_STAMPS               // timing experiment (results invalid): per-wave s_memtime ticks per section, written over `out`
#define STAMP(i_) do { __builtin_amdgcn_sched_barrier(0); stamp[i_] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i_) do { } while (0)
#endif

// The last block of a task of 32 b + 1 tokens holds ONE query; as a block it costs 1 / (b + 1) of the kernel, because the exponentials
// of a block are lane-parallel over QUERIES.  Here the scores are taken the other way round, S = Q . K^T -- the Q fragment the caller
// prefetched has that query in all 32 rows (rows past the last token are clamped to it) -- so a lane holds one KEY per tile and the
// softmax is NKT exponentials per lane.  The weights go through 576 B of the task's K rows that no query uses (rows ROWS - 8 ..: padding
// keys, masked in every block) back into the B-operand layout of the P.V MFMAs, which -- like the row sum -- are those of the block
// path, key for key: the row has the bits the block path gives it.  kv_off: byte offset of the task's K | V buffer in LDS.
// Run by the LOADER wave between two tasks' DMA bursts (inside the compute waves' block loop it cost more than it saved: its registers
// made the allocator spill in the block path, and every reload of a spill is a wait on vmcnt behind the Q prefetch).
template <int NKT>
__device__ __forceinline__ void attn_tail_query(bf16x8_t q0, bf16x8_t q1, bf16x8_t q2, bf16x8_t q3, int kv_off, int n_tok,
                                                          float scale_log2e, bf16_t* out, const float* out_inv, size_t qrow, int width,
                                                          int head, int stop_after) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = NKT * 32, MAT = ROWS * 128;
  int lane;                                                     // from the hardware, HERE: nothing derived from it is kept across the block path
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  const int r = lane & 31, h = lane >> 5;
  const char* Ks = smem + kv_off;
  const char* Vs = Ks + MAT;
  char* ps = smem + kv_off + (ROWS - 8) * 128;
  const bf16x8_t qf[4] = {q0, q1, q2, q3};
  float sc[NKT];
  float mx = -INFINITY;
  // K fragments of tile c + 1 are read before the MFMAs of tile c are issued, and a tile's result is taken one tile later (two
  // accumulators in turn): the four MFMAs of a tile are a dependent chain, two tiles' chains overlap
  bf16x8_t kfa[4], kfb[4];
  f32x16_t acc[2];
#pragma unroll
  for (int st = 0; st < 4; ++st) kfa[st] = *(const bf16x8_t*)(Ks + k_swz(r, st * 2 + h));
#pragma unroll
  for (int c = 0; c < NKT; ++c) {
    if (c + 1 < NKT) {
#pragma unroll
      for (int st = 0; st < 4; ++st) kfb[st] = *(const bf16x8_t*)(Ks + k_swz((c + 1) * 32 + r, st * 2 + h));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[c & 1][e] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) acc[c & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[st], kfa[st], acc[c & 1], 0, 0, 0);
    if (c >= 1) {                                               // every row of a tile is the query: [0] = <Q, K[32 (c - 1) + r]>
      sc[c - 1] = acc[(c - 1) & 1][0];
      mx = fmaxf(mx, sc[c - 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < 4; ++st) kfa[st] = kfb[st];
  }
  sc[NKT - 1] = ((NKT - 1) * 32 + r < n_tok) ? acc[(NKT - 1) & 1][0] : -INFINITY;     // only the last tile has padded keys
  mx = fmaxf(mx, sc[NKT - 1]);
  if (stop_after == 4) { if (mx == 12345.f) out[0] = 0; return; }                       // (timing experiments: 4 = scores only, 5 = + weights)
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  const float moff = mx * scale_log2e;
  if (h == 0) {
#pragma unroll
    for (int c = 0; c < NKT; ++c) {
      const float pw = __builtin_amdgcn_exp2f(fmaf(sc[c], scale_log2e, -moff));
      *(unsigned short*)(ps + (c * 32 + r) * 2) = (unsigned short)cvt_pk_bf16(pw, 0.f);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (stop_after == 5) return;
  // P.V and the row sum over 16-key steps; slot i of lane half h holds key 16 j + (i & 3) + 8 (i >> 2) + 4 h, as in the block path.
  // The fragments of step j + 1 are read before the three MFMAs of step j are issued.
  const int vi = lane & 15, vq = vi >> 2, vp = vi & 3, vg1 = (lane >> 4) & 1;
  const char* vbase0 = Vs + v_swz(4 * h + vq, (vg1 * 16 + vp * 4) >> 3) + ((vg1 * 16 + vp * 4) & 7) * 2;
  const char* vbase1 = Vs + v_swz(4 * h + vq, (32 + vg1 * 16 + vp * 4) >> 3) + ((vg1 * 16 + vp * 4) & 7) * 2;
  auto v_frag = [&](int j, const char* vb) -> bf16x8_t {
    typedef __attribute__((ext_vector_type(8))) short s16x8_t;
    s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + j * 2048));
    s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + j * 2048 + 1024));
    s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, vv);
  };
  auto p_frag = [&](int j) -> bf16x8_t {
    const uint2 plo = *(const uint2*)(ps + (16 * j + 4 * h) * 2), phi = *(const uint2*)(ps + (16 * j + 8 + 4 * h) * 2);
    const u32x4_t pw = {plo.x, plo.y, phi.x, phi.y};
    return __builtin_bit_cast(bf16x8_t, pw);
  };
  const u32x4_t ones_w = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};        // eight bf16 1.0
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_w);
  constexpr int J = 2 * NKT - 1;                                // steps that always hold a real key; step J only if n_tok > 16 J
  f32x16_t o[2], lacc;
#pragma unroll
  for (int e = 0; e < 16; ++e) { o[0][e] = 0.f; o[1][e] = 0.f; lacc[e] = 0.f; }
  bf16x8_t pf = p_frag(0), va = v_frag(0, vbase0), vb = v_frag(0, vbase1);
#pragma unroll
  for (int j = 0; j < J; ++j) {
    bf16x8_t pn = pf, van = va, vbn = vb;
    if (j + 1 < J) { pn = p_frag(j + 1); van = v_frag(j + 1, vbase0); vbn = v_frag(j + 1, vbase1); }
    __builtin_amdgcn_sched_barrier(0);
    lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lacc, 0, 0, 0);
    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf, o[0], 0, 0, 0);
    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, pf, o[1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    pf = pn; va = van; vb = vbn;
  }
  if (n_tok > J * 16) {                                         // wave-uniform: the last 16-key step holds real keys
    const bf16x8_t pl = p_frag(J), vl0 = v_frag(J, vbase0), vl1 = v_frag(J, vbase1);
    lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pl, lacc, 0, 0, 0);
    o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl0, pl, o[0], 0, 0, 0);
    o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl1, pl, o[1], 0, 0, 0);
  }
  const float inv = 1.0f / lacc[0];
  if (r == 0) {                                                 // lanes 0 and 32 hold the row: dims 32 dt + 8 g4 + 4 h + {0..3}
    if (out_inv) {
      uint8_t* ob = (uint8_t*)out + qrow * width + head * 64 + h * 4;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4_t is = *(const f32x4_t*)(out_inv + head * 64 + dt * 32 + g4 * 8 + h * 4);
          *(int*)(ob + dt * 32 + g4 * 8) = pack_fp8x4(o[dt][g4 * 4 + 0] * inv * is[0], o[dt][g4 * 4 + 1] * inv * is[1],
                                                      o[dt][g4 * 4 + 2] * inv * is[2], o[dt][g4 * 4 + 3] * inv * is[3]);
        }
    } else {
      bf16_t* ob = out + qrow * width + head * 64 + h * 4;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          *(uint2*)(ob + dt * 32 + g4 * 8) = uint2{cvt_pk_bf16(o[dt][g4 * 4 + 0] * inv, o[dt][g4 * 4 + 1] * inv),
                                                   cvt_pk_bf16(o[dt][g4 * 4 + 2] * inv, o[dt][g4 * 4 + 3] * inv)};
    }
  }
}

template <int NKT, int NCW, bool TAIL>
__global__ __launch_bounds__((NCW + 1) * 64, (NCW + 1) / 4) void attn_stream_kernel(
    const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int n_tok, int width, int heads, float scale_log2e,
    int n_tasks, int dbg_mode, const float* __restrict__ out_inv, int q_blocks) {
  // NCW compute waves + 1 loader; all NKT key tiles of a block's scores live in registers (one exact pass)
  // dbg_mode (timing experiments only, results invalid): 1 = loader alone, 2 = compute alone
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWS = NKT * 32, MAT = ROWS * 128, BUF = 2 * MAT;
  constexpr int TRB = NCW <= 7 ? 2048 : 1024;                   // wave-private O image: 16 or 8 rows x 128 B
  constexpr int TR_ROWS = TRB / 128;
  constexpr int CTRL = 2 * BUF + NCW * TRB;                     // [0]: landed, [4]: next block, [16..]: done[k] per task (<= 496 tasks)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x, wg = blockIdx.x;
  const int t0 = (int)(((long long)n_tasks * wg) / G), t1 = (int)(((long long)n_tasks * (wg + 1)) / G);
  const int ntask = t1 - t0;                                    // host guarantees 1 <= ntask <= 496
  // q_blocks: only the first 32-query blocks of every task.  TAIL (n_tok = 32 b + 1, all blocks asked for): the compute waves take the
  // b full blocks, the loader wave computes the one query left over (attn_tail_query) while the task's blocks are being computed
  const int n_qb = TAIL ? NKT - 1 : min((n_tok + 31) >> 5, q_blocks);
  const int total_blocks = ntask * n_qb;
  const size_t ld = (size_t)3 * width;
  const unsigned ldb = (unsigned)(ld * 2);
  char* ctrl = smem + CTRL;

  for (int i = tid; i < 512; i += (NCW + 1) * 64) ((unsigned*)ctrl)[i] = 0u;
  __syncthreads();

  if (wave == NCW) {
    // ------------------------------------ loader wave ------------------------------------
    if (dbg_mode == 2) {
      if (lane == 0) __atomic_store_n((unsigned*)ctrl, (unsigned)ntask, __ATOMIC_RELAXED);
      return;
    }
    if constexpr (TAIL) {
      const int h = lane >> 5;
      // This wave goes ahead of the compute wave it shares a SIMD with (which sleeps between its polls costs that wave nothing): the
      // sooner it is back at the counter below, the sooner the next task's K / V are requested.  At equal priority the one-query path
      // took four times as long and this wave, not the compute waves, paced the kernel.
      __builtin_amdgcn_s_setprio(3);
      for (int k = 0; k < ntask; ++k) {
        if (k >= 2 && dbg_mode != 1) {                           // buffer k&1 is free once every block of task k-2 has been computed
          unsigned spins = 0;
          while (lds_load_u32(ctrl + 16 + (k - 2) * 4) != (unsigned)n_qb && ++spins < SPIN_LIMIT) __builtin_amdgcn_s_sleep(2);
        }
        asm volatile("" ::: "memory");
        const int t = t0 + k;
        const int crop = t / heads, head = t - crop * heads;
        const char* tb = (const char*)(qkv + (size_t)crop * n_tok * ld + head * 64);
        // the task's last query, the same row in every lane row (what the A operand of S = Q . K^T wants); it lands with the K / V rows
        const size_t qrow = (size_t)crop * n_tok + (n_tok - 1);
        bf16x8_t qf[4];
        {
          const bf16_t* qp = qkv + qrow * ld + head * 64 + h * 8;
#pragma unroll
          for (int st = 0; st < 4; ++st) qf[st] = *(const bf16x8_t*)(qp + st * 16);
        }
        const int dst = (k & 1) * BUF;
#pragma unroll 4
        for (int j = 0; j < ROWS / 8; ++j) {
          const int row = 8 * j + (lane >> 3);
          const unsigned rb = (unsigned)min(row, n_tok - 1) * ldb;
          const int ck = (lane & 7) ^ ((row >> 1) & 7);
          const int cv = (lane & 7) ^ (((row >> 1) & 1) << 2);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + (size_t)width * 2 + (rb + ck * 16)),
                                           (__attribute__((address_space(3))) void*)(smem + dst + j * 1024), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + (size_t)width * 4 + (rb + cv * 16)),
                                           (__attribute__((address_space(3))) void*)(smem + dst + MAT + j * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // task k has landed (and the row stored in the previous round has left)
        if (lane == 0) __atomic_store_n((unsigned*)ctrl, (unsigned)(k + 1), __ATOMIC_RELAXED);
        if (dbg_mode != 3)                                       // (3: timing experiment without this path, results invalid)
          attn_tail_query<NKT>(qf[0], qf[1], qf[2], qf[3], dst, n_tok, scale_log2e, out, out_inv, qrow, width, head, dbg_mode);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // its K / V reads have returned before the buffer is refilled (two rounds on)
      }
      return;
    }
    for (int k = 0; k < ntask; ++k) {
      if (k >= 2 && dbg_mode != 1) {
        // buffer k&1 is free once every block of task k-2 has been computed
        if (lds_load_u32(ctrl + 16 + (k - 2) * 4) != (unsigned)n_qb) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // task k-1 has landed: publish it before waiting
          if (lane == 0) __atomic_store_n((unsigned*)ctrl, (unsigned)k, __ATOMIC_RELAXED);
          unsigned spins = 0;
          while (lds_load_u32(ctrl + 16 + (k - 2) * 4) != (unsigned)n_qb && ++spins < SPIN_LIMIT) __builtin_amdgcn_s_sleep(2);
        }
      }
      asm volatile("" ::: "memory");
      const int t = t0 + k;
      const int crop = t / heads, head = t - crop * heads;
      const char* tb = (const char*)(qkv + (size_t)crop * n_tok * ld + head * 64);
      const int dst = (k & 1) * BUF;
#pragma unroll 4
      for (int j = 0; j < ROWS / 8; ++j) {
        const int row = 8 * j + (lane >> 3);
        const unsigned rb = (unsigned)min(row, n_tok - 1) * ldb;
        const int ck = (lane & 7) ^ ((row >> 1) & 7);
        const int cv = (lane & 7) ^ (((row >> 1) & 1) << 2);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + (size_t)width * 2 + (rb + ck * 16)),
                                         (__attribute__((address_space(3))) void*)(smem + dst + j * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tb + (size_t)width * 4 + (rb + cv * 16)),
                                         (__attribute__((address_space(3))) void*)(smem + dst + MAT + j * 1024), 16, 0, 0);
      }
      // task k has >= 63 DMA instructions, so "at most 63 outstanding" means every older task has landed
      asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
      if (lane == 0) __atomic_store_n((unsigned*)ctrl, (unsigned)k, __ATOMIC_RELAXED);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __atomic_store_n((unsigned*)ctrl, (unsigned)ntask, __ATOMIC_RELAXED);
    return;
  }

  // ------------------------------------ compute waves ------------------------------------
  if (dbg_mode == 1) return;
  const int r = lane & 31, h = lane >> 5;
  char* tr = smem + 2 * BUF + wave * TRB;

  // Q fragments of a block: lane (r,h) holds Q[q0+r][16*step + 8h .. +7]; loaded one block ahead
  auto q_ptr = [&](int gg) {
    const int kk = gg / n_qb, qq = gg - kk * n_qb;
    const int tt = t0 + kk;
    const int cc = tt / heads, hh = tt - cc * heads;
    return qkv + (size_t)cc * n_tok * ld + hh * 64 + (size_t)min(qq * 32 + r, n_tok - 1) * ld + h * 8;
  };
  // blocks are handed out in order from an LDS counter: a wave that has its SIMD to itself takes more of them
  auto grab = [&]() -> int {
    unsigned v = 0;
    if (lane == 0) v = __atomic_fetch_add((unsigned*)(ctrl + 4), 1u, __ATOMIC_RELAXED);
    return (int)__builtin_amdgcn_readfirstlane(v);
  };
  const int g_first = grab();
  bf16x8_t qf[4];
  {
    const bf16_t* qp = q_ptr(min(g_first, total_blocks - 1));
#pragma unroll
    for (int st = 0; st < 4; ++st) qf[st] = *(const bf16x8_t*)(qp + st * 16);
  }

#ifdef ATTN_STAMPS
  unsigned long long stamp[6] = {0, 0, 0, 0, 0, 0}, sect[5] = {0, 0, 0, 0, 0}, nblk = 0;
  const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
  for (int g = g_first, g_next = 0; g < total_blocks; g = g_next) {
    STAMP(0);
    g_next = grab();                                            // the block after this one: its Q is prefetched below
    const int k = g / n_qb, qb = g - k * n_qb;
    const int t = t0 + k;
    const int crop = t / heads, head = t - crop * heads;
    const char* Ks = smem + (k & 1) * BUF;
    const char* Vs = Ks + MAT;

    {                                                           // wait until K/V of task k are readable
      unsigned spins = 0;
      while (lds_load_u32(ctrl) <= (unsigned)k && ++spins < SPIN_LIMIT) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
    }
    STAMP(1);

    const u32x4_t ones_w = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // eight bf16 1.0
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_w);


    f32x16_t o[2], lacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.f; o[1][e] = 0.f; lacc[e] = 0.f; }

    // ---- S^T tiles: s[c][reg] = <K[32 c + krow(reg,h)], Q[q]>.  K fragments of tile c + 1 are read before the MFMAs of tile c are
    // issued (register double buffer; hipcc otherwise reads each fragment right in front of its MFMA and waits), and the row max of
    // the finished tile c - 1 runs in the shadow of tile c's MFMAs, as four independent v_max3 chains (one chain of 72 dependent
    // v_max3 after the last tile took twice as long by the section stamps) ----
    f32x16_t s[NKT];
    bf16x8_t kfa[4], kfb[4];
    float mx4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int st = 0; st < 4; ++st) kfa[st] = *(const bf16x8_t*)(Ks + k_swz(r, st * 2 + h));
#pragma unroll
    for (int c = 0; c < NKT; ++c) {
      if (c + 1 < NKT) {
#pragma unroll
        for (int st = 0; st < 4; ++st) kfb[st] = *(const bf16x8_t*)(Ks + k_swz((c + 1) * 32 + r, st * 2 + h));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 16; ++e) s[c][e] = 0.f;
#pragma unroll
      for (int st = 0; st < 4; ++st) s[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfa[st], qf[st], s[c], 0, 0, 0);
      if (c >= 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) mx4[e & 3] = fmaxf(mx4[e & 3], s[c - 1][e]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int st = 0; st < 4; ++st) kfa[st] = kfb[st];
    }
    STAMP(2);
    // ---- the last tile: mask its padded keys, finish the max (exact two-pass softmax: the true row max) ----
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = (NKT - 1) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (key >= n_tok) s[NKT - 1][e] = -INFINITY;
      mx4[e & 3] = fmaxf(mx4[e & 3], s[NKT - 1][e]);
    }
    float mx = fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3]));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float moff = mx * scale_log2e;
    STAMP(3);

    // ---- P = exp2(s*c - m*c), row sum, O^T += V^T . P^T as a software pipeline over 16-key steps with the program order pinned:
    // the three MFMAs of step j (ones . P^T for the row sums of the ROUNDED weights, V^T . P^T for the two 32-wide halves of d) with
    // the 20 VALU instructions of step j + 1 between them -- an MFMA holds the matrix pipe for 32 cycles, v_fma_f32 takes 5.2 and
    // v_exp_f32 ~11 cycles of VALU issue in its shadow (tools/probes/issue_overlap_probe.hip) -- and each V fragment is read again
    // for step j + 1 right behind the MFMA that consumed it, a whole step ahead of its use.  Left to itself hipcc computes a step's
    // weights, reads its V fragments and waits for them in front of the MFMAs (section stamps: a quarter less time in this phase).
    // No v_pk_fma_f32 for the scaling: packed fp32 runs on the matrix pipe (8.5 cycles each, serialised with the MFMAs). ----
    {
      constexpr int J = 2 * NKT - 1;                            // steps that always hold a real key; step J only if n_tok > 16 J
      constexpr int QPF = J > 5 ? 5 : J - 1;                    // step at which the next block's Q loads are issued (s[0], s[1] are dead; short sweeps: the last step)
      const int vi = lane & 15, vq = vi >> 2, vp = vi & 3, vg1 = (lane >> 4) & 1;
      // V^T fragment of step j, half dt: rows 16 j + 4 h + q and + 8, 16-B chunk (32 dt + 16 g1 + 4 p) / 8; the swizzle term of
      // those rows is ((q >> 1) & 1) for every j, so a step is a constant offset from a per-lane base
      const char* vbase0 = Vs + v_swz(4 * h + vq, (vg1 * 16 + vp * 4) >> 3) + ((vg1 * 16 + vp * 4) & 7) * 2;
      const char* vbase1 = Vs + v_swz(4 * h + vq, (32 + vg1 * 16 + vp * 4) >> 3) + ((vg1 * 16 + vp * 4) & 7) * 2;
      auto p_scale = [&](int j, float (&t)[8]) {
        const int c = j >> 1, s2 = j & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = fmaf(s[c][s2 * 8 + i], scale_log2e, -moff);
      };
      auto v_frag = [&](int j, const char* vb) -> bf16x8_t {
        typedef __attribute__((ext_vector_type(8))) short s16x8_t;
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + j * 2048));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vb + j * 2048 + 1024));
        s16x8_t vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, vv);
      };
      auto p_frag = [&](int j) -> bf16x8_t {
        float t[8];
        p_scale(j, t);
        u32x4_t pw;
#pragma unroll
        for (int i = 0; i < 4; ++i) pw[i] = cvt_pk_bf16(__builtin_amdgcn_exp2f(t[2 * i]), __builtin_amdgcn_exp2f(t[2 * i + 1]));
        return __builtin_bit_cast(bf16x8_t, pw);
      };
      __builtin_amdgcn_sched_barrier(0);
      bf16x8_t va = v_frag(0, vbase0), vb = v_frag(0, vbase1);
      bf16x8_t pf = p_frag(0);
#pragma unroll
      for (int j = 0; j < J; ++j) {
        float t[8];
        u32x4_t pw = __builtin_bit_cast(u32x4_t, pf);
        const bool nx = j + 1 < J;
        __builtin_amdgcn_sched_barrier(0);
        if (j == QPF) {                                         // next block's Q: in flight for the rest of this block
          const bf16_t* qp = q_ptr(min(g_next, total_blocks - 1));
#pragma unroll
          for (int st = 0; st < 4; ++st) qf[st] = *(const bf16x8_t*)(qp + st * 16);
          __builtin_amdgcn_sched_barrier(0);
        }
        lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf, lacc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (nx) {
          p_scale(j + 1, t);
          // the empty asm makes the values exist HERE: instruction selection otherwise sinks them to their use behind the barriers
          asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
        }
        __builtin_amdgcn_sched_barrier(0);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf, o[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (nx) {
          va = v_frag(j + 1, vbase0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) t[i] = __builtin_amdgcn_exp2f(t[i]);
          asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
        }
        __builtin_amdgcn_sched_barrier(0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, pf, o[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (nx) {
          vb = v_frag(j + 1, vbase1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 4; i < 8; ++i) t[i] = __builtin_amdgcn_exp2f(t[i]);
#pragma unroll
          for (int i = 0; i < 4; ++i) pw[i] = cvt_pk_bf16(t[2 * i], t[2 * i + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        pf = __builtin_bit_cast(bf16x8_t, pw);
      }
      if (n_tok > J * 16) {                                     // wave-uniform: the last 16-key step holds real keys
        const bf16x8_t vl0 = v_frag(J, vbase0), vl1 = v_frag(J, vbase1);
        const bf16x8_t pl = p_frag(J);
        lacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pl, lacc, 0, 0, 0);
        o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl0, pl, o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl1, pl, o[1], 0, 0, 0);
      }
    }
    // every K/V read of this block has returned (the MFMAs consumed them): release the buffer share
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __atomic_fetch_add((unsigned*)(ctrl + 16 + k * 4), 1u, __ATOMIC_RELAXED);
    STAMP(4);

    const float inv = 1.0f / lacc[0];                           // the MFMA already summed both lane halves
    const int ol = lane, orr = r, oh = h;
    const int otr_base = (ol >> 3) * 128 + (((ol & 7) ^ (ol >> 3)) << 4);

    if (out_inv) {
      // ---- O as e4m3: fragment layout -> [TR_ROWS q rows][80-B pitch] image -> whole 64-B rows ----
      uint8_t* obase8 = (uint8_t*)out + (size_t)crop * n_tok * width + head * 64 + (ol & 3) * 16;
      int opk8[8];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x4_t is = *(const f32x4_t*)(out_inv + head * 64 + dt * 32 + g4 * 8 + oh * 4);
          opk8[dt * 4 + g4] = pack_fp8x4(o[dt][g4 * 4 + 0] * inv * is[0], o[dt][g4 * 4 + 1] * inv * is[1],
                                         o[dt][g4 * 4 + 2] * inv * is[2], o[dt][g4 * 4 + 3] * inv * is[3]);
        }
#pragma unroll
      for (int pass = 0; pass < 32 / TR_ROWS; ++pass) {
        if (orr / TR_ROWS == pass) {
          const int rr = orr % TR_ROWS;
#pragma unroll
          for (int c8 = 0; c8 < 8; ++c8) *(int*)(tr + rr * 80 + c8 * 8 + oh * 4) = opk8[c8];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int row = ol >> 2;
        if (row < TR_ROWS) {
          const uint4 v0 = *(const uint4*)(tr + row * 80 + (ol & 3) * 16);
          const int qa = qb * 32 + pass * TR_ROWS + row;
          if (qa < n_tok) *(uint4*)(obase8 + (size_t)qa * width) = v0;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      continue;
    }
    // ---- O: fragment layout -> [TR_ROWS q rows][64 d] bf16 image -> whole 128-B rows ----
    bf16_t* obase = out + (size_t)crop * n_tok * width + head * 64 + (ol & 7) * 8;
    uint2 opk[8];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        opk[dt * 4 + g4] = uint2{cvt_pk_bf16(o[dt][g4 * 4 + 0] * inv, o[dt][g4 * 4 + 1] * inv),
                                 cvt_pk_bf16(o[dt][g4 * 4 + 2] * inv, o[dt][g4 * 4 + 3] * inv)};
#pragma unroll
    for (int pass = 0; pass < 32 / TR_ROWS; ++pass) {
      if (orr / TR_ROWS == pass) {
        const int rr = orr % TR_ROWS;
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) *(uint2*)(tr + rr * 128 + ((c8 ^ (rr & 7)) << 4) + oh * 8) = opk[c8];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint4 v0 = *(const uint4*)(tr + otr_base);
      const int qa = qb * 32 + pass * TR_ROWS + (ol >> 3);
      if (qa < n_tok) *(uint4*)(obase + (size_t)qa * width) = v0;
      if (TR_ROWS == 16) {
        const uint4 v1 = *(const uint4*)(tr + 1024 + otr_base);
        if (qa + 8 < n_tok) *(uint4*)(obase + (size_t)(qa + 8) * width) = v1;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#ifdef ATTN_STAMPS
    STAMP(5);
    for (int i = 0; i < 5; ++i) sect[i] += stamp[i + 1] - stamp[i];
    ++nblk;
#endif
  }
#ifdef ATTN_STAMPS
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    unsigned long long* so = (unsigned long long*)out + ((size_t)wg * NCW + wave) * 8;
    if (lane == 0) { for (int i = 0; i < 5; ++i) so[i] = sect[i]; so[5] = nblk; so[6] = t_end - t_begin; so[7] = 0x5741505354414d50ull; }
  }
#endif
}

template <int NKT, int NCW, bool TAIL>
hipError_t launch_attn_stream(const bf16_t* qkv, bf16_t* out, int n_crops, int n_tok, int width, int heads,
                              const float* out_inv, int q_blocks, hipStream_t stream) {
  const int lds = 2 * 2 * NKT * 32 * 128 + NCW * (NCW <= 7 ? 2048 : 1024) + 2048;
  static DeviceKernelSetup setup;             // per device: LDS opt-in + CU count (common.h)
  int n_cu = 256;
  if (hipError_t e = setup.ensure((const void*)attn_stream_kernel<NKT, NCW, TAIL>, lds, &n_cu); e != hipSuccess) return e;
  const int n_tasks = n_crops * heads;
  int grid = n_tasks < n_cu ? n_tasks : n_cu;
  while ((n_tasks + grid - 1) / grid > 496) grid *= 2;      // per-workgroup task counters live in 2 KiB of LDS
  const float scale_log2e = 0.125f * 1.44269504088896340736f;
#ifdef CLIPENC_DIAG                         // timing experiments of tools/attn_ab.py (results invalid): 1 = loader alone, 2 = compute alone
  static const int dbg = [] { const char* e = getenv("CLIPENC_ATTN_DBG"); return e ? atoi(e) : 0; }();
#else
  constexpr int dbg = 0;
#endif
  hipLaunchKernelGGL((attn_stream_kernel<NKT, NCW, TAIL>), dim3(grid), dim3((NCW + 1) * 64), lds, stream, qkv, out, n_tok, width, heads,
                     scale_log2e, n_tasks, dbg, out_inv, q_blocks);
  return hipGetLastError();
}

template <int NKT>
hipError_t launch_attn(const bf16_t* qkv, bf16_t* out, int n_crops, int n_tok, int width, int heads,
                       const float* out_inv, int q_blocks, hipStream_t stream) {
  const int lds = NKT * 32 * 128 * 2;
  static DeviceKernelSetup setup;
  if (hipError_t e = setup.ensure((const void*)attn_kernel<NKT>, lds, nullptr); e != hipSuccess) return e;
  const float scale_log2e = 0.125f * 1.44269504088896340736f;   // 64^-0.5 * log2(e)
  hipLaunchKernelGGL((attn_kernel<NKT>), dim3(n_crops * heads), dim3(256), lds, stream, qkv, out, n_tok, width,
                     heads, scale_log2e, out_inv, q_blocks);
  return hipGetLastError();
}

}  // namespace

// qkv: [n_crops*n_tok][3*width] bf16 ([q|k|v], head = 64-wide slice); out: [n_crops*n_tok][width] bf16, or, when
// out_inv != NULL, e4m3 bytes: out8[t][c] = fp8(O[t][c] * out_inv[c])
hipError_t ce_attention(const void* qkv, void* out, int n_crops, int n_tok, int width, int heads,
                        const float* out_inv, int q_blocks, hipStream_t stream) {
  if (q_blocks < 1) q_blocks = 1 << 20;                         // all query blocks
  if (heads < 1 || width % heads != 0 || n_tok < 1 || n_crops < 1) return hipErrorInvalidValue;
  const int nkt = (n_tok + 31) / 32;
  const bf16_t* q = (const bf16_t*)qkv;
  bf16_t* o = (bf16_t*)out;
#define ATTN_HD_CASES(HD)                                                                                                  \
  switch (nkt) {                                                                                                           \
    case 1: return launch_attn_hd<1, HD>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);                   \
    case 2: return launch_attn_hd<2, HD>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);                   \
    case 3: return launch_attn_hd<3, HD>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);                   \
    case 4: return launch_attn_hd<4, HD>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);                   \
    case 5: return launch_attn_hd<5, HD>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);                   \
    case 6: return launch_attn_hd<6, HD>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);                   \
    case 7: return launch_attn_hd<7, HD>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);                   \
    case 8: return launch_attn_hd<8, HD>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);                   \
    case 9: return launch_attn_hd<9, HD>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);                   \
    default: return hipErrorInvalidValue;                                                                                  \
  }
  if (width == heads * 80) { ATTN_HD_CASES(80) }                  // ViT-H-14: its own kernel family, up to 288 tokens
  if (width == heads * 96) { ATTN_HD_CASES(96) }                  // ViT-g-14's 88-wide heads, zero-padded by clipenc_create
  if (width == heads * 112) { ATTN_HD_CASES(112) }                // ViT-bigG-14's 104-wide heads
  if (width == heads * 128) { ATTN_HD_CASES(128) }
#undef ATTN_HD_CASES
  if (width != heads * 64) return hipErrorInvalidValue;
#ifdef CLIPENC_DIAG                         // developer A/B: 0 = one workgroup per (crop, head) for every shape; CLIPENC_ATTN_TAIL=0: the odd query as a block
  static const int impl = [] { const char* e = getenv("CLIPENC_ATTN_IMPL"); return e ? atoi(e) : 2; }();
  static const bool tail_on_loader = [] { const char* e = getenv("CLIPENC_ATTN_TAIL"); return e ? atoi(e) != 0 : true; }();
  static const int qb_cap = [] { const char* e = getenv("CLIPENC_ATTN_QB"); return e ? atoi(e) : 0; }();   // timing experiment (results invalid)
  if (qb_cap > 0 && q_blocks > qb_cap) q_blocks = qb_cap;
  static const int long_impl = [] { const char* e = getenv("CLIPENC_ATTN_LONG_IMPL"); return e ? atoi(e) : 2; }();   // 1 = attn_long_kernel for every long launch
#else
  constexpr int impl = 2;
  constexpr bool tail_on_loader = true;
  constexpr int long_impl = 2;
#endif
  if (impl == 2 && nkt == 8 && n_crops * heads >= 64) return launch_attn_stream<8, 7, false>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
  // 65 .. 224 tokens (ViT-B-16 / L-16 at 224 px: 197), round 6: the same kernel with three to seven key tiles -- bit-identical to attn_kernel<NKT> and
  // 15-30 % faster from 96 tokens up (2 048 crops x 16 heads: 197 tokens 1.129 -> 0.793 ms, 170: 0.951 -> 0.666, 145: 0.699 -> 0.559, 101: 0.471 -> 0.392,
  // 96: 0.433 -> 0.365; 65: equal); two tiles (33 .. 64 tokens) are not worth the persistent form (33 tokens: 0.172 -> 0.200)
  if (impl == 2 && nkt >= 3 && nkt <= 7 && n_crops * heads >= 64) {
    switch (nkt) {
      case 3: return launch_attn_stream<3, 7, false>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
      case 4: return launch_attn_stream<4, 7, false>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
      case 5: return launch_attn_stream<5, 7, false>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
      case 6: return launch_attn_stream<6, 7, false>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
      default: return launch_attn_stream<7, 7, false>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
    }
  }
  if (impl == 2 && nkt == 9 && n_crops * heads >= 64) {
    // 32 b + 1 tokens (ViT-L/14: 257) with every block asked for: the odd query goes to the loader wave
    if (tail_on_loader && (n_tok & 31) == 1 && q_blocks >= nkt) return launch_attn_stream<9, 7, true>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
    return launch_attn_stream<9, 7, false>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
  }
  switch (nkt) {                 // NKT must equal ceil(n_tok/32): only the last key tile is masked
    case 1: return launch_attn<1>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
    case 2: return launch_attn<2>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
    case 3: return launch_attn<3>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
    case 4: return launch_attn<4>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
    case 5: return launch_attn<5>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
    case 6: return launch_attn<6>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
    case 7: return launch_attn<7>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
    case 8: return launch_attn<8>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
    case 9: return launch_attn<9>(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);
    default:
      // every block of >= 64 tasks, up to 608 tokens: the streaming form (K | V + 4 KiB of control words in 160 KiB)
      if (long_impl == 2 && nkt <= 19 && q_blocks >= nkt && n_crops * heads >= 64)
        return launch_attn_long_stream(q, o, n_crops, n_tok, width, heads, out_inv, stream);
      return launch_attn_long(q, o, n_crops, n_tok, width, heads, out_inv, q_blocks, stream);   // up to 640 tokens (K, V of one head in LDS)
  }
}
