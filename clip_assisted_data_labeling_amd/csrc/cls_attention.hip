// The LAST transformer block's attention for the class-token query only, without K and V (SURVEY.md Appendix A.2 steps 4-5:
// the embedding is ln_post + proj of token 0, /root/reference/utils/embedder.py:98 takes the pooled output).
//
// Rounds 1-3 ran the last block's K | V projection for EVERY token (2 T D^2 MACs: 1.84 ms of the 512-image step) only to feed one
// query per (crop, head).  With the LayerNorm folded as in the GEMMs (k_j[n] = rstd_j (x_j . W'_k[n] - mu_j colsum_k[n]) + bias'_k[n],
// W' = gamma (.) W, bias' = b + W beta) the class-token attention of head h needs no K or V row at all:
//   score_j = q_h . k_j[h] = rstd_j (x_j . r_h - mu_j s_h) + t_h        r_h = sum_{d in h} q_d W'_k[d, :]   (a D-vector per head)
//                                                                     s_h = sum q_d colsum_k[d],  t_h = sum q_d bias'_k[d]
//   o_h[d]  = sum_j p_j v_j[d] = z_h . W'_v[d, :] - m_h colsum_v[d] + bias'_v[d]
//                                                                     z_h = sum_j p_j rstd_j x_j,  m_h = sum_j p_j rstd_j mu_j
// i.e. an attention problem with ONE query per head, "keys" and "values" = the crop's raw residual rows x_j (D wide), which are read
// twice -- 2 T D bytes instead of a GEMM over them.  r_h and o_h are batched small GEMMs that the existing persistent GEMM runs
// (capi.hip: 16x redundant and still 65 us each); THIS file is the part in between:
//   cls_attn_kernel  one workgroup per crop: scores of all heads against all tokens, softmax, z_h and m_h     (HBM / Infinity-Cache bound)
//   cls_qmask_kernel Q_cls -> one row per (crop, head) with the other heads' columns zeroed (the operand of the r_h GEMM)
//   cls_finish_kernel the diagonal blocks of z . W'_v^T, the rank-1 mean term and the bias -> the class-token rows of `attn` (bf16 or e4m3)
// Arithmetic: bf16 operands, fp32 accumulation, every reduction in a fixed order (deterministic; a crop's result does not depend on
// its position in the batch).  Parity: tests/test_gpu_cls_only.py holds it to the standard last block and the towers to the oracle.
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

__device__ __forceinline__ int k_swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_swz(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4); }

// One LDS-DMA piece: 64 lanes x 16 B from (uniform base + per-lane 32-bit offset) to 1 KiB of LDS at lds_addr.  Inline asm, as in
// gemm_persist.hip: behind the builtin hipcc waits for EVERY outstanding piece in front of the next LDS read, i.e. the next tile's
// fetch would not overlap this tile's MFMAs; the waits below are counted by hand (a wave's own pieces retire in order).
__device__ __forceinline__ void glds16_at(const char* base, unsigned off, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_addr) : "memory");
}
// wait until at most `n` (0..4, wave-uniform) of this wave's vector-memory operations are outstanding
__device__ __forceinline__ void wait_vm(int n) {
  if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

constexpr int R_OFF = 0;                 // [D/64 segments][16 heads x 128 B], k_swz                                  <= 32 KiB
constexpr int X_OFF = 32768;             // 2 buffers x [D/64 segments][16 tokens x 128 B]                                64 KiB
constexpr int XBUF = 32768;
constexpr int PART_OFF = X_OFF + 2 * XBUF;        // [8 waves][16 heads][16 tokens] fp32 partial scores                   8 KiB
constexpr int ST_OFF = PART_OFF + 8192;           // [ntp] (mean, rstd)                                               <= 5 KiB
constexpr int HS_OFF = ST_OFF + 640 * 8;          // [16] (s_h, t_h)
constexpr int SC_OFF = HS_OFF + 128;              // [16 heads][ntp] fp32 scores, rewritten in place as bf16 weights (row pitch ntp * 4 + 16)

// x: [T][D] bf16 residual stream; stats: [parts][stats_ld][2] (sum, sumsq) of the rows; q: the class-token Q rows (already LayerNorm-
// folded GEMM output, bf16), row c at q + c * q_stride elements; R: [(c * H + h)][D] bf16 = r_h; outputs Zp [(c * H + h)][D] bf16 and
// mz [(c * H + h)]
__global__ __launch_bounds__(512, 2) void cls_attn_kernel(const bf16_t* __restrict__ x, const float* __restrict__ stats, int parts, int stats_ld,
                                                          const bf16_t* __restrict__ q, size_t q_stride, const float* __restrict__ colsum_k,
                                                          const float* __restrict__ bias_k, const bf16_t* __restrict__ R,
                                                          bf16_t* __restrict__ Zp, float* __restrict__ mz, int n_tok, int D, int H,
                                                          float inv_width, float eps, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int crop = blockIdx.x;
  const int segs = D >> 6;                                   // 64-column segments of a row (<= 16)
  const int ntiles = (n_tok + 15) >> 4, ntp = ntiles * 16;
  const int pitch = ntp * 4 + 16;                            // score row pitch in bytes (+16: the 16 heads' rows start in different banks)
  const char* xb = (const char*)(x + (size_t)crop * n_tok * D);
  const unsigned row_b = (unsigned)D * 2u;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const int np = (segs * 2 - w + 7) / 8;                     // DMA pieces this wave issues per staged tile (<= 4)

  // ---- r_h rows of the crop into LDS (segment-major, k_swz), by LDS-DMA: one piece = 8 head rows x 128 B
  {
    const char* rb = (const char*)(R + (size_t)crop * H * D);
    for (int pc = w; pc < segs * 2; pc += 8) {
      const int seg = pc >> 1, r8 = (pc & 1) * 8;
      const int row = r8 + (lane >> 3);
      const int ck = (lane & 7) ^ ((row >> 1) & 7);
      glds16_at(rb, (unsigned)min(row, H - 1) * row_b + (unsigned)(seg * 128 + ck * 16), lds0 + (unsigned)(R_OFF + seg * 2048 + r8 * 128));
    }
  }
  // ---- (mean, rstd) of the crop's rows; s_h, t_h
  for (int j = tid; j < ntp; j += 512) {
    float s_ = 0.f, ss_ = 0.f;
    const size_t row = (size_t)crop * n_tok + min(j, n_tok - 1);
    for (int p = 0; p < parts; ++p) {
      const float2 t = *(const float2*)(stats + ((size_t)p * stats_ld + row) * 2);
      s_ += t.x; ss_ += t.y;
    }
    const float mean = s_ * inv_width;
    const float var = fmaxf(ss_ * inv_width - mean * mean, 0.f);
    *(float2*)(smem + ST_OFF + j * 8) = float2{mean, rsqrtf(var + eps)};
  }
  for (int hh = w; hh < H; hh += 8) {                          // one head per wave and pass, lane = column of the head
    const float qv = bf16_to_f32(q[(size_t)crop * q_stride + hh * 64 + lane]);
    float a = qv * colsum_k[hh * 64 + lane], b = qv * bias_k[hh * 64 + lane];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (lane == 0) *(float2*)(smem + HS_OFF + hh * 8) = float2{a, b};
  }

  // one 16-token tile of x into buffer b: segs x 2 pieces (8 token rows x 128 B), swizzle for row reads (pass 1) or transposing reads (pass 2)
  auto stage = [&](int tile, int b, bool for_tr) {
    for (int pc = w; pc < segs * 2; pc += 8) {
      const int seg = pc >> 1, r8 = (pc & 1) * 8;
      const int row = r8 + (lane >> 3);
      const int tok = min(tile * 16 + row, n_tok - 1);          // rows beyond the crop: clamped copies (their weights are exactly 0)
      const int c = for_tr ? ((lane & 7) ^ (((row >> 1) & 1) << 2)) : ((lane & 7) ^ ((row >> 1) & 7));
      glds16_at(xb, (unsigned)tok * row_b + (unsigned)(seg * 128 + c * 16), lds0 + (unsigned)(X_OFF + b * XBUF + seg * 2048 + r8 * 128));
    }
  };

  // ================= pass 1: scores[h][j] = scale (rstd_j (x_j . r_h - mu_j s_h) + t_h) =================
  stage(0, 0, false);
  const int m16 = lane & 15, kg = lane >> 4;
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) stage(t + 1, (t + 1) & 1, false);
    // this wave's pieces of tile t (and of r_h) have landed: they are older than the np pieces just issued
    wait_vm(t + 1 < ntiles ? np : 0);
    __syncthreads();
    const char* xt = smem + X_OFF + (t & 1) * XBUF;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int ks = w; ks < D / 32; ks += 8) {                   // k steps of 32 columns, split over the waves
      const int seg = ks >> 1, ch = (ks & 1) * 4 + kg;
      const bf16x8_t a = *(const bf16x8_t*)(smem + R_OFF + seg * 2048 + k_swz(m16, ch));      // r_h[32 ks + 8 kg ..]
      const bf16x8_t b = *(const bf16x8_t*)(xt + seg * 2048 + k_swz(m16, ch));                // x_{16 t + n}[32 ks + 8 kg ..]
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);                      // D[head][token]
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *(float*)(smem + PART_OFF + ((w * 16 + kg * 4 + i) * 16 + m16) * 4) = acc[i];
    __syncthreads();
    if (tid < 256) {
      const int hh = tid >> 4, n = tid & 15, j = t * 16 + n;
      float raw = 0.f;
#pragma unroll
      for (int ww = 0; ww < 8; ++ww) raw += *(const float*)(smem + PART_OFF + ((ww * 16 + hh) * 16 + n) * 4);
      const float2 st = *(const float2*)(smem + ST_OFF + j * 8);
      const float2 hs = *(const float2*)(smem + HS_OFF + min(hh, H - 1) * 8);
      const float sc = j < n_tok ? (st.y * (raw - st.x * hs.x) + hs.y) * scale_log2e : -INFINITY;
      *(float*)(smem + SC_OFF + hh * pitch + j * 4) = sc;
    }
    __syncthreads();
  }

  // ================= softmax per head; weights p_j rstd_j as bf16 in place, m_h = sum p_j rstd_j mu_j =================
  stage(0, 0, true);                                            // (pass 2's first tile lands meanwhile)
  for (int hh = w; hh < 16; hh += 8) {
    char* rowp = smem + SC_OFF + hh * pitch;
    float v[10];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int j = lane + 64 * i;
      v[i] = j < ntp ? *(const float*)(rowp + j * 4) : -INFINITY;
      mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 10; ++i) { v[i] = __builtin_amdgcn_exp2f(v[i] - mx); sum += v[i]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    float mm = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every lane holds its scores before the row is overwritten
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int j = lane + 64 * i;
      if (j < ntp) {
        const float2 st = *(const float2*)(smem + ST_OFF + j * 8);
        const float pw = v[i] * inv * st.y;                       // p_j rstd_j (0 for the padded tokens)
        mm += pw * st.x;
        *(bf16_t*)(rowp + j * 2) = f32_to_bf16(pw);
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mm += __shfl_xor(mm, o);
    if (lane == 0 && hh < H) mz[(size_t)crop * H + hh] = mm;
  }

  // ================= pass 2: z_h = sum_j (p_j rstd_j) x_j  (wave w owns columns [w D / 8, (w + 1) D / 8)) =================
  const int nb = D / 128;                                       // 16-column blocks per wave (8 for D = 1024)
  f32x4_t z[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int ti = lane & 15;                                     // transposing read: lane i of a 16-lane group reads token row i >> 2, columns 4 (i & 3) ..
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) stage(t + 1, (t + 1) & 1, true);
    wait_vm(t + 1 < ntiles ? np : 0);
    __syncthreads();                                            // (also: the weights of every head are written)
    const char* xt = smem + X_OFF + (t & 1) * XBUF;
    const s16x4_t a = *(const s16x4_t*)(smem + SC_OFF + m16 * pitch + (t * 16 + kg * 4) * 2);   // weights of head m16, tokens 16 t + 4 kg ..
    const int trow = kg * 4 + (ti >> 2);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i < nb) {
        const int col = (w * nb + i) * 16 + (ti & 3) * 4;       // column of this lane's 4-wide piece
        const int seg = col >> 6, cs = col & 63;
        const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(xt + seg * 2048 + v_swz(trow, cs >> 3) + (cs & 7) * 2));
        z[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, z[i], 0, 0, 0);               // D[head][column]
      }
    }
    __syncthreads();                                            // the buffer is refilled two tiles on
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < nb) {
      const int col = (w * nb + i) * 16 + m16;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int hh = kg * 4 + e;
        if (hh < H) Zp[((size_t)crop * H + hh) * D + col] = f32_to_bf16(z[i][e]);
      }
    }
  }
}

// Qm[(c * H + h)][n] = q_c[n] for n in head h, 0 elsewhere (bf16): the operand whose product with W'_k^T is r_h
__global__ void cls_qmask_kernel(const bf16_t* __restrict__ q, size_t q_stride, bf16_t* __restrict__ Qm, int n_crops, int D, int H) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one 16-B piece (8 columns) per thread
  const size_t per_row = (size_t)D / 8;
  if (i >= (size_t)n_crops * H * per_row) return;
  const size_t row = i / per_row;
  const int c8 = (int)(i - row * per_row) * 8;
  const int crop = (int)(row / H), hh = (int)(row - (size_t)crop * H);
  uint4 v = {0, 0, 0, 0};
  if ((c8 >> 6) == hh) v = *(const uint4*)(q + (size_t)crop * q_stride + c8);
  *(uint4*)(Qm + row * D + c8) = v;
}

// o_c[h * 64 + d] = Of[(c * H + h)][h * 64 + d] - m_h colsum_v[..] + bias_v[..] -> the class-token row of crop c (bf16, or e4m3 with inv)
__global__ void cls_finish_kernel(const float* __restrict__ Of, const float* __restrict__ mz, const float* __restrict__ colsum_v,
                                  const float* __restrict__ bias_v, void* __restrict__ out, size_t out_stride, const float* __restrict__ out_inv,
                                  int n_crops, int D, int H) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n_crops * D) return;
  const int crop = (int)(i / D), n = (int)(i - (size_t)crop * D), hh = n >> 6;
  // Of is fp32: the mean term is subtracted BEFORE the one rounding of o (as the projected path's fp32 epilogue does); with real
  // checkpoints |z . W'_v| can be far larger than |v| (non-zero token means, outlier channels), and a bf16 Of put an error of
  // 2^-9 |Of| on o
  const float v = Of[((size_t)crop * H + hh) * D + n] - mz[(size_t)crop * H + hh] * colsum_v[n] + bias_v[n];
  if (out_inv) {
    const float s = __builtin_amdgcn_fmed3f(v * out_inv[n], -448.0f, 448.0f);
    const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(s, s, 0, false);
    ((uint8_t*)out)[(size_t)crop * out_stride + n] = (uint8_t)(pk & 0xff);
  } else {
    ((bf16_t*)out)[(size_t)crop * out_stride + n] = f32_to_bf16(v);
  }
}

}  // namespace

size_t ce_cls_attn_scratch_elems(int n_crops, int D, int H) { return (size_t)n_crops * H * D; }   // per buffer: Qm, R, Zp (bf16), Of (fp32)

// The shapes cls_attn_kernel is written for -- THE predicate: run_tower (capi.hip) asks it before it launches anything of the
// shortcut, ce_cls_attn refuses on it.
bool ce_cls_attn_supported(int n_crops, int n_tok, int D, int H) {
  if (n_crops < 1 || n_tok < 1 || n_tok > 640 || D < 256 || D > 1024 || D % 256 != 0 || H * 64 != D) return false;
  const int ntp = (n_tok + 15) / 16 * 16;
  return SC_OFF + 16 * (ntp * 4 + 16) <= 160 * 1024;
}

hipError_t ce_cls_qmask(const void* q, size_t q_stride, void* Qm, int n_crops, int D, int H, hipStream_t stream) {
  const size_t pieces = (size_t)n_crops * H * (D / 8);
  hipLaunchKernelGGL(cls_qmask_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)q, q_stride, (bf16_t*)Qm, n_crops, D, H);
  return hipGetLastError();
}

hipError_t ce_cls_attn(const void* x, const float* stats, int parts, int stats_ld, const void* q, size_t q_stride, const float* colsum_k,
                       const float* bias_k, const void* R, void* Zp, float* mz, int n_crops, int n_tok, int D, int H, float eps,
                       hipStream_t stream) {
  if (!ce_cls_attn_supported(n_crops, n_tok, D, H)) return hipErrorInvalidValue;
  const int ntp = (n_tok + 15) / 16 * 16;
  const int lds = SC_OFF + 16 * (ntp * 4 + 16);
  static DeviceKernelSetup setup;
  if (hipError_t e = setup.ensure((const void*)cls_attn_kernel, 160 * 1024, nullptr); e != hipSuccess) return e;
  hipLaunchKernelGGL(cls_attn_kernel, dim3(n_crops), dim3(512), lds, stream, (const bf16_t*)x, stats, parts, stats_ld, (const bf16_t*)q, q_stride,
                     colsum_k, bias_k, (const bf16_t*)R, (bf16_t*)Zp, mz, n_tok, D, H, 1.0f / D, eps, 0.125f * 1.44269504088896340736f);
  return hipGetLastError();
}

hipError_t ce_cls_finish(const void* Of, const float* mz, const float* colsum_v, const float* bias_v, void* out, size_t out_stride,
                         const float* out_inv, int n_crops, int D, int H, hipStream_t stream) {
  const size_t n = (size_t)n_crops * D;
  hipLaunchKernelGGL(cls_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const float*)Of, mz, colsum_v, bias_v, out,
                     out_stride, out_inv, n_crops, D, H);
  return hipGetLastError();
}
