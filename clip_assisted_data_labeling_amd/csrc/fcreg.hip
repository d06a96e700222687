// K10 of SURVEY.md §2.2: the whole SimpleFC regressor forward in ONE kernel.
//   y = sigmoid(W_n . lrelu(... lrelu(W_1 x + b_1) ...) + b_n)
// Follows /root/reference/utils/nn_model.py:21-33 (Linear, LeakyReLU(0.01), Dropout per hidden
// layer; final Linear + Sigmoid) and :38-41; Dropout is the identity at eval
// (/root/reference/_5_predict_labels.py:108).  fp32 throughout, like the reference's `.float()`
// (_5_predict_labels.py:135).
//
// One workgroup per FC_ROWS input rows.  The rows are staged in LDS; thread j owns output neuron j
// of the current layer and streams column j of the TRANSPOSED weight ([in][out], so a wave reads
// 256 contiguous bytes per k), accumulating FC_ROWS rows at once.  Activations ping-pong between
// two LDS buffers; only the final scores go back to HBM.  The input row may be gathered from
// `n_seg` segments (the per-crop embeddings selected by model.crop_names,
// /root/reference/_5_predict_labels.py:79) so the encoder output is consumed in place.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int FC_ROWS = 4;

__global__ void fcreg_kernel(const FcRegParams p, int max_width) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* buf0 = (float*)smem;                          // [FC_ROWS][max_width]
  float* buf1 = buf0 + FC_ROWS * max_width;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int row0 = blockIdx.x * FC_ROWS;
  const int in0 = p.sizes[0];

  // gather the input rows (zeros for rows past the end)
  for (int idx = tid; idx < FC_ROWS * in0; idx += nthr) {
    const int r = idx / in0, k = idx - r * in0;
    float v = 0.f;
    if (row0 + r < p.n_rows) {
      const int s = k / p.seg_len, o = k - s * p.seg_len;
      v = p.x[(size_t)(row0 + r) * p.row_stride + p.seg_off[s] + o];
    }
    buf0[r * max_width + k] = v;
  }
  __syncthreads();

  float* cur = buf0;
  float* nxt = buf1;
  for (int l = 0; l < p.n_layers; ++l) {
    const int in = p.sizes[l], on = p.sizes[l + 1];
    const float* __restrict__ Wt = p.Wt[l];
    const bool last = (l == p.n_layers - 1);
    for (int j = tid; j < on; j += nthr) {
      float acc[FC_ROWS];
#pragma unroll
      for (int r = 0; r < FC_ROWS; ++r) acc[r] = 0.f;
      int k = 0;
      for (; k + 4 <= in; k += 4) {
        const float w0 = Wt[(size_t)(k + 0) * on + j], w1 = Wt[(size_t)(k + 1) * on + j];
        const float w2 = Wt[(size_t)(k + 2) * on + j], w3 = Wt[(size_t)(k + 3) * on + j];
#pragma unroll
        for (int r = 0; r < FC_ROWS; ++r) {
          const float4 xv = *(const float4*)(cur + r * max_width + k);
          acc[r] = fmaf(xv.x, w0, acc[r]); acc[r] = fmaf(xv.y, w1, acc[r]);
          acc[r] = fmaf(xv.z, w2, acc[r]); acc[r] = fmaf(xv.w, w3, acc[r]);
        }
      }
      for (; k < in; ++k) {
        const float w0 = Wt[(size_t)k * on + j];
#pragma unroll
        for (int r = 0; r < FC_ROWS; ++r) acc[r] = fmaf(cur[r * max_width + k], w0, acc[r]);
      }
      const float bj = p.b[l][j];
#pragma unroll
      for (int r = 0; r < FC_ROWS; ++r) {
        float v = acc[r] + bj;
        if (!last) {
          v = v >= 0.f ? v : p.negative_slope * v;
          nxt[r * max_width + j] = v;
        } else {
          v = 1.0f / (1.0f + expf(-v));
          if (row0 + r < p.n_rows) p.y[(size_t)(row0 + r) * on + j] = v;
        }
      }
    }
    __syncthreads();
    float* t = cur; cur = nxt; nxt = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Store-scale path (scoring N stored rows, /root/reference/_5_predict_labels.py:133-135 over a whole dataset): the same
// SimpleFC in ONE kernel, on the fp32 matrix pipe.  v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain (no reduced
// precision) at the fp32 vector rate, with one VGPR per operand, so the arithmetic type stays what the reference computes
// in (`features.float()`).
//
// Orientation: every layer is computed TRANSPOSED, D' = W . h^T, i.e. A operand = 32 neurons of the layer, B operand =
// 32 input rows of this wave.  The 32 x 32 result then has the INPUT ROW on the lane and the neurons in the 16
// registers, which is exactly the B operand of the next layer's MFMA with no lane movement: k-step s of a 32-neuron tile
// takes register s of both lane halves (half h holds neuron 4h + (s & 3) + 8 (s >> 2)), and the next layer's weights are
// simply loaded with that same k order.  Activations never leave the registers; LDS only stages layer 1.
//
// One workgroup = 4 waves = 128 input rows (32 per wave); each wave keeps all T1 <= 9 neuron tiles of layer 1 for its 32
// rows (<= 144 accumulator registers).  Layer 1 streams K in stages of 32: the 128 x 32 block of X (16 KiB; rows gathered
// from the caller's segments) and the T1*32 x 32 block of W1 (<= 36 KiB) land in LDS by LDS-DMA (8 rows x 128 B per
// wave instruction), double-buffered, one barrier per stage (a stage is 16 k-steps x T1 MFMAs of 64 cycles = 9 216 cycles
// at T1 = 9, so the staging is far off the critical path and bank conflicts of the 128-B-row image do not matter).
// Per pair of k-steps a lane reads ONE 8-B pair per operand tile (lane half h: elements 2h, 2h + 1 of the 16-B chunk).
// Layers 2..n read their (tiny, L2-resident) weights straight from global memory, one dword per lane and k-step.
// Algorithmic rate: 2 x 853 056 FLOP per row at 3072-264-128-64-1; the fp32 matrix pipe peaks at 157 TFLOP/s, i.e. 92 M
// rows/s = 1.13 TB/s of input -- the kernel is bound by that pipe, not by HBM.  Measured (tools/bench_fcreg.py, 1 M rows x 3072
// fp32 resident in HBM): 18.3 ms = 703 GB/s of input = 97.6 TFLOP/s (0.62 of the fp32 matrix peak; SQ MFMA-busy 0.68 at 2.30 GHz;
// 264 neurons padded to 9 tiles cost 9 %), against 146 ms for the 4-rows-per-workgroup kernel on the same rows.
constexpr int MF_ROWS = 128, MF_KT = 32, MF_MAXT = 9, MF_MAXT2 = 4;   // layer 1 up to 288 neurons, later layers up to 128

template <int T, int N>
__device__ __forceinline__ void mf_bias_act(f32x16_t (&h)[N], int tiles, const float* bias, int width, int half, float slope,
                                            bool last) {
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t < tiles) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = t * 32 + 4 * half + (r & 3) + 8 * (r >> 2);
        float v = h[t][r] + (n < width ? bias[n] : 0.f);
        if (last) v = 1.0f / (1.0f + expf(-v));
        else v = v >= 0.f ? v : slope * v;
        h[t][r] = n < width ? v : 0.f;                 // padded neurons feed zeros into the next layer
      }
    }
  }
}

// One later layer, transposed: dst[to] = W_l[tile to] . src (all input tiles), then bias + activation.  For (to, ti) a lane
// needs the 16 weights W[to*32 + j][ti*32 + 4h + (s & 3) + 8 (s >> 2)], s = 0..15: four 16-B pieces at columns ti*32 + 8g + 4h,
// fetched with four dwordx4 loads ONE TILE PAIR AHEAD of their 16 MFMAs (issued as 16 dword loads right in front of the
// MFMAs this tail was latency-bound and took as long as layer 1).  `src` / `dst` are statically indexed register arrays.
template <int TS, int TD, int NS, int ND>
__device__ __forceinline__ void mf_layer(const FcRegParams& p, int l, f32x16_t (&src)[NS], f32x16_t (&dst)[ND], int half, int j) {
  const int Tin = (p.sizes[l] + 31) >> 5, Tout = (p.sizes[l + 1] + 31) >> 5;
  const int ldw = Tin * 32;
  const float* wbase = p.Wr[l] + (size_t)j * ldw + 4 * half;
  float4 cur[4], nxt[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) cur[g] = *(const float4*)(wbase + 8 * g);
#pragma unroll
  for (int to = 0; to < TD; ++to) {
    if (to < Tout) {
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ti = 0; ti < TS; ++ti) {
        if (ti < Tin) {
          int nto = to, nti = ti + 1;                          // the pair after this one (the last pair re-reads itself)
          if (nti >= Tin) { nti = 0; nto = to + 1; }
          if (nto >= Tout) { nto = to; nti = ti; }
          const float* wn = wbase + (size_t)nto * 32 * ldw + nti * 32;
#pragma unroll
          for (int g = 0; g < 4; ++g) nxt[g] = *(const float4*)(wn + 8 * g);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < 16; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[ks >> 2][ks & 3], src[ti][ks], acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < 4; ++g) cur[g] = nxt[g];
        }
      }
      dst[to] = acc;
    }
  }
  mf_bias_act<TD>(dst, Tout, p.b[l], p.sizes[l + 1], half, p.negative_slope, l == p.n_layers - 1);
}

template <int T1>                                              // 32-neuron tiles of layer 1, compile time: a branch-free main loop
__global__ __launch_bounds__(256) void fcreg_mfma_kernel(const FcRegParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, j = lane & 31;
  const int row0 = blockIdx.x * MF_ROWS;
  const int in0 = p.sizes[0];
  constexpr int xbytes = MF_ROWS * MF_KT * 4;                  // 16 KiB
  constexpr int stage_bytes = xbytes + T1 * 32 * MF_KT * 4;
  // LDS-DMA pieces of one stage: 8 rows x 128 B each; X: 16 pieces, W1: 4*T1 pieces, dealt round-robin to the 4 waves
  constexpr int n_pieces = 16 + 4 * T1;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)smem);
  const int prow = lane >> 3, pchunk = lane & 7;
  auto stage = [&](int buf, int k0) {
    const int seg = k0 / p.seg_len, ko = k0 - seg * p.seg_len;
    for (int q = w; q < n_pieces; q += 4) {
      // 16-B chunk c of tile row r is stored at position c ^ ((r >> 1) & 7) of its 128-B row (swizzle on the SOURCE address,
      // the LDS image of a piece stays lane-linear): the ds_read_b128 of 32 consecutive rows below is then conflict-free
      // (unswizzled it was 8-way and the loop LDS-bound: 29 ms per 1 M rows instead of 15)
      const int sw = (pchunk ^ ((4 * q + (prow >> 1)) & 7)) * 4;
      const float* src;
      if (q < 16) {
        int r = row0 + q * 8 + prow;
        r = r < p.n_rows ? r : p.n_rows - 1;                   // rows past the end re-read the last row (results are not stored)
        src = p.x + (size_t)r * p.row_stride + p.seg_off[seg] + ko + sw;
      } else {
        src = p.Wr[0] + (size_t)((q - 16) * 8 + prow) * in0 + k0 + sw;
      }
      // LDS-DMA as inline asm: behind the builtin hipcc drains the transfer (vmcnt(0)) in front of the first LDS read that
      // follows -- i.e. BEFORE the stage's MFMAs instead of behind them, which serialised load and compute (MFMA-busy 0.60).
      // The wait is hand-placed in front of the stage-end barrier below.
      const unsigned lds_dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * stage_bytes + q * 1024));   // wave-uniform, provably
      // (M0: the AMDGPU backend keeps m0 RESERVED -- it never holds a value of the register allocator's, and a "m0" clobber is
      //  rejected with -Winline-asm "reserved register"; the compiler's own M0 users (LDS-DMA builtins, movrel) re-initialise it
      //  in front of each use, and this translation unit has none: `tools/kernel_resources.sh fcreg.hip` counts the M0 writes in the ISA)
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_dst) : "memory");
    }
  };

  f32x16_t ha[MF_MAXT], hb[MF_MAXT2];
#pragma unroll
  for (int t = 0; t < MF_MAXT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ha[t][r] = 0.f;

  // ---- layer 1: ha[t] = W1[tile t] . X^T over K = in0, staged through LDS ----
  const int n_stages = in0 / MF_KT;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < n_stages; ++s) {
    const int buf = s & 1;
    if (s + 1 < n_stages) stage(buf ^ 1, (s + 1) * MF_KT);
    const char* xs = smem + buf * stage_bytes + (w * 32 + j) * 128;          // this lane's input row, 32 k
    const char* ws = smem + buf * stage_bytes + xbytes + j * 128;            // + t * 4096: neuron j of tile t
    const int rsw = (j >> 1) & 7;                                            // tile rows 32 w + j / 32 t + j: ((r >> 1) & 7) = (j >> 1) & 7
    // Per 16-B chunk (4 k) lane half h takes the 8-byte pair at offset 8h: k-step 0 multiplies element 2h, k-step 1 element
    // 2h + 1 -- the same k assignment for both operands, so every k is used exactly once -- and the two floats are MFMA
    // operands as they arrive: no per-lane select.  (Selecting elements [h], [h + 2] of a 16-B read cost one VALU write per
    // MFMA into a register the previous MFMA was still reading: the matrix pipe sat idle half the time, SQ MFMA-busy 0.50.)
    // Operands of chunk c + 1 are read before the MFMAs of chunk c are issued (register double buffer): one wave per SIMD has
    // nobody else to cover an LDS round trip.  All tiles for k-step 0, then all for k-step 1: no dependent MFMA pairs.
    const int hoff = half * 8;
    float2 xb = *(const float2*)(xs + ((0 ^ rsw) << 4) + hoff), xn = xb;
    float2 wa[T1], wn[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t) { wa[t] = *(const float2*)(ws + t * 4096 + ((0 ^ rsw) << 4) + hoff); wn[t] = wa[t]; }
#pragma unroll
    for (int c = 0; c < MF_KT / 4; ++c) {
      if (c + 1 < MF_KT / 4) {
        xn = *(const float2*)(xs + (((c + 1) ^ rsw) << 4) + hoff);
#pragma unroll
        for (int t = 0; t < T1; ++t) wn[t] = *(const float2*)(ws + t * 4096 + (((c + 1) ^ rsw) << 4) + hoff);
      }
      __builtin_amdgcn_sched_barrier(0);                       // the reads are ISSUED here; hipcc otherwise sinks them to their use
#pragma unroll
      for (int t = 0; t < T1; ++t) ha[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[t].x, xb.x, ha[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < T1; ++t) ha[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[t].y, xb.y, ha[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      xb = xn;
#pragma unroll
      for (int t = 0; t < T1; ++t) wa[t] = wn[t];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of stage s+1 have landed ...
    __syncthreads();                                           // ... and everybody's; buffer `buf` may be refilled
  }
  mf_bias_act<MF_MAXT>(ha, T1, p.b[0], p.sizes[1], half, p.negative_slope, p.n_layers == 1);

  // ---- layers 2..n: activations stay in registers (ha <-> hb), weights from global memory (mf_layer) ----
  if (p.n_layers > 1) mf_layer<MF_MAXT, MF_MAXT2>(p, 1, ha, hb, half, j);
  if (p.n_layers > 2) mf_layer<MF_MAXT2, MF_MAXT2>(p, 2, hb, ha, half, j);
  if (p.n_layers > 3) mf_layer<MF_MAXT2, MF_MAXT2>(p, 3, ha, hb, half, j);
  // ---- output: neuron n of the last layer sits in register (n & 3) + 4 ((n >> 3) & 3) of lane half (n >> 2) & 1, tile n >> 5 ----
  const int on = p.sizes[p.n_layers];
  const int row = row0 + w * 32 + j;
  if (row < p.n_rows) {
    const bool in_hb = (p.n_layers == 2 || p.n_layers == 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = 4 * half + (r & 3) + 8 * (r >> 2);
      if (n < on) p.y[(size_t)row * on + n] = in_hb ? hb[0][r] : ha[0][r];
    }
  }
}

// shape contract of the store-scale kernel
bool fcreg_mfma_ok(const FcRegParams& p) {
  if (p.n_rows < CE_FC_MFMA_MIN_ROWS || p.n_layers < 1 || p.n_layers > 4) return false;
  for (int l = 0; l < p.n_layers; ++l) if (!p.Wr[l]) return false;
  if (p.sizes[0] % MF_KT != 0 || p.seg_len % MF_KT != 0) return false;
  if (p.n_layers > 1 && p.sizes[1] > MF_MAXT * 32) return false;
  for (int l = 2; l < p.n_layers; ++l) if (p.sizes[l] > MF_MAXT2 * 32) return false;
  if (p.sizes[p.n_layers] > 32) return false;
  if (p.n_layers == 1 && p.sizes[1] > 32) return false;
  if ((p.row_stride % 4) != 0 || ((uintptr_t)p.x & 15)) return false;               // 16-B LDS-DMA chunks
  for (int s = 0; s < p.n_seg; ++s) if (p.seg_off[s] % 4 != 0) return false;
  return true;
}

}  // namespace

hipError_t ce_fcreg_forward(const FcRegParams& p, hipStream_t stream) {
  if (p.n_layers < 1 || p.n_layers > CE_FC_MAX_LAYERS || p.n_rows < 1) return hipErrorInvalidValue;
  if (p.n_seg < 1 || p.n_seg > CE_FC_MAX_SEG || p.n_seg * p.seg_len != p.sizes[0]) return hipErrorInvalidValue;
  if (fcreg_mfma_ok(p)) {
    const int T1 = (p.sizes[1] + 31) / 32;
    const int lds = 2 * (MF_ROWS * MF_KT * 4 + T1 * 32 * MF_KT * 4);
    const dim3 grid((p.n_rows + MF_ROWS - 1) / MF_ROWS);
    switch (T1) {
#define MF_CASE(n) case n: { static DeviceKernelSetup st_; if (hipError_t e = st_.ensure((const void*)fcreg_mfma_kernel<n>, lds, nullptr); e != hipSuccess) return e; \
                             hipLaunchKernelGGL(fcreg_mfma_kernel<n>, grid, dim3(256), lds, stream, p); break; }
      MF_CASE(1) MF_CASE(2) MF_CASE(3) MF_CASE(4) MF_CASE(5) MF_CASE(6) MF_CASE(7) MF_CASE(8) MF_CASE(9)
#undef MF_CASE
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  int max_width = 0, max_out = 0;
  for (int l = 0; l <= p.n_layers; ++l) max_width = p.sizes[l] > max_width ? p.sizes[l] : max_width;
  for (int l = 1; l <= p.n_layers; ++l) max_out = p.sizes[l] > max_out ? p.sizes[l] : max_out;
  max_width = (max_width + 3) & ~3;
  const size_t lds = (size_t)2 * FC_ROWS * max_width * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  int threads = ((max_out + 63) / 64) * 64;
  threads = threads < 64 ? 64 : (threads > 1024 ? 1024 : threads);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)fcreg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(fcreg_kernel, dim3((p.n_rows + FC_ROWS - 1) / FC_ROWS), dim3(threads), lds, stream, p, max_width);
  return hipGetLastError();
}
