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

}  // namespace

hipError_t ce_fcreg_forward(const FcRegParams& p, hipStream_t stream) {
  if (p.n_layers < 1 || p.n_layers > CE_FC_MAX_LAYERS || p.n_rows < 1) return hipErrorInvalidValue;
  if (p.n_seg < 1 || p.n_seg > CE_FC_MAX_SEG || p.n_seg * p.seg_len != p.sizes[0]) return hipErrorInvalidValue;
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
