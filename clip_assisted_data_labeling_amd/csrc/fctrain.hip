// Regressor training on the device: the optimisation loop of
// /root/reference/_4_train_model.py:199-207 for the SimpleFC of utils/nn_model.py:6-41 --
//   forward (Linear, LeakyReLU, Dropout per hidden layer; Linear, Sigmoid), MSELoss (mean over the batch),
//   backward, torch.optim.Adam (weight_decay added to the gradient, bias-corrected, eps 1e-8)
// as plain fp32 kernels on data that already lives in HBM (the embeddings never go back to the host).  The model is
// tiny (447 k parameters at 1536-264-128-64-1), a step is latency-bound: 12 small launches, all state on the device.
// Dropout masks come from a counter-based hash of (seed, step, layer, row, column): a function of the counters only, so
// the parity tests can evaluate the same masks on the CPU (torch's generator is not reproducible outside torch).
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int MAXL = CE_FC_MAX_LAYERS;

__host__ __device__ inline uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool keep_elem(uint32_t layer_key, int r, int c, uint32_t thr) {
  return lowbias32(layer_key ^ lowbias32((uint32_t)r * 0x85ebca6bu + (uint32_t)c + 1u)) >= thr;
}

// Everything that changes from step to step lives in this device block, so that the launches of a step are identical
// and a whole epoch can be replayed as ONE hipGraph (the loop is launch-bound: 13 tiny kernels per step).
struct Ctl {
  long long step;                // optimisation steps taken so far (Adam's t - 1, dropout counter)
  long long n_order;             // rows in this epoch
  const long long* order;        // row indices, or NULL for 0..n_order-1
  const float* X; const float* T;
  float* losses;                 // per-batch MSE out, or NULL
  int b0, batch, bi;             // first row of the current batch within the epoch, batch size, batch index
  float lr, wd, p_drop, inv_keep;
  uint32_t seed, thr;
  float step_size, inv_sqrt_c2;  // written by the delta kernel for the step's update launches
};
__device__ __forceinline__ int batch_rows(const Ctl* c) { return (int)min((long long)c->batch, c->n_order - c->b0); }
__device__ __forceinline__ uint32_t layer_key_dev(const Ctl* c, int layer) {
  return lowbias32(c->seed ^ lowbias32((uint32_t)c->step + 0x9e3779b9u * (uint32_t)(layer + 1)));
}

__global__ __launch_bounds__(256) void fct_gather_kernel(const Ctl* __restrict__ c, int d, float* __restrict__ a0, float* __restrict__ t) {
  const int r = blockIdx.x;
  if (r >= batch_rows(c)) return;
  const long long src = c->order ? c->order[c->b0 + r] : (long long)c->b0 + r;
  for (int k = threadIdx.x; k < d; k += 256) a0[(size_t)r * d + k] = c->X[(size_t)src * d + k];
  if (threadIdx.x == 0 && c->T) t[r] = c->T[src];
}

// one wave per output neuron j; rows in tiles of 16.  last: sigmoid; else LeakyReLU + dropout (train != 0 and p > 0)
__global__ __launch_bounds__(64) void fct_forward_kernel(const Ctl* __restrict__ c, int layer, int train, const float* __restrict__ a_in,
                                                         const float* __restrict__ W, const float* __restrict__ b, int n_in, int n_out,
                                                         int last, float slope, float* __restrict__ z, float* __restrict__ a_out) {
  const int j = blockIdx.x, lane = threadIdx.x;
  const int B = batch_rows(c);
  const uint32_t thr = (train && !last) ? c->thr : 0u;
  const uint32_t lkey = layer_key_dev(c, layer);
  const float inv_keep = c->inv_keep;
  const float* w = W + (size_t)j * n_in;
  for (int r0 = 0; r0 < B; r0 += 16) {
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nr = min(16, B - r0);
    for (int k = lane; k < n_in; k += 64) {
      const float wv = w[k];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (r < nr) acc[r] = fmaf(a_in[(size_t)(r0 + r) * n_in + k], wv, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[r] += __shfl_xor(acc[r], o);
    }
    if (lane < nr) {
      float zz = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) if (r == lane) zz = acc[r];
      zz += b[j];
      const int row = r0 + lane;
      z[(size_t)row * n_out + j] = zz;
      float av;
      if (last) av = 1.0f / (1.0f + expf(-zz));
      else {
        av = zz > 0.f ? zz : slope * zz;
        if (thr) av = keep_elem(lkey, row, j, thr) ? av * inv_keep : 0.f;
      }
      a_out[(size_t)row * n_out + j] = av;
    }
  }
}

// output layer (one neuron): dz = 2/B (y - t) y (1 - y); batch MSE -> losses[bi]; Adam's scalars of this step
__global__ __launch_bounds__(256) void fct_delta_kernel(Ctl* __restrict__ c, const float* __restrict__ y, const float* __restrict__ t,
                                                        float* __restrict__ dz) {
  __shared__ float red[256];
  const int B = batch_rows(c);
  float s = 0.f;
  for (int r = threadIdx.x; r < B; r += 256) {
    const float yy = y[r], e = yy - t[r];
    dz[r] = (2.0f / (float)B) * e * yy * (1.0f - yy);
    s = fmaf(e, e, s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    if (c->losses) c->losses[c->bi] = red[0] / (float)B;
    const double tt = (double)(c->step + 1);
    const double c1 = 1.0 - pow(0.9, tt), c2 = 1.0 - pow(0.999, tt);
    c->step_size = (float)((double)c->lr / c1);
    c->inv_sqrt_c2 = (float)(1.0 / sqrt(c2));
  }
}

// dz_prev[r][k] = (sum_j dz[r][j] W[j][k]) * lrelu'(z_prev[r][k]) * dropout;  W: [n_out][n_in], k < n_in; layer = the hidden layer of z_prev
__global__ __launch_bounds__(256) void fct_backward_input_kernel(const Ctl* __restrict__ c, int layer, const float* __restrict__ dz,
                                                                 const float* __restrict__ W, const float* __restrict__ z_prev, int n_in,
                                                                 int n_out, float slope, float* __restrict__ dz_prev) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= batch_rows(c) * n_in) return;
  const int r = idx / n_in, k = idx - r * n_in;
  float s = 0.f;
  for (int j = 0; j < n_out; ++j) s = fmaf(dz[(size_t)r * n_out + j], W[(size_t)j * n_in + k], s);
  if (c->thr) s = keep_elem(layer_key_dev(c, layer), r, k, c->thr) ? s * c->inv_keep : 0.f;
  dz_prev[idx] = s * (z_prev[idx] > 0.f ? 1.0f : slope);
}

// Adam on W [n_out][n_in] and b [n_out]:  g = dz^T a_in + wd * p
__global__ __launch_bounds__(256) void fct_update_kernel(const Ctl* __restrict__ c, const float* __restrict__ dz, const float* __restrict__ a_in,
                                                         int n_in, int n_out, float* __restrict__ W, float* __restrict__ mW,
                                                         float* __restrict__ vW, float* __restrict__ b, float* __restrict__ mb,
                                                         float* __restrict__ vb) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long nW = (long)n_out * n_in;
  if (idx >= nW + n_out) return;
  const int B = batch_rows(c);
  float g = 0.f;
  float *p, *m, *v;
  if (idx < nW) {
    const int j = (int)(idx / n_in), k = (int)(idx - (long)j * n_in);
    for (int r = 0; r < B; ++r) g = fmaf(dz[(size_t)r * n_out + j], a_in[(size_t)r * n_in + k], g);
    p = W + idx; m = mW + idx; v = vW + idx;
  } else {
    const int j = (int)(idx - nW);
    for (int r = 0; r < B; ++r) g += dz[(size_t)r * n_out + j];
    p = b + j; m = mb + j; v = vb + j;
  }
  g = fmaf(c->wd, *p, g);
  const float mm = 0.9f * *m + 0.1f * g;
  const float vv = 0.999f * *v + 0.001f * g * g;
  *m = mm; *v = vv;
  *p = *p - c->step_size * mm / (sqrtf(vv) * c->inv_sqrt_c2 + 1e-8f);
}

__global__ void fct_advance_kernel(Ctl* c) {
  c->step += 1; c->b0 += c->batch; c->bi += 1;
}

}  // namespace

struct FcTrainState {
  int n_layers = 0;
  int sizes[MAXL + 1];
  float slope = 0.01f;
  void* slab = nullptr;                       // W, b, mW, vW, mb, vb of every layer, then the control block
  float *W[MAXL], *b[MAXL], *mW[MAXL], *vW[MAXL], *mb[MAXL], *vb[MAXL];
  Ctl* ctl = nullptr;                         // device
  Ctl* ctl_host = nullptr;                    // page-locked staging of the epoch's control block
  void* ws = nullptr; size_t ws_bytes = 0; int ws_rows = 0;
  float *a[MAXL + 1], *z[MAXL], *dz[MAXL], *t = nullptr;
  long long step = 0;                         // optimisation steps taken (mirror of ctl->step)
  // one epoch as a graph: valid for (graph_batch, graph_steps) and the current workspace
  hipGraphExec_t graph = nullptr; int graph_batch = 0; long graph_steps = 0;
  hipStream_t cap_stream = nullptr;
  int use_graph = 1;
};

FcTrainState* ce_fctrain_create(int n_layers, const int* sizes, const float* const* W, const float* const* b, float slope,
                                hipError_t* err) {
  *err = hipErrorInvalidValue;
  if (n_layers < 1 || n_layers > MAXL || sizes[n_layers] != 1) return nullptr;
  FcTrainState* s = new FcTrainState();
  s->n_layers = n_layers; s->slope = slope;
  size_t total = 0;
  for (int l = 0; l <= n_layers; ++l) { if (sizes[l] < 1) { delete s; return nullptr; } s->sizes[l] = sizes[l]; }
  for (int l = 0; l < n_layers; ++l) total += 3 * ((size_t)sizes[l + 1] * sizes[l] + sizes[l + 1]);
  const size_t ctl_off = (total * 4 + 255) & ~(size_t)255;
  if ((*err = hipMalloc(&s->slab, ctl_off + sizeof(Ctl))) != hipSuccess) { delete s; return nullptr; }
  auto bail = [&](hipError_t e) -> FcTrainState* { *err = e; (void)hipFree(s->slab); if (s->ctl_host) (void)hipHostFree(s->ctl_host); delete s; return nullptr; };
  if ((*err = hipMemset(s->slab, 0, ctl_off + sizeof(Ctl))) != hipSuccess) return bail(*err);
  if ((*err = hipHostMalloc((void**)&s->ctl_host, sizeof(Ctl), hipHostMallocDefault)) != hipSuccess) return bail(*err);
  s->ctl = (Ctl*)((char*)s->slab + ctl_off);
  float* p = (float*)s->slab;
  for (int l = 0; l < n_layers; ++l) {
    const size_t nw = (size_t)sizes[l + 1] * sizes[l], nb = sizes[l + 1];
    s->W[l] = p; p += nw; s->b[l] = p; p += nb; s->mW[l] = p; p += nw; s->vW[l] = p; p += nw; s->mb[l] = p; p += nb; s->vb[l] = p; p += nb;
    if ((*err = hipMemcpy(s->W[l], W[l], nw * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (*err = hipMemcpy(s->b[l], b[l], nb * 4, hipMemcpyHostToDevice)) != hipSuccess) return bail(*err);
  }
  const char* env = getenv("CLIPENC_TRAIN_GRAPH");
  s->use_graph = env ? atoi(env) : 1;
  *err = hipSuccess;
  return s;
}

static void drop_graph(FcTrainState* s) {
  if (s->graph) { (void)hipGraphExecDestroy(s->graph); s->graph = nullptr; s->graph_batch = 0; s->graph_steps = 0; }
}

void ce_fctrain_destroy(FcTrainState* s) {
  if (!s) return;
  drop_graph(s);
  if (s->cap_stream) (void)hipStreamDestroy(s->cap_stream);
  if (s->slab) (void)hipFree(s->slab);
  if (s->ws) (void)hipFree(s->ws);
  if (s->ctl_host) (void)hipHostFree(s->ctl_host);
  delete s;
}

static hipError_t ensure_ws(FcTrainState* s, int rows) {
  if (rows <= s->ws_rows) return hipSuccess;
  drop_graph(s);                                                // the graph holds workspace pointers
  if (s->ws) { (void)hipFree(s->ws); s->ws = nullptr; s->ws_rows = 0; }
  size_t per_row = 1;                                           // t
  for (int l = 0; l <= s->n_layers; ++l) per_row += s->sizes[l];            // a[l]
  for (int l = 0; l < s->n_layers; ++l) per_row += 2 * (size_t)s->sizes[l + 1];   // z[l], dz[l]
  hipError_t e = hipMalloc(&s->ws, per_row * rows * 4);
  if (e != hipSuccess) return e;
  float* p = (float*)s->ws;
  for (int l = 0; l <= s->n_layers; ++l) { s->a[l] = p; p += (size_t)rows * s->sizes[l]; }
  for (int l = 0; l < s->n_layers; ++l) { s->z[l] = p; p += (size_t)rows * s->sizes[l + 1]; s->dz[l] = p; p += (size_t)rows * s->sizes[l + 1]; }
  s->t = p;
  s->ws_rows = rows;
  return hipSuccess;
}

static void launch_forward(FcTrainState* s, int train, hipStream_t st) {
  for (int l = 0; l < s->n_layers; ++l) {
    const int last = l == s->n_layers - 1;
    hipLaunchKernelGGL(fct_forward_kernel, dim3(s->sizes[l + 1]), dim3(64), 0, st, s->ctl, l, train, s->a[l], s->W[l], s->b[l], s->sizes[l],
                       s->sizes[l + 1], last, s->slope, s->z[l], s->a[l + 1]);
  }
}

// the launches of one optimisation step; every step-dependent value is read from the control block on the device
static void launch_step(FcTrainState* s, int batch, hipStream_t st) {
  const int L = s->n_layers;
  hipLaunchKernelGGL(fct_gather_kernel, dim3(batch), dim3(256), 0, st, s->ctl, s->sizes[0], s->a[0], s->t);
  launch_forward(s, 1, st);
  hipLaunchKernelGGL(fct_delta_kernel, dim3(1), dim3(256), 0, st, s->ctl, s->a[L], s->t, s->dz[L - 1]);
  for (int l = L - 1; l >= 1; --l) {                            // input gradients first: they need the weights before the update
    const int n = batch * s->sizes[l];
    hipLaunchKernelGGL(fct_backward_input_kernel, dim3((n + 255) / 256), dim3(256), 0, st, s->ctl, l - 1, s->dz[l], s->W[l], s->z[l - 1],
                       s->sizes[l], s->sizes[l + 1], s->slope, s->dz[l - 1]);
  }
  for (int l = 0; l < L; ++l) {
    const long n = (long)s->sizes[l + 1] * s->sizes[l] + s->sizes[l + 1];
    hipLaunchKernelGGL(fct_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s->ctl, s->dz[l], s->a[l], s->sizes[l],
                       s->sizes[l + 1], s->W[l], s->mW[l], s->vW[l], s->b[l], s->mb[l], s->vb[l]);
  }
  hipLaunchKernelGGL(fct_advance_kernel, dim3(1), dim3(1), 0, st, s->ctl);
}

static hipError_t upload_ctl(FcTrainState* s, const Ctl& c, hipStream_t st) {
  hipError_t e = hipStreamSynchronize(st);                      // the staging block is reused: previous upload done (and epoch finished)
  if (e != hipSuccess) return e;
  *s->ctl_host = c;
  return hipMemcpyAsync(s->ctl, s->ctl_host, sizeof(Ctl), hipMemcpyHostToDevice, st);
}

// One pass over `order` (n_order row indices into X / T; NULL = rows 0..n_order-1) in batches of `batch_size`: an Adam step
// per batch; losses[i] = MSE of batch i before its update.  The whole epoch is one hipGraph launch (captured once per
// (batch size, step count)); CLIPENC_TRAIN_GRAPH=0 replays the same launches directly.
hipError_t ce_fctrain_epoch(FcTrainState* s, const float* X, const float* T, const long long* order, long n_order, int batch_size,
                            float lr, float wd, float p_drop, uint32_t seed, float* losses, hipStream_t st) {
  if (!s || !X || !T || n_order < 1 || batch_size < 1 || batch_size > 65535 || p_drop < 0.f || p_drop >= 1.f) return hipErrorInvalidValue;
  hipError_t e = ensure_ws(s, batch_size);
  if (e != hipSuccess) return e;
  const long steps = (n_order + batch_size - 1) / batch_size;
  Ctl c{};
  c.step = s->step; c.n_order = n_order; c.order = order; c.X = X; c.T = T; c.losses = losses;
  c.b0 = 0; c.batch = batch_size; c.bi = 0; c.lr = lr; c.wd = wd; c.p_drop = p_drop;
  c.thr = p_drop > 0.f ? (uint32_t)std::min<double>((double)p_drop * 4294967296.0, 4294967295.0) : 0u;
  c.inv_keep = c.thr ? 1.0f / (1.0f - p_drop) : 1.0f;
  c.seed = seed;
  if ((e = upload_ctl(s, c, st)) != hipSuccess) return e;
  if (s->use_graph && (s->graph == nullptr || s->graph_batch != batch_size || s->graph_steps != steps)) {
    drop_graph(s);
    if (!s->cap_stream && (e = hipStreamCreateWithFlags(&s->cap_stream, hipStreamNonBlocking)) != hipSuccess) return e;
    hipGraph_t g = nullptr;
    if ((e = hipStreamBeginCapture(s->cap_stream, hipStreamCaptureModeThreadLocal)) == hipSuccess) {
      for (long i = 0; i < steps; ++i) launch_step(s, batch_size, s->cap_stream);
      e = hipStreamEndCapture(s->cap_stream, &g);
    }
    if (e == hipSuccess && g) e = hipGraphInstantiate(&s->graph, g, nullptr, nullptr, 0);
    if (g) (void)hipGraphDestroy(g);
    if (e != hipSuccess || !s->graph) { (void)hipGetLastError(); s->graph = nullptr; s->use_graph = 0; }   // fall back for good
    else { s->graph_batch = batch_size; s->graph_steps = steps; }
  }
  if (s->use_graph && s->graph) {
    if ((e = hipGraphLaunch(s->graph, st)) != hipSuccess) return e;
  } else {
    for (long i = 0; i < steps; ++i) launch_step(s, batch_size, st);
  }
  s->step += steps;
  return hipGetLastError();
}

// eval-mode forward of rows [0, n) of X: y [n]
hipError_t ce_fctrain_predict(FcTrainState* s, const float* X, long n, float* y, hipStream_t st) {
  if (!s || !X || !y || n < 0) return hipErrorInvalidValue;
  const int chunk = 4096;
  hipError_t e = ensure_ws(s, (int)std::min<long>(chunk, std::max<long>(n, 1)));
  if (e != hipSuccess) return e;
  for (long b0 = 0; b0 < n; b0 += s->ws_rows) {
    const int B = (int)std::min<long>(s->ws_rows, n - b0);
    Ctl c{};
    c.step = s->step; c.n_order = B; c.order = nullptr; c.X = X + (size_t)b0 * s->sizes[0]; c.T = nullptr; c.batch = B; c.inv_keep = 1.0f;
    if ((e = upload_ctl(s, c, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(fct_gather_kernel, dim3(B), dim3(256), 0, st, s->ctl, s->sizes[0], s->a[0], s->t);
    launch_forward(s, 0, st);
    if ((e = hipMemcpyAsync(y + b0, s->a[s->n_layers], (size_t)B * 4, hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
  }
  return hipGetLastError();
}

hipError_t ce_fctrain_get_params(FcTrainState* s, int layer, float* W_host, float* b_host) {
  if (!s || layer < 0 || layer >= s->n_layers) return hipErrorInvalidValue;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return e;
  if (W_host && (e = hipMemcpy(W_host, s->W[layer], (size_t)s->sizes[layer + 1] * s->sizes[layer] * 4, hipMemcpyDeviceToHost)) != hipSuccess) return e;
  if (b_host && (e = hipMemcpy(b_host, s->b[layer], (size_t)s->sizes[layer + 1] * 4, hipMemcpyDeviceToHost)) != hipSuccess) return e;
  return hipSuccess;
}
