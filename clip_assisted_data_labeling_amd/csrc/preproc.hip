// GPU front end of the embed path: the 4-crop geometry + CLIP validation transform
//   crop / black square pad  ->  Resize(R, bicubic, shorter side)  ->  CenterCrop(R)
// of /root/reference/utils/embedder.py:184-251 and :90-92, on a decoded uint8 RGB image that is already in HBM.
// Output: uint8 [n_crops][3][R][R], which clipenc_encode takes as CLIPENC_IN_U8 (ToTensor + Normalize are fused
// into its first kernel).
//
// The resize restates Pillow's two-pass fixed-point resampler (the reference calls it through torchvision's
// Resize on PIL images): per output coordinate a window [xmin, xmin+xmax) and double-precision bicubic weights
// (a = -0.5, support 2 x max(scale,1)), normalised, converted to 22-bit fixed point with round-half-away; the
// horizontal pass writes uint8 (rounded, clipped) and the vertical pass runs on that uint8 intermediate.
// Integer arithmetic: results are BIT-EXACT with Pillow (tests/test_cpu_preproc.py pins the tables and the
// integer passes against PIL on the CPU; tests/test_gpu_preproc.py pins the kernels).
// The coefficient tables are computed by a small kernel in front of the two passes (the same double-precision
// expressions as the host restatement ce_preproc_axis_tables, which the CPU tests pin against PIL; this file is built
// with -ffp-contract=off and IEEE fp64 division, so both produce the same integers): a call is three launches on the
// caller's stream with the per-crop plan passed by value -- no staging copy, no host synchronisation.
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__host__ __device__ double bicubic_filter(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

}  // namespace

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for output coordinates [out0, out0 + n_out) of a resize of
// `in_size` samples to `out_size` samples (box = whole input).  bounds: [n_out][2] = (xmin, xmax); kk: [n_out][ksize].
int ce_preproc_axis_tables(int in_size, int out_size, int out0, int n_out, std::vector<int>& bounds, std::vector<int>& kk) {
  const double support_base = 2.0;
  const float in0 = 0.0f, in1 = (float)in_size;
  double filterscale, scale;
  filterscale = scale = (double)(in1 - in0) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = support_base * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  bounds.assign((size_t)n_out * 2, 0);
  kk.assign((size_t)n_out * ksize, 0);
  std::vector<double> k(ksize);
  for (int i = 0; i < n_out; ++i) {
    const int xx = out0 + i;
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (int x = 0; x < xmax; ++x) {
      const double v = k[x];
      kk[(size_t)i * ksize + x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
    }
    bounds[(size_t)i * 2 + 0] = xmin;
    bounds[(size_t)i * 2 + 1] = xmax;
  }
  return ksize;
}

namespace {

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;                      // arithmetic shift, like Pillow's lookup index
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

struct CropPlan {
  // canvas (virtual source of the resize): canvas(x, y) = image(x - ox, y - oy) inside the image, else 0 (black pad)
  int ox, oy;
  int row0, n_rows;                          // canvas rows needed by the vertical pass: [row0, row0 + n_rows)
  int ksize_h, ksize_v;
  int bounds_h, kk_h, bounds_v, kk_v;        // offsets (in ints) into the table buffer
  int tmp_off;                               // offset (in pixels) into the temp buffer
  int in_h, out_h, off_h, in_v, out_v, off_v; // resize geometry per axis: in_size -> out_size, first output coordinate
  const uint8_t* img; int H, W, pitch;        // the decoded image this crop comes from
};
constexpr int MAX_CROPS = 32;
struct PlanPack { CropPlan p[MAX_CROPS]; };   // 32 x 88 B by value (kernel argument); larger batches pass a device array

// one output coordinate of Pillow's precompute_coeffs + normalize_coeffs_8bpc (see ce_preproc_axis_tables)
__host__ __device__ inline void axis_entry(int in_size, int out_size, int xx, int ksize, int* bounds2, int* kk_row) {
  const float in0 = 0.0f, in1 = (float)in_size;
  double filterscale, scale;
  filterscale = scale = (double)(in1 - in0) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  const double center = in0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) ww += bicubic_filter((x + xmin - center + 0.5) * ss);
  if (kk_row) {
    for (int x = 0; x < ksize; ++x) {
      int q = 0;
      if (x < xmax) {
        double v = bicubic_filter((x + xmin - center + 0.5) * ss);   // recomputed: the same value as in the sum above
        if (ww != 0.0) v /= ww;
        q = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
      }
      kk_row[x] = q;
    }
  }
  bounds2[0] = xmin;
  bounds2[1] = xmax;
}

// grid (2 axes, n_crops), one thread per output coordinate
__global__ __launch_bounds__(256) void preproc_tables_kernel(const PlanPack pack, const CropPlan* __restrict__ dev_plans,
                                                             int* __restrict__ tab, int R) {
  const CropPlan& pl = dev_plans ? dev_plans[blockIdx.y] : pack.p[blockIdx.y];
  const bool vert = blockIdx.x == 1;
  for (int i = threadIdx.x; i < R; i += blockDim.x) {
    if (vert) axis_entry(pl.in_v, pl.out_v, pl.off_v + i, pl.ksize_v, tab + pl.bounds_v + i * 2, tab + pl.kk_v + (size_t)i * pl.ksize_v);
    else axis_entry(pl.in_h, pl.out_h, pl.off_h + i, pl.ksize_h, tab + pl.bounds_h + i * 2, tab + pl.kk_h + (size_t)i * pl.ksize_h);
  }
}

// horizontal pass: tmp[crop][yy][X] = packed RGB of clip8(sum_k canvas(xmin_X + k, row0 + yy) * kh[X][k])
__global__ __launch_bounds__(256) void preproc_h_kernel(const PlanPack pack, const CropPlan* __restrict__ dev_plans,
                                                        const int* __restrict__ tab, uint32_t* __restrict__ tmp, int R) {
  const CropPlan& pl = dev_plans ? dev_plans[blockIdx.y] : pack.p[blockIdx.y];
  const uint8_t* __restrict__ img = pl.img;
  const int H = pl.H, W = pl.W, pitch = pl.pitch;
  const int yy = blockIdx.x;
  if (yy >= pl.n_rows) return;
  const int sy = pl.row0 + yy - pl.oy;       // image row
  for (int X = threadIdx.x; X < R; X += blockDim.x) {
    const int xmin = tab[pl.bounds_h + X * 2], xmax = tab[pl.bounds_h + X * 2 + 1];
    const int* k = tab + pl.kk_h + X * pl.ksize_h;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    if (sy >= 0 && sy < H) {
      const uint8_t* row = img + (size_t)sy * pitch;
      for (int i = 0; i < xmax; ++i) {
        const int sx = xmin + i - pl.ox;
        if (sx >= 0 && sx < W) {
          const int w = k[i];
          s0 += row[sx * 3 + 0] * w; s1 += row[sx * 3 + 1] * w; s2 += row[sx * 3 + 2] * w;
        }
      }
    }
    tmp[(size_t)pl.tmp_off + (size_t)yy * R + X] = (uint32_t)clip8(s0) | ((uint32_t)clip8(s1) << 8) | ((uint32_t)clip8(s2) << 16);
  }
}

// vertical pass: out[crop][c][Y][X] = clip8(sum_k tmp[ymin_Y + k - row0][X][c] * kv[Y][k])
__global__ __launch_bounds__(256) void preproc_v_kernel(const PlanPack pack, const CropPlan* __restrict__ dev_plans,
                                                        const int* __restrict__ tab, const uint32_t* __restrict__ tmp,
                                                        uint8_t* __restrict__ out, int R) {
  const CropPlan& pl = dev_plans ? dev_plans[blockIdx.y] : pack.p[blockIdx.y];
  const int Y = blockIdx.x;
  const int ymin = tab[pl.bounds_v + Y * 2], ymax = tab[pl.bounds_v + Y * 2 + 1];
  const int* k = tab + pl.kk_v + Y * pl.ksize_v;
  uint8_t* o = out + (size_t)blockIdx.y * 3 * R * R + (size_t)Y * R;
  for (int X = threadIdx.x; X < R; X += blockDim.x) {
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    const uint32_t* col = tmp + (size_t)pl.tmp_off + (size_t)(ymin - pl.row0) * R + X;
    for (int i = 0; i < ymax; ++i) {
      const uint32_t px = col[(size_t)i * R];
      const int w = k[i];
      s0 += (int)(px & 255u) * w; s1 += (int)((px >> 8) & 255u) * w; s2 += (int)((px >> 16) & 255u) * w;
    }
    o[X] = (uint8_t)clip8(s0);
    o[(size_t)R * R + X] = (uint8_t)clip8(s1);
    o[(size_t)2 * R * R + X] = (uint8_t)clip8(s2);
  }
}

// torchvision CenterCrop offset: int(round((size - R) / 2.0)) with Python's round-half-to-even
int center_crop_offset(int size, int R) {
  const int d = size - R;                    // >= 0 here
  if (d % 2 == 0) return d / 2;
  const int f = d / 2;                       // d/2.0 = f + 0.5 -> nearest even of {f, f+1}
  return (f % 2 == 0) ? f : f + 1;
}

}  // namespace

struct PreprocState {
  void* dev_tab = nullptr; size_t tab_bytes = 0;
  void* dev_tmp = nullptr; size_t tmp_bytes = 0;
  void* dev_plans = nullptr; size_t plans_bytes = 0;     // batched calls: plans of more than MAX_CROPS crops
  void* pinned = nullptr; size_t pinned_bytes = 0;       //                their host staging copy
  hipEvent_t staged = nullptr;                           //                recorded after the upload that last read `pinned`
};

static hipError_t grow(void** p, size_t* have, size_t need, bool pinned) {
  if (need <= *have) return hipSuccess;
  if (*p) { if (pinned) (void)hipHostFree(*p); else (void)hipFree(*p); *p = nullptr; *have = 0; }
  need = need + need / 2 + 4096;
  hipError_t e = pinned ? hipHostMalloc(p, need, hipHostMallocDefault) : hipMalloc(p, need);
  if (e == hipSuccess) *have = need;
  return e;
}

PreprocState* ce_preproc_create() { return new PreprocState(); }

void ce_preproc_destroy(PreprocState* s) {
  if (!s) return;
  if (s->dev_tab) (void)hipFree(s->dev_tab);
  if (s->dev_tmp) (void)hipFree(s->dev_tmp);
  if (s->dev_plans) (void)hipFree(s->dev_plans);
  if (s->pinned) (void)hipHostFree(s->pinned);
  if (s->staged) (void)hipEventDestroy(s->staged);
  delete s;
}

// boxes: [n_crops][5] = {kind, a, b, c, d}: kind 0 = crop box (left, top, right, bottom) in image coordinates;
// kind 1 = black square canvas (side, paste_x, paste_y, unused) with the image pasted at (paste_x, paste_y).
// One plan per crop.  box: {kind, a, b, c, d}, see the header.  Returns false on a malformed box.
static bool make_plan(CropPlan& pl, const uint8_t* img, int H, int W, int pitch, const int* b, int R) {
  int cw, ch;
  if (b[0] == 0) {
    if (b[1] < 0 || b[2] < 0 || b[3] > W || b[4] > H || b[3] <= b[1] || b[4] <= b[2]) return false;
    cw = b[3] - b[1]; ch = b[4] - b[2]; pl.ox = -b[1]; pl.oy = -b[2];
  } else if (b[0] == 1) {
    if (b[1] < W || b[1] < H || b[2] < 0 || b[3] < 0) return false;
    cw = ch = b[1]; pl.ox = b[2]; pl.oy = b[3];
  } else {
    return false;
  }
  pl.img = img; pl.H = H; pl.W = W; pl.pitch = pitch;
  // Resize(R): shorter side -> R, the other int(R * long / short) (torchvision), then CenterCrop(R)
  int nw, nh;
  if (cw <= ch) { nw = R; nh = (int)((long long)R * ch / cw); } else { nh = R; nw = (int)((long long)R * cw / ch); }
  pl.in_h = cw; pl.out_h = nw; pl.off_h = center_crop_offset(nw, R);
  pl.in_v = ch; pl.out_v = nh; pl.off_v = center_crop_offset(nh, R);
  auto ksize_of = [](int in_size, int out_size) {
    double fs = (double)((float)in_size - 0.0f) / out_size;
    if (fs < 1.0) fs = 1.0;
    return (int)ceil(2.0 * fs) * 2 + 1;
  };
  pl.ksize_h = ksize_of(cw, nw);
  pl.ksize_v = ksize_of(ch, nh);
  // canvas rows the vertical pass reads: window starts and ends are non-decreasing in the output coordinate
  int first[2], last[2];
  axis_entry(pl.in_v, pl.out_v, pl.off_v, pl.ksize_v, first, nullptr);
  axis_entry(pl.in_v, pl.out_v, pl.off_v + R - 1, pl.ksize_v, last, nullptr);
  pl.row0 = first[0]; pl.n_rows = last[0] + last[1] - first[0];
  return true;
}

// images: n_images decoded uint8 HWC images in HBM (host arrays of device pointers / sizes); image i contributes
// crops_per_image[i] consecutive boxes; out: uint8 [total crops][3][R][R] in box order.
hipError_t ce_preproc_crops_u8_batch(PreprocState* s, int n_images, const uint8_t* const* imgs, const int* Hs, const int* Ws,
                                     const int* pitches, const int* crops_per_image, const int* boxes, int R, uint8_t* out,
                                     hipStream_t stream) {
  if (!s || !imgs || !Hs || !Ws || !pitches || !crops_per_image || !boxes || !out || n_images < 1 || R < 1) return hipErrorInvalidValue;
  long total = 0;
  for (int i = 0; i < n_images; ++i) {
    if (!imgs[i] || Hs[i] < 1 || Ws[i] < 1 || pitches[i] < Ws[i] * 3 || crops_per_image[i] < 0) return hipErrorInvalidValue;
    total += crops_per_image[i];
  }
  if (total < 1 || total > 65535) return hipErrorInvalidValue;
  std::vector<CropPlan> plans((size_t)total);
  size_t tab_ints = 0, tmp_pixels = 0;
  int max_rows = 0;
  long c = 0;
  for (int i = 0; i < n_images; ++i)
    for (int j = 0; j < crops_per_image[i]; ++j, ++c) {
      CropPlan& pl = plans[c];
      if (!make_plan(pl, imgs[i], Hs[i], Ws[i], pitches[i], boxes + c * 5, R)) return hipErrorInvalidValue;
      max_rows = std::max(max_rows, pl.n_rows);
      pl.bounds_h = (int)tab_ints; tab_ints += (size_t)R * 2;
      pl.kk_h = (int)tab_ints; tab_ints += (size_t)R * pl.ksize_h;
      pl.bounds_v = (int)tab_ints; tab_ints += (size_t)R * 2;
      pl.kk_v = (int)tab_ints; tab_ints += (size_t)R * pl.ksize_v;
      if (tab_ints > 0x7fffffffull || tmp_pixels > 0x7fffffffull) return hipErrorInvalidValue;
      pl.tmp_off = (int)tmp_pixels;
      tmp_pixels += (size_t)pl.n_rows * R;
    }
  hipError_t e;
  if ((e = grow(&s->dev_tab, &s->tab_bytes, tab_ints * sizeof(int), false)) != hipSuccess) return e;
  if ((e = grow(&s->dev_tmp, &s->tmp_bytes, tmp_pixels * 4, false)) != hipSuccess) return e;
  PlanPack pack;
  const CropPlan* dev_plans = nullptr;
  if (total <= MAX_CROPS) {
    memcpy(pack.p, plans.data(), (size_t)total * sizeof(CropPlan));       // plans travel as the kernel argument
  } else {
    const size_t bytes = (size_t)total * sizeof(CropPlan);
    if (!s->staged && (e = hipEventCreateWithFlags(&s->staged, hipEventDisableTiming)) != hipSuccess) return e;
    else if (s->pinned && (e = hipEventSynchronize(s->staged)) != hipSuccess) return e;   // the previous upload has read `pinned`
    if ((e = grow(&s->pinned, &s->pinned_bytes, bytes, true)) != hipSuccess) return e;
    if ((e = grow(&s->dev_plans, &s->plans_bytes, bytes, false)) != hipSuccess) return e;
    memcpy(s->pinned, plans.data(), bytes);
    if ((e = hipMemcpyAsync(s->dev_plans, s->pinned, bytes, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    if ((e = hipEventRecord(s->staged, stream)) != hipSuccess) return e;
    dev_plans = (const CropPlan*)s->dev_plans;
  }
  // three launches, ordered by the stream (tables and the intermediate are reused by the next call on the same stream)
  hipLaunchKernelGGL(preproc_tables_kernel, dim3(2, (unsigned)total), dim3(256), 0, stream, pack, dev_plans, (int*)s->dev_tab, R);
  hipLaunchKernelGGL(preproc_h_kernel, dim3(max_rows, (unsigned)total), dim3(256), 0, stream, pack, dev_plans, (const int*)s->dev_tab,
                     (uint32_t*)s->dev_tmp, R);
  hipLaunchKernelGGL(preproc_v_kernel, dim3(R, (unsigned)total), dim3(256), 0, stream, pack, dev_plans, (const int*)s->dev_tab,
                     (const uint32_t*)s->dev_tmp, out, R);
  return hipGetLastError();
}

hipError_t ce_preproc_crops_u8(PreprocState* s, const uint8_t* img, int H, int W, int pitch, int n_crops, const int* boxes,
                               int R, uint8_t* out, hipStream_t stream) {
  return ce_preproc_crops_u8_batch(s, 1, &img, &H, &W, &pitch, &n_crops, boxes, R, out, stream);
}
