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
// The coefficient tables are tiny and are computed on the host (plain double arithmetic, no contraction).
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

double bicubic_filter(double x) {
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
};

// horizontal pass: tmp[crop][yy][X] = packed RGB of clip8(sum_k canvas(xmin_X + k, row0 + yy) * kh[X][k])
__global__ __launch_bounds__(256) void preproc_h_kernel(const uint8_t* __restrict__ img, int H, int W, int pitch,
                                                        const CropPlan* __restrict__ plans, const int* __restrict__ tab,
                                                        uint32_t* __restrict__ tmp, int R) {
  const CropPlan pl = plans[blockIdx.y];
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
__global__ __launch_bounds__(256) void preproc_v_kernel(const CropPlan* __restrict__ plans, const int* __restrict__ tab,
                                                        const uint32_t* __restrict__ tmp, uint8_t* __restrict__ out, int R) {
  const CropPlan pl = plans[blockIdx.y];
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
  void* dev_plans = nullptr; size_t plans_bytes = 0;
  void* dev_tmp = nullptr; size_t tmp_bytes = 0;
  void* pinned = nullptr; size_t pinned_bytes = 0;
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
  if (s->dev_plans) (void)hipFree(s->dev_plans);
  if (s->dev_tmp) (void)hipFree(s->dev_tmp);
  if (s->pinned) (void)hipHostFree(s->pinned);
  delete s;
}

// boxes: [n_crops][5] = {kind, a, b, c, d}: kind 0 = crop box (left, top, right, bottom) in image coordinates;
// kind 1 = black square canvas (side, paste_x, paste_y, unused) with the image pasted at (paste_x, paste_y).
hipError_t ce_preproc_crops_u8(PreprocState* s, const uint8_t* img, int H, int W, int pitch, int n_crops, const int* boxes,
                               int R, uint8_t* out, hipStream_t stream) {
  if (!s || !img || !boxes || !out || n_crops < 1 || n_crops > 64 || R < 1 || H < 1 || W < 1 || pitch < W * 3)
    return hipErrorInvalidValue;
  std::vector<int> tab;
  std::vector<CropPlan> plans(n_crops);
  size_t tmp_pixels = 0;
  int max_rows = 0;
  for (int c = 0; c < n_crops; ++c) {
    const int* b = boxes + c * 5;
    int cw, ch;
    CropPlan& pl = plans[c];
    if (b[0] == 0) {
      if (b[1] < 0 || b[2] < 0 || b[3] > W || b[4] > H || b[3] <= b[1] || b[4] <= b[2]) return hipErrorInvalidValue;
      cw = b[3] - b[1]; ch = b[4] - b[2]; pl.ox = -b[1]; pl.oy = -b[2];
    } else if (b[0] == 1) {
      if (b[1] < W || b[1] < H || b[2] < 0 || b[3] < 0) return hipErrorInvalidValue;
      cw = ch = b[1]; pl.ox = b[2]; pl.oy = b[3];
    } else {
      return hipErrorInvalidValue;
    }
    // Resize(R): shorter side -> R, the other int(R * long / short) (torchvision), then CenterCrop(R)
    int nw, nh;
    if (cw <= ch) { nw = R; nh = (int)((long long)R * ch / cw); } else { nh = R; nw = (int)((long long)R * cw / ch); }
    const int left = center_crop_offset(nw, R), top = center_crop_offset(nh, R);
    std::vector<int> bh, kh, bv, kv;
    pl.ksize_h = ce_preproc_axis_tables(cw, nw, left, R, bh, kh);
    pl.ksize_v = ce_preproc_axis_tables(ch, nh, top, R, bv, kv);
    int rmin = bv[0], rmax = bv[0] + bv[1];
    for (int y = 0; y < R; ++y) { rmin = std::min(rmin, bv[y * 2]); rmax = std::max(rmax, bv[y * 2] + bv[y * 2 + 1]); }
    pl.row0 = rmin; pl.n_rows = rmax - rmin;
    max_rows = std::max(max_rows, pl.n_rows);
    pl.bounds_h = (int)tab.size(); tab.insert(tab.end(), bh.begin(), bh.end());
    pl.kk_h = (int)tab.size(); tab.insert(tab.end(), kh.begin(), kh.end());
    pl.bounds_v = (int)tab.size(); tab.insert(tab.end(), bv.begin(), bv.end());
    pl.kk_v = (int)tab.size(); tab.insert(tab.end(), kv.begin(), kv.end());
    pl.tmp_off = (int)tmp_pixels;
    tmp_pixels += (size_t)pl.n_rows * R;
  }
  const size_t tab_b = tab.size() * sizeof(int), plan_b = plans.size() * sizeof(CropPlan);
  hipError_t e;
  if ((e = grow(&s->pinned, &s->pinned_bytes, tab_b + plan_b, true)) != hipSuccess) return e;
  if ((e = grow(&s->dev_tab, &s->tab_bytes, tab_b, false)) != hipSuccess) return e;
  if ((e = grow(&s->dev_plans, &s->plans_bytes, plan_b, false)) != hipSuccess) return e;
  if ((e = grow(&s->dev_tmp, &s->tmp_bytes, tmp_pixels * 4, false)) != hipSuccess) return e;
  // the pinned staging buffer is reused by the next call: wait for the previous call's copies on this stream
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
  memcpy(s->pinned, tab.data(), tab_b);
  memcpy((char*)s->pinned + tab_b, plans.data(), plan_b);
  if ((e = hipMemcpyAsync(s->dev_tab, s->pinned, tab_b, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
  if ((e = hipMemcpyAsync(s->dev_plans, (char*)s->pinned + tab_b, plan_b, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
  hipLaunchKernelGGL(preproc_h_kernel, dim3(max_rows, n_crops), dim3(256), 0, stream, img, H, W, pitch,
                     (const CropPlan*)s->dev_plans, (const int*)s->dev_tab, (uint32_t*)s->dev_tmp, R);
  hipLaunchKernelGGL(preproc_v_kernel, dim3(R, n_crops), dim3(256), 0, stream, (const CropPlan*)s->dev_plans,
                     (const int*)s->dev_tab, (const uint32_t*)s->dev_tmp, out, R);
  return hipGetLastError();
}
