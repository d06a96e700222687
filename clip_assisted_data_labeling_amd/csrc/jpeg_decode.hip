// Baseline JPEG -> RGB on the GPU: the input side of the embed path.  The reference decodes every file in DataLoader workers
// (/root/reference/utils/embedder.py:167, PIL.Image.open(...).convert('RGB')) and is bound by that on real data; here the host
// only walks the markers (jpeg_host.cpp) and the device does the rest with the arithmetic of jpeg_core.h, which is Pillow's
// (libjpeg-turbo defaults) bit for bit:
//   jpeg_entropy_kernel   one workgroup per image: its four Huffman tables go to LDS, lane 0 walks the scan (inherently
//                         serial: every code's position depends on all codes before it) and writes the nonzero coefficients;
//                         the parallelism is across the images of the batch, which is what the embed driver has plenty of
//   jpeg_idct_kernel      one thread per 8 x 8 block of any image: dequantise, integer inverse DCT, samples into the plane
//   jpeg_colour_kernel    one thread per output pixel: chroma upsampling (triangle filters) + YCbCr -> RGB, interleaved uint8
// A batch is planned on the host in one pass (per image: padded entropy segment, coefficient planes, sample planes) into one
// device arena that grows to the largest batch seen; descriptors and entropy segments travel in ONE host-to-device copy from
// page-locked memory.  The RGB output goes to memory the caller owns.
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "jpeg_host.h"
#include "kernels.h"

using jpg::ImageDesc;

namespace {

__global__ __launch_bounds__(64) void jpeg_entropy_kernel(const ImageDesc* __restrict__ descs, int* __restrict__ status,
                                                          uint8_t* __restrict__ arena) {
  __shared__ jpg::HuffTable tabs[4];
  __shared__ uint8_t zz[64];
  const ImageDesc& d = descs[blockIdx.x];
  {
    const uint32_t* src = (const uint32_t*)d.huff;
    uint32_t* dst = (uint32_t*)tabs;
    for (int i = threadIdx.x; i < (int)(sizeof(tabs) / 4); i += 64) dst[i] = src[i];
    zz[threadIdx.x] = (uint8_t)jpg::zigzag_to_natural(threadIdx.x);
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  int16_t* coef[jpg::MAX_COMPS];
  for (int c = 0; c < jpg::MAX_COMPS; ++c) coef[c] = (int16_t*)(arena + d.coef_off[c]);
  status[blockIdx.x] = jpg::decode_scan(jpg::scan_geom(d), arena + d.data_off, coef, tabs, zz);
}

// image of a batch-wide index: descs[i].base <= idx < descs[i + 1].base
template <typename F>
__device__ __forceinline__ int find_image(const ImageDesc* descs, int n, uint64_t idx, F base_of) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (base_of(descs[mid]) <= idx) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void jpeg_idct_kernel(const ImageDesc* __restrict__ descs, int n, uint64_t total_blocks,
                                                        uint8_t* __restrict__ arena) {
  const uint64_t gb = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (gb >= total_blocks) return;
  const int img = find_image(descs, n, gb, [](const ImageDesc& d) { return d.block_base; });
  const ImageDesc& d = descs[img];
  uint32_t b = (uint32_t)(gb - d.block_base);
  int c = 0;
  while (c < d.ncomp - 1 && b >= (uint32_t)(d.bw[c] * d.bh[c])) { b -= (uint32_t)(d.bw[c] * d.bh[c]); ++c; }
  const int by = (int)(b / (uint32_t)d.bw[c]), bx = (int)(b - (uint32_t)by * d.bw[c]);
  const int pitch = d.bw[c] * 8;
  __attribute__((aligned(16))) int16_t coef[64];
  const uint4* src = (const uint4*)(arena + d.coef_off[c] + (size_t)b * 128);
#pragma unroll
  for (int i = 0; i < 8; ++i) ((uint4*)coef)[i] = src[i];
  jpg::idct_block(coef, d.quant[c], arena + d.plane_off[c] + (size_t)by * 8 * pitch + bx * 8, pitch);
}

__global__ __launch_bounds__(256) void jpeg_colour_kernel(const ImageDesc* __restrict__ descs, const uint8_t* __restrict__ arena,
                                                          uint8_t* __restrict__ rgb) {
  const ImageDesc& d = descs[blockIdx.y];
  const uint32_t pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= (uint32_t)d.width * (uint32_t)d.height) return;
  const int y = (int)(pix / (uint32_t)d.width), x = (int)(pix - (uint32_t)y * d.width);
  uint8_t* o = rgb + d.rgb_off + (size_t)pix * 3;
  const int Y = arena[d.plane_off[0] + (size_t)y * d.bw[0] * 8 + x];
  if (d.ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; return; }
  const int h = d.hmax / d.hs[1], v = d.vmax / d.vs[1];
  const int cb = jpg::upsampled(arena + d.plane_off[1], d.bw[1] * 8, d.dw[1], d.dh[1], h, v, x, y);
  const int cr = jpg::upsampled(arena + d.plane_off[2], d.bw[2] * 8, d.dw[2], d.dh[2], h, v, x, y);
  uint8_t px[3];
  jpg::ycc_to_rgb(Y, cb, cr, px);
  o[0] = px[0]; o[1] = px[1]; o[2] = px[2];
}

size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct JpegDecState {
  // the plan of the current batch (made by ce_jpegdec_plan, run by ce_jpegdec_run)
  std::vector<ImageDesc> descs;            // decodable images only, in input order
  std::vector<int> input_index;            // descs[i] = input file input_index[i]
  std::vector<std::pair<const uint8_t*, size_t>> scans;
  size_t stage_bytes = 0, arena_bytes = 0, desc_begin = 0, coef_begin = 0, coef_bytes = 0, max_pixels = 0;
  uint64_t total_blocks = 0, rgb_bytes = 0;
  // buffers (grow only)
  void* arena = nullptr; size_t arena_cap = 0;
  void* stage = nullptr; size_t stage_cap = 0;          // page-locked: [descs][entropy segments]
};

JpegDecState* ce_jpegdec_create() { return new JpegDecState(); }

void ce_jpegdec_destroy(JpegDecState* s) {
  if (!s) return;
  if (s->arena) (void)hipFree(s->arena);
  if (s->stage) (void)hipHostFree(s->stage);
  delete s;
}

// Host pass over n files: status[i] = 0 (will be decoded) or the reason it cannot be (jpeg_host.h); widths / heights for every
// file whose header could be read; rgb_offsets[i] = offset of image i's uint8 [H][W][3] in the caller's output buffer of
// *rgb_bytes bytes (256-byte aligned images).
void ce_jpegdec_plan(JpegDecState* s, const void* const* files, const size_t* sizes, int n, int* status, int* widths, int* heights,
                     unsigned long long* rgb_offsets, unsigned long long* rgb_bytes) {
  s->descs.clear(); s->input_index.clear(); s->scans.clear();
  s->descs.reserve((size_t)n);
  size_t rgb = 0;
  uint64_t blocks = 0;
  s->max_pixels = 0;
  for (int i = 0; i < n; ++i) {
    ImageDesc d;
    size_t so = 0, sl = 0;
    const int rc = files[i] ? jpg::parse_jpeg((const uint8_t*)files[i], sizes[i], &d, &so, &sl) : jpg::JPG_NOT_JPEG;
    status[i] = rc; widths[i] = d.width; heights[i] = d.height; rgb_offsets[i] = 0;
    if (rc) continue;
    d.rgb_off = rgb; rgb_offsets[i] = rgb;
    rgb += up((size_t)d.width * d.height * 3, 256);
    d.block_base = blocks;
    uint32_t nb = 0;
    for (int c = 0; c < d.ncomp; ++c) nb += (uint32_t)(d.bw[c] * d.bh[c]);
    d.n_blocks = nb; blocks += nb;
    s->max_pixels = std::max(s->max_pixels, (size_t)d.width * d.height);
    s->descs.push_back(d);
    s->input_index.push_back(i);
    s->scans.emplace_back((const uint8_t*)files[i] + so, sl);
  }
  // arena: [status words][descs][entropy segments][coefficients][sample planes]; the first three are what the staging buffer holds
  const size_t m = s->descs.size();
  s->desc_begin = up(m * sizeof(int), 256);
  size_t off = s->desc_begin + up(m * sizeof(ImageDesc), 256);
  for (size_t i = 0; i < m; ++i) {
    ImageDesc& d = s->descs[i];
    d.data_off = off;
    d.data_len = (uint32_t)(up(s->scans[i].second, 16) + 32);
    d.data_real = (uint32_t)s->scans[i].second;
    off += d.data_len;
  }
  off = up(off, 256);
  s->stage_bytes = off;
  s->coef_begin = off;
  for (auto& d : s->descs)
    for (int c = 0; c < d.ncomp; ++c) { d.coef_off[c] = off; off += (size_t)d.bw[c] * d.bh[c] * 128; }
  off = up(off, 256);
  s->coef_bytes = off - s->coef_begin;
  for (auto& d : s->descs)
    for (int c = 0; c < d.ncomp; ++c) { d.plane_off[c] = off; off += up((size_t)d.bw[c] * d.bh[c] * 64, 16); }
  s->arena_bytes = up(off, 256);
  s->total_blocks = blocks;
  s->rgb_bytes = rgb;
  *rgb_bytes = rgb;
}

// Decodes the planned batch into rgb_dev (>= the plan's rgb_bytes); synchronises the stream; dev_status[i] for every INPUT file:
// unchanged for the ones the plan refused, 0 or 100 + code for the decoded ones (entropy data ran out / invalid code).
hipError_t ce_jpegdec_run(JpegDecState* s, void* rgb_dev, int* status, hipStream_t stream) {
  const size_t m = s->descs.size();
  if (m == 0) return hipSuccess;
  if (s->arena_cap < s->arena_bytes) {
    if (s->arena) (void)hipFree(s->arena);
    s->arena = nullptr; s->arena_cap = 0;
    const size_t want = s->arena_bytes + s->arena_bytes / 4;
    if (hipError_t e = hipMalloc(&s->arena, want); e != hipSuccess) return e;
    s->arena_cap = want;
  }
  if (s->stage_cap < s->stage_bytes) {
    if (s->stage) (void)hipHostFree(s->stage);
    s->stage = nullptr; s->stage_cap = 0;
    const size_t want = s->stage_bytes + s->stage_bytes / 4;
    if (hipError_t e = hipHostMalloc(&s->stage, want, hipHostMallocDefault); e != hipSuccess) return e;
    s->stage_cap = want;
  }
  uint8_t* st = (uint8_t*)s->stage;
  memset(st, 0xff, m * sizeof(int));                          // status words: -1 until the entropy kernel has written them
  memcpy(st + s->desc_begin, s->descs.data(), m * sizeof(ImageDesc));
  for (size_t i = 0; i < m; ++i) {
    const ImageDesc& d = s->descs[i];
    uint8_t* dst = st + d.data_off;
    memcpy(dst, s->scans[i].first, s->scans[i].second);
    for (size_t k = s->scans[i].second; k < d.data_len; ++k) dst[k] = ((k - s->scans[i].second) & 1) ? 0xD9 : 0xFF;   // EOI markers: a scan
  }                                                                                                                    // that runs long ends in them
  uint8_t* arena = (uint8_t*)s->arena;
  if (hipError_t e = hipMemcpyAsync(arena, st, s->stage_bytes, hipMemcpyHostToDevice, stream); e != hipSuccess) return e;
  if (hipError_t e = hipMemsetAsync(arena + s->coef_begin, 0, s->coef_bytes, stream); e != hipSuccess) return e;
  const ImageDesc* descs = (const ImageDesc*)(arena + s->desc_begin);
  hipLaunchKernelGGL(jpeg_entropy_kernel, dim3((unsigned)m), dim3(64), 0, stream, descs, (int*)arena, arena);
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((s->total_blocks + 255) / 256)), dim3(256), 0, stream, descs, (int)m,
                     s->total_blocks, arena);
  hipLaunchKernelGGL(jpeg_colour_kernel, dim3((unsigned)((s->max_pixels + 255) / 256), (unsigned)m), dim3(256), 0, stream, descs, arena,
                     (uint8_t*)rgb_dev);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  // the entropy kernel's verdicts (behind this copy the staging buffer is free for the next batch)
  if (hipError_t e = hipMemcpyAsync(st, arena, m * sizeof(int), hipMemcpyDeviceToHost, stream); e != hipSuccess) return e;
  if (hipError_t e = hipStreamSynchronize(stream); e != hipSuccess) return e;
  for (size_t i = 0; i < m; ++i) {
    const int ds = ((const int*)st)[i];
    status[s->input_index[i]] = ds ? 100 + ds : 0;
  }
  return hipSuccess;
}
