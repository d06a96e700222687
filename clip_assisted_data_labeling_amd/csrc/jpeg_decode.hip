// Baseline JPEG -> RGB on the GPU: the input side of the embed path.  The reference decodes every file in DataLoader workers
// (/root/reference/utils/embedder.py:167, PIL.Image.open(...).convert('RGB')) and is bound by that on real data; here the host
// only walks the markers (jpeg_host.cpp) and the device does the rest with the arithmetic of jpeg_core.h, which is Pillow's
// (libjpeg-turbo defaults) bit for bit:
//   jpeg_entropy_kernel   one workgroup per image, PARALLEL inside the image: the unstuffed scan is cut into subsequences of 2 KiB,
//                         one thread each, decoded speculatively and re-decoded until every subsequence's entry state is the
//                         exit state of the one before (jpeg_core.h, "parallel entropy decoding": Huffman streams
//                         re-synchronise, typically 2-3 passes); then a prefix sum of block counts and DC differences and one
//                         writing pass.  Worst case (periodic content) it degenerates to the serial walk, never to a wrong result
//   jpeg_idct_kernel      one thread per 8 x 8 block of any image: dequantise, integer inverse DCT, samples into the plane
//   jpeg_colour_kernel    one thread per output pixel: chroma upsampling (triangle filters) + YCbCr -> RGB, interleaved uint8
// A batch is planned on the host in one pass (per image: the entropy segment with its byte stuffing and restart markers
// removed -- host threads, memchr-paced --, restart-interval table, coefficient planes, sample planes) into one device arena that
// grows to the largest batch seen; descriptors and segments travel in ONE host-to-device copy from page-locked memory.  The RGB output goes to memory the caller owns.
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "common.h"
#include "jpeg_host.h"
#include "kernels.h"

using jpg::ImageDesc;

namespace {

constexpr int SUB_BYTES = 2048;              // subsequence length: 2-3 synchronisation passes on photographic and on noise content
constexpr int ENT_THREADS = 256;

// per-subsequence scratch (global memory, jpg::ImageDesc::sub_off)
struct SubScratch {
  uint64_t entry;                            // state the subsequence is decoded from
  uint64_t exit;                             // state its last decode ended in
  uint32_t blocks, du_base;                  // blocks completed in it / index of the block in progress at its entry
  int32_t dc[3], dc_base[3];                 // DC differences met in it / DC predictors at its entry
  uint32_t dirty, pad;                       // entry changed since the last decode
};

__global__ __launch_bounds__(ENT_THREADS) void jpeg_entropy_kernel(const ImageDesc* __restrict__ descs, int* __restrict__ status,
                                                                   uint8_t* __restrict__ arena) {
  __shared__ jpg::HuffTable tabs[4];
  __shared__ uint8_t zz[64];
  __shared__ int s_err;
  __shared__ jpg::ParGeom g;                                   // (its small tables are indexed by the decoding state: LDS, not registers)
  const ImageDesc& d = descs[blockIdx.x];
  const int tid = threadIdx.x;
  {
    if (tid == 0) g = jpg::par_geom(d);
    const uint32_t* src = (const uint32_t*)d.huff;
    uint32_t* dst = (uint32_t*)tabs;
    for (int i = tid; i < (int)(sizeof(tabs) / 4); i += ENT_THREADS) dst[i] = src[i];
    if (tid < 64) zz[tid] = (uint8_t)jpg::zigzag_to_natural(tid);
    if (tid == 0) s_err = 0;
  }
  __syncthreads();
  if (d.prog_off) return;                                      // a progressive file: jpeg_entropy_prog_kernel
  const uint32_t N = d.n_sub;
  if (N == 0) {                                                // the host could not unstuff the scan (no EOI, stray marker)
    if (tid == 0) status[blockIdx.x] = 3;
    return;
  }
  const uint8_t* clean = arena + d.clean_off;
  const uint32_t* iv_byte = (const uint32_t*)(arena + d.iv_off);
  const uint32_t* iv_sub = iv_byte + (d.n_iv + 1);
  const int n_iv = d.n_iv, L = d.sub_bytes;
  SubScratch* sc = (SubScratch*)(arena + d.sub_off);
  int16_t* coef[jpg::MAX_COMPS];
  for (int c = 0; c < jpg::MAX_COMPS; ++c) coef[c] = (int16_t*)(arena + d.coef_off[c]);

  for (uint32_t i = tid; i < N; i += ENT_THREADS) {
    const jpg::SubSeq q = jpg::subseq_of(iv_byte, iv_sub, n_iv, L, i);
    sc[i].entry = jpg::pack_state(jpg::SubState{q.start_bit, 0, 0});
    sc[i].dirty = 1;
  }
  __syncthreads();
  // ---- synchronisation passes: at most N (after pass t the entries 0 .. t are the true ones); typically 2-3.  Capped: files are
  // untrusted, and one made not to re-synchronise would walk its up to 32 768 subsequences one pass each (the serial walk's time,
  // tens of seconds on one workgroup while the host waits).  Content that needs more than MAX_SYNC_PASSES passes -- megabytes of
  // flat scan data, i.e. hundreds of megapixels -- is flagged like corrupt entropy data and goes to the caller's Pillow path.
  constexpr uint32_t MAX_SYNC_PASSES = 1024;
  bool converged = false;
  for (uint32_t pass = 0; pass <= N && pass < MAX_SYNC_PASSES; ++pass) {
    for (uint32_t i = tid; i < N; i += ENT_THREADS) {
      if (!sc[i].dirty) continue;
      sc[i].dirty = 0;
      const jpg::SubSeq q = jpg::subseq_of(iv_byte, iv_sub, n_iv, L, i);
      if (q.last) continue;                                      // nobody reads its exit; its block count follows from the interval's
      jpg::SubState st = jpg::unpack_state(sc[i].entry);
      int acc[3] = {0, 0, 0};
      uint32_t done = 0;
      jpg::decode_span<false>(g, clean, tabs, zz, st, q.end_bit, 0, 0, acc, coef, &done);
      sc[i].exit = jpg::pack_state(st);
      sc[i].blocks = done;
      sc[i].dc[0] = acc[0]; sc[i].dc[1] = acc[1]; sc[i].dc[2] = acc[2];
    }
    __syncthreads();
    int changed = 0;
    for (uint32_t i = tid; i + 1 < N; i += ENT_THREADS) {
      const jpg::SubSeq q = jpg::subseq_of(iv_byte, iv_sub, n_iv, L, i);
      if (q.last) continue;                                      // the next subsequence starts an interval: its entry is known
      if (sc[i + 1].entry != sc[i].exit) { sc[i + 1].entry = sc[i].exit; sc[i + 1].dirty = 1; changed = 1; }
    }
    if (!__syncthreads_or(changed)) { converged = true; break; }
  }
  if (!converged) {                                               // (uniform: every thread saw the same votes)
    if (tid == 0) status[blockIdx.x] = 4;
    return;
  }
  // ---- block index and DC predictors at every entry: prefix sums inside each restart interval
  for (int j = tid; j < n_iv; j += ENT_THREADS) {
    uint32_t du = g.du_per_interval ? (uint32_t)j * g.du_per_interval : 0u;
    int b0 = 0, b1 = 0, b2 = 0;
    for (uint32_t i = iv_sub[j]; i < iv_sub[j + 1]; ++i) {
      sc[i].du_base = du;
      sc[i].dc_base[0] = b0; sc[i].dc_base[1] = b1; sc[i].dc_base[2] = b2;
      du += sc[i].blocks; b0 += sc[i].dc[0]; b1 += sc[i].dc[1]; b2 += sc[i].dc[2];
    }
  }
  __syncthreads();
  // ---- the writing pass
  int err = 0;
  for (uint32_t i = tid; i < N; i += ENT_THREADS) {
    const jpg::SubSeq q = jpg::subseq_of(iv_byte, iv_sub, n_iv, L, i);
    const uint32_t iv_first = g.du_per_interval ? (uint32_t)q.interval * g.du_per_interval : 0u;
    const uint32_t iv_stop = g.du_per_interval ? min(iv_first + g.du_per_interval, g.total_du) : g.total_du;
    jpg::SubState st = jpg::unpack_state(sc[i].entry);
    int pred[3] = {sc[i].dc_base[0], sc[i].dc_base[1], sc[i].dc_base[2]};
    uint32_t done = 0;
    const uint32_t du0 = sc[i].du_base;
    if (du0 > iv_stop || iv_first > g.total_du) { err = 1; continue; }
    if (jpg::decode_span<true>(g, clean, tabs, zz, st, q.end_bit, du0, iv_stop, pred, coef, &done)) { err = 1; continue; }
    if (q.last) {
      if (du0 + done != iv_stop || st.u != 0 || st.k != 0) err = 1;     // the interval does not hold the blocks it must
      else if (st.bit > q.iv_end_bit) err = 2;                   // bits consumed that are not in the file
    } else if (jpg::pack_state(st) != sc[i + 1].entry) {
      err = 1;                                                   // stopped early at the interval's block limit
    }
  }
  if (err) atomicMax(&s_err, err);
  __syncthreads();
  if (tid == 0) status[blockIdx.x] = s_err;
}

// Progressive files: one workgroup per image, lane 0 walks the scans one after the other (jpeg_core.h: end-of-band runs and
// correction bits make a scan serial); the tables of the current scan are in LDS.  The parallelism is across the images of the batch.
__global__ __launch_bounds__(64) void jpeg_entropy_prog_kernel(const ImageDesc* __restrict__ descs, int* __restrict__ status,
                                                               uint8_t* __restrict__ arena) {
  __shared__ jpg::HuffTable tabs[7];                           // [0..2] DC tables of the scan's components, [3] its AC table (progressive),
  __shared__ uint8_t zz[64];                                   // [4..6] AC tables per component (full-band scans of a sequential file)
  const ImageDesc& d = descs[blockIdx.x];
  if (!d.prog_off) return;
  const jpg::ProgDesc& pd = *(const jpg::ProgDesc*)(arena + d.prog_off);
  const jpg::HuffTable* all = (const jpg::HuffTable*)(arena + pd.tabs_off);
  const int tid = threadIdx.x;
  zz[tid] = (uint8_t)jpg::zigzag_to_natural(tid);
  int16_t* coef[jpg::MAX_COMPS];
  for (int c = 0; c < jpg::MAX_COMPS; ++c) coef[c] = (int16_t*)(arena + d.coef_off[c]);
  int st = 0;
  for (int si = 0; si < pd.n_scans; ++si) {
    const jpg::ProgScan& ps = pd.scans[si];
    __syncthreads();                                             // lane 0 is done with the previous scan's tables
    const bool full_band = ps.ss == 0 && ps.se == 63;            // a scan of a sequential file: DC and AC table per component
    for (int slot = 0; slot < 7; ++slot) {
      int src_i = -1;
      if (slot < 3) { if (slot < ps.ncomp && ps.ss == 0 && ps.ah == 0) src_i = ps.dc_tab[slot]; }
      else if (slot == 3) { if (ps.ss > 0) src_i = ps.ac_tab; }
      else if (full_band && slot - 4 < ps.ncomp) src_i = ps.ac_tab3[slot - 4];
      if (src_i < 0) continue;
      const uint32_t* src = (const uint32_t*)(all + src_i);
      uint32_t* dst = (uint32_t*)(tabs + slot);
      for (int i = tid; i < (int)(sizeof(jpg::HuffTable) / 4); i += 64) dst[i] = src[i];
    }
    __syncthreads();
    if (tid == 0 && st == 0) {
      if (ps.n_iv <= 0) st = 3;                                  // the host could not prepare this scan
      else {
        jpg::ProgScan local = ps;
        local.dc_tab[0] = 0; local.dc_tab[1] = 1; local.dc_tab[2] = 2; local.ac_tab = 3;
        local.ac_tab3[0] = 4; local.ac_tab3[1] = 5; local.ac_tab3[2] = 6;
        st = jpg::prog_decode_scan(d, local, arena + ps.clean_off, (const uint32_t*)(arena + ps.iv_off), tabs, zz, coef);
      }
    }
  }
  if (tid == 0) status[blockIdx.x] = st;
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
  if (d.ncomp == 1) { o[0] = o[1] = o[2] = arena[d.plane_off[0] + (size_t)y * d.bw[0] * 8 + x]; return; }
  const int Y = jpg::upsampled(arena + d.plane_off[0], d.bw[0] * 8, d.dw[0], d.dh[0], d.hmax / d.hs[0], d.vmax / d.vs[0], x, y);
  const int cb = jpg::upsampled(arena + d.plane_off[1], d.bw[1] * 8, d.dw[1], d.dh[1], d.hmax / d.hs[1], d.vmax / d.vs[1], x, y);
  const int cr = jpg::upsampled(arena + d.plane_off[2], d.bw[2] * 8, d.dw[2], d.dh[2], d.hmax / d.hs[2], d.vmax / d.vs[2], x, y);
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
  std::vector<uint32_t> max_iv;            // per image: upper bound of its restart intervals
  std::vector<jpg::ProgInfo> prog;         // per image: its scans, if it is a progressive file (else empty)
  bool any_prog = false;
  std::vector<size_t> prog_tabs_off;       // per image: arena offset of its table array (progressive)
  std::vector<int> host_status;            // per image: 0, or why the host could not prepare its scan
  size_t stage_bytes = 0, arena_bytes = 0, desc_begin = 0, coef_begin = 0, coef_bytes = 0, max_pixels = 0;
  uint64_t total_blocks = 0, rgb_bytes = 0;
  // buffers (grow only)
  void* arena = nullptr; size_t arena_cap = 0;
  void* stage = nullptr; size_t stage_cap = 0;          // [descs][entropy segments]; page-locked while small (ce_jpegdec_reserve)
  bool stage_pinned = false;
};

JpegDecState* ce_jpegdec_create() { return new JpegDecState(); }

#ifndef JPEG_STAGE_PINNED_MAX
#define JPEG_STAGE_PINNED_MAX (64u << 20)
#endif
static constexpr size_t STAGE_PINNED_MAX = JPEG_STAGE_PINNED_MAX;
static void free_stage(JpegDecState* s) {
  if (s->stage) { if (s->stage_pinned) (void)hipHostFree(s->stage); else free(s->stage); }
  s->stage = nullptr; s->stage_cap = 0; s->stage_pinned = false;
}

void ce_jpegdec_destroy(JpegDecState* s) {
  if (!s) return;
  if (s->arena) (void)hipFree(s->arena);
  free_stage(s);
  delete s;
}

// Host pass over n files: status[i] = 0 (will be decoded) or the reason it cannot be (jpeg_host.h); widths / heights for every
// file whose header could be read; rgb_offsets[i] = offset of image i's uint8 [H][W][3] in the caller's output buffer of
// *rgb_bytes bytes (256-byte aligned images).
void ce_jpegdec_plan(JpegDecState* s, const void* const* files, const size_t* sizes, int n, int* status, int* widths, int* heights,
                     unsigned long long* rgb_offsets, unsigned long long* rgb_bytes) {
  s->descs.clear(); s->input_index.clear(); s->scans.clear(); s->prog.clear();
  s->any_prog = false;
  s->descs.reserve((size_t)n);
  size_t rgb = 0;
  uint64_t blocks = 0;
  s->max_pixels = 0;
  for (int i = 0; i < n; ++i) {
    ImageDesc d;
    size_t so = 0, sl = 0;
    jpg::ProgInfo pi;
    const int rc = files[i] ? jpg::parse_jpeg((const uint8_t*)files[i], sizes[i], &d, &so, &sl, &pi) : jpg::JPG_NOT_JPEG;
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
    s->scans.emplace_back((const uint8_t*)files[i] + so, sl);   // (progressive: the file start, its scans carry their own ranges)
    if (!pi.scans.empty()) s->any_prog = true;
    s->prog.push_back(std::move(pi));
  }
  // arena: [status words][descs][per image: unstuffed scan + restart-interval table][subsequence scratch][coefficients]
  // [sample planes]; the first three are what the staging buffer holds.  Sizes here are upper bounds (the scan is unstuffed in
  // ce_jpegdec_run): clean bytes <= scan bytes, intervals <= MCUs / restart interval + 1, subsequences <= bytes / SUB_BYTES + intervals
  const size_t m = s->descs.size();
  s->desc_begin = up(m * sizeof(int), 256);
  size_t off = s->desc_begin + up(m * sizeof(ImageDesc), 256);
  s->max_iv.assign(m, 1);
  s->prog_tabs_off.clear();
  for (size_t i = 0; i < m; ++i) {
    ImageDesc& d = s->descs[i];
    const size_t scan_len = s->scans[i].second;
    const size_t mcus = (size_t)d.mcus_x * d.mcus_y;
    const size_t max_iv = d.restart_interval ? (mcus + d.restart_interval - 1) / d.restart_interval : 1;
    s->max_iv[i] = (uint32_t)max_iv;
    d.sub_bytes = SUB_BYTES;
    d.data_off = 0; d.data_len = 0; d.data_real = 0;             // (the stuffed segment does not travel to the device)
    d.prog_off = 0;
    jpg::ProgInfo& pi = s->prog[i];
    if (!pi.scans.empty()) {
      // progressive: [ProgDesc][tables][per scan: unstuffed data, interval starts]
      d.prog_off = off;
      off += up(sizeof(jpg::ProgDesc), 16);
      const size_t tabs_off = off;
      off += up(pi.tabs.size() * sizeof(jpg::HuffTable), 16);
      for (auto& si : pi.scans) {
        const bool single = si.s.ncomp == 1;
        const size_t smcus = single ? (size_t)((d.dw[si.s.comp[0]] + 7) / 8) * ((d.dh[si.s.comp[0]] + 7) / 8) : mcus;
        const size_t siv = si.s.restart_interval ? (smcus + si.s.restart_interval - 1) / si.s.restart_interval : 1;
        si.s.clean_off = off;
        off += up(si.end - si.begin + 32, 16);
        si.s.iv_off = off;
        off += up((siv + 1) * sizeof(uint32_t), 16);
        si.s.n_iv = (int32_t)siv;                                // (expected count; verified when the scan is unstuffed)
      }
      s->prog_tabs_off.push_back(tabs_off);
      d.clean_off = 0; d.iv_off = 0;
      continue;
    }
    s->prog_tabs_off.push_back(0);
    d.clean_off = off;
    off += up(scan_len + 16, 16);
    d.iv_off = off;
    off += up((max_iv + 1) * 2 * sizeof(uint32_t), 16);
  }
  off = up(off, 256);
  s->stage_bytes = off;
  for (size_t i = 0; i < m; ++i) {
    ImageDesc& d = s->descs[i];
    const size_t max_sub = d.prog_off ? 0 : s->scans[i].second / SUB_BYTES + s->max_iv[i] + 1;
    d.sub_off = off;
    off += up(max_sub * sizeof(SubScratch), 16);
  }
  off = up(off, 256);
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

// Device scratch and page-locked staging of at least these sizes (never shrinks; 0 = leave as is).  On failure the state keeps
// no buffer of that kind (capacity 0) and the error is returned.
hipError_t ce_jpegdec_reserve(JpegDecState* s, size_t arena_bytes, size_t stage_bytes) {
  if (arena_bytes > s->arena_cap) {
    if (s->arena) (void)hipFree(s->arena);
    s->arena = nullptr; s->arena_cap = 0;
    if (hipError_t e = hipMalloc(&s->arena, arena_bytes); e != hipSuccess) { (void)hipGetLastError(); return hipErrorOutOfMemory; }
    s->arena_cap = arena_bytes;
  }
  if (stage_bytes > s->stage_cap) {
    // Staging up to STAGE_PINNED_MAX is page-locked (the copy is then one asynchronous DMA); beyond that it is ordinary memory:
    // page-locking costs ~0.35 s per GB, and a driver that doubles its chunks paid it again at every growth, on the path its
    // encoder waits for (rocprofv3: GPU idle gaps of 53 / 108 / 249 ms in front of the 512 / 1 024 / 2 048-file chunks); the
    // runtime's staged copy of a pageable buffer moves the same bytes at ~10 GB/s with no set-up cost.
    free_stage(s);
    if (stage_bytes <= STAGE_PINNED_MAX) {
      if (hipError_t e = hipHostMalloc(&s->stage, stage_bytes, hipHostMallocDefault); e != hipSuccess) { (void)hipGetLastError(); s->stage = nullptr; return hipErrorOutOfMemory; }
      s->stage_pinned = true;
    } else {
      s->stage = aligned_alloc(4096, (stage_bytes + 4095) / 4096 * 4096);
      if (!s->stage) return hipErrorOutOfMemory;
      s->stage_pinned = false;
    }
    s->stage_cap = stage_bytes;
  }
  return hipSuccess;
}

// Decodes the planned batch into rgb_dev (>= the plan's rgb_bytes); synchronises the stream; dev_status[i] for every INPUT file:
// unchanged for the ones the plan refused, 0 or 100 + code for the decoded ones (entropy data ran out / invalid code).
hipError_t ce_jpegdec_run(JpegDecState* s, void* rgb_dev, int* status, hipStream_t stream) {
  const size_t m = s->descs.size();
  if (m == 0) return hipSuccess;
  // (growth only when a batch exceeds what ce_jpegdec_reserve set aside: hipFree / hipMalloc synchronise the whole device, i.e.
  //  stall an encoder running on another stream -- the embed driver reserves once, before the encoder starts)
  if (hipError_t e = ce_jpegdec_reserve(s, s->arena_bytes > s->arena_cap ? s->arena_bytes + s->arena_bytes / 2 : 0,
                                        s->stage_bytes > s->stage_cap ? s->stage_bytes + s->stage_bytes / 2 : 0); e != hipSuccess) {
    // not even the exact size?  (the caller hands the batch to its own decoder on hipErrorOutOfMemory)
    if (hipError_t e2 = ce_jpegdec_reserve(s, s->arena_bytes, s->stage_bytes); e2 != hipSuccess) return e2;
  }
  uint8_t* st = (uint8_t*)s->stage;
  memset(st, 0xff, m * sizeof(int));                          // status words: -1 until the entropy kernel has written them
  // unstuff every scan into the staging buffer (host threads: memchr-paced, ~1 GB/s each) and fill in what depends on it
  s->host_status.assign(m, 0);
  {
    const int nt = (int)std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), std::min<size_t>(16, m));
    auto work = [&](int t) {
      for (size_t i = (size_t)t; i < m; i += (size_t)nt) {
        ImageDesc& d = s->descs[i];
        if (d.prog_off) {
          jpg::ProgInfo& pi = s->prog[i];
          jpg::ProgDesc* pd = (jpg::ProgDesc*)(st + d.prog_off);
          memset(pd, 0, sizeof *pd);
          pd->n_scans = (int32_t)pi.scans.size(); pd->n_tabs = (int32_t)pi.tabs.size(); pd->tabs_off = s->prog_tabs_off[i];
          memcpy(st + pd->tabs_off, pi.tabs.data(), pi.tabs.size() * sizeof(jpg::HuffTable));
          d.n_sub = 0; d.n_iv = 0; d.clean_len = 0;
          for (size_t k = 0; k < pi.scans.size(); ++k) {
            jpg::ProgScan ps = pi.scans[k].s;
            uint32_t* ivb = (uint32_t*)(st + ps.iv_off);
            int got_iv = 0;
            const long cl = jpg::unstuff_scan(s->scans[i].first + pi.scans[k].begin, pi.scans[k].end - pi.scans[k].begin, st + ps.clean_off, ivb,
                                              ps.n_iv, &got_iv, false);
            if (cl < 0 || got_iv != ps.n_iv) { ps.n_iv = 0; ps.clean_len = 0; s->host_status[i] = cl < 0 ? 3 : 1; }
            else { ivb[got_iv] = (uint32_t)cl; ps.clean_len = (uint32_t)cl; memset(st + ps.clean_off + cl, 0xFF, 32); }
            pd->scans[k] = ps;
          }
          continue;
        }
        uint32_t* iv_byte = (uint32_t*)(st + d.iv_off);
        int n_iv = 0;
        const long clen = jpg::unstuff_scan(s->scans[i].first, s->scans[i].second, st + d.clean_off, iv_byte, (int)s->max_iv[i], &n_iv);
        const uint32_t mcus = (uint32_t)d.mcus_x * (uint32_t)d.mcus_y;
        const uint32_t want_iv = d.restart_interval ? (mcus + d.restart_interval - 1) / d.restart_interval : 1u;
        d.n_sub = 0; d.n_iv = 0; d.clean_len = 0;
        if (clen < 0) { s->host_status[i] = 3; continue; }       // no EOI / a marker that does not belong into a scan
        if ((uint32_t)n_iv != want_iv) { s->host_status[i] = 1; continue; }
        iv_byte[n_iv] = (uint32_t)clen;
        uint32_t* iv_sub = iv_byte + (n_iv + 1);
        uint32_t nsub = 0;
        for (int j = 0; j < n_iv; ++j) {
          iv_sub[j] = nsub;
          const uint32_t bytes = iv_byte[j + 1] - iv_byte[j];
          nsub += bytes ? (bytes + SUB_BYTES - 1) / SUB_BYTES : 1u;
        }
        iv_sub[n_iv] = nsub;
        d.n_iv = n_iv; d.n_sub = nsub; d.clean_len = (uint32_t)clen;
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
  }
  memcpy(st + s->desc_begin, s->descs.data(), m * sizeof(ImageDesc));
  uint8_t* arena = (uint8_t*)s->arena;
  if (hipError_t e = hipMemcpyAsync(arena, st, s->stage_bytes, hipMemcpyHostToDevice, stream); e != hipSuccess) return e;
  if (hipError_t e = hipMemsetAsync(arena + s->coef_begin, 0, s->coef_bytes, stream); e != hipSuccess) return e;
  const ImageDesc* descs = (const ImageDesc*)(arena + s->desc_begin);
  hipLaunchKernelGGL(jpeg_entropy_kernel, dim3((unsigned)m), dim3(ENT_THREADS), 0, stream, descs, (int*)arena, arena);
  if (s->any_prog) hipLaunchKernelGGL(jpeg_entropy_prog_kernel, dim3((unsigned)m), dim3(64), 0, stream, descs, (int*)arena, arena);
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((s->total_blocks + 255) / 256)), dim3(256), 0, stream, descs, (int)m,
                     s->total_blocks, arena);
  hipLaunchKernelGGL(jpeg_colour_kernel, dim3((unsigned)((s->max_pixels + 255) / 256), (unsigned)m), dim3(256), 0, stream, descs, arena,
                     (uint8_t*)rgb_dev);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  // the entropy kernel's verdicts (behind this copy the staging buffer is free for the next batch)
  if (hipError_t e = hipMemcpyAsync(st, arena, m * sizeof(int), hipMemcpyDeviceToHost, stream); e != hipSuccess) return e;
  if (hipError_t e = hipStreamSynchronize(stream); e != hipSuccess) return e;
  for (size_t i = 0; i < m; ++i) {
    const int ds = s->host_status[i] ? s->host_status[i] : ((const int*)st)[i];
    status[s->input_index[i]] = ds ? 100 + ds : 0;
  }
  return hipSuccess;
}
