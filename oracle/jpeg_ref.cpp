// TEST INFRASTRUCTURE ONLY.  CPU run of the JPEG decoding arithmetic (clip_assisted_data_labeling_amd/csrc/jpeg_core.h +
// jpeg_host.cpp, the same sources the HIP kernels are compiled from) so that tests can compare it with Pillow -- the reference's
// own decoder, /root/reference/utils/embedder.py:167 -- without a GPU.  Never linked into, loaded or called by the product library.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../clip_assisted_data_labeling_amd/csrc/jpeg_host.h"

extern "C" {

// returns a jpg:: reason code (0 = decodable); width / height are filled whenever the header could be read
int jpeg_ref_info(const uint8_t* data, size_t len, int* width, int* height, int* ncomp) {
  static jpg::ImageDesc d;
  size_t so = 0, sl = 0;
  jpg::ProgInfo prog;
  const int rc = jpg::parse_jpeg(data, len, &d, &so, &sl, &prog);
  *width = d.width; *height = d.height; *ncomp = d.ncomp;
  return rc;
}

namespace {

void finish_image(const jpg::ImageDesc& d, int16_t* const cp[jpg::MAX_COMPS], uint8_t* rgb) {
  std::vector<std::vector<uint8_t>> plane(d.ncomp);
  for (int c = 0; c < d.ncomp; ++c) {
    plane[c].assign((size_t)d.bw[c] * d.bh[c] * 64, 0);
    const int pitch = d.bw[c] * 8;
    for (int by = 0; by < d.bh[c]; ++by)
      for (int bx = 0; bx < d.bw[c]; ++bx)
        jpg::idct_block(cp[c] + ((size_t)by * d.bw[c] + bx) * 64, d.quant[c], plane[c].data() + (size_t)by * 8 * pitch + bx * 8, pitch);
  }
  for (int y = 0; y < d.height; ++y)
    for (int x = 0; x < d.width; ++x) {
      uint8_t* o = rgb + ((size_t)y * d.width + x) * 3;
      if (d.ncomp == 1) { o[0] = o[1] = o[2] = plane[0][(size_t)y * d.bw[0] * 8 + x]; continue; }
      const int Y = jpg::upsampled(plane[0].data(), d.bw[0] * 8, d.dw[0], d.dh[0], d.hmax / d.hs[0], d.vmax / d.vs[0], x, y);
      const int cb = jpg::upsampled(plane[1].data(), d.bw[1] * 8, d.dw[1], d.dh[1], d.hmax / d.hs[1], d.vmax / d.vs[1], x, y);
      const int cr = jpg::upsampled(plane[2].data(), d.bw[2] * 8, d.dw[2], d.dh[2], d.hmax / d.hs[2], d.vmax / d.vs[2], x, y);
      jpg::ycc_to_rgb(Y, cb, cr, o);
    }
}

}  // namespace

namespace {

// a progressive file: its scans one after the other (jpeg_core.h: prog_decode_scan), exactly what the device's lane does
int decode_progressive(const uint8_t* data, const jpg::ImageDesc& d, jpg::ProgInfo& prog, uint8_t* rgb) {
  std::vector<std::vector<int16_t>> coef(d.ncomp);
  int16_t* cp[jpg::MAX_COMPS] = {nullptr, nullptr, nullptr};
  for (int c = 0; c < d.ncomp; ++c) { coef[c].assign((size_t)d.bw[c] * d.bh[c] * 64, 0); cp[c] = coef[c].data(); }
  uint8_t zz[64];
  for (int k = 0; k < 64; ++k) zz[k] = (uint8_t)jpg::zigzag_to_natural(k);
  for (auto& si : prog.scans) {
    const size_t sl = si.end - si.begin;
    std::vector<uint32_t> clean_words((sl + 32) / 4 + 8);
    uint8_t* clean = (uint8_t*)clean_words.data();
    std::vector<uint32_t> iv_byte(sl / 2 + 4);
    int n_iv = 0;
    const long clen = jpg::unstuff_scan(data + si.begin, sl, clean, iv_byte.data(), (int)iv_byte.size() - 1, &n_iv, false);
    if (clen < 0) return 103;
    memset(clean + clen, 0xFF, 32);
    jpg::ProgScan& ps = si.s;
    const bool single = ps.ncomp == 1;
    const uint32_t mcus = single ? (uint32_t)((d.dw[ps.comp[0]] + 7) / 8) * (uint32_t)((d.dh[ps.comp[0]] + 7) / 8)
                                 : (uint32_t)d.mcus_x * (uint32_t)d.mcus_y;
    const uint32_t want_iv = ps.restart_interval ? (mcus + ps.restart_interval - 1) / ps.restart_interval : 1;
    if ((uint32_t)n_iv != want_iv) return 105;
    iv_byte[n_iv] = (uint32_t)clen;
    ps.n_iv = n_iv; ps.clean_len = (uint32_t)clen;
    const int st = jpg::prog_decode_scan(d, ps, clean, iv_byte.data(), prog.tabs.data(), zz, cp);
    if (st) return 100 + st;
  }
  finish_image(d, cp, rgb);
  return 0;
}

}  // namespace

// rgb: [height][width][3]; returns 0, a parse reason code, or 100 + the entropy decoder's status.  The SERIAL walk of the scan.
int jpeg_ref_decode(const uint8_t* data, size_t len, uint8_t* rgb) {
  static jpg::ImageDesc d;                                     // (6 KiB of tables)
  size_t so = 0, sl = 0;
  jpg::ProgInfo prog;
  const int rc = jpg::parse_jpeg(data, len, &d, &so, &sl, &prog);
  if (rc) return rc;
  if (!prog.scans.empty()) return decode_progressive(data, d, prog, rgb);
  // the entropy-coded segment, 16-byte aligned and padded the way the product's host side pads it
  const size_t padded = (sl + 15) / 16 * 16 + 32;
  std::vector<uint64_t> seg(padded / 8 + 1);
  memset(seg.data(), 0xFF, padded);
  memcpy(seg.data(), data + so, sl);
  for (size_t i = sl; i + 1 < padded; i += 2) { ((uint8_t*)seg.data())[i] = 0xFF; ((uint8_t*)seg.data())[i + 1] = 0xD9; }
  d.data_len = (uint32_t)padded;
  d.data_real = (uint32_t)sl;
  std::vector<std::vector<int16_t>> coef(d.ncomp);
  int16_t* cp[jpg::MAX_COMPS] = {nullptr, nullptr, nullptr};
  for (int c = 0; c < d.ncomp; ++c) { coef[c].assign((size_t)d.bw[c] * d.bh[c] * 64, 0); cp[c] = coef[c].data(); }
  uint8_t zz[64];
  for (int k = 0; k < 64; ++k) zz[k] = (uint8_t)jpg::zigzag_to_natural(k);
  const int st = jpg::decode_scan(jpg::scan_geom(d), (const uint8_t*)seg.data(), cp, d.huff, zz);
  if (st) return 100 + st;
  finish_image(d, cp, rgb);
  return 0;
}

// The PARALLEL decode of the scan (jpeg_core.h: subsequences, entry / exit states passed on until nothing changes), its
// "threads" run one after the other here.  *passes = synchronisation passes it took.  Same return codes; 104 = not converged
// in max_passes, 105 = the restart intervals do not match the frame.
int jpeg_ref_decode_parallel(const uint8_t* data, size_t len, uint8_t* rgb, int sub_bytes, int max_passes, int* passes) {
  static jpg::ImageDesc d;
  size_t so = 0, sl = 0;
  jpg::ProgInfo prog;
  const int rc = jpg::parse_jpeg(data, len, &d, &so, &sl, &prog);
  if (rc) return rc;
  if (!prog.scans.empty()) { if (passes) *passes = 0; return decode_progressive(data, d, prog, rgb); }
  std::vector<uint32_t> clean_words((sl + 16) / 4 + 8);
  uint8_t* clean = (uint8_t*)clean_words.data();
  std::vector<uint32_t> iv_byte(sl / 2 + 4);
  int n_iv = 0;
  const long clen = jpg::unstuff_scan(data + so, sl, clean, iv_byte.data(), (int)iv_byte.size() - 1, &n_iv);
  if (clen < 0) return 103;
  const jpg::ParGeom g = jpg::par_geom(d);
  const uint32_t total_mcus = (uint32_t)d.mcus_x * d.mcus_y;
  const uint32_t want_iv = d.restart_interval ? (total_mcus + d.restart_interval - 1) / d.restart_interval : 1;
  if ((uint32_t)n_iv != want_iv) return 105;
  iv_byte[n_iv] = (uint32_t)clen;
  std::vector<uint32_t> iv_sub(n_iv + 1);
  uint32_t N = 0;
  for (int j = 0; j < n_iv; ++j) { iv_sub[j] = N; const uint32_t bytes = iv_byte[j + 1] - iv_byte[j]; N += bytes ? (bytes + sub_bytes - 1) / sub_bytes : 1; }
  iv_sub[n_iv] = N;
  std::vector<std::vector<int16_t>> coef(d.ncomp);
  int16_t* cp[jpg::MAX_COMPS] = {nullptr, nullptr, nullptr};
  for (int c = 0; c < d.ncomp; ++c) { coef[c].assign((size_t)d.bw[c] * d.bh[c] * 64, 0); cp[c] = coef[c].data(); }
  uint8_t zz[64];
  for (int k = 0; k < 64; ++k) zz[k] = (uint8_t)jpg::zigzag_to_natural(k);
  std::vector<uint64_t> entry(N), exitst(N);
  std::vector<uint32_t> cnt(N, 0), du_base(N, 0);
  std::vector<int> dcs(3 * (size_t)N, 0), dcb(3 * (size_t)N, 0);
  for (uint32_t i = 0; i < N; ++i) {
    const jpg::SubSeq q = jpg::subseq_of(iv_byte.data(), iv_sub.data(), n_iv, sub_bytes, i);
    entry[i] = jpg::pack_state(jpg::SubState{q.start_bit, 0, 0});
  }
  bool converged = false;
  int it = 0;
  for (; it < max_passes && !converged; ++it) {
    for (uint32_t i = 0; i < N; ++i) {                          // "all threads": read entry[], write exitst[]
      const jpg::SubSeq q = jpg::subseq_of(iv_byte.data(), iv_sub.data(), n_iv, sub_bytes, i);
      if (q.last) continue;
      jpg::SubState s = jpg::unpack_state(entry[i]);
      int acc[3] = {0, 0, 0};
      uint32_t done = 0;
      jpg::decode_span<false>(g, clean, d.huff, zz, s, q.end_bit, 0, 0, acc, cp, &done);
      exitst[i] = jpg::pack_state(s); cnt[i] = done;
      for (int c = 0; c < 3; ++c) dcs[3 * (size_t)i + c] = acc[c];
    }
    converged = true;
    for (uint32_t i = 0; i + 1 < N; ++i) {                      // barrier, then hand the exit states on
      const jpg::SubSeq q = jpg::subseq_of(iv_byte.data(), iv_sub.data(), n_iv, sub_bytes, i);
      if (q.last) continue;
      if (entry[i + 1] != exitst[i]) { entry[i + 1] = exitst[i]; converged = false; }
    }
  }
  if (passes) *passes = it;
  if (!converged) return 104;
  for (int j = 0; j < n_iv; ++j) {                              // prefix sums inside every interval
    uint32_t du = g.du_per_interval ? (uint32_t)j * g.du_per_interval : 0;
    int b[3] = {0, 0, 0};
    for (uint32_t i = iv_sub[j]; i < iv_sub[j + 1]; ++i) {
      du_base[i] = du;
      for (int c = 0; c < 3; ++c) { dcb[3 * (size_t)i + c] = b[c]; b[c] += dcs[3 * (size_t)i + c]; }
      du += cnt[i];
    }
  }
  for (uint32_t i = 0; i < N; ++i) {                            // the writing pass
    const jpg::SubSeq q = jpg::subseq_of(iv_byte.data(), iv_sub.data(), n_iv, sub_bytes, i);
    const uint32_t iv_first = g.du_per_interval ? (uint32_t)q.interval * g.du_per_interval : 0;
    const uint32_t iv_stop = g.du_per_interval ? std::min(iv_first + g.du_per_interval, g.total_du) : g.total_du;
    jpg::SubState s = jpg::unpack_state(entry[i]);
    int pred[3] = {dcb[3 * (size_t)i], dcb[3 * (size_t)i + 1], dcb[3 * (size_t)i + 2]};
    uint32_t done = 0;
    if (du_base[i] > iv_stop) return 101;
    if (jpg::decode_span<true>(g, clean, d.huff, zz, s, q.end_bit, du_base[i], iv_stop, pred, cp, &done)) return 101;
    if (q.last) {
      if (du_base[i] + done != iv_stop || s.u != 0 || s.k != 0) return 101;
      if (s.bit > q.iv_end_bit) return 102;                     // consumed bits that are not in the file
    } else if (jpg::pack_state(s) != entry[i + 1]) {
      return 101;                                               // (ran into the interval's block limit early)
    }
  }
  finish_image(d, cp, rgb);
  return 0;
}

}  // extern "C"
