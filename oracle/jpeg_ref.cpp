// TEST INFRASTRUCTURE ONLY.  CPU run of the JPEG decoding arithmetic (clip_assisted_data_labeling_amd/csrc/jpeg_core.h +
// jpeg_host.cpp, the same sources the HIP kernels are compiled from) so that tests can compare it with Pillow -- the reference's
// own decoder, /root/reference/utils/embedder.py:167 -- without a GPU.  Never linked into, loaded or called by the product library.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../clip_assisted_data_labeling_amd/csrc/jpeg_host.h"

extern "C" {

// returns a jpg:: reason code (0 = decodable); width / height are filled whenever the header could be read
int jpeg_ref_info(const uint8_t* data, size_t len, int* width, int* height, int* ncomp) {
  jpg::ImageDesc d;
  size_t so = 0, sl = 0;
  const int rc = jpg::parse_jpeg(data, len, &d, &so, &sl);
  *width = d.width; *height = d.height; *ncomp = d.ncomp;
  return rc;
}

// rgb: [height][width][3]; returns 0, a parse reason code, or 100 + the entropy decoder's status
int jpeg_ref_decode(const uint8_t* data, size_t len, uint8_t* rgb) {
  static jpg::ImageDesc d;                                     // (6 KiB of tables)
  size_t so = 0, sl = 0;
  const int rc = jpg::parse_jpeg(data, len, &d, &so, &sl);
  if (rc) return rc;
  // the entropy-coded segment, 16-byte aligned and padded the way the product's host side pads it
  const size_t padded = (sl + 15) / 16 * 16 + 32;
  std::vector<uint64_t> seg(padded / 8 + 1);
  memset(seg.data(), 0xFF, padded);
  memcpy(seg.data(), data + so, sl);
  for (size_t i = sl; i + 1 < padded; i += 2) { ((uint8_t*)seg.data())[i] = 0xFF; ((uint8_t*)seg.data())[i + 1] = 0xD9; }
  d.data_len = (uint32_t)padded;
  d.data_real = (uint32_t)sl;
  std::vector<std::vector<int16_t>> coef(d.ncomp);
  std::vector<std::vector<uint8_t>> plane(d.ncomp);
  int16_t* cp[jpg::MAX_COMPS] = {nullptr, nullptr, nullptr};
  for (int c = 0; c < d.ncomp; ++c) {
    coef[c].assign((size_t)d.bw[c] * d.bh[c] * 64, 0);
    plane[c].assign((size_t)d.bw[c] * d.bh[c] * 64, 0);
    cp[c] = coef[c].data();
  }
  uint8_t zz[64];
  for (int k = 0; k < 64; ++k) zz[k] = (uint8_t)jpg::zigzag_to_natural(k);
  const int st = jpg::decode_scan(jpg::scan_geom(d), (const uint8_t*)seg.data(), cp, d.huff, zz);
  if (st) return 100 + st;
  for (int c = 0; c < d.ncomp; ++c) {
    const int pitch = d.bw[c] * 8;
    for (int by = 0; by < d.bh[c]; ++by)
      for (int bx = 0; bx < d.bw[c]; ++bx)
        jpg::idct_block(cp[c] + ((size_t)by * d.bw[c] + bx) * 64, d.quant[c], plane[c].data() + (size_t)by * 8 * pitch + bx * 8, pitch);
  }
  for (int y = 0; y < d.height; ++y)
    for (int x = 0; x < d.width; ++x) {
      uint8_t* o = rgb + ((size_t)y * d.width + x) * 3;
      const int Y = plane[0][(size_t)y * d.bw[0] * 8 + x];
      if (d.ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; continue; }
      const int h = d.hmax / d.hs[1], v = d.vmax / d.vs[1];
      const int cb = jpg::upsampled(plane[1].data(), d.bw[1] * 8, d.dw[1], d.dh[1], h, v, x, y);
      const int cr = jpg::upsampled(plane[2].data(), d.bw[2] * 8, d.dw[2], d.dh[2], h, v, x, y);
      jpg::ycc_to_rgb(Y, cb, cr, o);
    }
  return 0;
}

}  // extern "C"
