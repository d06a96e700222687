// JPEG marker parser + Huffman table builder (host only; see jpeg_host.h).  Follows ITU-T T.81 Annex B (marker syntax) and
// Annex C (table generation); the colour-space decision is JFIF's / Adobe's APP14 convention as libjpeg applies it.
#include "jpeg_host.h"

#include <string.h>

namespace jpg {

const char* reason_text(int code) {
  static const char* const t[] = {"ok", "not a JPEG file", "progressive / lossless / hierarchical", "sample precision is not 8 bits",
                                  "neither 1 nor 3 components", "sampling factors that do not divide the largest ones",
                                  "scan components out of frame order", "colour space other than grey or YCbCr", "missing table",
                                  "arithmetic coding", "larger than 16384 per side or 178 956 970 pixels", "truncated", "corrupt header"};
  return code >= 0 && code <= JPG_CORRUPT ? t[code] : "?";
}

bool build_huff(const uint8_t counts[16], const uint8_t* vals, int nvals, HuffTable* t) {
  memset(t, 0, sizeof *t);
  int total = 0;
  for (int l = 0; l < 16; ++l) total += counts[l];
  if (total != nvals || total > 256) return false;
  memcpy(t->huffval, vals, (size_t)total);
  int code = 0, k = 0;
  for (int l = 1; l <= 16; ++l) {
    const int n = counts[l - 1];
    if (n) {
      t->valoffset[l] = k - code;
      for (int i = 0; i < n; ++i, ++k, ++code) {
        if (code >= (1 << l)) return false;                      // over-subscribed
        if (l <= 9) {
          const int first = code << (9 - l);
          for (int f = 0; f < (1 << (9 - l)); ++f) t->look[first + f] = (uint16_t)((l << 8) | vals[k]);
        }
      }
      t->maxcode[l] = code - 1;
    } else {
      t->maxcode[l] = -1;
      t->valoffset[l] = 0;
    }
    code <<= 1;
  }
  t->maxcode[17] = 0x7fffffff;
  return true;
}

namespace {
inline int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }
}

long unstuff_scan(const uint8_t* scan, size_t len, uint8_t* clean, uint32_t* interval_start, int max_intervals, int* n_intervals,
                  bool require_eoi) {
  size_t i = 0, o = 0;
  int niv = 1, next_rst = 0;
  interval_start[0] = 0;
  bool eoi = false;
  while (i < len) {
    const uint8_t* f = (const uint8_t*)memchr(scan + i, 0xFF, len - i);
    const size_t run = f ? (size_t)(f - (scan + i)) : len - i;
    memcpy(clean + o, scan + i, run);
    o += run; i += run;
    if (!f) break;
    // scan[i] == 0xFF
    size_t j = i + 1;
    while (j < len && scan[j] == 0xFF) ++j;                       // fill bytes
    if (j >= len) { i = len; break; }                             // the data ends inside a marker
    const int m = scan[j];
    if (m == 0x00) {
      if (j != i + 1) return -1;                                  // 0xFF 0xFF 0x00: not valid entropy data
      clean[o++] = 0xFF; i = j + 1;
    } else if (m >= 0xD0 && m <= 0xD7) {
      if (m != 0xD0 + next_rst) return -1;                        // restart markers count 0 .. 7 cyclically
      next_rst = (next_rst + 1) & 7;
      if (niv >= max_intervals) return -1;
      interval_start[niv++] = (uint32_t)o;
      i = j + 1;
    } else if (m == 0xD9) {
      eoi = true;
      break;
    } else {
      return -1;                                                  // another marker inside the scan (DNL, a second scan, ...)
    }
  }
  if (!eoi && require_eoi) return -1;
  memset(clean + o, 0xFF, 16);
  *n_intervals = niv;
  return (long)o;
}

namespace {

// end of the entropy-coded data that starts at `pos`: the position of the next marker that is not RSTn (stuffed zeros and fill
// bytes skipped); len if the file ends first
size_t find_scan_end(const uint8_t* data, size_t len, size_t pos) {
  while (pos < len) {
    const uint8_t* f = (const uint8_t*)memchr(data + pos, 0xFF, len - pos);
    if (!f) return len;
    size_t i = (size_t)(f - data), j = i + 1;
    while (j < len && data[j] == 0xFF) ++j;
    if (j >= len) return len;
    const int m = data[j];
    if (m == 0x00 || (m >= 0xD0 && m <= 0xD7)) { pos = j + 1; continue; }
    return i;
  }
  return len;
}

}  // namespace

int parse_jpeg(const uint8_t* data, size_t len, ImageDesc* d, size_t* scan_off, size_t* scan_len, ProgInfo* prog) {
  memset(d, 0, sizeof *d);
  *scan_off = 0; *scan_len = 0;
  if (len < 4 || data[0] != 0xFF || data[1] != 0xD8 || data[2] != 0xFF) return JPG_NOT_JPEG;   // (Pillow identifies JPEG by these three bytes)
  uint16_t qt[4][64];
  bool have_qt[4] = {false, false, false, false}, have_ht[4] = {false, false, false, false};
  int comp_id[MAX_COMPS] = {0, 0, 0}, comp_tq[MAX_COMPS] = {0, 0, 0};
  bool have_sof = false, jfif = false, adobe = false, progressive = false, multi = false, frame_done = false;
  int adobe_transform = -1;
  int cur_tab[4] = {-1, -1, -1, -1};                              // progressive: index in prog->tabs of DC 0, DC 1, AC 0, AC 1
  int last_al[MAX_COMPS][64];                                     // progressive: -1 = coefficient not coded yet, else the bit it is refined down to
  for (auto& row : last_al) for (int& v : row) v = -1;

  // colour space, sampling, geometry, quantisation tables: at the first SOS
  auto finish_frame = [&]() -> int {
    if (d->ncomp == 3) {
      // (libjpeg's rules): JFIF = YCbCr, Adobe transform 0 = RGB, 1 = YCbCr, otherwise by the component ids
      bool ycc = true;
      if (jfif) ycc = true;
      else if (adobe) ycc = adobe_transform == 1;
      else if (comp_id[0] == 'R' && comp_id[1] == 'G' && comp_id[2] == 'B') ycc = false;
      if (!ycc) return JPG_COLORSPACE;
      // any sampling libjpeg upsamples: every component's factors divide the largest ones, at most 10 blocks per MCU
      d->hmax = d->vmax = 1;
      int blocks = 0;
      for (int c = 0; c < 3; ++c) {
        if (d->hs[c] > d->hmax) d->hmax = d->hs[c];
        if (d->vs[c] > d->vmax) d->vmax = d->vs[c];
        blocks += d->hs[c] * d->vs[c];
      }
      if (blocks > 10) return JPG_CORRUPT;
      for (int c = 0; c < 3; ++c)
        if (d->hmax % d->hs[c] || d->vmax % d->vs[c]) return JPG_SAMPLING;       // (3 into 2 etc.: libjpeg refuses these too)
    } else {
      d->hs[0] = d->vs[0] = 1;                                    // a single-component scan is never interleaved: 8 x 8 "MCUs"
      d->hmax = d->vmax = 1;
    }
    d->mcus_x = (d->width + 8 * d->hmax - 1) / (8 * d->hmax);
    d->mcus_y = (d->height + 8 * d->vmax - 1) / (8 * d->vmax);
    for (int c = 0; c < d->ncomp; ++c) {
      if (!have_qt[comp_tq[c]]) return JPG_TABLES;
      memcpy(d->quant[c], qt[comp_tq[c]], sizeof d->quant[c]);
      d->bw[c] = d->mcus_x * d->hs[c]; d->bh[c] = d->mcus_y * d->vs[c];
      d->dw[c] = (d->width * d->hs[c] + d->hmax - 1) / d->hmax;
      d->dh[c] = (d->height * d->vs[c] + d->vmax - 1) / d->vmax;
    }
    frame_done = true;
    return JPG_OK;
  };

  size_t pos = 2;
  for (;;) {
    while (pos < len && data[pos] != 0xFF) ++pos;                // (garbage between segments is skipped, as libjpeg does)
    while (pos < len && data[pos] == 0xFF) ++pos;
    if (pos >= len) return JPG_TRUNCATED;
    const int m = data[pos++];
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (m == 0xD9) {                                              // EOI
      if (!(progressive || multi) || !prog || prog->scans.empty()) return JPG_TRUNCATED;     // before any scan
      for (int c = 0; c < d->ncomp; ++c)
        for (int k = 0; k < 64; ++k)
          if (last_al[c][k] != 0) return JPG_PROGRESSIVE;         // an unfinished progression (Pillow would smooth it)
      return JPG_OK;
    }
    if (pos + 2 > len) return JPG_TRUNCATED;
    const int seg = be16(data + pos);
    if (seg < 2 || pos + (size_t)seg > len) return JPG_TRUNCATED;
    const uint8_t* s = data + pos + 2;
    const int n = seg - 2;
    if (m == 0xE0 && n >= 14 && !memcmp(s, "JFIF\0", 5)) jfif = true;   // libjpeg's examine_app0: a JFIF APP0 has >= 14 data bytes
    else if (m == 0xEE && n >= 12 && !memcmp(s, "Adobe", 5)) { adobe = true; adobe_transform = s[11]; }
    else if (m == 0xDB) {                                         // DQT
      if (frame_done) return JPG_TABLES;                          // (redefined between scans: not met in practice)
      int i = 0;
      while (i < n) {
        const int pq = s[i] >> 4, tq = s[i] & 15;
        ++i;
        if (tq > 3 || pq > 1 || i + 64 * (pq + 1) > n) return JPG_CORRUPT;
        for (int k = 0; k < 64; ++k) {
          const int v = pq ? be16(s + i + 2 * k) : s[i + k];
          qt[tq][zigzag_to_natural(k)] = (uint16_t)v;
        }
        i += 64 * (pq + 1);
        have_qt[tq] = true;
      }
    } else if (m == 0xC4) {                                       // DHT
      int i = 0;
      while (i < n) {
        if (i + 17 > n) return JPG_CORRUPT;
        const int tc = s[i] >> 4, th = s[i] & 15;
        int total = 0;
        for (int l = 0; l < 16; ++l) total += s[i + 1 + l];
        if (tc > 1 || i + 17 + total > n) return JPG_CORRUPT;
        if (th > 1) return JPG_TABLES;                            // (ids 2, 3: extended sequential / progressive only; not met in practice)
        if (tc == 0)
          for (int k = 0; k < total; ++k)
            if (s[i + 17 + k] > 15) return JPG_CORRUPT;          // a DC table codes categories 0 .. 15
        if (!build_huff(s + i + 1, s + i + 17, total, &d->huff[tc * 2 + th])) return JPG_CORRUPT;
        have_ht[tc * 2 + th] = true;
        if (prog) { prog->tabs.push_back(d->huff[tc * 2 + th]); cur_tab[tc * 2 + th] = (int)prog->tabs.size() - 1; }
        i += 17 + total;
      }
    } else if (m == 0xDD) {                                       // DRI
      if (n != 2) return JPG_CORRUPT;
      d->restart_interval = be16(s);
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {             // SOF0 / SOF1: sequential Huffman; SOF2: progressive Huffman
      if (n < 6 || have_sof) return JPG_CORRUPT;
      if (m == 0xC2) {
        if (!prog) return JPG_PROGRESSIVE;
        progressive = true;
      }
      if (s[0] != 8) return JPG_PRECISION;
      d->height = be16(s + 1); d->width = be16(s + 3); d->ncomp = s[5];
      if (d->width < 1 || d->height < 1) return JPG_CORRUPT;     // (height 0 = DNL marker: not supported)
      // 16384 per side, and Pillow's decompression-bomb limit (2 x Image.MAX_IMAGE_PIXELS): Pillow raises on such a file, so it is
      // left to the caller's Pillow path to do so -- and a few header bytes cannot make the planner reserve gigabytes
      if (d->width > 16384 || d->height > 16384 || (uint64_t)d->width * (uint64_t)d->height > 178956970ull) return JPG_TOO_LARGE;
      if (d->ncomp != 1 && d->ncomp != 3) return JPG_COMPONENTS;
      if (n != 6 + 3 * d->ncomp) return JPG_CORRUPT;
      for (int c = 0; c < d->ncomp; ++c) {
        comp_id[c] = s[6 + 3 * c];
        d->hs[c] = s[7 + 3 * c] >> 4; d->vs[c] = s[7 + 3 * c] & 15;
        comp_tq[c] = s[8 + 3 * c];
        if (comp_tq[c] > 3 || d->hs[c] < 1 || d->vs[c] < 1 || d->hs[c] > 4 || d->vs[c] > 4) return JPG_CORRUPT;
      }
      have_sof = true;
    } else if ((m >= 0xC3 && m <= 0xCF) && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return (m == 0xC9 || m == 0xCA || m == 0xCB || m == 0xCD || m == 0xCE || m == 0xCF) ? JPG_ARITHMETIC : JPG_PROGRESSIVE;
    } else if (m == 0xCC) {
      return JPG_ARITHMETIC;
    } else if (!((m >= 0xE0 && m <= 0xEF) || m == 0xFE) && m != 0xDA) {
      return JPG_CORRUPT;                                         // not a segment libjpeg skips: Pillow decides what the file is
    } else if (m == 0xDA) {                                       // SOS
      if (!have_sof) return JPG_CORRUPT;
      const int ns = n >= 1 ? s[0] : 0;
      if (ns < 1 || ns > d->ncomp) return JPG_CORRUPT;
      if (n != 1 + 2 * ns + 3) return JPG_CORRUPT;
      if (!frame_done)
        if (const int rc = finish_frame()) return rc;
      const uint8_t* e = s + 1 + 2 * ns;
      if (!progressive && ns == d->ncomp) {
        for (int c = 0; c < d->ncomp; ++c) {
          if (s[1 + 2 * c] != comp_id[c]) return JPG_MULTI_SCAN;   // (components in frame order)
          const int td = s[2 + 2 * c] >> 4, ta = s[2 + 2 * c] & 15;
          if (td > 1 || ta > 1) return JPG_TABLES;
          if (!have_ht[td] || !have_ht[2 + ta]) return JPG_TABLES;
          d->dc_tab[c] = td; d->ac_tab[c] = 2 + ta;
        }
        if (e[0] != 0 || e[1] != 63 || e[2] != 0) return JPG_CORRUPT;
        *scan_off = pos + (size_t)seg;
        *scan_len = len - *scan_off;
        return JPG_OK;
      }
      // ---- a scan of a progressive file, or of a sequential file whose components come in several scans (full band each)
      if (!prog) return JPG_MULTI_SCAN;
      multi = true;
      if (prog->scans.size() >= MAX_SCANS) return JPG_PROGRESSIVE;
      ProgScanInfo si;
      memset(&si, 0, sizeof si);
      ProgScan& ps = si.s;
      ps.ncomp = ns;
      ps.ss = e[0]; ps.se = e[1]; ps.ah = e[2] >> 4; ps.al = e[2] & 15;
      ps.restart_interval = d->restart_interval;
      int prev = -1;
      for (int i = 0; i < ns; ++i) {
        int c = -1;
        for (int q = 0; q < d->ncomp; ++q) if (comp_id[q] == s[1 + 2 * i]) c = q;
        if (c <= prev) return JPG_CORRUPT;                        // unknown component / not in frame order
        prev = c;
        ps.comp[i] = c;
        const int td = s[2 + 2 * i] >> 4, ta = s[2 + 2 * i] & 15;
        if (td > 1 || ta > 1) return JPG_TABLES;
        if (ps.ss == 0 && ps.ah == 0) { if (cur_tab[td] < 0) return JPG_TABLES; ps.dc_tab[i] = cur_tab[td]; }
        if (ps.se > 0) { if (cur_tab[2 + ta] < 0) return JPG_TABLES; ps.ac_tab = cur_tab[2 + ta]; ps.ac_tab3[i] = cur_tab[2 + ta]; }
      }
      if (!progressive && !(ps.ss == 0 && ps.se == 63 && ps.ah == 0 && ps.al == 0)) return JPG_CORRUPT;   // sequential: full band, full precision
      // the scan must be a legal step of an orderly progression (T.81 G.1.1.1.1): DC scans cover coefficient 0 only and may
      // interleave components, AC scans cover one component; first pass Ah = 0, every later pass refines exactly the next bit
      if (ps.ss > ps.se || ps.se > 63 || ps.al > 13 || (progressive && ps.ss == 0 && ps.se != 0) || (ps.ss > 0 && ns != 1)) return JPG_CORRUPT;
      if (ps.ah != 0 && ps.al != ps.ah - 1) return JPG_CORRUPT;
      for (int i = 0; i < ns; ++i)
        for (int k = ps.ss; k <= ps.se; ++k) {
          int& la = last_al[ps.comp[i]][k];
          if (ps.ah == 0 ? la != -1 : la != ps.ah) return JPG_PROGRESSIVE;   // (out of order: libjpeg only warns; Pillow decodes it)
          la = ps.al;
        }
      if (ps.ss > 0 && last_al[ps.comp[0]][0] < 0) return JPG_PROGRESSIVE;   // AC before the component's DC
      si.begin = pos + (size_t)seg;
      si.end = find_scan_end(data, len, si.begin);
      if (si.end >= len) return JPG_TRUNCATED;
      prog->scans.push_back(si);
      pos = si.end;
      continue;
    }
    pos += (size_t)seg;
  }
}

}  // namespace jpg
