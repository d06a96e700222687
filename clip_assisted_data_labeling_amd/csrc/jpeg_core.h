// Baseline JPEG decoding arithmetic shared by the HIP kernels (jpeg_decode.hip) and the CPU checker of the test infrastructure (jpeg_ref.cpp):
// the entropy decoder of ITU-T T.81 Annex F.2.2 and the three sample-domain steps exactly as the reference's image loader
// performs them -- /root/reference/utils/embedder.py:167 opens every file with PIL.Image.open(...).convert('RGB'), and Pillow's
// JPEG plugin is libjpeg-turbo with its defaults: the "islow" integer inverse DCT (Loeffler-Ligtenberg-Moschytz, 13-bit
// constants, two passes), "fancy" (triangle-filter) chroma upsampling for 2x1 and 2x2 subsampled components, and the 16-bit
// fixed-point YCbCr -> RGB tables of JFIF.  All three are integer algorithms, so the decoded pixels can be -- and are tested to
// be -- identical to Pillow's, bit for bit (tests/test_cpu_jpeg.py through the checker, tests/test_gpu_jpeg.py on the device).
// Restated from the published algorithms (T.81; the LL&M factorisation with libjpeg's documented scaling; JFIF 1.02); the
// reference repository holds no JPEG code of its own.
#pragma once
#include <stdint.h>

#ifdef __HIP__               /* a HIP translation unit (jpeg_decode.hip); plain C++ everywhere else */
#define JPG_HD __host__ __device__ __forceinline__
#else
#define JPG_HD static inline
#endif

namespace jpg {

enum { MAX_COMPS = 3 };

// One Huffman table, prepared on the host (jpeg_host.cpp): 9-bit lookahead + the canonical-code arrays of T.81 F.2.2.3
struct HuffTable {
  uint16_t look[512];        // index = next 9 bits: (code length << 8) | symbol, 0 = the code is longer than 9 bits
  int32_t maxcode[18];       // largest code of length l (1..16), -1 if none; [17] = sentinel above every 16-bit code
  int32_t valoffset[17];     // huffval index of the first code of length l, minus that code
  uint8_t huffval[256];
};

// Per-image decode plan (host-built; every offset is relative to the batch's device arena)
struct ImageDesc {
  int32_t width, height;                 // luma / output size
  int32_t ncomp;                         // 1 (grey) or 3 (Y Cb Cr)
  int32_t hs[MAX_COMPS], vs[MAX_COMPS];  // sampling factors
  int32_t hmax, vmax;
  int32_t mcus_x, mcus_y;                // interleaved scan: MCUs per row / column (1 component: 8 x 8 blocks)
  int32_t bw[MAX_COMPS], bh[MAX_COMPS];  // blocks per row / column of the component's (padded) plane
  int32_t dw[MAX_COMPS], dh[MAX_COMPS];  // real ("downsampled") size of the component in samples
  int32_t restart_interval;              // MCUs between RSTn markers, 0 = none
  int32_t dc_tab[MAX_COMPS], ac_tab[MAX_COMPS];   // index into huff[4]: 0,1 = DC tables 0,1; 2,3 = AC tables 0,1
  uint16_t quant[MAX_COMPS][64];         // natural (row-major) order
  uint64_t data_off;                     // entropy-coded segment, 16-byte aligned, padded with 0xFF 0xD9 ...
  uint32_t data_len;
  uint64_t coef_off[MAX_COMPS];          // int16 [bh][bw][64], zeroed before the entropy kernel
  uint64_t plane_off[MAX_COMPS];         // uint8 [bh * 8][bw * 8]
  uint64_t rgb_off;                      // uint8 [height][width][3]
  uint64_t block_base;                   // index of the image's first 8 x 8 block in the batch-wide block numbering
  uint32_t n_blocks;
  uint32_t data_real;                    // bytes of data_len that come from the file (the rest is the host's padding)
  // parallel entropy decoding (below): the unstuffed stream, its restart intervals, per-subsequence scratch
  uint64_t clean_off;                    // 4-byte aligned, 16 bytes of 0xFF behind clean_len
  uint32_t clean_len;
  int32_t n_iv;                          // restart intervals (1 without DRI)
  uint64_t iv_off;                       // uint32 iv_byte[n_iv + 1], then uint32 iv_sub[n_iv + 1]
  uint32_t n_sub;                        // subsequences of the image
  int32_t sub_bytes;
  uint64_t sub_off;                      // scratch: n_sub x { u64 entry, u64 exit, u32 blocks, u32 du_base, i32 dc[3], i32 dc_base[3] }
  uint64_t prog_off;                     // 0 = a sequential file; else its ProgDesc (progressive: several scans, decoded serially)
  HuffTable huff[4];
};

JPG_HD int zigzag_to_natural(int k) {
  constexpr uint8_t t[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  return t[k & 63];
}
// (the entropy kernel keeps a copy of that table in LDS and passes it as `zz`: a constant array is a global-memory load per
// coefficient on the device)

// ------------------------------------------------------------------------------------------------ entropy-coded segment
// Bytes come from 8-byte words (the segment starts 16-byte aligned and is padded, so whole words can always be read);
// 0xFF 0x00 is a data byte 0xFF, 0xFF 0xFF.. are fill bytes, 0xFF <other> is a marker: it ends the data (zero bits are fed
// from there on, as libjpeg does) until a restart consumes it.
struct BitReader {
  const uint64_t* wp;
  uint64_t word, ahead;        // `ahead` is the word behind `word`, loaded one word early: its latency passes under 8 bytes of decoding
  int wleft;
  uint64_t bitbuf;             // next bits at the top
  int bits;
  int marker;                  // 0 = none pending
  int fake;                    // zero bits appended behind a marker that are still unconsumed at most `bits`: consuming one = the data ran out
  uint32_t consumed, limit;    // bytes taken from the segment / its padded length
};

JPG_HD void br_init(BitReader& br, const uint8_t* base, uint32_t padded_len) {
  br.wp = (const uint64_t*)base; br.word = 0; br.wleft = 0; br.bitbuf = 0; br.bits = 0; br.marker = 0; br.fake = 0; br.consumed = 0;
  br.limit = padded_len;
  br.ahead = *br.wp++;                                         // (the segment is at least 32 bytes long with its padding)
}

JPG_HD int br_raw_byte(BitReader& br) {
  if (br.consumed >= br.limit) return 0xD9;                    // (cannot happen with the host's padding; keeps a bad file inside its buffer)
  if (br.wleft == 0) {
    br.word = br.ahead; br.wleft = 8;
    if (br.consumed + 16 <= br.limit) br.ahead = *br.wp++;     // never reads past the padded segment
  }
  const int b = (int)(br.word & 0xffu);
  br.word >>= 8; br.wleft--; br.consumed++;
  return b;
}

JPG_HD void br_fill(BitReader& br) {
  while (br.bits <= 56) {
    int b = 0;
    if (!br.marker) {
      b = br_raw_byte(br);
      if (b == 0xFF) {
        int b2 = br_raw_byte(br);
        while (b2 == 0xFF) b2 = br_raw_byte(br);               // fill bytes
        if (b2 != 0) { br.marker = b2; b = 0; }
      }
    }
    if (br.marker) br.fake += 8;
    br.bitbuf |= (uint64_t)b << (56 - br.bits);
    br.bits += 8;
  }
}

JPG_HD int br_peek16(BitReader& br) {
  if (br.bits < 16) br_fill(br);
  return (int)(br.bitbuf >> 48);
}

JPG_HD void br_skip(BitReader& br, int n) { br.bitbuf <<= n; br.bits -= n; }

JPG_HD int br_get(BitReader& br, int n) {                      // n = 1 .. 16
  if (br.bits < n) br_fill(br);
  const int v = (int)(br.bitbuf >> (64 - n));
  br_skip(br, n);
  return v;
}

// one Huffman symbol (T.81 F.2.2.3 DECODE with a 9-bit lookahead); -1 = no such code
JPG_HD int huff_decode(BitReader& br, const HuffTable& t) {
  const int c = br_peek16(br);
  const int e = t.look[c >> 7];
  if (e) { br_skip(br, e >> 8); return e & 0xff; }
  int l = 10;
  while (l <= 16 && (c >> (16 - l)) > t.maxcode[l]) ++l;
  if (l > 16) return -1;
  br_skip(br, l);
  return t.huffval[((c >> (16 - l)) + t.valoffset[l]) & 0xff];
}

JPG_HD int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }   // T.81 F.2.2.1 EXTEND

// one 8 x 8 block: DC difference + AC run-lengths -> the NONZERO coefficients into coef (natural order; the rest stays 0)
JPG_HD bool decode_block(BitReader& br, const HuffTable& dc, const HuffTable& ac, const uint8_t* zz, int& pred, int16_t* coef) {
  int s = huff_decode(br, dc);
  if (s < 0 || s > 11) return false;
  if (s) pred += extend(br_get(br, s), s);
  coef[0] = (int16_t)pred;
  for (int k = 1; k < 64;) {
    const int rs = huff_decode(br, ac);
    if (rs < 0) return false;
    const int r = rs >> 4;
    s = rs & 15;
    if (s == 0) {
      if (r != 15) break;                                      // EOB
      k += 16;                                                 // ZRL
      continue;
    }
    k += r;
    if (k > 63) return false;
    coef[zz[k]] = (int16_t)extend(br_get(br, s), s);
    ++k;
  }
  return true;
}

// true if bits that were never in the file have been consumed (a truncated file: Pillow refuses those, "image file is truncated")
JPG_HD bool br_starved(const BitReader& br) { return br.bits < br.fake; }

// after a restart interval: drop the bit remainder, the pending marker must be RSTn
JPG_HD bool br_restart(BitReader& br) {
  if (br_starved(br)) return false;
  br.bitbuf = 0; br.bits = 0; br.fake = 0;
  if (!br.marker) {                                            // the marker has not been run into yet: it is the next thing in the data
    int b = br_raw_byte(br);
    if (b != 0xFF) return false;
    while (b == 0xFF) b = br_raw_byte(br);
    br.marker = b;
  }
  if (br.marker < 0xD0 || br.marker > 0xD7) return false;
  br.marker = 0;
  return true;
}

// what the scan walk needs of an ImageDesc, by value (registers on the device: the descriptor itself lives in global memory,
// and every coefficient store could alias it)
struct ScanGeom {
  int ncomp, mcus_x, mcus_y, restart_interval;
  int h[MAX_COMPS], v[MAX_COMPS], bw[MAX_COMPS], dc[MAX_COMPS], ac[MAX_COMPS];
  uint32_t data_len, data_real;
};

JPG_HD ScanGeom scan_geom(const ImageDesc& d) {
  ScanGeom g;
  g.ncomp = d.ncomp; g.mcus_x = d.mcus_x; g.mcus_y = d.mcus_y; g.restart_interval = d.restart_interval;
  for (int c = 0; c < MAX_COMPS; ++c) {
    g.h[c] = d.ncomp == 1 ? 1 : d.hs[c]; g.v[c] = d.ncomp == 1 ? 1 : d.vs[c];
    g.bw[c] = d.bw[c]; g.dc[c] = d.dc_tab[c]; g.ac[c] = d.ac_tab[c];
  }
  g.data_len = d.data_len; g.data_real = d.data_real;
  return g;
}

// the whole scan of one image, sequentially: coef planes of every component.  0 = ok, 1 = invalid code / coefficient index /
// restart marker, 2 = the data ran out, 3 = no EOI marker behind the scan
JPG_HD int decode_scan(const ScanGeom g, const uint8_t* arena_data, int16_t* const coef[MAX_COMPS], const HuffTable* huff, const uint8_t* zz) {
  BitReader br;
  br_init(br, arena_data, g.data_len);
  int pred0 = 0, pred1 = 0, pred2 = 0;
  int to_restart = g.restart_interval;
  for (int my = 0; my < g.mcus_y; ++my) {
    for (int mx = 0; mx < g.mcus_x; ++mx) {
      if (g.restart_interval) {
        if (to_restart == 0) {
          if (!br_restart(br)) return 1;
          pred0 = pred1 = pred2 = 0;
          to_restart = g.restart_interval;
        }
        --to_restart;
      }
#if defined(__HIP__)
#pragma unroll
#endif
      for (int c = 0; c < MAX_COMPS; ++c) {
        if (c < g.ncomp) {
          int& pred = c == 0 ? pred0 : c == 1 ? pred1 : pred2;
          for (int by = 0; by < g.v[c]; ++by)
            for (int bx = 0; bx < g.h[c]; ++bx) {
              const int brow = my * g.v[c] + by, bcol = mx * g.h[c] + bx;
              if (!decode_block(br, huff[g.dc[c]], huff[g.ac[c]], zz, pred, coef[c] + ((size_t)brow * g.bw[c] + bcol) * 64)) return 1;
            }
        }
      }
    }
  }
  if (br_starved(br)) return 2;
  // behind the last MCU the next marker must be EOI and must come from the file, not from the padding: Pillow refuses a file that
  // ends before it ("image file is truncated")
  if (!br.marker) {
    int b = br_raw_byte(br);
    while (b != 0xFF && br.consumed < br.limit) b = br_raw_byte(br);
    while (b == 0xFF && br.consumed < br.limit) b = br_raw_byte(br);
    br.marker = b;
  }
  return (br.marker == 0xD9 && br.consumed <= g.data_real) ? 0 : 3;
}

// ------------------------------------------------------------------------------------------------ parallel entropy decoding
// A Huffman-coded scan is serial as written -- every code's position depends on all codes before it -- but it re-synchronises:
// a decoder started at a wrong bit or in a wrong state (which block of the MCU, which coefficient index) falls back into step
// with the true symbol sequence after a few symbols.  That makes the scan decodable in parallel (Klein & Wiseman 2003;
// Weissenberger & Schmidt 2018/2021 for JPEG on GPUs): the unstuffed stream (jpeg_host.cpp: unstuff_scan) is cut into
// subsequences of `sub_bytes` bytes; each keeps an ENTRY state (bit position, block-in-MCU u, coefficient index k), a guess at
// first except for the first subsequence of every restart interval, whose state is known; every pass decodes each subsequence
// from its entry state up to the first symbol boundary at or behind its end and hands that EXIT state to the next one as its new
// entry state; when a pass changes nothing, entry(i + 1) = exit(i) holds for all i and entry(0) is true, so by induction every
// state is true.  Counting the blocks completed and summing the DC differences per component in that last pass gives, by a prefix
// sum, the block index and DC predictors each subsequence starts with; a final pass decodes every subsequence once more and
// writes the coefficients -- the same values, in the same places, as the serial walk.
struct SubState { uint32_t bit; int u, k; };
JPG_HD uint64_t pack_state(const SubState& s) { return (uint64_t)s.bit | (uint64_t)(uint32_t)s.u << 32 | (uint64_t)(uint32_t)s.k << 40; }
JPG_HD SubState unpack_state(uint64_t v) { SubState s; s.bit = (uint32_t)v; s.u = (int)((v >> 32) & 0xff); s.k = (int)((v >> 40) & 0xff); return s; }

struct ParGeom {
  int ncomp, B, mcus_x;                    // B: blocks ("data units") per MCU
  int comp_of_u[10], bx_of_u[10], by_of_u[10];   // (T.81: at most 10 blocks per MCU)
  int h[MAX_COMPS], v[MAX_COMPS], bw[MAX_COMPS], dc[MAX_COMPS], ac[MAX_COMPS];
  uint32_t total_du, du_per_interval;      // du_per_interval: restart_interval * B, 0 = one interval
};

JPG_HD ParGeom par_geom(const ImageDesc& d) {
  ParGeom g;
  g.ncomp = d.ncomp; g.mcus_x = d.mcus_x;
  int u = 0;
  for (int c = 0; c < MAX_COMPS; ++c) {
    g.h[c] = d.ncomp == 1 ? 1 : d.hs[c]; g.v[c] = d.ncomp == 1 ? 1 : d.vs[c];
    g.bw[c] = d.bw[c]; g.dc[c] = d.dc_tab[c]; g.ac[c] = d.ac_tab[c];
    if (c < d.ncomp)
      for (int by = 0; by < g.v[c]; ++by)
        for (int bx = 0; bx < g.h[c]; ++bx)
          if (u < 10) { g.comp_of_u[u] = c; g.bx_of_u[u] = bx; g.by_of_u[u] = by; ++u; }
  }
  g.B = u;
  for (; u < 10; ++u) { g.comp_of_u[u] = 0; g.bx_of_u[u] = 0; g.by_of_u[u] = 0; }
  g.total_du = (uint32_t)d.mcus_x * (uint32_t)d.mcus_y * (uint32_t)g.B;
  g.du_per_interval = (uint32_t)d.restart_interval * (uint32_t)g.B;
  return g;
}

JPG_HD uint32_t bswap32(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xff00u) | ((v << 8) & 0xff0000u) | (v << 24); }

// the 32 bits at bit position `bit` of the unstuffed stream (4-byte aligned, 16 bytes of 0xFF behind its end)
JPG_HD uint32_t peek32(const uint8_t* clean, uint32_t bit) {
  const uint32_t* w = (const uint32_t*)clean + (bit >> 5);
  const uint32_t a = bswap32(w[0]), b = bswap32(w[1]);
  const int sh = (int)(bit & 31);
  return sh ? (a << sh) | (b >> (32 - sh)) : a;
}

// where subsequence i lies: restart intervals start at clean-stream bytes iv_byte[j] (iv_byte[n_iv] = length of the stream) and
// hold the subsequences iv_sub[j] .. iv_sub[j + 1] - 1, each sub_bytes long except the last of an interval
struct SubSeq { uint32_t start_bit, end_bit, iv_end_bit; int interval; bool first, last; };
JPG_HD SubSeq subseq_of(const uint32_t* iv_byte, const uint32_t* iv_sub, int n_iv, int sub_bytes, uint32_t i) {
  int lo = 0, hi = n_iv - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (iv_sub[mid] <= i) lo = mid; else hi = mid - 1;
  }
  SubSeq q;
  q.interval = lo;
  const uint32_t local = i - iv_sub[lo];
  const uint32_t b0 = iv_byte[lo] + local * (uint32_t)sub_bytes;
  q.first = local == 0;
  q.last = i + 1 == iv_sub[lo + 1];
  q.iv_end_bit = iv_byte[lo + 1] * 8u;
  q.start_bit = b0 * 8u;
  q.end_bit = q.last ? q.iv_end_bit : (b0 + (uint32_t)sub_bytes) * 8u;
  return q;
}

// Decodes symbols from state s until the first symbol boundary at or behind end_bit (WRITE: or until block index du_stop is
// reached -- the last subsequence of an interval ends by that count; running into end_bit = the interval's end first means the
// data ran out, and it is also what keeps a damaged stream from being read past its buffer).  du: index of the block in progress at entry; pred: WRITE = the running DC predictors, else accumulators of the DC
// differences met; *blocks_done: blocks completed.  Returns 0, or (WRITE only; a speculative pass skips a bit instead) 1 for an
// invalid code / category / coefficient index or more blocks than the interval holds.
template <bool WRITE>
JPG_HD int decode_span(const ParGeom& g, const uint8_t* clean, const HuffTable* huff, const uint8_t* zz, SubState& s, uint32_t end_bit,
                       uint32_t du, uint32_t du_stop, int pred[MAX_COMPS], int16_t* const coef[MAX_COMPS], uint32_t* blocks_done) {
  uint32_t done = 0;
  int16_t* blk = nullptr;
  int mx = 0, my = 0;
  int p0 = pred[0], p1 = pred[1], p2 = pred[2];                 // (scalars: an array indexed by the component lives in scratch memory on the device)
  if (WRITE) {
    const uint32_t mcu = du / (uint32_t)g.B;
    my = (int)(mcu / (uint32_t)g.mcus_x); mx = (int)(mcu - (uint32_t)my * (uint32_t)g.mcus_x);
    if (du < g.total_du) {
      const int c = g.comp_of_u[s.u];
      blk = (c == 0 ? coef[0] : c == 1 ? coef[1] : coef[2]) + ((size_t)(my * g.v[c] + g.by_of_u[s.u]) * g.bw[c] + (mx * g.h[c] + g.bx_of_u[s.u])) * 64;
    }
  }
  for (;;) {
    if (WRITE && du + done >= du_stop) break;
    if (s.bit >= end_bit) break;
    const int c = g.comp_of_u[s.u];
    const uint32_t win = peek32(clean, s.bit);
    const HuffTable& t = huff[s.k == 0 ? g.dc[c] : g.ac[c]];
    const int c16 = (int)(win >> 16);
    int len, sym;
    const int e = t.look[c16 >> 7];
    if (e) { len = e >> 8; sym = e & 0xff; }
    else {
      len = 10;
      while (len <= 16 && (c16 >> (16 - len)) > t.maxcode[len]) ++len;
      if (len > 16) {
        if (WRITE) return 1;
        s.bit += 1;                                              // a speculative start inside garbage: slide on
        continue;
      }
      sym = t.huffval[((c16 >> (16 - len)) + t.valoffset[len]) & 0xff];
    }
    if (s.k == 0) {
      int sz = sym;
      if (sz > 11) { if (WRITE) return 1; sz &= 15; }
      int v = 0;
      if (sz) v = extend((int)((win << len) >> (32 - sz)), sz);
      s.bit += (uint32_t)(len + sz);
      const int pv = (c == 0 ? p0 : c == 1 ? p1 : p2) + v;
      if (c == 0) p0 = pv; else if (c == 1) p1 = pv; else p2 = pv;
      if (WRITE) blk[0] = (int16_t)pv;
      s.k = 1;
    } else {
      const int r = sym >> 4, sz = sym & 15;
      if (sz == 0) {
        s.bit += (uint32_t)len;
        s.k = r == 15 ? s.k + 16 : 64;
      } else {
        s.k += r;
        const int v = extend((int)((win << len) >> (32 - sz)), sz);
        s.bit += (uint32_t)(len + sz);
        if (s.k > 63) { if (WRITE) return 1; }
        else if (WRITE) blk[zz[s.k]] = (int16_t)v;
        s.k++;
      }
    }
    if (s.k >= 64) {                                             // block complete
      s.k = 0;
      ++done;
      if (++s.u == g.B) {
        s.u = 0;
        if (WRITE && ++mx == g.mcus_x) { mx = 0; ++my; }
      }
      if (WRITE && du + done < g.total_du) {
        const int cn = g.comp_of_u[s.u];
        blk = (cn == 0 ? coef[0] : cn == 1 ? coef[1] : coef[2]) + ((size_t)(my * g.v[cn] + g.by_of_u[s.u]) * g.bw[cn] + (mx * g.h[cn] + g.bx_of_u[s.u])) * 64;
      }
    }
  }
  *blocks_done = done;
  pred[0] = p0; pred[1] = p1; pred[2] = p2;
  return 0;
}

// ------------------------------------------------------------------------------------------------ progressive scans
// T.81 Annex G: the coefficients arrive in several scans -- spectral selection (a band Ss .. Se of the zigzag sequence per scan) and
// successive approximation (high bits first, then one refining bit per scan: Ah -> Al) -- into the same coefficient planes the
// baseline decoder fills; after the last scan the planes hold exactly the coefficients of the equivalent baseline file, and
// everything behind (inverse DCT, upsampling, colour) is shared.  A scan is walked serially by one lane (end-of-band runs and
// correction bits make the position of every code depend on the coefficients decoded before it); the parallelism is across the
// images of a batch.  All four procedures follow G.1.2 / G.2 (decoding side: Figures G.3 - G.7 mirrored).
enum { MAX_SCANS = 24 };
struct ProgScan {
  int32_t ncomp, comp[MAX_COMPS];
  int32_t ss, se, ah, al;
  int32_t dc_tab[MAX_COMPS], ac_tab;     // indices into the image's table array (ProgDesc::tabs_off)
  int32_t ac_tab3[MAX_COMPS];            // full-band scans of a sequential file: an AC table per component of the scan
  int32_t restart_interval;              // MCUs of THIS scan between restart markers, 0 = none
  int32_t n_iv;
  uint64_t clean_off;                    // the scan's unstuffed data (4-byte aligned, 16 bytes of 0xFF behind it)
  uint32_t clean_len, pad;
  uint64_t iv_off;                       // uint32 iv_byte[n_iv + 1]
};
struct ProgDesc {
  int32_t n_scans, n_tabs;
  uint64_t tabs_off;                     // HuffTable[n_tabs]
  ProgScan scans[MAX_SCANS];
};

// (reads stop at `limit` = a few bytes behind the scan's end, inside its padding: behind it every bit reads as 1, so a damaged scan
// can run on -- it is refused at the end of the block -- but never leaves its buffer)
struct PBits { const uint8_t* clean; uint32_t bit, limit; };
JPG_HD uint32_t pb_peek(const PBits& b) { return b.bit < b.limit ? peek32(b.clean, b.bit) : 0xffffffffu; }
JPG_HD int pb_get(PBits& b, int n) {                          // n = 0 .. 16
  if (n == 0) return 0;
  const uint32_t w = pb_peek(b);
  b.bit += (uint32_t)n;
  return (int)(w >> (32 - n));
}
JPG_HD int pb_huff(PBits& b, const HuffTable& t) {            // one symbol, -1 = no such code
  const int c16 = (int)(pb_peek(b) >> 16);
  const int e = t.look[c16 >> 7];
  if (e) { b.bit += (uint32_t)(e >> 8); return e & 0xff; }
  int len = 10;
  while (len <= 16 && (c16 >> (16 - len)) > t.maxcode[len]) ++len;
  if (len > 16) return -1;
  b.bit += (uint32_t)len;
  return t.huffval[((c16 >> (16 - len)) + t.valoffset[len]) & 0xff];
}

// a whole block of a SEQUENTIAL scan (a sequential file whose components come in several scans): DC difference, then AC run / size
// pairs up to EOB -- what decode_block does on the stuffed stream
JPG_HD bool seq_block(PBits& b, const HuffTable& dc, const HuffTable& ac, const uint8_t* zz, int& pred, int16_t* blk) {
  int s = pb_huff(b, dc);
  if (s < 0 || s > 11) return false;
  if (s) pred += extend(pb_get(b, s), s);
  blk[0] = (int16_t)pred;
  for (int k = 1; k < 64;) {
    const int rs = pb_huff(b, ac);
    if (rs < 0) return false;
    const int r = rs >> 4;
    s = rs & 15;
    if (s == 0) {
      if (r != 15) break;
      k += 16;
      continue;
    }
    k += r;
    if (k > 63) return false;
    blk[zz[k]] = (int16_t)extend(pb_get(b, s), s);
    ++k;
  }
  return true;
}

JPG_HD bool prog_dc_first(PBits& b, const HuffTable& t, int& pred, int16_t* blk, int al) {
  const int s = pb_huff(b, t);
  if (s < 0 || s > 11) return false;
  if (s) pred += extend(pb_get(b, s), s);
  blk[0] = (int16_t)((uint32_t)pred << al);
  return true;
}
JPG_HD void prog_dc_refine(PBits& b, int16_t* blk, int al) {
  if (pb_get(b, 1)) blk[0] = (int16_t)(blk[0] | (1 << al));
}
JPG_HD bool prog_ac_first(PBits& b, const HuffTable& t, const uint8_t* zz, int16_t* blk, int ss, int se, int al, int& eobrun) {
  if (eobrun > 0) { --eobrun; return true; }
  for (int k = ss; k <= se; ++k) {
    const int rs = pb_huff(b, t);
    if (rs < 0) return false;
    const int r = rs >> 4, s = rs & 15;
    if (s) {
      k += r;
      if (k > se) return false;
      blk[zz[k]] = (int16_t)((uint32_t)extend(pb_get(b, s), s) << al);
    } else if (r == 15) {
      k += 15;                                                   // ZRL
    } else {
      eobrun = 1 << r;                                           // EOBn: this band and the bands of the next eobrun - 1 blocks are done
      if (r) eobrun += pb_get(b, r);
      --eobrun;
      break;
    }
  }
  return true;
}
JPG_HD bool prog_ac_refine(PBits& b, const HuffTable& t, const uint8_t* zz, int16_t* blk, int ss, int se, int al, int& eobrun) {
  const int p1 = 1 << al, m1 = -(1 << al);
  int k = ss;
  if (eobrun == 0) {
    for (; k <= se; ++k) {
      const int rs = pb_huff(b, t);
      if (rs < 0) return false;
      int r = rs >> 4, s = rs & 15;
      if (s) {
        if (s != 1) return false;                                // a newly nonzero coefficient is +-1 at this bit
        s = pb_get(b, 1) ? p1 : m1;
      } else if (r != 15) {
        eobrun = 1 << r;
        if (r) eobrun += pb_get(b, r);
        break;                                                   // the rest of the band: correction bits only (below)
      }
      // pass the coefficients that are already nonzero (each takes a correction bit) and r zero-history ones
      do {
        int16_t& c = blk[zz[k]];
        if (c != 0) {
          if (pb_get(b, 1) && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
        } else if (--r < 0) {
          break;
        }
        ++k;
      } while (k <= se);
      if (s) {
        if (k > se) return false;
        blk[zz[k]] = (int16_t)s;
      }
    }
  }
  if (eobrun > 0) {
    for (; k <= se; ++k) {
      int16_t& c = blk[zz[k]];
      if (c != 0 && pb_get(b, 1) && (c & p1) == 0) c = (int16_t)(c + (c >= 0 ? p1 : m1));
    }
    --eobrun;
  }
  return true;
}

// one scan, serially.  0 = ok, 1 = invalid code / index, 2 = the data ran out
JPG_HD int prog_decode_scan(const ImageDesc& d, const ProgScan& sc, const uint8_t* clean, const uint32_t* iv_byte, const HuffTable* tabs,
                            const uint8_t* zz, int16_t* const coef[MAX_COMPS]) {
  PBits b{clean, 0, sc.clean_len * 8u + 64u};
  int pred[MAX_COMPS] = {0, 0, 0};
  int eobrun = 0;
  const bool single = sc.ncomp == 1;
  const int c0 = sc.comp[0];
  // a single-component scan walks the component's REAL blocks in raster order; an interleaved one walks MCUs
  const int nx = single ? (d.dw[c0] + 7) / 8 : d.mcus_x, ny = single ? (d.dh[c0] + 7) / 8 : d.mcus_y;
  int to_restart = sc.restart_interval, iv = 0;
  for (int my = 0; my < ny; ++my)
    for (int mx = 0; mx < nx; ++mx) {
      if (sc.restart_interval) {
        if (to_restart == 0) {
          if (b.bit > iv_byte[iv + 1] * 8u) return 2;
          ++iv;
          if (iv >= sc.n_iv) return 1;
          b.bit = iv_byte[iv] * 8u;
          pred[0] = pred[1] = pred[2] = 0; eobrun = 0;
          to_restart = sc.restart_interval;
        }
        --to_restart;
      }
      for (int ci = 0; ci < sc.ncomp; ++ci) {
        const int c = sc.comp[ci];
        const int h = single ? 1 : d.hs[c], v = single ? 1 : d.vs[c];
        for (int by = 0; by < v; ++by)
          for (int bx = 0; bx < h; ++bx) {
            int16_t* blk = coef[c] + ((size_t)(my * v + by) * d.bw[c] + (mx * h + bx)) * 64;
            bool ok = true;
            if (sc.ss == 0 && sc.se == 63) {
              ok = seq_block(b, tabs[sc.dc_tab[ci]], tabs[sc.ac_tab3[ci]], zz, pred[ci], blk);
            } else if (sc.ss == 0) {
              if (sc.ah == 0) ok = prog_dc_first(b, tabs[sc.dc_tab[ci]], pred[ci], blk, sc.al);
              else prog_dc_refine(b, blk, sc.al);
            } else if (sc.ah == 0) {
              ok = prog_ac_first(b, tabs[sc.ac_tab], zz, blk, sc.ss, sc.se, sc.al, eobrun);
            } else {
              ok = prog_ac_refine(b, tabs[sc.ac_tab], zz, blk, sc.ss, sc.se, sc.al, eobrun);
            }
            if (!ok) return 1;
            if (b.bit > b.limit) return 2;                        // (behind the end: refused here, see PBits)
          }
      }
    }
  return b.bit > sc.clean_len * 8u ? 2 : 0;
}

// ------------------------------------------------------------------------------------------------ inverse DCT ("islow")
// Dequantise + 8 x 8 inverse DCT, LL&M with CONST_BITS = 13, PASS1_BITS = 2: columns first into a workspace scaled by 4, then
// rows, then centre and clamp.  Written the way libjpeg-turbo's SIMD routine computes it (that is what Pillow executes): the
// dequantised coefficient is a 16-bit product (wraps), sums and products are 32-bit (wrap), the workspace saturates to 16 bits,
// the result saturates to a signed byte before the centre is added.  On valid data none of this ever triggers and the result is
// the textbook islow transform; on damaged data it keeps the pixels equal to Pillow's, and all of it is defined arithmetic
// (unsigned wrap-around, no signed overflow).
JPG_HD int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
JPG_HD int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
JPG_HD int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
JPG_HD int32_t sat16(int32_t x) { return x < -32768 ? -32768 : x > 32767 ? 32767 : x; }

JPG_HD void idct_1d(const int32_t in[8], int32_t out[8]) {     // outputs NOT descaled: out[0..7] = the eight sums
  constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
                F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069, F_2_053119869 = 16819,
                F_2_562915447 = 20995, F_3_072711026 = 25172;
  int32_t z2 = in[2], z3 = in[6];
  int32_t z1 = wmul(wadd(z2, z3), F_0_541196100);
  int32_t tmp2 = wadd(z1, wmul(z3, -F_1_847759065));
  int32_t tmp3 = wadd(z1, wmul(z2, F_0_765366865));
  z2 = in[0]; z3 = in[4];
  int32_t tmp0 = (int32_t)((uint32_t)wadd(z2, z3) << 13);
  int32_t tmp1 = (int32_t)((uint32_t)wsub(z2, z3) << 13);
  const int32_t tmp10 = wadd(tmp0, tmp3), tmp13 = wsub(tmp0, tmp3), tmp11 = wadd(tmp1, tmp2), tmp12 = wsub(tmp1, tmp2);
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = wadd(tmp0, tmp3); z2 = wadd(tmp1, tmp2); z3 = wadd(tmp0, tmp2);
  int32_t z4 = wadd(tmp1, tmp3);
  const int32_t z5 = wmul(wadd(z3, z4), F_1_175875602);
  tmp0 = wmul(tmp0, F_0_298631336); tmp1 = wmul(tmp1, F_2_053119869); tmp2 = wmul(tmp2, F_3_072711026); tmp3 = wmul(tmp3, F_1_501321110);
  z1 = wmul(z1, -F_0_899976223); z2 = wmul(z2, -F_2_562915447); z3 = wmul(z3, -F_1_961570560); z4 = wmul(z4, -F_0_390180644);
  z3 = wadd(z3, z5); z4 = wadd(z4, z5);
  tmp0 = wadd(tmp0, wadd(z1, z3)); tmp1 = wadd(tmp1, wadd(z2, z4)); tmp2 = wadd(tmp2, wadd(z2, z3)); tmp3 = wadd(tmp3, wadd(z1, z4));
  out[0] = wadd(tmp10, tmp3); out[7] = wsub(tmp10, tmp3);
  out[1] = wadd(tmp11, tmp2); out[6] = wsub(tmp11, tmp2);
  out[2] = wadd(tmp12, tmp1); out[5] = wsub(tmp12, tmp1);
  out[3] = wadd(tmp13, tmp0); out[4] = wsub(tmp13, tmp0);
}

JPG_HD int32_t descale(int32_t x, int n) { return wadd(x, 1 << (n - 1)) >> n; }

// coef: 64 int16 (natural order), quant: 64 uint16 (natural order); out: 8 rows, `pitch` bytes apart
JPG_HD void idct_block(const int16_t* coef, const uint16_t* quant, uint8_t* out, int pitch) {
  int32_t ws[64];
  for (int c = 0; c < 8; ++c) {
    int32_t in[8], o[8];
    for (int r = 0; r < 8; ++r) in[r] = (int16_t)(uint16_t)((uint32_t)(int32_t)coef[r * 8 + c] * (uint32_t)quant[r * 8 + c]);
    idct_1d(in, o);
    for (int r = 0; r < 8; ++r) ws[r * 8 + c] = sat16(descale(o[r], 13 - 2));
  }
  for (int r = 0; r < 8; ++r) {
    int32_t o[8];
    idct_1d(ws + r * 8, o);
    for (int c = 0; c < 8; ++c) {
      const int32_t v = sat16(descale(o[c], 13 + 2 + 3));
      out[r * pitch + c] = (uint8_t)((v < -128 ? -128 : v > 127 ? 127 : v) + 128);
    }
  }
}

// ------------------------------------------------------------------------------------------------ upsampling + colour
// Sample (x, y) of a component at full resolution.  p: the component's plane (pitch bytes per row), dw x dh its real size,
// (h, v) = (hmax / hs, vmax / vs) its expansion.  As libjpeg chooses its upsampler per component: 1 x 1 is a copy; 2 x 1, 2 x 2
// and 1 x 2 use triangle filters ("fancy upsampling": 3/4 + 1/4 in one direction with alternating rounding, 9/16 + 3/16 + 3/16 +
// 1/16 for 2 x 2; the rows above the first and below the last real row are those rows again) -- the two horizontal ones only when
// the plane is more than 2 samples wide; every other integral expansion replicates samples.
JPG_HD int upsampled(const uint8_t* p, int pitch, int dw, int dh, int h, int v, int x, int y) {
  if (h == 1 && v == 1) return p[(size_t)y * pitch + x];
  if (h == 2 && v == 1 && dw > 2) {
    const uint8_t* row = p + (size_t)y * pitch;
    const int i = x >> 1;
    if (x & 1) return i == dw - 1 ? row[i] : (row[i] * 3 + row[i + 1] + 2) >> 2;
    return i == 0 ? row[0] : (row[i] * 3 + row[i - 1] + 1) >> 2;
  }
  if (h == 2 && v == 2 && dw > 2) {
    const int i = x >> 1, r = y >> 1;
    int rn = (y & 1) ? r + 1 : r - 1;                          // the nearer neighbouring row
    rn = rn < 0 ? 0 : rn > dh - 1 ? dh - 1 : rn;
    const uint8_t* r0 = p + (size_t)r * pitch;
    const uint8_t* r1 = p + (size_t)rn * pitch;
    const int cur = r0[i] * 3 + r1[i];
    if (x & 1) return i == dw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + (r0[i + 1] * 3 + r1[i + 1]) + 7) >> 4;
    return i == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + (r0[i - 1] * 3 + r1[i - 1]) + 8) >> 4;
  }
  if (h == 1 && v == 2) {
    const int r = y >> 1;
    int rn = (y & 1) ? r + 1 : r - 1;
    rn = rn < 0 ? 0 : rn > dh - 1 ? dh - 1 : rn;
    return (p[(size_t)r * pitch + x] * 3 + p[(size_t)rn * pitch + x] + ((y & 1) ? 2 : 1)) >> 2;
  }
  return p[(size_t)(y / v) * pitch + x / h];                   // integral replication
}

JPG_HD uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

// JFIF YCbCr -> RGB in libjpeg's 16-bit fixed point (1.40200, 0.34414, 0.71414, 1.77200; the two green terms share one rounding)
JPG_HD void ycc_to_rgb(int y, int cb, int cr, uint8_t* rgb) {
  const int xb = cb - 128, xr = cr - 128;
  rgb[0] = clamp255(y + ((91881 * xr + 32768) >> 16));
  rgb[1] = clamp255(y + ((-22554 * xb + 32768 - 46802 * xr) >> 16));
  rgb[2] = clamp255(y + ((116130 * xb + 32768) >> 16));
}

}  // namespace jpg
