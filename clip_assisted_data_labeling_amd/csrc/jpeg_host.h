// Host side of the JPEG decoder: marker parsing and table preparation (no device code; also compiled into the CPU checker
// the tests' jpeg_ref.cpp).  Supported: baseline / extended-sequential Huffman JPEG (one interleaved scan, or the
// components in several full-band scans) and progressive Huffman JPEG, 8 bits, greyscale or YCbCr with any sampling libjpeg upsamples (4:4:4, 4:2:2, 4:2:0, 4:4:0, 4:1:1, ...),
// restart intervals.  Everything else is reported with a reason code and
// left to the caller (the embed driver hands such files to Pillow).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "jpeg_core.h"

namespace jpg {

enum {
  JPG_OK = 0, JPG_NOT_JPEG = 1, JPG_PROGRESSIVE = 2, JPG_PRECISION = 3, JPG_COMPONENTS = 4, JPG_SAMPLING = 5, JPG_MULTI_SCAN = 6,
  JPG_COLORSPACE = 7, JPG_TABLES = 8, JPG_ARITHMETIC = 9, JPG_TOO_LARGE = 10, JPG_TRUNCATED = 11, JPG_CORRUPT = 12
};

// A progressive file's scans as the parser found them: parameters, Huffman tables in force (indices into `tabs`), and the byte
// range of each scan's entropy-coded data in the file.
struct ProgScanInfo { ProgScan s; size_t begin, end; };
struct ProgInfo {
  std::vector<HuffTable> tabs;
  std::vector<ProgScanInfo> scans;
};

// Fills every field of *d that does not depend on the batch layout (sizes, sampling, tables); *scan_off / *scan_len = the
// entropy-coded data (from behind the SOS header to the end of the file).  Returns JPG_OK or the reason the file is not decodable here.
// A progressive file (SOF2) is taken only when `prog` is given: its scans go there (*scan_off / *scan_len are 0 then), after a
// check that the scan script is a complete, orderly progression (every coefficient of every component first coded with Ah = 0,
// then refined bit by bit down to 0) -- what Pillow decodes without its "block smoothing" of unfinished progressions.
int parse_jpeg(const uint8_t* data, size_t len, ImageDesc* d, size_t* scan_off, size_t* scan_len, ProgInfo* prog = nullptr);
// T.81 Annex C + F.2.2.3: code lengths -> lookahead / maxcode / valoffset; false if the lengths do not describe a prefix code
bool build_huff(const uint8_t counts[16], const uint8_t* vals, int nvals, HuffTable* t);
const char* reason_text(int code);

// The entropy-coded segment without its byte stuffing and markers, for the parallel decoder: `clean` receives the data bytes
// (0xFF 0x00 -> 0xFF, fill bytes and RSTn markers dropped), `interval_start` the byte offset in `clean` at which each restart
// interval begins (first entry 0).  Returns the number of clean bytes, or -1 if the data does not end in an EOI marker (a
// truncated file) or holds a marker that is neither RSTn in sequence nor EOI.  `clean` must have room for len + 16 bytes; the 16
// bytes behind the data are set to 0xFF (what the encoder pads with).
// require_eoi = false: the given range IS the scan (a progressive file's scans end at the next marker, found by the parser).
long unstuff_scan(const uint8_t* scan, size_t len, uint8_t* clean, uint32_t* interval_start, int max_intervals, int* n_intervals,
                  bool require_eoi = true);

}  // namespace jpg
