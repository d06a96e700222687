"""JPEG files -> uint8 RGB tensors on the GPU (libclipenc_hip.so: jpegdec_*), the decode step of the reference's image loader.

/root/reference/utils/embedder.py:167 opens every image with `PIL.Image.open(path).convert('RGB')` inside DataLoader workers;
on real data that host-side decode is what bounds the embed driver.  `GpuJpegDecoder.decode` takes the file BYTES of a batch,
parses the headers on the host and decodes on the device with Pillow's (libjpeg-turbo's default) integer arithmetic, so the
pixels are identical to Pillow's.  Files it does not take (CMYK, 4:4:0, arithmetic coding, ... — `reason`) come back as None
and the caller decodes them with Pillow.  Sequential files are decoded in parallel inside each file; progressive files are taken
too, one lane per image (serial scans).  No CPU fallback inside: without the HIP library this module raises.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib


def scratch_bytes_per_pixel(data: bytes) -> float:
    """Device scratch one decoded pixel of this file needs -- int16 coefficients + one sample byte per component sample:
    3 x sum_c(h_c v_c) / (h_max v_max) -- from the frame header's sampling factors (host only, a few hundred bytes walked):
    4:2:0 -> 4.5, 4:2:2 -> 6, 4:4:4 -> 9, grey -> 3.  9.0 when the header cannot be read."""
    try:
        i, n = 2, len(data)
        while i + 4 <= n:
            if data[i] != 0xFF:
                i += 1
                continue
            m = data[i + 1]
            if m == 0xFF:                                           # fill byte
                i += 1
                continue
            if m == 0xD8 or m == 0x01 or 0xD0 <= m <= 0xD7:          # markers without a length
                i += 2
                continue
            seg = (data[i + 2] << 8) | data[i + 3]
            if m in (0xC0, 0xC1, 0xC2):
                nc = data[i + 9]
                hv = [(data[i + 11 + 3 * c] >> 4, data[i + 11 + 3 * c] & 15) for c in range(nc)]
                hmax, vmax = max(h for h, _ in hv), max(v for _, v in hv)
                return 3.0 * sum(h * v for h, v in hv) / float(hmax * vmax)
            if m == 0xDA or m == 0xD9:
                break
            i += 2 + seg
    except Exception:
        pass
    return 9.0


class GpuJpegDecoder:
    def __init__(self, device="cuda"):
        self.lib = _lib.load()
        d = torch.device(device)
        self.device = torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())
        h = ctypes.c_void_p()
        _lib.check(self.lib.jpegdec_create(self.device.index, ctypes.byref(h)), "jpegdec_create")
        self.handle = h

    def takes(self, data: bytes, progressive: bool = True) -> bool:
        """host only, thread-safe: would `decode` take this file?  progressive=False: only sequential files (the parallel
        entropy kernel); a progressive file is walked by one lane per image, scan after scan."""
        n = ctypes.c_int(0)
        if self.lib.jpegdec_probe(data, len(data), None, None, ctypes.byref(n)) != 0:
            return False
        return progressive or n.value == 1

    def probe(self, data: bytes):
        """host only, thread-safe: (taken, width, height, n_scans) from the file's headers (0 x 0 when it is not a JPEG the decoder
        can read at all); what the driver's reader threads cut decode chunks by."""
        n, w, h = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        rc = self.lib.jpegdec_probe(data, len(data), ctypes.byref(w), ctypes.byref(h), ctypes.byref(n))
        return rc == 0, int(w.value), int(h.value), int(n.value)

    NO_MEMORY = 77            # CLIPENC_JPEGDEC_NO_MEMORY (include/clipenc.h); as a per-file status: "decode it yourself"

    def reserve(self, pixels: int, file_bytes: int, staging: bool = False, bytes_per_pixel: float = 9.0) -> bool:
        """Sets aside device scratch for batches of up to `pixels` decoded pixels from `file_bytes` of files (`bytes_per_pixel`
        per pixel + the files' bytes: int16 coefficients + sample planes; `scratch_bytes_per_pixel(file)` gives a file's own figure
        from its sampling factors -- 4.5 for 4:2:0, 9 for 4:4:4, the default: an arena sized short regrows in the middle of a run,
        and hipFree + hipMalloc synchronise the device) and, with staging=True, the page-locked staging buffer (the files' bytes; page-locking takes
        ~0.2 s per GB, so by default it grows with the batches instead).  Call it before other work runs on the device: growing
        the device scratch later (hipFree + hipMalloc) synchronises the device.  False when the memory is not there."""
        rc = self.lib.jpegdec_reserve(self.handle, int(pixels * max(3.0, min(9.0, float(bytes_per_pixel)))) + int(file_bytes * 1.25) + (1 << 20),
                                      int(file_bytes * 1.25) + (1 << 20) if staging else 0)
        if rc == self.NO_MEMORY:
            return False
        _lib.check(rc, "jpegdec_reserve")
        return True

    def reason(self, code: int) -> str:
        if int(code) == self.NO_MEMORY:
            return "no device scratch for this batch"
        return self.lib.jpegdec_reason(int(code)).decode()

    @torch.no_grad()
    def decode(self, files: Sequence[bytes], max_batch_pixels: int = 2_000_000_000, max_batch_files: int = 16384) -> Tuple[List[Optional[torch.Tensor]], List[int]]:
        """files: the bytes of each file -> (images, status): images[i] a uint8 [H, W, 3] tensor on the GPU (a view into a
        batch buffer) or None when status[i] != 0 (1..12: not decodable here, see `reason`; 77: no device memory for its group;
        >= 100: corrupt or truncated entropy data).  The device works on all files of a call at once -- the entropy decoder is one serial stream per file, so its
        throughput IS the number of files in flight -- except that a call is split into groups of at most `max_batch_pixels`
        pixels (7.5 bytes of device scratch + output per pixel: 15 GB at the default) and `max_batch_files` files (the C entry point takes
        65535 per call)."""
        n = len(files)
        if n == 0:
            return [], []
        bufs = [bytes(f) if not isinstance(f, bytes) else f for f in files]
        images: List[Optional[torch.Tensor]] = [None] * n
        status_all = [0] * n
        start = 0
        while start < n:
            # headers of the rest (host only, microseconds per file), then as many files as fit the pixel budget
            m = min(n - start, max(1, min(int(max_batch_files), 65535)))
            ptrs = (ctypes.c_char_p * m)(*bufs[start:start + m])
            sizes = (ctypes.c_size_t * m)(*[len(b) for b in bufs[start:start + m]])
            status = (ctypes.c_int * m)()
            widths = (ctypes.c_int * m)()
            heights = (ctypes.c_int * m)()
            offsets = (ctypes.c_ulonglong * m)()
            total = ctypes.c_ulonglong()
            _lib.check(self.lib.jpegdec_plan(self.handle, ptrs, sizes, m, status, widths, heights, offsets, ctypes.byref(total)), "jpegdec_plan")
            take, pixels = 0, 0
            while take < m:
                px = widths[take] * heights[take] if status[take] == 0 else 0
                if take > 0 and pixels + px > max_batch_pixels:
                    break
                pixels += px
                take += 1
            if take < m:                                          # plan again for the group that fits
                _lib.check(self.lib.jpegdec_plan(self.handle, ptrs, sizes, take, status, widths, heights, offsets, ctypes.byref(total)),
                           "jpegdec_plan")
            try:
                rgb = torch.empty(max(int(total.value), 1), dtype=torch.uint8, device=self.device)
                rc = self.lib.jpegdec_run(self.handle, rgb.data_ptr(), status, _lib.current_stream_ptr(self.device))
            except torch.OutOfMemoryError:
                rc = self.NO_MEMORY
            if rc == self.NO_MEMORY:                              # no room for this group: its files go back to the caller undecoded
                for i in range(take):
                    status_all[start + i] = int(status[i]) or self.NO_MEMORY
                start += take
                continue
            _lib.check(rc, "jpegdec_run")
            for i in range(take):
                status_all[start + i] = int(status[i])
                if status[i] == 0:
                    h, w, o = heights[i], widths[i], int(offsets[i])
                    images[start + i] = rgb[o:o + h * w * 3].view(h, w, 3)
            start += take
        return images, status_all

    def close(self):
        if getattr(self, "handle", None):
            self.lib.jpegdec_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
