"""GPU JPEG decoder (jpeg_decode.hip) against Pillow, the decoder behind /root/reference/utils/embedder.py:167
(`Image.open(path).convert('RGB')`): integer arithmetic on both sides, so the bar is bit-exact."""
import io

import numpy as np
import pytest
import torch
from PIL import Image, ImageFile

from clip_assisted_data_labeling_amd.jpeg_gpu import GpuJpegDecoder

pytestmark = pytest.mark.gpu
ImageFile.MAXBLOCK = 1 << 24


def _smooth(rs, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 17.0 + yy / 29.0), 128 + 90 * np.cos(xx / 11.0 - yy / 23.0), 128 + 80 * np.sin((xx + yy) / 7.0)], -1)
    return np.clip(img + rs.randn(h, w, 3) * 12, 0, 255).astype(np.uint8)


def _jpeg(arr, **kw):
    b = io.BytesIO()
    (arr if isinstance(arr, Image.Image) else Image.fromarray(arr)).save(b, "JPEG", **kw)
    return b.getvalue()


def _pil(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def test_decoder_matches_pillow_bit_for_bit_over_sizes_samplings_qualities(gpu):
    rs = np.random.RandomState(0)
    files = []
    for (h, w) in [(8, 8), (16, 16), (37, 53), (64, 48), (1, 1), (2, 3), (5, 2), (17, 1), (3, 5), (100, 133), (241, 319), (480, 640)]:
        for ss in (0, 1, 2):
            for q in (30, 90, 100):
                files.append(_jpeg(_smooth(rs, h, w), quality=q, subsampling=ss, optimize=bool((h + q) & 1)))
    noise = rs.randint(0, 256, (123, 211, 3), dtype=np.uint8)
    files += [_jpeg(noise, quality=q, subsampling=ss) for q in (5, 50, 100) for ss in (0, 1, 2)]
    files += [_jpeg(rs.randint(0, 256, (77, 91), dtype=np.uint8), quality=80)]                       # greyscale
    files += [_jpeg(noise, quality=85, subsampling=ss, **kw) for ss in (0, 2)
              for kw in ({"restart_marker_blocks": 1}, {"restart_marker_blocks": 5}, {"restart_marker_rows": 3})]
    ext = np.zeros((40, 40, 3), np.uint8)
    ext[:20, :20] = 255; ext[20:, :20] = (255, 0, 0); ext[:20, 20:] = (0, 0, 255)
    files += [_jpeg(ext, quality=100, subsampling=2), _jpeg(ext, quality=10, subsampling=2)]
    dec = GpuJpegDecoder(gpu)
    images, status = dec.decode(files)
    assert status == [0] * len(files)
    for i, (img, data) in enumerate(zip(images, files)):
        ref = _pil(data)
        assert tuple(img.shape) == ref.shape, i
        assert np.array_equal(img.cpu().numpy(), ref), f"file {i} ({ref.shape}) differs from Pillow"
    # a second batch through the same handle (buffers are reused, the arena may grow); then the same files under a pixel budget that
    # splits the call into several device batches
    big = [_jpeg(rs.randint(0, 256, (512, 512, 3), dtype=np.uint8), quality=90) for _ in range(5)] + [_jpeg(_smooth(rs, 1200, 1600), quality=92)]
    for budget in (2_000_000_000, 600_000):
        images, status = dec.decode(big, max_batch_pixels=budget)
        assert status == [0] * len(big)
        for img, data in zip(images, big):
            assert np.array_equal(img.cpu().numpy(), _pil(data))
    dec.close()


def test_decoder_reports_what_it_does_not_take_and_corrupt_data(gpu):
    rs = np.random.RandomState(1)
    noise = Image.fromarray(rs.randint(0, 256, (64, 80, 3), dtype=np.uint8))
    good = _jpeg(noise, quality=80)
    prog = _jpeg(noise, quality=80, progressive=True)
    files = [good, prog[: len(prog) // 2], _jpeg(noise.convert("CMYK"), quality=80), b"\x89PNG\r\n\x1a\n" + b"0" * 64,
             _jpeg(noise, quality=80, keep_rgb=True), good[: len(good) // 2], b"", good]
    dec = GpuJpegDecoder(gpu)
    images, status = dec.decode(files)
    assert status[0] == 0 and status[7] == 0
    assert status[1] in (11, 2) and status[2] == 4 and status[3] == 1 and status[4] == 7 and status[6] == 1   # [1]: a truncated progressive file
    assert "progressive" in dec.reason(2) and "colour" in dec.reason(7)
    # half a file: the header parses, the entropy data runs out -> flagged by the device (or, if the cut fell inside the header, by the parser)
    assert status[5] >= 100 or status[5] == 11
    assert all((im is None) == (s != 0) for im, s in zip(images, status))
    ref = _pil(good)
    assert np.array_equal(images[0].cpu().numpy(), ref) and np.array_equal(images[7].cpu().numpy(), ref)
    dec.close()


def test_decoded_images_feed_the_gpu_front_end_like_pillow_decoded_ones(gpu):
    """decode -> crop geometry + bicubic resize on the device: the same crops as from Pillow-decoded arrays."""
    from clip_assisted_data_labeling_amd.preprocess import GpuCropper
    rs = np.random.RandomState(2)
    files = [_jpeg(_smooth(rs, h, w), quality=88, subsampling=ss) for (h, w, ss) in [(300, 400, 2), (512, 384, 1), (257, 257, 0)]]
    dec = GpuJpegDecoder(gpu)
    images, status = dec.decode(files)
    assert status == [0, 0, 0]
    cropper = GpuCropper(224, gpu)
    a, names_a = cropper.batch(images)
    b, names_b = cropper.batch([torch.from_numpy(_pil(f).copy()) for f in files])
    assert names_a == names_b and torch.equal(a, b)
    cropper.close(); dec.close()


def test_damaged_files_on_the_device_equal_the_cpu_run_of_the_same_arithmetic(gpu):
    """Mutated files (random bytes, 0xFF insertions, truncations): the device must survive every one of them, refuse the same
    files as the CPU run of the same sources (oracle/jpeg_ref.cpp: the parallel entropy decoder with its threads in sequence)
    and produce the same pixels for the rest."""
    from oracle import jpeg_oracle
    rs = np.random.RandomState(11)
    seeds = [_jpeg(rs.randint(0, 256, (h, w, 3), dtype=np.uint8), quality=85, subsampling=ss, **kw)
             for (h, w, ss, kw) in [(64, 80, 2, {}), (33, 47, 1, {"optimize": True}), (40, 40, 0, {"restart_marker_blocks": 3}), (24, 24, 2, {}),
                                    (48, 56, 2, {"progressive": True}), (40, 40, 1, {"progressive": True, "restart_marker_blocks": 2})]]
    files = []
    for s in seeds:
        for t in range(100):
            a = bytearray(s)
            for _ in range(rs.randint(1, 6)):
                mode = t % 4
                pos = rs.randint(0, min(700, len(a))) if mode == 1 else rs.randint(min(600, len(a) - 1), len(a)) if mode == 2 else rs.randint(0, len(a))
                a[pos] = 0xFF if mode == 3 else rs.randint(0, 256)
            if t % 7 == 0:
                a = a[: rs.randint(2, len(a))]
            files.append(bytes(a))
    # restart intervals that come out too short / too long / renumbered / missing: the writing pass must stop at the interval's end
    for kw in ({"restart_marker_blocks": 1}, {"restart_marker_rows": 1}, {"restart_marker_blocks": 7}):
        s = _jpeg(rs.randint(0, 256, (64, 80, 3), dtype=np.uint8), quality=85, **kw)
        sos = s.index(b"\xff\xda")
        for t in range(60):
            a = bytearray(s)
            p = rs.randint(sos + 14, len(a) - 4)
            if t % 3 == 0:
                del a[p:p + rs.randint(1, 200)]
            elif t % 3 == 1:
                a[p:p] = a[p:p + rs.randint(1, 200)]
            else:
                idx = [i for i in range(sos, len(a) - 1) if a[i] == 0xFF and 0xD0 <= a[i + 1] <= 0xD7]
                q = idx[rs.randint(0, len(idx))]
                if t % 2:
                    del a[q:q + 2]
                else:
                    a[q + 1] = 0xD0 + rs.randint(0, 8)
            files.append(bytes(a))
    dec = GpuJpegDecoder(gpu)
    images, status = dec.decode(files)
    n_ok = 0
    for data, img, st in zip(files, images, status):
        try:
            ref = jpeg_oracle.decode_parallel(data, 2048, 1 << 20)[0]
        except ValueError:
            ref = None
        assert (ref is None) == (st != 0), (st, len(data))
        if ref is not None:
            n_ok += 1
            assert np.array_equal(img.cpu().numpy(), ref)
    assert n_ok >= 30
    # the handle is still good
    good = _jpeg(rs.randint(0, 256, (50, 60, 3), dtype=np.uint8), quality=90)
    images, status = dec.decode([good])
    assert status == [0] and np.array_equal(images[0].cpu().numpy(), _pil(good))
    dec.close()


def test_progressive_files_on_the_device_equal_pillow(gpu):
    """Progressive files mixed with baseline ones in one call (the progressive ones take the serial-per-image kernel)."""
    rs = np.random.RandomState(21)
    files = []
    for (h, w) in [(8, 8), (37, 53), (1, 1), (5, 2), (100, 133), (241, 319), (480, 640)]:
        for ss in (0, 1, 2):
            for q in (10, 90, 100):
                kw = [{}, {"optimize": True}, {"restart_marker_blocks": 3}, {"restart_marker_rows": 1}][(h + ss + q) % 4]
                files.append(_jpeg(_smooth(rs, h, w), quality=q, subsampling=ss, progressive=True, **kw))
                if q == 90:
                    files.append(_jpeg(_smooth(rs, h, w), quality=q, subsampling=ss))          # a baseline file in between
    noise = rs.randint(0, 256, (123, 211, 3), dtype=np.uint8)
    files += [_jpeg(noise, quality=q, progressive=True) for q in (5, 50, 100)]
    files += [_jpeg(rs.randint(0, 256, (77, 91), dtype=np.uint8), quality=80, progressive=True)]
    dec = GpuJpegDecoder(gpu)
    assert all(dec.takes(f) for f in files)
    images, status = dec.decode(files)
    assert status == [0] * len(files)
    for i, (img, data) in enumerate(zip(images, files)):
        assert np.array_equal(img.cpu().numpy(), _pil(data)), i
    dec.close()


def test_any_sampling_on_the_device_equals_pillow(gpu):
    """4:4:0, 4:1:1, ten-block MCUs, unequal chroma planes, luma below chroma (tests/jpeg_writer.py), restart intervals."""
    from tests.jpeg_writer import random_coefs, tables_from_pillow, write_baseline
    dqt, dht = tables_from_pillow(80)
    rs = np.random.RandomState(1)
    files = []
    for samp in ([(1, 2), (1, 1), (1, 1)], [(4, 1), (1, 1), (1, 1)], [(1, 4), (1, 1), (1, 1)], [(4, 2), (1, 1), (1, 1)], [(2, 4), (1, 1), (1, 1)],
                 [(2, 2), (2, 1), (1, 1)], [(2, 2), (1, 2), (2, 1)], [(1, 1), (2, 2), (2, 2)], [(2, 1), (1, 2), (1, 1)], [(2, 2)]):
        for (w, h) in [(8, 8), (33, 47), (100, 37), (3, 2), (257, 130)]:
            for restart in (0, 3):
                files.append(write_baseline(w, h, samp, random_coefs(rs, w, h, samp), dqt, dht, restart=restart))
    from tests.jpeg_writer import write_progressive
    for samp in ([(1, 2), (1, 1), (1, 1)], [(4, 1), (1, 1), (1, 1)], [(2, 2), (1, 2), (2, 1)], [(2, 2)]):          # ... and progressive ones
        for (w, h) in [(33, 47), (100, 37)]:
            files.append(write_progressive(w, h, samp, random_coefs(rs, w, h, samp), dqt, dht, restart=2 * (w > 50)))
    from tests.jpeg_writer import write_sequential_scans
    for samp in ([(2, 2), (1, 1), (1, 1)], [(1, 2), (1, 1), (1, 1)]):                                                # ... sequential files in several scans
        for scans in ([[0], [1], [2]], [[0], [1, 2]]):
            files.append(write_sequential_scans(100, 37, samp, random_coefs(rs, 100, 37, samp), dqt, dht, scans, restart=3))
    dec = GpuJpegDecoder(gpu)
    images, status = dec.decode(files)
    assert status == [0] * len(files)
    for i, (img, data) in enumerate(zip(images, files)):
        assert np.array_equal(img.cpu().numpy(), _pil(data)), i
    dec.close()


def test_a_call_is_split_by_file_count_and_by_pixels(gpu):
    """GpuJpegDecoder.decode groups the files of a call (at most max_batch_files and max_batch_pixels per device batch; the C entry
    point takes 65535 files): the result does not depend on the grouping."""
    import ctypes
    from clip_assisted_data_labeling_amd import _lib
    rs = np.random.RandomState(3)
    files = [_jpeg(_smooth(rs, 40 + 7 * i, 50 + 5 * i), quality=85, subsampling=i % 3) for i in range(9)] + [b"not a jpeg"]
    dec = GpuJpegDecoder(gpu)
    whole, st0 = dec.decode(files)
    for kw in ({"max_batch_files": 3}, {"max_batch_files": 1}, {"max_batch_pixels": 5000}):
        parts, st1 = dec.decode(files, **kw)
        assert st1 == st0 and st0[:9] == [0] * 9 and st0[9] == 1
        for a, b in zip(whole, parts):
            assert (a is None and b is None) or torch.equal(a, b)
    for im, f in zip(whole[:9], files):
        assert np.array_equal(im.cpu().numpy(), _pil(f))
    n = 65536                                                  # one file too many for a single plan: refused with a message
    arr = (ctypes.c_char_p * n)(); sizes = (ctypes.c_size_t * n)(); ints = (ctypes.c_int * n)(); offs = (ctypes.c_ulonglong * n)()
    total = ctypes.c_ulonglong()
    assert dec.lib.jpegdec_plan(dec.handle, arr, sizes, n, ints, ints, ints, offs, ctypes.byref(total)) != 0
    assert b"65535" in _lib.load().clipenc_last_error()
    dec.close()
