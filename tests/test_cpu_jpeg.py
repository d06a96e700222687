"""The JPEG decoding arithmetic of the HIP kernels (csrc/jpeg_core.h + jpeg_host.cpp), run on the CPU by the checker
oracle/jpeg_ref.cpp, against Pillow -- the decoder behind /root/reference/utils/embedder.py:167.  Integer algorithms on both
sides: every pixel must be equal.  (The device run of the same sources is tests/test_gpu_jpeg.py.)"""
import io
import struct

import numpy as np
import pytest
from PIL import Image, ImageFile

from oracle import jpeg_oracle

ImageFile.MAXBLOCK = 1 << 24


def _smooth(rs, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(xx / 17.0 + yy / 29.0), 128 + 90 * np.cos(xx / 11.0 - yy / 23.0), 128 + 80 * np.sin((xx + yy) / 7.0)], -1)
    return np.clip(img + rs.randn(h, w, 3) * 12, 0, 255).astype(np.uint8)


def _jpeg(img, **kw):
    b = io.BytesIO()
    (img if isinstance(img, Image.Image) else Image.fromarray(img)).save(b, "JPEG", **kw)
    return b.getvalue()


def _pil(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


@pytest.mark.parametrize("subsampling", [0, 1, 2])
def test_checker_equals_pillow_over_sizes_and_qualities(subsampling):
    rs = np.random.RandomState(subsampling)
    for (h, w) in [(8, 8), (16, 16), (37, 53), (1, 1), (2, 3), (5, 2), (17, 1), (3, 5), (4, 6), (100, 133), (241, 319)]:
        for q in (30, 75, 100):
            for opt in (False, True):
                data = _jpeg(_smooth(rs, h, w), quality=q, subsampling=subsampling, optimize=opt)
                assert np.array_equal(jpeg_oracle.decode(data), _pil(data)), (h, w, q, opt)


def test_checker_equals_pillow_on_noise_grey_restart_markers_and_saturated_colours():
    rs = np.random.RandomState(7)
    noise = rs.randint(0, 256, (123, 211, 3), dtype=np.uint8)
    files = [_jpeg(noise, quality=q, subsampling=ss) for q in (5, 50, 90, 100) for ss in (0, 1, 2)]
    files += [_jpeg(rs.randint(0, 256, (77, 91), dtype=np.uint8), quality=80), _jpeg(rs.randint(0, 256, (9, 200), dtype=np.uint8), quality=100, optimize=True)]
    files += [_jpeg(noise, quality=85, subsampling=ss, **kw) for ss in (0, 1, 2)
              for kw in ({"restart_marker_blocks": 1}, {"restart_marker_blocks": 5}, {"restart_marker_rows": 1}, {"restart_marker_rows": 3})]
    ext = np.zeros((40, 40, 3), np.uint8)
    ext[:20, :20] = 255; ext[20:, :20] = (255, 0, 0); ext[:20, 20:] = (0, 0, 255)
    files += [_jpeg(ext, quality=100, subsampling=2), _jpeg(ext, quality=10, subsampling=2), _jpeg(rs.randint(0, 256, (512, 512, 3), dtype=np.uint8), quality=90)]
    for i, data in enumerate(files):
        assert np.array_equal(jpeg_oracle.decode(data), _pil(data)), i


def test_parser_reasons_and_truncation_like_pillow():
    rs = np.random.RandomState(1)
    noise = Image.fromarray(rs.randint(0, 256, (64, 80, 3), dtype=np.uint8))
    good = _jpeg(noise, quality=80)
    assert jpeg_oracle.info(good) == (0, 80, 64, 3)
    assert jpeg_oracle.info(_jpeg(noise, quality=80, progressive=True))[0] == 0      # (progressive files are taken since round 3)
    assert jpeg_oracle.info(_jpeg(noise.convert("CMYK"), quality=80))[0] == 4
    assert jpeg_oracle.info(_jpeg(noise, quality=80, keep_rgb=True))[0] == 7
    assert jpeg_oracle.info(b"\x89PNG\r\n\x1a\n" + b"0" * 50)[0] == 1 and jpeg_oracle.info(b"")[0] == 1
    # Pillow refuses a file whose data or EOI marker is cut off and accepts bytes behind EOI; so does the decoder
    for cut in (len(good) // 2, len(good) - 10, len(good) - 2, len(good) - 1):
        with pytest.raises(ValueError):
            jpeg_oracle.decode(good[:cut])
        with pytest.raises(OSError):
            _pil(good[:cut])
    tail = good + b"abc\xff\x00xyz" * 5
    assert np.array_equal(jpeg_oracle.decode(tail), _pil(tail))


def test_damaged_files_are_refused_or_decoded_like_pillow():
    """600 mutated files (random bytes in headers / entropy data, 0xFF insertions, truncations).  The parser is stricter than
    libjpeg on purpose -- what it refuses goes to Pillow in the driver -- so the property to hold is one-sided: a file the
    decoder accepts is a file Pillow accepts too, and then (bar a few blocks of saturated garbage) with the same pixels."""
    import warnings
    rs = np.random.RandomState(5)
    seeds = [_jpeg(rs.randint(0, 256, (h, w, 3), dtype=np.uint8), quality=85, subsampling=ss, **kw)
             for (h, w, ss, kw) in [(64, 80, 2, {}), (33, 47, 1, {"optimize": True}), (40, 40, 0, {"restart_marker_blocks": 3}), (24, 24, 2, {}),
                                    (48, 56, 2, {"progressive": True}), (40, 40, 1, {"progressive": True, "restart_marker_blocks": 2})]]
    accepted = same = 0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for s in seeds:
            for t in range(150):
                a = bytearray(s)
                for _ in range(rs.randint(1, 6)):
                    mode = t % 4
                    pos = rs.randint(0, min(700, len(a))) if mode == 1 else rs.randint(min(600, len(a) - 1), len(a)) if mode == 2 else rs.randint(0, len(a))
                    a[pos] = 0xFF if mode == 3 else rs.randint(0, 256)
                if t % 7 == 0:
                    a = a[: rs.randint(2, len(a))]
                data = bytes(a)
                try:
                    got = jpeg_oracle.decode_parallel(data, 2048, 1 << 20)[0]      # the decoder the device runs
                except ValueError:
                    continue
                try:
                    assert np.array_equal(jpeg_oracle.decode(data), got)           # the serial walk, where it accepts the file too
                except ValueError:
                    pass
                accepted += 1
                ref = _pil(data)                                      # must not raise: accepted here => accepted by Pillow
                assert ref.shape == got.shape
                bad_blocks = {(y // 8, x // 8) for y, x in zip(*np.nonzero((ref != got).any(-1)))}
                same += not bad_blocks
                assert len(bad_blocks) <= 12, (len(data), bad_blocks)         # (blocks of saturated garbage: damaged TABLES reach several)
    assert accepted >= 50 and same >= 0.9 * accepted, (accepted, same)


@pytest.mark.parametrize("sub_bytes", [2048, 128, 16])
def test_parallel_entropy_decoder_equals_pillow(sub_bytes):
    """The subsequence-parallel entropy decoder (jpeg_core.h; its threads run one after the other in the checker): whatever the
    subsequence length, however many synchronisation passes it takes, the pixels are Pillow's.  Includes streams that never
    re-synchronise by themselves (flat / periodic content: as many passes as subsequences) and restart intervals."""
    rs = np.random.RandomState(sub_bytes)
    yy, xx = np.mgrid[0:256, 0:384]
    cases = [(_smooth(rs, h, w), dict(quality=q, subsampling=ss)) for (h, w) in [(8, 8), (37, 53), (1, 1), (100, 133), (241, 319)]
             for ss in (0, 1, 2) for q in (30, 100)]
    noise = rs.randint(0, 256, (123, 211, 3), dtype=np.uint8)
    cases += [(noise, dict(quality=85, subsampling=ss, **kw)) for ss in (0, 2)
              for kw in ({}, {"restart_marker_blocks": 1}, {"restart_marker_blocks": 5}, {"restart_marker_rows": 3})]
    cases += [(rs.randint(0, 256, (77, 91), dtype=np.uint8), dict(quality=80)),
              (np.zeros((256, 384, 3), np.uint8) + 120, dict(quality=75, optimize=True)),                       # flat
              (np.stack([xx // 8] * 3, -1).astype(np.uint8), dict(quality=90, subsampling=0)),                   # a staircase: periodic blocks
              ((((xx // 8 + yy // 8) & 1) * 255).astype(np.uint8)[..., None].repeat(3, -1), dict(quality=75, optimize=True))]
    worst = 0
    for img, kw in cases:
        data = _jpeg(img, **kw)
        got, passes = jpeg_oracle.decode_parallel(data, sub_bytes, max_passes=1 << 20)
        assert np.array_equal(got, _pil(data)), (img.shape, kw)
        assert np.array_equal(got, jpeg_oracle.decode(data))                  # ... and the serial walk's
        worst = max(worst, passes)
    assert worst >= 2
    # damaged data is refused by both walks alike
    good = _jpeg(noise, quality=80)
    for cut in (len(good) // 2, len(good) - 2):
        with pytest.raises(ValueError):
            jpeg_oracle.decode_parallel(good[:cut], sub_bytes)


@pytest.mark.parametrize("subsampling", [0, 1, 2])
def test_progressive_files_equal_pillow(subsampling):
    """Progressive JPEG (spectral selection + successive approximation, T.81 Annex G): DC first / refinement scans, AC first scans
    with end-of-band runs, AC refinement scans with correction bits, restart intervals inside scans -- into the same coefficient
    planes as the baseline decoder, then the same inverse DCT / upsampling / colour: Pillow's pixels."""
    rs = np.random.RandomState(100 + subsampling)
    for (h, w) in [(8, 8), (16, 16), (37, 53), (1, 1), (2, 3), (5, 2), (17, 1), (100, 133), (241, 319)]:
        for q in (10, 90, 100):
            for kw in ({}, {"optimize": True}, {"restart_marker_blocks": 3}, {"restart_marker_rows": 1}):
                data = _jpeg(_smooth(rs, h, w), quality=q, subsampling=subsampling, progressive=True, **kw)
                assert np.array_equal(jpeg_oracle.decode(data), _pil(data)), (h, w, q, kw)
    noise = rs.randint(0, 256, (123, 211, 3), dtype=np.uint8)
    for q in (5, 50, 100):
        data = _jpeg(noise, quality=q, subsampling=subsampling, progressive=True)
        assert np.array_equal(jpeg_oracle.decode_parallel(data)[0], _pil(data))
    grey = _jpeg(rs.randint(0, 256, (77, 91), dtype=np.uint8), quality=80, progressive=True)
    assert np.array_equal(jpeg_oracle.decode(grey), _pil(grey))
    # a progressive file cut off anywhere is refused (Pillow: truncated)
    good = _jpeg(noise, quality=80, progressive=True)
    for cut in (len(good) // 3, len(good) // 2, len(good) - 2):
        with pytest.raises(ValueError):
            jpeg_oracle.decode(good[:cut])


# ---------------------------------------------------------------------------------------------- files as other encoders write them
def _segments(data):
    """[(marker, payload bytes incl. length)] up to and including SOS header; then the rest (entropy data + following) as raw"""
    out=[]; pos=2
    while True:
        assert data[pos]==0xFF
        m=data[pos+1]; L=struct.unpack('>H',data[pos+2:pos+4])[0]
        out.append((m,data[pos+2:pos+2+L])); pos+=2+L
        if m==0xDA: break
    return out, data[pos:]

def _rebuild(segs, rest):
    b=bytearray(b'\xff\xd8')
    for m,p in segs: b+=bytes([0xFF,m])+p
    return bytes(b)+rest

def _variants(data):
    segs,rest=_segments(data)
    # 1. all DHT tables in ONE segment (what cameras write)
    dht=[p[2:] for m,p in segs if m==0xC4]
    if len(dht)>1:
        merged=b''.join(dht); one=(0xC4, struct.pack('>H',len(merged)+2)+merged)
        s2=[]; done=False
        for m,p in segs:
            if m==0xC4:
                if not done: s2.append(one); done=True
            else: s2.append((m,p))
        yield 'merged DHT', _rebuild(s2,rest)
    # 2. all DQT tables in one segment, as 16-bit entries
    dqt=[p[2:] for m,p in segs if m==0xDB]
    tabs=[]
    for payload in dqt:
        i=0
        while i<len(payload):
            pq,tq=payload[i]>>4,payload[i]&15; n=64*(pq+1); vals=payload[i+1:i+1+n]; i+=1+n
            v=[vals[k] for k in range(64)] if pq==0 else [struct.unpack('>H',vals[2*k:2*k+2])[0] for k in range(64)]
            tabs.append((tq,v))
    body=b''.join(bytes([0x10|tq])+b''.join(struct.pack('>H',x) for x in v) for tq,v in tabs)
    one=(0xDB, struct.pack('>H',len(body)+2)+body)
    s2=[]; done=False
    for m,p in segs:
        if m==0xDB:
            if not done: s2.append(one); done=True
        else: s2.append((m,p))
    yield '16-bit DQT', _rebuild(s2,rest)
    # 3. comment + EXIF-like APP1 + APP13 segments in front of and behind the frame header
    com=(0xFE, struct.pack('>H',2+11)+b'hello world'); app1=(0xE1, struct.pack('>H',2+300)+b'Exif\0\0'+bytes(294)); app13=(0xED, struct.pack('>H',2+40)+bytes(range(40)))
    s2=[]
    for m,p in segs:
        if m in (0xC0,0xC2): s2+=[com,app1]
        s2.append((m,p))
        if m in (0xC0,0xC2): s2+=[app13]
    yield 'COM/APPn', _rebuild(s2,rest)
    # 4. tables behind the frame header (DQT, DHT after SOF)
    s2=[x for x in segs if x[0] not in (0xDB,0xC4,0xDA)]+[x for x in segs if x[0] in (0xDB,0xC4)]+[x for x in segs if x[0]==0xDA]
    yield 'tables after SOF', _rebuild(s2,rest)
    # 5. fill bytes in front of every marker inside the entropy data / between scans, and in front of EOI
    r=bytearray(); i=0
    while i<len(rest):
        if rest[i]==0xFF and i+1<len(rest) and rest[i+1] not in (0x00,):
            r+=b'\xff\xff\xff'
        r.append(rest[i]); 
        if rest[i]==0xFF and i+1<len(rest) and rest[i+1]==0x00:
            r.append(0); i+=2; continue
        i+=1
    yield 'fill bytes', _rebuild(segs,bytes(r))
    # 6. no JFIF marker (component ids 1 2 3 decide), 7. trailing bytes
    yield 'no JFIF', _rebuild([x for x in segs if x[0]!=0xE0],rest)
    yield 'trailing bytes', data+b'\x00\x01\xff\x00\xffjunk'*3


def test_structural_variants_of_the_same_stream_decode_like_pillow():
    """Pillow writes one layout; cameras and other encoders write others.  Byte-level rewrites of Pillow's files -- all Huffman
    tables in one DHT segment, 16-bit quantisation tables in one DQT, COM / EXIF-like APP1 / APP13 segments around the frame
    header, tables behind the frame header, fill bytes in front of every marker of the entropy data, no JFIF marker, bytes
    behind EOI -- baseline, restart intervals, optimised tables, progressive, colour and grey."""
    import struct  # noqa: F401  (used by the helpers above)
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (67, 93, 3), dtype=np.uint8)
    grey = rs.randint(0, 256, (40, 50), dtype=np.uint8)
    n = 0
    for kw in (dict(quality=85), dict(quality=85, subsampling=0, restart_marker_blocks=4), dict(quality=60, subsampling=1, optimize=True),
               dict(quality=85, progressive=True), dict(quality=85, progressive=True, restart_marker_rows=1, subsampling=1)):
        for arr in (img, grey):
            data = _jpeg(arr, **kw)
            for name, v in _variants(data):
                ref = _pil(v)
                assert np.array_equal(jpeg_oracle.decode_parallel(v)[0], ref), (name, kw, arr.shape)
                n += 1
    assert n >= 60


def test_short_jfif_app0_does_not_count_as_jfif():
    """libjpeg (and so Pillow) sets saw_JFIF_marker only for an APP0 of >= 14 data bytes.  A 3-component file with a 13-byte
    'JFIF' APP0 and an Adobe marker with transform 0 is RGB-coded for libjpeg -- read as JFIF it would be converted from YCbCr,
    i.e. differ from Pillow.  The parser must take the Adobe rule (and leave RGB-coded files to the caller's Pillow path)."""
    import struct
    img = np.random.RandomState(1).randint(0, 256, (24, 31, 3), dtype=np.uint8)
    data = _jpeg(img, quality=90, subsampling=0)
    segs, rest = _segments(data)
    adobe = (0xEE, struct.pack('>H', 2 + 12) + b'Adobe' + struct.pack('>HHHB', 100, 0, 0, 0))           # transform = 0
    assert segs[0][0] == 0xE0                                              # Pillow writes the JFIF APP0 first
    for n_app0 in (12, 13):
        short = (0xE0, struct.pack('>H', 2 + n_app0) + (b'JFIF\0\x01\x01\x00\x00\x01\x00\x01\x00\x00')[:n_app0])
        v = _rebuild([short, adobe] + segs[1:], rest)
        ref = _pil(v)                                                  # Pillow / libjpeg: Adobe transform 0 = no colour conversion
        rc = jpeg_oracle.info(v)[0]
        if rc == 0:
            assert np.array_equal(jpeg_oracle.decode(v), ref), n_app0
        else:
            assert jpeg_oracle.REASONS[rc] == "colour space", (n_app0, rc)
    full = _rebuild([segs[0], adobe] + segs[1:], rest)                   # a proper 14-byte JFIF APP0 wins over Adobe, as in libjpeg
    assert np.array_equal(jpeg_oracle.decode(full), _pil(full))


def test_any_sampling_libjpeg_upsamples_equals_pillow():
    """Streams Pillow cannot be asked to write (tests/jpeg_writer.py entropy-codes random quantised coefficients): 4:4:0 (what a
    losslessly rotated 4:2:2 photo is), 4:1:1, 1x4, 4x2 and 2x4 luma (ten blocks per MCU), chroma planes sampled differently from
    each other, luma sampled below chroma, a grey file that declares 2x2 -- with and without restart intervals, zero runs over 16."""
    from tests.jpeg_writer import random_coefs, tables_from_pillow, write_baseline
    dqt, dht = tables_from_pillow(80)
    rs = np.random.RandomState(0)
    configs = [[(1, 1)] * 3, [(2, 1), (1, 1), (1, 1)], [(2, 2), (1, 1), (1, 1)], [(1, 2), (1, 1), (1, 1)], [(4, 1), (1, 1), (1, 1)],
               [(1, 4), (1, 1), (1, 1)], [(4, 2), (1, 1), (1, 1)], [(2, 4), (1, 1), (1, 1)], [(2, 2), (2, 1), (1, 1)], [(2, 2), (1, 2), (2, 1)],
               [(2, 2), (2, 2), (1, 1)], [(1, 1), (2, 2), (2, 2)], [(2, 1), (1, 2), (1, 1)], [(2, 2)], [(1, 1)]]
    for samp in configs:
        for (w, h) in [(8, 8), (17, 9), (33, 47), (100, 37), (3, 2)]:
            for restart in (0, 1, 5):
                data = write_baseline(w, h, samp, random_coefs(rs, w, h, samp), dqt, dht, restart=restart)
                ref = _pil(data)
                assert np.array_equal(jpeg_oracle.decode_parallel(data, 64)[0], ref), (samp, w, h, restart)
                assert np.array_equal(jpeg_oracle.decode(data), ref), (samp, w, h, restart)
    # factors that do not divide the largest ones (3 next to 2): libjpeg refuses them, so does the parser
    bad = write_baseline(32, 32, [(2, 2), (1, 1), (1, 1)], random_coefs(rs, 32, 32, [(2, 2), (1, 1), (1, 1)]), dqt, dht)
    i = bad.index(b"\xff\xc0")
    bad = bad[: i + 11] + bytes([0x31]) + bad[i + 12: i + 14] + bytes([0x21]) + bad[i + 15:]      # luma 3x1 next to a 2x1 chroma plane
    assert jpeg_oracle.info(bad)[0] == 5
    with pytest.raises(OSError):
        _pil(bad)


def test_progressive_files_with_any_sampling_equal_pillow():
    """Spectral-selection progressive files from tests/jpeg_writer.py (what a losslessly rotated progressive photo looks like when
    its sampling becomes 4:4:0): the single-component AC scans walk each component's REAL blocks, not the MCU grid."""
    from tests.jpeg_writer import random_coefs, tables_from_pillow, write_progressive
    dqt, dht = tables_from_pillow(80)
    rs = np.random.RandomState(3)
    for samp in ([(1, 1)] * 3, [(2, 2), (1, 1), (1, 1)], [(1, 2), (1, 1), (1, 1)], [(4, 1), (1, 1), (1, 1)], [(1, 4), (1, 1), (1, 1)],
                 [(4, 2), (1, 1), (1, 1)], [(2, 2), (2, 1), (1, 1)], [(2, 2), (1, 2), (2, 1)], [(1, 1), (2, 2), (2, 2)], [(2, 2)]):
        for (w, h) in [(8, 8), (17, 9), (33, 47), (100, 37), (3, 2)]:
            for restart in (0, 2):
                data = write_progressive(w, h, samp, random_coefs(rs, w, h, samp), dqt, dht, restart=restart)
                assert np.array_equal(jpeg_oracle.decode(data), _pil(data)), (samp, w, h, restart)


def test_sequential_files_in_several_scans_equal_pillow():
    """A sequential (SOF0) file may bring its components in several scans -- one each, or one alone and two interleaved; the decoder
    walks them with the machinery of the progressive files (tests/jpeg_writer.py writes them, Pillow gives the expected pixels)."""
    from tests.jpeg_writer import random_coefs, tables_from_pillow, write_sequential_scans
    dqt, dht = tables_from_pillow(80)
    rs = np.random.RandomState(4)
    for samp in ([(1, 1)] * 3, [(2, 2), (1, 1), (1, 1)], [(2, 1), (1, 1), (1, 1)], [(1, 2), (1, 1), (1, 1)], [(2, 2), (1, 2), (2, 1)]):
        for scans in ([[0], [1], [2]], [[0], [1, 2]], [[0, 1], [2]]):
            for (w, h) in [(8, 8), (17, 9), (33, 47), (100, 37), (3, 2)]:
                for restart in (0, 3):
                    data = write_sequential_scans(w, h, samp, random_coefs(rs, w, h, samp), dqt, dht, scans, restart=restart)
                    assert np.array_equal(jpeg_oracle.decode(data), _pil(data)), (samp, scans, w, h, restart)
    # a component that never arrives: refused (Pillow decodes what is there and leaves the rest grey)
    data = write_sequential_scans(33, 47, [(1, 1)] * 3, random_coefs(rs, 33, 47, [(1, 1)] * 3), dqt, dht, [[0], [1]])
    assert jpeg_oracle.info(data)[0] != 0


def test_header_claiming_a_huge_image_is_refused():
    """Pillow raises DecompressionBombError above 2 x MAX_IMAGE_PIXELS; the decoder hands such files back (reason 10) instead of
    planning gigabytes of scratch for a few header bytes."""
    data = bytearray(_jpeg(np.zeros((16, 16, 3), np.uint8), quality=80))
    i = data.index(b"\xff\xc0")
    data[i + 5:i + 9] = bytes([0x3c, 0x00, 0x3c, 0x00])            # 15360 x 15360 = 236 M pixels
    assert jpeg_oracle.info(bytes(data))[0] == 10
    data[i + 5:i + 9] = bytes([0x30, 0x00, 0x30, 0x00])            # 12288 x 12288 = 151 M pixels: parsed (and then found truncated)
    assert jpeg_oracle.info(bytes(data))[0] == 0


def test_scratch_bytes_per_pixel_follows_the_sampling_factors():
    """What the embed driver reserves the device decoder's scratch by (jpeg_gpu.scratch_bytes_per_pixel): int16 coefficients + sample
    planes per decoded pixel, from the frame header alone."""
    import io
    import numpy as np
    from PIL import Image
    from clip_assisted_data_labeling_amd.jpeg_gpu import scratch_bytes_per_pixel
    im = Image.fromarray(np.random.RandomState(0).randint(0, 256, (64, 80, 3), dtype=np.uint8))
    for subsampling, want in ((2, 4.5), (1, 6.0), (0, 9.0)):
        b = io.BytesIO(); im.save(b, "JPEG", quality=90, subsampling=subsampling)
        assert scratch_bytes_per_pixel(b.getvalue()) == want
    b = io.BytesIO(); im.convert("L").save(b, "JPEG")
    assert scratch_bytes_per_pixel(b.getvalue()) == 3.0
    b = io.BytesIO(); im.save(b, "JPEG", progressive=True)
    assert scratch_bytes_per_pixel(b.getvalue()) == 4.5
    assert scratch_bytes_per_pixel(b"not a jpeg") == 9.0 and scratch_bytes_per_pixel(b"") == 9.0
