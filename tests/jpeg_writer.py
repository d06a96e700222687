"""Test helper: a minimal baseline JPEG WRITER (entropy coding only) for streams Pillow cannot be asked to produce -- arbitrary
sampling factors (4:4:0, 4:1:1, a chroma plane sampled 2x1 next to one sampled 1x1, ...), long zero runs, restart intervals of any
length.  It takes quantised coefficients directly (no forward DCT): the point is the stream layout, and Pillow decodes the result
for the expected pixels.  Quantisation and Huffman tables are lifted from a file Pillow wrote (the standard Annex K tables)."""
import io
import struct

import numpy as np
from PIL import Image


def tables_from_pillow(quality=75):
    b = io.BytesIO()
    Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(b, "JPEG", quality=quality)
    data = b.getvalue()
    pos, dqt, dht = 2, {}, {}
    while True:
        m = data[pos + 1]
        ln = struct.unpack(">H", data[pos + 2:pos + 4])[0]
        body = data[pos + 4:pos + 2 + ln]
        if m == 0xDB:
            i = 0
            while i < len(body):
                dqt[body[i] & 15] = list(body[i + 1:i + 65]); i += 65
        elif m == 0xC4:
            i = 0
            while i < len(body):
                counts = list(body[i + 1:i + 17]); n = sum(counts)
                dht[(body[i] >> 4, body[i] & 15)] = (counts, list(body[i + 17:i + 17 + n])); i += 17 + n
        elif m == 0xDA:
            break
        pos += 2 + ln
    return dqt, dht


def _codes(counts, vals):
    out, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(counts[ln - 1]):
            out[vals[k]] = (code, ln); code += 1; k += 1
        code <<= 1
    return out


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, code, ln):
        self.acc = (self.acc << ln) | (code & ((1 << ln) - 1)); self.n += ln
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(byte)
            if byte == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _cat(v):
    return int(abs(int(v))).bit_length()


def write_baseline(width, height, samp, coefs, dqt, dht, restart=0, comp_tq=(0, 1, 1), comp_tabs=((0, 0), (1, 1), (1, 1))):
    """samp: [(h, v)] per component; coefs[c]: int array [bh][bw][64] in ZIGZAG order, bh / bw = MCU rows / columns x v / h."""
    nc = len(samp)
    hmax, vmax = (max(h for h, _ in samp), max(v for _, v in samp)) if nc > 1 else (1, 1)   # one component: never interleaved, 8 x 8 "MCUs"
    mx, my = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    out = bytearray(b"\xff\xd8\xff\xe0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    for tq in sorted(set(comp_tq[:nc])):
        out += b"\xff\xdb" + struct.pack(">H", 67) + bytes([tq]) + bytes(dqt[tq])
    out += b"\xff\xc0" + struct.pack(">HBHHB", 8 + 3 * nc, 8, height, width, nc)
    for c in range(nc):
        out += bytes([c + 1, (samp[c][0] << 4) | samp[c][1], comp_tq[c]])
    for key in sorted({(0, comp_tabs[c][0]) for c in range(nc)} | {(1, comp_tabs[c][1]) for c in range(nc)}):
        counts, vals = dht[key]
        out += b"\xff\xc4" + struct.pack(">H", 19 + len(vals)) + bytes([(key[0] << 4) | key[1]]) + bytes(counts) + bytes(vals)
    if restart:
        out += b"\xff\xdd\x00\x04" + struct.pack(">H", restart)
    out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * nc, nc)
    for c in range(nc):
        out += bytes([c + 1, (comp_tabs[c][0] << 4) | comp_tabs[c][1]])
    out += b"\x00\x3f\x00"
    dc = [_codes(*dht[(0, comp_tabs[c][0])]) for c in range(nc)]
    ac = [_codes(*dht[(1, comp_tabs[c][1])]) for c in range(nc)]
    bits, pred, n_mcu, rst = _Bits(), [0] * nc, 0, 0
    for y in range(my):
        for x in range(mx):
            if restart and n_mcu and n_mcu % restart == 0:
                bits.flush(); out += bits.out + bytes([0xFF, 0xD0 + rst]); rst = (rst + 1) & 7
                bits, pred = _Bits(), [0] * nc
            n_mcu += 1
            for c in range(nc):
                h, v = samp[c] if nc > 1 else (1, 1)
                for by in range(v):
                    for bx in range(h):
                        blk = coefs[c][y * v + by][x * h + bx]
                        d = int(blk[0]) - pred[c]; pred[c] = int(blk[0])
                        s = _cat(d)
                        bits.put(*dc[c][s])
                        if s:
                            bits.put(d if d > 0 else d + (1 << s) - 1, s)
                        run = 0
                        last = max([k for k in range(1, 64) if blk[k]] or [0])
                        for k in range(1, last + 1):
                            if blk[k] == 0:
                                run += 1
                                continue
                            while run > 15:
                                bits.put(*ac[c][0xF0]); run -= 16
                            s = _cat(blk[k])
                            bits.put(*ac[c][(run << 4) | s])
                            vv = int(blk[k])
                            bits.put(vv if vv > 0 else vv + (1 << s) - 1, s)
                            run = 0
                        if last < 63:
                            bits.put(*ac[c][0x00])
    bits.flush()
    return bytes(out + bits.out + b"\xff\xd9")


def random_coefs(rs, width, height, samp):
    """sparse, small quantised coefficients: smooth-ish blocks with a few AC terms, now and then a zero run longer than 16"""
    nc = len(samp)
    hmax, vmax = (max(h for h, _ in samp), max(v for _, v in samp)) if nc > 1 else (1, 1)
    mx, my = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    out = []
    for c in range(nc):
        h, v = samp[c] if nc > 1 else (1, 1)
        a = np.zeros((my * v, mx * h, 64), dtype=np.int32)
        a[..., 0] = rs.randint(-40, 41, a.shape[:2])
        for k in range(1, 12):
            a[..., k] = rs.randint(-12, 13, a.shape[:2]) * (rs.rand(*a.shape[:2]) < 0.3)
        far = rs.rand(*a.shape[:2]) < 0.1
        a[..., 40 + c] = rs.randint(-3, 4, a.shape[:2]) * far          # behind a run of 28+ zeros: ZRL
        out.append(a)
    return out


def write_progressive(width, height, samp, coefs, dqt, dht, restart=0, comp_tq=(0, 1, 1), comp_tabs=((0, 0), (1, 1), (1, 1)), bands=((1, 5), (6, 63))):
    """The same coefficients as a PROGRESSIVE file by spectral selection only (no successive approximation): one interleaved DC
    scan, then per component one AC scan per band (every block's band ends with EOB0 unless it is full) -- single-component scans
    walk the component's REAL blocks, ceil(dw / 8) x ceil(dh / 8), which is where exotic samplings differ from the MCU walk."""
    nc = len(samp)
    hmax, vmax = (max(h for h, _ in samp), max(v for _, v in samp)) if nc > 1 else (1, 1)
    mx, my = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    out = bytearray(b"\xff\xd8\xff\xe0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    for tq in sorted(set(comp_tq[:nc])):
        out += b"\xff\xdb" + struct.pack(">H", 67) + bytes([tq]) + bytes(dqt[tq])
    out += b"\xff\xc2" + struct.pack(">HBHHB", 8 + 3 * nc, 8, height, width, nc)
    for c in range(nc):
        out += bytes([c + 1, (samp[c][0] << 4) | samp[c][1], comp_tq[c]])
    for key in sorted({(0, comp_tabs[c][0]) for c in range(nc)} | {(1, comp_tabs[c][1]) for c in range(nc)}):
        counts, vals = dht[key]
        out += b"\xff\xc4" + struct.pack(">H", 19 + len(vals)) + bytes([(key[0] << 4) | key[1]]) + bytes(counts) + bytes(vals)
    if restart:
        out += b"\xff\xdd\x00\x04" + struct.pack(">H", restart)
    dc = [_codes(*dht[(0, comp_tabs[c][0])]) for c in range(nc)]
    ac = [_codes(*dht[(1, comp_tabs[c][1])]) for c in range(nc)]

    def scan(comps, ss, se, units):
        """units: iterable of lists of (component, block) = the MCUs of the scan in order"""
        nonlocal out
        out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(comps), len(comps))
        for c in comps:
            out += bytes([c + 1, (comp_tabs[c][0] << 4) | comp_tabs[c][1]])
        out += bytes([ss, se, 0])
        bits, pred, n, rst = _Bits(), [0] * nc, 0, 0
        for mcu in units:
            if restart and n and n % restart == 0:
                bits.flush(); out += bits.out + bytes([0xFF, 0xD0 + rst]); rst = (rst + 1) & 7
                bits, pred = _Bits(), [0] * nc
            n += 1
            for c, blk in mcu:
                if ss == 0:
                    d = int(blk[0]) - pred[c]; pred[c] = int(blk[0])
                    s_ = _cat(d)
                    bits.put(*dc[c][s_])
                    if s_:
                        bits.put(d if d > 0 else d + (1 << s_) - 1, s_)
                    continue
                run = 0
                last = max([k for k in range(ss, se + 1) if blk[k]] or [ss - 1])
                for k in range(ss, last + 1):
                    if blk[k] == 0:
                        run += 1
                        continue
                    while run > 15:
                        bits.put(*ac[c][0xF0]); run -= 16
                    s_ = _cat(blk[k])
                    bits.put(*ac[c][(run << 4) | s_])
                    vv = int(blk[k])
                    bits.put(vv if vv > 0 else vv + (1 << s_) - 1, s_)
                    run = 0
                if last < se:
                    bits.put(*ac[c][0x00])
        bits.flush()
        out += bits.out

    def mcus():
        for y in range(my):
            for x in range(mx):
                yield [(c, coefs[c][y * (samp[c][1] if nc > 1 else 1) + by][x * (samp[c][0] if nc > 1 else 1) + bx])
                       for c in range(nc) for by in range(samp[c][1] if nc > 1 else 1) for bx in range(samp[c][0] if nc > 1 else 1)]
    scan(list(range(nc)), 0, 0, mcus())
    for c in range(nc):
        h, v = samp[c] if nc > 1 else (1, 1)
        dw, dh = -(-width * h // hmax), -(-height * v // vmax)
        nbx, nby = -(-dw // 8), -(-dh // 8)
        for (ss, se) in bands:
            scan([c], ss, se, ([(c, coefs[c][by][bx])] for by in range(nby) for bx in range(nbx)))
    return bytes(out + b"\xff\xd9")


def write_sequential_scans(width, height, samp, coefs, dqt, dht, scans, restart=0, comp_tq=(0, 1, 1), comp_tabs=((0, 0), (1, 1), (1, 1))):
    """A SEQUENTIAL (SOF0) file whose components come in several scans, e.g. scans = [[0], [1], [2]] or [[0], [1, 2]]: every scan
    codes the full band of its components; one component alone walks its real blocks, several are interleaved MCU by MCU."""
    nc = len(samp)
    hmax, vmax = max(h for h, _ in samp), max(v for _, v in samp)
    mx, my = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    out = bytearray(b"\xff\xd8\xff\xe0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    for tq in sorted(set(comp_tq[:nc])):
        out += b"\xff\xdb" + struct.pack(">H", 67) + bytes([tq]) + bytes(dqt[tq])
    out += b"\xff\xc0" + struct.pack(">HBHHB", 8 + 3 * nc, 8, height, width, nc)
    for c in range(nc):
        out += bytes([c + 1, (samp[c][0] << 4) | samp[c][1], comp_tq[c]])
    for key in sorted({(0, comp_tabs[c][0]) for c in range(nc)} | {(1, comp_tabs[c][1]) for c in range(nc)}):
        counts, vals = dht[key]
        out += b"\xff\xc4" + struct.pack(">H", 19 + len(vals)) + bytes([(key[0] << 4) | key[1]]) + bytes(counts) + bytes(vals)
    if restart:
        out += b"\xff\xdd\x00\x04" + struct.pack(">H", restart)
    dc = [_codes(*dht[(0, comp_tabs[c][0])]) for c in range(nc)]
    ac = [_codes(*dht[(1, comp_tabs[c][1])]) for c in range(nc)]
    for comps in scans:
        out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(comps), len(comps))
        for c in comps:
            out += bytes([c + 1, (comp_tabs[c][0] << 4) | comp_tabs[c][1]])
        out += b"\x00\x3f\x00"
        if len(comps) == 1:
            c = comps[0]
            h, v = samp[c]
            dw, dh = -(-width * h // hmax), -(-height * v // vmax)
            units = ([(c, coefs[c][by][bx])] for by in range(-(-dh // 8)) for bx in range(-(-dw // 8)))
        else:
            units = ([(c, coefs[c][y * samp[c][1] + by][x * samp[c][0] + bx]) for c in comps for by in range(samp[c][1]) for bx in range(samp[c][0])]
                     for y in range(my) for x in range(mx))
        bits, pred, n, rst = _Bits(), [0] * nc, 0, 0
        for mcu in units:
            if restart and n and n % restart == 0:
                bits.flush(); out += bits.out + bytes([0xFF, 0xD0 + rst]); rst = (rst + 1) & 7
                bits, pred = _Bits(), [0] * nc
            n += 1
            for c, blk in mcu:
                d = int(blk[0]) - pred[c]; pred[c] = int(blk[0])
                s_ = _cat(d)
                bits.put(*dc[c][s_])
                if s_:
                    bits.put(d if d > 0 else d + (1 << s_) - 1, s_)
                run = 0
                last = max([k for k in range(1, 64) if blk[k]] or [0])
                for k in range(1, last + 1):
                    if blk[k] == 0:
                        run += 1
                        continue
                    while run > 15:
                        bits.put(*ac[c][0xF0]); run -= 16
                    s_ = _cat(blk[k])
                    bits.put(*ac[c][(run << 4) | s_])
                    vv = int(blk[k])
                    bits.put(vv if vv > 0 else vv + (1 << s_) - 1, s_)
                    run = 0
                if last < 63:
                    bits.put(*ac[c][0x00])
        bits.flush()
        out += bits.out
    return bytes(out + b"\xff\xd9")
