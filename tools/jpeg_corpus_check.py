#!/usr/bin/env python3
"""Developer: the GPU JPEG decoder against Pillow on a large random corpus (sizes 1 .. 1500, every sampling, qualities 1 .. 100,
optimised / default tables, restart intervals, grey, content from flat to noise, 30 % of the files progressive); prints mismatches,
exits 1 if any."""
import io, os, sys, time
import numpy as np, torch
from PIL import Image, ImageFile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd.jpeg_gpu import GpuJpegDecoder
ImageFile.MAXBLOCK = 1 << 26
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rs = np.random.RandomState(int(os.environ.get("SEED", "0")))
def content(h, w, grey):
    kind = rs.randint(0, 6)
    c = 1 if grey else 3
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == 0: a = rs.randint(0, 256, (h, w, c))
    elif kind == 1: a = np.zeros((h, w, c)) + rs.randint(0, 256)
    elif kind == 2: a = np.stack([(xx * rs.rand() * 2 + yy * rs.rand() * 2) % 256] * c, -1)
    elif kind == 3: a = np.stack([128 + 120 * np.sin(xx / (2 + 40 * rs.rand()) + yy / (2 + 40 * rs.rand()) + i) for i in range(c)], -1) + rs.randn(h, w, c) * rs.rand() * 30
    elif kind == 4: a = np.stack([(((xx // (1 + rs.randint(0, 20))) + (yy // (1 + rs.randint(0, 20)))) & 1) * 255] * c, -1)
    else:
        a = np.zeros((h, w, c)) + rs.randint(0, 256); hh, ww = max(1, h // 3), max(1, w // 3); a[:hh, :ww] = rs.randint(0, 256, (hh, ww, c))
    a = np.clip(a, 0, 255).astype(np.uint8)
    return a[..., 0] if grey else a
files, metas = [], []
t0 = time.time()
for i in range(n):
    big = rs.rand() < 0.05
    h = rs.randint(1, 1500 if big else 400); w = rs.randint(1, 1500 if big else 400)
    grey = rs.rand() < 0.1
    kw = dict(quality=int(rs.choice([1, 5, 20, 50, 75, 85, 90, 95, 100])), optimize=bool(rs.rand() < 0.3))
    if rs.rand() < float(os.environ.get("PROGRESSIVE", "0.3")): kw["progressive"] = True
    if not grey: kw["subsampling"] = int(rs.randint(0, 3))
    r = rs.rand()
    if r < 0.15: kw["restart_marker_blocks"] = int(rs.randint(1, 40))
    elif r < 0.3: kw["restart_marker_rows"] = int(rs.randint(1, 5))
    b = io.BytesIO(); Image.fromarray(content(h, w, grey)).save(b, "JPEG", **kw)
    files.append(b.getvalue()); metas.append((h, w, grey, kw))
print(f"{n} files, {sum(map(len, files)) / 1e6:.0f} MB, generated in {time.time() - t0:.0f} s", flush=True)
dec = GpuJpegDecoder(torch.device("cuda", 0))
bad = 0
for lo in range(0, n, 1000):
    imgs, st = dec.decode(files[lo:lo + 1000])
    for j, (im, s) in enumerate(zip(imgs, st)):
        ref = np.asarray(Image.open(io.BytesIO(files[lo + j])).convert("RGB"))
        if s != 0 or not np.array_equal(im.cpu().numpy(), ref):
            bad += 1; print("MISMATCH", lo + j, metas[lo + j], "status", s, flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
