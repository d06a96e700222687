#!/usr/bin/env python3
"""Developer: decode rate of the GPU JPEG decoder on generated files (noise = the bench's worst case, or smooth photo-like content)."""
import argparse, io, os, sys, time
import numpy as np, torch
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd.jpeg_gpu import GpuJpegDecoder
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=256); ap.add_argument("--size", type=int, default=512)
ap.add_argument("--kind", default="noise"); ap.add_argument("--quality", type=int, default=90); ap.add_argument("--subsampling", type=int, default=2)
ap.add_argument("--progressive", action="store_true")
args = ap.parse_args()
rs = np.random.RandomState(0)
def make(i):
    if args.kind == "noise":
        a = rs.randint(0, 256, (args.size, args.size, 3), dtype=np.uint8)
    else:
        yy, xx = np.mgrid[0:args.size, 0:args.size]
        a = np.clip(np.stack([128 + 100 * np.sin(xx / 17.0 + yy / 29.0 + i), 128 + 90 * np.cos(xx / 11.0 - yy / 23.0), 128 + 80 * np.sin((xx + yy) / 7.0)], -1)
                    + rs.randn(args.size, args.size, 3) * 6, 0, 255).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(a).save(b, "JPEG", quality=args.quality, subsampling=args.subsampling, progressive=args.progressive); return b.getvalue()
files = [make(i) for i in range(min(args.n, 32))]
files = (files * (args.n // len(files) + 1))[:args.n]
dev = torch.device("cuda", 0)
dec = GpuJpegDecoder(dev)
t0 = time.perf_counter(); ref = [np.asarray(Image.open(io.BytesIO(f)).convert("RGB")) for f in files[:32]]; t_pil = (time.perf_counter() - t0) / 32
imgs, st = dec.decode(files); assert not any(st)
assert all(np.array_equal(i.cpu().numpy(), r) for i, r in zip(imgs[:32], ref))
for _ in range(2): dec.decode(files)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): dec.decode(files)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"{args.kind}{' progressive' if args.progressive else ''} {args.size}x{args.size} q{args.quality} ss{args.subsampling}: {np.mean([len(f) for f in files]) / 1024:.0f} KB/file; "
      f"GPU batch of {args.n}: {dt * 1e3:.1f} ms = {args.n / dt:.0f} images/s;  Pillow on one host core: {t_pil * 1e3:.2f} ms/image = {1 / t_pil:.0f} images/s")
