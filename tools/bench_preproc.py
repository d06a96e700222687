#!/usr/bin/env python3
"""Developer timing: GPU front end (4 crops per image) vs the Pillow path on one host core."""
import os, sys, time
import numpy as np, torch
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd.preprocess import ClipValTransform, GpuCropper, extract_crops

dev = torch.device("cuda", 0)
cropper = GpuCropper(224, dev)
rs = np.random.RandomState(0)
for (w, h) in [(640, 480), (1024, 768), (2000, 1500), (4000, 3000)]:
    arr = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
    img_gpu = torch.from_numpy(arr).to(dev)
    for _ in range(3): cropper(img_gpu)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): cropper(img_gpu)
    torch.cuda.synchronize(); gpu_ms = (time.perf_counter() - t0) / n * 1e3
    pil = Image.fromarray(arr); tf = ClipValTransform(224)
    t0 = time.perf_counter()
    for _ in range(5): [tf.to_uint8(c) for c in extract_crops(pil)[0]]
    cpu_ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"{w}x{h}: GPU front end {gpu_ms:.3f} ms/image (image already in HBM), Pillow on one core {cpu_ms:.2f} ms/image")
