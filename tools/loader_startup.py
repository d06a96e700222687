#!/usr/bin/env python3
"""Developer: where the host-decode path's start-up goes (embed_e2e: first batch stored after ~1.8 s against 0.29 s with --gpu_decode).
Times the DataLoader's worker start and its first batch in a process that has (a) not and (b) already initialised the GPU and built an encoder.
  python tools/loader_startup.py [n_images] [workers]"""
import os, sys, tempfile, time
import numpy as np
import torch
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch.utils.data import DataLoader
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embed_driver import RawImageDataset, _collate

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 16
paths = []


def make_files():
    d = tempfile.mkdtemp(prefix="ldr_")
    rng = np.random.default_rng(0)
    for i in range(n):
        p = os.path.join(d, f"{i:05d}.jpg")
        Image.fromarray(rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)).save(p, quality=90)
        paths.append(p)


def first_batch(tag, ctx=None, pin=False):
    t0 = time.perf_counter()
    kw = dict(batch_size=max(4, 512 // workers), shuffle=False, num_workers=workers, collate_fn=_collate, prefetch_factor=4, pin_memory=pin)
    if ctx:
        kw["multiprocessing_context"] = ctx
    it = iter(DataLoader(RawImageDataset(paths), **kw))
    t1 = time.perf_counter()
    next(it)
    t2 = time.perf_counter()
    for _ in range(workers - 1):
        next(it)
    t3 = time.perf_counter()
    print(f"{tag:46s} workers started {t1 - t0:6.3f} s   first loader batch +{t2 - t1:6.3f} s   {workers} loader batches (one encode batch) +{t3 - t2:6.3f} s", flush=True)
    del it


def main():
    make_files()
    ds = RawImageDataset(paths)
    t = []
    for i in range(12):                                     # the same decode in THIS process: first calls against warm ones
        t0 = time.perf_counter(); ds[i]; t.append(time.perf_counter() - t0)
    print("in-process decode of the first 12 images, ms:", " ".join(f"{x * 1e3:.1f}" for x in t), flush=True)
    from concurrent.futures import ThreadPoolExecutor
    for nt in (4, 8, 16, 32):                              # the same decode on a pool of THREADS (Pillow's codecs release the GIL)
        with ThreadPoolExecutor(nt) as pool:
            t0 = time.perf_counter()
            first = None
            for i, r in enumerate(pool.map(ds.__getitem__, range(len(paths)))):
                if i == 511 and first is None:
                    first = time.perf_counter() - t0
            dt = time.perf_counter() - t0
        print(f"thread pool of {nt:2d}: first 512 images after {first:.3f} s, {len(paths) / dt:7.0f} images/s", flush=True)
    first_batch("fork, GPU not initialised")
    first_batch("fork, GPU not initialised, second loader")
    if torch.cuda.is_available():                           # (fork only: a spawn / forkserver child would initialise the GPU again)
        from clip_assisted_data_labeling_amd.embedder import HipViT
        cfg = vit_config.ARCHS["ViT-L-14"]
        vit = HipViT(cfg, vit_config.seeded_state_dict(cfg, 0), "cuda")
        vit.encode(torch.randn(2048, 3, 224, 224, device="cuda"))
        torch.cuda.synchronize()
        first_batch("fork, encoder built and run")
        first_batch("fork, encoder built, pinned batches", pin=True)


if __name__ == "__main__":
    main()
