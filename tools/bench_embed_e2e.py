#!/usr/bin/env python3
"""Developer timing: the whole embed driver on JPEG files (decode in DataLoader workers -> GPU front end -> encoder -> store)."""
import argparse, os, shutil, sys, tempfile, time
import numpy as np
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2048)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--workers", type=int, default=16)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--model", default="ViT-L-14/seed0")
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="e2e_")
try:
    rs = np.random.RandomState(0)
    base = rs.randint(0, 256, (a.size, a.size, 3), dtype=np.uint8)
    t = time.perf_counter()
    for i in range(a.n):
        img = np.roll(base, i * 7, axis=1)
        Image.fromarray(img).save(os.path.join(tmp, f"{i:06d}.jpg"), quality=90)
    print(f"wrote {a.n} JPEGs {a.size}x{a.size} in {time.perf_counter() - t:.1f}s; host cpus: {os.cpu_count()}")
    import torch
    from clip_assisted_data_labeling_amd import embed_driver
    from clip_assisted_data_labeling_amd.embedder import CLIP_Encoder
    enc = CLIP_Encoder(a.model, None, device="cuda")
    for mode in ("pt+cpu_preprocess", "pt+gpu_preprocess", "packed+gpu_preprocess"):
        for f in os.listdir(tmp):
            if f.endswith(".pt"):
                os.remove(os.path.join(tmp, f))
        store = os.path.join(tmp, "_store_" + mode.replace("+", "_")) if mode.startswith("packed") else None
        ds = embed_driver.Feature_Dataset(tmp, a.model, a.batch, shuffle_filenames=False, num_workers=a.workers, encoder=enc,
                                          device="cuda", gpu_preprocess="gpu" in mode, packed_store=store, force_reencode=True)
        torch.cuda.synchronize(); t = time.perf_counter()
        n_emb = ds.process()[0]
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"{mode:24s}: {n_emb} images in {dt:.2f} s = {n_emb / dt:,.0f} images/s (workers {a.workers}, batch {a.batch})")
finally:
    shutil.rmtree(tmp)
