#!/usr/bin/env python3
"""Developer: where the time of `embed_driver --gpu_decode` goes on generated files (cProfile of the main process, GPU idle gaps
show up as time in synchronize / decode)."""
import cProfile, io, os, pstats, shutil, sys, tempfile, time
import numpy as np, torch
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import embed_driver
from clip_assisted_data_labeling_amd.embedder import CLIP_Encoder
n = int(os.environ.get("N", "4096")); size = 512
tmp = tempfile.mkdtemp(prefix="prof_e2e_")
try:
    base = np.random.RandomState(0).randint(0, 256, (size, size, 3), dtype=np.uint8)
    for i in range(n):
        Image.fromarray(np.roll(base, i * 7, axis=1)).save(os.path.join(tmp, f"{i:06d}.jpg"), quality=90)
    enc = CLIP_Encoder("ViT-L-14/seed0", None, device="cuda:0")
    ds = embed_driver.Feature_Dataset(tmp, "ViT-L-14/seed0", 256, shuffle_filenames=False, num_workers=16, encoder=enc, device="cuda:0",
                                      force_reencode=True, gpu_decode=True, packed_store=os.environ.get("PACKED"))
    torch.cuda.synchronize()
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    ds.process()
    torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t0
    print(f"{n} images in {dt:.2f} s = {n / dt:.0f} images/s")
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[-3800:])
finally:
    shutil.rmtree(tmp, ignore_errors=True)
