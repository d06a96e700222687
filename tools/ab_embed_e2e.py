#!/usr/bin/env python3
"""Developer: the --gpu_decode embed driver on the bench's 4 096 generated JPEG files with several settings of its knobs, interleaved in
one process on one box (JSON knobs per variant: attribute -> value, set on the Feature_Dataset before process()).
    python tools/ab_embed_e2e.py --reps 3 '{}' '{"pt_writers": 4}' '{"batch_size": 256}'"""
import argparse, contextlib, io, json, os, shutil, sys, tempfile, time
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from PIL import Image
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="*", default=["{}"])
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--packed", action="store_true")
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="ab_e2e_")
try:
    base = np.random.RandomState(0).randint(0, 256, (a.size, a.size, 3), dtype=np.uint8)
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda i: Image.fromarray(np.roll(base, i * 7, axis=1)).save(os.path.join(tmp, f"{i:06d}.jpg"), quality=90), range(a.n)))
    import torch
    from clip_assisted_data_labeling_amd import embed_driver
    from clip_assisted_data_labeling_amd.embedder import CLIP_Encoder
    enc = CLIP_Encoder("ViT-L-14/seed0", None, device="cuda:0")
    for rep in range(a.reps):
        for vi, v in enumerate(a.variants):
            knobs = json.loads(v)
            drv = embed_driver
            if knobs.pop("driver", None) == "r03":          # (a copy of the round-3 driver placed next to the package for the A/B; not shipped)
                from clip_assisted_data_labeling_amd import _embed_driver_r03 as drv
            store = os.path.join(tmp, f"_store_{rep}_{vi}") if a.packed else None
            with contextlib.redirect_stdout(io.StringIO()):
                ds = drv.Feature_Dataset(tmp, "ViT-L-14/seed0", knobs.pop("batch_size", a.batch), shuffle_filenames=False, num_workers=16,
                                                  encoder=enc, device="cuda:0", force_reencode=True, gpu_decode=True, packed_store=store,
                                                  decode_chunk=knobs.pop("decode_chunk", 2048))
                for k, val in knobs.items():
                    setattr(ds, k, val)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                n_emb = ds.process()[0]
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
            m = {k: round(t - ds.marks["start"], 3) for k, t in ds.marks.items() if k != "start"} if hasattr(ds, "marks") else {}
            print(f"rep {rep} {v:40s} {n_emb / dt:8.1f} images/s  {dt:.3f} s  {m}", flush=True)
            if os.environ.get("STAGER_LOG") and hasattr(ds, "stager_log"):
                for row in ds.stager_log:
                    print("      ", row)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
