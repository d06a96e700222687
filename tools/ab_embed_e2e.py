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
ap.add_argument("--content", default="noise", choices=["noise", "photo"],
                help="noise: uniform noise, ~230 KB per 512x512 file, the entropy decoder's worst case; photo: low-pass structure + "
                     "sensor-like noise, the size a camera JPEG of that many pixels has")
a = ap.parse_args()
tmp = tempfile.mkdtemp(prefix="ab_e2e_")
try:
    rs = np.random.RandomState(0)
    if a.content == "noise":
        base = rs.randint(0, 256, (a.size, a.size, 3), dtype=np.uint8)
    else:
        coarse = Image.fromarray(rs.randint(0, 256, (a.size // 32, a.size // 32, 3), dtype=np.uint8)).resize((a.size, a.size), Image.BICUBIC)
        mid = Image.fromarray(rs.randint(96, 160, (a.size // 4, a.size // 4, 3), dtype=np.uint8)).resize((a.size, a.size), Image.BICUBIC)
        base = np.clip(np.asarray(coarse, np.float32) + (np.asarray(mid, np.float32) - 128) + rs.normal(0, 3, (a.size, a.size, 3)), 0, 255).astype(np.uint8)
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda i: Image.fromarray(np.roll(base, i * 7, axis=1)).save(os.path.join(tmp, f"{i:06d}.jpg"), quality=90), range(a.n)))
    print(f"{a.n} files of {a.size}x{a.size} ({a.content}), {sum(os.path.getsize(os.path.join(tmp, f)) for f in os.listdir(tmp)) / a.n / 1024:.0f} KB each", flush=True)
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
