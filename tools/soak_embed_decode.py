#!/usr/bin/env python3
"""Developer soak: the embed driver over a few thousand MIXED files -- baseline / progressive / grey / CMYK / multi-scan JPEG of
random sizes and qualities, restart markers, PNG, truncated and garbage files -- once with host decoding (DataLoader workers,
Pillow) and once with --gpu_decode (and once more with progressive files on the device): the packed stores must hold the same
rows bit for bit and the same files must be counted unreadable."""
import io, os, shutil, sys, tempfile, time
import numpy as np, torch
from PIL import Image, ImageFile
ImageFile.MAXBLOCK = 1 << 24
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import embed_driver, packed_store
from clip_assisted_data_labeling_amd.embedder import CLIP_Encoder
from tests.jpeg_writer import random_coefs, tables_from_pillow, write_sequential_scans

n = int(os.environ.get("N", "3000")); MODEL = os.environ.get("MODEL", "ViT-B-32/seed0")
rs = np.random.RandomState(int(os.environ.get("SEED", "1")))
tmp = tempfile.mkdtemp(prefix="soak_e2e_")
kinds = {}
try:
    root = os.path.join(tmp, "imgs"); os.makedirs(root)
    dqt, dht = tables_from_pillow(85)
    for i in range(n):
        w, h = int(rs.randint(8, 900)), int(rs.randint(8, 900))
        yy, xx = np.mgrid[0:h, 0:w]
        arr = np.clip(128 + 90 * np.sin(xx / rs.uniform(3, 40) + yy / rs.uniform(3, 40))[..., None] + rs.randn(h, w, 3) * rs.uniform(0, 60), 0, 255).astype(np.uint8)
        img = Image.fromarray(arr)
        r = rs.rand()
        name = os.path.join(root, f"{i:06d}.jpg")
        kw = dict(quality=int(rs.randint(20, 100)), subsampling=int(rs.randint(0, 3)))
        if r < 0.62: kind = "baseline"; img.save(name, **kw, optimize=bool(rs.rand() < 0.3))
        elif r < 0.70: kind = "restart"; img.save(name, **kw, restart_marker_blocks=int(rs.randint(1, 30)))
        elif r < 0.82: kind = "progressive"; img.save(name, **kw, progressive=True)
        elif r < 0.86: kind = "grey"; img.convert("L").save(name, quality=kw["quality"])
        elif r < 0.89: kind = "cmyk"; img.convert("CMYK").save(name, quality=kw["quality"])
        elif r < 0.92:
            kind = "multiscan"; samp = [(2, 2), (1, 1), (1, 1)]
            open(name, "wb").write(write_sequential_scans(w, h, samp, random_coefs(rs, w, h, samp), dqt, dht, [[0], [1, 2]], restart=int(rs.randint(0, 5))))
        elif r < 0.95: kind = "png"; name = name[:-4] + ".png"; img.save(name)
        elif r < 0.98:
            kind = "truncated"; b = io.BytesIO(); img.save(b, "JPEG", **kw); blob = b.getvalue()
            open(name, "wb").write(blob[: int(len(blob) * rs.uniform(0.2, 0.95))])
        else: kind = "garbage"; open(name, "wb").write(bytes(rs.randint(0, 256, int(rs.randint(0, 3000)), dtype=np.uint8)))
        kinds[kind] = kinds.get(kind, 0) + 1
    print("files:", kinds, flush=True)
    enc = CLIP_Encoder(MODEL, None, device="cuda:0")
    stores = {}
    for tag, kw in (("host", dict(gpu_preprocess=True, num_workers=16)), ("gpu", dict(gpu_decode=True, num_workers=8)), ("gpu+prog", dict(gpu_decode=True, num_workers=8))):
        out = os.path.join(tmp, "store_" + tag.replace("+", "_"))
        ds = embed_driver.Feature_Dataset(root, MODEL, 128, shuffle_filenames=False, encoder=enc, device="cuda:0", force_reencode=True,
                                          packed_store=out, **kw)
        if tag == "gpu+prog": ds.gpu_decode_progressive = True
        t0 = time.perf_counter(); res = ds.process(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        keys, data, _ = packed_store.PackedStore(out).load(MODEL)
        stores[tag] = (res, dict(zip(keys, range(len(keys)))), np.array(data))
        print(f"{tag}: {res} in {dt:.1f} s = {res[0] / dt:.0f} images/s, {len(keys)} rows", flush=True)
    (r0, k0, d0) = stores["host"]
    ok = True
    for tag in ("gpu", "gpu+prog"):
        (r1, k1, d1) = stores[tag]
        same = r1 == r0 and set(k1) == set(k0)
        if same:
            for k, i in k1.items():
                if not np.array_equal(d1[i].view(np.uint32), d0[k0[k]].view(np.uint32)):
                    same = False; print("differs:", k); break
        print(f"{tag} store == host store: {same}", flush=True)
        ok = ok and same
    sys.exit(0 if ok else 1)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
