#!/usr/bin/env python3
"""Developer timing: per-image `.pt` store (the reference's format) vs the packed store, host side only."""
import os, shutil, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd.packed_store import PackedStore, PackedStoreWriter
from clip_assisted_data_labeling_amd.preprocess import CROP_NAMES

n, E, B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 768, 512
tmp = tempfile.mkdtemp(prefix="storebench_")
emb = torch.randn(n, 4, E)
try:
    t = time.perf_counter()
    for i in range(n):
        torch.save({"ViT-L-14/openai": {c: emb[i, j:j + 1].clone() for j, c in enumerate(CROP_NAMES)}}, os.path.join(tmp, f"{i:07d}.pt"))
    w_pt = n / (time.perf_counter() - t)
    t = time.perf_counter()
    for i in range(n):
        d = torch.load(os.path.join(tmp, f"{i:07d}.pt"), map_location="cpu", weights_only=True)
        torch.cat([d["ViT-L-14/openai"][c] for c in CROP_NAMES], 0).flatten()
    r_pt = n / (time.perf_counter() - t)
    sd = os.path.join(tmp, "store")
    t = time.perf_counter()
    with PackedStoreWriter(sd, "ViT-L-14/openai", CROP_NAMES, E) as w:
        for b0 in range(0, n, B):
            w.append([f"{i:07d}" for i in range(b0, min(n, b0 + B))], emb[b0:b0 + B])
    w_pk = n / (time.perf_counter() - t)
    t = time.perf_counter()
    found, mat = PackedStore(sd).features(["ViT-L-14/openai"], CROP_NAMES, [f"{i:07d}" for i in range(n)])
    r_pk = n / (time.perf_counter() - t)
    assert found.all() and np.array_equal(mat, emb.reshape(n, -1).numpy())
    print(f"{n} images x 4 crops x {E}: per-image .pt write {w_pt:,.0f} img/s, read {r_pt:,.0f} img/s | "
          f"packed write {w_pk:,.0f} img/s, read+assemble {r_pk:,.0f} img/s (one process, this host's disk)")
finally:
    shutil.rmtree(tmp)
