#!/usr/bin/env python3
"""Developer: the diversity-ordering walk (500 steps, 100 candidates per step, the reference's defaults) on N stored
ViT-L/14 embeddings already in HBM, next to the oracle (the reference's torch-CPU arithmetic without its 50 000 file loads)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import diversity_driver
from oracle import diversity_oracle
dev = torch.device("cuda", 0)
for n in (10_000, 100_000, 1_000_000):
    g = torch.Generator().manual_seed(n)
    c = torch.randn(64, 768, generator=g)
    emb = (c[torch.randint(0, 64, (n,), generator=g)] + 0.5 * torch.randn(n, 768, generator=g))
    samples = diversity_oracle.draw_samples(n, 500, 100, seed=1)
    e = emb.to(dev)
    diversity_driver.diversity_order_indices(e, samples[:5])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = diversity_driver.diversity_order_indices(e, samples).cpu().tolist()
    t1 = time.perf_counter() - t0
    line = f"n={n}: 500 steps in {t1*1e3:.1f} ms = {n*768*4*500/t1/1e12:.2f} TB/s of embedding rows scanned"
    if n <= 100_000:
        t0 = time.perf_counter(); want = diversity_oracle.diversity_order(emb.numpy(), samples); t2 = time.perf_counter() - t0
        line += f"; oracle (torch CPU, embeddings in RAM) {t2*1e3:.0f} ms; identical walk: {got == want}"
    print(line, flush=True)
