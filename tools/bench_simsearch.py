#!/usr/bin/env python3
"""Developer timing of the similarity search kernels (HBM-bound scan + top-N)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
for (n, d, dt) in ((1_000_000, 768, torch.float32), (1_000_000, 768, torch.float16), (4_000_000, 768, torch.float16)):
    x = torch.randn(n, d, device=dev).to(dt)
    q = torch.randn(d, device=dev)
    dist = torch.empty(n, device=dev)
    top = 30
    wsb = lib.simsearch_topn_workspace(n, top)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev); idx = torch.empty(top, dtype=torch.int64, device=dev); val = torch.empty(top, device=dev)
    def run(m):
        lib.simsearch_distances(x.data_ptr(), 1 if dt == torch.float16 else 0, n, d, d, q.data_ptr(), m, dist.data_ptr(), st)
    def sel():
        lib.simsearch_topn(dist.data_ptr(), n, top, idx.data_ptr(), val.data_ptr(), ws.data_ptr(), wsb, st)
    for name, fn in (("l2", lambda: run(0)), ("cosine", lambda: run(1)), ("top-30", sel)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        extra = f"{n * d * x.element_size() / ms / 1e9:.2f} TB/s" if name != "top-30" else ""
        print(f"n={n} d={d} {str(dt)[6:]}: {name:7s} {ms:.3f} ms {extra}")
