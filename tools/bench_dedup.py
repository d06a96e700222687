#!/usr/bin/env python3
"""Developer timing of BASELINE.json configs[4]: cosine all-pairs dedup on N x 768 fp16 embeddings (default 100 000): the exact
float16 search and the screened one (e4m3 screen + exact recheck), alternating on one box."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import _lib
from tools.quick_bench import timeit

ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=100000); ap.add_argument("--d", type=int, default=768)
ap.add_argument("--planted", type=int, default=1000); args = ap.parse_args()
lib = _lib.load(); dev = torch.device("cuda", 0); st = _lib.current_stream_ptr(dev)
g = torch.Generator(device=dev).manual_seed(7)
e = torch.randn(args.n, args.d, device=dev, generator=g)
src = torch.randperm(args.n - args.planted, device=dev, generator=g)[:args.planted]
e[args.n - args.planted:] = e[src] + 0.1 * torch.randn(args.planted, args.d, device=dev, generator=g)
e16 = e.half().contiguous()
n_pad, d_pad = (args.n + 255) // 256 * 256, (args.d + 127) // 128 * 128
ws = torch.empty(n_pad * d_pad, dtype=torch.float16, device=dev)
cap = 1 << 20
pairs = torch.empty((cap, 2), dtype=torch.int64, device=dev); vals = torch.empty(cap, dtype=torch.float32, device=dev)
count = torch.zeros(1, dtype=torch.int64, device=dev)
fn = lambda: lib.dedup_find_pairs(e16.data_ptr(), args.n, args.d, 0.96, 1, ws.data_ptr(), pairs.data_ptr(), vals.data_ptr(), cap, count.data_ptr(), st)
cand_cap = 1 << 22
nbytes = int(lib.dedup_screen_ws_bytes(args.n, args.d, cand_cap))
sws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev); sptr = (sws.data_ptr() + 255) // 256 * 256
fs = lambda: lib.dedup_find_pairs_screened(e16.data_ptr(), args.n, args.d, 0.96, 1, ws.data_ptr(), sptr, nbytes, cand_cap, pairs.data_ptr(), vals.data_ptr(),
                                           cap, count.data_ptr(), st)
flop = float(args.n) * (args.n - 1) * args.d
for rnd in range(3):
    ms_s = timeit(fs, iters=5, warmup=2)
    cs = int(count.item()); cands = int(sws[sptr - sws.data_ptr():sptr - sws.data_ptr() + 8].view(torch.int64).item())
    ms = timeit(fn, iters=5, warmup=2)
    print(f"round {rnd}: screened {ms_s:.2f} ms ({cs} pairs from {cands} candidates, {flop / ms_s / 1e9:.0f} TFLOP/s algorithmic)   exact {ms:.2f} ms "
          f"({int(count.item())} pairs)", flush=True)
c = int(count.item())
p = pairs[:c].cpu().numpy()
planted_found = int(((p[:, 1] >= args.n - args.planted)).sum())
print(f"dedup N={args.n} d={args.d}: {ms:.2f} ms, {c} pairs ({planted_found} touch planted rows), "
      f"{flop / ms / 1e9:.1f} TFLOP/s on the strict upper triangle ({flop/1e12:.2f} TFLOP)")
