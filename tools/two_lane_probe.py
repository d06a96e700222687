#!/usr/bin/env python3
"""Developer probe: does running the batch as L independent sub-batches on L HIP streams fill the tail rounds of the persistent
kernels (8 224 tiles on 256 CUs = 32.125 rounds for out-proj / FC2) and the launch gaps between dependent kernels?

Each lane is its own handle (own workspace) on its own stream; lane i encodes crops [i * n / L, (i + 1) * n / L).  While lane A's
kernel is in its last, partly filled round, lane B's next kernel is dispatched onto the CUs that became free.  Compared with the
one-handle, one-stream step on the same box, interleaved.
    python tools/two_lane_probe.py [--dtype fp8] [--lanes 1,2,3,4] [--reps 3] [--steps 5]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_assisted_data_labeling_amd import vit_config  # noqa: E402
from clip_assisted_data_labeling_amd.embedder import HipViT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--lanes", default="1,2,4")
    ap.add_argument("--crops", type=int, default=2048)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--same-stream", action="store_true", help="lanes run one after the other on ONE stream (cost of the split alone)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = vit_config.ARCHS["ViT-L-14"]
    sd = vit_config.seeded_state_dict(cfg, 0)
    n = args.crops
    g = torch.Generator(device=dev).manual_seed(1)
    crops = torch.randn(n, 3, 224, 224, device=dev, generator=g)
    lane_counts = [int(x) for x in args.lanes.split(",")]
    sets = {}
    for L in lane_counts:
        per = [(n * i // L, n * (i + 1) // L) for i in range(L)]
        vits = [HipViT(cfg, sd, dev, chunk_crops=hi - lo, precision=args.dtype) for lo, hi in per]
        streams = [torch.cuda.Stream(device=dev) for _ in per]
        sets[L] = (per, vits, streams)

    def step(L):
        per, vits, streams = sets[L]
        outs = []
        if L == 1 or args.same_stream:
            for (lo, hi), v in zip(per, vits):
                outs.append(v.encode(crops[lo:hi]))
            return outs
        cur = torch.cuda.current_stream(dev)
        ev = torch.cuda.Event()
        ev.record(cur)
        for (lo, hi), v, s in zip(per, vits, streams):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                outs.append(v.encode(crops[lo:hi]))
        for s in streams:
            cur.wait_stream(s)
        return outs

    ref = torch.cat(step(1))
    for L in lane_counts:
        out = torch.cat(step(L))
        torch.cuda.synchronize()
        print(f"lanes {L}: bitwise equal to one lane: {bool(torch.equal(out, ref))}")
    for rep in range(args.reps):
        for L in lane_counts:
            step(L)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(L)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            print(f"rep {rep} lanes {L}: {dt * 1e3:8.2f} ms per step  {n / 4 / dt:8.1f} images/s", flush=True)


if __name__ == "__main__":
    main()
