#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): images/sec (4 crops each), ViT-L/14 encode + score @ 512 images per GPU.

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
clipenc_encode_score on 512 images x 4 crops of 224x224 (bf16 MFMA encode, fused fp32 regressor
3072->264->128->64->1 on the 4 crop embeddings), followed for N > 1 by the RCCL all-gather of the
embeddings and scores.  Images shard across ranks (weak scaling: 512 images per GPU).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2516.6      # MI355X dense bf16 MFMA: 256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz
PEAK_FP8_TFLOPS = 5033.2       # dense e4m3 on the block-scaled MFMA: 2x the bf16 rate per clock (MI355X_MICROARCH.md)
MODEL = "ViT-L-14"
IMAGES_PER_GPU = 512
CROPS_PER_IMAGE = 4
REG_SIZES = [4 * 768, 264, 128, 64, 1]


def fc_weights(seed):
    rs = np.random.RandomState(seed)
    Ws, bs = [], []
    for i in range(len(REG_SIZES) - 1):
        bound = 1.0 / np.sqrt(REG_SIZES[i])
        Ws.append(rs.uniform(-bound, bound, (REG_SIZES[i + 1], REG_SIZES[i])).astype(np.float32))
        bs.append(rs.uniform(-bound, bound, (REG_SIZES[i + 1],)).astype(np.float32))
    return Ws, bs


def synthetic_crops(n, size, seed, device):
    """uint8-valued pixels through the CLIP normalisation, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    u = torch.randint(0, 256, (n, 3, size, size), generator=g, device=device, dtype=torch.int32).float()
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], device=device).view(1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711], device=device).view(1, 3, 1, 1)
    return ((u / 255.0 - mean) / std).contiguous()


def cpu_baseline(cfg, sd, Ws, bs):
    """The oracle (a port of the reference's CPU encode_image + SimpleFC) timed on this box's host cores
    on a bounded sample of the same workload."""
    from oracle import fcreg_oracle, vit_oracle          # checker only: never on the product path
    cores = min(os.cpu_count() or 1, 32)      # more threads than this make torch's fp32 GEMMs at these sizes slower
    torch.set_num_threads(cores)
    n_img = 4
    crops = synthetic_crops(n_img * CROPS_PER_IMAGE, cfg.image_size, 99, "cpu")
    vit_oracle.encode_image(sd, cfg, crops[:4])           # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        emb = vit_oracle.encode_image(sd, cfg, crops)
        fcreg_oracle.forward_np(Ws, bs, emb.reshape(n_img, -1).numpy())
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= 4:
            break
    return {"value": round(n_img * reps / el, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x {n_img} images x 4 crops, fp32 torch CPU restatement (oracle/), {el:.1f} s"}


def pmc_traffic(kernel_name):
    """HBM-side bytes per launch of `kernel_name` from the latest committed rocprofv3 PMC passes
    (profiles/*/pmc_hbm_traffic_per_kernel.json: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 read
    correction, written by tools/summarize_profiles.py); None when no profile has been committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_hbm_traffic_per_kernel.json")))
    if not files:
        return None
    for path in reversed(files):                      # newest round first; bf16 and fp8 runs are summarised separately
        try:
            for k, v in json.load(open(path)).items():
                if kernel_name in k:
                    return round(float(v.get("bytes_per_launch", v.get("hbm_bytes", 0.0))), 1)
        except Exception:
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--images", type=int, default=IMAGES_PER_GPU, help="images per GPU per step (default 512)")
    ap.add_argument("--chunk", type=int, default=0, help="crops per pass through the layer chain (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["bf16", "fp8"], default="bf16",
                    help="arithmetic of the block GEMMs: bf16 = the headline (configs[1]+[2]); fp8 = configs[3] (e4m3 MFMA)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    import torch.distributed as dist
    from clip_assisted_data_labeling_amd import vit_config
    from clip_assisted_data_labeling_amd.embedder import HipViT
    from clip_assisted_data_labeling_amd.nn_model import HipRegressor

    # BENCH_DEVICE / BENCH_BACKEND exist only to exercise the N > 1 control flow on a 1-GPU box (all ranks on one
    # device, gloo with host staging); the real multi-GPU run uses one GPU per rank and RCCL.
    dev_index = int(os.environ.get("BENCH_DEVICE", local_rank))
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    cfg = vit_config.ARCHS[MODEL]
    sd = vit_config.seeded_state_dict(cfg, 0)               # random-init weights of the named architecture
    Ws, bs = fc_weights(1)
    vit = HipViT(cfg, sd, dev, chunk_crops=args.chunk or None, precision=args.dtype)
    reg = HipRegressor([torch.from_numpy(w) for w in Ws], [torch.from_numpy(b) for b in bs], 0.01, dev)
    n_img = args.images
    crops = synthetic_crops(n_img * CROPS_PER_IMAGE, cfg.image_size, 1234 + rank, dev)   # resident in HBM
    sel = list(range(CROPS_PER_IMAGE))
    gdev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        emb_all = torch.empty((world * n_img, CROPS_PER_IMAGE, cfg.embed_dim), device=gdev)
        score_all = torch.empty((world * n_img, 1), device=gdev)

    def step():
        emb, score = vit.encode_score(crops, reg, CROPS_PER_IMAGE, sel)
        if world > 1:                                       # the one exchange of the path: gather results
            dist.all_gather_into_tensor(emb_all, emb.to(gdev))
            dist.all_gather_into_tensor(score_all, score.to(gdev))
        return emb, score

    for _ in range(args.warmup):
        step()
    vit.profile_enable(True)                                # HIP events around every kernel, on the launch stream

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        emb, score = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=gdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = vit.profile_read()
    vit.profile_enable(False)
    assert torch.isfinite(emb).all() and torch.isfinite(score).all()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * n_img * args.steps / elapsed
        flop_per_image = 2.0 * cfg.macs_per_crop() * CROPS_PER_IMAGE + 2.0 * sum(
            REG_SIZES[i] * REG_SIZES[i + 1] for i in range(len(REG_SIZES) - 1))
        # dominant kernel = largest share of the step among the device kernels (names as rocprofv3 prints them)
        DOMINANT = max((k for k in prof if not k.startswith("shape:")), key=lambda k: prof[k][0])
        d_ms, d_n, d_fl = prof[DOMINANT]
        achieved = d_fl / (d_ms * 1e-3) / 1e12 if d_ms > 0 else 0.0
        fp8 = args.dtype == "fp8"
        d_peak = PEAK_FP8_TFLOPS if "fp8" in DOMINANT else PEAK_BF16_TFLOPS
        workload = ("BASELINE.json configs[3] on one GPU's shard: ViT-L/14 @224 encode with e4m3 MFMA block GEMMs "
                    "(per-token x per-channel scales), bf16 attention/residual" if fp8 else
                    "BASELINE.json configs[1]+[2]: ViT-L/14 @224 bf16 encode")
        line = {
            "metric": "images/sec (4 crops each) ViT-L/14 encode+score @ bs512",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{workload} of {n_img} images x 4 crops per GPU "
                                   "+ fused fp32 regressor 3072-264-128-64-1, seeded random-init weights, crops resident in HBM",
                       "images_per_gpu": n_img, "crops_per_image": CROPS_PER_IMAGE, "parallelism": f"image-sharded x{world}",
                       "chunk_crops": args.chunk or 2048},
            "end_to_end": {"tflops": round(value * flop_per_image / 1e12, 1),
                           "frac_of_bf16_peak": round(value * flop_per_image / 1e12 / (PEAK_BF16_TFLOPS * world), 4)},
            "roofline": {"bound": "mfma", "kernel": DOMINANT, "achieved": round(achieved, 1), "peak": d_peak,
                         "unit": "TFLOP/s", "frac": round(achieved / d_peak, 4), "traffic": pmc_traffic(DOMINANT),
                         "launches": d_n, "avg_launch_ms": round(d_ms / max(d_n, 1), 4),
                         "algorithmic_flop_per_launch": d_fl / max(d_n, 1)},
            "kernels_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in prof.items() if not k.startswith("shape:")},
            "kernels_tflops": {k.replace("shape:", ""): round(v[2] / (v[0] * 1e-3) / 1e12, 1) for k, v in prof.items()
                               if v[0] > 0 and v[2] > 1e12},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, Ws, bs)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
