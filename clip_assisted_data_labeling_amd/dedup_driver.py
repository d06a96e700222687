"""Near-duplicate driver: the counterpart of /root/reference/_2_remove_duplicates.py with the
similarity search on the HIP kernel (libclipenc_hip.so: dedup_find_pairs).

Keeps: the flags (:135-142), the per-sub-directory (jpg, pt) pairing and float16 cast (:8-49), the
default crop `square_padded_crop` / threshold 0.96 (:54, :137), row-major (i < j) pair order (:74-76),
the output folder name and `fix_duplicate` naming (:83-125), `--test` dry runs (:89).  The N x N
similarity matrix is never materialised, so `--chunk_size` can be the whole directory.
"""
from __future__ import annotations

import argparse
import os
import shutil
from typing import List, Tuple

import numpy as np
import torch

from . import _lib


def _packed_paths_and_embeddings(args, crop_to_use):
    """Same chunks as below, read from the packed shards (embed_driver --packed_store): one gather per sub-directory."""
    from .packed_store import PackedStore, image_key
    store = PackedStore(args.packed_store)
    if args.clip_model_to_use is None:
        args.clip_model_to_use = store.models()[0]
        print(f"\n ----> args.clip_model_to_use was not specified, defaulting to first found one: {args.clip_model_to_use} \n")
    keys, data, crop_names = store.load(args.clip_model_to_use)
    pos = {k: i for i, k in enumerate(keys)}
    c = crop_names.index(crop_to_use)
    for subdir, _, files in os.walk(args.root_dir):
        stems = sorted(os.path.splitext(f)[0] for f in files if f.endswith(".jpg"))
        rows = [(s, pos.get(image_key(os.path.join(subdir, s), args.root_dir), -1)) for s in stems]
        rows = [(s, r) for s, r in rows if r >= 0]
        for b0 in range(0, len(rows), args.chunk_size):
            part = rows[b0:b0 + args.chunk_size]
            emb = torch.from_numpy(np.asarray(data[[r for _, r in part], c, :])).to(torch.float16)      # :38
            yield [os.path.join(subdir, s + ".jpg") for s, _ in part], list(emb)


def get_paths_and_embeddings(args, crop_to_use):
    if getattr(args, "packed_store", None):
        yield from _packed_paths_and_embeddings(args, crop_to_use)
        return
    for subdir, _, files in os.walk(args.root_dir):
        stems = {}
        for f in files:
            stem, ext = os.path.splitext(f)
            stems.setdefault(stem, []).append(ext)
        paths, embeddings = [], []
        for stem in sorted(stems):
            exts = stems[stem]
            if ".jpg" in exts and ".pt" in exts:
                try:
                    d = torch.load(os.path.join(subdir, stem + ".pt"), map_location="cpu", weights_only=True)
                    if args.clip_model_to_use is None:
                        args.clip_model_to_use = list(d.keys())[0]
                        print(f"\n ----> args.clip_model_to_use was not specified, defaulting to first found one: {args.clip_model_to_use} \n")
                    emb = d[args.clip_model_to_use][crop_to_use].squeeze().to(torch.float16)      # :38
                    paths.append(os.path.join(subdir, stem + ".jpg"))
                    embeddings.append(emb)
                    if len(paths) == args.chunk_size:
                        yield paths, embeddings
                        paths, embeddings = [], []
                except Exception:
                    continue
        if paths:
            yield paths, embeddings


@torch.no_grad()
def near_duplicate_pairs(emb_fp16: torch.Tensor, threshold: float, device="cuda", fp16_compare: bool = True,
                         capacity: int = 1 << 20, screened: bool = True, candidate_capacity: int = 1 << 22) -> Tuple[np.ndarray, np.ndarray]:
    """[n, d] float16 -> (pairs int64 [P,2] sorted row-major with i < j, values float32 [P]).
    screened (default): the e4m3 screen + exact recheck of include/clipenc.h (dedup_find_pairs_screened) -- the same pairs and
    values as the exact float16 search, which it falls back to by itself when there are more than `candidate_capacity` candidates;
    screened=False: the exact search only (dedup_find_pairs)."""
    lib = _lib.load()
    dev = torch.device(device)
    x = emb_fp16.to(dev, torch.float16).contiguous()
    n, d = x.shape
    n_pad, d_pad = (n + 255) // 256 * 256, (d + 127) // 128 * 128
    ws = torch.empty(max(n_pad * d_pad, 1), dtype=torch.float16, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    sws = None
    if screened and n >= 2:
        nbytes = int(lib.dedup_screen_ws_bytes(n, d, candidate_capacity))
        sws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        sws_ptr = (sws.data_ptr() + 255) // 256 * 256
    while True:
        pairs = torch.empty((capacity, 2), dtype=torch.int64, device=dev)
        vals = torch.empty(capacity, dtype=torch.float32, device=dev)
        if sws is not None:
            _lib.check(lib.dedup_find_pairs_screened(x.data_ptr(), n, d, float(threshold), 1 if fp16_compare else 0, ws.data_ptr(), sws_ptr, nbytes,
                                                     candidate_capacity, pairs.data_ptr(), vals.data_ptr(), capacity, count.data_ptr(),
                                                     _lib.current_stream_ptr(dev)), "dedup_find_pairs_screened")
        else:
            _lib.check(lib.dedup_find_pairs(x.data_ptr(), n, d, float(threshold), 1 if fp16_compare else 0, ws.data_ptr(),
                                            pairs.data_ptr(), vals.data_ptr(), capacity, count.data_ptr(),
                                            _lib.current_stream_ptr(dev)), "dedup_find_pairs")
        c = int(count.item())
        if c <= capacity:
            break
        capacity = c                                        # overflow: the count is exact, rerun with room for all
    p = pairs[:c].cpu().numpy()
    v = vals[:c].cpu().numpy()
    order = np.lexsort((p[:, 1], p[:, 0]))
    return p[order], v[order]


def fix_duplicate(duplicate_index, img_paths, outdir, sim_value, mode):
    """_2_remove_duplicates.py:102-125."""
    dirname = os.path.dirname(img_paths[0])
    b1 = os.path.splitext(os.path.basename(img_paths[0]))[0]
    b2 = os.path.splitext(os.path.basename(img_paths[1]))[0]
    files1 = [os.path.join(dirname, f) for f in os.listdir(os.path.dirname(img_paths[0])) if b1 in f]
    files2 = [os.path.join(dirname, f) for f in os.listdir(os.path.dirname(img_paths[1])) if b2 in f]
    for f in files1:
        if mode == "copy":
            shutil.copy(f, os.path.join(outdir, f"{sim_value:.3f}_{duplicate_index:08d}_source_{os.path.basename(f)}"))
    for f in files2:
        if mode == "copy":
            shutil.copy(f, os.path.join(outdir, f"{sim_value:.3f}_{duplicate_index:08d}_target_{os.path.basename(f)}"))
        if mode == "move":
            os.rename(f, os.path.join(outdir, f"{sim_value:.3f}_{duplicate_index:08d}_target_{os.path.basename(f)}"))


def find_near_duplicates(args, crop_to_use="square_padded_crop", device="cuda") -> List[Tuple[str, str, float]]:
    found = []
    for paths, embeddings in get_paths_and_embeddings(args, crop_to_use):
        if not paths:
            continue
        e = torch.stack(embeddings)
        print(f"Got batch of embeddings of shape: {tuple(e.shape)}, searching near duplicates..")
        pairs, vals = near_duplicate_pairs(e, args.threshold, device)
        near = [(paths[i], paths[j]) for i, j in pairs.tolist()]
        output_dir = os.path.join(os.path.dirname(args.root_dir), f"near_duplicates_cosine_{args.threshold}")
        os.makedirs(output_dir, exist_ok=True)
        print(f"Found {len(near)} duplicates!")
        if near and not args.test:
            for k, (pp, sv) in enumerate(zip(near, vals.tolist())):
                fix_duplicate(k, pp, output_dir, sv, args.mode)
        found += [(a, b, float(s)) for (a, b), s in zip(near, vals.tolist())]
    return found


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--root_dir", type=str, help="Root directory of the dataset")
    parser.add_argument("--threshold", type=float, default=0.96, help="Cosine-similarity threshold for near-duplicate detection")
    parser.add_argument("--mode", type=str, default="copy", help="copy / move, Use copy to test the script, move after")
    parser.add_argument("--clip_model_to_use", type=str, default=None, help="Which CLIP model to use, if None, use the first one found")
    parser.add_argument("--chunk_size", type=int, default=1000000, help="Images compared at once (the HIP kernel never builds the N x N matrix)")
    parser.add_argument("--test", action="store_true", help="Test the script without doing anything")
    parser.add_argument("--packed_store", type=str, default=None,
                        help="Read embeddings from the packed shards in this directory (written by embed_driver --packed_store)")
    find_near_duplicates(parser.parse_args(argv))


if __name__ == "__main__":
    main()
