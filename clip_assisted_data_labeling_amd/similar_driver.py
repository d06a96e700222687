"""Similarity-search driver: the counterpart of /root/reference/tools/find_similar_imgs.py with the distance scan and
the top-N selection on the HIP kernels (libclipenc_hip.so: simsearch_distances, simsearch_topn).

Keeps: the flags (:142-151), the context embedding = mean over the context directory of each file's concatenated
[model][crop] embedding (:19-62, `--clip_models_to_use all` = the models of the first file), the candidate rule (a `.pt`
with a `.jpg` next to it whose name is not in the context set, :106-109), "l2" / "cosine" distances (:88-94), root
directories of several context sets (:154-158), the `_similar` output folder and `{distance:.3f}_{stem}.jpg` names
(:165-172).  All candidates are scored in one launch instead of one `torch.load` + Python distance per file; results
come back ascending (the reference lists its top-N in slot order; the set is the same whenever distances differ).
With `--packed_store DIR` the search set is read from the packed shards (embed_driver --packed_store).
"""
from __future__ import annotations

import argparse
import os
import shutil
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

MEASURES = {"l2": 0, "cosine": 1}


def get_filepaths(root_dir, extension=(".pt",)):
    out = []
    for root, _, files in os.walk(root_dir):
        for f in files:
            if f.endswith(tuple(extension)):
                out.append(os.path.join(root, f))
    return out


def _file_features(path: str, clip_models: Sequence[str], crop_name: str) -> torch.Tensor:
    full = torch.load(path, map_location="cpu", weights_only=True)
    return torch.cat([full[m][crop_name].flatten() for m in clip_models], dim=0)


def create_context_embedding(args, context_dir) -> Tuple[np.ndarray, List[str]]:
    feats, names, skips = [], [], 0
    for p in get_filepaths(context_dir):
        try:
            if args.clip_models_to_use[0] == "all":
                args.clip_models_to_use = list(torch.load(p, map_location="cpu", weights_only=True).keys())
                print(f"\n----> Using all found clip models: {args.clip_models_to_use}")
            feats.append(_file_features(p, args.clip_models_to_use, args.crop_name_to_use))
            names.append(Path(p).name)
        except Exception as e:                       # simply skip the sample if something goes wrong (:52-55)
            print(e)
            skips += 1
    print(f"Loaded {len(feats)} samples from {context_dir}")
    if skips:
        print(f"(skipped {skips} samples due to loading errors)..")
    if not feats:
        raise RuntimeError(f"no usable embeddings in {context_dir}")
    return torch.stack(feats, 0).float().mean(0).numpy(), names


@torch.no_grad()
def nearest(emb: torch.Tensor, query: np.ndarray, measure: str, top_n: int, device="cuda",
            row_offset: int = 0, row_stride: Optional[int] = None, d: Optional[int] = None):
    """emb: [n, ...] float32 / float16 tensor (moved to `device` if needed); row i = emb.flatten(1)[i, row_offset:row_offset+d].
    Returns (indices int64 [k], distances float32 [k]) of the k = min(top_n, n) nearest rows, ascending."""
    if measure not in MEASURES:
        raise NotImplementedError(f"Similarity measure {measure} not implemented!")
    lib = _lib.load()
    dev = torch.device(device)
    x = emb.to(dev)
    if x.dtype not in (torch.float32, torch.float16):
        x = x.float()
    x = x.contiguous()
    n = x.shape[0]
    flat = x.view(n, -1) if n else x.reshape(0, 0)
    stride = row_stride if row_stride is not None else (flat.shape[1] if n else 0)
    d = d if d is not None else stride - row_offset
    if n == 0 or top_n < 1:
        return np.zeros(0, np.int64), np.zeros(0, np.float32)
    q = torch.from_numpy(np.ascontiguousarray(query, dtype=np.float32)).to(dev)
    if q.numel() != d:
        raise ValueError(f"query has {q.numel()} elements, rows have {d}")
    dist = torch.empty(n, dtype=torch.float32, device=dev)
    st = _lib.current_stream_ptr(dev)
    base = flat.data_ptr() + row_offset * flat.element_size()
    _lib.check(lib.simsearch_distances(base, 1 if x.dtype == torch.float16 else 0, n, d, stride, q.data_ptr(), MEASURES[measure],
                                       dist.data_ptr(), st), "simsearch_distances")
    k = min(top_n, n)
    ws_bytes = lib.simsearch_topn_workspace(n, k)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    idx = torch.empty(k, dtype=torch.int64, device=dev)
    val = torch.empty(k, dtype=torch.float32, device=dev)
    _lib.check(lib.simsearch_topn(dist.data_ptr(), n, k, idx.data_ptr(), val.data_ptr(), ws.data_ptr(), ws_bytes, st), "simsearch_topn")
    return idx.cpu().numpy(), val.cpu().numpy()


def _search_set_from_pt(args, context_names):
    paths, feats, skips = [], [], 0
    for p in get_filepaths(args.search_dir):
        img = p[:-3] + ".jpg"
        if not os.path.exists(img) or Path(img).name in context_names:      # :106-109
            continue
        try:
            feats.append(_file_features(p, args.clip_models_to_use, args.crop_name_to_use))
            paths.append(img)
        except Exception as e:
            print(e)
            skips += 1
    if skips:
        print(f"(skipped {skips} samples due to loading errors)..")
    return paths, (torch.stack(feats, 0).float() if feats else torch.zeros(0, 0))


def _search_set_from_store(args, context_names):
    from .packed_store import PackedStore
    store = PackedStore(args.packed_store)
    blocks, keys0 = [], None
    for m in args.clip_models_to_use:
        keys, data, crops = store.load(m)
        if keys0 is None:
            keys0 = keys
        elif keys != keys0:
            raise RuntimeError("packed store: the requested models do not cover the same images in the same order")
        blocks.append(np.asarray(data[:, crops.index(args.crop_name_to_use), :]))
    mat = np.concatenate(blocks, axis=1) if len(blocks) > 1 else blocks[0]
    keep, paths = [], []
    for i, k in enumerate(keys0):
        img = os.path.join(args.search_dir, k.replace("/", os.sep) + ".jpg")
        if os.path.exists(img) and Path(img).name not in context_names:
            keep.append(i)
            paths.append(img)
    return paths, torch.from_numpy(np.ascontiguousarray(mat[keep]))


def find_similar_imgs(args, context_embedding: np.ndarray, context_names, device="cuda"):
    """[(img_path, distance)] of the args.top_n nearest search images, ascending."""
    names = set(n[:-3] + ".jpg" for n in context_names)                      # the reference compares jpg names to pt names:
    names |= set(context_names)                                             # (:109 never matches; both spellings excluded here)
    if getattr(args, "packed_store", None):
        paths, feats = _search_set_from_store(args, names)
    else:
        paths, feats = _search_set_from_pt(args, names)
    print(f"Searched through {len(paths)} samples from {args.search_dir}")
    idx, dist = nearest(feats, context_embedding, args.similarity_measure, args.top_n, device)
    return [(paths[i], float(v)) for i, v in zip(idx, dist) if i >= 0]


def main(argv=None):
    parser = argparse.ArgumentParser(description="Find similar images between the context and search directories using pre-computed CLIP embeddings")
    parser.add_argument("--context_dir", help="Directory to learn img context from")
    parser.add_argument("--search_dir", help="Directory to find similar imgs in")
    parser.add_argument("--output_dir", default=None, help="Directory to copy selected files to (default: <context_dir>/_similar)")
    parser.add_argument("--clip_models_to_use", metavar="S", type=str, nargs="+", default=["all"], help="Which CLIP model embeddings to use, default: use all found")
    parser.add_argument("--crop_name_to_use", default="square_padded_crop", help="From which img crop to use the CLIP embedding")
    parser.add_argument("--similarity_measure", default="l2", help="Similarity measure to use in CLIP-space (cosine or l2)")
    parser.add_argument("--top_n", default=30, type=int, help="How many similar images to find")
    parser.add_argument("--packed_store", type=str, default=None, help="Read the SEARCH set from packed shards (keys relative to --search_dir)")
    args = parser.parse_args(argv)
    if not any(f.endswith(".pt") for f in os.listdir(args.context_dir)):     # a root of several context sets (:154-158)
        context_dirs = [os.path.join(args.context_dir, d) for d in sorted(os.listdir(args.context_dir))]
    else:
        context_dirs = [args.context_dir]
    for context_dir in context_dirs:
        emb, names = create_context_embedding(args, context_dir)
        out_dir = args.output_dir or os.path.join(context_dir, "_similar")
        Path(out_dir).mkdir(parents=True, exist_ok=True)
        for img_path, distance in find_similar_imgs(args, emb, names):
            shutil.copy(img_path, os.path.join(out_dir, f"{distance:.3f}_{Path(img_path).stem}.jpg"))


if __name__ == "__main__":
    main()
