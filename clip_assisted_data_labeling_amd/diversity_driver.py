"""Diversity ordering of a fresh labeling session: the counterpart of `diversity_ordered_image_files`
(/root/reference/_3_label_images.py:135-177) with the sampled farthest-point walk on the HIP kernels
(libclipenc_hip.so: diversity_order).

Keeps: the signature and defaults (image_files, root_directory, total_n_ordered_imgs=500, sample_size=100), the start
at image_files[0] (:141), `min(total_n_ordered_imgs, len(image_files) - 1)` steps (:146), one
`random.sample(image_files, sample_size)` per step drawn from Python's global RNG exactly as the reference draws it
(:148; it raises ValueError when sample_size > len(image_files), as there), the "square_padded_crop" embedding of
`<root_directory>/<basename>.pt` (:142, :151-154), the appended image = first minimum of the column maxima (:161-167), and
the tail = every file not chosen, in its original order (:174).

Differs: the reference opens 100 `.pt` files per step (50 000 `torch.load`s for the default walk); here every embedding is
read ONCE (from the `.pt` files or the packed store), lives in HBM, and the whole walk is one asynchronous chain of
kernels whose state never leaves the device.  `.pt` files may hold the crop at the top level (what :142 indexes) or under
the model name (what _1_embed_with_CLIP.py writes); files without a usable embedding raise, as `torch.load` would there.
There is no CPU path: without the HIP library or a GPU this raises.
"""
from __future__ import annotations

import os
import random
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib

CROP = "square_padded_crop"


def _pt_path(image_file: str, root_directory: str) -> str:
    return os.path.join(root_directory, os.path.basename(image_file).replace(".jpg", ".pt"))     # :142


def _load_pt_embedding(path: str, crop_name: str, model_name: Optional[str]) -> np.ndarray:
    d = torch.load(path, map_location="cpu", weights_only=True)
    if crop_name in d:                                     # the layout :142 indexes
        t = d[crop_name]
    else:                                                  # Appendix B.1 of SURVEY.md: {model_name: {crop: [1, E]}}
        m = model_name if model_name is not None else next(k for k, v in d.items() if isinstance(v, dict) and crop_name in v)
        t = d[m][crop_name]
    return t.squeeze().float().numpy()


def load_embeddings(image_files: Sequence[str], root_directory: str, crop_name: str = CROP, model_name: Optional[str] = None,
                    packed_store: Optional[str] = None) -> np.ndarray:
    """float32 [len(image_files), E], row i = the crop embedding of image_files[i]."""
    if packed_store:
        from .packed_store import PackedStore, image_key
        store = PackedStore(packed_store)
        model = model_name or store.models()[0]
        keys, data, crop_names = store.load(model)
        pos = {k: i for i, k in enumerate(keys)}
        c = crop_names.index(crop_name)
        rows = [pos[image_key(os.path.join(root_directory, os.path.basename(f)), root_directory)] for f in image_files]
        return np.ascontiguousarray(np.asarray(data[:, c, :])[rows], dtype=np.float32)
    return np.stack([_load_pt_embedding(_pt_path(f, root_directory), crop_name, model_name) for f in image_files]).astype(np.float32)


def diversity_order_indices(emb: torch.Tensor, samples: np.ndarray, first: int = 0) -> torch.Tensor:
    """emb: float32 [n, E] on a cuda device (rows may be a strided view of the packed [n, crops, E] block);
    samples: int32 [steps, sample_size].  Returns the int32 [steps] index appended at each step (device tensor)."""
    if not emb.is_cuda:
        raise _lib.ClipencError("diversity ordering runs on the HIP kernels only: pass a cuda tensor")
    if emb.dtype != torch.float32 or emb.dim() != 2 or emb.stride(1) != 1:
        raise ValueError("emb must be float32 [n, E] with contiguous rows")
    lib = _lib.load()
    n, d = emb.shape
    samples = np.ascontiguousarray(samples, dtype=np.int32)
    steps, k = (samples.shape if samples.ndim == 2 else (0, 1))
    if steps and (samples.min() < 0 or samples.max() >= n):
        raise ValueError("sample index outside [0, n)")
    dev = emb.device
    order = torch.empty((max(steps, 1),), dtype=torch.int32, device=dev)
    if steps == 0:
        return order[:0]
    s_dev = torch.from_numpy(samples).to(dev)
    ws_bytes = int(lib.diversity_workspace(n))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    _lib.check(lib.diversity_order(emb.data_ptr(), n, d, emb.stride(0), int(first), s_dev.data_ptr(), steps, k, order.data_ptr(),
                                   ws.data_ptr(), ws_bytes, _lib.current_stream_ptr(dev)), "diversity_order")
    torch.cuda.current_stream(dev).synchronize()          # s_dev / ws are released on return
    return order


@torch.no_grad()
def diversity_ordered_image_files(image_files, root_directory, total_n_ordered_imgs=500, sample_size=100, crop_name: str = CROP,
                                  model_name: Optional[str] = None, packed_store: Optional[str] = None, device="cuda") -> List[str]:
    """Tries to order the first total_n_ordered_imgs in a way that maximizes the diversity of that set in CLIP space."""
    image_files = list(image_files)
    steps = min(total_n_ordered_imgs, len(image_files) - 1)                                     # :146
    print("Creating the most CLIP-diverse ordering of the first ", total_n_ordered_imgs, " images...")
    # the reference draws one sample per step from the global RNG while it walks; drawing them up front consumes the
    # same numbers in the same order
    samples = np.array([random.sample(range(len(image_files)), sample_size) for _ in range(max(steps, 0))], dtype=np.int32)
    if steps <= 0:
        return image_files
    emb = torch.from_numpy(load_embeddings(image_files, root_directory, crop_name, model_name, packed_store)).to(device)
    order = diversity_order_indices(emb, samples.reshape(steps, sample_size), first=0).cpu().tolist()
    img_files = [image_files[0]] + [image_files[i] for i in order]                              # :141, :167
    taken = set(img_files)
    return img_files + [f for f in image_files if f not in taken]                               # :174
