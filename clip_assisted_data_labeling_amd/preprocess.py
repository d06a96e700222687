"""CPU-side data prep of the embed path: the 4-crop geometry and the CLIP validation transform.

Restates /root/reference/utils/embedder.py:184-251 (`CustomImageDataset.extract_crops`) and the
open_clip validation transform returned by `CLIP_Encoder.get_preprocess_transform`
(/root/reference/utils/embedder.py:90-92; constants visible at :121-124) with PIL + torch only —
torchvision is not available in this environment. SURVEY.md Appendix C documents the geometry.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
from PIL import Image

OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)

CROP_NAMES = ["centre_crop", "square_padded_crop", "subcrop1", "subcrop2"]   # _1_embed_with_CLIP.py:200


def _center_crop_box(w: int, h: int, cw: int, ch: int) -> Tuple[int, int, int, int]:
    # torchvision CenterCrop: top = int(round((h - ch) / 2.0)), left likewise (Python banker's rounding)
    top = int(round((h - ch) / 2.0))
    left = int(round((w - cw) / 2.0))
    return left, top, left + cw, top + ch


def crop_boxes(width: int, height: int, crop_names: Sequence[str] = CROP_NAMES):
    """Integer geometry of the crops of a width x height image.

    Returns a list of (name, kind, box): kind 'crop' -> box (left, top, right, bottom) in the image;
    kind 'pad' -> box (side, paste_x, paste_y): black square canvas with the image pasted.
    Mirrors utils/embedder.py:196-245 including the clipping at :233-236.
    """
    out = []
    if "centre_crop" in crop_names:
        s = min(width, height)
        out.append(("centre_crop", "crop", _center_crop_box(width, height, s, s)))
    if "square_padded_crop" in crop_names:
        s = max(width, height)
        out.append(("square_padded_crop", "pad", (s, (s - width) // 2, (s - height) // 2)))
    if any("subcrop1" in n for n in crop_names) or any("subcrop2" in n for n in crop_names):
        s1 = int((width * height * 0.15) ** 0.5)
        s2 = int((width * height * 0.1) ** 0.5)
        if width >= height:
            centers = [(width // 4, height // 2), (width // 4 * 3, height // 2)]
        else:
            centers = [(width // 2, height // 4), (width // 2, height // 4 * 3)]
        for name, (cx, cy), s in zip(["subcrop1", "subcrop2"], centers, [s1, s2]):
            if name in crop_names:
                left = max(0, cx - s // 2)
                top = max(0, cy - s // 2)
                right = min(width, left + s)
                bottom = min(height, top + s)
                if right - left > 0 and bottom - top > 0:
                    out.append((name, "crop", (left, top, right, bottom)))
    return out


def extract_crops(pil_img: Image.Image, crop_names: Sequence[str] = CROP_NAMES) -> Tuple[List[Image.Image], List[str]]:
    crops, names = [], []
    for name, kind, box in crop_boxes(pil_img.width, pil_img.height, crop_names):
        if kind == "crop":
            crops.append(pil_img.crop(box))
        else:
            side, px, py = box
            canvas = Image.new("RGB", (side, side), (0, 0, 0))
            canvas.paste(pil_img, (px, py))
            crops.append(canvas)
        names.append(name)
    return crops, names


class ClipValTransform:
    """Resize(R, bicubic, shorter side) -> CenterCrop(R) -> RGB -> ToTensor -> Normalize(mean, std)."""

    def __init__(self, size: int, mean=OPENAI_CLIP_MEAN, std=OPENAI_CLIP_STD):
        self.size = size
        self.mean = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(3, 1, 1)

    def to_uint8(self, img: Image.Image) -> torch.Tensor:
        """Resize + CenterCrop + RGB only: uint8 [3,R,R]; ToTensor + Normalize are left to the device
        (CLIPENC_IN_U8), which computes the identical fp32 expression."""
        return torch.from_numpy(np.asarray(self._resize_crop(img), dtype=np.uint8).copy()).permute(2, 0, 1).contiguous()

    def _resize_crop(self, img: Image.Image) -> Image.Image:
        w, h = img.size
        R = self.size
        if w <= h:
            nw, nh = R, int(R * h / w)
        else:
            nh, nw = R, int(R * w / h)
        if (nw, nh) != (w, h):
            img = img.resize((nw, nh), Image.BICUBIC)
        img = img.crop(_center_crop_box(nw, nh, R, R))
        return img.convert("RGB")

    def __call__(self, img: Image.Image) -> torch.Tensor:
        arr = torch.from_numpy(np.asarray(self._resize_crop(img), dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
        return (arr - self.mean) / self.std

    def __repr__(self):
        return f"ClipValTransform(size={self.size})"


def clip_val_transform(size: int) -> ClipValTransform:
    return ClipValTransform(size)


def crop_box_table(width: int, height: int, crop_names: Sequence[str] = CROP_NAMES):
    """`crop_boxes` in the int[n][5] form the C ABI takes (preproc_crops_u8): kind 0 = crop box, kind 1 = square pad."""
    rows, names = [], []
    for name, kind, box in crop_boxes(width, height, crop_names):
        rows.append([0, *box] if kind == "crop" else [1, box[0], box[1], box[2], 0])
        names.append(name)
    return rows, names


class GpuCropper:
    """Device-side counterpart of `extract_crops` + `ClipValTransform.to_uint8`: decoded uint8 HWC image in HBM ->
    uint8 [n_crops, 3, R, R], bit-exact with the Pillow path (libclipenc_hip.so: preproc_crops_u8)."""

    def __init__(self, size: int, device="cuda", crop_names: Sequence[str] = CROP_NAMES):
        import ctypes
        from . import _lib
        self._lib_mod, self._ct = _lib, ctypes
        self.lib = _lib.load()
        self.size = size
        self.crop_names = list(crop_names)
        d = torch.device(device)
        self.device = torch.device("cuda", d.index if d.index is not None else torch.cuda.current_device())
        h = ctypes.c_void_p()
        _lib.check(self.lib.preproc_create(self.device.index, ctypes.byref(h)), "preproc_create")
        self.handle = h

    @torch.no_grad()
    def __call__(self, img_u8_hwc: torch.Tensor):
        """img: uint8 [H, W, 3] (any device) -> (uint8 [n_crops, 3, R, R] on the GPU, crop names)."""
        if img_u8_hwc.dtype != torch.uint8 or img_u8_hwc.dim() != 3 or img_u8_hwc.shape[2] != 3:
            raise ValueError(f"expected a uint8 [H, W, 3] image, got {tuple(img_u8_hwc.shape)} {img_u8_hwc.dtype}")
        img = img_u8_hwc.to(self.device, non_blocking=True).contiguous()
        H, W = int(img.shape[0]), int(img.shape[1])
        rows, names = crop_box_table(W, H, self.crop_names)
        flat = [v for r in rows for v in r]
        boxes = (self._ct.c_int * len(flat))(*flat)
        out = torch.empty((len(rows), 3, self.size, self.size), dtype=torch.uint8, device=self.device)
        self._lib_mod.check(self.lib.preproc_crops_u8(self.handle, img.data_ptr(), H, W, W * 3, len(rows), boxes, self.size,
                                                      out.data_ptr(), self._lib_mod.current_stream_ptr(self.device)),
                            "preproc_crops_u8")
        return out, names

    @torch.no_grad()
    def batch(self, images: Sequence[torch.Tensor]):
        """Several decoded images at once (three kernel launches for all of them): list of uint8 [H, W, 3] ->
        (uint8 [total crops, 3, R, R] on the GPU in image-major order, list of crop-name lists)."""
        ct = self._ct
        dev_imgs, boxes, counts, names_all = [], [], [], []
        for im in images:
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
                raise ValueError(f"expected a uint8 [H, W, 3] image, got {tuple(im.shape)} {im.dtype}")
            d = im.to(self.device, non_blocking=True).contiguous()
            rows, names = crop_box_table(int(d.shape[1]), int(d.shape[0]), self.crop_names)
            dev_imgs.append(d)
            boxes.extend(v for r in rows for v in r)
            counts.append(len(rows))
            names_all.append(names)
        n, total = len(dev_imgs), sum(counts)
        out = torch.empty((total, 3, self.size, self.size), dtype=torch.uint8, device=self.device)
        if total == 0:
            return out, names_all
        ptrs = (ct.c_void_p * n)(*[d.data_ptr() for d in dev_imgs])
        Hs = (ct.c_int * n)(*[int(d.shape[0]) for d in dev_imgs])
        Ws = (ct.c_int * n)(*[int(d.shape[1]) for d in dev_imgs])
        Ps = (ct.c_int * n)(*[int(d.shape[1]) * 3 for d in dev_imgs])
        Cs = (ct.c_int * n)(*counts)
        Bx = (ct.c_int * len(boxes))(*boxes)
        self._lib_mod.check(self.lib.preproc_crops_u8_batch(self.handle, n, ptrs, Hs, Ws, Ps, Cs, Bx, self.size, out.data_ptr(),
                                                            self._lib_mod.current_stream_ptr(self.device)), "preproc_crops_u8_batch")
        return out, names_all

    def close(self):
        if getattr(self, "handle", None):
            self.lib.preproc_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
