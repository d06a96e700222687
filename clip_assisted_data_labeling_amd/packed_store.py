"""Packed embedding store.

The reference keeps one pickle per image (`<image>.pt` = `{model_name: {crop_name: float32[1, E]}}`,
/root/reference/_1_embed_with_CLIP.py:136-168) and every consumer opens them one by one
(_5_predict_labels.py:69-88, _2_remove_duplicates.py:25-46).  At the rate the HIP encoder produces embeddings
(thousands of images per second per GPU) that is thousands of `torch.load` + `torch.save` per second.  This store
keeps the same information as a few large append-only shards and converts losslessly in both directions, so the
per-image `.pt` wire format stays available to the reference's own scripts:

    <store_dir>/<model>.r<rank>.<seq>.f32    raw little-endian float32 [n][n_crops][E], rows in append order
    <store_dir>/<model>.r<rank>.<seq>.json   {"format", "model_name", "crop_names", "embed_dim", "n", "keys"}

`keys` are image paths relative to the dataset root without extension (what the reference calls the uuid, plus the
sub-directory).  One process (rank) writes its own shards — nothing is shared between writers; a shard becomes visible
when its index is renamed into place, so a crash loses at most the shard being written.  When a key occurs in several
shards of a model the newest wins (that is how `--force_reencode` overwrites).
"""
from __future__ import annotations

import json
import os
import re
import time
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

FORMAT = 1
_SAFE = re.compile(r"[^A-Za-z0-9_.-]")


def _model_tag(model_name: str) -> str:
    return _SAFE.sub("_", model_name.replace("/", "__"))


def image_key(image_path: str, root_dir: str) -> str:
    """Key of an image: its path relative to the dataset root, extension stripped, '/' separators."""
    rel = os.path.relpath(os.path.splitext(image_path)[0], root_dir)
    return rel.replace(os.sep, "/")


class PackedStoreWriter:
    """Appends `[n, n_crops, E]` float32 blocks; `rotate_every` images per shard bounds what a crash can lose."""

    def __init__(self, store_dir: str, model_name: str, crop_names: Sequence[str], embed_dim: int, rank: int = 0,
                 rotate_every: int = 1 << 16):
        self.store_dir = store_dir
        self.model_name = model_name
        self.crop_names = list(crop_names)
        self.embed_dim = int(embed_dim)
        self.rank = int(rank)
        self.rotate_every = int(rotate_every)
        os.makedirs(store_dir, exist_ok=True)
        self._fh = None
        self._keys: List[str] = []
        self._base = None
        self.n_written = 0

    def _open(self):
        tag = _model_tag(self.model_name)
        seq = 0
        taken = set(os.listdir(self.store_dir))
        while True:                                            # never reuse a name, also not one left by a crashed writer
            base = f"{tag}.r{self.rank:03d}.{seq:05d}"
            if base + ".json" not in taken and base + ".f32" not in taken:
                break
            seq += 1
        self._base = os.path.join(self.store_dir, base)
        self._fh = open(self._base + ".f32", "wb")
        self._keys = []

    def append(self, keys: Sequence[str], emb) -> None:
        arr = emb.detach().cpu().numpy() if isinstance(emb, torch.Tensor) else np.asarray(emb)
        arr = np.ascontiguousarray(arr, dtype="<f4")
        if arr.ndim != 3 or arr.shape != (len(keys), len(self.crop_names), self.embed_dim):
            raise ValueError(f"expected [{len(keys)}, {len(self.crop_names)}, {self.embed_dim}] embeddings, got {arr.shape}")
        start = 0
        while start < len(keys):
            if self._fh is None:
                self._open()
            room = self.rotate_every - len(self._keys)
            stop = min(len(keys), start + room)
            self._fh.write(arr[start:stop].tobytes())
            self._keys.extend(keys[start:stop])
            self.n_written += stop - start
            start = stop
            if len(self._keys) >= self.rotate_every:
                self._seal()

    def _seal(self):
        if self._fh is None:
            return
        self._fh.flush()
        os.fsync(self._fh.fileno())
        self._fh.close()
        self._fh = None
        if not self._keys:
            os.remove(self._base + ".f32")
            return
        index = {"format": FORMAT, "model_name": self.model_name, "crop_names": self.crop_names,
                 "embed_dim": self.embed_dim, "dtype": "float32", "n": len(self._keys), "created": time.time(),
                 "keys": self._keys}
        tmp = self._base + ".json.tmp"
        with open(tmp, "w") as f:
            json.dump(index, f)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, self._base + ".json")                 # the shard exists from here on
        self._keys = []

    def close(self):
        self._seal()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class PackedStore:
    """Read side: all sealed shards under `store_dir`."""

    def __init__(self, store_dir: str):
        self.store_dir = store_dir
        self._shards: Dict[str, List[dict]] = {}
        self._cache: Dict[str, tuple] = {}
        if os.path.isdir(store_dir):
            for name in sorted(os.listdir(store_dir)):
                if not name.endswith(".json"):
                    continue
                with open(os.path.join(store_dir, name)) as f:
                    idx = json.load(f)
                if idx.get("format") != FORMAT:
                    raise ValueError(f"{name}: unknown packed-store format {idx.get('format')!r}")
                data = os.path.join(store_dir, name[:-5] + ".f32")
                want = idx["n"] * len(idx["crop_names"]) * idx["embed_dim"] * 4
                if not os.path.exists(data) or os.path.getsize(data) < want:
                    raise ValueError(f"{name}: data file missing or shorter than its index says ({want} bytes)")
                idx["_data"] = data
                self._shards.setdefault(idx["model_name"], []).append(idx)
            for shards in self._shards.values():               # oldest first, so that later shards override
                shards.sort(key=lambda s: (s["created"], s["_data"]))

    def models(self) -> List[str]:
        return sorted(self._shards)

    def keys(self, model_name: str) -> set:
        out = set()
        for s in self._shards.get(model_name, []):
            out.update(s["keys"])
        return out

    def _index(self, model_name: str):
        """Cached per model: (crop_names, E, [memmap per shard], {key: (shard, row)} newest shard wins, keys in first-seen
        order).  Built once per PackedStore: nothing is concatenated and no shard is read until rows are asked for."""
        cached = self._cache.get(model_name)
        if cached is not None:
            return cached
        shards = self._shards.get(model_name, [])
        if not shards:
            raise KeyError(f"no shards for model {model_name!r} in {self.store_dir}")
        crop_names, E = list(shards[0]["crop_names"]), shards[0]["embed_dim"]
        maps, where, order = [], {}, []
        for si, s in enumerate(shards):                       # oldest first: later shards override
            if s["crop_names"] != crop_names or s["embed_dim"] != E:
                raise ValueError(f"shards of {model_name!r} disagree on crop names / embedding width")
            maps.append(np.memmap(s["_data"], dtype="<f4", mode="r", shape=(s["n"], len(crop_names), E)))
            for row, k in enumerate(s["keys"]):
                if k not in where:
                    order.append(k)
                where[k] = (si, row)
        cached = self._cache[model_name] = (crop_names, E, maps, where, order)
        return cached

    def crop_names(self, model_name: str) -> List[str]:
        return list(self._index(model_name)[0])

    def rows(self, model_name: str, keys: Sequence[str]):
        """(found bool[n], float32 [n_found, n_crops, E]) for `keys`, read shard by shard."""
        crop_names, E, maps, where, _ = self._index(model_name)
        loc = [where.get(k) for k in keys]
        found = np.array([l is not None for l in loc], dtype=bool)
        sel = [l for l in loc if l is not None]
        out = np.empty((len(sel), len(crop_names), E), np.float32)
        if sel:
            sh = np.array([l[0] for l in sel], dtype=np.int64)
            rw = np.array([l[1] for l in sel], dtype=np.int64)
            for si in np.unique(sh):
                m = sh == si
                out[m] = _gather_rows(maps[si], rw[m])
        return found, out

    def load(self, model_name: str) -> Tuple[List[str], np.ndarray, List[str]]:
        """(keys, float32 [N, n_crops, E], crop_names) with every key once (newest shard wins), in first-seen order.
        A single shard without overwritten keys is returned as its memory map (no copy)."""
        crop_names, E, maps, where, order = self._index(model_name)
        if len(maps) == 1 and len(order) == maps[0].shape[0]:
            return list(order), maps[0], list(crop_names)
        _, data = self.rows(model_name, order)
        return list(order), data, list(crop_names)

    def features(self, clip_models: Sequence[str], crop_names: Sequence[str], keys: Sequence[str]):
        """Regressor input rows, assembled as _5_predict_labels.py:75-82 does: [model][crop in crop_names][E] flattened.
        Returns (found bool[n], float32 [n_found, n_models * n_crops * E]) for `keys`.

        A requested crop that the store does not hold raises ValueError naming it: the feature row is never narrowed
        silently (the reference's trainer raises 'Missing crops' and drops the sample, _4_train_model.py:52-54; in a
        packed store the crop set is store-wide, so every sample would be dropped)."""
        found = np.ones(len(keys), dtype=bool)
        cols_of = {}
        for m in clip_models:
            names = self._index(m)[0]
            missing = [c for c in crop_names if c not in names]
            if missing:
                raise ValueError(f"Missing crops {missing} for model {m!r}: the packed store {self.store_dir} holds {names}")
            cols_of[m] = [names.index(c) for c in crop_names]
            where = self._index(m)[3]
            found &= np.array([k in where for k in keys], dtype=bool)
        sel_keys = [k for k, ok in zip(keys, found) if ok]
        out = []
        for m in clip_models:
            _, block = self.rows(m, sel_keys)
            cols = cols_of[m]
            if cols != list(range(block.shape[1])):
                block = block[:, cols, :]
            out.append(block.reshape(len(sel_keys), -1))
        if len(out) == 1:
            return found, out[0]
        return found, (np.concatenate(out, axis=1) if out else np.zeros((len(sel_keys), 0), np.float32))


def _gather_rows(data: np.ndarray, rows: np.ndarray) -> np.ndarray:
    """data[rows] for a (memory-mapped) [N, C, E] array: one sequential read of the covering range when the rows are
    dense in it (the usual case: a directory's images were appended together), a gather otherwise."""
    if len(rows) == 0:
        return np.zeros((0,) + data.shape[1:], np.float32)
    lo, hi = int(rows.min()), int(rows.max()) + 1
    if hi - lo <= 4 * len(rows):
        buf = np.array(data[lo:hi])
        if hi - lo == len(rows) and np.array_equal(rows, np.arange(lo, hi)):
            return buf
        return buf[rows - lo]
    return np.asarray(data[rows])


def export_pt(store_dir: str, root_dir: str, models: Optional[Iterable[str]] = None) -> int:
    """Compatibility exporter: writes / merges the reference's per-image `<key>.pt` files under `root_dir`
    (`{model_name: {crop_name: float32[1, E]}}`, other models already in a file are kept, :136-168)."""
    store = PackedStore(store_dir)
    models = list(models) if models is not None else store.models()
    per_key: Dict[str, dict] = {}
    for m in models:
        keys, data, crop_names = store.load(m)
        for i, k in enumerate(keys):
            per_key.setdefault(k, {})[m] = {c: torch.from_numpy(np.array(data[i, j], dtype=np.float32)).unsqueeze(0)
                                             for j, c in enumerate(crop_names)}
    for k, entry in per_key.items():
        path = os.path.join(root_dir, k.replace("/", os.sep) + ".pt")
        final = {}
        if os.path.exists(path):
            try:
                final = torch.load(path, map_location="cpu", weights_only=True)
            except Exception as e:
                print(f"Warning: Failed to load existing {path} for update: {e}")
        final.update(entry)
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        torch.save(final, path)
    return len(per_key)


def import_pt(root_dir: str, store_dir: str, rank: int = 0) -> Dict[str, int]:
    """Packs the `.pt` files found under `root_dir` (one shard set per model); files whose crops differ from the first
    file of their model are skipped with a message.  Returns {model_name: images packed}."""
    writers: Dict[str, PackedStoreWriter] = {}
    counts: Dict[str, int] = {}
    for sub, _, files in os.walk(root_dir):
        for name in sorted(files):
            if not name.endswith(".pt"):
                continue
            path = os.path.join(sub, name)
            try:
                d = torch.load(path, map_location="cpu", weights_only=True)
            except Exception as e:
                print(f"Warning: could not read {path}: {e}")
                continue
            if not isinstance(d, dict):
                continue
            for m, crops in d.items():
                if not isinstance(crops, dict) or not crops:
                    continue
                names = list(crops)
                E = int(next(iter(crops.values())).numel())
                w = writers.get(m)
                if w is None:
                    w = writers[m] = PackedStoreWriter(store_dir, m, names, E, rank)
                if names != w.crop_names or any(crops[c].numel() != w.embed_dim for c in names):
                    print(f"Skipping {path} for {m}: crops {names} differ from the store's {w.crop_names}")
                    continue
                emb = torch.stack([crops[c].reshape(-1).float() for c in names]).unsqueeze(0)
                w.append([image_key(path, root_dir)], emb)
                counts[m] = counts.get(m, 0) + 1
    for w in writers.values():
        w.close()
    return counts


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Convert between the packed embedding store and per-image .pt files")
    ap.add_argument("direction", choices=["export", "import"], help="export: store -> .pt files; import: .pt files -> store")
    ap.add_argument("--root_dir", required=True)
    ap.add_argument("--packed_store", required=True)
    ap.add_argument("--models", nargs="*", default=None)
    a = ap.parse_args(argv)
    if a.direction == "export":
        print(f"wrote {export_pt(a.packed_store, a.root_dir, a.models)} .pt files under {a.root_dir}")
    else:
        print(f"packed: {import_pt(a.root_dir, a.packed_store)}")


if __name__ == "__main__":
    main()
