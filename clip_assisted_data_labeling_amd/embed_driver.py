"""Embed driver: the counterpart of /root/reference/_1_embed_with_CLIP.py (`Feature_Dataset`, CLI).

Same flags (:187-197), same file discovery (:47, :54-58), same `<image_basename>.pt` store
(`{model_name: {crop_name: float32[1,E]}}`, several models merged per file,
:136-168) and the same skip-if-already-embedded rule (:117-128), written the way the downstream readers
(_5_predict_labels.py:79, _2_remove_duplicates.py:38) expect it: every image gets ALL of its crop keys,
each holding that image's own crop embedding (the reference's writer mis-keys them for batch sizes
above one; that defect is not reproduced).  The 22 `img_stat_*` scalars of
utils/image_features.py are not produced (never consumed: _4_train_model.py:274).

Differences that matter for speed: the resume check happens BEFORE images are decoded, per image
rather than per batch; with torch.distributed initialised, each rank embeds a contiguous shard of the
sorted file list (clip_assisted_data_labeling_amd/sharding.py) on its own GPU.  With `--packed_store DIR` the
embeddings go to a few large append-only shards instead of one pickle per image (packed_store.py; `python -m
clip_assisted_data_labeling_amd.packed_store export` recreates the per-image `.pt` files).
"""
from __future__ import annotations

import argparse
import os
import random
from typing import List, Optional, Sequence

import torch
from PIL import Image
from torch.utils.data import DataLoader, Dataset

from .preprocess import CROP_NAMES, extract_crops
from .sharding import shard_list

IMG_EXTENSIONS = (".png", ".jpg", ".jpeg", ".JPEG", ".JPG", ".PNG")      # _1_embed_with_CLIP.py:47


class CustomImageDataset(Dataset):
    """utils/embedder.py:153-181: path -> (crops[n_crops,3,R,R], crop_names, path)."""

    def __init__(self, image_paths: Sequence[str], crop_names: Sequence[str], preprocess_transform):
        self.image_paths = list(image_paths)
        self.crop_names = list(crop_names)
        self.preprocess_transform = preprocess_transform

    def __len__(self):
        return len(self.image_paths)

    def __getitem__(self, idx):
        img_path = self.image_paths[idx]
        try:
            pil_img = Image.open(img_path).convert("RGB")
            raw_crops, names = extract_crops(pil_img, self.crop_names)
            crops = torch.stack([self.preprocess_transform(c) for c in raw_crops])
            return crops, ",".join(names), img_path, True
        except Exception as e:  # unreadable image: report it, never substitute another one (Appendix D #4)
            print(f"Error loading or processing image {img_path}: {e}")
            return torch.zeros(0), "", img_path, False


class RawImageDataset(Dataset):
    """Workers only decode: path -> uint8 [H, W, 3]; crops, resize and normalise run on the GPU (GpuCropper)."""

    def __init__(self, image_paths: Sequence[str]):
        self.image_paths = list(image_paths)

    def __len__(self):
        return len(self.image_paths)

    def __getitem__(self, idx):
        img_path = self.image_paths[idx]
        try:
            import numpy as np
            arr = np.asarray(Image.open(img_path).convert("RGB"), dtype=np.uint8)
            return torch.from_numpy(arr.copy()), "", img_path, True
        except Exception as e:
            print(f"Error loading or processing image {img_path}: {e}")
            return torch.zeros(0), "", img_path, False


class _Shape:
    """Stands in for a crop tensor once it has been sent to the GPU: only `.shape[0]` (its crop count) is still needed."""

    def __init__(self, n):
        self.shape = (n,)


def _collate(batch):
    ok = [b for b in batch if b[3]]
    bad = [b[2] for b in batch if not b[3]]
    return ok, bad


def chunk_cut(pixels: Sequence[int], max_files: int, max_pixels: int) -> int:
    """How many of the next files form one decode chunk: at most `max_files`, and as many as fit `max_pixels` decoded pixels --
    but always at least one (a single image over the budget is decoded alone).  `pixels[i]` = width * height of file i as its
    header states it (0 for files that do not go to the device)."""
    take, total = 0, 0
    for px in pixels[:max(1, max_files)]:
        if take > 0 and total + px > max_pixels:
            break
        total += px
        take += 1
    return take


def find_images(root_dir: str) -> List[str]:
    paths = []
    for root, _, files in os.walk(root_dir):
        for name in files:
            if name.endswith(IMG_EXTENSIONS):
                paths.append(os.path.join(root, name))
    return paths


def already_embedded(feature_path: str, model_name: str) -> bool:
    if not os.path.exists(feature_path):
        return False
    try:
        return model_name in torch.load(feature_path, map_location="cpu", weights_only=True)
    except Exception as e:
        print(f"Warning: Could not load existing feature file {feature_path}: {e}")
        return False


class Feature_Dataset:
    def __init__(self, root_dir, model_name, batch_size, model_path=None, force_reencode=False,
                 shuffle_filenames=True, num_workers=0, crop_names=None, encoder=None, device="cuda",
                 gpu_preprocess=False, packed_store=None, shard_images=8192, gpu_decode=False, decode_chunk=2048,
                 precision="bf16"):
        self.device = device
        # gpu_decode: no decoding workers at all -- the main process reads file bytes (a small thread pool), the GPU decodes
        # the JPEGs (jpeg_gpu.GpuJpegDecoder, bit-identical to Pillow) `decode_chunk` files at a time (the entropy decoder is
        # one serial stream per file: its parallelism is the number of files in flight) and the GPU front end crops / resizes.
        # Files the decoder does not take (progressive, CMYK, PNG, ...) are decoded by Pillow here, as the reference does.
        self.gpu_decode = bool(gpu_decode)
        self.decode_chunk = max(int(decode_chunk), 1)
        self.gpu_decode_max_bytes = 64 << 20                # larger files: Pillow in the reader threads (see gpu_decoded_batches)
        # Device memory of one decode chunk: its decoded images (3 bytes per pixel) live from the decode until the chunk's crops
        # are cut, next to ~4.5 bytes per pixel of decoder scratch.  A chunk is therefore cut at `decode_chunk` files OR at this
        # many pixels, whichever comes first (1 Gpx = 7.5 GB; 2 048 twelve-megapixel photos would be 74 GB of RGB alone).
        self.gpu_decode_max_pixels = 1_000_000_000
        self.gpu_decode_read_ahead_bytes = 2 << 30          # file bytes the reader pool may hold ahead of the decoder
        self.pt_writers = 1                                 # threads writing the per-image .pt files of a batch (pickling holds the GIL:
                                                            # 2 and 4 measured slower than 1 next to the stager thread, tools/ab_embed_e2e.py)
        # where the stager's decode + crop kernels go: "priority" = a high-priority stream (they start at the next boundary
        # between two of the encoder's kernels), "side" = an ordinary second stream, "same" = the encoder's own stream
        self.gpu_decode_stream = "priority"
        self.gpu_decode_queue = 6                           # encode batches the stager may be ahead of the encoder
        self.first_chunk_div = 4                            # the first decode chunk is batch_size / this many files
        # Progressive files: the device takes them, but walks each with ONE lane (scans are serial), 0.25 - 0.5 s for a chunk
        # during which its small workgroups sit on every CU and the encoder's persistent kernels cannot be placed -- with host
        # cores to spare Pillow in the reader threads is the better deal; set True on a box without them.
        self.gpu_decode_progressive = False
        gpu_preprocess = gpu_preprocess or self.gpu_decode
        self.packed_store = packed_store
        self.shard_images = int(shard_images)     # images per sealed shard = the most a killed rank can lose
        self.root_dir = root_dir
        self.model_name = model_name
        self.force_reencode = force_reencode
        self.batch_size = batch_size
        self.crop_names = list(crop_names) if crop_names is not None else list(CROP_NAMES)
        print("Searching images..")
        self.img_filepaths = find_images(root_dir)
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        if shuffle_filenames and not dist_on:
            random.shuffle(self.img_filepaths)             # :60-61 (unseeded, single process only)
        else:
            self.img_filepaths.sort()
        print(f"---> Found {len(self.img_filepaths)} images in {root_dir}")
        if dist_on:                                        # one process per GPU: contiguous shard of the sorted list
            self.img_filepaths = shard_list(self.img_filepaths, torch.distributed.get_rank(),
                                            torch.distributed.get_world_size())
        if encoder is not None:
            self.encoder = encoder
        elif "/" in model_name:
            from .embedder import CLIP_Encoder
            self.encoder = CLIP_Encoder(model_name, model_path, device=self.device, precision=precision)   # "bf16" | "fp8" (e4m3 block GEMMs)
        else:
            raise ValueError(f"Unknown model format: {model_name}. Expected 'Arch/Dataset'.")     # :75
        self.preprocess = self.encoder.get_preprocess_transform()
        # the HIP encoder takes raw uint8 crops (ToTensor + Normalize fused into its first kernel): 1 byte per pixel
        # crosses the DataLoader pipes and PCIe instead of 4
        if getattr(self.encoder, "accepts_uint8", False) and hasattr(self.preprocess, "to_uint8"):
            self.preprocess = self.preprocess.to_uint8
        self.num_workers = num_workers
        self.cropper = None
        if gpu_preprocess:                                 # crop geometry + bicubic resize on the GPU (bit-exact with Pillow)
            from .preprocess import GpuCropper
            self.cropper = GpuCropper(self.encoder.img_resolution, self.device, self.crop_names)
        self.jpeg = None
        if self.gpu_decode:
            from .jpeg_gpu import GpuJpegDecoder
            self.jpeg = GpuJpegDecoder(self.device)

    def __len__(self):
        return len(self.img_filepaths)

    @torch.no_grad()
    def process(self):
        n_embedded, n_skipped, n_failed = 0, 0, 0
        print(f"Embedding dataset of {len(self.img_filepaths)} images using {self.model_name}...")
        todo = []
        writer, done_keys = None, set()
        if self.packed_store:
            from .packed_store import PackedStore, PackedStoreWriter, image_key
            if not self.force_reencode:
                done_keys = PackedStore(self.packed_store).keys(self.model_name)
            rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        for p in self.img_filepaths:                       # resume check first: nothing is decoded for done images
            if self.packed_store:
                done = image_key(p, self.root_dir) in done_keys
            else:
                done = not self.force_reencode and already_embedded(os.path.splitext(p)[0] + ".pt", self.model_name)
            if done:
                n_skipped += 1
            else:
                todo.append(p)
        # A DataLoader worker produces one whole loader batch by itself, so the loader batch is a fraction of the encode
        # batch: all workers then decode images of the SAME encode batch in parallel and the first one is complete after
        # batch_size / num_workers decodes instead of batch_size.
        loader_bs = self.batch_size if self.num_workers <= 1 else max(4, -(-self.batch_size // self.num_workers))
        kwargs = dict(batch_size=loader_bs, shuffle=False, num_workers=self.num_workers, collate_fn=_collate)
        if self.num_workers > 0:
            kwargs["prefetch_factor"] = 4
        dataset = RawImageDataset(todo) if self.cropper else CustomImageDataset(todo, self.crop_names, self.preprocess)
        on_gpu = torch.device(self.device).type == "cuda" and torch.cuda.is_available()
        # decoded images reach the main process in page-locked memory (the loader's pin thread copies them there in the
        # background), so the per-image upload is an asynchronous DMA instead of a 0.6 ms synchronous pageable copy
        loader = DataLoader(dataset, pin_memory=on_gpu, **kwargs) if self.jpeg is None else []   # (gpu_decode: no workers)

        import time as _time
        self.progress_log = []                             # (perf_counter, images stored so far) after every batch: start-up vs steady state
        self.marks = {"start": _time.perf_counter()}       # first_batch_ready / first_encode_issued / last_encode_issued / done
        self.stager_log = []                               # gpu_decode: per decode chunk, where the stager's time went

        def finish(batch, features):
            """Host side of one batch: embeddings [sum crops, E] (CPU fp32) -> the store."""
            nonlocal writer, n_embedded
            try:
                _finish(batch, features)
            finally:
                self.progress_log.append((_time.perf_counter(), n_embedded))

        def _finish(batch, features):
            nonlocal writer, n_embedded
            counts = [b[0].shape[0] for b in batch]
            if self.packed_store:                          # one sequential write per batch instead of one pickle per image
                if any(n != len(self.crop_names) for n in counts):
                    raise RuntimeError("packed store needs every image to carry all crops " + str(self.crop_names))
                if writer is None:
                    writer = PackedStoreWriter(self.packed_store, self.model_name, self.crop_names, features.shape[-1], rank,
                                               rotate_every=self.shard_images)
                writer.append([image_key(b[2], self.root_dir) for b in batch],
                              features.view(len(batch), len(self.crop_names), features.shape[-1]))
                n_embedded += len(batch)
                return
            # one pickle per image (the reference's store): a few threads -- torch.save spends part of its time in file I/O
            # outside the GIL, and the LAST batch of a run is written with nothing left to hide behind
            rows, jobs = 0, []
            for (crops, names, img_path, _), n in zip(batch, counts):
                jobs.append((img_path, names, rows, n))
                rows += n

            def write_one(job):
                img_path, names, row, n = job
                feature_save_path = os.path.splitext(img_path)[0] + ".pt"
                final = {}
                if os.path.exists(feature_save_path) and not self.force_reencode:
                    try:
                        final = torch.load(feature_save_path, map_location="cpu", weights_only=True)
                    except Exception as e:
                        print(f"Warning: Failed to load existing {feature_save_path} for update: {e}")
                per_model = {}
                for j, crop_name in enumerate(names.split(",")):
                    per_model[crop_name] = features[row + j].unsqueeze(0).clone()       # float32 [1, E] on CPU (:157-161)
                final[self.model_name] = per_model                                     # :164
                try:
                    torch.save(final, feature_save_path)
                except Exception as e:
                    print(f"Error saving features to {feature_save_path}: {e}")
                return 1
            n_embedded += sum(pt_pool.map(write_one, jobs))

        # Three stages, each on its own thread, handing batches on through bounded queues:
        #   stager  (gpu_decode only): file bytes read ahead by a reader pool -> JPEG decode of one chunk on a SIDE stream -> the
        #           chunk's crops cut on that stream batch by batch (the decoded images are dropped as soon as their crops exist)
        #   encoder (this thread): waits for a batch's crops (stream event), encodes, starts the copy of the embeddings to
        #           page-locked memory and moves on -- kernel launches are asynchronous, so the GPU never waits for the host
        #   writer: waits for a batch's copy event and writes the store (one torch.save per image, or one append per batch)
        import queue
        import threading
        from concurrent.futures import ThreadPoolExecutor as _Pool
        pt_pool = _Pool(max(1, int(self.pt_writers)))
        out_q: "queue.Queue" = queue.Queue(maxsize=4)
        writer_errors: List[BaseException] = []

        def writer_loop():
            while True:
                item = out_q.get()
                if item is None:
                    return
                if writer_errors:                           # keep draining so that the encoder never blocks on a dead writer
                    continue
                try:
                    meta, host, ev = item
                    if ev is not None:
                        ev.synchronize()
                    finish(meta, host)
                except BaseException as e:                  # noqa: BLE001 -- re-raised in the main thread
                    writer_errors.append(e)

        def gpu_decoded_batches():
            """-> (batch meta [(_Shape(crops), names, path, True)], uint8 crops [sum crops, 3, R, R] on the GPU, stream event)"""
            nonlocal n_failed
            import collections
            import io
            import numpy as np
            from concurrent.futures import ThreadPoolExecutor

            def read(path):
                """-> (payload, pixels): the file's bytes (pixels = width * height from its header) -- or, for a file the device
                decoder does not take (a header check, microseconds) or one over `gpu_decode_max_bytes` (64 MiB: a guard, not a
                tuning knob), the image decoded right here with Pillow (pixels = 0): the pool's threads run in parallel, Pillow
                and the check release the GIL.  (None, 0): unreadable."""
                try:
                    with open(path, "rb") as f:
                        blob = f.read()
                    ok, w, h, scans = self.jpeg.probe(blob) if len(blob) <= self.gpu_decode_max_bytes else (False, 0, 0, 0)
                    if not ok or (scans != 1 and not self.gpu_decode_progressive):
                        # CMYK / PNG / ... and, unless asked for, progressive JPEG: Pillow, as the reference -- here, in the
                        # pool, so that such files are decoded in parallel and not one after the other
                        return torch.from_numpy(np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"), dtype=np.uint8).copy()), 0
                    return blob, w * h
                except Exception as e:
                    print(f"Error loading or processing image {path}: {e}")
                    return None, 0

            # The stager's decode and crop kernels go to a HIGH-PRIORITY stream: they are placed at the next boundary between two of the
            # encoder's kernels and then have the GPU to themselves for their few milliseconds.  On an ordinary second stream they
            # only found CUs in the encoder's tail rounds -- its persistent kernels hold every CU -- and a 10-ms entropy workgroup
            # that got one kept the next persistent kernel waiting for that CU; in the encoder's own stream the decode's final
            # synchronisation waits for the whole batch queued before it, so the stager is never more than one batch ahead.
            mode = self.gpu_decode_stream
            side = (torch.cuda.current_stream(self.device) if mode == "same" else
                    torch.cuda.Stream(device=self.device, priority=-1 if mode == "priority" else 0))
            in_q: "queue.Queue" = queue.Queue(maxsize=int(self.gpu_decode_queue))
            stop = threading.Event()

            def put(item):
                while not stop.is_set():
                    try:
                        in_q.put(item, timeout=0.2)
                        return True
                    except queue.Full:
                        continue
                return False

            def stager():
                try:
                    with ThreadPoolExecutor(max(2, min(16, self.num_workers or 8))) as readers:
                        ahead = collections.deque()             # (path, future, file bytes) in file order
                        # the first chunk is a QUARTER of an encode batch (the encoder starts after a few milliseconds of decoding)
                        nxt, ahead_bytes, limit, reserved = 0, 0, max(min(32, self.batch_size), self.batch_size // max(1, int(self.first_chunk_div))), False
                        while (nxt < len(todo) or ahead) and not stop.is_set():
                            # chunks double up to `decode_chunk` files: the entropy decoder's parallelism is the files in flight
                            # (read ahead by two chunks of the CURRENT size: submitting 1 024 reads before the first 128-file chunk was looked
                            #  at cost 50 ms of start-up)
                            while nxt < len(todo) and len(ahead) < limit * 2 and ahead_bytes < self.gpu_decode_read_ahead_bytes:
                                try:
                                    size = os.path.getsize(todo[nxt])
                                except OSError:
                                    size = 0
                                ahead.append((todo[nxt], readers.submit(read, todo[nxt]), size))
                                ahead_bytes += size
                                nxt += 1
                            tl = {"t": _time.perf_counter() - self.marks["start"]}
                            window = []
                            for path, fut, size in list(ahead)[:limit]:
                                window.append((path, fut.result(), size))
                            tl["reads_s"] = _time.perf_counter() - self.marks["start"] - tl["t"]
                            take = chunk_cut([w[1][1] for w in window], limit, self.gpu_decode_max_pixels)
                            chunk = window[:take]
                            if not reserved:
                                # scratch for a FULL chunk of files like these, set aside now -- the encoder has not started, so
                                # the allocation's device synchronisation costs nothing (later growth would stall it)
                                reserved = True
                                dev_files = [w for w in window if isinstance(w[1][0], bytes)]
                                if dev_files:
                                    scale = self.decode_chunk / len(dev_files)
                                    px = min(self.gpu_decode_max_pixels, int(sum(w[1][1] for w in dev_files) * scale * 1.1))
                                    # bytes per pixel from the files' own sampling factors (the largest of the window: a 4:2:0 set reserves
                                    # 4.5, one 4:4:4 file makes it 9 -- a 9 GB arena where 4.6 GB do costs ~0.3 s of start-up in hipMalloc)
                                    from .jpeg_gpu import scratch_bytes_per_pixel
                                    # (over the WHOLE first window -- the header walk is microseconds per file; a directory that starts with
                                    #  4:2:0 files and holds 4:4:4 ones later in the window would otherwise under-reserve by 2x and regrow mid-run)
                                    bpp = max(scratch_bytes_per_pixel(w[1][0]) for w in dev_files)
                                    if not self.jpeg.reserve(px, int(sum(len(w[1][0]) for w in dev_files) * scale * 1.1), bytes_per_pixel=bpp):
                                        print(f"Warning: no device memory for the JPEG decoder's scratch ({px / 1e6:.0f} Mpx per chunk); "
                                              "groups it cannot hold are decoded by Pillow on the host (slow)")
                            for _ in range(take):
                                ahead_bytes -= ahead.popleft()[2]
                            limit = min(self.decode_chunk, limit * 2)
                            tl["files"] = take
                            t_dec = _time.perf_counter()
                            with torch.cuda.stream(side):
                                images, status = self.jpeg.decode([c[1][0] if isinstance(c[1][0], bytes) else b"" for c in chunk],
                                                                  max_batch_pixels=max(self.gpu_decode_max_pixels, 1))
                                tl["decode_s"] = _time.perf_counter() - t_dec
                                acc, failed = [], 0
                                for (path, (payload, _px), _sz), img, st in zip(chunk, images, status):
                                    if isinstance(payload, torch.Tensor):   # decoded by Pillow in a reader thread
                                        img = payload
                                    elif img is None and payload is not None:   # the device flagged its entropy data: Pillow decides, as the reference
                                        if st == self.jpeg.NO_MEMORY and not getattr(self, "_warned_no_memory", False):
                                            self._warned_no_memory = True
                                            print("Warning: the JPEG decoder had no device scratch for a group of files; "
                                                  "they are decoded by Pillow on the host (slow) -- free device memory or lower --batch_size")
                                        try:
                                            img = torch.from_numpy(np.asarray(Image.open(io.BytesIO(payload)).convert("RGB"), dtype=np.uint8).copy())
                                        except Exception as e:
                                            print(f"Error loading or processing image {path}: {e}")
                                    if img is None:
                                        failed += 1
                                        continue
                                    acc.append((img, path))
                                del images
                                # encode batches of this chunk; the job's LAST batch goes out as two halves, so that the store
                                # writes of the first half (one pickle per image: ~0.3 ms each) run under the encode of the second
                                bounds = list(range(0, len(acc), self.batch_size)) + [len(acc)] if acc else [0]
                                if len(bounds) > 1 and nxt >= len(todo) and not ahead and len(acc) - bounds[-2] > 64:
                                    bounds.insert(-1, bounds[-2] + (len(acc) - bounds[-2] + 1) // 2)
                                for b0, b1 in zip(bounds[:-1], bounds[1:]):
                                    part = acc[b0:b1]
                                    stacked, names_all = self.cropper.batch([im for im, _ in part])
                                    ev = torch.cuda.Event()
                                    ev.record(side)
                                    meta = [(_Shape(len(n)), ",".join(n), path, True) for n, (_, path) in zip(names_all, part)]
                                    if not put(("batch", meta, stacked, ev, failed)):
                                        return
                                    failed = 0
                                if failed:
                                    put(("batch", [], None, None, failed))
                                tl["crop_and_queue_s"] = _time.perf_counter() - t_dec - tl["decode_s"]
                                self.stager_log.append({k: round(v, 4) if isinstance(v, float) else v for k, v in tl.items()})
                                del acc                          # the chunk's decoded images: their crops are cut (stream-ordered free)
                    put(("end",))
                except BaseException as e:                      # noqa: BLE001 -- re-raised in the main thread
                    put(("error", e))

            th = threading.Thread(target=stager, name="embed-stager", daemon=True)
            th.start()
            try:
                while True:
                    item = in_q.get()
                    if item[0] == "end":
                        break
                    if item[0] == "error":
                        raise item[1]
                    _, meta, stacked, ev, failed = item
                    n_failed += failed
                    if meta:
                        yield meta, stacked, ev
            finally:
                stop.set()
                th.join()

        def encode_batches():
            """-> (batch meta, crops on the GPU, event or None): loader batches regrouped into encode batches of `batch_size` images"""
            nonlocal n_failed
            if self.jpeg is not None:
                yield from gpu_decoded_batches()
                return

            def ready(ok):
                if self.cropper:
                    # all images of the batch go through the GPU front end in three launches (crop geometry on the host,
                    # uploads from page-locked memory are asynchronous)
                    stacked, names_all = self.cropper.batch([b[0] for b in ok])
                    return [(_Shape(len(n)), ",".join(n), b[2], True) for n, b in zip(names_all, ok)], stacked, None
                stacked = torch.cat([b[0] for b in ok], 0).to(self.device, non_blocking=True)   # [sum crops, 3, R, R], row = image-major
                return [(_Shape(b[0].shape[0]), b[1], b[2], b[3]) for b in ok], stacked, None    # the crop tensors are not kept: only their counts
            acc = []
            for ok_part, bad in loader:
                n_failed += len(bad)
                acc.extend(ok_part)
                while len(acc) >= self.batch_size:
                    yield ready(acc[:self.batch_size])
                    acc = acc[self.batch_size:]
            if acc:
                yield ready(acc)

        def run():
            wt = threading.Thread(target=writer_loop, name="embed-writer", daemon=True)
            wt.start()
            try:
                for meta, stacked, ev in encode_batches():
                    self.marks.setdefault("first_batch_ready", _time.perf_counter())
                    if writer_errors:
                        break
                    if ev is not None:                          # crops cut on the side stream: order this stream behind them
                        cur = torch.cuda.current_stream(stacked.device)
                        cur.wait_event(ev)
                        stacked.record_stream(cur)
                    features = self.encoder.encode_image(stacked).float()                        # :130
                    if on_gpu:
                        host = torch.empty(features.shape, dtype=torch.float32, pin_memory=True)
                        host.copy_(features, non_blocking=True)
                        done = torch.cuda.Event()
                        done.record()
                        out_q.put((meta, host, done))
                    else:
                        out_q.put((meta, features.cpu(), None))
                    self.marks.setdefault("first_encode_issued", _time.perf_counter())
                    self.marks["last_encode_issued"] = _time.perf_counter()
            finally:
                out_q.put(None)
                wt.join()
                self.marks["done"] = _time.perf_counter()
            if writer_errors:
                raise writer_errors[0]

        try:
            run()
        finally:
            # seal the shard on EVERY exit path (an exception in the loop, Ctrl-C, a failed write): an unsealed shard is
            # invisible to readers and to the resume check, i.e. every image already embedded into it would be lost
            if writer is not None:
                writer.close()
            pt_pool.shutdown(wait=True)
        print("\n--- Feature encoding done! ---\n")
        print(f"Embedded {n_embedded} images ({n_skipped} images were already embedded, {n_failed} unreadable). "
              f"Features saved with model key '{self.model_name}'.")
        return n_embedded, n_skipped, n_failed


def build_parser() -> argparse.ArgumentParser:
    """The flags of /root/reference/_1_embed_with_CLIP.py:187-199 (same names, same defaults for the model) + this driver's own."""
    parser = argparse.ArgumentParser()
    parser.add_argument("--root_dir", type=str, required=True, help="Root directory of the dataset (can contain subdirectories)")
    # default = the reference's (/root/reference/_1_embed_with_CLIP.py:190): the shipped regressor checkpoint names this tower
    # in its clip_models, so `.pt` files written without the flag carry the key that checkpoint looks for
    parser.add_argument("--models_to_use", type=str, nargs="+", default=["ViT-L-14-336/openai"],
                        help="Which CLIP models to use, '<arch>/<pretrained>'")
    parser.add_argument("--batch_size", type=int, default=128, help="Number of images to encode at once")
    parser.add_argument("--num_workers", type=int, default=4, help="Number of workers for the dataloader")
    parser.add_argument("--force_reencode", action="store_true", help="Force re-encoding of all images for the specified models")
    parser.add_argument("--model_path", type=str, default=None, help="Local directory (or file) holding the model weights")
    parser.add_argument("--gpu_preprocess", action="store_true",
                        help="Workers only decode; crops, bicubic resize and normalise run on the GPU (bit-exact with Pillow)")
    parser.add_argument("--gpu_decode", action="store_true",
                        help="JPEG files are decoded on the GPU too (bit-identical to Pillow; implies --gpu_preprocess); other "
                             "formats and JPEG variants the decoder does not take go through Pillow in the reader threads")
    parser.add_argument("--packed_store", type=str, default=None,
                        help="Write embeddings to packed shards in this directory instead of one .pt per image")
    parser.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp8"],
                        help="Arithmetic of the encoder's GEMMs: bf16 (embeddings within 7e-5 of the fp32 path, 1 - cos) or e4m3 block "
                             "GEMMs (within 6e-4; 1.45x the images per second)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        # device_id binds the communicator to this rank's GPU at once (no lazy guess from the first collective's tensors)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cuda"
    print(f"Embedding all imgs with {len(args.models_to_use)} models: \n--> {args.models_to_use}")
    for model_name in args.models_to_use:
        print(f"\n--- Processing model: {model_name} ---")
        Feature_Dataset(args.root_dir, model_name, args.batch_size, model_path=args.model_path,
                        force_reencode=args.force_reencode, num_workers=args.num_workers, crop_names=CROP_NAMES,
                        device=device, gpu_preprocess=args.gpu_preprocess, packed_store=args.packed_store,
                        gpu_decode=args.gpu_decode, precision=args.precision).process()


if __name__ == "__main__":
    main()
