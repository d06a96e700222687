"""Predict driver: the counterpart of /root/reference/_5_predict_labels.py (`CustomDataset`,
`predict_labels`, CLI) with the regressor forward on the HIP kernel.

Keeps: the flags (:193-199), the recursive directory rule (:204-210), `find_model` (:48-56), the feature
assembly order [model][crop in model.crop_names][E] (:69-88), the CSV columns / update rule (new value
overrides, :144-155), the JSON `predicted_label` field (:159-167), the `{score:.3f}_{uuid}.jpg` previews
(:170-177) and the autosave cadence (:179-182).  The per-batch pandas outer merge (quadratic overall)
is replaced by an in-memory table merged once per save — same file contents.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import time
from typing import List

import numpy as np
import pandas as pd
import torch

from .nn_model import load_regressor


def find_model(model_name, model_dir="models"):
    if os.path.exists(model_name) and os.path.isfile(model_name):
        return model_name
    if os.path.isdir(model_dir):
        for model_file in os.listdir(model_dir):
            if model_name in model_file:
                return os.path.join(model_dir, model_file)
    return None


def assemble_features(feature_path: str, clip_models: List[str], crop_names: List[str]) -> torch.Tensor:
    """_5_predict_labels.py:75-82."""
    full = torch.load(feature_path, map_location="cpu", weights_only=True)
    parts = []
    for m in clip_models:
        d = full[m]
        parts.append(torch.cat([d[c] for c in crop_names if c in d], dim=0).flatten())
    return torch.cat(parts, dim=0).flatten()


@torch.no_grad()
def predict_labels(args, device="cuda"):
    model_file = find_model(args.model_file)
    if model_file is None or not os.path.exists(model_file):
        raise FileNotFoundError(f"could not find model file {args.model_file!r}")
    output_dir = args.root_dir + "_predicted_scores"
    os.makedirs(output_dir, exist_ok=True)
    model = load_regressor(model_file)
    model.eval()
    clip_models = model.clip_models
    print("Loaded regression model trained on the following CLIP models:")
    print(clip_models)

    label_file = os.path.join(os.path.dirname(args.root_dir), os.path.basename(args.root_dir) + ".csv")
    if os.path.exists(label_file):
        database = pd.read_csv(label_file)
        print(f"Loaded existing database: {label_file}.\nDatabase contains {len(database)} entries")
    else:
        database = pd.DataFrame(columns=["uuid", "label", "timestamp", "predicted_label"])
    if "predicted_label" not in database.columns:
        database["predicted_label"] = np.nan

    img_files = sorted(os.path.splitext(f)[0] for f in os.listdir(args.root_dir) if f.endswith(".jpg"))
    print(f"Predicting labels for {len(img_files)} images...")
    updates = {}                                            # uuid -> (predicted_label, timestamp)

    def save():
        db = database
        if updates:
            new = pd.DataFrame({"uuid": list(updates), "predicted_label_new": [v[0] for v in updates.values()],
                                "timestamp_new": [v[1] for v in updates.values()]})
            db = database.merge(new, on="uuid", how="outer")
            db["predicted_label"] = db["predicted_label_new"].where(db["predicted_label_new"].notna(), db["predicted_label"])
            db["timestamp"] = db["timestamp_new"].where(db["timestamp_new"].notna(), db["timestamp"])
            db = db.drop(columns=["predicted_label_new", "timestamp_new"])
        db.to_csv(label_file, index=False)
        return db

    n_predictions = 0
    rng = np.random
    store = None
    if getattr(args, "packed_store", None):                 # features from the packed shards: no per-image torch.load
        from .packed_store import PackedStore, image_key
        store = getattr(args, "_store", None)               # main() opens the store ONCE for all sub-directories; its
        if store is None:                                   # per-model key index is built once and cached inside it
            store = args._store = PackedStore(args.packed_store)
        store_root = getattr(args, "store_root", None) or args.root_dir
    for b0 in range(0, len(img_files), args.batch_size):
        uuids, img_paths, feats = [], [], []
        batch = img_files[b0:b0 + args.batch_size]
        if store is not None:
            keys = [image_key(os.path.join(args.root_dir, u), store_root) for u in batch]
            found, mat = store.features(clip_models, model.crop_names, keys)
            for u, ok in zip(batch, found):
                if ok:
                    uuids.append(u)
                    img_paths.append(os.path.join(args.root_dir, u + ".jpg"))
                else:
                    print(f"WARNING: no packed embedding for {u}, skipping this sample..")
            feats = list(torch.from_numpy(np.ascontiguousarray(mat)))
        else:
            for uuid in batch:
                try:
                    feats.append(assemble_features(os.path.join(args.root_dir, uuid + ".pt"), clip_models, model.crop_names))
                    uuids.append(uuid)
                    img_paths.append(os.path.join(args.root_dir, uuid + ".jpg"))
                except Exception as e:                      # :84-86: skip the sample
                    print(f"WARNING: {str(e)} for {uuid}, skipping this sample..")
        if not uuids:
            continue
        features = torch.stack(feats)
        predicted = model(features.to(device).float()).cpu().numpy().reshape(-1)      # :135
        now = int(time.time())
        for uuid, label in zip(uuids, predicted):
            updates[uuid] = (float(label), now)
            json_file = os.path.join(args.root_dir, uuid + ".json")
            if os.path.exists(json_file):
                with open(json_file, "r") as f:
                    data = json.load(f)
                data["predicted_label"] = float(label)
                with open(json_file, "w") as f:
                    json.dump(data, f)
        if args.copy_imgs_fraction > 0:
            pick = np.arange(len(uuids))[rng.random(len(uuids)) < args.copy_imgs_fraction]
            for i in pick:
                shutil.copy(img_paths[i], os.path.join(output_dir, f"{predicted[i]:.3f}_{uuids[i]}.jpg"))
        n_before = n_predictions
        n_predictions += len(uuids)
        if n_predictions // 100 != n_before // 100:
            save()
    final = save()
    print("Done!")
    print(f"{n_predictions} of {len(img_files)} img predicted. (the rest was skipped due to errors)")
    if len(final):
        print(f"Average predicted label: {final['predicted_label'].mean():.3f}")
    print(f"Database saved at {label_file}")
    return final


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--root_dir", type=str, help="Root directory of the dataset")
    parser.add_argument("--model_file", type=str, help="Path to the model file (.pth)")
    parser.add_argument("--batch_size", type=int, default=1024, help="Batch size for predicting")
    parser.add_argument("--copy_imgs_fraction", type=float, default=0.01,
                        help="Fraction of images to copy to the _predicted_scores directory with prepended prediction score")
    parser.add_argument("--num_workers", type=int, default=4, help="(kept for CLI compatibility)")
    parser.add_argument("--packed_store", type=str, default=None,
                        help="Read embeddings from the packed shards in this directory (written by embed_driver --packed_store)")
    args = parser.parse_args(argv)
    top = args.root_dir
    args.store_root = top                                  # packed keys are relative to the top-level dataset root
    for root, _, files in os.walk(top):                    # :204-210
        if any(f.endswith(".jpg") for f in files) and "_predicted_scores" not in root:
            args.root_dir = root
            print(f"\n\nPredicting labels for {root}...")
            predict_labels(args)


if __name__ == "__main__":
    main()
