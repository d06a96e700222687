"""Counterpart of /root/reference/predict_simple.py: score every image of a directory with a trained
regressor (image -> crops -> encode -> score in one fused device call) and copy it with the score prepended."""
from __future__ import annotations

import argparse
import os
import shutil

from PIL import Image

from .embedder import AestheticRegressor

IMG_EXTENSIONS = [".jpg", ".png", ".jpeg", ".bmp", ".webp"]          # predict_simple.py:39


def predict_images(img_paths, model_path, device, output_dir=None, clip_model_path=None, batch_size=32):
    aesthetic_regressor = AestheticRegressor(model_path, device=device, clip_model_path=clip_model_path)
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
    print("\nPredicting aesthetic scores...")
    results = []
    for b0 in range(0, len(img_paths), batch_size):
        batch = img_paths[b0:b0 + batch_size]
        scores, _ = aesthetic_regressor.predict_scores([Image.open(p) for p in batch])
        for image_path, score in zip(batch, scores.tolist()):
            print(f"Score: {score:.3f} for {os.path.basename(image_path)}")
            results.append((image_path, score))
            if output_dir is not None:
                shutil.copy(image_path, os.path.join(output_dir, f"{score:.3f}_" + os.path.basename(image_path)))
    return results


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--input_img_dir", type=str, help="Directory with the images to score")
    parser.add_argument("--model_path", type=str, required=True, help="Path to the regressor file (.pth)")
    parser.add_argument("--clip_model_path", type=str, default=None, help="Local directory (or file) holding the CLIP weights")
    args = parser.parse_args(argv)
    paths = [os.path.join(args.input_img_dir, n) for n in sorted(os.listdir(args.input_img_dir))
             if os.path.splitext(n)[1].lower() in IMG_EXTENSIONS]
    print(f"Found {len(paths)} images in {args.input_img_dir}")
    predict_images(paths, args.model_path, "cuda", args.input_img_dir + "_aesthetic_scores", args.clip_model_path)


if __name__ == "__main__":
    main()
