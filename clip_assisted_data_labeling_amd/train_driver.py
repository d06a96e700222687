"""Training driver: the counterpart of /root/reference/_4_train_model.py with the optimisation loop on the HIP kernels
(libclipenc_hip.so: fctrain_*), so that embed -> train -> predict never leaves the GPU.

Keeps: the flags (:240-262), label loading from `<train_data_dir>/<name>.csv` with NaN labels dropped (:30-36), the
feature assembly [model][crop in crop_names][E] from `<uuid>.pt` (or from `--packed_store`) with samples that fail to
load skipped (:41-72), label normalisation to [0, 1] (:87-91), the train/test split sizes (:108-109), SimpleFC at
`hidden_sizes` with LeakyReLU / Dropout / Sigmoid (:118-122), Adam + weight decay, CosineAnnealingWarmRestarts stepped
per epoch, MSE (:125-129, :199-212), the per-epoch test loss as the mean of per-batch MSEs with the constant-mean
"dummy" loss next to it (:132-166), the progress lines (:217-219) and the saved artifact: `torch.save(model)` of a
SimpleFC in eval mode whose pickle names the reference's own class (`utils.nn_model.SimpleFC`), under
`models/<name>_<timestamp>_<k>k_imgs_<epochs>_epochs_<mse>_mse.pth` (:227-237).
Differences: shuffling, the split and dropout use numpy / a counter-based hash seeded by --random_seed instead of
torch's global generator (the reference's runs are not reproducible across torch versions either); no matplotlib plots.
"""
from __future__ import annotations

import argparse
import ctypes
import math
import os
import time
from typing import List, Sequence

import numpy as np
import pandas as pd
import torch

from . import _lib
from .nn_model import SimpleFC

CROP_NAMES_DEFAULT = ["centre_crop", "subcrop2_0.1"]          # _4_train_model.py:266 (the uncommented choice)


class FcTrainer:
    """Owner of one `fctrain_t` handle: parameters + Adam state on one GPU."""

    def __init__(self, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], negative_slope: float = 0.01, device="cuda"):
        from .embedder import _device_index
        self.lib = _lib.load()
        self.device_index = _device_index(device)
        self.device = torch.device("cuda", self.device_index)
        Ws = [w.detach().to(torch.float32).contiguous().cpu() for w in weights]
        bs = [b.detach().to(torch.float32).contiguous().cpu() for b in biases]
        n = len(Ws)
        self.sizes = [int(Ws[0].shape[1])] + [int(w.shape[0]) for w in Ws]
        Wp = (_lib.c_float_p * n)(*[ctypes.cast(w.data_ptr(), _lib.c_float_p) for w in Ws])
        bp = (_lib.c_float_p * n)(*[ctypes.cast(b.data_ptr(), _lib.c_float_p) for b in bs])
        h = ctypes.c_void_p()
        _lib.check(self.lib.fctrain_create(n, (ctypes.c_int * (n + 1))(*self.sizes), Wp, bp, ctypes.c_float(negative_slope),
                                           self.device_index, ctypes.byref(h)), "fctrain_create")
        self.handle = h

    @torch.no_grad()
    def epoch(self, X: torch.Tensor, T: torch.Tensor, order, batch_size: int, lr: float, weight_decay: float,
              dropout_prob: float = 0.0, seed: int = 0) -> torch.Tensor:
        """One pass over `order` (int64 row indices, or None); returns the per-batch MSEs (GPU tensor)."""
        assert X.device == self.device and T.device == self.device and X.dtype == T.dtype == torch.float32 and X.is_contiguous()
        if order is not None:
            order = torch.as_tensor(order, dtype=torch.int64).to(self.device).contiguous()
        n = int(order.numel()) if order is not None else int(X.shape[0])
        losses = torch.empty((n + batch_size - 1) // batch_size, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.fctrain_epoch(self.handle, X.data_ptr(), T.data_ptr(), order.data_ptr() if order is not None else None, n,
                                          int(batch_size), ctypes.c_float(lr), ctypes.c_float(weight_decay), ctypes.c_float(dropout_prob),
                                          int(seed) & 0xFFFFFFFF, losses.data_ptr(), _lib.current_stream_ptr(self.device)), "fctrain_epoch")
        return losses

    @torch.no_grad()
    def predict(self, X: torch.Tensor) -> torch.Tensor:
        X = X.to(self.device, torch.float32).contiguous()
        y = torch.empty(X.shape[0], dtype=torch.float32, device=self.device)
        _lib.check(self.lib.fctrain_predict(self.handle, X.data_ptr(), int(X.shape[0]), y.data_ptr(), _lib.current_stream_ptr(self.device)),
                   "fctrain_predict")
        return y

    def parameters(self):
        Ws, bs = [], []
        for l in range(len(self.sizes) - 1):
            W = torch.empty(self.sizes[l + 1], self.sizes[l]); b = torch.empty(self.sizes[l + 1])
            _lib.check(self.lib.fctrain_get_params(self.handle, l, W.data_ptr(), b.data_ptr()), "fctrain_get_params")
            Ws.append(W); bs.append(b)
        return Ws, bs

    def close(self):
        if getattr(self, "handle", None):
            self.lib.fctrain_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def cosine_warm_restarts_lr(base_lr: float, eta_min: float, T_0: int, epoch: int) -> float:
    """CosineAnnealingWarmRestarts(T_0, T_mult=1, eta_min) after `epoch` scheduler steps (:126, :210)."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * (epoch % T_0) / T_0)) / 2.0


def save_reference_compatible(model: SimpleFC, path: str) -> None:
    """torch.save(model) (:237) with the pickle naming the reference's class, so that the reference's own
    `torch.load(model_file)` (_5_predict_labels.py:107) and this repo's `load_regressor` both read it."""
    cls = type(model)
    old = cls.__module__
    cls.__module__ = "utils.nn_model"
    try:
        import sys, types
        shim_pkg = sys.modules.get("utils")
        made = []
        if shim_pkg is None:
            shim_pkg = types.ModuleType("utils"); sys.modules["utils"] = shim_pkg; made.append("utils")
        shim = sys.modules.get("utils.nn_model")
        if shim is None:
            shim = types.ModuleType("utils.nn_model"); sys.modules["utils.nn_model"] = shim; made.append("utils.nn_model")
        had = getattr(shim, "SimpleFC", None)
        shim.SimpleFC = cls                                   # pickle checks that the named attribute is this class
        try:
            torch.save(model, path)
        finally:
            if had is None:
                del shim.SimpleFC
            else:
                shim.SimpleFC = had
            for m in made:
                sys.modules.pop(m, None)
    finally:
        cls.__module__ = old


def load_training_set(args, crop_names: List[str]):
    """(features float32 [n, F], labels float32 [n]) exactly as :21-84 assembles them."""
    feats, labels = [], []
    store = None
    if getattr(args, "packed_store", None):
        from .packed_store import PackedStore
        store = PackedStore(args.packed_store)
    rng = np.random.RandomState(args.random_seed)
    for name in args.train_data_names:
        data = pd.read_csv(os.path.join(args.train_data_dir, name + ".csv")).dropna(subset=["label"])
        data = data.sample(frac=1, random_state=rng).reset_index(drop=True)           # :36
        n_samples = skips = 0
        print(f"\nLoading {name} features from disk...")
        if store is not None:
            if args.clip_models_to_use[0] == "all":
                args.clip_models_to_use = store.models()
                print(f"\n----> Using all found clip models: {args.clip_models_to_use}")
            keys = [f"{name}/{u}" for u in data["uuid"]]
            found, mat = store.features(args.clip_models_to_use, crop_names, keys)
            feats.append(torch.from_numpy(np.ascontiguousarray(mat)))
            labels.extend(data["label"][found].tolist())
            n_samples, skips = int(found.sum()), int((~found).sum())
        else:
            for _, row in data.iterrows():
                try:
                    full = torch.load(f"{args.train_data_dir}/{name}/{row['uuid']}.pt", map_location="cpu", weights_only=True)
                    if args.clip_models_to_use[0] == "all":
                        args.clip_models_to_use = list(full.keys())
                        print(f"\n----> Using all found clip models: {args.clip_models_to_use}")
                    parts = []
                    for m in args.clip_models_to_use:
                        d = full[m]
                        missing = set(crop_names) - set(d.keys())
                        if missing:
                            raise Exception(f"Missing crops {missing} for {row['uuid']}")
                        parts.append(torch.cat([d[c] for c in crop_names], dim=0).flatten())
                    feats.append(torch.cat(parts, dim=0).unsqueeze(0))
                    labels.append(row["label"])
                    n_samples += 1
                except Exception:                           # simply skip the sample if something goes wrong (:73-75)
                    skips += 1
        print(f"Loaded {n_samples} samples from {name}!")
        if skips:
            print(f"(skipped {skips} samples due to loading errors)..")
    if not labels:
        raise RuntimeError("no training samples could be loaded")
    return torch.cat(feats, 0).float(), torch.tensor(labels, dtype=torch.float32)


@torch.no_grad()
def train(args, crop_names=None, device="cuda"):
    crop_names = list(crop_names or CROP_NAMES_DEFAULT)
    torch.manual_seed(args.random_seed)
    np.random.seed(args.random_seed)
    features, labels = load_training_set(args, crop_names)
    print("Normalizing labels to [0,1]...")
    print(f"min: {labels.min()}, max: {labels.max()}")
    labels = (labels - labels.min()) / (labels.max() - labels.min())
    print("\n--- All data loaded ---")
    print("Features shape:", features.shape)
    print("Labels shape:", labels.shape)
    n = features.shape[0]
    train_size = int((1 - args.test_fraction) * n)
    test_size = n - train_size
    print(f"Training on {train_size} samples, testing on {test_size} samples.")
    rs = np.random.RandomState(args.random_seed)
    perm = rs.permutation(n)                                  # random_split (:111)
    tr_idx, te_idx = perm[:train_size], perm[train_size:]
    dev = torch.device(device)
    X, T = features.to(dev).contiguous(), labels.to(dev).contiguous()
    Xte, Tte = X[torch.from_numpy(te_idx).to(dev)].contiguous(), T[torch.from_numpy(te_idx).to(dev)].contiguous()

    model = SimpleFC(features.shape[1], args.hidden_sizes, 1, args.clip_models_to_use, crop_names=crop_names,
                     dropout_prob=args.dropout_prob, verbose=args.print_network_layout)
    lin = model._linears()
    trainer = FcTrainer([m.weight for m in lin], [m.bias for m in lin], model._negative_slope(), dev)

    def test_loss():
        if test_size == 0:
            return -1.0, -1.0
        y = trainer.predict(Xte)
        tl, dl, nb = 0.0, 0.0, 0
        for b0 in range(0, test_size, args.batch_size):       # mean of per-batch MSEs, dummy = the batch's label mean (:139-146)
            yy, tt = y[b0:b0 + args.batch_size], Tte[b0:b0 + args.batch_size]
            tl += float(((yy - tt) ** 2).mean()); dl += float(((tt.mean() - tt) ** 2).mean()); nb += 1
        return tl / nb, dl / nb

    losses = [[], []]
    lrs = []
    tl, dl = test_loss()
    print(f"\nBefore training, test mse-loss: {tl:.4f} (dummy: {dl:.4f})")
    t0 = time.perf_counter()
    for epoch in range(args.n_epochs):
        lr = cosine_warm_restarts_lr(args.lr, args.min_lr, args.restart_epochs, epoch)
        order = tr_idx[rs.permutation(train_size)]            # DataLoader(shuffle=True) (:112)
        batch_losses = trainer.epoch(X, T, order, args.batch_size, lr, args.weight_decay, args.dropout_prob, args.random_seed)
        train_loss = float(batch_losses.mean())
        current_lr = cosine_warm_restarts_lr(args.lr, args.min_lr, args.restart_epochs, epoch + 1)   # scheduler.step() then get_last_lr (:210-212)
        lrs.append(current_lr)
        tl, dl = test_loss()
        losses[0].append(train_loss); losses[1].append(tl)
        if epoch % 2 == 0:
            test_str = f", test mse: {tl:.4f} (dummy: {dl:.4f})" if tl > 0 else ""
            print(f"Epoch {epoch+1}/{args.n_epochs}, train-mse: {train_loss:.4f}, lr: {current_lr:.6f}{test_str}")
    torch.cuda.synchronize()
    print(f"({args.n_epochs} epochs in {time.perf_counter() - t0:.2f} s on {dev})")
    if tl > 0:
        print(f"---> Best test mse loss: {min(losses[1]):.4f} in epoch {int(np.argmin(losses[1])) + 1}")
    Ws, bs = trainer.parameters()
    for m, W, b in zip(lin, Ws, bs):
        m.weight.copy_(W); m.bias.copy_(b)
    trainer.close()
    model.eval()
    path = None
    if not args.dont_save:
        timestamp = pd.Timestamp.now().strftime("%Y-%m-%d_%H:%M:%S")
        name = f"{args.model_name}_{timestamp}_{(train_size / 1000):.1f}k_imgs_{args.n_epochs}_epochs_{losses[1][-1]:.4f}_mse"
        os.makedirs("models", exist_ok=True)
        path = f"models/{name}.pth"
        save_reference_compatible(model, path)
        print("Final model saved to /model dir as:\n", f"{name}.pth")
    return model, losses, lrs, path


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--train_data_dir', type=str, help='Root directory of the (optionally multiple) datasets')
    parser.add_argument('--train_data_names', type=str, nargs='+', help='Names of the dataset files to train on (space separated)')
    parser.add_argument('--model_name', type=str, default='regressor', help='Name of the model when saved to disk')
    parser.add_argument('--dont_save', action='store_true', help='skip saving the model to disk')
    parser.add_argument('--clip_models_to_use', metavar='S', type=str, nargs='+', default=['all'], help='Which CLIP model embeddings to use, default: use all found')
    parser.add_argument('--test_fraction', type=float, default=0.25, help='Fraction of the training data to use for testing')
    parser.add_argument('--n_epochs', type=int, default=60, help='Number of epochs to train for')
    parser.add_argument('--batch_size', type=int, default=16, help='Batch size for training')
    parser.add_argument('--lr', type=float, default=0.0002, help='Initial learning rate')
    parser.add_argument('--min_lr', type=float, default=1e-6, help='Minimum learning rate for cosine scheduler')
    parser.add_argument('--restart_epochs', type=int, default=10, help='Number of epochs before learning rate restart')
    parser.add_argument('--weight_decay', type=float, default=0.0006, help='Weight decay for the Adam optimizer')
    parser.add_argument('--dropout_prob', type=float, default=0.5, help='Dropout probability')
    parser.add_argument('--hidden_sizes', type=int, nargs='+', default=[264, 128, 64], help='Hidden sizes of the FC neural network')
    parser.add_argument('--print_network_layout', action='store_true', help='Print the network layout')
    parser.add_argument('--random_seed', type=int, default=42, help='Random seed for reproducibility')
    parser.add_argument('--crop_names', type=str, nargs='+', default=CROP_NAMES_DEFAULT, help='Crops whose embeddings form the input (the reference edits this list in the source, :265-272)')
    parser.add_argument('--packed_store', type=str, default=None, help='Read embeddings from packed shards (keys <dataset name>/<uuid>)')
    args = parser.parse_args(argv)
    train(args, args.crop_names)


if __name__ == "__main__":
    main()
