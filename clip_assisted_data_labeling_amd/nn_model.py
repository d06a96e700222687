"""Host-side mirror of the reference's regressor, backed by the fused HIP kernel.

`SimpleFC` keeps the constructor signature, attributes (`clip_models`, `crop_names`,
`use_img_stat_features`, `data_min`, `data_max`), `layers` ModuleList and therefore the
`layers.{0,3,6,9}.{weight,bias}` state-dict keys of /root/reference/utils/nn_model.py:6-41, so that
pickled reference checkpoints (`torch.save(model)`, /root/reference/_4_train_model.py:237) load into
it through `load_regressor`.  `forward` on a GPU tensor runs the whole MLP in one HIP kernel
(libclipenc_hip.so: fcreg_forward); there is no CPU forward in this build.
"""
from __future__ import annotations

import ctypes
import io
import pickle
from typing import List, Optional, Sequence

import torch
from torch import nn

from . import _lib


class SimpleFC(nn.Module):
    def __init__(self, input_size, hidden_sizes, output_size, clip_models,
                 crop_names=['centre_crop', 'square_padded_crop', 'subcrop1', 'subcrop2'],
                 use_img_stat_features=False,
                 dropout_prob=0.0,
                 data_min=None, data_max=None,
                 verbose=0):
        super().__init__()
        self.clip_models = clip_models
        self.crop_names = crop_names
        self.use_img_stat_features = use_img_stat_features
        layer_sizes = [input_size] + list(hidden_sizes) + [output_size]
        self.data_min, self.data_max = data_min, data_max
        layers = []
        for i in range(len(layer_sizes) - 1):
            layers.append(nn.Linear(layer_sizes[i], layer_sizes[i + 1]))
            if i < len(layer_sizes) - 2:
                layers.append(nn.LeakyReLU())
                layers.append(nn.Dropout(p=dropout_prob))
        layers.append(nn.Sigmoid())
        self.layers = nn.ModuleList(layers)
        if verbose > 0:
            print(self)

    # ---- HIP backing -------------------------------------------------------------------------
    def _linears(self) -> List[nn.Linear]:
        return [m for m in self.layers if isinstance(m, nn.Linear)]

    def _negative_slope(self) -> float:
        for m in self.layers:
            if isinstance(m, nn.LeakyReLU):
                return float(m.negative_slope)
        return 0.01

    def hip_regressor(self, device) -> "HipRegressor":
        key = str(torch.device(device))
        cache = self.__dict__.setdefault("_hip_cache", {})
        if key not in cache:
            lin = self._linears()
            cache[key] = HipRegressor([m.weight for m in lin], [m.bias for m in lin], self._negative_slope(), device)
        return cache[key]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training and any(isinstance(m, nn.Dropout) and m.p > 0 for m in self.layers):
            raise _lib.ClipencError("SimpleFC HIP forward is inference-only: call model.eval() first "
                                    "(the reference does, _5_predict_labels.py:108)")
        if x.device.type != "cuda":
            raise _lib.ClipencError("SimpleFC.forward needs a GPU tensor: this build has no CPU path")
        return self.hip_regressor(x.device)(x)

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_hip_cache", None)      # device handles never travel in a pickle
        return state


class HipRegressor:
    """Owner of one `fcreg_t` handle."""

    def __init__(self, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], negative_slope: float, device):
        from .embedder import _device_index
        self.lib = _lib.load()
        self.device_index = _device_index(device)
        self.device = torch.device("cuda", self.device_index)
        Ws = [w.detach().to(torch.float32).contiguous().cpu() for w in weights]
        bs = [b.detach().to(torch.float32).contiguous().cpu() for b in biases]
        n = len(Ws)
        self.sizes = [int(Ws[0].shape[1])] + [int(w.shape[0]) for w in Ws]
        for l, (w, b) in enumerate(zip(Ws, bs)):
            if tuple(w.shape) != (self.sizes[l + 1], self.sizes[l]) or tuple(b.shape) != (self.sizes[l + 1],):
                raise ValueError(f"layer {l}: inconsistent shapes {tuple(w.shape)} / {tuple(b.shape)}")
        Wp = (_lib.c_float_p * n)(*[ctypes.cast(w.data_ptr(), _lib.c_float_p) for w in Ws])
        bp = (_lib.c_float_p * n)(*[ctypes.cast(b.data_ptr(), _lib.c_float_p) for b in bs])
        h = ctypes.c_void_p()
        _lib.check(self.lib.fcreg_create(n, (ctypes.c_int * (n + 1))(*self.sizes), Wp, bp,
                                         ctypes.c_float(negative_slope), self.device_index, ctypes.byref(h)),
                   "fcreg_create")
        self.handle = h

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1]).to(self.device, torch.float32).contiguous()
        if x2.shape[1] != self.sizes[0]:
            raise ValueError(f"expected {self.sizes[0]} input features, got {x2.shape[1]}")
        y = torch.empty((x2.shape[0], self.sizes[-1]), dtype=torch.float32, device=self.device)
        if x2.shape[0]:
            _lib.check(self.lib.fcreg_forward(self.handle, x2.data_ptr(), x2.shape[0], x2.shape[1], 1, x2.shape[1],
                                              (ctypes.c_int * 1)(0), y.data_ptr(),
                                              _lib.current_stream_ptr(self.device)), "fcreg_forward")
        return y.reshape(*lead, self.sizes[-1])

    def close(self):
        if getattr(self, "handle", None):
            self.lib.fcreg_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class _RemapUnpickler(pickle.Unpickler):
    """Resolves the reference's `utils.nn_model.SimpleFC` (the class path stored by
    /root/reference/_4_train_model.py:237) to this module's SimpleFC."""

    def find_class(self, module, name):
        if name == "SimpleFC" and module.split(".")[-1] == "nn_model":
            return SimpleFC
        return super().find_class(module, name)


class _RemapPickle:
    __name__ = "clipenc_remap_pickle"
    Unpickler = _RemapUnpickler

    @staticmethod
    def load(f, **kw):
        return _RemapUnpickler(f, **kw).load()


def load_regressor(model_file: str) -> SimpleFC:
    """`torch.load(model_file)` of /root/reference/_5_predict_labels.py:107 for torch >= 2.6
    (whole-module pickles need weights_only=False) with the class path remapped."""
    model = torch.load(model_file, map_location="cpu", weights_only=False, pickle_module=_RemapPickle)
    if not isinstance(model, SimpleFC):
        raise TypeError(f"{model_file} does not hold a SimpleFC module (got {type(model).__name__})")
    return model
