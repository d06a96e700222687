"""Host-side mirror of the reference's encoder wrapper, backed by the HIP library.

`CLIP_Encoder` keeps the constructor, attributes and methods of
/root/reference/utils/embedder.py:58-100 (`CLIP_Encoder(model_name, model_path=None, device=None)`,
`.img_resolution/.model_name/.device`, `get_preprocess_transform()`, `encode_image(Tensor)`), so
`Feature_Dataset` (/root/reference/_1_embed_with_CLIP.py:69-78,130) can use it unchanged.
The arithmetic runs in libclipenc_hip.so on an MI355X; there is no CPU path.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import torch

from . import _lib
from .preprocess import clip_val_transform
from .vit_config import ViTConfig, config_for, load_weights, normalise_state_dict

_DEVICE = "cuda"


def _device_index(device) -> int:
    d = torch.device(device)
    if d.type != "cuda":
        raise _lib.ClipencError(
            f"CLIP_Encoder needs an AMD GPU device ('cuda[:i]' under PyTorch-ROCm), got {device!r}: "
            "this build has no CPU path")
    return d.index if d.index is not None else torch.cuda.current_device()


class _WeightPack:
    """Keeps the ctypes view (and the tensors behind it) alive for the duration of clipenc_create."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg: ViTConfig):
        self.keep = []
        w = _lib.clipenc_weights()

        def ptr(t):
            t = t.detach().to(torch.float32).contiguous().cpu()
            self.keep.append(t)
            return ctypes.cast(t.data_ptr(), _lib.c_float_p)

        def per_layer(fmt):
            arr = (_lib.c_float_p * cfg.layers)(*[ptr(sd[fmt.format(l)]) for l in range(cfg.layers)])
            self.keep.append(arr)
            return ctypes.cast(arr, _lib.c_float_pp)

        w.conv1_weight = ptr(sd["conv1.weight"])
        w.class_embedding = ptr(sd["class_embedding"])
        w.positional_embedding = ptr(sd["positional_embedding"])
        w.ln_pre_w, w.ln_pre_b = ptr(sd["ln_pre.weight"]), ptr(sd["ln_pre.bias"])
        p = "transformer.resblocks.{}."
        w.ln_1_w, w.ln_1_b = per_layer(p + "ln_1.weight"), per_layer(p + "ln_1.bias")
        w.in_proj_w, w.in_proj_b = per_layer(p + "attn.in_proj_weight"), per_layer(p + "attn.in_proj_bias")
        w.out_proj_w, w.out_proj_b = per_layer(p + "attn.out_proj.weight"), per_layer(p + "attn.out_proj.bias")
        w.ln_2_w, w.ln_2_b = per_layer(p + "ln_2.weight"), per_layer(p + "ln_2.bias")
        w.c_fc_w, w.c_fc_b = per_layer(p + "mlp.c_fc.weight"), per_layer(p + "mlp.c_fc.bias")
        w.c_proj_w, w.c_proj_b = per_layer(p + "mlp.c_proj.weight"), per_layer(p + "mlp.c_proj.bias")
        w.ln_post_w, w.ln_post_b = ptr(sd["ln_post.weight"]), ptr(sd["ln_post.bias"])
        w.proj = ptr(sd["proj"])
        self.struct = w


class HipViT:
    """Owner of one `clipenc_t` handle (weights + workspace on one GPU)."""

    PRECISIONS = {"bf16": 0, "fp8": 1}

    def __init__(self, cfg: ViTConfig, state_dict: Dict[str, torch.Tensor], device="cuda", chunk_crops: Optional[int] = None,
                 precision: str = "bf16"):
        self.lib = _lib.load()
        self.cfg = cfg
        self.device_index = _device_index(device)
        self.device = torch.device("cuda", self.device_index)
        sd = normalise_state_dict(state_dict, cfg)
        pack = _WeightPack(sd, cfg)
        c = _lib.clipenc_config(cfg.image_size, cfg.patch, cfg.width, cfg.layers, cfg.heads, cfg.mlp_dim,
                                cfg.embed_dim, cfg.act, cfg.ln_eps)
        h = ctypes.c_void_p()
        _lib.check(self.lib.clipenc_create(ctypes.byref(c), ctypes.byref(pack.struct), self.device_index,
                                           ctypes.byref(h)), "clipenc_create")
        self.handle = h
        if chunk_crops:
            self.set_chunk(chunk_crops)
        self.precision = "bf16"
        if precision != "bf16":
            self.set_precision(precision)

    def set_precision(self, precision: str) -> None:
        """"bf16" (default) or "fp8" (e4m3 block GEMMs, BASELINE.json configs[3]); switchable at any time."""
        if precision not in self.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(self.PRECISIONS)}, got {precision!r}")
        _lib.check(self.lib.clipenc_set_precision(self.handle, self.PRECISIONS[precision]), "clipenc_set_precision")
        self.precision = precision

    def set_chunk(self, chunk_crops: int) -> None:
        _lib.check(self.lib.clipenc_set_chunk(self.handle, int(chunk_crops)), "clipenc_set_chunk")

    def _check_crops(self, crops: torch.Tensor):
        R = self.cfg.image_size
        if crops.dim() != 4 or tuple(crops.shape[1:]) != (3, R, R):
            raise ValueError(f"expected crops of shape [n, 3, {R}, {R}], got {tuple(crops.shape)}")
        if crops.device != self.device:
            crops = crops.to(self.device)
        if crops.dtype == torch.float32:
            dt = 0
        elif crops.dtype == torch.float16:
            dt = 1
        elif crops.dtype == torch.uint8:          # raw pixels: ToTensor + Normalize happen in the patchify kernel
            dt = 2
        else:
            crops, dt = crops.float(), 0
        return crops.contiguous(), dt

    @torch.no_grad()
    def encode(self, crops: torch.Tensor, normalize: bool = True) -> torch.Tensor:
        crops, dt = self._check_crops(crops)
        n = crops.shape[0]
        out = torch.empty((n, self.cfg.embed_dim), dtype=torch.float32, device=self.device)
        if n:
            _lib.check(self.lib.clipenc_encode(self.handle, crops.data_ptr(), n, dt, out.data_ptr(),
                                               1 if normalize else 0, _lib.current_stream_ptr(self.device)),
                       "clipenc_encode")
        return out

    @torch.no_grad()
    def forward_tokens(self, crops: torch.Tensor, n_layers: Optional[int] = None) -> torch.Tensor:
        """Token-level features: the bf16 residual stream [n, tokens, width] after `n_layers` blocks (default: all)."""
        crops, dt = self._check_crops(crops)
        n = crops.shape[0]
        n_layers = self.cfg.layers if n_layers is None else n_layers
        x = torch.empty((n, self.cfg.tokens, self.cfg.width), dtype=torch.bfloat16, device=self.device)
        _lib.check(self.lib.clipenc_forward_tokens(self.handle, crops.data_ptr(), n, dt, n_layers, x.data_ptr(),
                                                   _lib.current_stream_ptr(self.device)), "clipenc_forward_tokens")
        return x

    debug_run_layers = forward_tokens      # name used by the parity tests

    @torch.no_grad()
    def encode_score(self, crops: torch.Tensor, regressor, crops_per_image: int, crop_select):
        """Fused encode + regressor (clipenc_encode_score): returns (emb [n_img, crops, E], score [n_img, out])."""
        crops, dt = self._check_crops(crops)
        n = crops.shape[0]
        if n % crops_per_image:
            raise ValueError(f"{n} crops is not a multiple of {crops_per_image} crops per image")
        n_img = n // crops_per_image
        emb = torch.empty((n_img, crops_per_image, self.cfg.embed_dim), dtype=torch.float32, device=self.device)
        score = torch.empty((n_img, regressor.sizes[-1]), dtype=torch.float32, device=self.device)
        sel = (ctypes.c_int * len(crop_select))(*crop_select)
        _lib.check(self.lib.clipenc_encode_score(self.handle, regressor.handle, crops.data_ptr(), n_img, crops_per_image,
                                                 dt, sel, len(crop_select), emb.data_ptr(), score.data_ptr(),
                                                 _lib.current_stream_ptr(self.device)), "clipenc_encode_score")
        return emb, score

    def profile_enable(self, on: bool) -> None:
        _lib.check(self.lib.clipenc_profile_enable(self.handle, 1 if on else 0), "clipenc_profile_enable")

    def profile_read(self, reset: bool = True):
        """{kernel name: (total_ms, launches, algorithmic_flops)} since the last reset."""
        out = {}
        for k in range(self.lib.clipenc_profile_kinds()):
            name = ctypes.c_char_p(); ms = ctypes.c_double(); n = ctypes.c_longlong(); fl = ctypes.c_double()
            _lib.check(self.lib.clipenc_profile_read(self.handle, k, ctypes.byref(name), ctypes.byref(ms), ctypes.byref(n),
                                                     ctypes.byref(fl), 1 if reset else 0), "clipenc_profile_read")
            out[name.value.decode()] = (ms.value, n.value, fl.value)
        return out

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.clipenc_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class CLIP_Encoder:
    """Same surface as /root/reference/utils/embedder.py:58-100."""

    accepts_uint8 = True      # encode_image also takes uint8 [n,3,R,R] (output of get_preprocess_transform().to_uint8)

    def __init__(self, model_name, model_path=None, device=None, precision="bf16"):
        self.device = device if device else _DEVICE
        self.precision = precision                   # the reference runs fp16 on cuda (:61); here bf16 MFMA, or "fp8"
        self.model_name = model_name
        self.model_architecture, self.pretrained_dataset = self.model_name.split("/", 2)   # :63
        print(f"Loading CLIP model {self.model_name}...")
        self.config = config_for(model_name)
        self.model = HipViT(self.config, load_weights(model_name, model_path), self.device, precision=precision)
        self.preprocess = clip_val_transform(self.config.image_size)
        self.img_resolution = self.config.image_size                                       # :76-85
        print(f"CLIP model {self.model_name} with img_resolution {self.img_resolution} loaded on {self.device}!")

    def get_preprocess_transform(self):
        return self.preprocess

    @torch.no_grad()
    def encode_image(self, preprocessed_images: torch.Tensor) -> torch.Tensor:
        """[n,3,R,R] -> L2-normalised [n,E] float32 on the same device (utils/embedder.py:94-100)."""
        return self.model.encode(preprocessed_images, normalize=True)


class AestheticRegressor:
    """The intended behaviour of /root/reference/utils/embedder.py:277-311 (defective at HEAD, SURVEY.md
    Appendix D #3): image -> the regressor's crops -> encode -> [model][crop][E] features -> score, with
    encode + crop selection + regressor fused in one library call (`clipenc_encode_score`) when the
    regressor was trained on a single CLIP model.

    `predict_score(pil_img)` returns `(score: float, features: Tensor[1, n_models*n_crops*E])` like the
    reference; `predict_scores(list_of_pil)` is the batched form.
    """

    def __init__(self, model_path, device="cuda", clip_model_path=None, verbose=1):
        from .nn_model import load_regressor
        from .preprocess import CROP_NAMES, extract_crops
        self.device = device
        self._extract_crops = extract_crops
        self._all_crops = list(CROP_NAMES)
        self.model = load_regressor(model_path).eval()
        if verbose:
            print("Loaded regression model")
            print("Aesthetic Regressor was trained on embeddings from CLIP models:")
            print(self.model.clip_models)
            print("Aesthetic Regressor used crops:")
            print(self.model.crop_names)
        unknown = [c for c in self.model.crop_names if c not in self._all_crops]
        if unknown:
            raise ValueError(f"regressor uses crop names this build does not produce: {unknown}")
        self.clip_models = [CLIP_Encoder(name, clip_model_path, device=self.device) for name in self.model.clip_models]
        self._reg = self.model.hip_regressor(self.clip_models[0].model.device)
        self._select = [self._all_crops.index(c) for c in self.model.crop_names]

    @torch.no_grad()
    def predict_scores(self, pil_imgs):
        n = len(pil_imgs)
        feats = []
        score = None
        for enc in self.clip_models:
            tf = enc.get_preprocess_transform()
            crops = torch.stack([tf(c) for img in pil_imgs for c in self._extract_crops(img.convert("RGB"), self._all_crops)[0]])
            if len(self.clip_models) == 1:
                emb, score = enc.model.encode_score(crops, self._reg, len(self._all_crops), self._select)
            else:
                emb = enc.model.encode(crops).view(n, len(self._all_crops), -1)
            feats.append(emb[:, self._select, :].reshape(n, -1))
        features = torch.cat(feats, dim=1)
        if score is None:
            score = self._reg(features)
        return score.reshape(n, -1)[:, 0], features

    @torch.no_grad()
    def predict_score(self, pil_img):
        score, features = self.predict_scores([pil_img])
        return float(score[0].item()), features
