"""ViT image-tower configurations and weight containers for the CLIP embed path.

The reference never spells these shapes out: it asks open_clip for a model by name
(/root/reference/utils/embedder.py:59-74, name convention "<arch>/<pretrained>" split at :63).
The table below restates the open_clip architectures the reference can name (SURVEY.md
Appendix A.1) so that the HIP library can be configured without open_clip installed.
"""
from __future__ import annotations

import dataclasses
import math
import os
import re
from typing import Dict, Optional

import torch

ACT_QUICK_GELU = 0  # x * sigmoid(1.702 x): every "<arch>/openai" checkpoint
ACT_GELU_ERF = 1    # erf GELU: laion* / datacomp* tags


@dataclasses.dataclass(frozen=True)
class ViTConfig:
    image_size: int
    patch: int
    width: int
    layers: int
    heads: int
    mlp_dim: int
    embed_dim: int
    act: int = ACT_QUICK_GELU
    ln_eps: float = 1e-5

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + 1

    @property
    def patch_k(self) -> int:
        return 3 * self.patch * self.patch

    def macs_per_crop(self) -> int:
        """Algorithmic MACs of one crop (unpadded tokens, CLS-only projection)."""
        n, d, m = self.tokens, self.width, self.mlp_dim
        per_layer = n * d * 3 * d + 2 * n * n * d + n * d * d + 2 * n * d * m
        return (n - 1) * self.patch_k * d + self.layers * per_layer + d * self.embed_dim

    def macs_per_crop_executed(self) -> int:
        """MACs the bf16 encode path actually performs per crop: the LAST block runs its Q projection, attention,
        out-projection and MLP on the class-token row only (nothing downstream reads another token; capi.hip run_tower),
        and its attention scores that one query against the raw residual rows instead of projecting K and V
        (cls_attention.hip: Q d^2; r_h and o_h as two [heads x d] x [d x d] products per crop; scores and weighted row sums
        2 x 16 n d).  Reported next to `macs_per_crop` so that the throughput priced with the reference's full-block count
         can be told from the arithmetic really issued."""
        n, d, m = self.tokens, self.width, self.mlp_dim
        per_layer = n * d * 3 * d + 2 * n * n * d + n * d * d + 2 * n * d * m
        last = d * d + 2 * self.heads * d * d + 32 * n * d + d * d + 2 * d * m
        return (n - 1) * self.patch_k * d + (self.layers - 1) * per_layer + last + d * self.embed_dim


ARCHS: Dict[str, ViTConfig] = {
    "ViT-B-32": ViTConfig(224, 32, 768, 12, 12, 3072, 512),
    "ViT-B-16": ViTConfig(224, 16, 768, 12, 12, 3072, 512),
    "ViT-L-14": ViTConfig(224, 14, 1024, 24, 16, 4096, 768),
    "ViT-L-14-336": ViTConfig(336, 14, 1024, 24, 16, 4096, 768),
    # open_clip's ViT-H-14 (laion2b tags: erf-GELU through config_for): width 1280 = 16 heads of 80, five 256-column statistics parts;
    # the e4m3 tower runs its unfused form (row-quantised operands) for it
    "ViT-H-14": ViTConfig(224, 14, 1280, 32, 16, 5120, 1024),
    "ViT-L-16": ViTConfig(224, 16, 1024, 24, 16, 4096, 768),
    "ViT-H-16": ViTConfig(224, 16, 1280, 32, 16, 5120, 1024),
    # towers whose shapes the kernels are not built for run zero-padded (capi.hip, clipenc_create: exact arithmetic):
    # ViT-g-14's 16 heads of 88 run as heads of 96 (width 1408 -> 1536); ViT-B-16-plus-240's 14 heads of 64 as 16 (896 -> 1024)
    "ViT-g-14": ViTConfig(224, 14, 1408, 40, 16, 6144, 1024),
    "ViT-B-16-plus-240": ViTConfig(240, 16, 896, 12, 14, 3584, 640),
    # open_clip's smaller and odd-resolution towers (padding where needed: ViT-S's 6 heads run as 8; 280 / 320 px are the long attention kernel's)
    "ViT-S-32": ViTConfig(224, 32, 384, 12, 6, 1536, 384), "ViT-S-16": ViTConfig(224, 16, 384, 12, 6, 1536, 384),
    "ViT-M-32": ViTConfig(224, 32, 512, 12, 8, 2048, 512), "ViT-M-16": ViTConfig(224, 16, 512, 12, 8, 2048, 512),
    "ViT-B-32-256": ViTConfig(256, 32, 768, 12, 12, 3072, 512), "ViT-B-16-plus": ViTConfig(224, 16, 896, 12, 14, 3584, 640),
    "ViT-L-14-280": ViTConfig(280, 14, 1024, 24, 16, 4096, 768), "ViT-L-16-320": ViTConfig(320, 16, 1024, 24, 16, 4096, 768),
    "ViT-bigG-14": ViTConfig(224, 14, 1664, 48, 16, 8192, 1280),  # 16 heads of 104 run as heads of 112: 1792 columns on the device
    # small shapes used by the parity tests (not open_clip names)
    "ViT-tiny-test": ViTConfig(28, 14, 256, 2, 4, 512, 32),
    "ViT-small-test": ViTConfig(98, 14, 256, 3, 4, 1024, 64),
    "ViT-long-test": ViTConfig(336, 14, 256, 2, 4, 512, 32),      # 577 tokens like ViT-L-14-336
    "ViT-H-tiny-test": ViTConfig(28, 14, 1280, 2, 16, 5120, 64),  # head dim 80, five statistics parts, 5 tokens
    "ViT-H-mid-test": ViTConfig(98, 14, 1280, 3, 16, 5120, 64),   # ... 50 tokens: two key tiles, a partial one
    "ViT-g-tiny-test": ViTConfig(28, 14, 704, 2, 8, 1024, 32),    # 8 heads of 88 -> heads of 96 (width 768 on the device), 5 tokens
    "ViT-g-mid-test": ViTConfig(98, 14, 704, 3, 8, 1024, 64),     # ... 50 tokens
    "ViT-g-wide-test": ViTConfig(98, 14, 1232, 2, 14, 1024, 64),  # 14 heads of 88 -> 16 of 96: 1536 columns on the device (six statistics parts)
    "ViT-bigG-tiny-test": ViTConfig(28, 14, 1664, 2, 16, 2048, 1280),   # heads of 104 -> 112 (1792 columns), the 1280-wide embedding, 5 tokens
    "ViT-bigG-mid-test": ViTConfig(98, 14, 1664, 2, 16, 2048, 96),     # ... 50 tokens
    "ViT-hd120-test": ViTConfig(98, 14, 240, 2, 2, 512, 32),            # two heads of 120 -> 128 (256 columns)
    "ViT-pad-test": ViTConfig(98, 14, 288, 3, 6, 640, 32),        # every kind of padding: heads of 48 -> 64, 6 heads -> 8, mlp 640 -> 768
}


def config_for(model_name: str) -> ViTConfig:
    """'ViT-L-14/openai' -> ViTConfig. Unknown arch raises ValueError like _1_embed_with_CLIP.py:75."""
    arch, _, pretrained = model_name.partition("/")
    if arch not in ARCHS:
        raise ValueError(f"Unknown model architecture: {arch!r} (known: {sorted(ARCHS)})")
    cfg = ARCHS[arch]
    if pretrained and pretrained != "openai" and not pretrained.startswith("seed"):
        cfg = dataclasses.replace(cfg, act=ACT_GELU_ERF)
    return cfg


def state_dict_keys(cfg: ViTConfig):
    """(key, shape) pairs of the OpenAI / open_clip `visual.` state dict."""
    d, m = cfg.width, cfg.mlp_dim
    keys = [
        ("conv1.weight", (d, 3, cfg.patch, cfg.patch)),
        ("class_embedding", (d,)),
        ("positional_embedding", (cfg.tokens, d)),
        ("ln_pre.weight", (d,)), ("ln_pre.bias", (d,)),
    ]
    for l in range(cfg.layers):
        p = f"transformer.resblocks.{l}."
        keys += [
            (p + "ln_1.weight", (d,)), (p + "ln_1.bias", (d,)),
            (p + "attn.in_proj_weight", (3 * d, d)), (p + "attn.in_proj_bias", (3 * d,)),
            (p + "attn.out_proj.weight", (d, d)), (p + "attn.out_proj.bias", (d,)),
            (p + "ln_2.weight", (d,)), (p + "ln_2.bias", (d,)),
            (p + "mlp.c_fc.weight", (m, d)), (p + "mlp.c_fc.bias", (m,)),
            (p + "mlp.c_proj.weight", (d, m)), (p + "mlp.c_proj.bias", (d,)),
        ]
    keys += [("ln_post.weight", (d,)), ("ln_post.bias", (d,)), ("proj", (d, cfg.embed_dim))]
    return keys


def seeded_state_dict(cfg: ViTConfig, seed: int = 0, randomize_affine: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded fp32 weights with OpenAI-style init scales.

    There is no network in the build/bench environment, so parity and throughput runs use these
    instead of the `openai` checkpoint. LayerNorm affine terms and biases are randomised (unlike a
    fresh init) so that every term of the forward is exercised by the parity tests.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    d, L = cfg.width, cfg.layers
    attn_std = d ** -0.5
    proj_std = (d ** -0.5) * ((2 * L) ** -0.5)
    fc_std = (2 * d) ** -0.5
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in state_dict_keys(cfg):
        if key == "conv1.weight":
            t = torch.randn(shape, generator=g) * (cfg.patch_k ** -0.5)
        elif key in ("class_embedding", "positional_embedding", "proj"):
            t = torch.randn(shape, generator=g) * attn_std
        elif key.endswith("in_proj_weight"):
            t = torch.randn(shape, generator=g) * attn_std
        elif key.endswith("out_proj.weight") or key.endswith("c_proj.weight"):
            t = torch.randn(shape, generator=g) * proj_std
        elif key.endswith("c_fc.weight"):
            t = torch.randn(shape, generator=g) * fc_std
        elif key.endswith("ln_1.weight") or key.endswith("ln_2.weight") or key in ("ln_pre.weight", "ln_post.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g) if randomize_affine else torch.ones(shape)
        else:  # biases
            t = 0.02 * torch.randn(shape, generator=g) if randomize_affine else torch.zeros(shape)
        sd[key] = t.float().contiguous()
    return sd


def normalise_state_dict(raw: Dict[str, torch.Tensor], cfg: ViTConfig) -> Dict[str, torch.Tensor]:
    """Accept a full CLIP state dict ('visual.' prefix), a bare visual dict, or a
    `transformers` CLIPVisionModelWithProjection dict, and return the OpenAI `visual.` layout in fp32."""
    if any(k.startswith("visual.") for k in raw):
        raw = {k[len("visual."):]: v for k, v in raw.items() if k.startswith("visual.")}
    if any(k.startswith("vision_model.") for k in raw):
        raw = _from_transformers(raw, cfg)
    out = {}
    for key, shape in state_dict_keys(cfg):
        if key not in raw:
            raise KeyError(f"weight {key!r} missing from state dict")
        t = raw[key].detach().to(torch.float32).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"weight {key!r} has shape {tuple(t.shape)}, expected {shape}")
        out[key] = t
    return out


def _from_transformers(hf: Dict[str, torch.Tensor], cfg: ViTConfig) -> Dict[str, torch.Tensor]:
    """Key mapping of SURVEY.md Appendix A.3 (split q/k/v -> in_proj, visual_projection^T -> proj)."""
    sd = {
        "conv1.weight": hf["vision_model.embeddings.patch_embedding.weight"],
        "class_embedding": hf["vision_model.embeddings.class_embedding"],
        "positional_embedding": hf["vision_model.embeddings.position_embedding.weight"],
        "ln_pre.weight": hf["vision_model.pre_layrnorm.weight"],
        "ln_pre.bias": hf["vision_model.pre_layrnorm.bias"],
        "ln_post.weight": hf["vision_model.post_layernorm.weight"],
        "ln_post.bias": hf["vision_model.post_layernorm.bias"],
        "proj": hf["visual_projection.weight"].t().contiguous(),
    }
    for l in range(cfg.layers):
        s = f"vision_model.encoder.layers.{l}."
        p = f"transformer.resblocks.{l}."
        sd[p + "ln_1.weight"] = hf[s + "layer_norm1.weight"]
        sd[p + "ln_1.bias"] = hf[s + "layer_norm1.bias"]
        sd[p + "ln_2.weight"] = hf[s + "layer_norm2.weight"]
        sd[p + "ln_2.bias"] = hf[s + "layer_norm2.bias"]
        sd[p + "attn.in_proj_weight"] = torch.cat(
            [hf[s + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        sd[p + "attn.in_proj_bias"] = torch.cat(
            [hf[s + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        sd[p + "attn.out_proj.weight"] = hf[s + "self_attn.out_proj.weight"]
        sd[p + "attn.out_proj.bias"] = hf[s + "self_attn.out_proj.bias"]
        sd[p + "mlp.c_fc.weight"] = hf[s + "mlp.fc1.weight"]
        sd[p + "mlp.c_fc.bias"] = hf[s + "mlp.fc1.bias"]
        sd[p + "mlp.c_proj.weight"] = hf[s + "mlp.fc2.weight"]
        sd[p + "mlp.c_proj.bias"] = hf[s + "mlp.fc2.bias"]
    return sd


# file names open_clip leaves in `cache_dir` for pretrained='openai' (its download keeps the URL's basename; these are
# OpenAI's TorchScript archives) [upstream: open_clip pretrained.py]
_OPENAI_JIT_NAMES = {"ViT-B-32": "ViT-B-32.pt", "ViT-B-16": "ViT-B-16.pt", "ViT-L-14": "ViT-L-14.pt",
                     "ViT-L-14-336": "ViT-L-14-336px.pt"}
_HUB_FILES = ("open_clip_model.safetensors", "open_clip_pytorch_model.bin", "model.safetensors", "pytorch_model.bin")
_TIMM_ARCH = {"ViT-B-32": "vit_base_patch32_clip_224", "ViT-B-16": "vit_base_patch16_clip_224",
              "ViT-L-14": "vit_large_patch14_clip_224", "ViT-L-14-336": "vit_large_patch14_clip_336"}


def _read_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """One checkpoint file -> flat {name: tensor}.  Handles safetensors, plain `torch.save`d state dicts (optionally
    wrapped in {'state_dict': ...}) and TorchScript archives (OpenAI's `ViT-L-14.pt` is one: `torch.load` with
    `weights_only=True` rejects it, so the archive is opened with `torch.jit.load` and only its state dict is kept --
    the scripted code is never run)."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    try:
        raw = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as plain_err:                 # TorchScript archive (RuntimeError) or pickled non-tensor objects
        try:
            raw = torch.jit.load(path, map_location="cpu").state_dict()
        except Exception as jit_err:
            raise RuntimeError(f"cannot read checkpoint {path}: not a state dict ({plain_err}) nor a TorchScript "
                               f"archive ({jit_err})") from jit_err
    if isinstance(raw, dict) and "state_dict" in raw and isinstance(raw["state_dict"], dict):
        raw = raw["state_dict"]
    if isinstance(raw, dict) and any(k.startswith("module.") for k in raw):      # DataParallel-saved training checkpoints
        raw = {k[len("module."):] if k.startswith("module.") else k: v for k, v in raw.items()}
    return {k: v for k, v in raw.items() if isinstance(v, torch.Tensor)}


# pretrained tag -> what the hub repository of that tag is called (lower case, '_' -> '-').  open_clip resolves a tag to
# exactly one repository [upstream: open_clip pretrained.py]; a snapshot is accepted only when its repository name carries
# BOTH the architecture and one of the tag's names, so that e.g. a LAION ViT-L-14 snapshot is never loaded for 'openai'.
_TAG_ALIASES = {"openai": ("openai",), "laion2b_s32b_b82k": ("laion2b-s32b-b82k", "laion2b"),
                "laion2b_s34b_b79k": ("laion2b-s34b-b79k", "laion2b"), "laion400m_e31": ("laion400m-e31", "laion400m"),
                "laion400m_e32": ("laion400m-e32", "laion400m"), "datacomp_xl_s13b_b90k": ("datacomp-xl-s13b-b90k", "datacomp"),
                "dfn2b": ("dfn2b",),
                # ViT-H-14, ViT-g-14, ViT-bigG-14 (the bigG repository spells its tag without the 's': CLIP-ViT-bigG-14-laion2B-39B-b160k)
                "laion2b_s32b_b79k": ("laion2b-s32b-b79k", "laion2b"), "laion2b_s12b_b42k": ("laion2b-s12b-b42k", "laion2b"),
                "laion2b_s34b_b88k": ("laion2b-s34b-b88k", "laion2b"), "laion2b_s39b_b160k": ("laion2b-39b-b160k", "laion2b-s39b-b160k", "laion2b")}


# what open_clip appends to an architecture name to make ANOTHER architecture: 'ViT-B-16-plus-240', 'ViT-B-32-quickgelu',
# 'ViT-L-14-CLIPA-336', 'ViT-B-16-SigLIP-256', 'ViT-L-14-336', 'ViT-B-32-256', ...
_ARCH_QUALIFIERS = r"-(?:\d|plus|quickgelu|clipa|siglip|so400m|worldwide|xl|bigg|g\b)"


def _has_component(name: str, part: str) -> bool:
    """`part` occurs in `name` as a whole architecture name: not glued to more letters/digits on either side and not followed
    by a qualifier that names another architecture ('vit-l-14' must not accept a 'vit-l-14-336' repository, 'vit-b-16' not a
    'vit-b-16-plus-240' or 'vit-b-16-siglip' one)."""
    for m in re.finditer(re.escape(part), name):
        before = name[m.start() - 1] if m.start() > 0 else "-"
        if not before.isalnum() and not re.match(r"[a-z0-9]|" + _ARCH_QUALIFIERS, name[m.end():]):
            return True
    return False


def _hub_snapshots(cache_dir: str, arch: str, pretrained: str):
    """Checkpoint files inside a Hugging Face hub cache tree (`models--<org>--<repo>/snapshots/<rev>/<file>`), which is
    what open_clip leaves in `cache_dir` for hf-hub hosted tags.  Only repositories whose name holds the architecture
    AND the pretrained tag qualify; the longest tag match comes first."""
    import glob
    found = []
    arch_names = [arch.lower()]
    if arch in _TIMM_ARCH:
        arch_names.append(_TIMM_ARCH[arch].replace("_", "-"))
    tag_l = pretrained.lower().replace("_", "-")
    tag_names = tuple(dict.fromkeys((tag_l,) + _TAG_ALIASES.get(pretrained.lower(), ())))
    for repo in sorted(glob.glob(os.path.join(cache_dir, "models--*"))):
        name = os.path.basename(repo).lower().replace("_", "-")
        arch_hit = any(_has_component(name, a) for a in arch_names)
        tag_hit = max((len(t) for t in tag_names if t and t in name), default=0)
        if not (arch_hit and tag_hit):
            continue
        for fname in _HUB_FILES:
            hits = sorted(glob.glob(os.path.join(repo, "snapshots", "*", fname)))
            if hits:
                found.append((-tag_hit, hits[-1]))
                break
    return [p for _, p in sorted(found)]


def load_weights(model_name: str, model_path: Optional[str]) -> Dict[str, torch.Tensor]:
    """Resolve the weights the reference would have fetched through open_clip
    (/root/reference/utils/embedder.py:66-73, `cache_dir=model_path`).

    * pretrained tag 'seed<N>'  -> seeded synthetic weights (bench / parity);
    * `model_path` is a file     -> that file;
    * `model_path` is a directory, searched in this order:
        1. `<arch>-<pretrained>.{pt,pth,bin,safetensors}` / `<arch>_<pretrained>.*` (a hand-placed file: the exact name);
        2. what open_clip's own download leaves there for the `openai` tag: `ViT-L-14.pt`, `ViT-L-14-336px.pt`,
           `ViT-B-32.pt`, `ViT-B-16.pt` (TorchScript archives, fp16);
        3. a Hugging Face hub cache tree `models--*/snapshots/*/open_clip_model.safetensors|open_clip_pytorch_model.bin`
           (hf-hub hosted tags, and `openai` in newer open_clip releases) -- only repositories whose name carries both
           the architecture and the pretrained tag: open_clip resolves exactly the named tag, and so does this.
      Files may hold an OpenAI / open_clip state dict (with or without the `visual.` prefix), a `transformers`
      CLIPVisionModelWithProjection dict, in any float dtype (fp16 archives are up-cast).
    No download is attempted: a missing file raises FileNotFoundError.
    """
    cfg = config_for(model_name)
    arch, _, pretrained = model_name.partition("/")
    if pretrained.startswith("seed"):
        return seeded_state_dict(cfg, int(pretrained[4:] or 0))
    candidates = []
    if model_path and os.path.isfile(model_path):
        candidates.append(model_path)
    elif model_path:
        for ext in ("pt", "pth", "bin", "safetensors"):       # the exact, hand-placed name wins over every search
            candidates.append(os.path.join(model_path, f"{arch}-{pretrained}.{ext}"))
            candidates.append(os.path.join(model_path, f"{arch}_{pretrained}.{ext}"))
        if pretrained == "openai" and arch in _OPENAI_JIT_NAMES:
            candidates.append(os.path.join(model_path, _OPENAI_JIT_NAMES[arch]))
        candidates += _hub_snapshots(model_path, arch, pretrained)
    errors = []
    for path in candidates:
        if not os.path.isfile(path):
            continue
        try:
            return normalise_state_dict(_read_checkpoint(path), cfg)
        except (KeyError, ValueError) as e:          # another architecture's checkpoint in the same cache: keep looking
            errors.append(f"{path}: {e}")
    if errors:
        raise ValueError(f"No checkpoint under {model_path!r} fits {model_name!r}: " + "; ".join(errors))
    raise FileNotFoundError(
        f"No local weights for {model_name!r}: looked for {candidates or '(no model_path given)'}; "
        "this build never downloads checkpoints. Pass model_path=<dir or file>, or use the "
        f"synthetic tag '{arch}/seed0'.")
