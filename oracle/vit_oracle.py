"""TEST INFRASTRUCTURE ONLY — fp32 CPU restatement of `CLIP_Encoder.encode_image`.

Follows /root/reference/utils/embedder.py:94-100 (wrapper: encode, then in-place L2 normalise
with no epsilon) and, for the arithmetic the reference delegates to the un-vendored, un-pinned
third-party module `open_clip` (`open_clip_torch`, version not pinned anywhere in the reference;
call sites utils/embedder.py:66-73 and :98), the published OpenAI-CLIP / open_clip
`VisionTransformer.forward` as restated in SURVEY.md Appendix A.2.

Parity pinning: the reference holds no test, golden vector or fixture for this boundary
("parity unpinned" at the open_clip call). The restatement is instead pinned against an
independent implementation of the same tower, `transformers.CLIPVisionModelWithProjection`
(tests/golden/make_golden.py asserts max-abs < 1e-5 on seeded weights through the key mapping of
SURVEY.md Appendix A.3), and the resulting vectors are committed under tests/golden/ -- since round 5
also at FULL size (ViT-L-14 and ViT-L-14-336: 1024 wide x 24 blocks x 257 / 577 tokens, measured
8.2e-8), with the transformers embeddings stored next to this restatement's (`emb_transformers`).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


def _act(u: torch.Tensor, act: int) -> torch.Tensor:
    if act == 0:  # QuickGELU (every */openai checkpoint)
        return u * torch.sigmoid(1.702 * u)
    return F.gelu(u)  # erf GELU


@torch.no_grad()
def vit_forward(sd: Dict[str, torch.Tensor], cfg, crops: torch.Tensor,
                taps: Optional[dict] = None) -> torch.Tensor:
    """Un-normalised image features [C, E]; `crops` is float32 [C, 3, R, R] (already normalised)."""
    x = crops.to(torch.float32)
    C = x.shape[0]
    d, H = cfg.width, cfg.heads
    dh = d // H
    # A.2 step 1: patch conv, stride = patch, no bias; patches row-major over (gy, gx)
    p = F.conv2d(x, sd["conv1.weight"], bias=None, stride=cfg.patch)      # [C, D, g, g]
    p = p.reshape(C, d, -1).permute(0, 2, 1)                               # [C, g*g, D]
    # step 2: class token first, then add positional embedding
    cls = sd["class_embedding"].reshape(1, 1, d).expand(C, 1, d)
    x = torch.cat([cls, p], dim=1) + sd["positional_embedding"].unsqueeze(0)
    # step 3
    x = F.layer_norm(x, (d,), sd["ln_pre.weight"], sd["ln_pre.bias"], cfg.ln_eps)
    if taps is not None:
        taps["ln_pre"] = x.clone()
    for l in range(cfg.layers):
        pre = f"transformer.resblocks.{l}."
        a = F.layer_norm(x, (d,), sd[pre + "ln_1.weight"], sd[pre + "ln_1.bias"], cfg.ln_eps)
        qkv = a @ sd[pre + "attn.in_proj_weight"].t() + sd[pre + "attn.in_proj_bias"]
        q, k, v = qkv.split(d, dim=-1)                                      # [q | k | v]
        q = q.reshape(C, -1, H, dh).transpose(1, 2)
        k = k.reshape(C, -1, H, dh).transpose(1, 2)
        v = v.reshape(C, -1, H, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
        o = torch.softmax(s, dim=-1) @ v                                    # no mask
        o = o.transpose(1, 2).reshape(C, -1, d)
        x = x + o @ sd[pre + "attn.out_proj.weight"].t() + sd[pre + "attn.out_proj.bias"]
        b = F.layer_norm(x, (d,), sd[pre + "ln_2.weight"], sd[pre + "ln_2.bias"], cfg.ln_eps)
        h = _act(b @ sd[pre + "mlp.c_fc.weight"].t() + sd[pre + "mlp.c_fc.bias"], cfg.act)
        x = x + h @ sd[pre + "mlp.c_proj.weight"].t() + sd[pre + "mlp.c_proj.bias"]
        if taps is not None:
            taps[f"block{l}"] = x.clone()
    c = F.layer_norm(x[:, 0, :], (d,), sd["ln_post.weight"], sd["ln_post.bias"], cfg.ln_eps)
    return c @ sd["proj"]


@torch.no_grad()
def encode_image(sd, cfg, crops: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
    """utils/embedder.py:94-100 on the CPU (precision 'fp32' branch): features / ||features||_2."""
    f = vit_forward(sd, cfg, crops, taps)
    f = f / f.norm(dim=-1, keepdim=True)
    return f
