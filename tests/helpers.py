import numpy as np
import torch

CLIP_MEAN = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 3, 1, 1)
CLIP_STD = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 3, 1, 1)


def synthetic_crops(n, size, seed):
    """uint8-valued pixels pushed through the CLIP normalisation."""
    g = torch.Generator().manual_seed(seed)
    u = torch.randint(0, 256, (n, 3, size, size), generator=g).float()
    return (u / 255.0 - CLIP_MEAN) / CLIP_STD


def np_fc_weights(sizes, seed):
    rs = np.random.RandomState(seed)
    Ws, bs = [], []
    for i in range(len(sizes) - 1):
        bound = 1.0 / np.sqrt(sizes[i])
        Ws.append(rs.uniform(-bound, bound, size=(sizes[i + 1], sizes[i])).astype(np.float32))
        bs.append(rs.uniform(-bound, bound, size=(sizes[i + 1],)).astype(np.float32))
    return Ws, bs


def one_minus_cos(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    a, b = a.double(), b.double()
    return 1.0 - (a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))
