"""Diversity ordering on the GPU (diversity_order) against the oracle (reference _3_label_images.py:128-177)."""
import random

import numpy as np
import pytest
import torch

from clip_assisted_data_labeling_amd import diversity_driver
from oracle import diversity_oracle

pytestmark = pytest.mark.gpu
# index work: the walk must be identical step for step.  Steps whose two best candidates are closer than the fp32
# accumulation-order noise of a 768-term dot product have no single right answer; the seeded cases below have none
# (asserted), so the comparison is exact.
MARGIN = 1e-6


def _emb(n, d, seed, clustered=False):
    g = torch.Generator().manual_seed(seed)
    e = torch.randn(n, d, generator=g)
    if clustered:                                          # CLIP-like: a few tight clusters, so that the maxima matter
        c = torch.randn(12, d, generator=g)
        e = c[torch.randint(0, 12, (n,), generator=g)] + 0.35 * e
    return e * (0.5 + torch.rand(n, 1, generator=g))      # rows are NOT unit norm: the kernel normalises like :130-131


@pytest.mark.parametrize("n,d,steps,k,clustered", [(400, 768, 60, 100, True), (37, 20, 16, 9, False), (5000, 512, 500, 100, True),
                                                  (2, 8, 1, 2, False), (300, 770, 40, 50, False)])
def test_walk_matches_oracle_step_for_step(gpu, n, d, steps, k, clustered):
    emb = _emb(n, d, n + d, clustered)
    samples = diversity_oracle.draw_samples(n, steps, k, seed=steps)
    assert diversity_oracle.margins(emb.numpy(), samples).min() > MARGIN
    want = diversity_oracle.diversity_order(emb.numpy(), samples, first=0)
    got = diversity_driver.diversity_order_indices(emb.to(gpu), samples, first=0).cpu().tolist()
    assert got == want


def test_strided_rows_of_a_packed_block_and_other_start(gpu):
    n, crops, d = 600, 4, 768
    block = _emb(n * crops, d, 9, True).view(n, crops, d).to(gpu)
    samples = diversity_oracle.draw_samples(n, 80, 100, seed=1)
    view = block[:, 1, :]                                  # one crop, rows crops*E apart, scanned in place
    want = diversity_oracle.diversity_order(view.cpu().numpy(), samples, first=17)
    got = diversity_driver.diversity_order_indices(view, samples, first=17).cpu().tolist()
    assert got == want


def test_ties_and_repeated_candidates(gpu):
    emb = _emb(50, 64, 4)
    emb[31] = emb[12]                                      # identical images: the earlier CANDIDATE POSITION wins (torch.argmin)
    s = np.array([[31, 12, 3, 7], [12, 31, 3, 7]], dtype=np.int32)
    for row in s:
        want = diversity_oracle.diversity_order(emb.numpy(), [row], first=0)
        got = diversity_driver.diversity_order_indices(emb.to(gpu), row.reshape(1, -1), first=0).cpu().tolist()
        assert got == want
    # a step that offers an already chosen image next to a fresh one takes the fresh one (cosine 1 with itself loses);
    # a step with ONLY chosen images is decided by the rounding of cos(x, x) in the reference, so it is not pinned here
    samples = np.array([[5, 6], [0, 7], [5, 8]], dtype=np.int32)
    want = diversity_oracle.diversity_order(emb.numpy(), samples, first=0)
    got = diversity_driver.diversity_order_indices(emb.to(gpu), samples, first=0).cpu().tolist()
    assert got == want and got[1] == 7


def test_driver_end_to_end_from_pt_files(gpu, tmp_path):
    n, d = 120, 768
    emb = _emb(n, d, 21, True)
    files = [str(tmp_path / f"im{i:03d}.jpg") for i in range(n)]
    for i, f in enumerate(files):
        torch.save({"ViT-L-14/openai": {"square_padded_crop": emb[i].unsqueeze(0)}}, f.replace(".jpg", ".pt"))
    random.seed(5)
    got = diversity_driver.diversity_ordered_image_files(files, str(tmp_path), total_n_ordered_imgs=50, sample_size=30, device=gpu)
    order = diversity_oracle.diversity_order(emb.numpy(), diversity_oracle.draw_samples(n, 50, 30, seed=5), first=0)
    assert got[:51] == [files[0]] + [files[i] for i in order] and sorted(got) == sorted(files)


def test_argument_errors(gpu):
    emb = _emb(10, 8, 0).to(gpu)
    with pytest.raises(ValueError):
        diversity_driver.diversity_order_indices(emb, np.array([[0, 10]], dtype=np.int32))
    with pytest.raises(Exception):
        diversity_driver.diversity_order_indices(emb, np.array([[0, 1]], dtype=np.int32), first=10)
