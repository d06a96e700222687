"""GPU front end (-m gpu): preproc_crops_u8 is bit-exact with the Pillow path the reference uses, for every crop."""
import numpy as np
import pytest
import torch
from PIL import Image

from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
from clip_assisted_data_labeling_amd.preprocess import CROP_NAMES, ClipValTransform, GpuCropper, extract_crops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h", [(640, 427), (300, 500), (224, 224), (1000, 60), (97, 333), (2000, 1500), (225, 224), (5000, 300), (1100, 1400)])
def test_gpu_crops_are_bit_exact_with_pillow(gpu, w, h):
    """Sizes: down- and up-scaling, windows of 5 to 83 taps, a canvas wider than any staging buffer one might add (5000)."""
    rs = np.random.RandomState(w + h)
    arr = rs.randint(0, 256, (h, w, 3), dtype=np.uint8)
    # smooth content as well as noise: half of the image is a gradient
    arr[:, : w // 2] = (np.add.outer(np.arange(h), np.arange(w // 2))[:, :, None] * np.array([1, 2, 3]) % 256).astype(np.uint8)
    for R in (224, 98):
        cropper = GpuCropper(R, gpu)
        got, names = cropper(torch.from_numpy(arr))
        crops, names_ref = extract_crops(Image.fromarray(arr))
        assert names == names_ref == CROP_NAMES
        want = torch.stack([ClipValTransform(R).to_uint8(c) for c in crops])
        assert got.shape == want.shape and got.dtype == torch.uint8
        diff = (got.cpu().int() - want.int()).abs()
        assert diff.max().item() == 0, (R, [int(diff[i].max()) for i in range(4)])
        cropper.close()


def test_gpu_front_end_feeds_encoder_identically(gpu):
    cfg = vit_config.ARCHS["ViT-small-test"]
    vit = HipViT(cfg, vit_config.seeded_state_dict(cfg, 4), gpu)
    rs = np.random.RandomState(1)
    arr = rs.randint(0, 256, (333, 517, 3), dtype=np.uint8)
    cropper = GpuCropper(cfg.image_size, gpu)
    u8_gpu, _ = cropper(torch.from_numpy(arr).to(gpu))
    tf = ClipValTransform(cfg.image_size)
    f32 = torch.stack([tf(c) for c in extract_crops(Image.fromarray(arr))[0]])
    assert torch.equal(vit.encode(u8_gpu), vit.encode(f32.to(gpu)))
    with pytest.raises(ValueError):
        cropper(torch.zeros(10, 10, 4, dtype=torch.uint8))
    vit.close(); cropper.close()


def test_batched_call_equals_per_image_calls(gpu):
    """preproc_crops_u8_batch (three launches for many images; plans by kernel argument up to 32 crops, by device array
    beyond) gives the same bytes as one call per image, for mixed sizes and repeated batches."""
    from clip_assisted_data_labeling_amd.preprocess import GpuCropper
    rs = np.random.RandomState(9)
    cropper = GpuCropper(224, gpu)
    for n_img in (1, 3, 8, 9, 40):                          # 4, 12, 32 (argument path) | 36, 160 crops (device-array path)
        imgs = [torch.from_numpy(rs.randint(0, 256, (rs.randint(60, 700), rs.randint(60, 700), 3), dtype=np.uint8))
                for _ in range(n_img)]
        out, names = cropper.batch(imgs)
        assert out.shape == (4 * n_img, 3, 224, 224) and all(len(n) == 4 for n in names)
        single = torch.cat([cropper(im)[0] for im in imgs], 0)
        assert torch.equal(out, single)
    cropper.close()
