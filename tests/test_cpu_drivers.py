"""Host logic of the drivers on CPU: file discovery, `.pt` schema, resume rule, feature assembly.
The encoder is replaced by a test double here; the real HIP path is exercised in test_gpu_drivers.py."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from clip_assisted_data_labeling_amd import embed_driver, predict_driver
from clip_assisted_data_labeling_amd.preprocess import CROP_NAMES, ClipValTransform


class FakeEncoder:
    """Deterministic stand-in: embedding = per-crop mean colour statistics (3 numbers, normalised)."""
    img_resolution = 32

    def __init__(self):
        self.calls = 0

    def get_preprocess_transform(self):
        return ClipValTransform(32)

    def encode_image(self, x):
        self.calls += 1
        f = x.float().mean(dim=(2, 3)) + torch.tensor([0.1, 0.2, 0.3])
        return f / f.norm(dim=-1, keepdim=True)


def _make_images(root, n, seed=0):
    rs = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, "sub"), exist_ok=True)
    paths = []
    for i in range(n):
        w, h = rs.randint(40, 90), rs.randint(40, 90)
        p = os.path.join(root, "sub" if i % 2 else "", f"img{i:03d}" + (".jpg" if i % 3 else ".PNG"))
        Image.fromarray(rs.randint(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
        paths.append(p)
    open(os.path.join(root, "notes.txt"), "w").write("x")
    open(os.path.join(root, "broken.jpg"), "wb").write(b"not an image")
    return paths


def test_embed_driver_schema_resume_and_merge(tmp_path):
    root = str(tmp_path / "data")
    os.makedirs(root)
    paths = _make_images(root, 7)
    enc = FakeEncoder()
    ds = embed_driver.Feature_Dataset(root, "Fake-A/test", 3, shuffle_filenames=False, encoder=enc, device="cpu")
    assert len(ds) == 8                                     # 7 images + broken.jpg; notes.txt ignored
    n_emb, n_skip, n_fail = ds.process()
    assert (n_emb, n_skip, n_fail) == (7, 0, 1)
    for p in paths:
        d = torch.load(os.path.splitext(p)[0] + ".pt", weights_only=True)      # plain tensors/strs only
        assert list(d) == ["Fake-A/test"]
        assert list(d["Fake-A/test"]) == CROP_NAMES           # every image has all four crop keys
        for c in CROP_NAMES:
            t = d["Fake-A/test"][c]
            assert t.shape == (1, 3) and t.dtype == torch.float32
        # each key holds THAT crop of THAT image (the reference's B>=2 mis-keying is not reproduced)
        from clip_assisted_data_labeling_amd.preprocess import extract_crops
        crops, _ = extract_crops(Image.open(p).convert("RGB"))
        ref = enc.encode_image(torch.stack([ClipValTransform(32)(c) for c in crops]))
        got = torch.cat([d["Fake-A/test"][c] for c in CROP_NAMES])
        assert torch.allclose(got, ref, atol=1e-6)
    assert not os.path.exists(os.path.join(root, "broken.pt"))
    # resume: nothing is re-encoded, nothing is decoded
    calls = enc.calls
    assert embed_driver.Feature_Dataset(root, "Fake-A/test", 3, shuffle_filenames=False, encoder=enc, device="cpu").process()[:2] == (0, 7)
    assert enc.calls == calls
    # a second model merges into the same files; --force_reencode rewrites
    embed_driver.Feature_Dataset(root, "Fake-B/test", 4, shuffle_filenames=False, encoder=enc, device="cpu").process()
    d = torch.load(os.path.splitext(paths[0])[0] + ".pt", weights_only=True)
    assert set(d) == {"Fake-A/test", "Fake-B/test"}
    assert embed_driver.Feature_Dataset(root, "Fake-A/test", 8, force_reencode=True, shuffle_filenames=False,
                                        encoder=enc, device="cpu").process()[0] == 7
    with pytest.raises(ValueError):
        embed_driver.Feature_Dataset(root, "no-slash-name", 2)


def test_packed_store_shard_is_sealed_when_the_encode_loop_dies(tmp_path):
    """A killed / failing rank must not lose what it has already embedded: the open shard is sealed on every exit path
    and the resume check then skips exactly those images."""
    from clip_assisted_data_labeling_amd.packed_store import PackedStore
    root, sd = str(tmp_path / "data"), str(tmp_path / "store")
    os.makedirs(root)
    _make_images(root, 9)

    class Dying(FakeEncoder):
        def encode_image(self, x):
            if self.calls == 2:
                raise RuntimeError("simulated GPU failure in the third batch")
            return super().encode_image(x)

    with pytest.raises(RuntimeError, match="simulated"):
        embed_driver.Feature_Dataset(root, "Fake-A/test", 3, shuffle_filenames=False, encoder=Dying(), device="cpu",
                                     packed_store=sd).process()
    done = PackedStore(sd).keys("Fake-A/test")
    assert len(done) == 6                                    # two complete batches survived in a SEALED shard
    enc = FakeEncoder()
    n_emb, n_skip, _ = embed_driver.Feature_Dataset(root, "Fake-A/test", 3, shuffle_filenames=False, encoder=enc, device="cpu",
                                                    packed_store=sd).process()
    assert (n_emb, n_skip) == (3, 6)
    assert len(PackedStore(sd).keys("Fake-A/test")) == 9


def test_feature_assembly_order(tmp_path):
    E = 4
    d = {"M1/x": {c: torch.full((1, E), float(i)) for i, c in enumerate(CROP_NAMES)},
         "M2/y": {c: torch.full((1, E), 10.0 + i) for i, c in enumerate(CROP_NAMES)}}
    p = str(tmp_path / "u.pt")
    torch.save(d, p)
    f = predict_driver.assemble_features(p, ["M2/y", "M1/x"], ["subcrop2", "centre_crop", "missing_crop"])
    assert f.tolist() == [13.0] * E + [10.0] * E + [3.0] * E + [0.0] * E     # [model][crop in crop_names order][E]
    assert predict_driver.find_model("nope", str(tmp_path)) is None


def test_saved_regressor_is_loadable_by_the_reference_class(tmp_path):
    """train_driver.save_reference_compatible writes `torch.save(model)` (_4_train_model.py:237) naming the reference's own
    class.  Where the reference checkout is present (the authoring container), load it with THAT class in a clean
    interpreter and compare its CPU forward with the oracle; elsewhere only the pickle's class path is checked."""
    import subprocess, sys
    from clip_assisted_data_labeling_amd.nn_model import SimpleFC
    from clip_assisted_data_labeling_amd.train_driver import save_reference_compatible
    from oracle import fcreg_oracle
    torch.manual_seed(0)
    m = SimpleFC(12, [7, 5], 1, ["M/x"], crop_names=["centre_crop"], dropout_prob=0.3).eval()
    path = str(tmp_path / "m.pth")
    save_reference_compatible(m, path)
    assert type(m).__module__ == "clip_assisted_data_labeling_amd.nn_model" and "utils" not in sys.modules
    raw = open(path, "rb").read()
    assert b"utils.nn_model" in raw and b"clip_assisted_data_labeling_amd" not in raw
    if not os.path.isdir("/root/reference/utils"):
        pytest.skip("reference checkout not present: class-path check only")
    x = np.random.RandomState(0).randn(4, 12).astype(np.float32)
    np.save(str(tmp_path / "x.npy"), x)
    code = ("import sys, numpy as np, torch; sys.path.insert(0, '/root/reference');"
            f"m = torch.load({path!r}, map_location='cpu', weights_only=False);"
            "assert type(m).__module__ == 'utils.nn_model' and not m.training, type(m);"
            f"x = torch.from_numpy(np.load({str(tmp_path / 'x.npy')!r}));"
            f"np.save({str(tmp_path / 'y.npy')!r}, m(x).detach().numpy()); print(m.clip_models, m.crop_names)")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "['M/x'] ['centre_crop']" in out.stdout
    lin = m._linears()
    ref = fcreg_oracle.forward_np([l.weight.detach().numpy() for l in lin], [l.bias.detach().numpy() for l in lin], x)
    assert np.abs(np.load(str(tmp_path / "y.npy")) - ref).max() < 1e-6


def test_decode_chunks_are_cut_by_files_and_by_pixels():
    """embed_driver --gpu_decode: a decode chunk holds at most `decode_chunk` files AND at most `gpu_decode_max_pixels` decoded
    pixels (3 bytes of RGB + ~4.5 bytes of scratch each on the device) -- but always at least one file."""
    from clip_assisted_data_labeling_amd.embed_driver import chunk_cut
    mp12 = 4000 * 3000
    assert chunk_cut([mp12] * 2048, 2048, 1_000_000_000) == 83                  # 83 x 12 Mpx = 996 Mpx (2 048 of them: 74 GB of RGB)
    assert chunk_cut([512 * 512] * 5000, 2048, 1_000_000_000) == 2048            # small files: the file count cuts
    assert chunk_cut([mp12] * 3, 10, 1000) == 1                                  # one image over the budget is decoded alone
    assert chunk_cut([0, 0, mp12, mp12], 10, mp12) == 3                          # files that go to Pillow (0 pixels) cost nothing
    assert chunk_cut([], 10, 100) == 0 and chunk_cut([5], 0, 100) == 1
    sizes, pos, chunks = [100] * 7 + [1000] + [100] * 4, 0, []
    while pos < len(sizes):
        n = chunk_cut(sizes[pos:], 5, 450)
        chunks.append(n)
        pos += n
    assert chunks == [4, 3, 1, 4] and sum(chunks) == len(sizes)


def test_embed_driver_defaults_to_the_references_model():
    """/root/reference/_1_embed_with_CLIP.py:190: `--models_to_use` defaults to ViT-L-14-336/openai, the tower the shipped
    regressor checkpoint names in clip_models -- a run without the flag must write the key that checkpoint reads."""
    args = embed_driver.build_parser().parse_args(["--root_dir", "x"])
    assert args.models_to_use == ["ViT-L-14-336/openai"]
    assert args.model_path is None and not args.force_reencode
