"""CPU suite (-m "not gpu"): the oracle against the committed golden vectors, the host logic,
and that the C-ABI library loads and exports every symbol include/clipenc.h declares."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

from clip_assisted_data_labeling_amd import _lib, vit_config
from clip_assisted_data_labeling_amd.nn_model import SimpleFC, load_regressor
from clip_assisted_data_labeling_amd.preprocess import ClipValTransform, crop_boxes, extract_crops
from oracle import dedup_oracle, fcreg_oracle, vit_oracle
from tests.helpers import np_fc_weights, synthetic_crops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ oracle vs golden vectors
def test_regressor_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "regressor_shipped.npz"))
    n = int(g["n_layers"])
    Ws, bs = [g[f"W{i}"] for i in range(n)], [g[f"b{i}"] for i in range(n)]
    assert [w.shape for w in Ws] == [(264, 768), (128, 264), (64, 128), (1, 64)]
    for fwd in (fcreg_oracle.forward_c, fcreg_oracle.forward_np):
        y = fwd(Ws, bs, g["x"], float(g["negative_slope"]))
        assert np.abs(y - g["y"]).max() < 2e-6
    # values observed when the reference checkpoint was probed
    assert np.allclose(g["y"][:4, 0], [0.3588, 0.3029, 0.2688, 0.2122], atol=1e-4)


def test_regressor_oracle_4crop_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "regressor_4crop.npz"))
    Ws, bs = np_fc_weights(list(g["sizes"]), int(g["weight_seed"]))
    y = fcreg_oracle.forward_c(Ws, bs, g["x"])
    assert np.abs(y - g["y"]).max() < 2e-6


def golden_cfg(g):
    """ViTConfig of a committed encoder fixture: '<arch>/<pretrained>' through the product's own name parsing (fixtures from
    before the erf-GELU one carry no tag: 'openai')."""
    tag = str(g["pretrained"]) if "pretrained" in g.files else "openai"
    return vit_config.config_for(f"{str(g['arch'])}/{tag}")


@pytest.mark.parametrize("arch", ["ViT-tiny-test", "ViT-small-test", "ViT-B-32", "ViT-small-test-erf"])
def test_vit_oracle_matches_golden(golden_dir, arch):
    g = np.load(os.path.join(golden_dir, f"encoder_{arch}.npz"))
    cfg = golden_cfg(g)
    assert (cfg.act == vit_config.ACT_GELU_ERF) == arch.endswith("-erf")
    sd = vit_config.seeded_state_dict(cfg, int(g["weight_seed"]))
    wsum = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(wsum - float(g["weight_abs_sum"])) < 1e-6 * wsum, "seeded weights drifted from the fixture"
    crops = synthetic_crops(int(g["n_crops"]), cfg.image_size, int(g["input_seed"]))
    assert abs(float(crops.double().abs().sum()) - float(g["crops_abs_sum"])) < 1e-3
    taps = {}
    emb = vit_oracle.encode_image(sd, cfg, crops, taps)
    assert np.abs(emb.numpy() - g["emb"]).max() < 2e-6
    assert np.abs(taps["ln_pre"][:, 0].numpy() - g["ln_pre_cls"]).max() < 1e-5
    assert torch.allclose(emb.norm(dim=-1), torch.ones(emb.shape[0]), atol=1e-6)


def test_vit_oracle_at_full_size_matches_the_transformers_vectors(golden_dir):
    """The 1024-wide x 24-block x 257-token tower: the oracle on ONE crop of the fixture (a few seconds of CPU) against the
    embedding transformers.CLIPVisionModelWithProjection produced for it in the authoring container (`make_golden.py full`)."""
    g = np.load(os.path.join(golden_dir, "encoder_ViT-L-14.npz"))
    cfg = golden_cfg(g)
    assert (cfg.width, cfg.layers, cfg.tokens, cfg.mlp_dim) == (1024, 24, 257, 4096)
    sd = vit_config.seeded_state_dict(cfg, int(g["weight_seed"]))
    crops = synthetic_crops(int(g["n_crops"]), cfg.image_size, int(g["input_seed"]))
    emb = vit_oracle.encode_image(sd, cfg, crops[1:2])
    assert np.abs(emb.numpy()[0] - g["emb_transformers"][1]).max() < 1e-5
    assert np.abs(g["emb"] - g["emb_transformers"]).max() < 1e-5 and float(g["oracle_vs_transformers_max_abs"]) < 1e-5
    g336 = np.load(os.path.join(golden_dir, "encoder_ViT-L-14-336.npz"))
    assert golden_cfg(g336).tokens == 577 and np.abs(g336["emb"] - g336["emb_transformers"]).max() < 1e-5
    # ViT-H-14 (`make_golden.py vit_h`: 1280 wide x 32 blocks, 16 heads of 80, erf-GELU): the committed vectors only -- a crop of it
    # through the oracle is the GPU suite's business (tests/test_gpu_vit_h.py)
    gh = np.load(os.path.join(golden_dir, "encoder_ViT-H-14-erf.npz"))
    ch = golden_cfg(gh)
    assert (ch.width, ch.layers, ch.heads, ch.mlp_dim, ch.embed_dim, ch.act) == (1280, 32, 16, 5120, 1024, vit_config.ACT_GELU_ERF)
    assert np.abs(gh["emb"] - gh["emb_transformers"]).max() < 1e-5 and float(gh["oracle_vs_transformers_max_abs"]) < 1e-5
    # ViT-g-14 (`make_golden.py vit_g`: 1408 wide = 16 heads of 88, 40 blocks; runs zero-padded on the device, tests/test_gpu_padded_towers.py)
    gg = np.load(os.path.join(golden_dir, "encoder_ViT-g-14-erf.npz"))
    cg = golden_cfg(gg)
    assert (cg.width, cg.layers, cg.heads, cg.mlp_dim, cg.embed_dim, cg.act) == (1408, 40, 16, 6144, 1024, vit_config.ACT_GELU_ERF)
    assert np.abs(gg["emb"] - gg["emb_transformers"]).max() < 1e-5 and float(gg["oracle_vs_transformers_max_abs"]) < 1e-5
    gb = np.load(os.path.join(golden_dir, "encoder_ViT-bigG-14-erf.npz"))       # `make_golden.py vit_bigg`: 16 heads of 104, a 1280-wide embedding
    cb = golden_cfg(gb)
    assert (cb.width, cb.layers, cb.heads, cb.mlp_dim, cb.embed_dim) == (1664, 48, 16, 8192, 1280) and gb["emb_transformers"].shape == (2, 1280)
    assert np.abs(gb["emb"] - gb["emb_transformers"]).max() < 1e-5 and float(gb["oracle_vs_transformers_max_abs"]) < 1e-5
    # ... and the small tower with every kind of padding, this one THROUGH the oracle here
    gp = np.load(os.path.join(golden_dir, "encoder_ViT-pad-test.npz"))
    cp = golden_cfg(gp)
    assert cp.width // cp.heads == 48 and cp.width % 256 and cp.mlp_dim % 256
    sdp = vit_config.seeded_state_dict(cp, int(gp["weight_seed"]))
    embp = vit_oracle.encode_image(sdp, cp, synthetic_crops(int(gp["n_crops"]), cp.image_size, int(gp["input_seed"])))
    assert np.abs(embp.numpy() - gp["emb_transformers"]).max() < 1e-5


def test_dedup_oracle_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "dedup_planted.npz"))
    pairs, vals = dedup_oracle.near_duplicates(torch.from_numpy(g["emb_fp16"]), float(g["threshold"]))
    assert pairs.shape[0] == g["pairs"].shape[0] > 0
    assert (pairs.numpy() == g["pairs"]).all()
    assert np.abs(vals.float().numpy() - g["values"]).max() == 0.0
    # blocked evaluation is the same arithmetic
    p2, _ = dedup_oracle.near_duplicates(torch.from_numpy(g["emb_fp16"]), float(g["threshold"]), block=333)
    assert (p2.numpy() == g["pairs"]).all()


# ------------------------------------------------------------------ C ABI: load + symbols
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "clipenc.h")).read()
    declared = set(re.findall(r"\b((?:clipenc|fcreg|fctrain|dedup|preproc|simsearch|diversity|jpegdec)_[a-z_0-9]+)\s*\(", header))
    assert {"clipenc_create", "clipenc_encode", "fcreg_forward", "clipenc_encode_score", "dedup_find_pairs"} <= declared
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/clipenc.h but not exported"
    assert lib.clipenc_last_error() is not None
    # diagnostics live in include/clipenc_diag.h and in the separately built libclipenc_hip_diag.so only
    diag = open(os.path.join(ROOT, "include", "clipenc_diag.h")).read()
    for name in re.findall(r"\b(clipenc_[a-z_0-9]+)\s*\(", diag):
        assert name in _lib.DIAG_SIGNATURES and name not in _lib.SIGNATURES
        if "diag" not in os.path.basename(_lib.LIB_PATH):
            assert not hasattr(lib, name), f"diagnostic entry point {name} exported by the product library"


def test_library_argument_errors_without_gpu():
    lib = _lib.load()
    rc = lib.clipenc_create(None, None, 0, None)
    assert rc != 0 and b"NULL" in lib.clipenc_last_error()
    assert lib.clipenc_encode(None, None, 4, 0, None, 1, None) != 0
    assert lib.fcreg_forward(None, None, 4, 8, 1, 8, None, None, None) != 0
    assert lib.clipenc_destroy(None) == 0 and lib.fcreg_destroy(None) == 0


def test_product_has_no_cpu_path():
    from clip_assisted_data_labeling_amd.embedder import CLIP_Encoder
    with pytest.raises(_lib.ClipencError):
        CLIP_Encoder("ViT-tiny-test/seed0", device="cpu")
    m = SimpleFC(8, [4], 1, clip_models=["x"]).eval()
    with pytest.raises(_lib.ClipencError):
        m(torch.zeros(2, 8))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "clip_assisted_data_labeling_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "oracle/" not in src, f


# ------------------------------------------------------------------ host logic
def test_model_name_parsing_and_configs():
    cfg = vit_config.config_for("ViT-L-14/openai")
    assert (cfg.width, cfg.layers, cfg.heads, cfg.mlp_dim, cfg.embed_dim, cfg.tokens) == (1024, 24, 16, 4096, 768, 257)
    assert cfg.act == vit_config.ACT_QUICK_GELU
    assert vit_config.config_for("ViT-L-14/laion2b_s32b_b82k").act == vit_config.ACT_GELU_ERF
    assert cfg.macs_per_crop() == 81_012_768_768
    assert vit_config.config_for("ViT-B-32/openai").macs_per_crop() == 4_408_811_520
    with pytest.raises(ValueError):
        vit_config.config_for("RN50/openai")                     # _1_embed_with_CLIP.py:75
    with pytest.raises(FileNotFoundError):
        vit_config.load_weights("ViT-B-32/openai", None)          # never downloads
    n_params = sum(int(np.prod(s)) for _, s in vit_config.state_dict_keys(cfg))
    assert n_params == 303_966_208


def test_state_dict_round_trip_through_file(tmp_path):
    cfg = vit_config.ARCHS["ViT-tiny-test"]
    sd = vit_config.seeded_state_dict(cfg, 9)
    torch.save({"visual." + k: v for k, v in sd.items()}, tmp_path / "ViT-tiny-test-openai.pt")
    back = vit_config.load_weights("ViT-tiny-test/openai", str(tmp_path))
    assert all(torch.equal(back[k], sd[k]) for k in sd)


def test_crop_geometry_matches_committed_box_table(golden_dir):
    """tests/golden/crop_boxes.json: 30 image sizes (square, wide, tall, odd, sub-crops clipped at the border, sizes whose
    sub-crop side is 0) from the statement-by-statement restatement of utils/embedder.py:196-245 in make_golden.py."""
    import json
    table = json.load(open(os.path.join(golden_dir, "crop_boxes.json")))["table"]
    assert len(table) >= 30
    clipped = dropped = 0
    for e in table:
        W, H = e["size"]
        got = {n: (k, tuple(b)) for n, k, b in crop_boxes(W, H)}
        assert got["centre_crop"] == ("crop", tuple(e["centre_crop"])), (W, H)
        assert got["square_padded_crop"] == ("pad", tuple(e["square_padded_crop"])), (W, H)
        for n in ("subcrop1", "subcrop2"):
            if e[n] is None:
                assert n not in got, (W, H, n)
                dropped += 1
            else:
                assert got[n] == ("crop", tuple(e[n])), (W, H, n)
                l, t, r, b = e[n]
                clipped += (r - l) != (b - t)
    assert clipped >= 4 and dropped >= 2               # the table does exercise the clipped and the zero-size branches


def _module_tree(named_tensors, half=True):
    """An nn.Module whose state dict has exactly the given dotted names (nested containers), like the CLIP object inside
    OpenAI's TorchScript archives: parameters in fp16, `visual.` prefix, plus non-visual entries that must be ignored."""
    root = torch.nn.Module()
    for name, t in named_tensors.items():
        mod = root
        parts = name.split(".")
        for p_ in parts[:-1]:
            if not hasattr(mod, p_):
                mod.add_module(p_, torch.nn.Module())
            mod = getattr(mod, p_)
        mod.register_parameter(parts[-1], torch.nn.Parameter((t.half() if half else t).clone(), requires_grad=False))
    return root


def test_load_weights_from_what_open_clip_leaves_in_cache_dir(tmp_path):
    """/root/reference/utils/embedder.py:66-73 passes `cache_dir=model_path`; open_clip then leaves there either OpenAI's
    TorchScript archive named after the download URL (`ViT-L-14.pt`, `ViT-L-14-336px.pt`, ...: fp16, whole CLIP model) or
    a Hugging Face hub tree `models--*/snapshots/*/open_clip_*`.  Both must load, and `weights_only=True` alone cannot
    open the first kind."""
    cfg = vit_config.ARCHS["ViT-tiny-test"]
    sd = {k: v.half().float() for k, v in vit_config.seeded_state_dict(cfg, 9).items()}    # representable in fp16
    full = {"visual." + k: v for k, v in sd.items()}
    full["logit_scale"] = torch.tensor(4.6)                                  # non-visual entries of the real archives
    full["token_embedding.weight"] = torch.zeros(8, 4)

    # OpenAI's archive holds the model itself (keys start at `visual.`)
    tree = _module_tree(full)
    jit_dir = tmp_path / "jit"
    jit_dir.mkdir()
    vit_config._OPENAI_JIT_NAMES["ViT-tiny-test"] = "ViT-tiny-test.pt"
    try:
        scripted = torch.jit.trace(_TraceWrap(tree), torch.zeros(1), check_trace=False)
        scripted.save(str(jit_dir / "ViT-tiny-test.pt"))
        with pytest.raises(Exception):
            torch.load(str(jit_dir / "ViT-tiny-test.pt"), map_location="cpu", weights_only=True)   # what round 1 did
        back = vit_config.load_weights("ViT-tiny-test/openai", str(jit_dir))
        assert all(back[k].dtype == torch.float32 and torch.equal(back[k], sd[k]) for k in sd)
    finally:
        del vit_config._OPENAI_JIT_NAMES["ViT-tiny-test"]
    # HF hub tree, safetensors and .bin, next to ANOTHER architecture's checkpoint that must be skipped
    from safetensors.torch import save_file
    hub = tmp_path / "hub"
    snap = hub / "models--laion--CLIP-ViT-tiny-test-laion2B-s32B-b82K" / "snapshots" / "0123abcd"
    snap.mkdir(parents=True)
    save_file({k: v.contiguous() for k, v in full.items()}, str(snap / "open_clip_model.safetensors"))
    other_cfg = vit_config.ARCHS["ViT-small-test"]
    other = hub / "models--laion--CLIP-ViT-small-test-laion2B-s32B-b82K" / "snapshots" / "ffff"
    other.mkdir(parents=True)
    torch.save({"visual." + k: v for k, v in vit_config.seeded_state_dict(other_cfg, 1).items()},
               str(other / "open_clip_pytorch_model.bin"))
    back = vit_config.load_weights("ViT-tiny-test/laion2b_s32b_b82k", str(hub))
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    back2 = vit_config.load_weights("ViT-small-test/laion2b_s32b_b82k", str(hub))
    assert back2["conv1.weight"].shape[0] == other_cfg.width
    with pytest.raises(FileNotFoundError):                                    # no repository of that architecture
        vit_config.load_weights("ViT-long-test/laion2b_s32b_b82k", str(hub))
    with pytest.raises(FileNotFoundError):
        vit_config.load_weights("ViT-tiny-test/openai", str(tmp_path / "empty"))
    # open_clip resolves exactly the named tag (/root/reference/utils/embedder.py:66-73): a snapshot of the right
    # architecture but ANOTHER pretrained tag must not be loaded silently
    with pytest.raises(FileNotFoundError):
        vit_config.load_weights("ViT-tiny-test/openai", str(hub))
    with pytest.raises(FileNotFoundError):
        vit_config.load_weights("ViT-tiny-test/laion400m_e32", str(hub))
    # ... and a repository of a LONGER architecture name does not answer for the shorter one (ViT-L-14 vs ViT-L-14-336)
    assert vit_config._has_component("models--laion--clip-vit-l-14-laion2b-s32b-b82k", "vit-l-14")
    assert not vit_config._has_component("models--laion--clip-vit-l-14-336-laion2b", "vit-l-14")
    assert not vit_config._has_component("models--x--clip-convit-l-14-laion2b", "vit-l-14")
    # qualifiers that name ANOTHER architecture (their tags overlap: laion400m-e32 exists for vit-b-16 and vit-b-16-plus-240)
    assert not vit_config._has_component("models--timm--vit-b-16-plus-240-laion400m-e32", "vit-b-16")
    assert not vit_config._has_component("models--x--vit-b-32-quickgelu-metaclip-400m", "vit-b-32")
    assert not vit_config._has_component("models--timm--vit-b-16-siglip-webli", "vit-b-16")
    assert not vit_config._has_component("models--x--vit-l-14-clipa-datacomp1b", "vit-l-14")
    assert vit_config._has_component("models--laion--clip-vit-b-16-laion400m-e32", "vit-b-16")
    assert vit_config._has_component("models--laion--clip-vit-g-14-laion2b-s34b-b88k", "vit-g-14")
    assert vit_config._has_component("models--laion--clip-vit-bigg-14-laion2b-39b-b160k", "vit-bigg-14")
    assert not vit_config._has_component("models--laion--clip-vit-bigg-14-laion2b-39b-b160k", "vit-g-14")
    assert "laion2b-39b-b160k" in vit_config._TAG_ALIASES["laion2b_s39b_b160k"]      # the repository's own spelling of that tag
    assert vit_config._has_component("models--laion--clip-vit-b-32-datacomp-xl-s13b-b90k", "vit-b-32")    # '-xl' belongs to the TAG here
    # the exact hand-placed file wins over a hub snapshot of the same name pair
    exact = {k: v + 1.0 for k, v in sd.items()}
    torch.save(exact, str(hub / "ViT-tiny-test-laion2b_s32b_b82k.pt"))
    back3 = vit_config.load_weights("ViT-tiny-test/laion2b_s32b_b82k", str(hub))
    assert torch.equal(back3["conv1.weight"], exact["conv1.weight"])
    # a file that is there but holds another architecture still reports ValueError, not "not found"
    torch.save({k: v for k, v in vit_config.seeded_state_dict(other_cfg, 1).items()}, str(hub / "ViT-long-test-openai.pt"))
    with pytest.raises(ValueError):
        vit_config.load_weights("ViT-long-test/openai", str(hub))


class _TraceWrap(torch.nn.Module):
    """Gives the parameter tree a traceable forward while keeping the state-dict names unprefixed."""
    def __init__(self, tree):
        super().__init__()
        for n, c in tree.named_children():
            self.add_module(n, c)
        for n, p_ in tree.named_parameters(recurse=False):
            self.register_parameter(n, p_)

    def forward(self, x):
        return x + self.logit_scale.float()


def test_crop_geometry_matches_survey_appendix_c():
    boxes = {n: (k, b) for n, k, b in crop_boxes(224, 224)}
    assert boxes["centre_crop"] == ("crop", (0, 0, 224, 224))
    assert boxes["square_padded_crop"] == ("pad", (224, 0, 0))
    l, t, r, b = boxes["subcrop1"][1]
    assert (r - l, b - t) == (86, 86) and (l + 43, t + 43) == (56, 112)
    l, t, r, b = boxes["subcrop2"][1]
    assert (r - l, b - t) == (70, 70) and (l + 35, t + 35) == (168, 112)
    # tall image: centres (W//2, H//4) and (W//2, H//4*3); very wide: clipped, non-square
    tall = {n: b for n, _, b in crop_boxes(100, 400)}
    assert tall["centre_crop"] == (0, 150, 100, 250)
    s1 = int((100 * 400 * 0.15) ** 0.5)
    assert tall["subcrop1"] == (max(0, 50 - s1 // 2), 100 - s1 // 2, min(100, max(0, 50 - s1 // 2) + s1), 100 - s1 // 2 + s1)
    wide = {n: b for n, _, b in crop_boxes(1000, 60)}
    l, t, r, b = wide["subcrop1"]
    assert (t, b) == (0, 60) and r - l == int((1000 * 60 * 0.15) ** 0.5)      # height clipped to the image
    assert [n for n, _, _ in crop_boxes(224, 224, ["centre_crop"])] == ["centre_crop"]


def test_preprocess_transform_shapes_and_identity_case():
    from PIL import Image
    rs = np.random.RandomState(0)
    arr = rs.randint(0, 256, (224, 224, 3), dtype=np.uint8)
    img = Image.fromarray(arr)
    t = ClipValTransform(224)(img)
    assert t.shape == (3, 224, 224) and t.dtype == torch.float32
    ref = (torch.from_numpy(arr).permute(2, 0, 1).float() / 255.0 - torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(3, 1, 1)) \
        / torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(3, 1, 1)
    assert torch.equal(t, ref)                                   # 224x224 input: resize and crop are identities
    crops, names = extract_crops(Image.fromarray(rs.randint(0, 256, (300, 500, 3), dtype=np.uint8)))
    assert names == ["centre_crop", "square_padded_crop", "subcrop1", "subcrop2"]
    assert crops[0].size == (300, 300) and crops[1].size == (500, 500)
    stacked = torch.stack([ClipValTransform(224)(c) for c in crops])
    assert stacked.shape == (4, 3, 224, 224)


def test_simplefc_is_pickle_compatible_with_reference_layout(tmp_path):
    m = SimpleFC(768, [264, 128, 64], 1, clip_models=["ViT-L-14/openai"], crop_names=["centre_crop"], dropout_prob=0.5)
    assert list(m.state_dict()) == [f"layers.{i}.{p}" for i in (0, 3, 6, 9) for p in ("weight", "bias")]
    torch.save(m, tmp_path / "m.pth")
    m2 = load_regressor(str(tmp_path / "m.pth"))
    assert m2.clip_models == ["ViT-L-14/openai"] and m2.crop_names == ["centre_crop"]
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_dedup_screen_scratch_size_is_host_arithmetic():
    """include/clipenc.h: dedup_screen_ws_bytes -- counter block + e4m3 rows (n and d padded) + a margin per row + 8 bytes per
    candidate slot; no device work."""
    from clip_assisted_data_labeling_amd import _lib
    lib = _lib.load()
    b = lib.dedup_screen_ws_bytes
    assert b(100_000, 768, 0) == 256 + 100_096 * 768 + 100_096 * 4            # 391 x 256 rows; 768 = 3 x 256 columns
    assert b(100_000, 768, 1 << 20) - b(100_000, 768, 0) == 8 << 20
    assert b(1000, 100, 0) == 256 + 1024 * 512 + 1024 * 4                     # d 100 -> 128 -> at least 512 e4m3 columns
    assert b(10, 1000, 5) == 256 + 256 * 1024 + 256 * 4 + 40
    assert b(-1, 768, 10) == 0 and b(10, 0, 10) == 0


def test_bench_self_launch_builds_a_torchrun_child_and_forwards_one_line(monkeypatch, capsys):
    """bench.py --gpus N without torchrun: the parent (no GPU call made) starts N ranks as a child `torch.distributed.run` on
    127.0.0.1, forwards rank 0's JSON line only, and returns the child's status."""
    import subprocess
    import types
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=seen.get("rc", 0),
                                     stdout='noise from a rank\n{"metric": "images/sec", "value": 1.0, "n_gpus": 4}\n')
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert cmd[-5].endswith("bench.py") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "images/sec", "value": 1.0, "n_gpus": 4}' and "noise from a rank" in out.err
    seen["rc"] = 3
    assert bench.self_launch(4) == 3                        # a failing child fails the command
