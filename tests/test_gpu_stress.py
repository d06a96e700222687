"""Race screens (-m gpu): the persistent GEMM ring, the in-place residual epilogue and the streaming attention's
LDS flag protocol are hand-synchronised; a rare ordering bug shows up as run-to-run differences or as rare wrong
tiles.  Every case is run many times back to back (warm L2/LDS, uneven shapes) and must be bitwise stable and
equal to an independent reference."""
import ctypes

import pytest
import torch

from clip_assisted_data_labeling_amd import _lib, vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
from tests.helpers import one_minus_cos

pytestmark = pytest.mark.gpu


def _gemm_bf16(lib, a, w, out, st):
    m, k = a.shape
    _lib.check(lib.clipenc_op_gemm_nt(a.data_ptr(), w.data_ptr(), m, w.shape[0], k, 0, 1, None, out.data_ptr(), st), "gemm")


@pytest.mark.parametrize("m,n,k", [(70001, 768, 384), (131329, 256, 1024), (9999, 2304, 128), (300000, 1024, 256)])
def test_persistent_gemm_is_stable_over_repeats(gpu, m, n, k):
    lib = _lib.load()
    st = _lib.current_stream_ptr(gpu)
    g = torch.Generator(device=gpu).manual_seed(m)
    a = torch.randn(m, k, device=gpu, generator=g).to(torch.bfloat16)
    w = torch.randn(n, k, device=gpu, generator=g).to(torch.bfloat16)
    ref = (a[-3000:].float() @ w.float().t())
    first = None
    for it in range(12):
        out = torch.full((m, n), float("nan"), device=gpu, dtype=torch.bfloat16)
        _gemm_bf16(lib, a, w, out, st)
        if first is None:
            first = out
            assert torch.isfinite(out).all()
            assert (out[-3000:].float() - ref).abs().max().item() <= 0.01 * ref.abs().max().item() + 0.05
            # every tile of a strided sample of rows against fp32
            rows = torch.arange(0, m, 997, device=gpu)
            r2 = a[rows].float() @ w.float().t()
            assert (out[rows].float() - r2).abs().max().item() <= 0.01 * r2.abs().max().item() + 0.05
        else:
            assert torch.equal(out, first), f"run {it} differs from run 0"


def test_streaming_attention_is_stable_over_repeats(gpu):
    lib = _lib.load()
    st = _lib.current_stream_ptr(gpu)
    n_crops, n_tok, heads = 300, 257, 16                         # 4800 tasks, ~19 per workgroup: buffers recycle often
    width = heads * 64
    g = torch.Generator(device=gpu).manual_seed(1)
    qkv = (torch.randn(n_crops * n_tok, 3 * width, device=gpu, generator=g) * 1.5).to(torch.bfloat16)
    first = None
    for it in range(10):
        out = torch.full((n_crops * n_tok, width), float("nan"), dtype=torch.bfloat16, device=gpu)
        _lib.check(lib.clipenc_op_attention(qkv.data_ptr(), out.data_ptr(), n_crops, n_tok, width, heads, st), "attention")
        if first is None:
            first = out
            assert torch.isfinite(out).all()
            for crop in (0, 137, 299):                            # first / middle / last task ranges vs fp32
                blk = qkv[crop * n_tok:(crop + 1) * n_tok].float().view(n_tok, 3, heads, 64).permute(1, 2, 0, 3)
                ref = (torch.softmax(blk[0] @ blk[1].transpose(-1, -2) * 0.125, -1) @ blk[2]).permute(1, 0, 2).reshape(n_tok, width)
                got = out[crop * n_tok:(crop + 1) * n_tok].float()
                assert (got - ref).abs().max().item() < 0.03
        else:
            assert torch.equal(out, first), f"run {it} differs from run 0"


@pytest.mark.parametrize("n_crops,n_tok,heads,planted", [(150, 577, 16, False), (90, 400, 8, True), (300, 321, 4, False)])
def test_long_streaming_attention_is_stable_over_repeats(gpu, n_crops, n_tok, heads, planted):
    """attn_long_stream_kernel's tile hand-over (the next task's key tiles replace the current task's in place: `landed` / `done[]` words in
    LDS, a loader wave, no barrier): 2 400 / 720 / 1 200 tasks = 3-10 per workgroup, run eight times -- every run must give the first
    run's bits, and sampled crops the fp32 softmax.  `planted`: every seventh crop carries a key hundreds of nats above its first tile, so
    that flagged blocks take the exact second sweep in most workgroups."""
    lib = _lib.load()
    st = _lib.current_stream_ptr(gpu)
    width = heads * 64
    g = torch.Generator(device=gpu).manual_seed(n_tok)
    qkv = (torch.randn(n_crops * n_tok, 3 * width, device=gpu, generator=g) * 1.5).to(torch.bfloat16)
    if planted:
        for crop in range(0, n_crops, 7):
            qkv[crop * n_tok + 211, width:2 * width] = 24.0
    first = None
    for it in range(8):
        out = torch.full((n_crops * n_tok, width), float("nan"), dtype=torch.bfloat16, device=gpu)
        _lib.check(lib.clipenc_op_attention(qkv.data_ptr(), out.data_ptr(), n_crops, n_tok, width, heads, st), "attention")
        if first is None:
            first = out
            assert torch.isfinite(out).all()
            for crop in (0, 7, n_crops // 2, n_crops - 1):
                blk = qkv[crop * n_tok:(crop + 1) * n_tok].double().view(n_tok, 3, heads, 64).permute(1, 2, 0, 3)
                ref = (torch.softmax(blk[0] @ blk[1].transpose(-1, -2) * 0.125, -1) @ blk[2]).permute(1, 0, 2).reshape(n_tok, width).float()
                got = out[crop * n_tok:(crop + 1) * n_tok].float()
                assert (got - ref).abs().max().item() < 0.04, crop
        else:
            assert torch.equal(out, first), f"run {it} differs from run 0"


def test_encoder_repeatability_under_load(gpu):
    cfg = vit_config.ARCHS["ViT-small-test"]
    vit = HipViT(cfg, vit_config.seeded_state_dict(cfg, 2), gpu)
    g = torch.Generator(device=gpu).manual_seed(4)
    crops = torch.randn(700, 3, cfg.image_size, cfg.image_size, device=gpu, generator=g)   # 35 000 token rows, 137 M-tiles
    first = vit.encode(crops)
    for _ in range(8):
        assert torch.equal(vit.encode(crops), first)
    sub = vit.encode(crops[123:130])
    assert one_minus_cos(sub.cpu(), first[123:130].cpu()).max().item() < 1e-6
    vit.close()


@pytest.mark.parametrize("m,k2", [(131329, 1024), (70001, 4096)])
def test_fused_fp8_chain_is_stable_over_repeats_and_row_prefixes(gpu, m, k2):
    """The fused fp8 tower's hand-synchronised pieces in one chain: the residual GEMM that quantises its own rows (cross-wave
    exchange of the row maxima through LDS + one extra workgroup barrier per tile), the row constants, and the LayerNorm-folded
    consumer (exponent dwords of the NEXT tile fetched by LDS-DMA under the current one and carried through registers across
    the epilogue).  Many tiles per workgroup, a ragged last M-tile, twelve back-to-back runs bitwise equal; and every row's
    result must not depend on how many rows follow it (the first 5 000 rows of the full problem = the 5 000-row problem)."""
    lib = _lib.load()
    st = _lib.current_stream_ptr(gpu)
    F8 = torch.float8_e4m3fn
    n = 1024
    g = torch.Generator(device=gpu).manual_seed(m)
    a8 = (torch.randn(m, k2, device=gpu, generator=g) * 40).clamp(-448, 448).to(F8).view(torch.uint8)
    w8 = (torch.randn(n, k2, device=gpu, generator=g) * 40).clamp(-448, 448).to(F8).view(torch.uint8)
    w2 = (torch.randn(768, n, device=gpu, generator=g) * 40).clamp(-448, 448).to(F8).view(torch.uint8)
    sw = torch.exp2(torch.randint(-16, -12, (n,), device=gpu, generator=g).float())        # powers of two: the fused ops' weight scales
    sw2 = torch.exp2(torch.randint(-13, -10, (768,), device=gpu, generator=g).float())
    bias = torch.randn(n, device=gpu, generator=g)
    cs2, b2 = torch.randn(768, device=gpu, generator=g), torch.randn(768, device=gpu, generator=g)
    x0 = (torch.randn(m, n, device=gpu, generator=g) * torch.logspace(-2, 2, m, device=gpu).view(m, 1)).to(torch.bfloat16)

    def run(rows):
        ld = (rows + 255) // 256 * 256
        x = x0[:rows].clone()
        q = torch.full((rows, n), 0x7f, dtype=torch.uint8, device=gpu)
        eb = torch.full((rows, 4), 0xff, dtype=torch.uint8, device=gpu)
        stt = torch.full((n // 256, ld, 2), float("nan"), device=gpu)
        r = torch.empty(rows, device=gpu)
        d = torch.empty(rows, device=gpu)
        out = torch.full((rows, 768), float("nan"), dtype=torch.bfloat16, device=gpu)
        _lib.check(lib.clipenc_op_gemm_fp8_resid_q(a8.data_ptr(), w8.data_ptr(), rows, n, k2, sw.data_ptr(), bias.data_ptr(), x.data_ptr(),
                                                   q.data_ptr(), eb.data_ptr(), stt.data_ptr(), ld, st), "resid_q")
        _lib.check(lib.clipenc_op_row_norm_consts(stt.data_ptr(), n // 256, ld, rows, n, 1e-5, r.data_ptr(), d.data_ptr(), st), "consts")
        _lib.check(lib.clipenc_op_gemm_fp8_lnf(q.data_ptr(), eb.data_ptr(), w2.data_ptr(), rows, 768, n, r.data_ptr(), d.data_ptr(),
                                               sw2.data_ptr(), cs2.data_ptr(), b2.data_ptr(), -1, None, out.data_ptr(), st), "lnf")
        return x, q, eb, out

    first = run(m)
    assert all(torch.isfinite(t.float()).all() for t in (first[0], first[3]))
    assert not torch.isnan(first[1].view(F8).float()).any() and (first[2] != 0xff).all()
    for it in range(11):
        again = run(m)
        for t0, t1, name in zip(first, again, ("x", "x8", "exponents", "consumer output")):
            assert torch.equal(t0, t1), f"run {it + 1}: {name} differs from run 0"
    head = run(5000)
    for t0, t1, name in zip(first, head, ("x", "x8", "exponents", "consumer output")):
        assert torch.equal(t0[:5000], t1), f"{name}: the first 5000 rows depend on the rows behind them"
