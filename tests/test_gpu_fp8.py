"""fp8 (OCP e4m3) block GEMMs — BASELINE.json configs[3].  The row quantiser and the fp8 GEMM are pinned on their own
against torch's float8_e4m3fn arithmetic; the whole encoder in CLIPENC_PREC_FP8 is held to the same 1e-3 cosine bound
against the fp32 oracle as the bf16 path."""
import pytest
import torch

from clip_assisted_data_labeling_amd import _lib, vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
from oracle import vit_oracle
from tests.helpers import one_minus_cos, synthetic_crops

pytestmark = pytest.mark.gpu

COS_TOL = 1e-3
F8 = torch.float8_e4m3fn


def _stream(dev):
    return _lib.current_stream_ptr(dev)


def _quant(gpu, x, ln=False, eps=1e-5):
    lib = _lib.load()
    n, k = x.shape
    out = torch.full((n, k), 0x7f, dtype=torch.uint8, device=gpu)           # 0x7f = NaN in e4m3fn
    sc = torch.full((n,), float("nan"), dtype=torch.float32, device=gpu)
    _lib.check(lib.clipenc_op_quant_rows_fp8(x.data_ptr(), 1 if x.dtype == torch.float32 else 0, n, k, 1 if ln else 0,
                                             eps, out.data_ptr(), sc.data_ptr(), _stream(gpu)), "quant")
    torch.cuda.synchronize()
    return out, sc


def _gemm8(gpu, a8, w8, sa, sw, bias, act=-1, resid=None):
    lib = _lib.load()
    m, k = a8.shape
    n = w8.shape[0]
    out = resid.clone() if resid is not None else torch.full((m, n), float("nan"), device=gpu, dtype=torch.bfloat16)
    _lib.check(lib.clipenc_op_gemm_fp8(a8.data_ptr(), w8.data_ptr(), m, n, k, sa.data_ptr(), sw.data_ptr(), bias.data_ptr(),
                                       act, out.data_ptr() if resid is not None else None, out.data_ptr(), _stream(gpu)),
               "gemm_fp8")
    torch.cuda.synchronize()
    return out


def _deq(u8):
    return u8.view(F8).float()


# ------------------------------------------------------------------------------------------ quantiser
@pytest.mark.parametrize("n,k,dtype", [(1, 256, torch.bfloat16), (1029, 1024, torch.bfloat16), (517, 4096, torch.bfloat16),
                                       (64, 768, torch.float32), (3, 8, torch.float32)])
def test_quant_rows_matches_torch_e4m3(gpu, n, k, dtype):
    g = torch.Generator().manual_seed(n + k)
    x = (torch.randn(n, k, generator=g) * torch.logspace(-3, 3, n).view(n, 1)).to(dtype)
    x[0, :4] = torch.tensor([0.0, -0.0, 1e-30, -1e-30]).to(dtype)
    q, sc = _quant(gpu, x.to(gpu))
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    assert torch.equal(sc.cpu(), amax * (1.0 / 448.0))
    ref = (xf * (448.0 / amax).view(n, 1)).clamp(-448, 448).to(F8)           # RNE, same as v_cvt_pk_fp8_f32
    got = q.cpu().view(F8)
    assert not torch.isnan(got.float()).any()
    # the device evaluates 448/absmax with its own fp32 division; where x * inv lands within an ulp of a rounding
    # boundary the code may differ by one step from torch's -- nowhere else
    diff = (q.cpu().to(torch.int16) - ref.view(torch.uint8).to(torch.int16)).abs()
    assert diff.max().item() <= 1 and (diff != 0).float().mean().item() < 1e-3, (diff.max(), (diff != 0).sum())
    # every row uses the full range: its absmax element maps to +-448
    assert torch.equal(got.float().abs().amax(dim=1), torch.full((n,), 448.0))


def test_quant_rows_zero_row_and_layernorm(gpu):
    x = torch.zeros(4, 512, dtype=torch.bfloat16)
    x[1] = 3.0                                                              # constant row: LN -> 0
    x[2] = torch.randn(512).to(torch.bfloat16)
    x[3] = (torch.randn(512) * 50 + 20).to(torch.bfloat16)
    q, sc = _quant(gpu, x.to(gpu))
    assert torch.equal(q[0].cpu(), torch.zeros(512, dtype=torch.uint8)) and sc[0].item() == 1.0
    q, sc = _quant(gpu, x.to(gpu), ln=True, eps=1e-5)
    xf = x.float()
    xhat = (xf - xf.mean(1, keepdim=True)) * torch.rsqrt(xf.var(1, unbiased=False, keepdim=True) + 1e-5)
    deq = _deq(q.cpu()) * sc.cpu().view(4, 1)
    assert torch.isfinite(deq).all()
    # e4m3 keeps 3 mantissa bits: relative error <= 2^-4 of the element, plus the subnormal floor of the row
    err = (deq - xhat).abs()
    bound = xhat.abs() * 2.0 ** -4 + xhat.abs().amax(1, keepdim=True) * 2.0 ** -9 / 448 * 2 + 1e-6
    assert (err <= bound).all()
    assert deq[0].abs().max().item() == 0.0 and deq[1].abs().max().item() < 1e-3


# ------------------------------------------------------------------------------------------ fp8 GEMM
def _rand8(m, k, g, pow2=False):
    """random e4m3 rows and their scales; pow2: the scale rounded up to a power of two (what the block-exponent GEMMs want of the
    weight scales: the MFMA applies the exponent, include/clipenc.h)"""
    x = torch.randn(m, k, generator=g)
    amax = x.abs().amax(1, keepdim=True)
    if pow2:
        sc = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))
        return (x / sc).to(F8).view(torch.uint8), sc.flatten()
    return (x * (448.0 / amax)).to(F8).view(torch.uint8), (amax / 448.0).flatten()


@pytest.mark.parametrize("m,n,k", [(256, 256, 256), (1, 256, 256), (255, 512, 512), (1285, 768, 1024), (4099, 1024, 4096),
                                   (2056, 3072, 1024)])
def test_gemm_fp8_matches_torch(gpu, m, n, k):
    g = torch.Generator().manual_seed(m + 3 * n + 7 * k)
    a8, sa = _rand8(m, k, g)
    w8, sw = _rand8(n, k, g)
    bias = torch.randn(n, generator=g)
    scale = sa.double().view(m, 1) * sw.double().view(1, n)
    ref = (_deq(a8).double() @ _deq(w8).double().t()) * scale + bias.double()
    mag = (_deq(a8).double().abs() @ _deq(w8).double().abs().t()) * scale          # sum_k |a||w|
    out = _gemm8(gpu, a8.to(gpu), w8.to(gpu), sa.to(gpu), sw.to(gpu), bias.to(gpu)).float().cpu()
    assert torch.isfinite(out).all()
    # final bf16 rounding (half an ulp = 2^-8 relative at most) + the accumulation error of the f8f6f4 MFMA, which is
    # NOT an IEEE fp32 sum: measured <= 4e-6 * sum|a||w| on gfx950 (tools/diag_fp8.py; fp32 would give 3e-8)
    tol = ref.abs() * (2.0 ** -8 * 1.001) + 1e-5 * mag + 1e-6
    assert ((out.double() - ref).abs() <= tol).all()


def test_gemm_fp8_is_not_transposed_or_permuted(gpu):
    m, n, k = 512, 256, 512
    a = torch.zeros(m, k)
    a[torch.arange(m), (torch.arange(m) * 37) % k] = 1.0                     # one-hot rows pick single W columns
    w = ((torch.arange(n).view(n, 1) * 5 + torch.arange(k).view(1, k) * 3) % 17).float() - 8.0   # exact in e4m3
    ones_m, ones_n, zb = torch.ones(m), torch.ones(n), torch.zeros(n)
    out = _gemm8(gpu, a.to(F8).view(torch.uint8).to(gpu), w.to(F8).view(torch.uint8).to(gpu), ones_m.to(gpu), ones_n.to(gpu),
                 zb.to(gpu)).float().cpu()
    assert torch.equal(out, a @ w.t())


@pytest.mark.parametrize("act", [0, 1])
def test_gemm_fp8_activation_and_residual(gpu, act):
    m, n, k = 771, 512, 256
    g = torch.Generator().manual_seed(act)
    a8, sa = _rand8(m, k, g)
    w8, sw = _rand8(n, k, g)
    bias = torch.randn(n, generator=g)
    lin = (_deq(a8) @ _deq(w8).t()) * sa.view(m, 1) * sw.view(1, n) + bias
    acc_tol = 2e-5 * (_deq(a8).abs() @ _deq(w8).abs().t()) * sa.view(m, 1) * sw.view(1, n)   # MFMA accumulation, see above
    ref = lin * torch.sigmoid(1.702 * lin) if act == 0 else torch.nn.functional.gelu(lin)
    out = _gemm8(gpu, a8.to(gpu), w8.to(gpu), sa.to(gpu), sw.to(gpu), bias.to(gpu), act=act).float().cpu()
    assert ((out - ref).abs() <= ref.abs() * 2.0 ** -7 + 2e-3 + 1.2 * acc_tol).all()
    resid = torch.randn(m, n, generator=g).to(torch.bfloat16)
    out = _gemm8(gpu, a8.to(gpu), w8.to(gpu), sa.to(gpu), sw.to(gpu), bias.to(gpu), resid=resid.to(gpu)).float().cpu()
    ref = lin + resid.float()
    assert ((out - ref).abs() <= ref.abs() * 2.0 ** -8 * 1.001 + 1e-5 + acc_tol).all()


@pytest.mark.parametrize("m,n,k,act", [(771, 512, 256, 0), (300, 256, 1024, 1), (2056, 1024, 512, -1)])
def test_gemm_fp8_e4m3_output_with_static_column_scales(gpu, m, n, k, act):
    lib = _lib.load()
    g = torch.Generator().manual_seed(m + act)
    a8, sa = _rand8(m, k, g)
    w8, sw = _rand8(n, k, g)
    bias = torch.randn(n, generator=g)
    lin = (_deq(a8).double() @ _deq(w8).double().t()) * sa.double().view(m, 1) * sw.double().view(1, n) + bias.double()
    val = lin * torch.sigmoid(1.702 * lin) if act == 0 else (torch.nn.functional.gelu(lin) if act == 1 else lin)
    inv_s = (448.0 / (val.abs().amax(0) * torch.linspace(0.5, 40.0, n, dtype=torch.float64))).float()   # some columns saturate
    out = torch.full((m, n), 0x7f, dtype=torch.uint8, device=gpu)
    dev = [t.to(gpu) for t in (a8, w8, sa, sw, bias, inv_s)]                  # keep the device copies alive across the call
    _lib.check(lib.clipenc_op_gemm_fp8_q(dev[0].data_ptr(), dev[1].data_ptr(), m, n, k, dev[2].data_ptr(), dev[3].data_ptr(),
                                         dev[4].data_ptr(), act, dev[5].data_ptr(), out.data_ptr(), _stream(gpu)), "gemm_fp8_q")
    torch.cuda.synchronize()
    got = _deq(out.cpu()).double()
    assert not torch.isnan(got).any()
    want = (val * inv_s.double()).clamp(-448, 448)
    # e4m3 rounding: half an ulp = 2^-4 relative for normals, 2^-10 absolute below 2^-6; plus the fp32/MFMA noise upstream
    tol = want.abs() * 2.0 ** -4 * 1.02 + 2.0 ** -10 * 1.02 + 1e-4 * inv_s.double() * (k ** 0.5)
    assert ((got - want).abs() <= tol).all()
    assert (got.abs().amax(0)[: n // 100 + 1] == 448.0).all()                  # the columns scaled past 448 saturate, no NaN


def test_gemm_fp8_rejects_bad_shapes(gpu):
    lib = _lib.load()
    z = torch.zeros(1 << 20, dtype=torch.uint8, device=gpu)
    f = torch.zeros(4096, dtype=torch.float32, device=gpu)
    for (m, n, k) in [(256, 128, 256), (256, 256, 128), (256, 256, 320), (0, 256, 256)]:
        rc = lib.clipenc_op_gemm_fp8(z.data_ptr(), z.data_ptr(), m, n, k, f.data_ptr(), f.data_ptr(), f.data_ptr(), -1, None,
                                     z.data_ptr(), _stream(gpu))
        assert rc != 0, (m, n, k)
    rc = lib.clipenc_op_gemm_fp8(z.data_ptr(), z.data_ptr(), 256, 256, 256, f.data_ptr(), f.data_ptr(), f.data_ptr(), 0,
                                 z.data_ptr(), z.data_ptr(), _stream(gpu))
    assert rc != 0                                                           # activation + residual


# ------------------------------------------------------------------------------------------ block-exponent rows (fused tower)
def _block_exp_rule(x):
    """exp byte of every (row, 256-column block) of a float32 matrix: max(biased exponent of the block's max |x| - 7, 0)."""
    n, k = x.shape
    amax = x.abs().view(n, k // 256, 256).amax(2)
    ex = (amax.view(torch.int32) >> 23) & 0xff
    return (ex - 7).clamp(min=0)


def _deq_block(q8, eb):
    """e4m3 bytes [n][k] and exponent bytes [n][k/256] -> float64 values."""
    n, k = q8.shape
    return (_deq(q8).double().view(n, k // 256, 256) * torch.pow(2.0, eb.double() - 127.0).view(n, k // 256, 1)).view(n, k)


@pytest.mark.parametrize("n,k", [(1, 256), (1029, 1024), (517, 768), (64, 512)])
def test_quant_block_matches_the_exponent_rule_and_torch_e4m3(gpu, n, k):
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + k)
    x = (torch.randn(n, k, generator=g) * torch.logspace(-4, 4, n).view(n, 1)).to(torch.bfloat16)
    x[0, :4] = torch.tensor([0.0, -0.0, 1e-30, -1e-30]).to(torch.bfloat16)
    if n > 3:
        x[2] = 0.0                                                           # an all-zero row
        x[3, 256:] *= 1000.0                                                 # blocks of one row far apart in magnitude
    xd = x.to(gpu)
    q = torch.full((n, k), 0x7f, dtype=torch.uint8, device=gpu)
    eb = torch.full((n, 4), 0xff, dtype=torch.uint8, device=gpu)
    st = torch.full((n, 2), float("nan"), device=gpu)
    _lib.check(lib.clipenc_op_quant_block_fp8(xd.data_ptr(), n, k, q.data_ptr(), eb.data_ptr(), st.data_ptr(), _stream(gpu)), "quant_block")
    torch.cuda.synchronize()
    xf = x.float()
    want_e = _block_exp_rule(xf)
    got_e = eb.cpu().to(torch.int32)
    assert torch.equal(got_e[:, : k // 256], want_e) and (got_e[:, k // 256:] == 0).all()
    # scaling by a power of two is exact, v_cvt_pk_fp8_f32 rounds to nearest even like torch: bit for bit
    scaled = (xf.view(n, k // 256, 256) * torch.pow(2.0, 127.0 - want_e.float()).view(n, k // 256, 1)).view(n, k)
    assert scaled.abs().max().item() < 256.0
    assert torch.equal(q.cpu(), scaled.to(F8).view(torch.uint8))
    assert torch.allclose(st.cpu()[:, 0].double(), xf.double().sum(1), rtol=1e-5, atol=1e-3 * xf.abs().amax(1).double().clamp(min=1e-30).max().item())
    assert torch.allclose(st.cpu()[:, 1].double(), (xf.double() ** 2).sum(1), rtol=1e-4)


def test_row_norm_consts(gpu):
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    n, width, parts, ld = 1000, 1024, 16, 1024
    x = torch.randn(n, width, generator=g).double() * 3 + torch.randn(n, 1, generator=g).double() * 5
    st = torch.zeros(parts, ld, 2, dtype=torch.float64)
    xs = x.view(n, parts, width // parts)
    st[:, :n, 0] = xs.sum(2).t()
    st[:, :n, 1] = (xs ** 2).sum(2).t()
    st_dev = st.float().to(gpu)
    r = torch.full((n,), float("nan"), device=gpu)
    d = torch.full((n,), float("nan"), device=gpu)
    _lib.check(lib.clipenc_op_row_norm_consts(st_dev.data_ptr(), parts, ld, n, width, 1e-5, r.data_ptr(), d.data_ptr(), _stream(gpu)), "consts")
    torch.cuda.synchronize()
    mean, var = x.mean(1), x.var(1, unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    assert torch.allclose(r.cpu().double(), rstd, rtol=2e-4)                 # the single-pass variance in fp32
    assert torch.allclose(d.cpu().double(), -mean * rstd, rtol=2e-4, atol=1e-5)


def _rand_block_rows(m, k, g, lo=110, hi=140):
    """random block-exponent rows: e4m3 bytes with |value| < 256 and exponent bytes in [lo, hi)."""
    v = torch.randn(m, k, generator=g) * 60.0
    q8 = v.clamp(-255, 255).to(F8).view(torch.uint8)
    eb = torch.zeros(m, 4, dtype=torch.uint8)
    eb[:, : k // 256] = torch.randint(lo, hi, (m, k // 256), generator=g).to(torch.uint8)
    return q8, eb


@pytest.mark.parametrize("m,n,k,act,q_out", [(256, 256, 256, -1, False), (1, 256, 1024, -1, False), (1285, 768, 768, -1, False),
                                             (2056, 3072, 1024, -1, False), (771, 512, 1024, 0, True), (300, 256, 512, 1, True),
                                             (4099, 1024, 1024, 0, True), (515, 512, 1024, 0, False)])
def test_gemm_fp8_lnf_matches_torch(gpu, m, n, k, act, q_out):
    """The LayerNorm-folded consumer: hardware block scales from the exponent bytes, row constants and column sums in the
    epilogue; several tiles per workgroup in the larger cases (the exponent rows of the NEXT tile travel through the image)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(m + 3 * n + 7 * k)
    a8, eb = _rand_block_rows(m, k, g)
    w8, sw = _rand8(n, k, g, pow2=True)
    bias, cs = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3
    r, d = torch.rand(m, generator=g) * 2 + 0.05, torch.randn(m, generator=g) * 2
    a = _deq_block(a8, eb[:, : k // 256].to(torch.int32))
    w = _deq(w8).double() * sw.double().view(n, 1)
    lin = r.double().view(m, 1) * (a @ w.t()) + d.double().view(m, 1) * cs.double().view(1, n) + bias.double()
    mag = r.double().view(m, 1) * (a.abs() @ w.abs().t())
    val = lin * torch.sigmoid(1.702 * lin) if act == 0 else (torch.nn.functional.gelu(lin) if act == 1 else lin)
    dev = [t.to(gpu) for t in (a8, eb, w8, r, d, sw, cs, bias)]
    if q_out:
        inv_s = (448.0 / (val.abs().amax(0).clamp(min=1e-6) * torch.linspace(0.5, 40.0, n, dtype=torch.float64))).float()
        inv_dev = inv_s.to(gpu)
        out = torch.full((m, n), 0x7f, dtype=torch.uint8, device=gpu)
    else:
        inv_dev = None
        out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device=gpu)
    _lib.check(lib.clipenc_op_gemm_fp8_lnf(dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(), m, n, k, dev[3].data_ptr(),
                                           dev[4].data_ptr(), dev[5].data_ptr(), dev[6].data_ptr(), dev[7].data_ptr(), act,
                                           inv_dev.data_ptr() if q_out else None, out.data_ptr(), _stream(gpu)), "gemm_fp8_lnf")
    torch.cuda.synchronize()
    acc_tol = 1e-5 * mag + 1e-6                                              # MFMA accumulation (test_gemm_fp8_matches_torch)
    if q_out:
        got = _deq(out.cpu()).double()
        assert not torch.isnan(got).any()
        want = (val * inv_s.double()).clamp(-448, 448)
        tol = want.abs() * 2.0 ** -4 * 1.02 + 2.0 ** -10 * 1.02 + 1.2 * acc_tol * inv_s.double() + 1e-4 * inv_s.double()
        assert ((got - want).abs() <= tol).all()
    else:
        got = out.float().cpu().double()
        assert torch.isfinite(got).all()
        act_slack = 0.0 if act == -1 else 2e-3
        assert ((got - val).abs() <= val.abs() * (2.0 ** -8 * 1.001 if act == -1 else 2.0 ** -7) + 1.2 * acc_tol + act_slack).all()


@pytest.mark.parametrize("m,n,k", [(256, 256, 256), (1, 256, 256), (1285, 768, 1024), (4099, 1024, 4096), (515, 1024, 1024)])
def test_gemm_fp8_resid_q_writes_rows_copy_exponents_and_statistics(gpu, m, n, k):
    """The producing GEMM of the fused tower: the bf16 rows are those of the plain residual GEMM bit for bit; the e4m3 copy
    decodes to them within the e4m3 step; exponents follow the block rule; statistics are those of the stored rows."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(m + n + k)
    a8, sa = _rand8(m, k, g)
    a_scaled = (_deq(a8) * sa.view(m, 1) * torch.logspace(-2, 2, m).view(m, 1))         # rows of very different magnitude ...
    a8 = a_scaled.clamp(-448, 448).to(F8).view(torch.uint8)                             # ... in one static-scale operand
    w8, sw = _rand8(n, k, g, pow2=True)
    bias = torch.randn(n, generator=g)
    resid = (torch.randn(m, n, generator=g) * torch.logspace(-1, 1, m).view(m, 1)).to(torch.bfloat16)
    ones = torch.ones(m)
    plain = _gemm8(gpu, a8.to(gpu), w8.to(gpu), ones.to(gpu), sw.to(gpu), bias.to(gpu), resid=resid.to(gpu)).cpu()
    x = resid.to(gpu).clone()
    q = torch.full((m, n), 0x7f, dtype=torch.uint8, device=gpu)
    eb = torch.full((m, 4), 0xff, dtype=torch.uint8, device=gpu)
    ld = (m + 255) // 256 * 256
    st = torch.full((n // 256, ld, 2), float("nan"), device=gpu)
    dev = [t.to(gpu) for t in (a8, w8, sw, bias)]
    _lib.check(lib.clipenc_op_gemm_fp8_resid_q(dev[0].data_ptr(), dev[1].data_ptr(), m, n, k, dev[2].data_ptr(), dev[3].data_ptr(),
                                               x.data_ptr(), q.data_ptr(), eb.data_ptr(), st.data_ptr(), ld, _stream(gpu)), "resid_q")
    torch.cuda.synchronize()
    assert torch.equal(x.cpu().view(torch.int16), plain.view(torch.int16))
    xf = x.float().cpu()
    got_e = eb.cpu().to(torch.int32)[:, : n // 256]
    rule = _block_exp_rule(xf)                                               # from the ROUNDED rows: the kernel sees the values
    assert (got_e - rule).abs().max().item() <= 1                            # before rounding, a block maximum next to a power
    assert ((got_e != rule).float().mean().item()) < 0.02                    # of two may land one exponent lower
    deq = _deq_block(q.cpu(), got_e)
    assert _deq(q.cpu()).abs().max().item() <= 256.0
    blockmax = xf.abs().view(m, n // 256, 256).amax(2, keepdim=True).expand(m, n // 256, 256).reshape(m, n).double()
    # e4m3 of the unrounded value: 2^-4 relative (+ the bf16 step between the two), subnormal floor 2^-10 * 2^e <= blockmax 2^-17
    assert ((deq - xf.double()).abs() <= xf.double().abs() * (2.0 ** -4 + 2.0 ** -7) + blockmax * 2.0 ** -16 + 1e-30).all()
    parts = xf.double().view(m, n // 256, 256)
    assert torch.allclose(st.cpu()[:, :m, 0].t().double(), parts.sum(2), rtol=1e-5, atol=1e-4 * xf.abs().max().item())
    assert torch.allclose(st.cpu()[:, :m, 1].t().double(), (parts ** 2).sum(2), rtol=1e-5, atol=1e-30)


def test_fused_block_ops_chain_like_the_tower(gpu):
    """quantise -> constants -> LayerNorm-folded GEMM reproduces LayerNorm(x) . W^T + b of the fp32 path within the e4m3 noise
    (rows with a mean several times their spread included: the mean term is folded, not subtracted before quantising)."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    m, n, k = 700, 512, 1024
    x = (torch.randn(m, k, generator=g) * torch.logspace(-1, 2, m).view(m, 1) + torch.linspace(-4, 4, m).view(m, 1)
         * torch.logspace(-1, 2, m).view(m, 1)).to(torch.bfloat16)
    gamma, beta = torch.rand(k, generator=g) + 0.5, torch.randn(k, generator=g) * 0.1
    W = torch.randn(n, k, generator=g) * 0.05
    b = torch.randn(n, generator=g)
    Wf = W * gamma.view(1, k)                                                 # gamma folded into the rows, beta into the bias
    bf = b + W @ beta
    amax = Wf.abs().amax(1, keepdim=True)
    sc2 = torch.exp2(torch.ceil(torch.log2(amax / 448.0)))                     # power-of-two weight scales (the MFMA applies the exponent)
    w8 = (Wf / sc2).to(F8).view(torch.uint8)
    sw = sc2.flatten()
    cs = (_deq(w8) * sw.view(n, 1)).sum(1)
    xd = x.to(gpu)
    q = torch.empty((m, k), dtype=torch.uint8, device=gpu)
    eb = torch.empty((m, 4), dtype=torch.uint8, device=gpu)
    st = torch.empty((m, 2), device=gpu)
    r = torch.empty(m, device=gpu)
    d = torch.empty(m, device=gpu)
    out = torch.empty((m, n), dtype=torch.bfloat16, device=gpu)
    dev = [t.to(gpu) for t in (w8, sw, cs, bf)]
    _lib.check(lib.clipenc_op_quant_block_fp8(xd.data_ptr(), m, k, q.data_ptr(), eb.data_ptr(), st.data_ptr(), _stream(gpu)), "quant_block")
    _lib.check(lib.clipenc_op_row_norm_consts(st.data_ptr(), 1, m, m, k, 1e-5, r.data_ptr(), d.data_ptr(), _stream(gpu)), "consts")
    _lib.check(lib.clipenc_op_gemm_fp8_lnf(q.data_ptr(), eb.data_ptr(), dev[0].data_ptr(), m, n, k, r.data_ptr(), d.data_ptr(),
                                           dev[1].data_ptr(), dev[2].data_ptr(), dev[3].data_ptr(), -1, None, out.data_ptr(),
                                           _stream(gpu)), "lnf")
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.double(), (k,), gamma.double(), beta.double(), 1e-5) @ W.double().t() + b.double()
    err = (out.float().cpu().double() - ref)
    # e4m3 on both operands: ~3.7 % relative noise per product (two roundings of up to 2^-4), which a sum of k random-sign
    # products keeps; a row whose mean is t sigma carries sqrt(1 + t^2) times that, because the mean is quantised with the
    # row and only removed in the epilogue (t runs from -4 to 4 over the rows here)
    t = torch.linspace(-4, 4, m).abs()
    rel = err.pow(2).mean(1).sqrt() / ref.pow(2).mean(1).sqrt()
    assert rel[t < 1].mean().item() < 0.05, rel[t < 1].mean()
    assert (rel <= 0.06 * torch.sqrt(1 + t * t).double() + 0.01).all(), (rel / torch.sqrt(1 + t * t).double()).max()
    assert one_minus_cos(out.float().cpu(), ref.float())[t < 1].max().item() < 3e-3


# ------------------------------------------------------------------------------------------ attention with e4m3 output
@pytest.mark.parametrize("n_crops,n_tok,heads", [(2, 50, 12), (3, 257, 16), (8, 257, 16), (90, 257, 16), (20, 250, 4), (2, 577, 16)])
def test_attention_e4m3_output_with_static_channel_scales(gpu, n_crops, n_tok, heads):
    lib = _lib.load()
    width = heads * 64
    g = torch.Generator().manual_seed(n_tok + heads)
    qkv = (torch.randn(n_crops * n_tok, 3 * width, generator=g) * 1.5).to(torch.bfloat16)
    q, k, v = qkv.float().view(n_crops, n_tok, 3, heads, 64).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).permute(0, 2, 1, 3).reshape(n_crops * n_tok, width)
    inv_s = (448.0 / (ref.abs().amax(0) * torch.linspace(0.8, 30.0, width))).float()      # the first channels saturate
    out = torch.full((n_crops * n_tok, width), 0x7f, dtype=torch.uint8, device=gpu)
    qkv_dev, inv_dev = qkv.to(gpu), inv_s.to(gpu)
    _lib.check(lib.clipenc_op_attention_q(qkv_dev.data_ptr(), out.data_ptr(), n_crops, n_tok, width, heads, inv_dev.data_ptr(),
                                          _stream(gpu)), "attention_q")
    torch.cuda.synchronize()
    got = _deq(out.cpu())
    assert not torch.isnan(got).any()
    want = (ref * inv_s).clamp(-448, 448)
    # e4m3 rounding (2^-4 relative / 2^-10 absolute) on top of the bf16-P attention error (0.03 absolute before scaling)
    tol = want.abs() * 2.0 ** -4 * 1.02 + 2.0 ** -10 * 1.02 + 0.03 * inv_s
    assert ((got - want).abs() <= tol).all()
    assert one_minus_cos(got / inv_s, ref.clamp(-448 / inv_s, 448 / inv_s)).max().item() < 2e-3


# ------------------------------------------------------------------------------------------ encoder in fp8
@pytest.mark.parametrize("arch", ["ViT-tiny-test", "ViT-small-test", "ViT-B-32", "ViT-small-test/laion2b_s32b_b82k"])
def test_encoder_fp8_within_tolerance_of_oracle(gpu, arch):
    cfg = vit_config.config_for(arch)                      # the laion tag: erf-GELU in FC1's quantising epilogue
    assert (cfg.act == vit_config.ACT_GELU_ERF) == ("laion" in arch)
    sd = vit_config.seeded_state_dict(cfg, 3)
    crops = synthetic_crops(8, cfg.image_size, 12)
    ref = vit_oracle.encode_image(sd, cfg, crops)
    vit = HipViT(cfg, sd, gpu)
    bf = vit.encode(crops.to(gpu)).cpu()
    vit.set_precision("fp8")
    f8 = vit.encode(crops.to(gpu))
    assert torch.equal(f8, vit.encode(crops.to(gpu)))                        # deterministic
    f8 = f8.cpu()
    assert not torch.equal(f8, bf)                                           # the fp8 kernels did run
    omc = one_minus_cos(f8, ref)
    assert omc.max().item() < COS_TOL, omc
    vit.set_precision("bf16")
    assert torch.equal(vit.encode(crops.to(gpu)).cpu(), bf)                  # and switching back restores bf16 exactly
    vit.close()


_UNFUSED_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[2])
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
from tests.helpers import synthetic_crops
dev = torch.device("cuda", 0)
out = {}
for arch in ("ViT-small-test", "ViT-B-32"):
    cfg = vit_config.ARCHS[arch]
    vit = HipViT(cfg, vit_config.seeded_state_dict(cfg, 3), dev, precision="fp8")
    out[arch] = vit.encode(synthetic_crops(8, cfg.image_size, 12).to(dev)).cpu()
    vit.close()
torch.save(out, sys.argv[1])
"""


def test_encoder_fp8_fused_and_unfused_towers_agree(gpu, tmp_path):
    """Widths over 1024 still run the separate LayerNorm-quantise pass; the diagnostic library runs it at every width
    (CLIPENC_FP8_UNFUSED=1).  Both towers against the fp32 oracle, and against each other."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "clip_assisted_data_labeling_amd", "libclipenc_hip_diag.so")
    assert os.path.exists(diag), f"{diag} missing: __graft_entry__.build() makes it"
    env = {k: v for k, v in os.environ.items() if k not in ("CLIPENC_FP8_UNFUSED", "CLIPENC_LIB_PATH")}
    env.update({"CLIPENC_LIB_PATH": diag, "CLIPENC_FP8_UNFUSED": "1"})
    path = str(tmp_path / "unfused.pt")
    subprocess.run([sys.executable, "-c", _UNFUSED_CHILD, path, root], env=env, check=True, timeout=600)
    unfused = torch.load(path)
    for arch in ("ViT-small-test", "ViT-B-32"):
        cfg = vit_config.ARCHS[arch]
        sd = vit_config.seeded_state_dict(cfg, 3)
        crops = synthetic_crops(8, cfg.image_size, 12)
        ref = vit_oracle.encode_image(sd, cfg, crops)
        vit = HipViT(cfg, sd, gpu, precision="fp8")
        fused = vit.encode(crops.to(gpu)).cpu()
        vit.close()
        assert not torch.equal(fused, unfused[arch])                         # two different towers did run
        print(arch, "1-cos fused", one_minus_cos(fused, ref).max().item(), "unfused", one_minus_cos(unfused[arch], ref).max().item())
        assert one_minus_cos(fused, ref).max().item() < COS_TOL
        assert one_minus_cos(unfused[arch], ref).max().item() < COS_TOL
        assert one_minus_cos(fused, unfused[arch]).max().item() < 2 * COS_TOL      # two independent roundings of the same tower


def test_encoder_fp8_chunk_and_row_invariance(gpu):
    cfg = vit_config.ARCHS["ViT-small-test"]
    vit = HipViT(cfg, vit_config.seeded_state_dict(cfg, 4), gpu, precision="fp8")
    crops = synthetic_crops(37, cfg.image_size, 5).to(gpu)
    crops[5] = crops[20]
    e = vit.encode(crops)
    assert torch.equal(e[5], e[20])                  # per-token scales: a crop's embedding does not depend on its batch
    vit.set_chunk(16)
    assert one_minus_cos(e.cpu(), vit.encode(crops).cpu()).max().item() < 1e-6
    vit.close()


def test_encoder_fp8_with_outlier_channels(gpu):
    """Same planted outliers as the bf16 test (test_gpu_parity.py): a few residual channels 60x larger than the rest.
    The per-(row, 256 columns) exponent puts those near the top of the e4m3 range and the floating-point format keeps 3 mantissa
    bits for the small channels down to 2^-14 of the block maximum; the rows also carry a mean of 1.5 (ln_pre.bias), which the
    fused tower folds into the consuming GEMM's epilogue instead of subtracting it before quantising."""
    cfg = vit_config.ARCHS["ViT-small-test"]
    sd = vit_config.seeded_state_dict(cfg, 8)
    g = torch.Generator().manual_seed(0)
    hot = torch.randperm(cfg.width, generator=g)[:3]
    sd["ln_pre.weight"][hot] *= 60.0
    sd["ln_pre.bias"] += 1.5
    sd["ln_pre.bias"][hot] += 40.0
    for l in range(cfg.layers):
        sd[f"transformer.resblocks.{l}.attn.out_proj.bias"][hot[0]] += 25.0
        sd[f"transformer.resblocks.{l}.mlp.c_proj.bias"][hot[1]] -= 25.0
    crops = synthetic_crops(6, cfg.image_size, 31)
    ref = vit_oracle.encode_image(sd, cfg, crops)
    vit = HipViT(cfg, sd, gpu, precision="fp8")
    omc = one_minus_cos(vit.encode(crops.to(gpu)).cpu(), ref)
    print("fp8 outlier 1-cos:", omc)
    assert omc.max().item() < COS_TOL, omc
    vit.close()


def test_gemm_fp8_e4m3_output_of_huge_and_non_finite_values(gpu):
    """What the static-scale e4m3 epilogue (EPI 2) does beyond the representable range, pinned byte for byte: a FINITE value of any
    size saturates to +-448 (0x7e / 0xfe: MODE.FP16_OVFL, gemm_fp8.hip), a non-finite one (+-inf from an overflowing row scale, or
    0 x inf) becomes the e4m3 NaN code and poisons what consumes it -- it is not clamped (the hardware conversion's rule, established
    by tools/probes/fp8_ovfl_probe.hip; accepted: a non-finite pre-activation means the tower has already diverged)."""
    lib = _lib.load()
    m, n, k = 256, 256, 256
    a8 = torch.full((m, k), 0x38, dtype=torch.uint8)                          # 1.0
    w8 = torch.where((torch.arange(n) % 2 == 0).view(n, 1).expand(n, k), torch.tensor(0x38, dtype=torch.uint8),
                     torch.tensor(0xb8, dtype=torch.uint8)).contiguous()      # +1.0 / -1.0 rows: acc = +-256
    sa = torch.ones(m)
    sa[1], sa[2], sa[3] = 1e30, float("inf"), 3.0e38                          # 256 * 3e38 overflows fp32 in the epilogue's multiply
    sw, bias, inv_s = torch.ones(n), torch.zeros(n), torch.ones(n)
    for act in (-1, 0):
        out = torch.full((m, n), 0x55, dtype=torch.uint8, device=gpu)
        dev = [t.to(gpu) for t in (a8, w8, sa, sw, bias, inv_s)]
        _lib.check(lib.clipenc_op_gemm_fp8_q(dev[0].data_ptr(), dev[1].data_ptr(), m, n, k, dev[2].data_ptr(), dev[3].data_ptr(),
                                             dev[4].data_ptr(), act, dev[5].data_ptr(), out.data_ptr(), _stream(gpu)), "gemm_fp8_q")
        torch.cuda.synchronize()
        o = out.cpu()
        pos = torch.arange(n) % 2 == 0
        if act == -1:
            assert (o[0][pos] == 0x78).all() and (o[0][~pos] == 0xf8).all()   # +-256 exactly
            assert (o[1][pos] == 0x7e).all() and (o[1][~pos] == 0xfe).all()   # 2.56e32: saturated
            assert ((o[2] & 0x7f) == 0x7f).all() and ((o[3] & 0x7f) == 0x7f).all()   # +-inf: NaN code, not +-448
        else:                                                                 # QuickGELU: u sigmoid(1.702 u) -> u for u >> 0, -> -0 for u << 0
            assert (o[0][pos] == 0x78).all() and ((o[0][~pos] & 0x7f) == 0).all()
            assert (o[1][pos] == 0x7e).all() and ((o[1][~pos] & 0x7f) == 0).all()
            assert ((o[2][pos] & 0x7f) == 0x7f).all() and ((o[3][pos] & 0x7f) == 0x7f).all()
