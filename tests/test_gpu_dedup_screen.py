"""The screened near-duplicate search (include/clipenc.h: dedup_find_pairs_screened -- an e4m3 MFMA screen, then the exact float16
value of the candidates only) against the exact search (dedup_find_pairs, which the other tests hold to the oracle and to the
reference's golden pairs): the SAME pairs with the SAME value bits, whatever the data, and the exact search by itself when the
candidates do not fit.  /root/reference/_2_remove_duplicates.py:63-80."""
import numpy as np
import pytest
import torch

from clip_assisted_data_labeling_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _search(gpu, e16, thr, fp16_compare=1, capacity=1 << 16, screened=True, cand_cap=1 << 20):
    lib = _lib.load()
    n, d = e16.shape
    n_pad, d_pad = (n + 255) // 256 * 256, (d + 127) // 128 * 128
    x = e16.to(gpu).contiguous()
    ws = torch.empty(n_pad * d_pad, dtype=torch.float16, device=gpu)
    pairs = torch.full((capacity, 2), -1, dtype=torch.int64, device=gpu)
    vals = torch.zeros(capacity, dtype=torch.float32, device=gpu)
    count = torch.full((1,), 123, dtype=torch.int64, device=gpu)
    st = _lib.current_stream_ptr(gpu)
    cands = None
    if screened:
        nbytes = int(lib.dedup_screen_ws_bytes(n, d, cand_cap))
        sws = torch.empty(nbytes + 256, dtype=torch.uint8, device=gpu)
        ptr = (sws.data_ptr() + 255) // 256 * 256
        _lib.check(lib.dedup_find_pairs_screened(x.data_ptr(), n, d, thr, fp16_compare, ws.data_ptr(), ptr, nbytes, cand_cap, pairs.data_ptr(),
                                                 vals.data_ptr(), capacity, count.data_ptr(), st), "screened")
        torch.cuda.synchronize()
        off = ptr - sws.data_ptr()
        cands = int(sws[off:off + 8].view(torch.int64).item())                 # the candidate counter (first word of the scratch)
    else:
        _lib.check(lib.dedup_find_pairs(x.data_ptr(), n, d, thr, fp16_compare, ws.data_ptr(), pairs.data_ptr(), vals.data_ptr(), capacity,
                                        count.data_ptr(), st), "exact")
        torch.cuda.synchronize()
    c = int(count.item())
    p = pairs[:min(c, capacity)].cpu().numpy()
    v = vals[:min(c, capacity)].cpu().numpy()
    order = np.lexsort((p[:, 1], p[:, 0]))
    return c, p[order], v[order], cands


def _same(gpu, e16, thr, **kw):
    c0, p0, v0, _ = _search(gpu, e16, thr, screened=False, **{k: v for k, v in kw.items() if k != "cand_cap"})
    c1, p1, v1, cands = _search(gpu, e16, thr, screened=True, **kw)
    assert c1 == c0, (c1, c0)
    assert np.array_equal(p1, p0)
    assert np.array_equal(v1.view(np.uint32), v0.view(np.uint32))                # the same value BITS
    return c0, cands


def _clustered(n, d, seed, gpu, spread):
    """rows around cluster centres: pair cosines all the way from ~0 to 1, dense around the threshold"""
    g = torch.Generator(device=gpu).manual_seed(seed)
    centres = torch.randn(n // 8, d, device=gpu, generator=g)
    which = torch.randint(0, n // 8, (n,), device=gpu, generator=g)
    noise = torch.randn(n, d, device=gpu, generator=g) * (spread * torch.rand(n, 1, device=gpu, generator=g))
    return (centres[which] + noise).half()


@pytest.mark.parametrize("n,d", [(6000, 768), (3000, 512), (1111, 1000), (700, 128)])
def test_screened_equals_exact_on_clustered_rows(gpu, n, d):
    e16 = _clustered(n, d, n + d, gpu, 0.6)
    for thr in (0.96, 0.8):
        c, cands = _same(gpu, e16, thr)
        assert c > 50 and c <= cands < (1 << 20)                                 # pairs were found, through the candidates
    _same(gpu, e16, 0.9, fp16_compare=0)


def test_screened_planted_pairs_and_few_candidates(gpu):
    """random rows + planted pairs well above the threshold: the candidates are the planted pairs and nothing else (the screen's
    margin is ~0.06 of cosine, random rows stay below 0.2)"""
    g = torch.Generator(device=gpu).manual_seed(5)
    n, d, planted = 20_000, 768, 300
    e = torch.randn(n, d, device=gpu, generator=g)
    src = torch.randperm(n - planted, device=gpu, generator=g)[:planted]
    e[n - planted:] = e[src] + 0.1 * torch.randn(planted, d, device=gpu, generator=g)
    c, cands = _same(gpu, e.half(), 0.96)
    assert c == planted and cands == planted


def test_more_candidates_than_slots_runs_the_exact_search(gpu):
    e16 = _clustered(3000, 256, 9, gpu, 0.3)
    c, cands = _same(gpu, e16, 0.9, cand_cap=64)
    assert c > 64 and cands > 64                                                 # the counter ran over: the exact search answered
    c, p, v, _ = _search(gpu, e16, 0.9, capacity=10, cand_cap=64)                # and the output overflow keeps its meaning
    assert c > 10 and len(p) == 10


def test_rows_that_cannot_be_screened(gpu):
    g = torch.Generator(device=gpu).manual_seed(2)
    e = torch.randn(900, 384, device=gpu, generator=g)
    e[5] = 0                                                                      # zero rows: 0 / 0 = NaN under the reference's rule, no one's duplicate
    e[600] = 0
    e[7] = e[300]; e[899] = e[1]
    e[20] = 1e-7 * e[21]                                                          # a norm in fp16's subnormals (sends the call to the exact search)
    c, cands = _same(gpu, e.half(), 0.96)
    assert c >= 2
    c, cands = _same(gpu, torch.ones(5, 100).half(), 0.5, fp16_compare=0)         # identical rows, d not a multiple of 128
    assert c == 10
    c, _, _, _ = _search(gpu, e[:1].half(), 0.5)                                  # a single row
    assert c == 0


def test_config4_screened_100k(gpu):
    """BASELINE.json configs[4] through the screened search: exactly the 1 000 planted pairs, twice."""
    n, d, planted, thr = 100_000, 768, 1000, 0.96
    g = torch.Generator(device=gpu).manual_seed(7)
    e = torch.randn(n, d, device=gpu, generator=g)
    src = torch.randperm(n - planted, device=gpu, generator=g)[:planted]
    e[n - planted:] = e[src] + 0.1 * torch.randn(planted, d, device=gpu, generator=g)
    e16 = e.half().contiguous()
    del e
    want = {(int(s), n - planted + t) for t, s in enumerate(src.cpu().tolist())}
    c0, p0, v0, _ = _search(gpu, e16, thr, screened=False)
    for _ in range(2):
        c, p, v, cands = _search(gpu, e16, thr)
        assert c == planted and cands == planted
        assert {tuple(r) for r in p.tolist()} == want
        assert np.array_equal(p, p0) and np.array_equal(v.view(np.uint32), v0.view(np.uint32))
