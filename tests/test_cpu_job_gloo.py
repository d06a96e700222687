"""Control flow of the whole-job runner (BASELINE.json configs[3]) on CPU: world_size-2 gloo processes, a stand-in encoder.
Covers the shard bounds, the ragged last batch, results landing at their global image index, and the single final gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clip_assisted_data_labeling_amd.job import run_embed_job, synthetic_u8_source

CROPS, E = 4, 6


def _source(first_image, n_images):
    """Stand-in generator: every pixel of crop c of image i holds (i * 4 + c) % 251 -- the stand-in encoder reads it back."""
    idx = torch.arange(first_image * CROPS, (first_image + n_images) * CROPS) % 251
    return idx.to(torch.uint8).view(-1, 1, 1, 1).expand(-1, 3, 2, 2).contiguous()


def _encode_score(crops):
    v = crops[:, 0, 0, 0].float()
    emb = v.view(-1, CROPS, 1).expand(-1, CROPS, E).contiguous()
    return emb, emb[:, 0, :1] * 0.5


def _expected(n):
    v = (torch.arange(n * CROPS) % 251).float().view(n, CROPS, 1).expand(n, CROPS, E)
    return v, v[:, 0, :1] * 0.5


def test_job_single_process_ragged_batches():
    calls = []
    res = run_embed_job(11, 4, CROPS, E, 1, _source, _encode_score, "cpu", progress=lambda d, t: calls.append((d, t)))
    e, s = _expected(11)
    assert res["batches"] == 3 and (res["lo"], res["hi"]) == (0, 11) and calls[-1] == (11, 11)
    assert torch.equal(res["emb"], e) and torch.equal(res["score"], s)
    assert run_embed_job(0, 4, CROPS, E, 1, _source, _encode_score, "cpu")["emb"].shape == (0, CROPS, E)
    with pytest.raises(ValueError):
        run_embed_job(5, 0, CROPS, E, 1, _source, _encode_score, "cpu")


def test_synthetic_source_is_reproducible_per_rank():
    a = synthetic_u8_source(8, CROPS, 7, 0, "cpu")
    b = synthetic_u8_source(8, CROPS, 7, 0, "cpu")
    c = synthetic_u8_source(8, CROPS, 7, 1, "cpu")
    x, y = a(0, 3), a(3, 2)
    assert x.dtype == torch.uint8 and x.shape == (12, 3, 8, 8)
    assert torch.equal(x, b(0, 3)) and torch.equal(y, b(3, 2))          # same seed, same batch sequence -> same stream
    assert not torch.equal(x, c(0, 3))                                   # another rank draws another stream


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, batch, gather_dst, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = run_embed_job(n_total, batch, CROPS, E, 1, _source, _encode_score, "cpu", rank, world, gather=True,
                            gather_dst=gather_dst)
        e, s = _expected(n_total)
        if gather_dst is None or rank == gather_dst:
            ok = torch.equal(res["emb"], e) and torch.equal(res["score"], s)
        else:
            ok = res["emb"] is None and res["score"] is None           # a true gather: only the destination holds the job
        q.put((rank, bool(ok), res["n_local"], res["batches"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gather_dst", [None, 0])
@pytest.mark.parametrize("n_total,batch,expect", [(11, 3, [(0, True, 6, 2), (1, True, 5, 2)]), (1, 4, [(0, True, 1, 1), (1, True, 0, 0)]),
                                                  (12, 4, [(0, True, 6, 2), (1, True, 6, 2)])])
def test_job_world2_gloo(n_total, batch, expect, gather_dst):
    assert _run_world(2, n_total, batch, gather_dst) == expect


def _run_world(world, n_total, batch, gather_dst):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, batch, gather_dst, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("gather_dst", [None, 0])
def test_job_world8_gloo_ragged_shards(gather_dst):
    """The rank count of BASELINE.json configs[3] (8 ranks) with a job that does not divide: 1003 images -> shards of 126 x3 and
    125 x5, batches of 50 -> every rank ends on a ragged batch (26 / 25 images); rank 0 (or every rank) holds all rows in order."""
    res = _run_world(8, 1003, 50, gather_dst)
    assert res == [(r, True, 126 if r < 3 else 125, 3) for r in range(8)]


def test_job_world8_gloo_fewer_images_than_ranks():
    res = _run_world(8, 5, 4, 0)                                          # ranks 5..7 own nothing and still take part in the gather
    assert res == [(r, True, 1 if r < 5 else 0, 1 if r < 5 else 0) for r in range(8)]
