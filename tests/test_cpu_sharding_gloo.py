"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding and the gather collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clip_assisted_data_labeling_amd.sharding import gather_rows, shard_bounds, shard_list


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 8, 9, 1000003):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_list(list("abcde"), 1, 2) == ["d", "e"]
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_bounds(n_total, rank, world)
        # stand-in for the per-rank encode+score result: row i of the job is filled with i
        emb = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1).expand(hi - lo, 4, 6).contiguous()
        score = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1) * 0.5
        full_e = gather_rows(emb, n_total)
        full_s = gather_rows(score, n_total)
        ok = (full_e.shape == (n_total, 4, 6) and torch.equal(full_e[:, 0, 0], torch.arange(n_total, dtype=torch.float32))
              and torch.equal(full_s[:, 0], torch.arange(n_total, dtype=torch.float32) * 0.5))
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, bool(ok), float(t.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [10, 7, 1])
def test_gather_rows_world2_gloo(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, True, 2.0), (1, True, 2.0)]


def test_gather_rows_single_process_passthrough():
    x = torch.randn(5, 3)
    assert gather_rows(x, 5) is x
    with pytest.raises(ValueError):
        gather_rows(x, 6)
