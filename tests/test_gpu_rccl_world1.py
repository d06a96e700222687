"""RCCL on the hardware that is reachable from here: a process group of ONE rank on the box's GPU (backend "nccl" is RCCL on ROCm).
The collectives of the path -- all_gather_into_tensor of the step's results, the MAX all-reduce of the timing, the barrier, the
object gather of the rank report and sharding.gather_rows -- run through the library on device tensors; with one rank they move no
data between GPUs, but communicator set-up, stream ordering and the torch -> RCCL call path are the ones the 8-GPU run takes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from clip_assisted_data_labeling_amd.sharding import gather_rows
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
emb = torch.randn(37, 4, 768, device=dev)
out = torch.empty_like(emb)
dist.all_gather_into_tensor(out, emb)
t = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
objs = [None]
dist.all_gather_object(objs, {"rank": 0, "uuid": str(torch.cuda.get_device_properties(dev).uuid)})
full = gather_rows(emb, 37, dst=0)
torch.cuda.synchronize()
assert torch.equal(out, emb) and float(t.item()) == 1.25 and objs[0]["rank"] == 0 and torch.equal(full, emb)
dist.destroy_process_group()
print("rccl world-1 ok", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
"""


def test_rccl_process_group_of_one_rank_runs_the_paths_collectives(gpu, tmp_path):
    script = tmp_path / "child.py"
    script.write_text(_CHILD)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(script), ROOT, "29683"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "rccl world-1 ok" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])


_JOB_CHILD = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from clip_assisted_data_labeling_amd import vit_config
from clip_assisted_data_labeling_amd.embedder import HipViT
from clip_assisted_data_labeling_amd.nn_model import HipRegressor
from clip_assisted_data_labeling_amd.job import run_embed_job, synthetic_u8_source
from oracle import fcreg_oracle                                   # checker only
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1")
N, B = int(sys.argv[3]), 512
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
cfg = vit_config.ARCHS["ViT-L-14"]
vit = HipViT(cfg, vit_config.seeded_state_dict(cfg, 0), dev, precision="fp8")
rng = np.random.RandomState(21)
sizes = [4 * cfg.embed_dim, 264, 128, 64, 1]
Ws = [(rng.randn(sizes[i + 1], sizes[i]) / np.sqrt(sizes[i])).astype(np.float32) for i in range(4)]
bs = [(0.1 * rng.randn(sizes[i + 1])).astype(np.float32) for i in range(4)]
reg = HipRegressor([torch.from_numpy(w) for w in Ws], [torch.from_numpy(b) for b in bs], 0.01, dev)
sel = [0, 1, 2, 3]
enc = lambda c: vit.encode_score(c, reg, 4, sel)
src = synthetic_u8_source(cfg.image_size, 4, 4242, 0, dev)
res = run_embed_job(N, B, 4, cfg.embed_dim, 1, src, enc, dev, rank=0, world=1, gather=True, sync=torch.cuda.synchronize, gather_dst=0)
emb, score = res["emb"], res["score"]
assert res["batches"] == (N + B - 1) // B and emb.shape == (N, 4, cfg.embed_dim) and score.shape == (N, 1)
assert emb.data_ptr() != res["emb_local"].data_ptr()              # the gathered result came through the backend, not by reference
assert torch.equal(emb, res["emb_local"]) and torch.equal(score, res["score_local"])
assert torch.isfinite(emb).all() and torch.isfinite(score).all()
assert float((emb.norm(dim=-1) - 1.0).abs().max()) < 1e-5
# first and last batch again from a fresh generator with the same seed: the stored rows bit for bit
again = synthetic_u8_source(cfg.image_size, 4, 4242, 0, dev)
first = again(0, B)
e0, s0 = enc(first)
assert torch.equal(e0, emb[:B]) and torch.equal(s0, score[:B])
nb_last = N - (res["batches"] - 1) * B
for b0 in range(B, N - nb_last, B):
    again(b0, B)                                                  # (the counter stream: draw what the job drew)
last = again(N - nb_last, nb_last)
e1, s1 = enc(last)
assert torch.equal(e1, emb[N - nb_last:]) and torch.equal(s1, score[N - nb_last:])
# every score is the C oracle's on the stored embedding (utils/nn_model.py:38-41), tolerance of BASELINE.json's north_star
ref = fcreg_oracle.forward_c(Ws, bs, emb.cpu().numpy().reshape(N, -1))
err = float(np.abs(score.cpu().numpy() - ref).max())
assert err < 1e-4, err
dist.barrier()
dist.destroy_process_group()
print(f"config3 shard ok: {N} images, {res['batches']} batches, {N / res['t_encode']:.0f} images/s, gather {res['t_gather'] * 1e3:.1f} ms, score err {err:.2e}")
"""


def test_config3_fp8_job_of_20k_images_inside_a_one_rank_rccl_group(gpu, tmp_path):
    """BASELINE.json configs[3] at a 1-GPU shard's scale under the driver: 20 480 synthetic images (40 batches of 512, uint8 crops
    generated on the device per batch) through job.run_embed_job with the e4m3 block GEMMs inside an initialised RCCL process group
    of one rank, results gathered with gather_dst=0 THROUGH the backend (/root/reference/_1_embed_with_CLIP.py:100-170 is the loop
    this replaces).  Checked: every row finite and unit norm, first and last batch reproduced bit for bit from a fresh generator,
    every score against the C oracle.  The 8-rank half of the config needs a multi-GPU node."""
    script = tmp_path / "job_child.py"
    script.write_text(_JOB_CHILD)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(script), ROOT, "29684", "20480"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and "config3 shard ok" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])
    print(p.stdout.strip().splitlines()[-1])
