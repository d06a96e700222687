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
