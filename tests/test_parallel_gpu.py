"""GPU: the collectives of the N > 1 paths on RCCL itself (backend "nccl"), with the one rank a 1-GPU box offers.  World size 1 cannot
show ordering between ranks (tests/test_parallel_cpu.py does that over gloo, world size 2) -- it shows that the calls the multi-GPU
paths make (`all_gather_into_tensor`, `reduce_scatter_tensor`, `all_reduce`, the bench's barrier + MAX reduction) are accepted by RCCL
with the tensors the product hands them (dtypes, alignment, device), on the real model.  Runs in a child process so that the process
group does not outlive the test."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["UFV_ROOT"]); sys.path.insert(0, os.path.join(os.environ["UFV_ROOT"], "tests"))
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["UFV_PORT"], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from ufvideo_amd import parallel as P, train as T
from test_model_gpu import tiny_model
from conftest import t
# 1. frame-sharded encode == the model's own encode (one rank holds every frame; the all-gather is a real RCCL call)
m, a, w = tiny_model()
video = t(a["video"]).to(dev)
with torch.no_grad():
    whole = m.encode_images_or_videos([(video, "video")])[0]
# tiny model: spatial_conv (Conv3d padding 1) cannot be frame-sharded -> refused; the v35 geometry is covered by the bench mode below
try:
    P.encode_frame_sharded(m, video)
    raise SystemExit("expected a refusal for the padding-1 connector")
except ValueError:
    pass
toks = P.all_gather_tokens(whole.float().contiguous(), [whole.shape[0]])
assert torch.equal(toks, whole.float())
# 2. the trainer's exchange primitives on an RCCL group
full = torch.randn(4096, device=dev); shard = torch.empty(4096, device=dev)
T.reduce_scatter_mean(shard, full); assert torch.equal(shard, full)
out = torch.zeros(4096, device=dev, dtype=torch.bfloat16); out.copy_(full)
T.all_gather_shards(out); assert torch.equal(out, full.to(torch.bfloat16))
# 3. the bench protocol's collectives (barrier, MAX over ranks of a float64 device scalar)
dist.barrier(); torch.cuda.synchronize()
tt = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(tt, op=dist.ReduceOp.MAX); assert float(tt) == 1.25
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK")
'''


def test_collectives_on_rccl_world_size_one():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UFV_ROOT=root, UFV_PORT="29517", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_gpus2_on_a_one_gpu_box_fails_loudly():
    """The driver's command form `python3 bench.py --gpus N` on a host with fewer than N GPUs must not report a line at all (round 2's
    bench ignored --gpus and printed n_gpus: 1)."""
    import torch
    have = torch.cuda.device_count()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(have + 1), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "visible" in r.stderr and "{" not in r.stdout
