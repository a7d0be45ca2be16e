"""The RCCL leg of the multi-GPU path on the one GPU this box has: a SINGLE-rank ``nccl`` group made to issue every collective
an 8-GPU job issues (m4depth_amd.dist.single_rank_shortcut = False) -- RCCL initialisation, the metric-state
all_gather_into_tensor, the float64 MAX all-reduce of the timing contract, the device barrier, the per-rank rate gather, the
flat gradient all-reduce -- and bench.py's timed job on top of them.  (World size 2 needs two GPUs: RCCL refuses two ranks on one
device; the sharding logic itself is covered over gloo in tests/test_dist_gloo.py and tests/test_bench_dist.py.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import os, sys, types
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from m4depth_amd import dist as D, metrics as MT
import bench as B
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29400 + os.getpid() %% 500))
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
D.single_rank_shortcut = False
assert D._collectives_on() and dist.get_backend() == "nccl"
mets = MT.default_metrics()
gt = torch.rand(2, 8, 8, device=dev) + 1.0
for m in mets:
    m.update_state(gt, gt * 1.01)
g = D.all_gather_metric_states(mets, dev)
local = torch.stack([m.state(dev) for m in mets]).to(torch.float32)
assert tuple(g.shape) == (1, len(mets), 2) and torch.equal(g[0], local), g
red = D.reduce_metric_states(g)
assert abs(float(red[0]) - 0.01) < 1e-4, red
assert D.max_over_ranks(3.25, dev) == 3.25
assert D.all_gather_floats(2.5, dev) == [2.5]
D.barrier(dev)
p = torch.nn.Parameter(torch.ones(5, device=dev)); p.grad = torch.full((5,), 3.0, device=dev)
q = torch.nn.Parameter(torch.ones(2, 2, device=dev)); q.grad = torch.full((2, 2), -1.0, device=dev)
D.all_reduce_gradients([p, q])
assert torch.equal(p.grad, torch.full((5,), 3.0, device=dev)) and torch.equal(q.grad, torch.full((2, 2), -1.0, device=dev))
# bench.py's timed job (barrier-bracketed region, max over ranks, per-rank gather) over the same group
args = types.SimpleNamespace(gpus=1, steps=3, warmup=1, batch=1, seq_len=3, height=16, width=32, levels=2, dscv_range=4, sncv_range=3,
                             repeat_regions=2)
x = torch.zeros(1, device=dev)
def step():
    x.add_(1.0)
dt, per_rank_s, ms_runs, stag = B.timed_job(step, args, D, dev, torch.cuda.synchronize, stagger_us=9)
assert len(per_rank_s) == 1 and dt >= per_rank_s[0] > 0 and len(ms_runs) == 2 and stag == [9.0]
assert float(x.item()) == 3 * 2
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
''' % ROOT


@pytest.mark.gpu
def test_rccl_single_rank_runs_every_collective_of_the_multi_gpu_path():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
