"""Data-parallel evaluation across the GPUs of one MI355X node.

The reference is single-GPU (main.py:44-46).  Trajectories are independent and
the recurrent state is per sample, so the path shards along the batch with NO
collective on the data path: rank r evaluates sequences [r*B/N, (r+1)*B/N) with
replicated weights.  The only exchange is one all-gather of each rank's Keras
``Mean`` accumulators -- (total, count) x 7 metrics = 14 floats / rank -- over
RCCL (``backend="nccl"`` is RCCL on ROCm; xGMI) at the end of the evaluation.
On CPU (tests) the same code runs over ``gloo``.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


# A single-rank group skips the collectives (nothing to exchange).  False = issue them anyway: on a 1-GPU box this is the only way
# to run the very calls an 8-GPU job makes -- RCCL initialisation, all_gather_into_tensor on the metric states, the float64 MAX
# all-reduce, the device barrier -- on real hardware (tests/test_gpu_dist.py).
single_rank_shortcut = True


def _collectives_on():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or not single_rank_shortcut)


def init_from_env(backend=None):
    """One process per GPU, launched by torch.distributed.run.  Returns
    (rank, world_size, local_rank, device).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device(f"cuda:{local_rank}") if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    global single_rank_shortcut
    lone_collectives = world == 1 and os.environ.get("M4D_DIST_SINGLE_RANK_COLLECTIVES", "0") == "1"
    if (world > 1 or lone_collectives) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world)
    if lone_collectives:
        # one rank of an N-GPU job, alone on a 1-GPU box: the same RCCL calls around the same per-rank workload
        # (bench.py --batch 32 = BASELINE configs[3]'s per-rank leg)
        single_rank_shortcut = False
    return rank, world, local_rank, device


def shutdown():
    """Tear the process group down.  bench.py deliberately does NOT call this: its ranks exit right after their last collective
    (process exit releases everything; torch prints a warning), because a teardown that could wait on a peer is a risk the
    never-yet-measured 8-GPU run should not carry.  For callers that keep the process alive."""
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def shard_range(global_batch, rank, world):
    """Contiguous shard of sequences owned by ``rank``."""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def all_gather_metric_states(metrics, device):
    """All-gather every rank's (total, count) pairs; returns [world, n_metrics, 2].
    One small collective; launched after the last batch."""
    local = torch.stack([m.state(device) for m in metrics]).to(torch.float32).contiguous()   # [n,2]
    if not _collectives_on():
        return local.unsqueeze(0)
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(local.shape), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out.view(-1), local.view(-1)) if device.type == "cuda" else \
        dist.all_gather(list(out.unbind(0)), local)
    return out


def reduce_metric_states(gathered):
    """Global Keras-``Mean`` result: sum of totals / sum of counts per metric."""
    total = gathered[..., 0].sum(dim=0)
    count = gathered[..., 1].sum(dim=0).clamp_min(1.0)
    return total / count


def max_over_ranks(value, device):
    """MAX-reduce a python float over ranks (bench timing contract)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if _collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_floats(value, device):
    """One python float per rank -> the list over ranks (report only: per-rank frames/s of the scaling bench)."""
    if not _collectives_on():
        return [float(value)]
    world = dist.get_world_size()
    local = torch.tensor([float(value)], dtype=torch.float32, device=device)
    out = torch.empty(world, dtype=torch.float32, device=device)
    if device.type == "cuda":
        dist.all_gather_into_tensor(out, local)
    else:
        dist.all_gather(list(out.unbind(0)), local[0])
    return [float(v) for v in out.tolist()]


def barrier(device):
    if _collectives_on():
        if device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def all_reduce_gradients(params, average=True):
    """Data-parallel training (an extension: the reference trains on one GPU): ONE flat
    all-reduce of every gradient.  The model has ~4.6 M parameters (18 MB), below the size where
    splitting into buckets overlapped with backward would pay on xGMI (a ring step per link is
    latency-bound under ~32 MB), so the whole set travels as a single bucket."""
    if not _collectives_on():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    if average:
        flat /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
