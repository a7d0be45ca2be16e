"""The N > 1 path on CPU: two processes over gloo shard a global batch, accumulate
Keras-Mean metric states independently and all-gather them (the RCCL all-gather of
bench.py / main.py on the GPU box)."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from m4depth_amd import dist as D
    from m4depth_amd import metrics as MT
    r, w, _, dev = D.init_from_env(backend="gloo")
    rng = np.random.default_rng(0)
    gt = torch.from_numpy((1 + 70 * rng.random([8, 2, 8, 8, 1])).astype(np.float32))      # 8 batches of 2
    est = gt * torch.from_numpy((1 + 0.1 * rng.standard_normal(gt.shape)).astype(np.float32))
    lo, hi = D.shard_range(8, r, w)
    mets = MT.default_metrics()
    for i in range(lo, hi):
        for m in mets:
            m.update_state(gt[i], est[i])
    D.barrier(dev)
    gathered = D.all_gather_metric_states(mets, dev)
    res = D.reduce_metric_states(gathered)
    tmax = D.max_over_ranks(float(r + 1), dev)
    per_rank = D.all_gather_floats(10.0 * (r + 1), dev)
    q.put((r, tuple(gathered.shape), res.tolist(), tmax, per_rank))
    torch.distributed.destroy_process_group()


def test_two_rank_metric_allgather():
    sys.path.insert(0, ROOT)
    from m4depth_amd import metrics as MT
    world = 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference over all 8 batches
    rng = np.random.default_rng(0)
    gt = torch.from_numpy((1 + 70 * rng.random([8, 2, 8, 8, 1])).astype(np.float32))
    est = gt * torch.from_numpy((1 + 0.1 * rng.standard_normal(gt.shape)).astype(np.float32))
    mets = MT.default_metrics()
    for i in range(8):
        for m in mets:
            m.update_state(gt[i], est[i])
    ref = np.array([float(m.result()) for m in mets])
    for r, shape, res, tmax, per_rank in outs:
        assert shape == (2, 7, 2)
        assert np.allclose(res, ref, rtol=1e-6)
        assert tmax == 2.0
        assert per_rank == [10.0, 20.0]                      # bench.py's per-rank report


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from m4depth_amd import dist as D
    r, w, _, dev = D.init_from_env(backend="gloo")
    params = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2))]
    params[0].grad = torch.full((3, 4), float(r + 1))
    params[1].grad = torch.arange(5, dtype=torch.float32) * (r + 1)
    # params[2] has no gradient on any rank (a frozen layer): must be skipped consistently
    D.all_reduce_gradients(params)
    q.put((r, params[0].grad.tolist(), params[1].grad.tolist(), params[2].grad is None))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_allreduce():
    """Data-parallel training: gradients of both ranks are averaged by one flat all-reduce."""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, g0, g1, none2 in outs:
        assert np.allclose(g0, 1.5)
        assert np.allclose(g1, 1.5 * np.arange(5))
        assert none2
