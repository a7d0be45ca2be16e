"""Train / eval / predict driver -- the counterpart of the reference ``main.py --mode=train``
(main.py:73-109), ``--mode=eval`` (:111-148) and ``--mode=predict`` (:150-172) over a seeded
synthetic dataset (no dataset or checkpoint exists in the build environment).

    python -m m4depth_amd.main --mode=eval --arch_depth=6 --seq_len=4 --batch_size=2 \\
           --n_batches=4 --ckpt_dir=/tmp/m4d
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
           -m m4depth_amd.main --mode=eval --batch_size=256        # global batch, sharded

Flag names follow ``m4depth_options.py`` where the flag exists there.  Like the
reference it writes the 7 metrics with ``%.18e`` (one per line, as np.savetxt does for the 1-D list) to
``<ckpt_dir>/perfs-<dataset>.txt`` (main.py:144-148).  With several ranks the batch
is sharded (m4depth_amd.dist) and the metric accumulators are all-gathered over RCCL.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

from . import dist as D
from . import synthetic as S
from .metrics import default_metrics
from .network import M4Depth, M4depthAblationParameters, GraphedSequence


def build_parser():
    p = argparse.ArgumentParser(description="M4Depth (MI355X native) train / eval / predict driver")
    p.add_argument("--mode", default="eval", choices=["train", "finetune", "eval", "validation", "predict"])
    p.add_argument("--dataset", default="synthetic", choices=["synthetic", "midair", "tartanair", "kitti-raw"],
                   help="a dataset of the reference (m4depth_options.py:11-14) or the seeded synthetic one")
    p.add_argument("--db_path_config", default=None, help="json file with the dataset roots (m4depth_options.py:27-29)")
    p.add_argument("--records_path", default=None, help="directory of trajectory csv files (m4depth_options.py:33-35)")
    p.add_argument("--no_augmentation", action="store_true")
    p.add_argument("--epochs", type=int, default=1, help="train on a real dataset: passes over the records")
    p.add_argument("--arch_depth", type=int, default=6, help="number of pyramid levels (m4depth_options.py:64)")
    p.add_argument("--seq_len", type=int, default=4)
    p.add_argument("--db_seq_len", type=int, default=None)
    p.add_argument("--batch_size", type=int, default=1, help="GLOBAL batch (sharded over ranks)")
    p.add_argument("--n_batches", type=int, default=2, help="synthetic dataset: batches to generate")
    p.add_argument("--height", type=int, default=None, help="frame height (default: 384 synthetic, else the dataset's)")
    p.add_argument("--width", type=int, default=None)
    p.add_argument("--ckpt_dir", default="ckpt")
    p.add_argument("--tf_checkpoint", default=None,
                   help="TensorFlow checkpoint of the reference model (prefix, or a directory holding a 'checkpoint' state "
                        "file, e.g. pretrained_weights/midair): used instead of <ckpt_dir>/train for the initial weights")
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--graph", action="store_true", help="replay the sequence forward from a hipGraph")
    p.add_argument("--host_input", action="store_true",
                   help="synthetic eval: hand every batch over from pinned HOST memory (the PCIe copy is part of the step) and "
                        "print the frames/s of the loop -- the PCIe-inclusive rate, next to bench.py's HBM-resident one")
    p.add_argument("--learning_rate", type=float, default=1e-4, help="Adam step size (main.py:88)")
    p.add_argument("--save_every", type=int, default=0, help="train: checkpoint every N steps (0 = at the end)")
    # ablation flags, spelled as in m4depth_options.py:67-84
    for flag in ("DINL", "SNCV", "time_recurr", "feature_normalization", "feature_subdivision", "level_memory"):
        p.add_argument(f"--no_{flag}", action="store_true")
    return p


def ablation_from_args(args):
    return M4depthAblationParameters(not args.no_DINL, not args.no_SNCV, not args.no_time_recurr,
                                     not args.no_feature_normalization, not args.no_feature_subdivision,
                                     not args.no_level_memory)                  # m4depth_options.py:96-98


def synthetic_batches(args, rank, world, dev, on_host=False):
    """``on_host``: the batch stays in pinned host memory (the consumer uploads it: PCIe inside the step)."""
    lo, hi = D.shard_range(args.batch_size, rank, world)
    h, w = args.height or 384, args.width or 1280
    put = (lambda t: t.pin_memory()) if on_host else (lambda t: t.to(dev))
    for i in range(args.n_batches):
        samples, cam = S.make_sequence(args.batch_size, args.seq_len, h, w, seed=args.seed + i)
        data = {k: put(torch.from_numpy(np.stack([s[k][lo:hi] for s in samples], axis=1)))
                for k in ("depth", "RGB_im", "rot", "trans")}
        data["new_traj"] = torch.from_numpy(np.stack([s["new_traj"][lo:hi] for s in samples], axis=1))
        data["camera"] = {k: put(torch.from_numpy(v[lo:hi])) for k, v in cam.items()}
        yield data


def _upload(data, dev):
    """Host-resident batch -> device (non-blocking copies on the current stream); new_traj stays on the host."""
    out = {k: (v.to(dev, non_blocking=True) if isinstance(v, torch.Tensor) and k != "new_traj" else v) for k, v in data.items()
           if k != "camera"}
    out["camera"] = {k: v.to(dev, non_blocking=True) for k, v in data["camera"].items()}
    return out


def dataset_batches(args, usecase, rank, world, dev):
    """Batches of one of the reference's datasets (main.py:67,77,120,152).  Returns (iterable, depth_type).
    With several ranks the PER-RANK batch is batch_size / world; the loader shards the chunk list itself
    (``shard=(rank, world)``): every rank gets the same number of batches and reads / decodes only its own
    sequences.  Streaming evaluation is inherently sequential and stays single-rank."""
    import json
    from . import dataloaders as dl
    if args.db_path_config is None or args.records_path is None:
        raise SystemExit("--dataset %s needs --db_path_config and --records_path" % args.dataset)
    cfg_dir = os.path.dirname(os.path.abspath(args.db_path_config))
    with open(args.db_path_config) as fh:
        roots = {k: (v if os.path.isabs(v) else os.path.normpath(os.path.join(cfg_dir, v)))
                 for k, v in json.load(fh).items() if not k.startswith("_")}    # m4depth_options.py:88-94
    loader = dl.get_loader(args.dataset)
    settings = dl.DataloaderParameters(roots, args.records_path, args.db_seq_len, args.seq_len, not args.no_augmentation)
    streaming = usecase in ("eval", "predict") and args.db_seq_len is None
    if streaming and world > 1:
        raise SystemExit("streaming evaluation (no --db_seq_len) is sequential: run it on one rank")
    per_rank = 1 if streaming else D.shard_range(args.batch_size, rank, world)[1] - D.shard_range(args.batch_size, rank, world)[0]
    kw = {"out_size": [args.height, args.width]} if args.height and args.width else {}
    ds = loader.get_dataset(usecase, settings, batch_size=per_rank, device=dev, seed=args.seed,
                            shard=(0, 1) if streaming else (rank, world), **kw)
    return ds, loader.depth_type


def _train_dir(args):
    return os.path.join(args.ckpt_dir, "train")


def latest_checkpoint(args):
    """Newest ``ckpt-<step>.npz`` of <ckpt_dir>/train (the role of tf.train.latest_checkpoint,
    callbacks.py:84), or None."""
    d = _train_dir(args)
    if not os.path.isdir(d):
        return None
    steps = sorted(int(f[5:-4]) for f in os.listdir(d) if f.startswith("ckpt-") and f.endswith(".npz"))
    return (os.path.join(d, f"ckpt-{steps[-1]}.npz"), steps[-1]) if steps else None


def load_weights(args, ablation):
    """Weights for eval / predict: the latest training checkpoint when there is one (callbacks.py:104-111),
    otherwise the seeded random initialisation."""
    if args.tf_checkpoint:
        from . import tf_checkpoint as TC
        prefix = TC.latest_checkpoint(args.tf_checkpoint) if os.path.isdir(args.tf_checkpoint) else args.tf_checkpoint
        if prefix is None:
            raise SystemExit("no TensorFlow checkpoint found in %s" % args.tf_checkpoint)
        print("Restoring weights from TensorFlow checkpoint %s" % prefix)
        return TC.load_m4depth_weights(prefix, args.arch_depth)
    ck = latest_checkpoint(args)
    if ck is not None:
        print("Restoring weights from %s" % ck[0])
        with np.load(ck[0]) as z:
            return {k: z[k] for k in z.files if not k.startswith("opt.")}
    return S.init_weights(args.arch_depth, seed=42, ablation=ablation)


def train(args, ablation, rank, world, dev):
    """main.py:73-109: Adam(1e-4), RMSE_log as the tracked metric, resume from the latest checkpoint.
    Several ranks = data parallel: the global batch is sharded and the gradients are averaged with
    one RCCL all-reduce per step (dist.all_reduce_gradients)."""
    from . import training as TR
    from .metrics import RootMeanSquaredLogError
    torch.manual_seed(42)                                                       # tf.random.set_seed(42), main.py:76
    depth_type = {"kitti-raw": "velodyne"}.get(args.dataset, "map")              # the loader's depth_type (main.py:79)
    model = M4Depth(depth_type=depth_type, nbre_levels=args.arch_depth, ablation_settings=ablation, is_training=True)
    model.load_numpy_weights(load_weights(args, ablation), dev)
    TR.set_trainable(model)
    opt = torch.optim.Adam(model.parameters(), lr=args.learning_rate, eps=1e-7)   # Keras Adam: epsilon 1e-7
    model.compile(optimizer=opt, metrics=[RootMeanSquaredLogError()])
    ck = latest_checkpoint(args)
    step0 = 0
    if ck is not None:
        step0 = ck[1]
        opt_path = ck[0][:-4] + ".opt.pt"
        if os.path.isfile(opt_path):
            opt.load_state_dict(torch.load(opt_path, map_location=dev))
    sync = D.all_reduce_gradients if world > 1 else None

    def save(step):
        if rank != 0:
            return
        os.makedirs(_train_dir(args), exist_ok=True)
        path = os.path.join(_train_dir(args), f"ckpt-{step}.npz")
        np.savez(path, **model.numpy_weights())
        torch.save(opt.state_dict(), path[:-4] + ".opt.pt")
        print("saved", path)

    step = step0
    for epoch in range(args.epochs if args.dataset != "synthetic" else 1):
        if args.dataset == "synthetic":
            batches = synthetic_batches(args, rank, world, dev)
        else:
            batches, _ = dataset_batches(args, args.mode, rank, world, dev)
        for data in batches:
            out = model.train_step(data, grad_sync=sync)
            step += 1
            if rank == 0:
                print(f"step {step}: loss {float(out['loss']):.6f}  RMSE_log {float(out['RMSE_log']):.6f}")
            if args.save_every and step % args.save_every == 0:
                save(step)
    if not args.save_every or step % args.save_every != 0:
        save(step)
    return 0


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, world, _, dev = D.init_from_env()
    if dev.type != "cuda":
        raise SystemExit("m4depth_amd.main needs a GPU: the hot path has no CPU fallback")
    ablation = ablation_from_args(args)
    if args.mode in ("train", "finetune"):
        return train(args, ablation, rank, world, dev)
    model = M4Depth(nbre_levels=args.arch_depth, ablation_settings=ablation)
    model.load_numpy_weights(load_weights(args, ablation), dev)
    model.compile(metrics=default_metrics())
    runner = None
    preds = []
    host_input = args.host_input and args.dataset == "synthetic" and args.mode != "predict"
    if args.dataset == "synthetic":
        batches = synthetic_batches(args, rank, world, dev, on_host=host_input)
        if host_input:
            batches = list(batches)                # generation is not part of the measured loop
    else:
        batches, _ = dataset_batches(args, "predict" if args.mode == "predict" else "eval", rank, world, dev)
    import time
    n_frames, t_loop = 0, None
    for batch_idx, data in enumerate(batches):
        if host_input:
            if batch_idx == 1:                                 # the first batch is the warm-up (allocations, graph capture)
                torch.cuda.synchronize()
                t_loop, n_frames = time.perf_counter(), 0
            data = _upload(data, dev)
            n_frames += data["RGB_im"].shape[0] * data["RGB_im"].shape[1]
        if data["RGB_im"].dim() == 4:            # streaming frame of a real dataset (db_seq_len None)
            if args.mode == "predict":
                preds.append(model.predict_step(data)["depth"].cpu().numpy())
            else:
                model.test_step(data)
            continue
        if args.mode == "predict":
            for t in range(data["RGB_im"].shape[1]):
                frame = {k: data[k][:, t] for k in ("depth", "RGB_im", "rot", "trans", "new_traj")}
                frame["camera"] = data["camera"]
                preds.append(model.predict_step(frame)["depth"].cpu().numpy())
            continue
        if args.graph:
            if runner is None:
                model.test_step(data)              # eager warm-up (state and scratch allocation, packed weights)
                for m in model.compiled_metrics:
                    m.reset_state()
                runner = GraphedSequence(model, data)
            model.graphed_test_step(data, runner)
        else:
            model.test_step(data)
    if args.mode == "predict":
        if rank == 0:
            os.makedirs(args.ckpt_dir, exist_ok=True)
            np.save(os.path.join(args.ckpt_dir, "predictions.npy"), np.concatenate(preds, axis=0))
        return 0
    if host_input and t_loop is not None:
        torch.cuda.synchronize()
        dt = time.perf_counter() - t_loop
        if rank == 0 and n_frames:
            print(f"host-resident input (pinned memory, PCIe copy inside the step): {n_frames / dt:.1f} frames/s per rank "
                  f"over {n_frames} frames")
    gathered = D.all_gather_metric_states(model.compiled_metrics, dev)
    metrics = D.reduce_metric_states(gathered).cpu().numpy()
    if rank == 0:
        os.makedirs(args.ckpt_dir, exist_ok=True)
        path = os.path.join(args.ckpt_dir, "perfs-" + args.dataset + ".txt")
        np.savetxt(path, metrics, fmt='%.18e', delimiter='\t', newline='\n')      # main.py:147 (1-D: one metric per line)
        for m, v in zip(model.compiled_metrics, metrics):
            print(f"{m.name}: {v:.6f}")
        print("wrote", path)
    return 0


if __name__ == "__main__":
    sys.exit(main())
