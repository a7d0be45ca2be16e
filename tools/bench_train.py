"""Training-step throughput on the MI355X (the reference trains on Mid-Air 384x384 crops,
batch 3, seq_len 4: scripts/1a-train-midair.sh).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=384)
    ap.add_argument("--width", type=int, default=384)
    ap.add_argument("--batch", type=int, default=3)
    ap.add_argument("--seq_len", type=int, default=4)
    ap.add_argument("--levels", type=int, default=6)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--eager", action="store_true", help="no hipGraph capture of the step")
    args = ap.parse_args()
    import m4depth_amd as M
    from m4depth_amd import synthetic as S, training as TR
    from m4depth_amd.metrics import RootMeanSquaredLogError
    dev = torch.device("cuda:0")
    model = M.M4Depth(nbre_levels=args.levels, is_training=True).load_numpy_weights(S.init_weights(args.levels, seed=42), dev)
    TR.set_trainable(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, eps=1e-7, capturable=not args.eager)
    model.compile(optimizer=opt, metrics=[RootMeanSquaredLogError()])
    samples, cam = S.make_sequence(args.batch, args.seq_len, args.height, args.width, seed=1)
    data = {k: torch.from_numpy(np.stack([s[k] for s in samples], axis=1)).to(dev) for k in ("depth", "RGB_im", "rot", "trans")}
    data["new_traj"] = np.stack([s["new_traj"] for s in samples], axis=1)
    data["camera"] = {k: torch.from_numpy(v).to(dev) for k, v in cam.items()}
    losses = []
    if args.eager:
        step = lambda: model.train_step(data)["loss"]
    else:
        runner = TR.GraphedTrainStep(model, data, opt, warmup=args.warmup)
        step = lambda: runner()[0]
    for _ in range(args.warmup):
        losses.append(float(step()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    losses.append(float(loss))
    print(json.dumps({"metric": "train sequences/s", "value": round(args.batch / dt, 2), "ms_per_step": round(dt * 1e3, 2),
                      "frames_per_s": round(args.batch * args.seq_len / dt, 1),
                      "config": f"{args.height}x{args.width} L={args.levels} batch {args.batch} seq_len {args.seq_len}",
                      "loss_first": losses[0], "loss_last": losses[-1],
                      "launch": "eager" if args.eager else "hipGraph replay of the whole step", "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
