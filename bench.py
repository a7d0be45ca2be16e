#!/usr/bin/env python
"""bench.py -- frames/s of the M4Depth per-frame inference path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it
is launched under ``python -m torch.distributed.run`` with one rank per GPU.

One *step* = one pass of the hot path over one batch of synthetic input = the
reference's ``test_step`` on a 5-D sequence batch (m4depth_network.py:433-474):
encoder + 6-level parallax-cost-volume decoder over ``seq_len`` frames (frame 0
carries ``new_traj`` and only seeds the recurrent state, exactly as
dataloaders/generic.py:139 produces it) + the 7 depth metrics on the last frame.
Default workload = BASELINE.json configs[1]: 384x1280, 6 levels, seq_len 4, DSCV
range 4 / SNCV range 3, batch 1 per GPU.  Inputs are resident in HBM before the
timed region.  ``value`` = all frames processed by all ranks / max-over-ranks time.

The JSON line also carries
  roofline     -- the dominant hand-written kernel of the step: the level-1 refiner
                  128->128 convolution (Winograd F(2x2,3x3) on fp32 MFMA): the layer's algorithmic
                  (direct-convolution) flops / HIP-event time on the launch stream, against the 157.3 TFLOP/s
                  fp32-MFMA peak (> 1 possible: Winograd executes 2.25x fewer), + the executed-flops fraction;
                  roofline_dscv / roofline_sncv: the level-1 cost-volume kernels, algorithmic
                  bytes / time against the 8 TB/s HBM3E peak;
  cpu_baseline -- the CPU oracle (a numpy restatement of the reference: TensorFlow is
                  not installable here, so kind = "port") timed on a bounded sample.
Multi-GPU: sequences are independent -> batch sharded across ranks, weights
replicated, no data-path collective ("weak" scaling); one RCCL all-gather of the
14 metric accumulators per rank at the end.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy
FP32_MFMA_PEAK_TFLOPS = 157.3  # dense f32-input MFMA peak (= fp32 vector peak), MI355X_MICROARCH.md


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=1, help="sequences per GPU (configs[1]: 1, configs[2]: 32)")
    p.add_argument("--seq-len", type=int, default=4)
    p.add_argument("--height", type=int, default=384)
    p.add_argument("--width", type=int, default=1280)
    p.add_argument("--levels", type=int, default=6)
    p.add_argument("--dscv-range", type=int, default=4)
    p.add_argument("--sncv-range", type=int, default=3)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-timing", action="store_true")
    p.add_argument("--eager", action="store_true", help="do not replay the step from a hipGraph")
    p.add_argument("--in-flight", type=int, default=1,
                   help="independent sequence batches in flight (each on its own stream and model state); 1 = the quoted number")
    p.add_argument("--launch", default="graph", choices=["tasks", "graph"],
                   help="tasks: one hipGraph per (frame, level) task on one stream per frame; graph: one hipGraph per step")
    return p.parse_args()


class EventTimer:
    """Brackets chosen kernels with HIP events on torch's current stream (the stream
    the C-ABI launches on)."""

    def __init__(self, torch, level):
        self.torch = torch
        self.targets = {("dscv", level), ("sncv", level), ("conv", f"lvl{level}.conv1")}
        self.enabled = False
        self.events = {}

    def run(self, name, level, thunk):
        if not self.enabled or (name, level) not in self.targets:
            return thunk()
        e0 = self.torch.cuda.Event(enable_timing=True)
        e1 = self.torch.cuda.Event(enable_timing=True)
        e0.record()
        r = thunk()
        e1.record()
        self.events.setdefault(name, []).append((e0, e1))
        return r

    def summary(self):
        return {k: (len(v), float(np.mean([a.elapsed_time(b) for a, b in v])) * 1e-3) for k, v in self.events.items()}


def make_batch(args, rank, dev, torch):
    from m4depth_amd import synthetic as S
    uniq = min(args.batch, 2)
    samples, cam = S.make_sequence(uniq, args.seq_len, args.height, args.width, seed=1235 + 7919 * rank)
    reps = -(-args.batch // uniq)

    def tile(x):
        return np.concatenate([x] * reps, axis=0)[:args.batch]

    data = {}
    for key in ("depth", "RGB_im", "rot", "trans"):
        data[key] = torch.from_numpy(np.stack([tile(s[key]) for s in samples], axis=1)).to(dev)
    data["new_traj"] = torch.from_numpy(np.stack([tile(s["new_traj"]) for s in samples], axis=1))   # host: control flow
    data["camera"] = {k: torch.from_numpy(tile(v)).to(dev) for k, v in cam.items()}
    return data


def level_bytes(args, b, lvl=1):
    """Algorithmic HBM bytes of the two cost-volume kernels at pyramid level ``lvl``
    (each unique input read once, each output written once)."""
    from m4depth_amd.synthetic import ENCODER_CHANNELS, nbre_cuts_for
    h, w = args.height >> lvl, args.width >> lvl
    C = ENCODER_CHANNELS[lvl - 1]
    k = nbre_cuts_for(lvl)
    px = b * h * w
    ncp = 2 * args.dscv_range + 1
    mo = 2 * args.sncv_range + 1
    return {"dscv": 4 * px * (2 * C + 2 + ncp * k + 1),        # c1, c2, 2 parallax maps | cv, log feature
            "sncv": 4 * px * (C + mo * mo * k)}                 # c (c1 == c2) | cost volume


def cpu_baseline(args):
    """Oracle (numpy restatement, kind 'port') on a bounded sample of the same workload:
    ONE full frame (frame 1 of a reset+full pair), batch 1, same resolution / levels."""
    from threadpoolctl import threadpool_limits
    from oracle import m4depth_oracle as O
    from m4depth_amd import synthetic as S
    cores = min(os.cpu_count() or 1, 16)
    W = S.init_weights(args.levels, seed=42, dscv_range=args.dscv_range, sncv_range=args.sncv_range)
    samples, cam = S.make_sequence(1, 2, args.height, args.width, seed=1235)
    with threadpool_limits(limits=cores):
        model = O.M4Depth(W, args.levels, dscv_range=args.dscv_range, sncv_range=args.sncv_range)
        model(samples[:1], cam)
        t0 = time.perf_counter()
        out, _ = model(samples[1:], cam)
        dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 full frame (second frame of a 2-frame sequence), batch 1, {args.height}x{args.width}, "
                      f"{args.levels} levels, numpy float32 oracle; BLAS limited to {cores} threads, "
                      "elementwise numpy is single-threaded"}, (W, samples, cam, out)


def main():
    args = parse()
    import torch
    from m4depth_amd import dist as D
    rank, world, local_rank, dev = D.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    import m4depth_amd as M
    from m4depth_amd import network as net
    from m4depth_amd import synthetic as S

    torch.backends.cudnn.benchmark = True
    weights = S.init_weights(args.levels, seed=42, dscv_range=args.dscv_range, sncv_range=args.sncv_range)
    model = M.M4Depth(nbre_levels=args.levels, dscv_range=args.dscv_range, sncv_range=args.sncv_range)
    model.load_numpy_weights(weights, dev)
    model.compile(metrics=M.default_metrics())
    data = make_batch(args, rank, dev, torch)

    timer = EventTimer(torch, level=1)
    if not args.no_kernel_timing:
        net.kernel_timer = timer

    for _ in range(max(args.warmup, 1)):                 # eager warm-up: MIOpen solver search, state allocation
        model.test_step(data)
    runner = None
    replicas = [model]
    if not args.eager:
        runner = net.TaskGraphSequence(model, data) if args.launch == "tasks" and args.seq_len > 1 else net.GraphedSequence(model, data)
        step = lambda: model.graphed_test_step(data, runner)
        if args.in_flight > 1:
            # Sequence batches are independent (every one starts with new_traj): keep several in flight, each with its own
            # recurrent state, graph and stream, so that the latency-bound coarse levels at the head of one overlap with the
            # chip-filling level-1 convolutions at the tail of the previous one.  Weights are shared read-only.
            runners, streams = [runner], [torch.cuda.Stream() for _ in range(args.in_flight)]
            for _ in range(args.in_flight - 1):
                mr = M.M4Depth(nbre_levels=args.levels, dscv_range=args.dscv_range, sncv_range=args.sncv_range)
                mr.load_numpy_weights(weights, dev)
                mr.compile(metrics=M.default_metrics())
                mr.test_step(data)
                replicas.append(mr)
                runners.append(net.GraphedSequence(mr, data))
            counter = [0]

            def step():
                i = counter[0] % args.in_flight
                counter[0] += 1
                streams[i].wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(streams[i]):
                    replicas[i].graphed_test_step(data, runners[i])
        for _ in range(args.warmup):
            step()
    else:
        step = lambda: model.test_step(data)
    torch.cuda.synchronize()
    for mr in replicas:
        for m in mr.compiled_metrics:
            m.reset_state()
    D.barrier(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    D.barrier(dev)
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, dev)
    for mr in replicas[1:]:                                                   # fold the replicas' Keras-Mean accumulators together
        for m, m2 in zip(model.compiled_metrics, mr.compiled_metrics):
            if m2.total is not None:
                m.total = m2.total if m.total is None else m.total + m2.total
                m.count += m2.count
    gathered = D.all_gather_metric_states(model.compiled_metrics, dev)       # the one collective (RCCL)
    metrics = D.reduce_metric_states(gathered).tolist()

    # Per-kernel roofline: the SAME workload, eager launches bracketed by HIP events on the
    # launch stream (events cannot bracket nodes of a replayed graph), right after the timed region.
    if not args.no_kernel_timing:
        timer.enabled = True
        for _ in range(min(args.steps, 5)):
            model.test_step(data)
        torch.cuda.synchronize()
        timer.enabled = False

    if rank != 0:
        return

    frames = world * args.batch * args.seq_len * args.steps
    value = frames / dt
    out = {
        "metric": "frames/s", "value": round(value, 2), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "per_gpu": round(value / world, 2),
        "full_frames_per_s": round(value * (args.seq_len - 1) / args.seq_len, 2),
        "config": {"workload": f"{args.height}x{args.width} {args.levels}-level seq_len={args.seq_len} "
                               f"dscv_range={args.dscv_range} sncv_range={args.sncv_range} batch {args.batch}/GPU "
                               "(BASELINE.json configs[1] when batch=1, configs[2] when batch=32)",
                   "global_batch": world * args.batch, "seq_len": args.seq_len, "parallelism": f"dp{world}",
                   "weights": "random-init (He normal), seed 42", "frame0": "new_traj (state reset only)",
                   "conv_backend": "hand-written fp32-MFMA convolutions with fused bias+leaky-relu (libm4depth_hip.so): "
                                   "Winograd F(2x2,3x3) for the wide stride-1 layers of levels 1-3, direct implicit GEMM "
                                   "(stride 1 / 2, split-K on the coarse levels) elsewhere; encoder head (3->16 convolution + DINL) as two fused "
                                   "HIP kernels -- no MIOpen / framework kernel in the forward",
                   "hot_path": "libm4depth_hip.so (HIP, gfx950)"},
        "AbsRel": round(metrics[0], 6), "sequence_batches_in_flight": args.in_flight,
        "launch": "eager" if args.eager else "hipGraph replay of the sequence forward; frames pipelined over the decoder "
                                                  "levels on one HIP stream per frame (M4D_LEVEL_PIPELINE)",
    }
    if timer.events:
        summ = timer.summary()
        bytes_l1 = level_bytes(args, args.batch, 1)
        # HBM traffic per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950 +
        # WRITE_SIZE, KiB; profiles/r01_pmc_traffic.json), measured on the same kernels at the same
        # geometry; null when no measurement exists for this batch size.
        traffic = {}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            traffic = tj.get(f"batch{args.batch}", {})
        except Exception:
            pass
        # Dominant hand-written kernel of the step: the level-1 128->128 refiner convolution (fp32 MFMA
        # implicit GEMM; algorithmic flops = 2*9*Cin*Cout per output pixel) -> "roofline"; the two
        # level-1 cost-volume kernels (HBM-bound by bytes) -> "roofline_dscv" / "roofline_sncv".
        h1, w1 = args.height >> 1, args.width >> 1
        for name, (n, sec) in summ.items():
            if name == "conv":
                # The layer's algorithmic work is the direct convolution's 2*9*Cin*Cout flops per pixel.  The kernel that
                # runs it is Winograd F(2x2,3x3) (2.25x fewer multiply-adds, m4d_wino.hip) unless M4D_WINOGRAD=0, so the
                # contract's `achieved` (algorithmic flops / time) can exceed the peak; the flops the kernel actually executes on
                # the matrix cores (2*4*Cin*Cout per pixel, whole 2x2 tiles) and their fraction of the peak are reported next to it.
                flops_direct = 2.0 * 9 * 128 * 128 * h1 * w1 * args.batch
                wino = net.winograd_conv
                flops_exec = 2.0 * 16 * 128 * 128 * ((h1 + 1) // 2) * ((w1 + 1) // 2) * args.batch if wino else flops_direct
                tf = flops_direct / sec / 1e12                 # the contract's definition: ALGORITHMIC flops / time
                tf_exec = flops_exec / sec / 1e12
                out["roofline"] = {"kernel": ("conv3x3_wino4_kernel (level-1 refiner 128->128, Winograd F(2x2,3x3) on fp32 MFMA, "
                                              "bias+leaky-relu fused)") if wino else
                                             "conv3x3_mfma_kernel<4,3,1> (level-1 refiner 128->128, bias+leaky-relu fused)",
                                   "bound": "mfma", "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4),
                                   "traffic": traffic.get("wino_l1_128_128" if wino else "conv_l1_128_128"),
                                   "algorithmic_flops_per_launch": flops_direct,
                                   "note": ("achieved / frac use the layer's algorithmic (direct-convolution) flops, 2*9*Cin*Cout per "
                                            "pixel; the Winograd kernel executes 2.25x fewer on the matrix cores, so frac can exceed 1 -- "
                                            "executed_* is the matrix-core utilisation") if wino else "direct convolution",
                                   "executed_mfma_flops_per_launch": flops_exec,
                                   "executed_tflops": round(tf_exec, 2), "executed_frac": round(tf_exec / FP32_MFMA_PEAK_TFLOPS, 4),
                                   "algorithmic_bytes_per_launch": 4 * (2 * 128 * h1 * w1 * args.batch + 9 * 128 * 128),
                                   "avg_launch_us": round(sec * 1e6, 2), "launches": n}
                continue
            gbs = bytes_l1[name] / sec / 1e9
            out[f"roofline_{name}"] = {"kernel": f"{name}_level1", "bound": "hbm", "achieved": round(gbs, 1),
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                       "traffic": traffic.get(name), "algorithmic_bytes_per_launch": bytes_l1[name],
                                       "avg_launch_us": round(sec * 1e6, 2), "launches": n}
        if "roofline" not in out and "roofline_dscv" in out:            # hand-written convolutions disabled
            out["roofline"] = out["roofline_dscv"]
    if not args.no_cpu_baseline and world == 1:
        cb, (W, samples, cam, ref) = cpu_baseline(args)
        out["cpu_baseline"] = cb
        # parity of the same sample on the GPU: depth and AbsRel vs the oracle
        model.reset_state()

        def dv(x):
            if isinstance(x, dict):
                return {k: dv(v) for k, v in x.items()}
            if isinstance(x, list):
                return [dv(v) for v in x]
            return torch.from_numpy(x) if x.dtype == np.bool_ else torch.from_numpy(x).to(dev)

        got = model([dv(samples), dv(cam)])["depth"].cpu().numpy()
        from oracle import m4depth_oracle as O
        rel = np.abs(got - ref["depth"]) / np.maximum(np.abs(ref["depth"]), 1e-9)
        a_gpu = float(O.metrics_batch(samples[-1]["depth"], got)[0])
        a_ref = float(O.metrics_batch(samples[-1]["depth"], ref["depth"])[0])
        out["parity"] = {"depth_rel_median": float(np.median(rel)), "depth_within_1e-4": float(np.mean(rel < 1e-4)),
                         "AbsRel_gpu": a_gpu, "AbsRel_oracle": a_ref, "AbsRel_rel_diff": abs(a_gpu - a_ref) / a_ref}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
