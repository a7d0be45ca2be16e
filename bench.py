#!/usr/bin/env python
"""bench.py -- frames/s of the M4Depth per-frame inference path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 it
is launched under ``python -m torch.distributed.run`` with one rank per GPU -- and when it is
invoked PLAINLY with ``--gpus N`` > 1 (no WORLD_SIZE in the environment) it launches those N ranks
itself (``torchrun_command``); it never prints a line whose ``n_gpus`` differs from ``--gpus``.

One *step* = one pass of the hot path over one batch of synthetic input = the
reference's ``test_step`` on a 5-D sequence batch (m4depth_network.py:433-474):
encoder + 6-level parallax-cost-volume decoder over ``seq_len`` frames (frame 0
carries ``new_traj`` and only seeds the recurrent state, exactly as
dataloaders/generic.py:139 produces it) + the 7 depth metrics on the last frame.
Workload: 1 GPU -> BASELINE.json configs[1] (384x1280, 6 levels, seq_len 4, DSCV range 4 / SNCV range 3,
batch 1); N > 1 GPUs -> configs[3] (global batch 32 N: 32 sequences per rank, 256 on 8 GPUs); ``--batch`` overrides
(32 on one GPU = configs[2]).  Inputs are resident in HBM before the timed region.
``value`` = all frames processed by all ranks / max-over-ranks time.

The JSON line also carries
  roofline          -- the dominant kernel of the step, the level-1 refiner 128->128 convolution (Winograd F(2x2,3x3),
                       float32 operands split into three bf16 terms on the bf16 matrix cores): the MFMA flops the kernel
                       EXECUTES / HIP-event time on the launch stream, against the dense peak of the MFMA type it issues
                       (frac <= 1); the float32-equivalent and algorithmic (direct-convolution) rates are side fields;
  roofline_hotpath  -- SURVEY 8(d): the hot path's algorithmic bytes per full frame (106.72 MB at the default config) /
                       the summed time of the hand-written level kernels of one full frame / 8 TB/s;
  roofline_<kernel> -- the level-1 cost-volume kernels alone, algorithmic bytes / time against the 8 TB/s HBM3E peak;
  ``traffic``       -- HBM bytes per launch from rocprofv3 --pmc passes (tools/pmc_traffic.py), accepted only when the
                       recorded kernel name and the hash of the kernel's source files match this checkout, else null;
  cpu_baseline      -- the CPU oracle (a numpy restatement of the reference: TensorFlow is not installable here, so
                       kind = "port"), SURVEY 8(d) protocol: batch 1, T = 4, one warm-up + 3 repeats, median.
Multi-GPU: sequences are independent -> batch sharded across ranks, weights
replicated, no data-path collective ("weak" scaling); one RCCL all-gather of the
14 metric accumulators per rank at the end, one of the per-rank wall times for the report.
"""
import argparse
import hashlib
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
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (no sparsity), MI355X_MICROARCH.md
# What a bare register-only loop of v_mfma_f32_32x32x16_bf16 sustains on this pool with NON-trivial operand values, every CU
# busy (tools/micro/mfma_operand_variety.hip, profiles/r03_mfma_rate_vs_operand_values.txt: 20.5 ns per MFMA and SIMD on
# pseudo-random operands against 13.7-15.0 ns on all-ones -- the nominal rate is only reached on trivial data)
BF16_MFMA_SUSTAINED_TFLOPS = round(1024 * 32768 / 20.5e-9 / 1e12, 0)
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")
HOT_KERNELS = ("pre", "norm", "front", "dscv", "sncv", "dscv_sncv", "tail", "post", "resize")     # network._timed names of the hot path


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=None,
                   help="sequences per GPU; default 1 on one GPU (configs[1]), 32 per rank on several (configs[3]); 32 on one GPU = configs[2]")
    p.add_argument("--seq-len", type=int, default=4)
    p.add_argument("--height", type=int, default=384)
    p.add_argument("--width", type=int, default=1280)
    p.add_argument("--levels", type=int, default=6)
    p.add_argument("--dscv-range", type=int, default=4)
    p.add_argument("--sncv-range", type=int, default=3)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-timing", action="store_true")
    p.add_argument("--no-configs2", action="store_true", help="skip the batch-32 (BASELINE configs[2]) leg of the default run")
    p.add_argument("--configs2-steps", type=int, default=5)
    p.add_argument("--eager", action="store_true", help="do not replay the step from a hipGraph")
    p.add_argument("--host-input", action="store_true",
                   help="also report the PCIe-inclusive rate: the RGB / pose batch starts in pinned host memory every step (never `value`)")
    p.add_argument("--schedule", choices=["graph", "tape"], default=os.environ.get("M4D_BENCH_SCHEDULE", "graph"),
                   help="graph (default) = ONE hipGraph per step, the frames pipelined on the graph's internal streams; tape = the same "
                        "(frame, level) wavefront as launch tapes of libm4depth_hip.so replayed as plain stream launches (no hipGraph; "
                        "measured 3 %% slower, kept as the graph-free launcher)")
    p.add_argument("--repeat-regions", type=int, default=3,
                   help="timed regions of --steps steps run back to back; the first is the line's value, all are in ms_per_step_runs")
    p.add_argument("--in-flight", type=int, default=1,
                   help="independent sequence batches in flight (each on its own stream and model state); 1 = the quoted number")
    return p.parse_args()


class EventTimer:
    """Brackets the hand-written kernels with HIP events on torch's current stream (the stream
    the C-ABI launches on).  Keys: (name, level)."""

    def __init__(self, torch):
        self.torch = torch
        self.enabled = False
        self.events = {}

    def run(self, name, level, thunk):
        if not self.enabled or not (name in HOT_KERNELS or (name == "conv" and level == "lvl1.conv1")):
            return thunk()
        e0 = self.torch.cuda.Event(enable_timing=True)
        e1 = self.torch.cuda.Event(enable_timing=True)
        e0.record()
        r = thunk()
        e1.record()
        self.events.setdefault((name, level), []).append((e0, e1))
        return r

    def summary(self):
        """{(name, level): (launches, mean seconds)} -- the RAW mean over every launch (the "average launch duration" of the
        roofline contract).  ``self.stats[key]`` carries the median, the mean without launches slower than twice the median
        (a box hiccup: one such launch doubled a batch-32 average in round 3) and how many those were, for every group."""
        out, self.stats, self.outliers = {}, {}, {}
        for k, v in self.events.items():
            d = np.array([a.elapsed_time(b) for a, b in v], np.float64)
            keep = d <= 2.0 * np.median(d)
            self.outliers[k] = int((~keep).sum())
            self.stats[k] = {"mean_us": round(float(d.mean()) * 1e3, 2), "median_us": round(float(np.median(d)) * 1e3, 2),
                             "trimmed_mean_us": round(float(d[keep].mean()) * 1e3, 2), "launches": int(d.size),
                             "slower_than_2x_median": int((~keep).sum())}
            out[k] = (int(d.size), float(d.mean()) * 1e-3)
        return out

    def reset(self):
        self.events = {}


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


def level_geometry(args, lvl):
    from m4depth_amd.synthetic import ENCODER_CHANNELS, nbre_cuts_for, f_input_channels
    h, w = args.height >> lvl, args.width >> lvl
    C = ENCODER_CHANNELS[lvl - 1]
    k = nbre_cuts_for(lvl)
    return h, w, C, k, f_input_channels(k, args.dscv_range, args.sncv_range)


def level_bytes(args, b, lvl=1):
    """Algorithmic HBM bytes of the two cost-volume kernels at pyramid level ``lvl``
    (each unique input read once, each output written once)."""
    h, w, C, k, _ = level_geometry(args, lvl)
    px = b * h * w
    ncp = 2 * args.dscv_range + 1
    mo = 2 * args.sncv_range + 1
    return {"dscv": 4 * px * (2 * C + 2 + ncp * k + 1),        # c1, c2, 2 parallax maps | cv, log feature
            "sncv": 4 * px * (C + mo * mo * k),                 # c (c1 == c2) | cost volume
            # the fused level front (normalise + level_pre + DSCV + SNCV): raw features, previous frame's features, depth
            # memory, the coarser level's parallax + other maps (5 floats per 4 pixels) | normalised features (the new
            # state), the whole refiner-input row
            "front": int(px * (4 * (3 * C + 1 + (ncp * k + mo * mo * k + 6)) + 5))}


def hotpath_bytes_per_frame(args, b):
    """per full frame sum_l 4 h w (3C + F_in + 13) + 6 h w [l < L], + the final nearest x2 (4 (H W + H W / 4))."""
    total = 0
    for lvl in range(1, args.levels + 1):
        h, w, C, k, f_in = level_geometry(args, lvl)
        total += 4 * h * w * (3 * C + f_in + 13) + (6 * h * w if lvl < args.levels else 0)
    return b * total, b * 4 * (args.height * args.width + (args.height // 2) * (args.width // 2))


def _sha(paths):
    hsh = hashlib.sha256()
    for p in paths:
        with open(os.path.join(ROOT, p), "rb") as fh:
            hsh.update(fh.read())
    return hsh.hexdigest()[:16]


def load_traffic(batch, prefix=""):
    """{name: {"bytes": n, "kernel": "..."}} of profiles/pmc_traffic.json for this batch size (``prefix`` "config4_": the
    BASELINE configs[4] geometry), only entries whose kernel sources are unchanged since the counters were collected
    (tools/pmc_traffic.py stamps a hash of them); + a note."""
    try:
        tj = json.load(open(TRAFFIC_JSON))
    except Exception:
        return {}, "no profiles/pmc_traffic.json"
    out, stale = {}, []
    for name, ent in tj.get(f"{prefix}batch{batch}", {}).items():
        try:
            fresh = _sha(ent["sources"]) == ent["sources_sha"]
        except Exception:
            fresh = False
        if fresh:
            out[name] = ent
        else:
            stale.append(name)
    note = f"rocprofv3 --pmc (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), tools/pmc_traffic.py, collected {tj.get('collected', '?')}"
    if stale:
        note += f"; REFUSED as stale (kernel sources changed since): {', '.join(stale)}"
    return out, note


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(height, width, levels, rd, rs, seq_len=4, repeats=3, seed=1235, keep=False):
    """SURVEY 8(d) protocol on the numpy oracle (kind 'port'): batch 1, T = seq_len frames (frame 0 = new_traj), one
    warm-up run + ``repeats`` timed runs, median; BLAS limited to ``cores`` threads (elementwise numpy is single-threaded)."""
    from threadpoolctl import threadpool_limits
    from oracle import m4depth_oracle as O
    from m4depth_amd import synthetic as S
    cores = min(os.cpu_count() or 1, 16)
    W = S.init_weights(levels, seed=42, dscv_range=rd, sncv_range=rs)
    samples, cam = S.make_sequence(1, seq_len, height, width, seed=seed)
    times, out = [], None
    with threadpool_limits(limits=cores):
        for i in range(1 + repeats):
            model = O.M4Depth(W, levels, dscv_range=rd, sncv_range=rs)
            t0 = time.perf_counter()
            out, seq = model(samples, cam)
            if i > 0:
                times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    res = {"value": round(seq_len / med, 4), "unit": "frames/s", "cores": cores, "kind": "port",
           "full_frames_per_s": round((seq_len - 1) / med, 4), "cpu": cpu_model_string(), "host_cores": os.cpu_count(),
           "seconds_per_sequence": [round(t, 3) for t in times],
           "sample": f"one {seq_len}-frame sequence (frame 0 = new_traj), batch 1, {height}x{width}, {levels} levels, "
                     f"dscv_range={rd} sncv_range={rs}; numpy float32 oracle (a restatement, not TensorFlow); 1 warm-up + "
                     f"{repeats} timed runs, median; BLAS limited to {cores} threads, elementwise numpy single-threaded"}
    return res, ((W, samples, cam, out, seq) if keep else None)


def kernel_rooflines(args, batch, timer, net, eager_steps):
    """The roofline objects of one batch size from the HIP-event brackets of ``eager_steps`` eager steps: the dominant kernel
    (level-1 128 -> 128 refiner convolution) against the MFMA peak, the level-1 cost-volume kernels and the whole hand-written
    hot path of a full frame against the HBM peak."""
    rep = {}
    summ = timer.summary()
    if (args.height, args.width, args.levels, args.dscv_range, args.sncv_range) == (384, 1280, 6, 4, 3):
        traffic, traffic_note = load_traffic(batch)
    elif (args.height, args.width, args.levels, args.dscv_range, args.sncv_range) == (768, 2560, 6, 6, 6):
        traffic, traffic_note = load_traffic(batch, "config4_")          # BASELINE configs[4]: tools/pmc_traffic.py --config4
    else:                                   # the PMC passes are collected at the level-1 geometry of these two pyramids only
        traffic, traffic_note = {}, "no PMC traffic for this geometry (tools/pmc_traffic.py measures the 384x1280 / ranges 4,3 and the 768x2560 / ranges 6,6 pyramids)"
    rep["traffic_note"] = traffic_note

    def tr(name):
        ent = traffic.get(name)
        return None if ent is None else ent["bytes"]

    # -- the dominant kernel of the step: the level-1 128->128 refiner convolution
    conv = summ.get(("conv", "lvl1.conv1"))
    h1, w1 = args.height >> 1, args.width >> 1
    if conv is not None:
        n, sec = conv
        wino = net._use_winograd(batch, h1, w1, 128, 128, 1)
        flops_direct = 2.0 * 9 * 128 * 128 * h1 * w1 * batch
        flops_wino = 2.0 * 16 * 128 * 128 * ((h1 + 1) // 2) * ((w1 + 1) // 2) * batch      # one multiply-add per (position, cin, cout)
        if wino == 6:           # float32 operands as 3 bf16 terms each: 6 bf16 MFMA products per float32 multiply
            flops_exec, peak, key = 6.0 * flops_wino, BF16_MFMA_PEAK_TFLOPS, "wino6_l1_128_128"
            units = batch * ((h1 + 15) // 16) * ((w1 + 15) // 16) * 2
            from m4depth_amd._lib import lib as _mlib
            persistent = net.wino6_kernel == 2 or (net.wino6_kernel == 0 and units >= int(_mlib.m4d_wino6_persistent_min_units()))
            kname = (("conv3x3_wino6p_kernel (persistent workgroups, csrc/m4d_wino6p.hip; " if persistent else
                      "conv3x3_wino6_kernel (one workgroup per (tile, 64 couts), csrc/m4d_wino6.hip; ") +
                     "level-1 refiner 128->128, Winograd F(2x2,3x3), float32 operands split into 3 bf16 "
                     "terms, 6 bf16 MFMA products each, float32 accumulate; bias+leaky-relu fused)")
        elif wino:
            flops_exec, peak, key = flops_wino, FP32_MFMA_PEAK_TFLOPS, "wino_l1_128_128"
            kname = "conv3x3_wino4_kernel (level-1 refiner 128->128, Winograd F(2x2,3x3) on fp32 MFMA, bias+leaky-relu fused)"
        else:
            flops_exec, peak, key = flops_direct, FP32_MFMA_PEAK_TFLOPS, "conv_l1_128_128"
            kname = "conv3x3_mfma_kernel<4,3,1> (level-1 refiner 128->128, bias+leaky-relu fused)"
        tf_exec = flops_exec / sec / 1e12
        rep["roofline"] = {
            "kernel": kname, "bound": "mfma", "achieved": round(tf_exec, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(tf_exec / peak, 4), "traffic": tr(key),
            "pmc_mfma_busy_frac": None if key not in traffic else traffic[key].get("mfma_busy_frac"),
            "traffic_kernel": None if key not in traffic else traffic[key].get("kernel"),
            "executed_mfma_flops_per_launch": flops_exec,
            "frac_of_sustained_mfma_rate": round(tf_exec / BF16_MFMA_SUSTAINED_TFLOPS, 4) if wino == 6 else None,
            "sustained_mfma_tflops_on_random_operands": BF16_MFMA_SUSTAINED_TFLOPS if wino == 6 else None,
            "note": "achieved / frac = MFMA flops the kernel executes / time against the dense peak of the MFMA type it issues "
                    "(bf16 2500 TFLOP/s for the split kernel: 6 bf16 products per float32 multiply-add of the Winograd form, "
                    "2*16*Cin*Cout per 2x2 output tile; fp32 MFMA 157.3 otherwise).  frac_of_sustained_mfma_rate = the same against what "
                    "a bare MFMA loop sustains on real operand values on this pool (0.66 of nominal, "
                    "profiles/r03_mfma_rate_vs_operand_values.txt).  Timing ablations of the kernel (DESIGN.md section 6): MFMAs + "
                    "row-transform reads + prologue / epilogue 73 % of the launch, LDS-DMA traffic 13 %, the exact 3-way operand "
                    "split + input transform 8.5 %, fragment reads 5 %; float32_equivalent_tflops = the float32 multiply-adds of "
                    "the Winograd form it replaces / time (fp32-MFMA peak 157.3), algorithmic_* = the layer's "
                    "direct-convolution flops / time",
            "float32_equivalent_tflops": round(flops_wino / sec / 1e12, 2) if wino else round(tf_exec, 2),
            "float32_equivalent_frac_of_fp32_mfma_peak": round((flops_wino if wino else flops_direct) / sec / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
            "algorithmic_flops_per_launch": flops_direct,
            "algorithmic_tflops": round(flops_direct / sec / 1e12, 2),
            "algorithmic_bytes_per_launch": 4 * (2 * 128 * h1 * w1 * batch + 9 * 128 * 128),
            "avg_launch_us": round(sec * 1e6, 2), "launches": n,
            "launch_time_stats": timer.stats.get(("conv", "lvl1.conv1"))}
    # -- the level-1 cost-volume kernels alone
    bytes_l1 = level_bytes(args, batch, 1)
    for name in ("front", "dscv", "sncv"):
        ent = summ.get((name, 1))
        if ent is None:
            continue
        n, sec = ent
        gbs = bytes_l1[name] / sec / 1e9
        rep[f"roofline_{name}"] = {"kernel": f"{name}_level1", "bound": "hbm", "achieved": round(gbs, 1),
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                   "traffic": tr(name),
                                   "traffic_kernel": None if name not in traffic else traffic[name].get("kernel"),
                                   "algorithmic_bytes_per_launch": bytes_l1[name],
                                   "avg_launch_us": round(sec * 1e6, 2), "launches": n,
                                   "launch_time_stats": timer.stats.get((name, 1))}
    # -- the whole hand-written hot path of one full frame against the HBM roofline
    full_frames = eager_steps * (args.seq_len - 1)
    per_kernel = {}
    for (name, lvl), (n, sec) in summ.items():
        if name in HOT_KERNELS and name != "resize" and lvl != "reset":
            per_kernel[name] = per_kernel.get(name, 0.0) + n * sec / max(full_frames, 1)
    if per_kernel:
        hp_bytes, resize_bytes = hotpath_bytes_per_frame(args, batch)
        t_all = sum(per_kernel.values())
        t_no_tail = sum(v for k, v in per_kernel.items() if k != "tail")
        gbs = hp_bytes / t_all / 1e9
        rep["roofline_hotpath"] = {
            "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_frame": hp_bytes, "us_per_frame": round(t_all * 1e6, 1),
            "us_per_frame_by_kernel": {k: round(v * 1e6, 1) for k, v in sorted(per_kernel.items())},
            "us_per_launch_by_kernel_and_level": {f"{k[0]}.{k[1]}": round(v[1] * 1e6, 1) for k, v in sorted(summ.items(), key=lambda kv: str(kv[0]))
                                                  if k[0] in HOT_KERNELS},
            "frac_excluding_tail": round(hp_bytes / t_no_tail / 1e9 / HBM_PEAK_GBS, 4) if t_no_tail > 0 else None,
            "note": "SURVEY 8(d) bytes of one full frame (all levels; x batch) / the summed HIP-event time of the hand-written "
                    "level kernels of that frame (level_pre + normalise, DSCV, SNCV, refiner tail) / 8 TB/s.  The tail kernel "
                    "also contains the last two refiner convolutions (32->16, 16->5), so `frac` is a lower bound for the "
                    "path proper; frac_excluding_tail drops that kernel's time but keeps its level_post bytes.  The "
                    "kernels timed are the ones the graph replays (levels <= 6000 pixels: DSCV and SNCV as ONE launch, dscv_sncv)"}
    if "roofline" not in rep and "roofline_hotpath" in rep:
        rep["roofline"] = rep["roofline_hotpath"]
    return rep


def eager_timed_steps(model, data, timer, batch, steps, torch):
    """``steps`` eager test_steps with the kernel timer on, each queued behind a GPU-side spin (see main)."""
    timer.enabled = True
    spin_cycles = int(60e6 * max(1, batch) ** 0.5)
    # one pass that is NOT kept: with the timer on the forward takes its single-stream, separately-launched form, whose first call
    # allocates scratch and output buffers (a host stall of milliseconds between an event and the launch it brackets would be read
    # as kernel time: one 58-ms "launch" among the six of a batch-32 leg, round 5)
    model.test_step(data)
    torch.cuda.synchronize()
    timer.reset()
    for _ in range(steps):
        torch.cuda._sleep(spin_cycles)
        model.test_step(data)
        torch.cuda.synchronize()
    timer.enabled = False
    return steps


def configs2_leg(args, weights, dev, timer, torch, M, net, D):
    """BASELINE configs[2] ("384x1280 6-level seq_len=4, batch 32, 1xMI355X (HBM-roofline run)") inside the default run: the
    same test_step at batch 32 -- own model (own recurrent state), hipGraph replay, ``--configs2-steps`` timed replays between
    synchronisations, then two eager steps under the kernel timer for the roofline fractions.  Returns the ``configs2``
    object of the JSON line (same definitions as the top-level keys)."""
    import argparse
    a2 = argparse.Namespace(**vars(args))
    a2.batch, a2.steps = 32, args.configs2_steps
    model = M.M4Depth(nbre_levels=args.levels, dscv_range=args.dscv_range, sncv_range=args.sncv_range)
    model.load_numpy_weights(weights, dev)
    model.compile(metrics=M.default_metrics())
    data = make_batch(a2, 0, dev, torch)
    was = net.kernel_timer
    net.kernel_timer = None
    model.test_step(data)                                # eager warm-up: state and scratch allocation
    runner = net.GraphedSequence(model, data)
    data.update({k: v for k, v in runner.input_buffers().items()})
    step = lambda: model.graphed_test_step(data, runner)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    dt, _ = timed_region(step, a2.steps, D, dev, torch.cuda.synchronize)
    frames = a2.batch * a2.seq_len * a2.steps
    rep = {"workload": "BASELINE configs[2]: 384x1280, 6 levels, seq_len 4 (frame 0 = new_traj), batch 32, one GPU, hipGraph replay",
           "value": round(frames / dt, 2), "unit": "frames/s", "steps": a2.steps, "ms_per_step": round(dt / a2.steps * 1e3, 3),
           "full_frames_per_s": round(frames * (a2.seq_len - 1) / a2.seq_len / dt, 2)}
    if was is not None:
        net.kernel_timer = timer
        timer.reset()
        n = eager_timed_steps(model, data, timer, a2.batch, 2, torch)
        rep.update(kernel_rooflines(a2, a2.batch, timer, net, n))
    net.kernel_timer = was
    return rep


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def torchrun_command(gpus, argv, port=None):
    """The command line that runs this script as ``gpus`` ranks of one node (one process per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()), os.path.abspath(__file__)] + list(argv)


def check_world(args, environ, device_count):
    """What a process started as ``bench.py --gpus N`` has to do: "run" (its world size is N), or "relaunch" (N > 1 and it
    was invoked plainly: no WORLD_SIZE -> start the N ranks under torch.distributed.run).  Anything else is refused: a line
    whose n_gpus is not the N that was asked for is worse than no line."""
    if args.gpus < 1:
        raise SystemExit(f"--gpus {args.gpus}: need at least one GPU")
    world = environ.get("WORLD_SIZE")
    if world is None:
        if args.gpus == 1:
            return "run"
        if device_count < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {device_count} GPU(s) visible on this node")
        return "relaunch"
    if int(world) != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: refusing to report a line whose "
                         f"n_gpus differs from --gpus (start it with --nproc-per-node {args.gpus}, or plainly)")
    return "run"


def timed_region(step, steps, D, dev, sync):
    """EXACTLY ``steps`` calls of ``step`` bracketed by a barrier + device synchronisation on both sides.  Returns
    (max-over-ranks wall time of the region including the closing barrier, the per-rank wall times without it)."""
    D.barrier(dev)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    dt_local = time.perf_counter() - t0
    D.barrier(dev)
    dt = time.perf_counter() - t0
    dt = D.max_over_ranks(dt, dev)
    per_rank_s = D.all_gather_floats(dt_local, dev)                          # RCCL all-gather (report only)
    return dt, per_rank_s


def workload_name(args, world):
    """Which BASELINE.json configuration this run IS -- decided from the whole geometry, not from (GPUs, batch) alone (round 5
    labelled the 768x2560 ranges-6/6 line configs[1])."""
    geo = (args.height, args.width, args.levels, args.seq_len, args.dscv_range, args.sncv_range)
    if geo == (384, 1280, 6, 4, 4, 3):
        if (world, args.batch) == (1, 1):
            return "BASELINE.json configs[1]"
        if (world, args.batch) == (1, 32):
            return "BASELINE.json configs[2]"
        if world > 1 and args.batch == 32:
            return "BASELINE.json configs[3] (32 sequences per rank)"
        return "the configs[1] geometry at a custom batch"
    if geo[:3] == (768, 2560, 6) and geo[4] == 6 and (world, args.batch) == (1, 1):
        return "BASELINE.json configs[4]" + ("" if geo[3] == 4 and geo[5] == 6 else " (custom seq_len / SNCV range)")
    if geo[:3] == (128, 256, 3) and geo[4] == 2 and (world, args.batch) == (1, 1):
        return "BASELINE.json configs[0] geometry"
    return "custom workload"


def run_spread(ms_runs):
    """min / median / max of the repeated timed regions' ms per step (the first one is the line's ``ms_per_step``)."""
    v = sorted(float(x) for x in ms_runs)
    return {"runs": [round(float(x), 3) for x in ms_runs], "min": round(v[0], 3), "median": round(float(np.median(v)), 3),
            "max": round(v[-1], 3), "spread_pct": round(100.0 * (v[-1] - v[0]) / v[0], 2) if v[0] > 0 else None}


def box_kind_probe(stagger_autotune_ms, chosen_us):
    """The capture-time timings of the lock-step and the staggered graph (GraphedSequence._capture_autotuned) as ONE record: the
    boxes of the pool come in two kinds -- on the 'lock-step-slow' kind the unstaggered graph runs ~4.5 % slower than the staggered
    one, on the 'lock-step-fast' kind ~3 % faster and the whole step ~7 % faster -- so a reader can tell a
    box difference from a code change.  None when the graph was captured once."""
    if not stagger_autotune_ms:
        return None
    lock = stagger_autotune_ms.get(0)
    stag = next((v for k, v in stagger_autotune_ms.items() if k), None)
    if not lock or not stag:
        return None
    ratio = lock[-1] / stag[-1]
    return {"lock_step_ms_per_step": lock, "staggered_ms_per_step": stag, "lock_step_over_staggered": round(ratio, 4),
            "kind": "lock-step-slow (staggered graph kept)" if ratio > 1.0 else "lock-step-fast (lock-step graph kept)",
            "chosen_stagger_us": chosen_us}


def timed_job(step, args, D, dev, sync, stagger_us=None):
    """The timed part of a run, identical on every rank (every call inside is a collective or rank-local; nothing waits for
    rank 0's report work): the contract's timed region -- ``value`` comes from THIS one --, then the same region
    ``--repeat-regions`` - 1 more times back to back in the same process (the boxes of the pool differ by up
    to 7 % and one 50-ms region cannot separate a 3 % change from run-to-run noise), then one small gather of every rank's
    Winograd first-round choice (the graph a rank replays depends on ITS capture-time timing at batch <= 4; -1 = not applicable).
    Returns (dt, per_rank_s, ms_per_step of every region, per-rank stagger us)."""
    dt, per_rank_s = timed_region(step, args.steps, D, dev, sync)
    ms_runs = [1e3 * dt / args.steps]
    for _ in range(max(int(getattr(args, "repeat_regions", 1)) - 1, 0)):
        dt_r, _ = timed_region(step, args.steps, D, dev, sync)
        ms_runs.append(1e3 * dt_r / args.steps)
    per_rank_stagger = D.all_gather_floats(-1.0 if stagger_us is None else float(stagger_us), dev)
    return dt, per_rank_s, ms_runs, per_rank_stagger


def report_head(args, world, dt, per_rank_s):
    """The contract fields of the JSON line: whole-job frames/s = frames of ALL ranks / max-over-ranks time."""
    if world != args.gpus or len(per_rank_s) != world:
        raise SystemExit(f"bench.py: world size {world} / {len(per_rank_s)} gathered ranks, but --gpus {args.gpus}")
    frames = world * args.batch * args.seq_len * args.steps
    value = frames / dt
    cfg_name = workload_name(args, world)
    return {
        "metric": "frames/s", "value": round(value, 2), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic", "per_gpu": round(value / world, 2),
        "full_frames_per_s": round(value * (args.seq_len - 1) / args.seq_len, 2),
        "per_rank_frames_per_s": [round(args.batch * args.seq_len * args.steps / s, 2) for s in per_rank_s],
        "config": {"workload": f"{args.height}x{args.width} {args.levels}-level seq_len={args.seq_len} "
                               f"dscv_range={args.dscv_range} sncv_range={args.sncv_range} batch {args.batch}/GPU = {cfg_name}",
                   "global_batch": world * args.batch, "seq_len": args.seq_len, "parallelism": f"dp{world}"}}


def main():
    args = parse()
    import torch
    if check_world(args, os.environ, torch.cuda.device_count()) == "relaunch":
        cmd = torchrun_command(args.gpus, sys.argv[1:])
        print(f"[bench] --gpus {args.gpus} invoked without a launcher: starting the ranks with {' '.join(cmd[1:10])} ...",
              file=sys.stderr, flush=True)
        os.execv(cmd[0], cmd)
    from m4depth_amd import dist as D
    rank, world, local_rank, dev = D.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if args.batch is None:
        args.batch = 1 if world == 1 else 32                 # configs[1] / configs[3] (32 sequences per rank)
    import m4depth_amd as M
    from m4depth_amd import network as net
    from m4depth_amd import synthetic as S
    from m4depth_amd import _lib

    weights = S.init_weights(args.levels, seed=42, dscv_range=args.dscv_range, sncv_range=args.sncv_range)
    model = M.M4Depth(nbre_levels=args.levels, dscv_range=args.dscv_range, sncv_range=args.sncv_range)
    model.load_numpy_weights(weights, dev)
    model.compile(metrics=M.default_metrics())
    data = make_batch(args, rank, dev, torch)

    timer = EventTimer(torch)
    if not args.no_kernel_timing:
        net.kernel_timer = timer

    for _ in range(max(args.warmup, 1)):                 # eager warm-up: state and scratch allocation
        model.test_step(data)
    runner = None
    extra_warmup = 0
    launches_per_step = None
    replicas = [model]
    if not args.eager:
        if args.schedule == "tape":
            runner = net.TapedSequence(model, data)
        else:
            n0 = int(_lib.lib.m4d_launch_count())
            runner = net.make_runner(model, data, warmup=1)
            launches_per_step = (int(_lib.lib.m4d_launch_count()) - n0) // (1 + runner.capture_passes)   # one eager warm-up pass + the capture pass(es)
        # the batch lives in the graph's own input buffers (inputs resident in HBM before the timed region: no hand-over copy)
        data.update({k: v for k, v in runner.input_buffers().items()})
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
                runners.append(net.GraphedSequence(mr, data))         # (replicas copy the batch into their own buffers per step)
            counter = [0]

            def step():
                i = counter[0] % args.in_flight
                counter[0] += 1
                streams[i].wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(streams[i]):
                    replicas[i].graphed_test_step(data, runners[i])
        for _ in range(args.warmup):
            step()
        # The GPU idled while the host captured and instantiated the graph (tens of ms): an idle MI355X drops its clocks and the
        # first replays after it run ~1.5 % slow (tools/step_jitter.py --idle-ms 200).  Keep replaying, UNTIMED, until the device
        # has been busy for a fifth of a second; the count is reported (`extra_warmup_steps`), the timed region below is
        # untouched: exactly --steps steps between barriers.
        torch.cuda.synchronize()
        t_busy = time.perf_counter()
        while time.perf_counter() - t_busy < 0.2 and extra_warmup < 200:
            step()
            torch.cuda.synchronize()
            extra_warmup += 1
    else:
        step = lambda: model.test_step(data)
    torch.cuda.synchronize()
    for mr in replicas:
        for m in mr.compiled_metrics:
            m.reset_state()
    stag = getattr(runner, "stagger_us", None) if runner is not None else None
    dt, per_rank_s, ms_runs, per_rank_stagger = timed_job(step, args, D, dev, torch.cuda.synchronize, stag)
    for mr in replicas[1:]:                                                   # fold the replicas' Keras-Mean accumulators together
        for m, m2 in zip(model.compiled_metrics, mr.compiled_metrics):
            if m2.total is not None:
                m.total = m2.total if m.total is None else m.total + m2.total
                m.count += m2.count
    gathered = D.all_gather_metric_states(model.compiled_metrics, dev)       # the one data collective (RCCL): 14 floats / rank
    metrics = D.reduce_metric_states(gathered).tolist()

    # PCIe-inclusive rate (never `value`): the batch's network inputs start in pinned host memory every step
    host_rate = None
    if args.host_input and runner is not None:
        pinned = {k: data[k].cpu().pin_memory() for k in ("RGB_im", "rot", "trans")}
        pinned["camera"] = {k: v.cpu().pin_memory() for k, v in data["camera"].items()}
        pinned["new_traj"] = data["new_traj"]
        pinned["depth"] = data["depth"]
        for _ in range(2):
            model.graphed_test_step(pinned, runner)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            model.graphed_test_step(pinned, runner)
        torch.cuda.synchronize()
        host_rate = args.batch * args.seq_len * args.steps / (time.perf_counter() - t1)

    # Per-kernel timing: the SAME workload, eager launches bracketed by HIP events on the
    # launch stream (events cannot bracket nodes of a replayed graph), right after the timed region.
    # Every timed eager step is queued behind a GPU-side spin (torch.cuda._sleep): the host enqueues the step's launches and
    # event records while the GPU spins, so an event pair brackets the kernel's execution, not the host's launch latency
    # (without it the small kernels of the coarse levels read 3-10x too long: the GPU runs ahead of the eager host loop).
    n_eager = 0
    if not args.no_kernel_timing and rank == 0:          # (report work of rank 0 alone: the other ranks are done after the gathers)
        n_eager = eager_timed_steps(model, data, timer, args.batch, min(args.steps, 5), torch)

    roof1 = kernel_rooflines(args, args.batch, timer, net, n_eager) if (timer.events and rank == 0) else {}
    # BASELINE configs[2] beside the default configs[1] line (one GPU, default workload only): the same step at batch 32
    configs2 = None
    if (world == 1 and args.batch == 1 and not args.no_configs2 and not args.eager and args.in_flight == 1 and args.schedule == "graph"
            and (args.height, args.width, args.levels, args.seq_len) == (384, 1280, 6, 4)):
        configs2 = configs2_leg(args, weights, dev, timer, torch, M, net, D)

    if rank != 0:
        return

    out = report_head(args, world, dt, per_rank_s)
    out.update({
        "dtype_note": "every tensor, operand and accumulator is float32 (the reference's float16 DSCV products excepted, as in "
                      "the reference).  The wide Winograd convolutions feed the bf16 matrix cores with float32 operands split "
                      "EXACTLY into three bf16 terms (6 of the 9 term products, float32 accumulation): float32 accuracy -- "
                      "measured error against float64 0.8x that of the fp32-MFMA kernels (profiles/r02_bf16_split_probe.txt, "
                      "tools/bench_wino6.py; parity.vs_float64_oracle below) -- not a reduced-precision path; "
                      "M4D_CONV_ARITH=f32 runs the fp32-MFMA kernels instead",
        "metric_note": "value counts every frame of the sequence the reference's test_step processes, including frame 0, "
                       "which carries new_traj and only runs the encoder + state reset; full_frames_per_s excludes it.  "
                       "north_star's target (>= 500 frames/s/GPU at 6 levels) is compared with full_frames_per_s -- the "
                       "stricter reading: frames that run the whole cost-volume path",
        "AbsRel": round(metrics[0], 6), "sequence_batches_in_flight": args.in_flight, "extra_warmup_steps": extra_warmup,
        "launch": "eager" if args.eager else ("launch tapes (csrc/m4d_tape.hip) replayed as plain stream launches, one HIP stream per frame"
                                                  if args.schedule == "tape" else
                                                  "hipGraph replay of the sequence forward; frames pipelined over the decoder "
                                                  "levels on one HIP stream per frame (M4D_LEVEL_PIPELINE)"),
    })
    out["ms_per_step_runs"] = run_spread(ms_runs)
    out["per_rank_winograd_stagger_us"] = [int(v) for v in per_rank_stagger]
    tuned = getattr(runner, "stagger_autotune_ms", None) if not args.eager else None
    probe = box_kind_probe(tuned, getattr(runner, "stagger_us", None))
    out["box_kind_probe"] = probe
    if tuned is not None:
        out["winograd_first_round"] = {
            "chosen": f"staggered over {runner.stagger_us} us" if runner.stagger_us else "lock step (no stagger)",
            "ms_per_step_at_capture": {("staggered" if k else "lock_step"): v for k, v in tuned.items()},
            "note": "GraphedSequence captures the sequence with and without the staggered first round of the one-per-CU Winograd "
                    "kernel (a per-launch argument of m4d_conv3x3_wino6_bias_act_ks; same bits) and keeps the faster graph -- "
                    "before the warm-up, outside the timed region: which one wins depends on the box (DESIGN.md section 6)"}
    out["config"].update({
        "weights": "random-init (He normal), seed 42", "frame0": "new_traj (state reset only)",
        "kernels": "every kernel of the captured forward is hand-written HIP (libm4depth_hip.so, gfx950): MFMA "
                   "convolutions with fused bias+leaky-relu (Winograd F(2x2,3x3) on the wide stride-1 layers -- float32 "
                   "operands as exact 3 x bf16 splits on the bf16 matrix cores, or fp32 MFMA --, direct fp32-MFMA "
                   "implicit GEMM elsewhere), fused encoder head / refiner tail, the level kernels; no MIOpen, rocBLAS "
                   "or PyTorch kernel in the graph (torch only launches the 7-metric kernel's host wrapper eagerly)",
        "hot_path": "libm4depth_hip.so (HIP, gfx950)"})
    if host_rate is not None:
        out["host_input_frames_per_s"] = round(host_rate, 2)
    out.update(roof1)
    if configs2 is not None:
        out["configs2"] = configs2
    # The driver's record keeps the `roofline` dict of the line, not the extra top-level keys: the north-star numbers (the HBM
    # fractions of the hand-written cost-volume path, BASELINE configs[2]) are repeated INSIDE it as plain numbers.
    if isinstance(out.get("roofline"), dict):
        rf, hp, fr = out["roofline"], out.get("roofline_hotpath") or {}, out.get("roofline_front") or {}
        c2 = configs2 or {}
        rf["hotpath_hbm_frac"] = hp.get("frac")
        rf["hotpath_us_per_frame"] = hp.get("us_per_frame")
        rf["front_hbm_frac"] = fr.get("frac")
        rf["front_avg_launch_us"] = fr.get("avg_launch_us")
        rf["front_traffic_ratio"] = (round(fr["traffic"] / fr["algorithmic_bytes_per_launch"], 4)
                                     if fr.get("traffic") and fr.get("algorithmic_bytes_per_launch") else None)
        rf["traffic_ratio"] = (round(rf["traffic"] / rf["algorithmic_bytes_per_launch"], 4)
                               if rf.get("traffic") and rf.get("algorithmic_bytes_per_launch") else None)
        rf["configs2_frames_per_s"] = c2.get("value")
        rf["configs2_frac"] = (c2.get("roofline") or {}).get("frac")
        rf["configs2_front_hbm_frac"] = (c2.get("roofline_front") or {}).get("frac")
        rf["configs2_hotpath_hbm_frac"] = (c2.get("roofline_hotpath") or {}).get("frac")
        rf["launches_per_step"] = launches_per_step
        rf["ms_per_step_runs"] = out["ms_per_step_runs"]
        rf["box_kind_probe"] = probe
    if not args.no_cpu_baseline and world == 1:
        cb, (W, samples, cam, ref, ref_seq) = cpu_baseline(args.height, args.width, args.levels, args.dscv_range, args.sncv_range,
                                                           seq_len=args.seq_len, keep=True)
        out["cpu_baseline"] = cb
        cb1, _ = cpu_baseline(128, 256, 3, 2, 2, seq_len=4, seed=1234)
        cb1["sample"] = "BASELINE.json configs[0] geometry (128x256, 3 levels, ranges 2/2): " + cb1["sample"]
        out["cpu_baseline_config1"] = cb1
        # parity of the same sequence on the GPU: depth and AbsRel vs the float32 oracle, and both float32 evaluations
        # against the float64 evaluation of the oracle (the rounding-noise floor of the 1e-4 tolerance)
        model.reset_state()

        def dv(x):
            if isinstance(x, dict):
                return {k: dv(v) for k, v in x.items()}
            if isinstance(x, list):
                return [dv(v) for v in x]
            return torch.from_numpy(x) if x.dtype == np.bool_ else torch.from_numpy(x).to(dev)

        got = model([dv(samples), dv(cam)])["depth"].cpu().numpy()
        from oracle import m4depth_oracle as O
        with O.float64_reference():
            truth, _ = O.M4Depth(W, args.levels, dscv_range=args.dscv_range, sncv_range=args.sncv_range)(samples, cam)
        rel = np.abs(got - ref["depth"]) / np.maximum(np.abs(ref["depth"]), 1e-9)
        rel_g64 = np.abs(got - truth["depth"]) / np.maximum(np.abs(truth["depth"]), 1e-9)
        rel_o64 = np.abs(ref["depth"] - truth["depth"]) / np.maximum(np.abs(truth["depth"]), 1e-9)
        a_gpu = float(O.metrics_batch(samples[-1]["depth"], got)[0])
        a_ref = float(O.metrics_batch(samples[-1]["depth"], ref["depth"])[0])
        out["parity"] = {"sample": f"the cpu_baseline sequence ({args.seq_len} frames), last frame's depth",
                         "depth_rel_median": float(np.median(rel)), "depth_within_1e-4": float(np.mean(rel < 1e-4)),
                         "AbsRel_gpu": a_gpu, "AbsRel_oracle": a_ref, "AbsRel_rel_diff": abs(a_gpu - a_ref) / a_ref,
                         "vs_float64_oracle": {"gpu_depth_within_1e-4": float(np.mean(rel_g64 < 1e-4)),
                                               "oracle_f32_depth_within_1e-4": float(np.mean(rel_o64 < 1e-4)),
                                               "gpu_depth_rel_median": float(np.median(rel_g64)),
                                               "oracle_f32_depth_rel_median": float(np.median(rel_o64))}}
        # the well-conditioned fixture (tests/golden/model_wc_full.npz: one 384x1280 / 6-level frame pair, golden depths of
        # the float32 oracle): the north-star tolerance on EVERY pixel
        try:
            g = dict(np.load(os.path.join(ROOT, "tests", "golden", "model_wc_full.npz")))
            L, rd, rs, H, Wd, T, b, seed = [int(v) for v in g["meta"]]
            Wc, s_wc, cam_wc = S.well_conditioned_case(L, b, T, H, Wd, seed, rd, rs)
            m_wc = M.M4Depth(nbre_levels=L, dscv_range=rd, sncv_range=rs)
            m_wc.load_numpy_weights(Wc, dev)
            m_wc([dv(s_wc), dv(cam_wc)])
            worst = 0.0
            for l in range(L):
                d = m_wc.last_estimates[-1][l]["depth"].cpu().numpy()
                worst = max(worst, float(np.max(np.abs(d - g[f"l{l}_depth"]) / np.abs(g[f"l{l}_depth"]))))
            out["parity"]["well_conditioned_fixture"] = {
                "sample": f"tests/golden/model_wc_full.npz: {H}x{Wd}, {L} levels, one reset + one full frame, last refiner layer x "
                          f"{S.WELL_CONDITIONED_GAIN}, lateral motion (m4depth_amd.synthetic.well_conditioned_case)",
                "depth_rel_max_over_all_levels": worst, "depth_within_1e-4": float(worst < 1e-4),
                "oracle_f32_vs_float64_rel_max": float(g["f32_vs_f64_max_rel_depth"])}
        except FileNotFoundError:
            pass
    print(json.dumps(out))


if __name__ == "__main__":
    main()
