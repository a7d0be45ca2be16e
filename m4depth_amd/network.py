"""M4Depth network, inference path -- same layer names and call signatures as
the reference ``m4depth_network.py``.

Host code is Python on PyTorch-ROCm (modules, control flow, device memory, streams, hipGraph capture); every kernel
the inference forward launches on the GPU is hand-written HIP from libm4depth_hip.so: the encoder (``FeaturePyramid``)
and decoder (``DispRefiner``) 3x3 convolutions with TF ``SAME`` padding (fp32-MFMA implicit GEMM / Winograd, fused
bias + leaky_relu, fused encoder head and refiner tail), and everything inside ``DepthEstimatorLevel`` -- per-cut
normalisation, x2 upsampling of the coarser estimate, prev_d2para, the DSCV and SNCV cost volumes, the log-parallax
features, exp/clip and parallax2depth -- writing straight into the refiner's input tensor.  No MIOpen / rocBLAS kernel
is involved.  Activations are NHWC float32 (``[b,h,w,c]``), exactly the reference's layout.
"""
from __future__ import annotations

import ctypes
from collections import namedtuple

import numpy as np
import torch

from . import network_ops as nops
from ._lib import lib, dptr, stream_ptr, check, as_f32
from .synthetic import ENCODER_CHANNELS, REFINER_CHANNELS, f_input_channels, nbre_cuts_for

# m4depth_network.py:21-22
M4depthAblationParameters = namedtuple('M4depthAblationParameters',
                                       ('DINL', 'SNCV', 'time_recurr', 'normalize_features',
                                        'subdivide_features', 'level_memory'),
                                       defaults=(True, True, True, True, True, True))

_CV_ACCUM = {"fp32_round": 0, "fp16_seq": 1}

import os as _os

# bench.py installs an object with ``run(name, level, thunk)`` here to bracket the
# hand-written kernels with HIP events on the launch stream; None = no overhead.
kernel_timer = None
# tools/determinism_stress.py installs ``tap(name, level, tensor)`` here to keep copies of intermediate activations (the
# encoder maps and the refiner layers, which nothing else retains); None = no overhead.
debug_tap = None
# Winograd F(2x2,3x3) for the wide stride-1 layers (csrc/m4d_wino.hip): 2.25x fewer MFMA flops, equal to the direct
# convolution up to float32 rounding (max difference ~1e-6 of the output range).  Measured per layer (tools/bench_wino.py):
# 1.26-1.49x at batch 1, 1.16-1.39x at batch 8 for 64/128 output channels; the 96- and 32-wide layers (one N-tile per
# workgroup) gain only on small grids.  0 = always the direct convolution.
winograd_conv = _os.environ.get("M4D_WINOGRAD", "1") == "1"


# Smallest grid (workgroups of Winograd kernel 2) for which a layer leaves the direct convolution.  Round 1 used 200 (a
# launch that fills the chip on its own); inside the frame pipeline the chip is shared with other frames' kernels anyway and
# the 2.25x fewer MFMA flops win from 60 workgroups on: level 3 of the 384x1280 pyramid (30 tiles x 2-4 N-tiles) moves to
# Winograd, +2 % frames/s at batch 1 (1050 -> 1074; 30: 1047).
winograd2_min_workgroups = int(_os.environ.get("M4D_WINO2_MIN_WG", "60"))


# Arithmetic of the wide Winograd layers: "bf16x3" = float32 operands split exactly into three bf16 terms, six bf16 MFMA
# products, float32 accumulation (csrc/m4d_wino6.hip: float32 accuracy -- error against float64 0.8x that of the fp32
# MFMA kernels, tools/bench_wino6.py -- at 2.67x less matrix-core time); "f32" = the fp32-MFMA kernels everywhere.
conv_arith = _os.environ.get("M4D_CONV_ARITH", "bf16x3")
# Smallest grid ((tile, 64-cout) units) those kernels take: 30 = the narrow layers of level 3 (30 tiles: 96 -> 64, and 64 -> 32 as
# half units) as well, +0.8 % frames/s at batch 1 against 40 (rounds 2-4: those two on the fp32-MFMA Winograd / direct kernels);
# 20 (level 4's wide layers too) measured the same as 30
wino6_min_workgroups = int(_os.environ.get("M4D_WINO6_MIN_WG", "30"))
# Which kernel serves those layers (an ARGUMENT of m4d_conv3x3_wino6_bias_act_k, same bits either way): 0 = the library chooses
# from the grid (persistent workgroups, csrc/m4d_wino6p.hip, wherever a CU gets more than one (tile, 64-cout) unit), 1 = one
# workgroup per unit always (csrc/m4d_wino6.hip), 2 = persistent always.  A/B timing only.
wino6_kernel = int(_os.environ.get("M4D_WINO6_KERNEL", "0"))
# Round 6: launches of wino6_pair_min_units ... (persistent threshold - 1) units at a launch batch <= 4 (the level-1 refiner layers
# at batch 1: 960 / 480 units) on persistent workgroups of wino6_pair_units CONSECUTIVE units each (kernel = 16 + n of
# m4d_conv3x3_wino6_bias_act_ks): a tile's cout groups share a halo fetch and the DMA stream runs through the unit boundary
# (4.4 us instead of ~7 for prologue + epilogue), while the dispatcher still places the workgroups as CUs free (the static
# one-range-per-CU form did not pay inside the frame pipeline).  0 = off (one workgroup per unit).  Same bits.
wino6_pair_min_units = int(_os.environ.get("M4D_WINO6_PAIR_MIN_UNITS", "0"))
wino6_pair_units = int(_os.environ.get("M4D_WINO6_PAIR_UNITS", "2"))
# Narrowest layer the bf16-split Winograd kernels serve: a cout group of <= 32 channels runs as a HALF unit (one N-tile), so the
# refiner's 64 -> 32 layer is one half unit per tile: 35.7 -> 29.0 us on level 1, 20.8 -> 12.4 on level 2 against the fp32-MFMA
# Winograd kernel that served it in rounds 2-4 (= 64 here), +0.9 % / +1.1 % frames/s at batch 1 / 32
# (profiles/r04_wino6_half_units.txt)
wino6_min_cout = int(_os.environ.get("M4D_WINO6_MIN_COUT", "32"))
# (experiments build only, include/m4depth_hip_experiments.h) which kernel serves the bf16-split layers -- same bits either
# way: 2 = the wide kernel m4d_wino6w.hip wherever it applies, 3 = the half-tile kernel m4d_wino6h.hip
if _os.environ.get("M4D_WINO6_VARIANT") or _os.environ.get("M4D_WINO6_HALF_MAX_WG"):
    from ._lib import require_experiments as _require_experiments
    _require_experiments("M4D_WINO6_VARIANT / M4D_WINO6_HALF_MAX_WG")
    if _os.environ.get("M4D_WINO6_VARIANT"):
        lib.m4d_wino6_set_variant(int(_os.environ["M4D_WINO6_VARIANT"]))
    if _os.environ.get("M4D_WINO6_HALF_MAX_WG"):       # grids up to this many workgroups take the half-tile kernel (default 0 = none)
        lib.m4d_wino6_set_half_tile_max_workgroups(int(_os.environ["M4D_WINO6_HALF_MAX_WG"]))
# The one-launch small-map convolution in the same arithmetic (csrc/m4d_conv.hip conv3x3_small6_kernel).  These launches are
# bound by how fast ONE CU streams its slice of the weights, not by the matrix core: -2 us per layer on the 240-channel first
# layers, nothing elsewhere (tools/bench_small_convs.py); +0.8 % frames/s at batch 1.  0 = the fp32-MFMA small-map kernel.
_small_conv_split_env = _os.environ.get("M4D_SMALL_CONV_SPLIT", "1") == "1"
small_conv_split = _small_conv_split_env and conv_arith == "bf16x3"


# The fused level tail (conv 32->16, conv 16->5, level_post) in the same bf16-split arithmetic (csrc/m4d_tail6.hip) when
# conv_arith is "bf16x3"; 0 = the fp32-MFMA tail (csrc/m4d_tail.hip) always.
tail_split = _os.environ.get("M4D_TAIL_SPLIT", "1") == "1"

# The encoder's convolutions choose their kernel from the per-image grid x the sequence batch (1 = from the per-image grid alone:
# rounds 1-3, which ran the coarse encoder levels of a batch-32 evaluation on the single-small-map latency kernels).
encoder_batch_dispatch = _os.environ.get("M4D_ENCODER_BATCH_DISPATCH", "1") == "1"

import contextlib as _contextlib


@_contextlib.contextmanager
def conv_arithmetic(arith):
    """Run the enclosed calls with ``conv_arith`` = "bf16x3" or "f32" (what ``M4D_CONV_ARITH`` selects at import) and
    everything derived from it switched consistently; the previous setting is restored on exit.  Not thread-safe."""
    global conv_arith, small_conv_split
    if arith not in ("bf16x3", "f32"):
        raise ValueError(f"conv arithmetic must be 'bf16x3' or 'f32', not {arith!r}")
    old = (conv_arith, small_conv_split)
    conv_arith, small_conv_split = arith, _small_conv_split_env and arith == "bf16x3"
    try:
        yield
    finally:
        conv_arith, small_conv_split = old


def _use_winograd(b, h, w, cin, cout, stride):
    """0 = direct convolution; 6 = the bf16-split Winograd kernel (m4d_wino6.hip); 1 = Winograd kernel 1 (16x8 tile,
    16-channel chunks, 2 N-tiles per workgroup); 2 = kernel 2 (16x16 tile, 8-channel chunks: half the weight traffic per flop -- ahead on the large maps).
    Winograd only pays when the launch fills the chip: the thresholds are workgroup counts (measured per layer of the
    384x1280 pyramid, profiles/r01_step_launches_b1_wino.txt: level 3 and coarser stay on the direct kernel, which can
    split N and K further)."""
    if not winograd_conv or stride != 1 or cin < 16 or cin % 2 != 0:
        return 0
    n32 = -(-cout // 32)
    t16 = b * (-(-h // 16)) * (-(-w // 16))
    if conv_arith == "bf16x3" and cin % 16 == 0 and cin >= 32 and cout >= wino6_min_cout and t16 * (-(-cout // 64)) >= wino6_min_workgroups:
        return 6                                                # 64-cout units; a last group of <= 32 channels runs one N-tile
    if cin % 4 == 0 and t16 * n32 >= winograd2_min_workgroups:
        return 2
    t8 = b * (-(-h // 8)) * (-(-w // 16))
    if cout % 64 == 0 and t8 * (cout // 64) >= 400:
        return 1
    if cout >= 96 and 400 <= t8 * n32 <= 3000:                 # one N-tile per workgroup: only ahead on small grids
        return 1
    return 0


# Refiner input padded to a multiple of 8 channels (levels 2-6: 58k + 6 = 122, 238, 470): the padding channels are
# zero in a persistent buffer and meet zero weights, so every sum is unchanged bit for bit, and the first refiner layer
# of level 2 becomes eligible for the 8-channel-chunk Winograd kernels (122 -> 128).  0 = exact-width input.
pad_refiner_input = _os.environ.get("M4D_PAD_REFINER_INPUT", "1") == "1"

# DispRefiner layers on maps of at most this many pixels (batch included) and at most 256 input channels run as ONE
# launch with the K split inside the workgroup (m4d_conv3x3_small_bias_act) instead of split-K + reduce: levels 4-6 at
# batch 1, +2.5 % frames/s (tools/ab_bench.sh).  Refiner layers only: their batch is the caller's batch in every launch
# mode, so the pipelined and the single-stream forward keep choosing the same kernel (bitwise-neutrality test).  0 = off.
small_map_conv_pixels = int(_os.environ.get("M4D_CONV_SMALL_PX", "2048"))
small_map_stride2 = _os.environ.get("M4D_CONV_SMALL_S2", "1") == "1"     # the coarse stride-2 encoder layers on the one-launch kernel too

# The latency-first small-map convolution (csrc/m4d_convlat.hip, round 5): every layer that may take the one-launch small-map
# kernels (``small_maps_ok``, stride 1, Cin >= 16, Cin % 4 == 0) on maps of at most this many pixels (batch included) when
# conv_arith is "bf16x3".  K is split over waves and workgroups until a wave owns one or two 16-channel chunks and requests
# everything it needs in one memory round trip; the K slices of different workgroups reach the NEXT layer as partial-sum slabs
# (network_ops.PartialAct) that it adds while staging.  Levels 4-6 at batch 1: 59 / 50 / 52 us of refiner convolutions per
# level -> 38 / 37 / 48 (tools/bench_lat_convs.py).  0 = the round-2/3 small-map kernels.
lat_conv_max_pixels = int(_os.environ.get("M4D_LAT_CONV_PX", "2048"))
# ... and the narrow layers (Cout <= 32: one cout group, the Winograd kernels run them as half units on 30-120 workgroups) up to
# this many pixels: level 3's 64 -> 32 layer at batch 1 (48x160)
lat_conv_narrow_max_pixels = int(_os.environ.get("M4D_LAT_CONV_NARROW_PX", "0"))
# ... and the stride-2 encoder layers whose OUTPUT has at most this many pixels (sequence batch included)
lat_conv_s2_max_pixels = int(_os.environ.get("M4D_LAT_CONV_S2_PX", "0"))
# ... and the narrow short-K stride-2 encoder layers (Cin, Cout <= 64: 32 -> 32 at 192x640 and 64 -> 64 at 96x320 of the 384x1280
# pyramid, fp32-MFMA direct kernels of ~25 us each at batch 1) in the kernel's M-over-waves form, up to this many OUTPUT pixels
lat_conv_mw_max_pixels = int(_os.environ.get("M4D_LAT_CONV_MW_PX", "0"))

# DSCV and SNCV of a small level (<= 6000 pixels) in one launch (m4d_dscv_sncv_fwd).  0 = two launches.
fused_cost_volumes = _os.environ.get("M4D_FUSED_COST_VOLUMES", "1") == "1"
# ... and the level's opening glue (level_pre) in that launch too (m4d_level_front_small, round 6): a coarse level opens with ONE
# launch.  Its features arrive per-cut normalised: the normalisation depends on the encoder alone, so the coarse levels' maps of all
# frames of an encoder batch are normalised in one launch right behind the encoder (m4d_normalize_levels), off the levels'
# coarse-to-fine latency chains.  0 = level_pre_normalize + dscv_sncv (two dependent launches per level).  Same bits.
small_level_front = _os.environ.get("M4D_SMALL_LEVEL_FRONT", "1") == "1"
# level_pre and the per-cut normalisation of a level in one launch (m4d_level_pre_normalize).  0 = two launches.
fused_level_front = _os.environ.get("M4D_FUSED_LEVEL_FRONT", "1") == "1"

# The whole level front (normalise + level_pre + DSCV + SNCV -> complete refiner-input rows) as ONE kernel
# (csrc/m4d_front.hip) wherever it has an instantiation: the default settings and 16 channels x 1-2 cuts or 32 x 2 (levels 1-3
# of the 6-level pyramid); maps of at most fused_front_min_pixels pixels (batch included) and every other configuration keep
# the separate kernels.  M4D_FUSED_FRONT=0 = the separate kernels everywhere (same bits).
fused_front = _os.environ.get("M4D_FUSED_FRONT", "1") == "1"
fused_front_min_pixels = int(_os.environ.get("M4D_FUSED_FRONT_MIN_PX", "0"))
# The geometries of levels 4-6 (24 / 32 channels per cut x 4 cuts, 24 x 8) have instantiations too (round 3); on maps of at
# most this many pixels (batch included: levels 4-6 at batch 1-3) the one-launch small-map cost-volume kernel stays -- a
# handful of 8x8 tiles cannot fill the chip -- above it (batch >= 4) the fused front writes whole refiner-input rows.
fused_front_coarse_min_pixels = int(_os.environ.get("M4D_FUSED_FRONT_COARSE_MIN_PX", "6000"))

# Encoder level 0 as two fused kernels (direct 3->16 convolution + bias + DINL statistics; DINL apply fused into the
# stride-2 convolution's input staging) instead of MIOpen conv + bias pass + 3 DINL passes + conv: no MIOpen kernel is
# left in the inference path.  0 = the unfused sequence.
fused_encoder_head = _os.environ.get("M4D_FUSED_ENCODER_HEAD", "1") == "1"

# The last two refiner convolutions (32->16, 16->5) and the level tail in ONE kernel (csrc/m4d_tail.hip).  0 = three launches.
fused_refiner_tail = _os.environ.get("M4D_FUSED_TAIL", "1") == "1"

# Frame pipeline of the decoder: level l of frame t+1 depends on level l+1 of its own frame and on
# level l of frame t only, so consecutive frames of a sequence run on two HIP streams as a
# wavefront: the launch-latency-bound coarse levels of frame t+1 execute underneath the
# chip-filling level-1/2 convolutions of frame t.  One stream per frame (up to this many), so that
# dependencies only ever flow from stream t to stream t+1: ROCm 7.2's hipStreamEndCapture segfaults on a
# capture in which two streams wait on each other's events alternately (tools/debug_multistream_capture.py).
# 0 disables (single stream).
level_pipeline_streams = int(_os.environ.get("M4D_LEVEL_PIPELINE", "8"))
# Redundant (transitively implied) cross-stream waits of the pipelined forward.  They change how ROCm's hipGraph executor lays the
# captured graph out on its streams, so they are knobs, measured (profiles/r04_graph_executor.txt):
#   M4D_PIPE_FORK_FRAMES = frames (beside frame 0 and the late encoder's first frame) whose stream waits for the fork event
#   explicitly: "all" (rounds 1-3) or a list like "1,2"; M4D_PIPE_SKIP_ENC_WAIT=1: a frame of the late encoder batch does not
#   wait for that batch's event when the previous frame (same batch) is waited for anyway.
_pf = _os.environ.get("M4D_PIPE_FORK_FRAMES", "1,2")
pipeline_fork_frames = None if _pf == "all" else {int(v) for v in _pf.split(",") if v.strip() != ""}
pipeline_skip_implied_encoder_wait = _os.environ.get("M4D_PIPE_SKIP_ENC_WAIT", "1") == "1"
# A reset frame (new_traj: six state-seeding launches of a few microseconds) shares the stream of the frame that follows it: the
# first full frame's coarse-to-fine chain then has no cross-stream wait in front of every level (round 5: each such wait is a
# ~4.7 us gap in the executor's queue, profiles/r05_queue_trace_b1_graph.txt, on a chain that runs with the chip otherwise idle).
pipeline_merge_reset_frame = _os.environ.get("M4D_PIPE_MERGE_RESET", "1") == "1"
# The staggered first round of the one-per-CU Winograd kernel (csrc/m4d_wino6.hip): an ARGUMENT of every launch
# (m4d_conv3x3_wino6_bias_act_ks, ABI 6 -- no library state).  The value is a per-MODEL setting (M4Depth.set_wino6_stagger ->
# every layer's ``wino6_stagger_us``); a layer that was never told falls back to this module default.  Which launches carry it:
# grids of >= wino6_stagger_min_wg workgroups at a launch batch <= wino6_stagger_max_batch (the launches that put one workgroup on
# every CU at batch 1-4: the level-1 / level-2 refiner layers and the encoder's 240-workgroup layers).  Whether GraphedSequence
# captures the sequence both ways and keeps the faster graph (the effect's sign depends on the box): wino6_stagger_autotune.
wino6_stagger_us = int(_os.environ.get("M4D_WINO6_STAGGER_US", "9"))
wino6_stagger_phases = int(_os.environ.get("M4D_WINO6_STAGGER_PHASES", "16"))
wino6_stagger_min_wg = int(_os.environ.get("M4D_WINO6_STAGGER_MIN_WG", "200"))
wino6_stagger_max_batch = int(_os.environ.get("M4D_WINO6_STAGGER_MAX_BATCH", "4"))
wino6_stagger_autotune = _os.environ.get("M4D_STAGGER_AUTOTUNE", "1") == "1"
# The sequence step as four hipGraphs on two real streams (SegmentedSequence) where the sequence allows it: make_runner().
segmented_step = _os.environ.get("M4D_SEGMENTED", "0") == "1"
segmented_resumed_first = _os.environ.get("M4D_SEG_RESUMED_FIRST", "1") == "1"
segmented_resumed_on_main = _os.environ.get("M4D_SEG_RESUMED_ON_MAIN", "0") == "1"
wino6_stagger_force = _os.environ.get("M4D_STAGGER_FORCE", "")           # "staggered" / "lock_step": keep that graph whatever the timing
# The reset frame of all levels in ONE launch (m4d_pyramid_reset) instead of one state-seeding launch per level: the six launches
# depend on each other only through the upsampling of constant maps, and they sit on the critical path of a batch-1 step.
fused_pyramid_reset = _os.environ.get("M4D_FUSED_RESET", "1") == "1"
# Encoding every frame on its own stream too (instead of one encoder pass batched over the frames, before the
# decoder) was measured slightly slower (633 vs 643 frames/s at batch 1): the batched pass has 4x fewer launches.
# Encoder level 0 (conv 3->16, DINL, conv 16->16 stride 2) as m4d_enc_level0_fwd: three passes that recompute the first
# convolution from the RGB frames instead of writing / re-reading its full-resolution 16-channel output.  0 = the two-call head.
fused_encoder_level0 = _os.environ.get("M4D_FUSED_ENC0", "1") == "1"
pipeline_encoder_per_frame = _os.environ.get("M4D_PIPELINE_ENCODER", "0") == "1"
# Encoder in two batches instead: frames [0, split) before the decoder, frames [split, T) on frame `split`'s stream.
pipeline_encoder_split = int(_os.environ.get("M4D_PIPELINE_ENCODER_SPLIT", "2"))
# With the split: level 0's DINL statistics of every frame are taken in the first batch's launches (per-image arithmetic, same bits).
encoder_stats_up_front = _os.environ.get("M4D_ENC_STATS_UP_FRONT", "1") == "1"
# A/B knob (rounds 5's behaviour): a stride-1 encoder layer on the latency-first kernel always hands partial slabs on, also when
# its stride-2 consumer cannot add them (then m4d_partial_finish runs as a launch of its own).
encoder_s1_partial_always = _os.environ.get("M4D_ENC_S1_PARTIAL", "0") == "1"


def _stack_frames(samples):
    """The RGB frames of ``samples`` as ONE encoder batch, frame-major (image t*b + i).  When the frames are views of one
    [b,T,H,W,3] sequence tensor (what test_step's unstacking produces, m4depth_network.py:439-447) the result is a
    ``FrameStack`` the fused encoder head reads in place; otherwise they are concatenated."""
    ims = [s['RGB_im'] for s in samples]
    x0 = ims[0]
    if (x0.is_cuda and x0.dtype == torch.float32 and x0.dim() == 4 and x0[0].is_contiguous()
            and all(x.dtype == x0.dtype and x.shape == x0.shape and x.stride() == x0.stride()
                    and x.untyped_storage().data_ptr() == x0.untyped_storage().data_ptr() for x in ims)):
        if len(ims) == 1:
            return nops.FrameStack(x0.unsqueeze(1))
        st = ims[1].storage_offset() - x0.storage_offset()
        if st > 0 and all(ims[t].storage_offset() - x0.storage_offset() == t * st for t in range(len(ims))):
            b, h, w, c = x0.shape
            return nops.FrameStack(torch.as_strided(x0, (b, len(ims), h, w, c), (x0.stride(0), st, w * c, c, 1),
                                                    x0.storage_offset()))
    return torch.cat([as_f32(x, "RGB_im") for x in ims], dim=0)


def _timed(name, level, thunk):
    kt = kernel_timer
    return thunk() if kt is None else kt.run(name, level, thunk)


def _to_device_f32(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float32)
    return torch.as_tensor(np.asarray(x), dtype=torch.float32, device=device)


def _stamp(*params):
    """Identity + in-place version of parameters: changes after ``optimizer.step()``, ``load_state_dict`` or any other
    in-place update, so packed copies keyed on it can never go stale silently (training.py keys its caches the same way)."""
    return tuple((p.data_ptr(), p._version, str(p.device)) for p in params)


class _PackCache:
    """Packed / transformed copies of a layer's weights, each remembered with the ``_stamp`` of the parameters it was
    built from.  A stale entry is rebuilt INTO THE SAME DEVICE TENSOR (``copy_``), so a hipGraph that captured the
    packed tensor's address sees the new weights at its next replay."""

    def __init__(self):
        self.entries = {}

    def get(self, key, stamp, build):
        hit = self.entries.get(key)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("packed convolution weights are missing or stale inside a hipGraph capture: call "
                               "M4Depth.prepack() before capturing")
        fresh = build()
        if hit is not None:
            old = hit[1]
            olds, news = (old if isinstance(old, tuple) else (old,)), (fresh if isinstance(fresh, tuple) else (fresh,))
            if len(olds) == len(news) and all(isinstance(o, torch.Tensor) == isinstance(n, torch.Tensor) and (
                    not isinstance(o, torch.Tensor) or (o.shape == n.shape and o.device == n.device)) for o, n in zip(olds, news)):
                for o, n in zip(olds, news):
                    if isinstance(o, torch.Tensor):
                        o.copy_(n)
                fresh = old
        self.entries[key] = (stamp, fresh)
        return fresh

    def clear(self):
        self.entries.clear()


class _Conv3x3SameTF(torch.nn.Module):
    """Keras Conv2D(filters, 3, strides, padding='same') on NHWC tensors.

    TF 'SAME': out = ceil(in/stride), total pad = max((out-1)*stride + 3 - in, 0),
    pad_before = total // 2 -- so stride 2 on an even size pads bottom/right only,
    which torch's symmetric ``padding=1`` does not reproduce.

    On the GPU every call is one hand-written HIP kernel (m4d_conv.hip / m4d_wino.hip) with the bias add and the
    leaky_relu fused; the padding rule lives inside the kernels."""

    def __init__(self, out_channels, stride, in_channels=None):
        super().__init__()
        self.out_channels = out_channels
        self.stride = stride
        self.weight = None
        self.bias = None
        self._cache = _PackCache()
        self.tag = None                  # e.g. "lvl1.conv1": lets bench.py bracket one layer with HIP events
        self.small_maps_ok = False       # DispRefiner layers: may take the one-launch small-map kernel
        self.per_image_dispatch = False  # encoder layers: kernel choice from the per-image grid x dispatch_batch (see forward)
        self.dispatch_batch = 1
        self.wino6_stagger_us = None     # this layer's Winograd launches: staggered first round (us); None = the module default
        if in_channels is not None:
            self._build(in_channels, None)

    def _build(self, in_channels, device):
        w = torch.empty(self.out_channels, in_channels, 3, 3, device=device)
        torch.nn.init.kaiming_normal_(w, nonlinearity='relu')       # ks.initializers.HeNormal (:61,:100)
        self.weight = torch.nn.Parameter(w.contiguous(memory_format=torch.channels_last), requires_grad=False)
        self.bias = torch.nn.Parameter(torch.zeros(self.out_channels, device=device), requires_grad=False)
        self._cache.clear()

    def load_hwio(self, kernel, bias, device):
        """Load a TF-layout [3,3,Cin,Cout] kernel.  When the layer already holds parameters of the same shape on the same
        device they are overwritten IN PLACE (``copy_``: same addresses, version bumped), so the packed copies are rebuilt
        into their existing tensors and a hipGraph captured before the load (``GraphedSequence``) replays the new weights;
        otherwise the parameters are replaced and any earlier capture is invalid (``GraphedSequence`` detects the changed
        addresses and raises)."""
        w = _to_device_f32(kernel, device).permute(3, 2, 0, 1)
        b = _to_device_f32(bias, device).contiguous()
        if (self.weight is not None and self.bias is not None and self.weight.shape == w.shape and self.bias.shape == b.shape
                and self.weight.device == w.device):
            with torch.no_grad():
                self.weight.copy_(w)
                self.bias.copy_(b)
            return
        self.weight = torch.nn.Parameter(w.contiguous(memory_format=torch.channels_last), requires_grad=False)
        self.bias = torch.nn.Parameter(b, requires_grad=False)
        self._cache.clear()

    def invalidate_packed(self):
        """Mark every packed copy stale.  Needed only after a write that bypasses autograd's version counter
        (``param.data.copy_(...)``, raw pointer writes): weights must otherwise be updated in place ON THE PARAMETER
        (``copy_`` / ``load_state_dict`` / an optimizer step), which the version-keyed caches follow by themselves."""
        with torch.no_grad():
            if self.weight is not None:
                self.weight.add_(0)                     # bumps ._version: every cache keyed on _stamp() rebuilds in place

    def _hwio_numpy(self, cin_pad=None):
        """The TF-layout kernel on the host; ``cin_pad`` > Cin appends zero input channels (for a zero-padded input)."""
        hwio = self.weight.detach().permute(2, 3, 1, 0).cpu().numpy()
        if cin_pad is not None and cin_pad > hwio.shape[2]:
            hwio = np.concatenate([hwio, np.zeros(hwio.shape[:2] + (cin_pad - hwio.shape[2], hwio.shape[3]), hwio.dtype)], axis=2)
        return hwio

    def _packed_weights(self, cin_pad=None):
        """(wp, CoutPad) for m4d_conv3x3_bias_act, packed from the live OIHW parameter (re-packed when it changes)."""
        if cin_pad is not None and cin_pad == self.weight.shape[1]:
            cin_pad = None

        def build():
            wp, cpad = nops.pack_conv_weights(self._hwio_numpy(cin_pad))
            return torch.from_numpy(wp).to(self.weight.device), cpad
        return self._cache.get(("direct", cin_pad), _stamp(self.weight), build)

    def _packed_weights_winograd(self, chunk=16, cin_pad=None):
        """(wu, CoutPad) for m4d_conv3x3_wino_bias_act (chunk 16) / wino2 (chunk 8): U = G g G^T, transformed on the host."""
        if cin_pad is not None and cin_pad == self.weight.shape[1]:
            cin_pad = None

        def build():
            wu, cpad = nops.pack_conv_weights_winograd(self._hwio_numpy(cin_pad), chunk=chunk)
            return torch.from_numpy(wu).to(self.weight.device), cpad
        return self._cache.get(("wino", chunk, cin_pad), _stamp(self.weight), build)

    def _packed_weights_small6(self, cin_pad=None):
        """(wp6 int16 bits, CoutPad) for m4d_conv3x3_small6_bias_act: the TF kernel split into three bf16 terms on the host."""
        if cin_pad is not None and cin_pad == self.weight.shape[1]:
            cin_pad = None

        def build():
            wp, cpad = nops.pack_conv_weights_small6(self._hwio_numpy(cin_pad))
            return torch.from_numpy(wp.view("int16")).to(self.weight.device), cpad
        return self._cache.get(("small6", cin_pad), _stamp(self.weight), build)

    def _packed_weights_lat(self, cin_pad=None):
        """wp (int16 bits) for m4d_conv3x3_lat: the TF kernel split into three bf16 terms, MFMA B-fragment order."""
        if cin_pad is not None and cin_pad == self.weight.shape[1]:
            cin_pad = None

        def build():
            wp = nops.pack_conv_weights_lat(self._hwio_numpy(cin_pad))
            return torch.from_numpy(wp.view("int16")).to(self.weight.device)
        return self._cache.get(("lat", cin_pad), _stamp(self.weight), build)

    def lat_eligible(self, b, h, w, cin):
        """Would ``forward`` run this layer on the latency-first small-map kernel for an input of this shape?"""
        if not (self.small_maps_ok and conv_arith == "bf16x3" and cin >= 16 and cin % 4 == 0 and self.stride in (1, 2)):
            return False
        eff_b = self.dispatch_batch if self.per_image_dispatch else b
        if self.stride == 2:                              # the stride-2 encoder layers: by OUTPUT pixels
            opx = eff_b * (-(-h // 2)) * (-(-w // 2))
            return self._lat_mw(b, h, w, cin) or (small_map_stride2 and opx <= lat_conv_s2_max_pixels)
        limit = max(lat_conv_max_pixels, lat_conv_narrow_max_pixels) if self.out_channels <= 32 else lat_conv_max_pixels
        return eff_b * h * w <= limit

    def _lat_mw(self, b, h, w, cin):
        """The M-over-waves form of the latency-first kernel for this (stride-2, narrow, short-K) layer?"""
        eff_b = self.dispatch_batch if self.per_image_dispatch else b
        return (self.stride == 2 and cin % 16 == 0 and cin <= 64 and self.out_channels <= 64
                and small_map_conv_pixels < eff_b * (-(-h // 2)) * (-(-w // 2)) <= lat_conv_mw_max_pixels)

    def _packed_weights_wino6(self, cin_pad=None):
        """(wu6 int16 bits, CoutPad) for m4d_conv3x3_wino6_bias_act: U = G g G^T split into three bf16 terms on the host."""
        if cin_pad is not None and cin_pad == self.weight.shape[1]:
            cin_pad = None

        def build():
            wu, cpad = nops.pack_conv_weights_wino6(self._hwio_numpy(cin_pad))
            return torch.from_numpy(wu.view("int16")).to(self.weight.device), cpad
        return self._cache.get(("wino6", cin_pad), _stamp(self.weight), build)

    def _hwio_device(self):
        """The TF-layout [3,3,Cin,Cout] kernel as a contiguous device tensor (the encoder-head kernel reads it directly)."""
        return self._cache.get(("hwio",), _stamp(self.weight), lambda: self.weight.detach().permute(2, 3, 1, 0).contiguous())

    def same_pads(self, h, w):
        """TF 'SAME' (before, after) pads for rows and columns at this stride."""
        s = self.stride
        ph = max((-(-h // s) - 1) * s + 3 - h, 0)
        pw = max((-(-w // s) - 1) * s + 3 - w, 0)
        return (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2)

    def launch_stagger_us(self, b, h, w):
        """The staggered-first-round argument of this layer's Winograd launch on a [b,h,w] map: the layer's (= its model's) setting
        on launches that put a workgroup on every CU, 0 elsewhere.  Pure function of the layer and the shape -- no global state."""
        us = wino6_stagger_us if self.wino6_stagger_us is None else self.wino6_stagger_us
        if us <= 0 or b > wino6_stagger_max_batch:
            return 0
        units = b * (-(-h // 16)) * (-(-w // 16)) * (-(-self.out_channels // 64))
        return int(us) if units >= wino6_stagger_min_wg else 0

    def forward(self, x_nhwc, slope=None, final=True):
        """Convolution + bias (+ leaky_relu(slope) when ``slope`` is given): ONE hand-written HIP kernel.  There is no CPU /
        framework form of this layer in the product: a CPU tensor raises (the host-logic tests patch a torch stand-in over
        this method from tests/helpers.py to check the layer wiring and the ``same_pads`` rule without a GPU).
        ``final`` = False: the caller hands the result to another layer of this class, so on the latency-first small-map
        kernel it may come back as K-slice partial sums (``network_ops.PartialAct``) that layer finishes while staging; a
        ``PartialAct`` input is finished here first (one launch) if this layer runs on any other kernel."""
        if isinstance(x_nhwc, nops.PartialAct) and not self.lat_eligible(*x_nhwc.shape):
            x_nhwc = x_nhwc.dense()
        if not x_nhwc.is_cuda:
            raise RuntimeError("m4depth_amd convolutions run on the GPU only (libm4depth_hip.so): got a CPU tensor; "
                               "there is no CPU fallback")
        if self.weight is None:
            self._build(x_nhwc.shape[-1], x_nhwc.device)
        if self.stride not in (1, 2):
            raise ValueError(f"stride {self.stride} is not supported by the HIP convolution")
        b_, h_, w_, cin_ = x_nhwc.shape
        act = 1.0 if slope is None else slope
        # The encoder is run on frames stacked along the batch axis, and how many are stacked depends on the launch mode
        # (all T frames, two batches in the pipelined forward, one frame when streaming): its kernel choice must not
        # depend on that, or the modes stop being bit-identical -- decide from the per-image grid x the SEQUENCE batch
        # (``dispatch_batch``, set by the model from its input: the same in every launch mode), so that a batch-32 evaluation
        # does not run its coarse encoder levels on the latency kernels of a single small map.
        eff_b = self.dispatch_batch if self.per_image_dispatch else b_
        if self.lat_eligible(b_, h_, w_, cin_):
            wl = self._packed_weights_lat(cin_)
            # (configuration from the per-image grid x the dispatch batch, like the kernel choice: the same in every launch mode)
            cfg = nops.lat_config(eff_b, h_, w_, cin_, self.out_channels, final, self.stride, mw=self._lat_mw(b_, h_, w_, cin_))
            return _timed("conv", self.tag, lambda: nops.conv3x3_lat(x_nhwc, wl, self.bias, self.out_channels, act, config=cfg,
                                                                     stride=self.stride))
        wino = _use_winograd(eff_b, h_, w_, cin_, self.out_channels, self.stride)
        if wino == 6:
            wu, cpad = self._packed_weights_wino6(cin_)
            stag = self.launch_stagger_us(b_, h_, w_)
            kern = wino6_kernel
            if kern == 0 and wino6_pair_min_units > 0 and b_ <= wino6_stagger_max_batch and cin_ >= 32:
                units = b_ * (-(-h_ // 16)) * (-(-w_ // 16)) * (cpad // 64)
                if wino6_pair_min_units <= units < int(lib.m4d_wino6_persistent_min_units()):
                    kern = 16 + max(1, min(wino6_pair_units, 15))
            return _timed("conv", self.tag, lambda: nops.conv3x3_wino6_bias_act(
                x_nhwc, wu, self.bias, self.out_channels, cpad, act, kernel=kern, stagger_us=stag,
                stagger_phases=wino6_stagger_phases))
        if wino:
            wu, cpad = self._packed_weights_winograd(16 if wino == 1 else 8, cin_)     # cin_ > Cin: zero-padded input
            fn = nops.conv3x3_wino_bias_act if wino == 1 else nops.conv3x3_wino2_bias_act
            return _timed("conv", self.tag, lambda: fn(x_nhwc, wu, self.bias, self.out_channels, cpad, act))
        wp, cpad = self._packed_weights(cin_)
        if (self.small_maps_ok and self.stride == 1 and eff_b * h_ * w_ <= small_map_conv_pixels and 16 <= cin_ <= 256
                and cin_ % 4 == 0):
            if small_conv_split:
                wp6, cpad6 = self._packed_weights_small6(cin_)
                return _timed("conv", self.tag, lambda: nops.conv3x3_small6_bias_act(
                    x_nhwc, wp6, self.bias, self.out_channels, cpad6, act))
            return _timed("conv", self.tag, lambda: nops.conv3x3_small_bias_act(
                x_nhwc, wp, self.bias, self.out_channels, cpad, act))
        if (self.small_maps_ok and self.stride == 2 and small_map_stride2 and 16 <= cin_ <= 256 and cin_ % 4 == 0
                and eff_b * (-(-h_ // 2)) * (-(-w_ // 2)) <= small_map_conv_pixels):
            return _timed("conv", self.tag, lambda: nops.conv3x3_small_bias_act(      # one launch instead of split-K + reduce
                x_nhwc, wp, self.bias, self.out_channels, cpad, act, stride=2))
        return _timed("conv", self.tag, lambda: nops.conv3x3_bias_act(
            x_nhwc, wp, self.bias, self.out_channels, cpad, act, stride=self.stride))


class DomainNormalization(torch.nn.Module):
    """m4depth_network.py:24-48 (Zhang et al., domain-invariant stereo matching):
    (x - mean_hw) / (var_hw + 1e-12) -- var, not std -- then l2-normalise over
    channels, scale and bias."""

    def __init__(self, regularizer_weight=0.0004):
        super().__init__()
        self.regularizer_weight = regularizer_weight
        self.scale = None
        self.bias = None

    def _build(self, channels, device):
        self.scale = torch.nn.Parameter(torch.ones(1, 1, 1, channels, device=device), requires_grad=False)
        self.bias = torch.nn.Parameter(torch.zeros(1, 1, 1, channels, device=device), requires_grad=False)

    def forward(self, f_map, slope=1.0):
        """``slope`` != 1 additionally applies the leaky_relu that follows the layer in the
        encoder (fused into the same HIP pass).  One hand-written kernel (m4d_dinl_fwd_padded); like the convolutions this
        layer has no CPU / framework form in the product: a CPU tensor or a width the kernel has no instantiation for raises
        (the host-logic tests patch a torch stand-in from tests/helpers.py; the TRAINING graph records the layer with torch
        autograd ops, training.dinl_autograd)."""
        if self.scale is None:
            self._build(f_map.shape[-1], f_map.device)
        if not f_map.is_cuda:
            raise RuntimeError("m4depth_amd DomainNormalization runs on the GPU only (libm4depth_hip.so): got a CPU tensor; "
                               "there is no CPU fallback")
        if f_map.shape[-1] not in (16, 32):
            raise ValueError(f"DomainNormalization: the HIP kernel is instantiated for 16 and 32 channels, not {f_map.shape[-1]}")
        return nops.dinl_act(f_map, self.scale, self.bias, slope)


class FeaturePyramid(torch.nn.Module):
    """Encoder of the network (m4depth_network.py:51-90): per level conv s1 ->
    [DINL on level 0] -> leaky_relu(0.1) -> conv s2 -> leaky_relu(0.1)."""

    def __init__(self, settings, regularizer_weight=0.0004, trainable=True):
        super().__init__()
        self.use_dinl = settings["ablation"].DINL
        self.out_sizes = ENCODER_CHANNELS[:settings["nbre_lvls"]]
        cin = [3] + self.out_sizes[:-1]
        self.conv_layers_s1 = torch.nn.ModuleList([_Conv3x3SameTF(n, 1, ci) for n, ci in zip(self.out_sizes, cin)])
        self.conv_layers_s2 = torch.nn.ModuleList([_Conv3x3SameTF(n, 2, n) for n in self.out_sizes])
        self.dn_layers = torch.nn.ModuleList([DomainNormalization(regularizer_weight) for _ in self.out_sizes])
        for conv in list(self.conv_layers_s1) + list(self.conv_layers_s2):
            conv.per_image_dispatch = True
            conv.small_maps_ok = True            # the stride-1 layers of the coarsest levels: one launch instead of split-K + reduce

    def set_sequence_batch(self, b):
        """The sequence batch the stacked frames belong to: the convolutions choose their kernel from the per-image grid x this
        (not from how many frames a launch mode stacks)."""
        b = max(int(b), 1) if encoder_batch_dispatch else 1
        for conv in list(self.conv_layers_s1) + list(self.conv_layers_s2):
            conv.dispatch_batch = b

    def _head_ok(self, images):
        return (self.use_dinl and fused_encoder_head and images.is_cuda and images.shape[-1] == 3
                and self.conv_layers_s1[0].out_channels == 16 and self.conv_layers_s2[0].out_channels <= 32
                and self.conv_layers_s1[0].weight is not None)

    def head_stats(self, images):
        """DINL statistics (mean, var: [n,16]) of level 0 for all the ``images`` (a FrameStack of a whole sequence), or None
        when level 0 does not run as ``nops.encoder_level0``.  ``forward(batch, head_stats=rows of it)`` then skips its own
        statistics passes: the later encoder batch of a split sequence loses four dependent launches (~47 us)."""
        if not (self._head_ok(images) and fused_encoder_level0 and self.conv_layers_s2[0].out_channels == 16):
            return None
        conv_s1 = self.conv_layers_s1[0]
        return _timed("enc0_stats", 0, lambda: nops.encoder_level0_stats(images, conv_s1._hwio_device(), conv_s1.bias))

    def forward(self, images, head_stats=None):
        """``images``: [b,H,W,3], or a ``network_ops.FrameStack`` (the frames of a sequence batch, encoded in one pass)."""
        head_ok = self._head_ok(images)
        if isinstance(images, nops.FrameStack):
            feature_maps = images if head_ok else images.dense()
        else:
            feature_maps = as_f32(images, "images")
        outputs = []
        for i, (conv_s1, conv_s2, dn_layer) in enumerate(zip(self.conv_layers_s1, self.conv_layers_s2, self.dn_layers)):
            if i == 0 and head_ok:
                # level 0 in two fused calls: conv 3->16 + bias + DINL statistics; stride-2 conv normalising its input on the fly
                if dn_layer.scale is None:
                    dn_layer._build(16, feature_maps.device)
                if fused_encoder_level0 and conv_s2.out_channels == 16:
                    # one call, no [b,H,W,16] intermediate: the 3 -> 16 convolution recomputed on the matrix cores per pass
                    feature_maps = _timed("enc0", 0, lambda: nops.encoder_level0(
                        feature_maps, conv_s1._hwio_device(), conv_s1.bias, dn_layer.scale, dn_layer.bias,
                        conv_s2._hwio_device(), conv_s2.bias, 0.1, stats=head_stats))
                    outputs.append(feature_maps)
                    continue
                wp2, cpad2 = conv_s2._packed_weights()
                feature_maps = nops.encoder_head(feature_maps, conv_s1._hwio_device(), conv_s1.bias, dn_layer.scale, dn_layer.bias,
                                                 wp2, conv_s2.bias, conv_s2.out_channels, cpad2, 0.1)
                outputs.append(feature_maps)
                continue
            if self.use_dinl and i == 0:
                tmp = dn_layer(conv_s1(feature_maps), slope=0.1)
            else:
                # (consumed by conv_s2 alone: on the latency-first small-map kernels it may stay K-slice partial sums -- but only
                #  when conv_s2 is itself such a kernel and adds the slabs while it stages; otherwise the slabs would need a
                #  finishing launch of their own, a dependent launch on the encoder's latency chain)
                fb, fh, fw = feature_maps.shape[:3]
                takes_slabs = conv_s2.lat_eligible(fb, fh, fw, conv_s1.out_channels)
                tmp = conv_s1(feature_maps, slope=0.1, final=not (takes_slabs or encoder_s1_partial_always))
            feature_maps = conv_s2(tmp, slope=0.1)
            outputs.append(feature_maps)
        return outputs


class DispRefiner(torch.nn.Module):
    """Sub-network refining an input parallax estimate (m4depth_network.py:93-135):
    convs (128,128,96) then (64,32,16,5), leaky_relu(0.1) after all but the last.
    Returns ``[out5, prep96]`` like the reference (its ``zip`` quirk, :125-135)."""

    def __init__(self, regularizer_weight=0.0004, in_channels=None):
        super().__init__()
        chans = REFINER_CHANNELS
        cin = [in_channels] + chans[:-1]
        self.prep_conv_layers = torch.nn.ModuleList([_Conv3x3SameTF(n, 1, ci) for n, ci in zip(chans[:3], cin[:3])])
        self.est_d_conv_layers = torch.nn.ModuleList([_Conv3x3SameTF(n, 1, ci) for n, ci in zip(chans[3:], cin[3:])])
        for conv in list(self.prep_conv_layers) + list(self.est_d_conv_layers):
            conv.small_maps_ok = True

    def forward(self, feature_map):
        prev_out = feature_map
        for conv in self.prep_conv_layers:
            prev_out = conv(prev_out, slope=0.1)                          # (finished tensors: ``prep`` is returned)
        prep = prev_out
        n = len(self.est_d_conv_layers)
        for i, conv in enumerate(self.est_d_conv_layers):
            prev_out = conv(prev_out, slope=0.1 if i < n - 1 else None, final=(i == n - 1))   # last convolution: no activation
        return [prev_out, prep]


class DepthEstimatorLevel(torch.nn.Module):
    """Stackable decoder level (m4depth_network.py:138-262): outputs a depth and
    a parallax map.  In inference mode the level owns its temporal memory
    (``prev_f_maps``, ``depth_prev_t``, :159-163).

    ``settings`` may carry two build extensions next to the reference's keys:
    ``dscv_range`` (default 4, hard-coded at :221) and ``sncv_range`` (default 3,
    :232), plus ``cv_accum`` (see oracle [UNPINNED] note on the fp16 mean)."""

    def __init__(self, settings, depth, regularizer_weight=0.0004):
        super().__init__()
        self.is_training = settings["is_training"]
        self.ablation = settings["ablation"]
        self.dscv_range = int(settings.get("dscv_range", 4))
        self.sncv_range = int(settings.get("sncv_range", 3))
        self.cv_accum = settings.get("cv_accum", "fp32_round")
        self.lvl_depth = depth
        self.lvl_mul = depth - 3
        self.nbre_cuts = nbre_cuts_for(depth, self.ablation.subdivide_features)       # :173-176
        self.f_in = f_input_channels(self.nbre_cuts, self.dscv_range, self.sncv_range,
                                     self.ablation.level_memory, self.ablation.SNCV, self.ablation.time_recurr)
        self.disp_refiner = DispRefiner(regularizer_weight=regularizer_weight, in_channels=self.f_in)
        for i, conv in enumerate(list(self.disp_refiner.prep_conv_layers) + list(self.disp_refiner.est_d_conv_layers)):
            conv.tag = f"lvl{depth}.conv{i}"
        self.prev_f_maps = None
        self.depth_prev_t = None
        self._own = None                  # the level's two persistent feature-state buffers (see _spare_f)
        self.last_f_input = None          # kept for inspection / parity tests
        self.last_cv_inputs = None
        self.last_front_inputs = None

    # -- temporal memory (build(), :153-165) ------------------------------------------
    def _ensure_state(self, shape, device):
        if self._own is None or tuple(self._own[0].shape) != tuple(shape) or self._own[0].device != device:
            b, h, w, c = shape
            self._own = [torch.zeros(shape, dtype=torch.float32, device=device), torch.empty(shape, dtype=torch.float32, device=device)]
            self.prev_f_maps = self._own[0]
            self.depth_prev_t = torch.ones((b, h, w, 1), dtype=torch.float32, device=device)

    @property
    def _spare_f(self):
        """The level's own state buffer that is NOT the current ``prev_f_maps``: where the next normalised features are written
        (they then become ``prev_f_maps``: a pointer hand-over, :211 / :259).  The level OWNS its two buffers for its lifetime --
        a captured hipGraph has their addresses baked in, so neither may ever be dropped -- and never writes anywhere else:
        a ``prev_f_maps`` that came from outside (``forward(..., curr_f_normalized=...)``) is borrowed and only read."""
        if self._own is None:
            return None
        return self._own[1] if self.prev_f_maps is self._own[0] else self._own[0]

    def reset_state(self):
        self.prev_f_maps = None
        self.depth_prev_t = None
        self._own = None

    def reset_job(self, curr_f_maps):
        """This level's buffers for the one-launch pyramid reset (``nops.pyramid_reset``), or None when its reset branch has to
        go through ``forward`` (training, features kept raw, a cut width the group kernel does not cover)."""
        if self.is_training or not fused_level_front or not self.ablation.normalize_features:
            return None
        if not isinstance(curr_f_maps, torch.Tensor) or not curr_f_maps.is_cuda or curr_f_maps.dim() != 4:
            return None
        b, h, w, c = curr_f_maps.shape
        if not lib.m4d_pyramid_reset_supported(c, self.nbre_cuts):
            return None
        self._ensure_state((b, h, w, c), curr_f_maps.device)
        return {"features": curr_f_maps, "cuts": self.nbre_cuts, "state_features": self._spare_f, "depth_state": self.depth_prev_t}

    def reset_commit(self):
        """After ``nops.pyramid_reset``: the normalised features ARE the new prev_f_maps (:211) -- the swap ``forward`` does."""
        self.prev_f_maps = self._spare_f

    def _tail_weights(self, convs, split=False):
        """Packed weights of the fused level tail (conv 32->16, conv 16->5), built once per device: the fp32-MFMA kernel's
        (m4d_refiner_tail) or, ``split``, the bf16-split kernel's B fragments (m4d_refiner_tail6)."""
        def build():
            k6 = convs[5].weight.detach().permute(2, 3, 1, 0).cpu().numpy()
            k7 = convs[6].weight.detach().permute(2, 3, 1, 0).cpu().numpy()
            w6, w7 = nops.pack_refiner_tail_weights(k6, k7)
            dev = convs[5].weight.device
            return torch.from_numpy(w6).to(dev), torch.from_numpy(w7).to(dev)
        def build6():
            k6 = convs[5].weight.detach().permute(2, 3, 1, 0).cpu().numpy()
            k7 = convs[6].weight.detach().permute(2, 3, 1, 0).cpu().numpy()
            w6, w7 = nops.pack_refiner_tail_weights6(k6, k7)
            dev = convs[5].weight.device
            return torch.from_numpy(w6.view("int16")).to(dev), torch.from_numpy(w7.view("int16")).to(dev)
        if getattr(self, "_tail_cache", None) is None:
            self._tail_cache = _PackCache()
        if split:
            return self._tail_cache.get("tail6", _stamp(convs[5].weight, convs[6].weight), build6)
        return self._tail_cache.get("tail", _stamp(convs[5].weight, convs[6].weight), build)

    def _vector_processing(self, f_map, out=None):
        if self.ablation.normalize_features:                                          # :179-182
            return nops.normalize_cuts(f_map, self.nbre_cuts, out=out)
        if out is not None:
            out.copy_(f_map)
            return out
        return f_map

    def _front_by_size(self, b, h, w, c):
        """Does a [b,h,w,c] map of this level take the big-tile fused front (``m4d_level_front_r``)?  (size and shape part of
        ``forward``'s decision)"""
        k = self.nbre_cuts
        F_st = (self.f_in + 7) // 8 * 8 if pad_refiner_input else self.f_in
        return (fused_front and b * h * w > (fused_front_min_pixels if k <= 2 else max(fused_front_min_pixels, fused_front_coarse_min_pixels))
                and bool(lib.m4d_level_front_supported(c, k, self.dscv_range, self.sncv_range, F_st)))

    def wants_prenormalized(self, b, h, w, c):
        """Would a full (non-reset) inference step of this level on a [b,h,w,c] map open with ``m4d_level_front_small`` -- i.e.
        should the caller hand it per-cut normalised features (``forward(..., curr_f_normalized=...)``,
        ``network_ops.normalize_levels``)?"""
        ab = self.ablation
        return (small_level_front and fused_cost_volumes and fused_level_front and not self.is_training
                and ab.SNCV and ab.time_recurr and ab.normalize_features and ab.level_memory
                and b * h * w <= 6000 and not self._front_by_size(b, h, w, c)
                and bool(lib.m4d_pyramid_reset_supported(c, self.nbre_cuts))
                and bool(lib.m4d_level_front_small_supported(c, self.nbre_cuts, self.dscv_range, self.sncv_range)))

    def forward(self, curr_f_maps, prev_l_est, rot, trans, camera, new_traj, prev_f_maps=None, prev_t_depth=None,
                curr_f_normalized=None):
        """``curr_f_normalized`` (inference, optional): the per-cut normalised ``curr_f_maps`` computed ahead of the level
        (``wants_prenormalized``): the level then opens with ONE launch and the tensor becomes its ``prev_f_maps``."""
        curr_f_maps = as_f32(curr_f_maps, "curr_f_maps")
        b, h, w, c = curr_f_maps.shape
        dev = curr_f_maps.device
        k = self.nbre_cuts
        use_state = (not self.is_training) and prev_f_maps is None and prev_t_depth is None    # :192
        if not self.is_training:
            self._ensure_state((b, h, w, c), dev)
        nt = new_traj
        if isinstance(nt, torch.Tensor):
            nt = bool(nt.reshape(-1)[0].item())       # "sequences are synchronized over the batch" (:207)
        elif not isinstance(nt, bool):
            nt = bool(np.asarray(nt).reshape(-1)[0])
        ab = self.ablation
        F_in = self.f_in
        F_st = (F_in + 7) // 8 * 8 if (pad_refiner_input and dev.type == "cuda" and not self.is_training) else F_in
        # the fused level front: normalisation, upsampling, both cost volumes and the log features in one launch
        use_front = (fused_front and use_state and not nt and dev.type == "cuda" and self._spare_f is not None
                     and ab.SNCV and ab.time_recurr and ab.normalize_features and ab.level_memory
                     and b * h * w > (fused_front_min_pixels if k <= 2 else max(fused_front_min_pixels, fused_front_coarse_min_pixels))
                     and bool(lib.m4d_level_front_supported(c, k, self.dscv_range, self.sncv_range, F_st)))
        # a coarse level whose features arrive normalised: level_pre + DSCV + SNCV in one launch (m4d_level_front_small)
        use_small = (curr_f_normalized is not None and not use_front and use_state and not nt and dev.type == "cuda"
                     and kernel_timer_allows_small_front() and self.wants_prenormalized(b, h, w, c)
                     and tuple(curr_f_normalized.shape) == (b, h, w, c))
        # normalised current features land in the spare state buffer: after the level
        # ran they ARE the new prev_f_maps (:211, :259) -- a pointer swap, not a copy.
        # (inference, state mode) the normalisation shares a launch with level_pre below: both open the level, neither
        # depends on the other
        norm_job = None
        if use_front:
            curr_f = self._spare_f
        elif use_small:
            curr_f = as_f32(curr_f_normalized, "curr_f_normalized")
        elif (fused_level_front and self.ablation.normalize_features and dev.type == "cuda" and not self.is_training
                and prev_f_maps is None and self._spare_f is not None and c % self.nbre_cuts == 0):
            curr_f = self._spare_f
            norm_job = (curr_f_maps, self.nbre_cuts, curr_f)
        else:
            curr_f = self._vector_processing(curr_f_maps, out=self._spare_f if not self.is_training else None)
        if prev_f_maps is not None:
            prev_f_maps = self._vector_processing(as_f32(prev_f_maps, "prev_f_maps"))
        if use_state:
            prev_t_depth = self.depth_prev_t
            prev_f_maps = self.prev_f_maps

        if prev_t_depth is None or nt:                                                 # :208-214
            para_prev_l, depth_prev_l, other_prev_l, _ = _timed("pre", "reset", lambda: nops.level_pre(
                prev_l_est, None, None, None, b, h, w, dev, normalize=norm_job,
                depth_state_reset=None if self.is_training else self.depth_prev_t))         # :209, in the same launch
            if not self.is_training:
                self.prev_f_maps = curr_f                      # (curr_f is the spare buffer: the other one becomes the spare)
            return {"depth": depth_prev_l, "parallax": para_prev_l, "other": other_prev_l}

        r = self.dscv_range
        ncp = 2 * r + 1
        if F_st != F_in:                              # channel stride of the refiner input buffer > its width: zero padding
            f_buf = nops.zeroed_workspace(("f_input", self.lvl_depth), (b, h, w, F_st), dev)
        else:
            f_buf = torch.empty((b, h, w, F_in), dtype=torch.float32, device=dev)
        f_input = f_buf
        log_off = ncp * k
        other_off = log_off + 1 if self.ablation.level_memory else -1
        sncv_off = log_off + 1 + (4 if self.ablation.level_memory else 0)
        scale = float(2.0 ** self.lvl_mul)
        # "preprocessor" (:216-242): upsample coarser estimate, prev_d2para, log / memory features
        if not use_front and not use_small:
            para_prev_l, depth_prev_l, other_prev_l, para_prev_t = _timed("pre", self.lvl_depth, lambda: nops.level_pre(
                prev_l_est, as_f32(prev_t_depth, "prev_t_depth"), trans, camera, b, h, w, dev,
                f_input=f_input, log_off=log_off, other_off=other_off, log_scale=scale, normalize=norm_job))
        rot_t = as_f32(rot, "rot")
        tr = as_f32(trans, "trans").reshape(b, 3)
        cf = as_f32(camera["f"], "camera['f']").reshape(b, 2)
        cc = as_f32(camera["c"], "camera['c']").reshape(b, 2)
        if rot_t.dim() != 2 or rot_t.shape[1] not in (3, 4):
            raise ValueError('Rotation must be expressed as a small angle (x,y,z) or a quaternion (w,x,y,z)')
        fin_ptr = f_input.data_ptr()
        time_recurr = self.ablation.time_recurr
        log_ptr = ctypes.c_void_p(fin_ptr + 4 * (F_in - 1)) if time_recurr else None      # channel F_in - 1 of a stride-F_st row
        # DSCV (:220-221) -> f_input[..., 0:9k]; time-recurrence feature (:238) -> f_input[..., -1]
        prev_f = as_f32(prev_f_maps, "prev_f_maps")
        kt_on = kernel_timer is not None and getattr(kernel_timer, "enabled", True)
        if use_front:
            pp = po = None
            ph = pw = 0
            if prev_l_est is not None:
                pp = as_f32(prev_l_est["parallax"], "prev_l_est['parallax']")
                po = as_f32(prev_l_est["other"], "prev_l_est['other']")
                ph, pw = pp.shape[1:3]
            check(_timed("front", self.lvl_depth, lambda: lib.m4d_level_front_r(
                dptr(curr_f_maps, "curr_f_maps"), dptr(curr_f), dptr(prev_f), dptr(as_f32(prev_t_depth, "prev_t_depth")),
                dptr(pp), dptr(po), ph, pw, dptr(rot_t), rot_t.shape[1], dptr(tr), dptr(cf), dptr(cc),
                b, h, w, c, k, self.dscv_range, self.sncv_range, _CV_ACCUM[self.cv_accum], ctypes.c_void_p(fin_ptr), F_st, scale,
                stream_ptr())), "m4d_level_front_r")
            para_prev_t = para_prev_l = None
        elif use_small:
            pp = po = None
            ph = pw = 0
            if prev_l_est is not None:
                pp = as_f32(prev_l_est["parallax"], "prev_l_est['parallax']")
                po = as_f32(prev_l_est["other"], "prev_l_est['other']")
                ph, pw = pp.shape[1:3]
            # (timed as "dscv_sncv" by bench.py: the launch that opens the level in the graph)
            check(_timed("dscv_sncv", self.lvl_depth, lambda: lib.m4d_level_front_small(
                dptr(curr_f), dptr(prev_f), dptr(as_f32(prev_t_depth, "prev_t_depth")), dptr(pp), dptr(po), ph, pw,
                dptr(rot_t), rot_t.shape[1], dptr(tr), dptr(cf), dptr(cc), b, h, w, c, k, self.dscv_range, self.sncv_range,
                _CV_ACCUM[self.cv_accum], ctypes.c_void_p(fin_ptr), F_st, scale, stream_ptr())), "m4d_level_front_small")
            para_prev_t = para_prev_l = None
        elif fused_cost_volumes and self.ablation.SNCV and dev.type == "cuda" and b * h * w <= 6000:
            # small maps: both (independent) cost volumes in one launch (timed as "dscv_sncv" by bench.py: the launch the graph replays)
            check(_timed("dscv_sncv", self.lvl_depth, lambda: lib.m4d_dscv_sncv_fwd(
                dptr(curr_f), dptr(prev_f), dptr(para_prev_t), dptr(para_prev_l), dptr(rot_t), rot_t.shape[1],
                dptr(tr), dptr(cf), dptr(cc), b, h, w, c, r, k, _CV_ACCUM[self.cv_accum], ctypes.c_void_p(fin_ptr),
                F_st, None, log_ptr, F_st, scale, self.sncv_range, ctypes.c_void_p(fin_ptr + 4 * sncv_off), F_st,
                stream_ptr())), "m4d_dscv_sncv_fwd")
        else:
            check(_timed("dscv", self.lvl_depth, lambda: lib.m4d_dscv_fwd(
                dptr(curr_f), dptr(prev_f), dptr(para_prev_t), dptr(para_prev_l), dptr(rot_t), rot_t.shape[1],
                dptr(tr), dptr(cf), dptr(cc), b, h, w, c, r, k, _CV_ACCUM[self.cv_accum], ctypes.c_void_p(fin_ptr),
                F_st, None, log_ptr, F_st, scale, None, stream_ptr())), "m4d_dscv_fwd")
            if self.ablation.SNCV:                                                     # :231-233
                check(_timed("sncv", self.lvl_depth, lambda: lib.m4d_sncv_fwd(
                    dptr(curr_f), dptr(curr_f), b, h, w, c, self.sncv_range, 1, k,
                    ctypes.c_void_p(fin_ptr + 4 * sncv_off), F_st, stream_ptr())), "m4d_sncv_fwd")
        self.last_f_input = f_input if F_st == F_in else f_input[..., :F_in]      # the reference-width view (inspection / tests)
        self.last_cv_inputs = (curr_f, prev_f, para_prev_t, para_prev_l, rot_t, tr, cf, cc)   # for tools/bench_kernels.py
        self.last_front_inputs = (curr_f_maps, prev_l_est, prev_t_depth)
        if debug_tap is not None:
            debug_tap("encoder_map", self.lvl_depth, curr_f_maps)
            debug_tap("f_input", self.lvl_depth, self.last_f_input)
        # "depth_estimator" (:244-260)
        convs = list(self.disp_refiner.prep_conv_layers) + list(self.disp_refiner.est_d_conv_layers)
        if (fused_refiner_tail and dev.type == "cuda" and len(convs) == 7 and convs[5].weight is not None
                and tuple(convs[5].weight.shape[:2]) == (16, 32) and tuple(convs[6].weight.shape[:2]) == (5, 16)):
            x = f_input
            for ci, conv in enumerate(convs[:5]):
                x = conv(x, slope=0.1, final=(ci == 4))                   # the fused tail reads a finished tensor
                if debug_tap is not None:
                    debug_tap(f"refiner_conv{ci + 1}", self.lvl_depth, x.dense() if isinstance(x, nops.PartialAct) else x)
            split = tail_split and conv_arith == "bf16x3"
            w6p, w7p = self._tail_weights(convs, split)
            tail_fn = nops.refiner_tail6 if split else nops.refiner_tail
            para_curr_l, depth, other = _timed("tail", self.lvl_depth, lambda: tail_fn(
                x, w6p, convs[5].bias, w7p, convs[6].bias, rot_t, tr, {"f": cf, "c": cc}, scale,
                depth_state=self.depth_prev_t if not self.is_training else None))
        else:
            prev_out = self.disp_refiner(f_input)
            para_curr_l, depth, other = _timed("post", self.lvl_depth, lambda: nops.level_post(
                prev_out[0], rot_t, tr, {"f": cf, "c": cc}, scale, depth_state=self.depth_prev_t if not self.is_training else None))
        if not self.is_training:
            # the normalised features ARE the new prev_f_maps (:211, :259): the level's spare buffer, or -- the one-launch opening of
            # a coarse level -- the caller's tensor, borrowed and only read (the level's own two buffers stay its own: _spare_f)
            self.prev_f_maps = curr_f
        return {"other": other, "depth": depth, "parallax": para_curr_l}


def kernel_timer_allows_small_front():
    """(bench.py's per-kernel timer brackets the launches the graph replays: nothing to switch off -- kept as the one place
    where a measurement mode could ask for the separate kernels)"""
    return True


class DepthEstimatorPyramid(torch.nn.Module):
    """Decoder (m4depth_network.py:265-323): coarse -> fine loop over levels for
    every sequence step; level of depth l sees intrinsics / 2**l (:300-302)."""

    def __init__(self, settings, regularizer_weight=0.0004, trainable=True):
        super().__init__()
        self.levels = torch.nn.ModuleList(
            [DepthEstimatorLevel(settings, i + 1, regularizer_weight=regularizer_weight)
             for i in range(settings["nbre_lvls"])])
        self.is_training = settings["is_training"]
        self.is_unsupervised = False

    def prenormalize(self, pyr, bsz):
        """The per-cut normalised feature maps of the levels that open with ``m4d_level_front_small`` (``wants_prenormalized``),
        for a feature pyramid ``pyr`` (one tensor per level, possibly several frames stacked along the batch axis; ``bsz`` = the
        sequence batch) in ONE launch: [normalised tensor or None per level], or None when no level wants them.  Issued right
        behind the encoder: the normalisation does not depend on the decoder's state, so it leaves the levels' latency chains."""
        if self.is_training or pyr is None or not isinstance(pyr[0], torch.Tensor) or not pyr[0].is_cuda:
            return None
        jobs = [(lvl, t) for lvl, t in enumerate(pyr[:len(self.levels)])
                if t.dim() == 4 and self.levels[lvl].wants_prenormalized(bsz, t.shape[1], t.shape[2], t.shape[3])]
        if not jobs:
            return None
        outs = _timed("norm", "levels", lambda: nops.normalize_levels([(t, self.levels[lvl].nbre_cuts) for lvl, t in jobs]))
        res = [None] * len(pyr)
        for (lvl, _), o in zip(jobs, outs):
            res[lvl] = o
        return res

    @staticmethod
    def _frame_slice(npyr, j, bsz):
        """Frame ``j``'s slices of a stacked ``prenormalize`` result (None stays None)."""
        return None if npyr is None else [None if t is None else t[j * bsz:(j + 1) * bsz] for t in npyr]

    def pipeline_streams_for(self, traj_samples, dev):
        """Number of HIP streams the (frame, level) wavefront would use for this call; 0 = single stream."""
        n_streams = level_pipeline_streams if (dev.type == "cuda" and not self.is_training and len(traj_samples) > 1
                                               and not (kernel_timer is not None and getattr(kernel_timer, "enabled", True))) else 0
        if n_streams < 2:
            return 0
        frames = len(traj_samples)
        if frames <= n_streams:
            return frames
        return 0 if torch.cuda.is_current_stream_capturing() else n_streams      # eager: streams reused round-robin

    def forward(self, f_maps_pyrs, traj_samples, camera, training=False, encoder=None, n_maps_pyrs=None):
        """``f_maps_pyrs`` = per-frame feature pyramids, or None with ``encoder`` given: the pipelined path then
        encodes every frame on that frame's stream (overlapping with the decoder of the frames before it).
        ``n_maps_pyrs`` (optional) = per frame the ``prenormalize``d maps of that frame's pyramid (None entries allowed)."""
        d_est_seq = []
        n_lvls = len(self.levels)
        # level-local intrinsics camera / 2**depth (:300-302): the same for every sequence step
        dev = camera["f"].device
        if isinstance(camera["f"], torch.Tensor) and camera["f"].is_cuda:
            local_cameras = nops.camera_pyramid(camera, n_lvls)       # all levels in one launch
        else:
            local_cameras = [{"f": camera["f"] / 2. ** (lvl + 1), "c": camera["c"] / 2. ** (lvl + 1)}
                             for lvl in range(n_lvls)]
        n_pipe = self.pipeline_streams_for(traj_samples, dev)
        if n_pipe >= 2:
            return self._forward_pipelined(f_maps_pyrs, traj_samples, local_cameras, n_pipe, encoder, n_maps_pyrs)
        if f_maps_pyrs is None:
            f_maps_pyrs = [encoder(sample['RGB_im']) for sample in traj_samples]
        if n_maps_pyrs is None:
            n_maps_pyrs = [None] * len(f_maps_pyrs)
        for seq_i, (f_pyr_curr, sample) in enumerate(zip(f_maps_pyrs, traj_samples)):
            rot = sample['rot']
            trans = sample['trans']
            cnter = float(n_lvls)
            d_est_curr = None
            fused = self._reset_all_levels(f_pyr_curr, sample["new_traj"])
            if fused is not None:
                d_est_seq.append(fused[::-1])
                continue
            for l in range(n_lvls):
                lvl = n_lvls - 1 - l
                f_maps_curr, level = f_pyr_curr[lvl], self.levels[lvl]
                f_maps_prev = None
                d_est_prev = None
                if self.is_training and seq_i != 0:
                    f_maps_prev = f_maps_pyrs[seq_i - 1][-l - 1]
                    d_est_prev = d_est_seq[-1][-l - 1]["depth"]
                local_camera = local_cameras[lvl]
                d_est = None if d_est_curr is None else dict(d_est_curr[-1])
                if l == 0 and n_maps_pyrs[seq_i] is None and not self.is_training:
                    n_maps_pyrs[seq_i] = self.prenormalize(f_pyr_curr, f_pyr_curr[0].shape[0])   # (a frame that arrived without)
                n_curr = None if n_maps_pyrs[seq_i] is None else n_maps_pyrs[seq_i][lvl]
                est = level(f_maps_curr, d_est, rot, trans, local_camera, sample["new_traj"],
                            prev_f_maps=f_maps_prev, prev_t_depth=d_est_prev, curr_f_normalized=n_curr)
                d_est_curr = [est] if d_est_curr is None else d_est_curr + [est]
                cnter -= 1.
            d_est_seq.append(d_est_curr[::-1])
        return d_est_seq

    def _reset_all_levels(self, f_pyr, new_traj):
        """A new-trajectory frame in one launch: every level's state seeded, the estimates returned coarse -> fine; None when the
        frame is not a reset frame or a level cannot take part (then the per-level loop runs)."""
        if not fused_pyramid_reset or self.is_training:
            return None
        nt = new_traj
        if isinstance(nt, torch.Tensor):
            nt = bool(nt.reshape(-1)[0].item())
        elif not isinstance(nt, bool):
            nt = bool(np.asarray(nt).reshape(-1)[0])
        if not nt or not isinstance(f_pyr[0], torch.Tensor) or not f_pyr[0].is_cuda:
            return None
        order = list(range(len(self.levels) - 1, -1, -1))                             # coarse -> fine
        jobs = [self.levels[lvl].reset_job(as_f32(f_pyr[lvl], "curr_f_maps")) for lvl in order]
        if any(j is None for j in jobs):
            return None
        ests = _timed("pre", "reset", lambda: nops.pyramid_reset(jobs, jobs[0]["features"].shape[0]))
        for lvl in order:
            self.levels[lvl].reset_commit()
        return ests

    def _forward_pipelined(self, f_maps_pyrs, traj_samples, local_cameras, n_streams, encoder=None, n_maps_pyrs=None,
                           resume=None):
        """The same loop as a wavefront over (frame, level) on ``n_streams`` HIP streams: frame t runs on
        stream t % n; before level l of frame t it waits for the event recorded after level l of frame
        t-1 (the level's temporal memory).  Inside a stream the levels stay in coarse-to-fine order.
        Works eagerly and under hipGraph capture (the side streams fork from / join the capturing one;
        under capture the caller guarantees one stream per frame).

        ``resume`` = {frame: its estimates so far, coarse -> fine}: those (frame, level) cells ran BEFORE this call, ordered ahead
        of it on the calling stream (``SegmentedSequence``: the reset frame and the first full frame's coarse levels are graphs of
        their own); the wavefront continues from there -- every feature pyramid is given, only the streams with work left fork
        and join, and a wait for a cell that ran earlier is dropped (the calling stream orders it)."""
        n_lvls = len(self.levels)
        main = torch.cuda.current_stream()
        if getattr(self, "_streams", None) is None or len(self._streams) < n_streams:
            self._streams = [torch.cuda.Stream() for _ in range(n_streams)]
        streams = self._streams[:n_streams]
        keep = []                                     # events must outlive a hipGraph capture they are part of
        fork = torch.cuda.Event()
        fork.record(main)
        keep.append(fork)
        n_fr = len(traj_samples)
        ran = set() if resume is None else {(f, l) for f, ests in resume.items() for l in range(len(ests))}

        def is_reset(f):
            nt = traj_samples[f]["new_traj"]
            return bool(nt.reshape(-1)[0].item()) if isinstance(nt, torch.Tensor) else bool(np.asarray(nt).reshape(-1)[0])
        # frame -> stream: one per frame; with pipeline_merge_reset_frame a reset frame rides on the next frame's stream
        stream_of = list(range(n_fr))
        if resume is not None and segmented_resumed_on_main:
            # a frame under way continues on the CALLING stream (-1), the others fork
            stream_of = [-1 if 0 < len(resume.get(f, [])) < n_lvls else f for f in range(n_fr)]
        if pipeline_merge_reset_frame and resume is None:
            for f in range(n_fr - 1):
                if is_reset(f) and not is_reset(f + 1):
                    stream_of[f] = f + 1

        def launches_encoder(f):
            """Frame ``f`` issues an encoder pass on its own stream (reads the input images, produced on the main stream)."""
            return f_maps_pyrs is None or (f_maps_pyrs[f] is None and (f == 0 or f_maps_pyrs[f - 1] is not None))

        def needs_fork(i_st):
            """Does stream ``i_st`` wait for the fork event itself?  It MUST when it hosts frame 0 (nothing else orders it behind
            the main stream's encoder -- also when a merged reset frame moved frame 0 onto it) or a frame that launches an encoder
            pass.  For the others the wait is implied by their wait for the previous frame's level -- but the redundant edge is not
            neutral: ROCm 7.2's hipGraph executor assigns nodes to its four streams from the edges it sees.
            ``pipeline_fork_frames`` = the streams (= frames, one stream per frame under capture) that keep the explicit wait."""
            hosted = [f for f in range(n_fr) if stream_of[f] % n_streams == i_st]
            if resume is not None:                    # nothing but the fork orders a resumed frame behind the calling stream
                return True
            if i_st == 0 or i_st >= n_fr or f_maps_pyrs is None:
                return True
            if any(f == 0 or launches_encoder(f) for f in hosted):
                return True
            return pipeline_fork_frames is None or i_st in pipeline_fork_frames
        # the streams that host a (frame, level) cell of this call (all of them unless the call resumes a sequence)
        active = [i_st for i_st in range(len(streams))
                  if resume is None or any((f, l) not in ran for f in range(n_fr)
                                           if stream_of[f] >= 0 and stream_of[f] % n_streams == i_st for l in range(n_lvls))]
        for i_st in active:
            if needs_fork(i_st):
                streams[i_st].wait_event(fork)        # encoder outputs / inputs are produced on the main stream
        done = {}
        late_encoder = None
        n_pyrs = list(n_maps_pyrs) if n_maps_pyrs is not None else [None] * n_fr
        f_pyrs = [None] * n_fr if resume is None else list(f_maps_pyrs)
        d_est = [None] * n_fr                         # per frame: estimates so far, coarse -> fine
        if resume is not None:
            for f, ests in resume.items():
                d_est[f] = list(ests)
        # (Measured, tools/step_profile.py + tools/ab_bench.sh: the late encoder batch below is independent of the first
        # frames' decoder, yet issuing it FIRST so that it runs beside their coarse-level chain costs 4.5 % -- chip-filling
        # kernels delay the chain's small ones; where it is issued now it overlaps the first level-1 pass instead.)
        # Issue order = anti-diagonals of the (frame, level) grid: level l of frame t right after level l of frame
        # t-1.  Every stream still sees its own frame coarse -> fine, but the launch (and hipGraph node) order puts
        # the next frame's coarse levels ahead of the current frame's fine ones.
        # (The capture order matters to ROCm's hipGraph executor: a node's FIRST-captured child continues its parent's node list
        # on the same executor stream.  Frame-major order gives the same layout as this one; later-frame-first inside a diagonal
        # or level-major order turn the lists level-major and cost 16 %: profiles/r04_graph_executor.txt.)
        order = [(seq_i, diag - seq_i) for diag in range(n_fr + n_lvls - 1)
                 for seq_i in range(max(0, diag - n_lvls + 1), min(n_fr, diag + 1))]

        if pipeline_merge_reset_frame:
            # a merged reset frame must be ISSUED before the frame it shares the stream with reaches the same level: issue the
            # whole reset frame first (it is six tiny launches)
            # (only the reset frame that OPENS the sequence: a later one waits on the frame before it, level by level, and keeps
            # its place on the anti-diagonals -- still ahead of the frame it shares the stream with)
            merged = [f for f in range(n_fr) if stream_of[f] != f and f == 0]
            order = [(f, l) for f in merged for l in range(n_lvls)] + [(f, l) for (f, l) in order if f not in merged]
        fused_reset_frames = set()
        order = [cell for cell in order if cell not in ran]
        if resume is not None and segmented_resumed_first:
            # the cells of a frame that was under way when the call resumed go FIRST: they depend on nothing inside this call, and
            # the hipGraph executor lays a graph out in capture order
            part = {f for f in resume if 0 < len(resume[f]) < n_lvls}
            order = [c for c in order if c[0] in part] + [c for c in order if c[0] not in part]
        for seq_i, l in order:
            lvl = n_lvls - 1 - l
            sample = traj_samples[seq_i]
            st = main if stream_of[seq_i] < 0 else streams[stream_of[seq_i] % n_streams]
            with torch.cuda.stream(st):
                if l == 0:
                    if f_maps_pyrs is None:          # per-frame encoder on the frame's own stream
                        f_pyrs[seq_i] = encoder(sample['RGB_im'])
                        n_pyrs[seq_i] = self.prenormalize(f_pyrs[seq_i], sample['RGB_im'].shape[0])
                    elif f_maps_pyrs[seq_i] is not None:
                        f_pyrs[seq_i] = f_maps_pyrs[seq_i]
                    elif f_pyrs[seq_i] is None:      # first frame of the late encoder batch: encode all remaining frames here
                        rest = [i for i in range(seq_i, n_fr) if f_maps_pyrs[i] is None]
                        bsz = sample['RGB_im'].shape[0]
                        tail = encoder(_stack_frames([traj_samples[i] for i in rest]))
                        ntail = self.prenormalize(tail, bsz)     # the coarse levels' normalised maps of the whole batch: one launch
                        for j, i in enumerate(rest):
                            f_pyrs[i] = [lvl[j * bsz:(j + 1) * bsz] for lvl in tail]
                            n_pyrs[i] = self._frame_slice(ntail, j, bsz)
                        enc_done = torch.cuda.Event()
                        enc_done.record(st)
                        keep.append(enc_done)
                        late_encoder = (seq_i, enc_done)
                    elif not (pipeline_skip_implied_encoder_wait and seq_i - 1 >= late_encoder[0]):
                        # a later frame of that batch: its features come from another stream (implied by the wait on the
                        # previous frame's level below when that frame is of the same encoder batch)
                        st.wait_event(late_encoder[1])
                if seq_i > 0 and stream_of[seq_i - 1] != stream_of[seq_i] and (seq_i - 1, l) not in ran:
                    st.wait_event(done[(seq_i - 1, lvl)])           # (same stream: ordered by the stream itself)
                if l == 0 and seq_i == 0:
                    # a reset frame that opens the sequence: ONE launch seeds every level (a later reset frame would have to wait
                    # for all levels of the frame before it -- events that do not exist yet in this issue order -- and takes
                    # the per-level path)
                    fused = self._reset_all_levels(f_pyrs[seq_i], sample["new_traj"])
                    if fused is not None:
                        d_est[seq_i] = fused
                        ev = torch.cuda.Event()
                        ev.record(st)
                        for any_lvl in range(n_lvls):
                            done[(seq_i, any_lvl)] = ev
                        fused_reset_frames.add(seq_i)
                if seq_i in fused_reset_frames:
                    continue
                prev = None if d_est[seq_i] is None else dict(d_est[seq_i][-1])
                est = self.levels[lvl](f_pyrs[seq_i][lvl], prev, sample['rot'], sample['trans'], local_cameras[lvl],
                                       sample["new_traj"],
                                       curr_f_normalized=None if n_pyrs[seq_i] is None else n_pyrs[seq_i][lvl])
                ev = torch.cuda.Event()
                ev.record(st)
                done[(seq_i, lvl)] = ev
                d_est[seq_i] = [est] if d_est[seq_i] is None else d_est[seq_i] + [est]
        d_est_seq = [ests[::-1] for ests in d_est]
        for st in (streams[i_st] for i_st in active):  # join
            ev = torch.cuda.Event()
            ev.record(st)
            main.wait_event(ev)
            keep.append(ev)
        keep.extend(done.values())
        self._events = keep
        if not torch.cuda.is_current_stream_capturing():
            for ests in d_est_seq:                    # results are consumed on the main stream from here on
                for est in ests:
                    for t in est.values():
                        t.record_stream(main)
        return d_est_seq


class M4Depth(torch.nn.Module):
    """MI355X-native model of M4Depth (m4depth_network.py:325-369, 433-489).

    ``model([traj_samples, camera])`` -> ``{"depth": [b,H,W,1]}`` (nearest x2
    upsample of the finest level); ``test_step`` / ``predict_step`` /
    ``compile`` / ``evaluate`` mirror the Keras harness used by ``main.py``."""

    def __init__(self, depth_type="map", nbre_levels=6, is_training=False, ablation_settings=None,
                 dscv_range=4, sncv_range=3, cv_accum="fp32_round"):
        super().__init__()
        self.ablation_settings = M4depthAblationParameters() if ablation_settings is None else ablation_settings
        self.model_settings = {
            "nbre_lvls": nbre_levels,
            "is_training": is_training,
            "ablation": self.ablation_settings,
            "dscv_range": dscv_range,
            "sncv_range": sncv_range,
            "cv_accum": cv_accum,
        }
        self.depth_type = depth_type
        self.encoder = FeaturePyramid(self.model_settings, regularizer_weight=0.)
        self.d_estimator = DepthEstimatorPyramid(self.model_settings, regularizer_weight=0.)
        self.step_counter = 0
        self.compiled_metrics = []
        self.last_estimates = None
        self.wino6_stagger_us = None        # set_wino6_stagger: None = the module default

    # -- weights -----------------------------------------------------------------------
    def load_numpy_weights(self, weights, device):
        """Load the dict produced by ``synthetic.init_weights`` (TF HWIO kernels)."""
        for i in range(self.model_settings["nbre_lvls"]):
            self.encoder.conv_layers_s1[i].load_hwio(weights[f"enc.s1.{i}.kernel"], weights[f"enc.s1.{i}.bias"], device)
            self.encoder.conv_layers_s2[i].load_hwio(weights[f"enc.s2.{i}.kernel"], weights[f"enc.s2.{i}.bias"], device)
        dn = self.encoder.dn_layers[0]
        sc = _to_device_f32(weights["enc.dn.0.scale"], device).reshape(1, 1, 1, -1)
        bi = _to_device_f32(weights["enc.dn.0.bias"], device).reshape(1, 1, 1, -1)
        if dn.scale is not None and dn.scale.shape == sc.shape and dn.scale.device == sc.device:
            with torch.no_grad():                      # in place: kernels (and captured graphs) read these by address
                dn.scale.copy_(sc)
                dn.bias.copy_(bi)
        else:
            dn.scale = torch.nn.Parameter(sc, requires_grad=False)
            dn.bias = torch.nn.Parameter(bi, requires_grad=False)
        for lvl in self.d_estimator.levels:
            convs = list(lvl.disp_refiner.prep_conv_layers) + list(lvl.disp_refiner.est_d_conv_layers)
            for i, conv in enumerate(convs):
                conv.load_hwio(weights[f"lvl.{lvl.lvl_depth}.conv.{i}.kernel"], weights[f"lvl.{lvl.lvl_depth}.conv.{i}.bias"], device)
        return self.prepack()

    def prepack(self):
        """Build every packed weight layout a layer can be dispatched to (direct, Winograd 16 / 8-channel chunks) NOW:
        which one is used depends on the batch and map size, packing is host work (a device-to-host copy), and that must
        never happen lazily inside a hipGraph capture."""
        for conv in self.modules():
            if isinstance(conv, _Conv3x3SameTF) and conv.weight is not None and conv.weight.is_cuda:
                cin = conv.weight.shape[1]
                conv._packed_weights()
                if small_conv_split and conv.small_maps_ok and conv.stride == 1 and 16 <= cin <= 256 and cin % 4 == 0:
                    conv._packed_weights_small6()
                if conv.small_maps_ok and conv_arith == "bf16x3" and cin >= 16 and cin % 4 == 0 and (
                        (conv.stride == 1 and max(lat_conv_max_pixels, lat_conv_narrow_max_pixels) > 0)
                        or (conv.stride == 2 and max(lat_conv_s2_max_pixels, lat_conv_mw_max_pixels) > 0)):
                    conv._packed_weights_lat()
                if cin == 3 or (conv.stride == 2 and cin == 16 and conv.out_channels == 16):
                    conv._hwio_device()                # the encoder's level-0 kernels read the TF layout directly
                if conv.stride == 1 and cin >= 16 and cin % 2 == 0:
                    conv._packed_weights_winograd(16)
                    if cin % 4 == 0:
                        conv._packed_weights_winograd(8)
                    if cin % 16 == 0 and cin >= 32 and conv.out_channels >= min(wino6_min_cout, 32):
                        conv._packed_weights_wino6()    # the dispatch's own condition (_use_winograd): a replayed hipGraph is
                                                        # refreshed through prepack() alone, so every layout a layer CAN be
                                                        # dispatched to must be (re)built here (round 4 left 64 -> 32 out)
        for lvl in self.d_estimator.levels:
            convs = list(lvl.disp_refiner.prep_conv_layers) + list(lvl.disp_refiner.est_d_conv_layers)
            c0 = convs[0] if convs else None
            if pad_refiner_input and c0 is not None and c0.weight is not None and c0.weight.is_cuda and c0.weight.shape[1] % 8 != 0:
                cin_pad = (c0.weight.shape[1] + 7) // 8 * 8           # the zero-padded refiner input (DepthEstimatorLevel.forward)
                c0._packed_weights(cin_pad)
                c0._packed_weights_winograd(16, cin_pad)
                c0._packed_weights_winograd(8, cin_pad)
                if small_conv_split and cin_pad <= 256:
                    c0._packed_weights_small6(cin_pad)
                if conv_arith == "bf16x3" and lat_conv_max_pixels > 0:
                    c0._packed_weights_lat(cin_pad)
                if cin_pad % 16 == 0 and cin_pad >= 32 and c0.out_channels >= min(wino6_min_cout, 32):
                    c0._packed_weights_wino6(cin_pad)
            if len(convs) == 7 and convs[5].weight is not None and convs[5].weight.is_cuda \
                    and tuple(convs[5].weight.shape[:2]) == (16, 32) and tuple(convs[6].weight.shape[:2]) == (5, 16):
                lvl._tail_weights(convs, tail_split and conv_arith == "bf16x3")
        return self

    def invalidate_packed(self):
        """Every layer's ``invalidate_packed`` + ``prepack``: after weights were written behind autograd's back
        (``param.data``-style writes); in-place updates on the parameters themselves need nothing."""
        for conv in self.modules():
            if isinstance(conv, _Conv3x3SameTF):
                conv.invalidate_packed()
        return self.prepack()

    def weights_stamp(self):
        """Cheap identity/version fingerprint of every convolution parameter (see ``_stamp``)."""
        return tuple((c.weight.data_ptr(), c.weight._version) for c in self.modules()
                     if isinstance(c, _Conv3x3SameTF) and c.weight is not None)

    def load_tf_checkpoint(self, prefix_or_dir, device):
        """Weights from a TensorFlow object-based checkpoint of the reference model (callbacks.py:98-111,
        the published pretrained weights): a checkpoint prefix, or a directory with a ``checkpoint`` state file."""
        from . import tf_checkpoint as TC
        prefix = prefix_or_dir
        if _os.path.isdir(prefix_or_dir):
            prefix = TC.latest_checkpoint(prefix_or_dir)
            if prefix is None:
                raise FileNotFoundError(f"no checkpoint state in {prefix_or_dir}")
        return self.load_numpy_weights(TC.load_m4depth_weights(prefix, self.model_settings["nbre_lvls"]), device)

    def numpy_weights(self):
        """Inverse of ``load_numpy_weights``: the dict of TF-layout (HWIO) arrays, e.g. for a checkpoint."""
        def hwio(conv):
            return conv.weight.detach().permute(2, 3, 1, 0).cpu().numpy().copy(), conv.bias.detach().cpu().numpy().copy()
        out = {}
        for i in range(self.model_settings["nbre_lvls"]):
            out[f"enc.s1.{i}.kernel"], out[f"enc.s1.{i}.bias"] = hwio(self.encoder.conv_layers_s1[i])
            out[f"enc.s2.{i}.kernel"], out[f"enc.s2.{i}.bias"] = hwio(self.encoder.conv_layers_s2[i])
        dn = self.encoder.dn_layers[0]
        out["enc.dn.0.scale"] = dn.scale.detach().reshape(-1).cpu().numpy().copy()
        out["enc.dn.0.bias"] = dn.bias.detach().reshape(-1).cpu().numpy().copy()
        for lvl in self.d_estimator.levels:
            convs = list(lvl.disp_refiner.prep_conv_layers) + list(lvl.disp_refiner.est_d_conv_layers)
            for i, conv in enumerate(convs):
                out[f"lvl.{lvl.lvl_depth}.conv.{i}.kernel"], out[f"lvl.{lvl.lvl_depth}.conv.{i}.bias"] = hwio(conv)
        return out

    def reset_state(self):
        for lvl in self.d_estimator.levels:
            lvl.reset_state()

    def set_wino6_stagger(self, us):
        """The staggered first round of THIS model's one-per-CU Winograd launches: ``us`` microseconds (0 = lock step, None = the
        module default ``network.wino6_stagger_us``).  A per-model setting handed to every launch as an argument
        (m4d_conv3x3_wino6_bias_act_ks): models with different settings, on different host threads, never see each other's."""
        for conv in self.modules():
            if isinstance(conv, _Conv3x3SameTF):
                conv.wino6_stagger_us = None if us is None else int(us)
        self.wino6_stagger_us = None if us is None else int(us)
        return self

    # -- forward -------------------------------------------------------------------------
    def forward(self, data, training=False):
        """``training=True`` on a model built with ``is_training=True`` and trainable parameters
        records the autograd graph (training.model_forward_train); everything else runs the
        fused inference kernels without recording anything."""
        if training and self.model_settings["is_training"] and torch.is_grad_enabled() \
                and any(p.requires_grad for p in self.parameters()):
            from . import training as TR
            self.step_counter += 1
            return TR.model_forward_train(self, data[0], data[1])
        with torch.no_grad():
            return self._forward_inference(data, training)

    def _forward_inference(self, data, training=False):
        traj_samples, camera = data[0], data[1]
        self.step_counter += 1
        # rot / trans dense float32 ONCE per frame (test_step's unstacking, m4depth_network.py:439-447, hands over [:, t] slices:
        # non-contiguous from batch 2 on -- converted here they cost one tiny copy per frame instead of one per level)
        traj_samples = [dict(s, rot=as_f32(s["rot"], "rot"), trans=as_f32(s["trans"], "trans"))
                        if isinstance(s.get("rot"), torch.Tensor) and isinstance(s.get("trans"), torch.Tensor) else s
                        for s in traj_samples]
        # The encoder is independent per frame (:358-360): run it ONCE on the frames stacked along
        # the batch axis (same per-sample arithmetic incl. the per-sample DINL statistics, a
        # quarter of the launches), then hand each frame its slice.
        n_fr = len(traj_samples)
        dev = camera["f"].device
        late_encoder_fn = self.encoder
        n_maps_pyrs = None
        same_shape = all(s['RGB_im'].shape == traj_samples[0]['RGB_im'].shape for s in traj_samples)
        self.encoder.set_sequence_batch(traj_samples[0]['RGB_im'].shape[0])
        if pipeline_encoder_per_frame and self.d_estimator.pipeline_streams_for(traj_samples, dev) >= 2:
            f_maps_pyrs = None                     # encoded per frame, on the frame's stream, inside the decoder pipeline
        elif pipeline_encoder_split > 0 and n_fr > pipeline_encoder_split and same_shape \
                and self.d_estimator.pipeline_streams_for(traj_samples, dev) >= 2:
            # frames [0, split) now, the rest on the stream of frame `split`: the second encoder batch runs underneath
            # the (launch-latency-bound) coarse levels of the first full frame instead of in front of them
            k = pipeline_encoder_split
            bsz = traj_samples[0]['RGB_im'].shape[0]
            # the DINL statistics of ALL frames in the first batch's launches (frame-major rows): the second batch, whose
            # duration is on the step's critical path, starts at its fused pass
            stats = self.encoder.head_stats(_stack_frames(traj_samples)) if encoder_stats_up_front else None
            if stats is not None:
                head = self.encoder(_stack_frames(traj_samples[:k]), head_stats=(stats[0][:k * bsz], stats[1][:k * bsz]))
                late_stats = (stats[0][k * bsz:], stats[1][k * bsz:])
                late_encoder_fn = lambda images: self.encoder(images, head_stats=late_stats)
            else:
                head = self.encoder(_stack_frames(traj_samples[:k]))
            f_maps_pyrs = [[lvl[t * bsz:(t + 1) * bsz] for lvl in head] for t in range(k)] + [None] * (n_fr - k)
            nhead = None if training else self.d_estimator.prenormalize(head, bsz)
            n_maps_pyrs = [self.d_estimator._frame_slice(nhead, t, bsz) for t in range(k)] + [None] * (n_fr - k)
        elif n_fr > 1 and same_shape:
            bsz = traj_samples[0]['RGB_im'].shape[0]
            stacked = self.encoder(_stack_frames(traj_samples))
            f_maps_pyrs = [[lvl[t * bsz:(t + 1) * bsz] for lvl in stacked] for t in range(n_fr)]
            nst = None if training else self.d_estimator.prenormalize(stacked, bsz)
            n_maps_pyrs = [self.d_estimator._frame_slice(nst, t, bsz) for t in range(n_fr)]
        else:
            f_maps_pyrs = [self.encoder(sample['RGB_im']) for sample in traj_samples]
        d_maps_pyrs = self.d_estimator(f_maps_pyrs, traj_samples, camera, training, encoder=late_encoder_fn,
                                       n_maps_pyrs=n_maps_pyrs)
        self.last_estimates = d_maps_pyrs          # per step, per level {depth, parallax, other} (fine -> coarse)
        if training:
            return d_maps_pyrs
        h, w = traj_samples[-1]['RGB_im'].shape[1:3]
        return {"depth": _timed("resize", 0, lambda: nops.resize_nearest(d_maps_pyrs[-1][0]["depth"], h, w))}

    # -- Keras-harness mirror (main.py:127-133) -------------------------------------------
    def compile(self, metrics=None, optimizer=None, **_unused):
        self.compiled_metrics = list(metrics or [])
        self.optimizer = optimizer

    def m4depth_loss(self, gts, preds):
        """m4depth_network.py:491-536."""
        from . import training as TR
        return TR.m4depth_loss(gts, preds, self.depth_type)

    def train_step(self, data, grad_sync=None):
        """m4depth_network.py:371-431; needs ``compile(optimizer=...)`` and trainable parameters
        (``training.set_trainable``)."""
        from . import training as TR
        if getattr(self, "optimizer", None) is None:
            raise RuntimeError("train_step: compile the model with an optimizer first")
        return TR.train_step(self, data, self.optimizer, grad_sync)

    def _update_metrics(self, gt_raw, est_raw, max_d=80.):
        """compiled_metrics.update_state on the clipped maps (:465-470).  With the default
        metric list on the GPU all 7 values come from ONE fused HIP pass."""
        from . import metrics as MT
        ms = self.compiled_metrics
        default = (len(ms) == 7 and gt_raw.is_cuda and isinstance(ms[0], MT.AbsRelError) and isinstance(ms[1], MT.SqRelError)
                   and isinstance(ms[2], MT.RootMeanSquaredError) and isinstance(ms[3], MT.RootMeanSquaredLogError)
                   and all(isinstance(ms[4 + i], MT.ThresholdRelError) and ms[4 + i].threshold == i + 1 for i in range(3)))
        if default:
            # the 7 Keras-Mean totals live in ONE tensor the metric objects hold views of; the metric kernel adds the batch's
            # values to it and writes the running means (what test_step returns) in the same launch
            acc = getattr(self, "_metric_acc", None)
            shared = (acc is not None and acc.device == gt_raw.device and all(
                m.total is not None and m.total.data_ptr() == acc[i].data_ptr() and m.count == ms[0].count
                for i, m in enumerate(ms)))
            if not shared and all(m.total is None for m in ms):
                acc = self._metric_acc = torch.zeros(7, dtype=torch.float32, device=gt_raw.device)
                for i, m in enumerate(ms):
                    m.total = acc[i]
                    m.count = 0
                shared = True
            if shared:
                # a FRESH result tensor per step (the caching allocator makes that a host-side pointer bump): what test_step
                # returns must not alias the next step's results, and a clone of one shared buffer was a 5-us copy kernel plus a
                # launch gap at the end of every step
                self._metric_mean = torch.empty(7, dtype=torch.float32, device=gt_raw.device)
                nops.depth_metrics(gt_raw, est_raw, max_d, total=acc, count=ms[0].count + 1, mean=self._metric_mean)
                for m in ms:
                    m.count += 1
                self._metric_mean_count = ms[0].count
            else:                                      # accumulators were touched from outside: per-metric update
                vals = nops.depth_metrics(gt_raw, est_raw, max_d)
                self._metric_acc = None
                for i, m in enumerate(ms):
                    m._update(vals[i])
            return
        gt = torch.clamp(gt_raw, 0.0, max_d)
        est = torch.clamp(est_raw, 0.001, max_d)
        for m in ms:
            m.update_state(gt, est)

    def _metric_results(self):
        """{name: result} of the compiled metrics (what test_step returns, m4depth_network.py:473-474)."""
        ms = self.compiled_metrics
        acc = getattr(self, "_metric_acc", None)
        if acc is not None and ms and all(m.total is not None and m.total.data_ptr() == acc[i].data_ptr()
                                          and m.count == ms[0].count for i, m in enumerate(ms)):
            if getattr(self, "_metric_mean_count", -1) == ms[0].count and ms[0].count > 0:
                res = self._metric_mean                      # written by the metric kernel itself, into this step's own tensor
            else:
                res = acc / float(max(ms[0].count, 1))
            return {m.name: res[i] for i, m in enumerate(ms)}
        return {m.name: m.result() for m in ms}

    @torch.no_grad()
    def test_step(self, data):
        """m4depth_network.py:433-474."""
        if data["depth"].dim() == 5:            # sequence input: metrics on the last frame only (:439-455)
            seq_len = data["depth"].shape[1]
            traj_samples = [{} for _ in range(seq_len)]
            for key in ["depth", "RGB_im", "new_traj", "rot", "trans"]:
                for i, item in enumerate(torch.unbind(data[key], dim=1)):
                    traj_samples[i][key] = item
            preds = self([traj_samples, data["camera"]], training=False)
            gt = data["depth"][:, -1]
            est = preds["depth"]
            new_traj = False
        else:                                   # one frame of a stream (:457-460)
            preds = self([[data], data["camera"]], training=False)
            gt = data["depth"]
            est = preds["depth"]
            nt = data["new_traj"]
            new_traj = bool(nt.reshape(-1)[0].item()) if isinstance(nt, torch.Tensor) else bool(np.asarray(nt).reshape(-1)[0])
        if not new_traj:                                                               # :465-470
            self._update_metrics(gt, est, 80.)
        return self._metric_results()

    @torch.no_grad()
    def predict_step(self, data):
        """m4depth_network.py:476-489."""
        preds = self([[data], data["camera"]], training=False)
        return {"image": data["RGB_im"], "depth": preds["depth"], "new_traj": data["new_traj"]}

    def graphed_test_step(self, data, runner):
        """``test_step`` for a 5-D sequence batch with the forward replayed from a
        ``GraphedSequence`` (metrics stay eager: Keras ``Mean`` keeps host-side counts)."""
        est = runner(data)
        self._update_metrics(data["depth"][:, -1], est, 80.)
        return self._metric_results()

    def evaluate(self, dataset):
        for m in self.compiled_metrics:
            m.reset_state()
        out = {}
        for batch in dataset:
            out = self.test_step(batch)
        return [float(out[m.name]) for m in self.compiled_metrics]


class GraphedSequence:
    """One ``M4Depth`` sequence forward captured in a hipGraph.

    A 384x1280 / 6-level frame is ~110 kernel launches; at batch 1 the eager loop is host-launch bound (GPU ~50 % idle,
    profiles/).  Capturing the whole sequence once and replaying it removes the host from the loop.  The capture
    is valid because the step has no host synchronisation: ``new_traj`` lives on the
    host and is baked in as the control flow of the captured sequence, every kernel of
    libm4depth_hip.so is enqueued on the capturing stream, and the level state buffers
    are persistent allocations made before the capture.

    Every instance warms up and captures on ITS OWN stream: the scratch buffers of network_ops are keyed by stream, so
    two replicas (bench.py --in-flight N) never share scratch, and everything a replay touches is allocated by the
    eager warm-up, not inside the capture.

    ``runner(data)`` copies a new batch into the static input buffers (device-side
    copies), replays, and returns the static ``depth`` output tensor.  The batch must have the captured shapes and
    the captured ``new_traj`` pattern (it is control flow here); anything else raises."""

    def __init__(self, model, example, warmup=2, autotune=None):
        """``autotune``: None = the module setting (``wino6_stagger_autotune``: capture the sequence with and without the staggered
        Winograd first round and keep the faster graph, batch <= 4 only); False = ONE capture with the model's own stagger
        setting (``M4Depth.set_wino6_stagger``); True = force the double capture."""
        self.model = model
        nt = example["new_traj"]
        self.new_traj = nt.clone() if isinstance(nt, torch.Tensor) else torch.as_tensor(np.asarray(nt))
        # rot / trans are kept FRAME-major ([T,b,k]) so that a frame's slice is dense at every batch size: a [:, t] slice of the
        # batch-major tensor is non-contiguous from batch 2 on and every level wrapper would copy it with a framework kernel
        # INSIDE the graph (round 4, batch 32: ~43 copy nodes per step on the latency chain).  The batch-major [b,T,k] views of
        # the same memory are what input_buffers() hands out and what __call__ copies new batches into.
        # (clone, not .contiguous(): at batch 1 the transposed view already counts as dense and would ALIAS the example's own
        # tensors -- a later batch copied into the static buffers then overwrote the caller's example)
        self._rot = as_f32(example["rot"], "rot").transpose(0, 1).clone(memory_format=torch.contiguous_format)
        self._trans = as_f32(example["trans"], "trans").transpose(0, 1).clone(memory_format=torch.contiguous_format)
        self.static = {"RGB_im": example["RGB_im"].clone(), "rot": self._rot.transpose(0, 1), "trans": self._trans.transpose(0, 1)}
        self.camera = {k: as_f32(v, f"camera[{k}]").clone() for k, v in example["camera"].items()}
        self.seq_len = self.static["RGB_im"].shape[1]
        model.prepack()
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(max(warmup, 1)):
                self._run()
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        self.capture_passes = 1                     # forward passes issued under capture (2 when both forms were captured)
        self.stagger_autotune_ms = None
        batch = int(self.static["RGB_im"].shape[0])
        model_us = getattr(model, "wino6_stagger_us", None)
        base_us = wino6_stagger_us if model_us is None else model_us
        if self.wants_autotune(batch, base_us, autotune):
            self._capture_autotuned(base_us)
        else:
            self.graph, self.depth = self._capture()
            self.stagger_us = base_us if batch <= wino6_stagger_max_batch else 0    # what the captured Winograd launches carry
        self.weights_stamp = model.weights_stamp()

    @staticmethod
    def wants_autotune(batch, stagger_us, autotune=None):
        """Is the sequence captured twice (staggered / lock-step Winograd first round) and the faster graph kept?  Only where the
        stagger applies at all: batch <= ``wino6_stagger_max_batch`` -- never at BASELINE configs[2] / configs[3]'s 32 sequences per
        rank, so the ranks of an 8-GPU run all replay the same graph form and none spends the 1.4 s."""
        want = wino6_stagger_autotune if autotune is None else bool(autotune)
        return bool(want and stagger_us > 0 and batch <= wino6_stagger_max_batch)

    def _capture(self):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=self.stream):
            out = self._run()
        return graph, out

    def _inspection_state(self):
        """The tensors a forward pass leaves on the model for inspection (per-level estimates, refiner inputs, ...): they live in
        the memory pool of the capture that produced them."""
        lv = self.model.d_estimator.levels
        return (self.model.last_estimates,
                [(l.last_f_input, l.last_cv_inputs, getattr(l, "last_front_inputs", None), l.prev_f_maps) for l in lv])

    def _restore_inspection_state(self, state):
        """... and each level's ``prev_f_maps``: after a coarse level's one-launch opening it is a tensor of the capture's pool
        (the state a following eager frame continues from must be the KEPT graph's)."""
        self.model.last_estimates = state[0]
        for l, (f_in, cv_in, front_in, prev_f) in zip(self.model.d_estimator.levels, state[1]):
            l.last_f_input, l.last_cv_inputs, l.last_front_inputs = f_in, cv_in, front_in
            l.prev_f_maps = prev_f

    def _capture_autotuned(self, stagger_us):
        """Two captures -- the Winograd launches with their first round staggered and without -- timed against each other here,
        the faster one kept.  Whether the stagger pays depends on the BOX: on some MI355X boxes of the pool the coarse-level
        kernels of the next frame queue behind the lock-step rounds of a level-1 layer and the stagger is worth +4.5 %; on others
        they do not (the same library runs 7 % faster there to begin with) and it costs 3 %.  The stagger is
        a launch ARGUMENT baked into the captured graph (ABI 6: no library state): the candidates differ in the per-model setting
        (``M4Depth.set_wino6_stagger``) they were captured under, the model keeps the winner's, and no other model, thread or
        later eager launch of another model is affected."""
        cands = []
        self.capture_passes = 2
        for us in (stagger_us, 0):
            self.model.set_wino6_stagger(us)
            graph, out = self._capture()
            cands.append([us, graph, out, [], self._inspection_state()])
        # Timed the way the graph will be used: back-to-back replays in ITS OWN steady state.  (A few replays right after the
        # capture, on a chip that idled through it, picked the wrong graph on every box; so did interleaved blocks of 24 replays,
        # and 60 + 40 replays per form on one box in six: the lock-step form runs ~2.5-4 % faster for its first 100-400 ms --
        # longer after the staggered form has run -- than it does afterwards, as if the synchronised load were throttled with a
        # delay.)  Per candidate and round: `settle` untimed replays, then `reps` timed; two rounds, the LAST one decides;
        # ~1.4 s in all at 384x1280 (kept short: the device's rate sags ~1 % over seconds of load, and the timed region follows).
        settle, reps = 100, 40
        with torch.cuda.stream(self.stream):
            for rnd in range(2):
                for c in cands:
                    self.graph = c[1]
                    for _ in range(settle):
                        self._replay()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        self._replay()
                    e1.record()
                    e1.synchronize()
                    c[3].append(e0.elapsed_time(e1))
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        best = min(cands, key=lambda c: c[3][-1])
        if wino6_stagger_force in ("staggered", "lock_step"):          # tests: either graph must serve
            best = cands[0] if wino6_stagger_force == "staggered" else cands[1]
        self.stagger_us, self.graph, self.depth = best[0], best[1], best[2]
        # the inspection tensors of the KEPT capture (the loser's pool is freed below: nothing may keep pointing into it)
        self._restore_inspection_state(best[4])
        self.stagger_autotune_ms = {int(c[0]): [round(v / reps, 4) for v in c[3]] for c in cands}
        self.model.set_wino6_stagger(self.stagger_us)  # this model's eager launches from here on follow the choice
        for c in cands:
            if c is not best:
                c[4] = None
                self._reset_graph(c[1])                 # the loser's graph and its memory pool

    @staticmethod
    def _reset_graph(graph):
        for g in (graph.values() if isinstance(graph, dict) else [graph]):
            g.reset()

    def _samples(self):
        nt = torch.unbind(self.new_traj, dim=1)
        return [{"RGB_im": self.static["RGB_im"][:, t], "rot": self._rot[t], "trans": self._trans[t], "new_traj": nt[t]}
                for t in range(self.seq_len)]

    def _run(self):
        return self.model([self._samples(), self.camera])["depth"]

    def input_buffers(self):
        """The graph's static input tensors ({"RGB_im", "rot", "trans", "camera": {"f", "c"}}): a producer (data loader,
        host-to-device copy) that writes the next batch straight into them and passes them back to ``__call__`` skips the
        device-to-device hand-over copies (``__call__`` only copies tensors that live elsewhere)."""
        return dict(self.static, camera=self.camera)

    def _check(self, name, got, want):
        if tuple(got.shape) != tuple(want.shape):
            raise ValueError(f"GraphedSequence: {name} has shape {tuple(got.shape)}, the graph was captured for {tuple(want.shape)}")

    def __call__(self, data=None):
        if data is not None:
            nt = data.get("new_traj") if isinstance(data, dict) else None
            if nt is not None:
                nt = nt if isinstance(nt, torch.Tensor) else torch.as_tensor(np.asarray(nt))
                # only new_traj[0, t] steers the control flow (m4depth_network.py:207-208), so that is what must match
                if tuple(nt.shape) != tuple(self.new_traj.shape) or not torch.equal(nt[0].cpu(), self.new_traj[0].cpu()):
                    raise ValueError("GraphedSequence: the batch's new_traj pattern differs from the captured control flow")
            for k in self.static:
                self._check(k, data[k], self.static[k])
                if data[k].data_ptr() != self.static[k].data_ptr():
                    self.static[k].copy_(data[k], non_blocking=True)
            for k in self.camera:
                self._check(f"camera[{k}]", data["camera"][k], self.camera[k])
                if data["camera"][k].data_ptr() != self.camera[k].data_ptr():
                    self.camera[k].copy_(data["camera"][k], non_blocking=True)
        stamp = self.model.weights_stamp()
        if stamp != self.weights_stamp:                 # the parameters changed (optimizer step, load_state_dict): refresh the
            if [p for p, _ in stamp] != [p for p, _ in self.weights_stamp]:
                raise RuntimeError("a convolution parameter was REPLACED (new address) after this hipGraph was captured: the "
                                   "graph still reads the old buffers -- update weights in place or build a new GraphedSequence")
            self.model.prepack()                        # packed copies in place -- the graph reads the same addresses
            self.weights_stamp = stamp
        self._replay()
        return self.depth

    def _replay(self):
        self.graph.replay()


class SegmentedSequence(GraphedSequence):
    """The same sequence forward as FOUR hipGraphs replayed on two real HIP streams, for the one overlap ROCm 7.2's hipGraph
    executor does not give a single captured graph (inside one graph the second encoder batch starts only
    when the first full frame's coarse-to-fine chain has run -- ~350 us of a nearly idle chip per batch-1 step):

        calling stream:  [a: encoder batch a (reset frame + first full frame), statistics of all frames, pyramid reset]
                         [c: first full frame, levels n..2 -- a latency chain of small kernels] -> wait(b) ->
                         [d: that frame's level 1 + the remaining frames as the (frame, level) wavefront, final upsampling]
        side stream:     wait(a) -> [b: encoder batch b (the remaining frames) + their normalised coarse maps]

    so that b's chip-filling kernels run BESIDE c's small ones.  Same kernels, same arguments, same order per buffer as
    ``GraphedSequence`` (bit-identical results: tests/test_gpu_model.py).  Needs a sequence that opens with a reset frame followed
    by at least two full frames (``eligible``); every other shape of work stays with ``GraphedSequence``."""

    @staticmethod
    def eligible(model, example):
        nt = example["new_traj"]
        nt = nt if isinstance(nt, torch.Tensor) else torch.as_tensor(np.asarray(nt))
        pat = [bool(v) for v in nt[0].reshape(-1).tolist()]
        return bool(example["RGB_im"].is_cuda and not model.model_settings["is_training"] and fused_pyramid_reset
                    and len(pat) >= 3 and pat[0] and not any(pat[1:]) and level_pipeline_streams >= len(pat))

    def __init__(self, model, example, warmup=2, autotune=None):
        if not self.eligible(model, example):
            raise ValueError("SegmentedSequence: needs a CUDA inference sequence [reset frame, >= 2 full frames]")
        self.stream_b = torch.cuda.Stream()
        self._ev_a, self._ev_b = torch.cuda.Event(), torch.cuda.Event()
        super().__init__(model, example, warmup=warmup, autotune=autotune)

    # -- the forward, cut into its four segments; ``run(name, fn)`` issues one --------------------------------------------
    def _segments(self, run):
        model, enc, dest = self.model, self.model.encoder, self.model.d_estimator
        samples = self._samples()
        n_fr, n_lvls, k = self.seq_len, len(dest.levels), 2
        bsz = int(self.static["RGB_im"].shape[0])
        S = {}

        def split(pyr, n):
            return [[lvl[t * bsz:(t + 1) * bsz] for lvl in pyr] for t in range(n)]

        def seg_a():
            model.step_counter += 1
            enc.set_sequence_batch(bsz)
            stats = enc.head_stats(_stack_frames(samples)) if encoder_stats_up_front else None
            if stats is not None:
                head = enc(_stack_frames(samples[:k]), head_stats=(stats[0][:k * bsz], stats[1][:k * bsz]))
                S["late_stats"] = (stats[0][k * bsz:], stats[1][k * bsz:])
            else:
                head = enc(_stack_frames(samples[:k]))
                S["late_stats"] = None
            S["f"] = split(head, k)
            nhead = dest.prenormalize(head, bsz)
            S["n"] = [dest._frame_slice(nhead, t, bsz) for t in range(k)]
            S["cams"] = nops.camera_pyramid(self.camera, n_lvls)
            fused = dest._reset_all_levels(S["f"][0], samples[0]["new_traj"])
            if fused is None:
                raise RuntimeError("SegmentedSequence: the one-launch pyramid reset refused this model")
            S["est"] = {0: fused}

        def seg_b():
            tail = enc(_stack_frames(samples[k:]), head_stats=S["late_stats"]) if S["late_stats"] is not None \
                else enc(_stack_frames(samples[k:]))
            ntail = dest.prenormalize(tail, bsz)
            S["f"] = S["f"] + split(tail, n_fr - k)
            S["n"] = S["n"] + [dest._frame_slice(ntail, j, bsz) for j in range(n_fr - k)]

        def seg_c():
            ests = []
            smp = samples[1]
            for l in range(n_lvls - 1):
                lvl = n_lvls - 1 - l
                ests.append(dest.levels[lvl](S["f"][1][lvl], dict(ests[-1]) if ests else None, smp["rot"], smp["trans"],
                                             S["cams"][lvl], smp["new_traj"],
                                             curr_f_normalized=None if S["n"][1] is None else S["n"][1][lvl]))
            S["est"][1] = ests

        def seg_d():
            d_maps = dest._forward_pipelined(S["f"], samples, S["cams"], n_fr, n_maps_pyrs=S["n"], resume=S["est"])
            model.last_estimates = d_maps
            h, w = samples[-1]["RGB_im"].shape[1:3]
            S["depth"] = nops.resize_nearest(d_maps[-1][0]["depth"], h, w)

        for name, fn in (("a", seg_a), ("b", seg_b), ("c", seg_c), ("d", seg_d)):
            run(name, fn)
        return S["depth"]

    def _play(self, issue):
        """The four segments on the current stream and the side stream, with the two cross-stream edges."""
        cur = torch.cuda.current_stream()
        issue("a")
        self._ev_a.record(cur)
        with torch.cuda.stream(self.stream_b):
            self.stream_b.wait_event(self._ev_a)
            issue("b")
            self._ev_b.record(self.stream_b)
        issue("c")
        cur.wait_event(self._ev_b)
        issue("d")

    def _run(self):
        """Eager pass (warm-up: allocates the per-stream scratch the captures will reuse)."""
        fns = {}

        def collect(name, fn):
            fns[name] = fn
            if name == "d":                               # all four closures exist: issue them in replay order
                self._play(lambda n: fns[n]())
        return self._segments(collect)

    def _capture(self):
        graphs = {}

        def run(name, fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream_b if name == "b" else self.stream):
                fn()
            graphs[name] = g
        out = self._segments(run)
        return graphs, out

    def _replay(self):
        self._play(lambda n: self.graph[n].replay())


def make_runner(model, example, warmup=2, autotune=None):
    """The hipGraph runner of a sequence step: ``SegmentedSequence`` where the module setting asks for it and the sequence is
    eligible, ``GraphedSequence`` otherwise."""
    if segmented_step and SegmentedSequence.eligible(model, example):
        return SegmentedSequence(model, example, warmup=warmup, autotune=autotune)
    return GraphedSequence(model, example, warmup=warmup, autotune=autotune)


class TapedSequence:
    """The sequence forward as LAUNCH TAPES (csrc/m4d_tape.hip) replayed on real HIP streams -- the (frame, level) wavefront
    of ``DepthEstimatorPyramid._forward_pipelined`` WITHOUT hipGraph:

        stream 0        enc_a: camera pyramid, encoder batch of frames < split, the reset frame's levels
        encoder stream  enc_b: encoder batch of the remaining frames
        stream t        level L, L-1, ... 1 of full frame t, one tape per (frame, level); level l of frame t waits for the
                        event recorded after level l of frame t - 1 (its temporal memory) and for its encoder batch

    A tape replay costs the host one C loop (~3 us per launch, ~180 launches per 4-frame step) and issues plain stream
    launches.  Measured (round 3): a chain of small kernels is NOT slowed by chip-filling kernels when
    both are plain stream launches (340 us against ~700 us as soon as either side is a hipGraph replay;
    profiles/r03_stream_vs_graph_probe.txt), yet the whole step is 3 % slower this way than as ONE hipGraph (1326-1338 against
    1379 frames/s): the step is bound by chip-time, not by the chains.  An EXPERIMENT since round 4: needs a
    ``make EXPERIMENTS=1`` build of the library (raises otherwise); same kernels in the same per-stream order: bit-identical
    results (tests/test_gpu_model.py, skipped on the product library); ``bench.py --schedule tape``."""

    def __init__(self, model, example, warmup=2, encoder_split=None):
        from ._lib import require_experiments
        require_experiments("network.TapedSequence (the launch tape, csrc/m4d_tape.hip)")
        self.model = model
        nt = example["new_traj"]
        self.new_traj = nt.clone() if isinstance(nt, torch.Tensor) else torch.as_tensor(np.asarray(nt))
        flags = [bool(v) for v in self.new_traj[0].reshape(-1).tolist()]
        if len(flags) < 2 or not flags[0] or any(flags[1:]):
            raise ValueError("TapedSequence: expected a sequence that starts with new_traj and continues without")
        # rot / trans are kept frame-major ([T,b,k]) so that a frame's slice is contiguous at every batch size: a non-contiguous
        # slice would make the level wrappers launch a framework copy kernel, which a tape cannot record.  The batch-major
        # [b,T,k] views of the same memory are what input_buffers() hands out.
        self._rot = example["rot"].transpose(0, 1).contiguous()
        self._trans = example["trans"].transpose(0, 1).contiguous()
        self.static = {"RGB_im": example["RGB_im"].clone(), "rot": self._rot.transpose(0, 1), "trans": self._trans.transpose(0, 1)}
        self.camera = {k: v.clone().contiguous() for k, v in example["camera"].items()}
        self.seq_len = T = self.static["RGB_im"].shape[1]
        self.n_lvls = L = len(model.d_estimator.levels)
        self.split = min(max(1, pipeline_encoder_split if encoder_split is None else int(encoder_split)), T)
        model.prepack()
        # Stream priorities (plain streams honour them, a hipGraph's executor streams do not): the frames are consumed in
        # order, so frame t's kernels go ahead of frame t + 1's whenever both have workgroups waiting for a CU.
        # M4D_TAPE_PRIORITIES = comma list for streams[0..T-1] + the late-encoder stream (torch: lower = more urgent).
        prio = [int(v) for v in _os.environ.get("M4D_TAPE_PRIORITIES", "").split(",") if v.strip() != ""]
        pr = lambda i: prio[min(i, len(prio) - 1)] if prio else 0
        self.streams = [torch.cuda.Stream(priority=pr(t)) for t in range(T)]   # [0]: encoder batch a + reset frame; [t]: full frame t
        self.s_enc = torch.cuda.Stream(priority=pr(T))
        self.ctx = {}
        segs = [("enc_a", self.streams[0], self._seg_enc_a)]
        if self.split < T:
            segs.append(("enc_b", self.s_enc, lambda: self._encode(self.split, T)))
        for t in range(1, T):
            for l in range(L):
                segs.append(((t, l), self.streams[t], lambda t=t, l=l: self._seg_level(t, l)))
        cur = torch.cuda.current_stream()
        for _ in range(max(warmup, 1)):                     # eager passes: state, scratch (keyed by stream), packed weights
            for _, st, fn in segs:
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    fn()
                torch.cuda.synchronize()
        self._pools, self.tapes = [], {}
        import warnings
        for name, st, fn in segs:
            # Recorded under a torch graph capture: the library's launches go to the tape, the torch graph stays EMPTY (the
            # forward launches nothing but libm4depth_hip.so kernels) and only its private memory pool matters -- it pins
            # the address of every tensor the segment allocates.
            g = torch.cuda.CUDAGraph()
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                with torch.cuda.graph(g, stream=st):
                    tid = int(lib.m4d_tape_begin())
                    if tid < 0:
                        raise RuntimeError("m4d_tape_begin: this thread is already recording a tape")
                    try:
                        fn()
                    finally:
                        n_rec = int(lib.m4d_tape_end())
            if not any("CUDA Graph is empty" in str(w.message) for w in caught):
                # torch captured something: a framework kernel (a copy of a non-contiguous input, a fill) sits in this
                # segment of the forward and would be missing from the replay
                raise RuntimeError(f"TapedSequence: segment {name} of the forward launched framework kernels beside "
                                   "libm4depth_hip.so's; it cannot be replayed from a launch tape (use GraphedSequence)")
            self._pools.append(g)
            self.tapes[name] = (tid, n_rec, st)
        torch.cuda.synchronize()
        self.depth = self.ctx["depth"]
        self.weights_stamp = model.weights_stamp()
        self._ev = {name: torch.cuda.Event() for name in self.tapes}
        self._ev_start, self._ev_join = torch.cuda.Event(), torch.cuda.Event()
        # issue order: anti-diagonals of the (frame, level) grid, as in DepthEstimatorPyramid._forward_pipelined
        self._order = [(t, diag - (t - 1)) for diag in range((T - 1) + L - 1)
                       for t in range(max(1, diag - L + 2), min(T - 1, diag + 1) + 1)]

    def __del__(self):
        try:
            for tid, _, _ in self.tapes.values():
                lib.m4d_tape_free(tid)
        except Exception:
            pass

    def launches_per_step(self):
        return sum(n for _, n, _ in self.tapes.values())

    def _samples(self):
        nt = torch.unbind(self.new_traj, dim=1)
        return [{"RGB_im": self.static["RGB_im"][:, t], "rot": self._rot[t], "trans": self._trans[t], "new_traj": nt[t]}
                for t in range(self.seq_len)]

    def _encode(self, a, b):
        samples = self._samples()
        bsz = samples[0]["RGB_im"].shape[0]
        self.model.encoder.set_sequence_batch(bsz)
        out = self.model.encoder(_stack_frames(samples[a:b]))
        for j, t in enumerate(range(a, b)):
            self.ctx[("f", t)] = [lvl[j * bsz:(j + 1) * bsz] for lvl in out]

    def _seg_enc_a(self):
        self.ctx["cams"] = nops.camera_pyramid(self.camera, self.n_lvls)
        self._encode(0, self.split)
        self.ctx[("est", 0)] = []
        for l in range(self.n_lvls):
            self._level(0, l)                              # the reset frame: state seeding only

    def _seg_level(self, t, l):
        if l == 0:
            self.ctx[("est", t)] = []
        self._level(t, l)
        if l == self.n_lvls - 1 and t == self.seq_len - 1:
            h, w = self.static["RGB_im"].shape[2:4]
            self.ctx["depth"] = nops.resize_nearest(self.ctx[("est", t)][-1]["depth"], h, w)
            self.model.last_estimates = [self.ctx[("est", i)][::-1] for i in range(self.seq_len)]

    def _level(self, t, l):
        """Level l (0 = coarsest) of frame t."""
        sample = self._samples()[t]
        ests = self.ctx[("est", t)]
        lvl = self.n_lvls - 1 - l
        prev = None if not ests else dict(ests[-1])
        ests.append(self.model.d_estimator.levels[lvl](self.ctx[("f", t)][lvl], prev, sample["rot"], sample["trans"],
                                                      self.ctx["cams"][lvl], sample["new_traj"]))

    def input_buffers(self):
        return dict(self.static, camera=self.camera)

    def _replay(self, name):
        tid, _, st = self.tapes[name]
        check(lib.m4d_tape_replay(tid, ctypes.c_void_p(st.cuda_stream)), "m4d_tape_replay")
        self._ev[name].record(st)

    def __call__(self, data=None):
        if data is not None:
            for k in self.static:
                if tuple(data[k].shape) != tuple(self.static[k].shape):
                    raise ValueError(f"TapedSequence: {k} has shape {tuple(data[k].shape)}, recorded {tuple(self.static[k].shape)}")
                if data[k].data_ptr() != self.static[k].data_ptr():
                    self.static[k].copy_(data[k], non_blocking=True)
            for k in self.camera:
                if data["camera"][k].data_ptr() != self.camera[k].data_ptr():
                    self.camera[k].copy_(data["camera"][k], non_blocking=True)
        stamp = self.model.weights_stamp()
        if stamp != self.weights_stamp:
            if [p for p, _ in stamp] != [p for p, _ in self.weights_stamp]:
                raise RuntimeError("a convolution parameter was REPLACED after these tapes were recorded: build a new TapedSequence")
            self.model.prepack()
            self.weights_stamp = stamp
        cur = torch.cuda.current_stream()
        T, ev = self.seq_len, self._ev
        self._ev_start.record(cur)
        for st in self.streams + [self.s_enc]:
            st.wait_event(self._ev_start)
        self._replay("enc_a")
        if "enc_b" in self.tapes:
            self._replay("enc_b")
        for (t, l) in self._order:
            st = self.streams[t]
            if l == 0:
                st.wait_event(ev["enc_a"])                     # the reset frame's state, the camera pyramid (+ frame t's features)
                if t >= self.split:
                    st.wait_event(ev["enc_b"])
            if t > 1:
                st.wait_event(ev[(t - 1, l)])
            self._replay((t, l))
        self._ev_join.record(self.streams[T - 1])
        cur.wait_event(self._ev_join)
        return self.depth

