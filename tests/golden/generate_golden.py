"""Generates the committed golden vectors under tests/golden/.

PARITY UNPINNED (see oracle/m4depth_oracle.py header): the reference ships no
tests, fixtures or golden outputs and TensorFlow cannot be imported in the
build container, so these vectors are produced by this repo's own CPU
restatement of the reference algorithm.  They pin the oracle against silent
drift and travel to the GPU box as data (inputs + expected outputs only).

Run from the repo root:  python tests/golden/generate_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import m4depth_oracle as O          # noqa: E402
from m4depth_amd import synthetic as S          # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
F = np.float32


def motion(rng, b, quat=True):
    aa = rng.normal(0.0, 0.02, [b, 3])
    if quat:
        ang = np.linalg.norm(aa, axis=1, keepdims=True)
        rot = np.concatenate([np.cos(ang / 2), aa / np.maximum(ang, 1e-12) * np.sin(ang / 2)], axis=1).astype(F)
    else:
        rot = aa.astype(F)
    trans = rng.normal([0.0, 0.0, 0.3], 0.05, [b, 3]).astype(F)
    return rot, trans


def camera(b, h, w):
    return {"f": np.tile(np.array([[0.5 * w, 0.5 * h]], F), [b, 1]), "c": np.tile(np.array([[0.5 * w, 0.5 * h]], F), [b, 1])}


def unit_features(rng, b, h, w, C, k):
    return O.normalize_cuts(rng.standard_normal([b, h, w, C]).astype(F), k)


def gen_ops():
    rng = np.random.default_rng(20260928)
    d = {}
    # --- dense_image_warp (+ index grid) and back_project, incl. out-of-image flow
    img = rng.standard_normal([2, 9, 11, 5]).astype(F)
    flow = (rng.standard_normal([2, 9, 11, 2]) * 3.0).astype(F)
    flow[0, 0, 0] = [-20.0, -20.0]
    flow[0, 1, 1] = [20.0, 20.0]
    flow[1, 4, 5] = [0.0, 0.0]
    flow[1, 4, 6] = [1.0, -2.0]
    out, y0, x0 = O.dense_image_warp(img, flow, return_index=True)
    d.update(warp_img=img, warp_flow=flow, warp_out=out, warp_idx=np.stack([y0, x0], -1).astype(np.int32))
    bp_in = rng.standard_normal([2, 6, 7, 2, 3]).astype(F)
    bp_co = (rng.random([2, 6, 7, 3, 2, 2]) * np.array([8.0, 7.0]) - 1.0).astype(F)
    bp_co[0, 0, 0, 0, 0] = [3.0, 2.0]          # exact integer coordinates (x1 == x0)
    bp_co[0, 0, 1, 0, 0] = [6.0, 5.0]          # last pixel
    bp_g = rng.standard_normal([2, 6, 7, 3, 2, 3]).astype(F)
    gi, gc = O.back_project_grad(bp_in, bp_co, bp_g)
    d.update(bp_in=bp_in, bp_coords=bp_co, bp_out=O.back_project(bp_in, bp_co), bp_grad=bp_g, bp_gin=gi, bp_gco=gc)
    # --- converters
    b, h, w = 2, 10, 14
    cam = camera(b, h, w)
    for tag, quat in (("q", True), ("e", False)):
        rot, trans = motion(rng, b, quat)
        depth = (1.0 + 60.0 * rng.random([b, h, w, 1])).astype(F)
        para = O.depth2parallax(depth, rot, trans, cam)
        d.update({f"cv_{tag}_rot": rot, f"cv_{tag}_trans": trans, f"cv_{tag}_depth": depth, f"cv_{tag}_d2p": para,
                  f"cv_{tag}_p2d": O.parallax2depth(para, rot, trans, cam),
                  f"cv_{tag}_pd2p": O.prev_d2para(depth, rot, trans, cam),
                  f"cv_{tag}_recompute": O.recompute_depth(depth, rot, trans, cam)})
    rot, trans = motion(rng, b)
    rp_map = rng.standard_normal([b, h, w, 3]).astype(F)
    rp_depth = (2.0 + 30.0 * rng.random([b, h, w, 1])).astype(F)
    rp_out, (pmr, rotc) = O.reproject(rp_map, rp_depth, rot, trans, cam)
    d.update(rp_rot=rot, rp_trans=trans, rp_map=rp_map, rp_depth=rp_depth, rp_out=rp_out, rp_pmr=pmr, rp_rotc=rotc)
    # --- resizes / normalisation
    x = rng.standard_normal([2, 5, 7, 4]).astype(F)
    d.update(rs_x=x, rs_bil_x2=O.resize_bilinear_v1(x, 10, 14), rs_bil_odd=O.resize_bilinear_v1(x, 9, 13),
             rs_near_x2=O.resize_nearest(x, 10, 14), rs_near_odd=O.resize_nearest(x, 11, 15))
    nx = rng.standard_normal([2, 5, 6, 24]).astype(F)
    d.update(nm_x=nx, nm_k1=O.normalize_cuts(nx, 1), nm_k4=O.normalize_cuts(nx, 4), nm_k3=O.normalize_cuts(nx, 3))
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **d)


def gen_cost_volumes():
    rng = np.random.default_rng(20260929)
    d = {}
    cases = [("a", 2, 12, 16, 16, 1, 4, 3), ("b", 1, 10, 12, 32, 2, 2, 2), ("c", 1, 8, 10, 96, 4, 4, 3),
             ("d", 1, 6, 8, 20, 2, 3, 1)]          # d: nc = 10 -> the generic (non-float4) kernels
    for tag, b, h, w, C, k, rd, rs in cases:
        cam = camera(b, h, w)
        rot, trans = motion(rng, b)
        trans = (trans * np.array([4.0, 4.0, 1.0])).astype(F)          # a few pixels of parallax at this size
        c1 = unit_features(rng, b, h, w, C, k)
        c2 = unit_features(rng, b, h, w, C, k)
        disp = (0.3 + 4.0 * rng.random([b, h, w, 1])).astype(F)
        dpt = (0.3 + 4.0 * rng.random([b, h, w, 1])).astype(F)
        for acc in ("fp32_round", "fp16_seq"):
            cv, pd, y0, x0 = O.get_parallax_sweeping_cv(c1, c2, dpt, disp, rot, trans, cam, rd, k, cv_accum=acc,
                                                        return_index=True)
            d[f"{tag}_cv_{acc}"] = cv
        d.update({f"{tag}_c1": c1, f"{tag}_c2": c2, f"{tag}_disp": disp, f"{tag}_dpt": dpt, f"{tag}_rot": rot,
                  f"{tag}_trans": trans, f"{tag}_prev_disp": pd,
                  f"{tag}_idx": np.stack([y0, x0], -1).astype(np.int32),
                  f"{tag}_sncv": O.cost_volume(c1, c2, rs, nbre_cuts=k),
                  f"{tag}_sncv_auto": O.cost_volume(c1, c1, rs, nbre_cuts=k),
                  f"{tag}_meta": np.array([b, h, w, C, k, rd, rs], np.int32)})
    d["a_sncv_dil2"] = O.cost_volume(d["a_c1"], d["a_c2"], 2, dilation_rate=2, nbre_cuts=1)
    np.savez_compressed(os.path.join(OUT, "cost_volumes.npz"), **d)


def gen_model():
    """BASELINE config 1 (oracle config): 128x256, 3 levels, dscv/sncv range 2, b=1,
    T=3 (one reset frame + two full frames).  Inputs are regenerated from the seed
    by the tests (m4depth_amd.synthetic), only outputs are stored."""
    L, rd, rs, H, Wd, T, b, seed = 3, 2, 2, 128, 256, 3, 1, 1235
    W = S.init_weights(L, seed=42, dscv_range=rd, sncv_range=rs)
    samples, cam = S.make_sequence(b, T, H, Wd, seed=seed)
    model = O.M4Depth(W, L, dscv_range=rd, sncv_range=rs)
    out, seq = model(samples, cam)
    d = {"meta": np.array([L, rd, rs, H, Wd, T, b, seed], np.int32), "depth": out["depth"],
         "metrics": O.metrics_batch(samples[-1]["depth"], out["depth"])}
    for t in range(T):
        for l in range(L):
            for key in ("depth", "parallax", "other"):
                d[f"t{t}_l{l}_{key}"] = seq[t][l][key]
    for l in range(L):
        d[f"f_input_l{l}"] = model.levels[l].last_f_input
    np.savez_compressed(os.path.join(OUT, "model_cfg1.npz"), **d)


def gen_model_well_conditioned():
    """The WELL-CONDITIONED fixture (m4depth_amd.synthetic.well_conditioned_case: last refiner layer x 0.25, lateral
    motion), on which the north-star tolerance -- depth within 1e-4 relative -- is asserted on EVERY pixel: at BASELINE
    config-1 size (128x256, 3 levels, ranges 2/2, one reset + two full frames) and for one 384x1280 / 6-level frame pair
    (configs[1]'s geometry).  Inputs are regenerated from the seeds by the tests; stored: the per-level depth and
    parallax of the last frame (the model output is the nearest x2 upsampling of level 0's depth), the 7 metrics, and
    the float32 oracle's own largest relative depth error against its float64 evaluation (the fixture's noise floor)."""
    for tag, (L, rd, rs, H, Wd, T, b, seed) in (("cfg1", (3, 2, 2, 128, 256, 3, 1, 1235)),
                                                ("full", (6, 4, 3, 384, 1280, 2, 1, 1236))):
        W, samples, cam = S.well_conditioned_case(L, b, T, H, Wd, seed, rd, rs)
        out, seq = O.M4Depth(W, L, dscv_range=rd, sncv_range=rs)(samples, cam)
        with O.float64_reference():
            _, seq64 = O.M4Depth(W, L, dscv_range=rd, sncv_range=rs)(samples, cam)
        floor = max(float(np.max(np.abs(seq[-1][l]["depth"] - seq64[-1][l]["depth"]) / np.abs(seq64[-1][l]["depth"])))
                    for l in range(L))
        d = {"meta": np.array([L, rd, rs, H, Wd, T, b, seed], np.int32),
             "metrics": O.metrics_batch(samples[-1]["depth"], out["depth"]),
             "f32_vs_f64_max_rel_depth": np.array(floor, np.float64)}
        for l in range(L):
            d[f"l{l}_depth"] = seq[-1][l]["depth"]
            d[f"l{l}_parallax"] = seq[-1][l]["parallax"]
        print(f"well-conditioned {tag}: float32 oracle vs float64 evaluation, max relative depth error {floor:.2e}")
        np.savez_compressed(os.path.join(OUT, f"model_wc_{tag}.npz"), **d)


if __name__ == "__main__":
    gen_ops()
    gen_cost_volumes()
    gen_model()
    gen_model_well_conditioned()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
