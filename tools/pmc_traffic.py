#!/usr/bin/env python
"""HBM traffic per launch of the kernels bench.py prices, from rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in
separate passes, counters only -- never combined with tracing), written to profiles/pmc_traffic.json TOGETHER WITH the
kernel name the counters were read from and a hash of that kernel's source files: bench.py refuses an entry whose sources
have changed since (a stale measurement reads as null, not as a number).

bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB counters x 1024; FETCH_SIZE doubled per the gfx950 note of
MI355X_MICROARCH.md section HBM: wide coalesced reads are tallied at half their bytes).

usage (on the GPU box, from the repo root):  python tools/pmc_traffic.py --batch 1 [--batch 32]
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = "m4depth_amd/csrc/"
# bench name -> (driver command, kernel-name substring(s) to read, source files whose hash stamps the entry)
TARGETS = {
    "dscv": (["tools/bench_kernels.py", "--iters", "5", "--which", "dscv[wave]"], ["dscv_wave_kernel"],
             [CS + "m4d_dscv.hip", CS + "m4d_common.h"]),
    "sncv": (["tools/bench_kernels.py", "--iters", "5", "--which", "sncv"], ["sncv7_kernel", "sncv_"],
             [CS + "m4d_sncv.hip", CS + "m4d_sncv_small.h", CS + "m4d_common.h"]),
    "front": (["tools/bench_kernels.py", "--iters", "5", "--which", "front"], ["level_front_kernel"],
              [CS + "m4d_front.hip", CS + "m4d_common.h"]),
    # level 4 of the pyramid (24x80, C = 96, 4 cuts): the fused front against the separate kernels it replaces at batch >= 4
    "front_l4": (["tools/bench_kernels.py", "--iters", "5", "--level", "4", "--which", "front"], ["level_front_kernel"],
                 [CS + "m4d_front.hip", CS + "m4d_common.h"]),
    "dscv_l4": (["tools/bench_kernels.py", "--iters", "5", "--level", "4", "--which", "dscv[wave]"], ["dscv_wave_kernel", "dscv_"],
                [CS + "m4d_dscv.hip", CS + "m4d_common.h"]),
    "sncv_l4": (["tools/bench_kernels.py", "--iters", "5", "--level", "4", "--which", "sncv"], ["sncv7_kernel", "sncv_"],
                [CS + "m4d_sncv.hip", CS + "m4d_sncv_small.h", CS + "m4d_common.h"]),
    "wino_l1_128_128": (["tools/bench_conv_one.py", "--iters", "5", "--winograd", "2"], ["conv3x3_wino4_kernel"],
                        [CS + "m4d_wino.hip"]),
    "wino6_l1_128_128": (["tools/bench_conv_one.py", "--iters", "5", "--winograd", "6"], ["conv3x3_wino6_kernel", "conv3x3_wino6p_kernel"],
                         [CS + "m4d_wino6.hip", CS + "m4d_wino6p.hip", CS + "Makefile"]),      # (the Makefile: W6FLAGS picks the split)
    "conv_l1_128_128": (["tools/bench_conv_one.py", "--iters", "5", "--winograd", "0"], ["conv3x3_mfma_kernel"],
                        [CS + "m4d_conv.hip"]),
}


# BASELINE configs[4] (768x2560, search ranges 6 / 6, batch 1): the fused level-1 front with 13 hypotheses / a 13x13 window and the
# level-1 128 -> 128 refiner layer at 384x1280 (3840 units: the persistent kernel) -- `--config4`, stored under "config4_batch1"
C4_GEO = ["--height", "768", "--width", "2560", "--dscv-range", "6", "--sncv-range", "6"]
TARGETS_CONFIG4 = {
    "front": (["tools/bench_kernels.py", "--iters", "5", "--which", "front"] + C4_GEO, ["level_front_kernel"],
              [CS + "m4d_front.hip", CS + "m4d_common.h"]),
    "wino6_l1_128_128": (["tools/bench_conv_one.py", "--iters", "5", "--winograd", "6", "--h", "384", "--w", "1280"],
                         ["conv3x3_wino6_kernel", "conv3x3_wino6p_kernel"], [CS + "m4d_wino6.hip", CS + "m4d_wino6p.hip", CS + "Makefile"]),
}


def sha(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(os.path.join(ROOT, p), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def counter_pass(counter, cmd, tag):
    out_dir = f"/tmp/pmct_{tag}_{counter}"
    subprocess.run(["rm", "-rf", out_dir])
    env = dict(os.environ, TMPDIR="/tmp")
    try:                                       # a counter pass that does not come back must not take the GPU box with it
        res = subprocess.run(["timeout", "-k", "10", "240", "rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out_dir,
                              "-o", "p", "--", sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        print(f"  [{tag}] counter pass {counter} timed out")
        return {}
    files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(f"  [{tag}] no counter csv for {counter}: {res.stderr[-300:]}")
        return {}
    acc = defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r.get("Counter_Name") == counter:
            acc[r.get("Kernel_Name", "")].append(float(r["Counter_Value"]))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, action="append")
    ap.add_argument("--only", default="")
    ap.add_argument("--config4", action="store_true", help="the BASELINE configs[4] geometry (batch 1) -> key config4_batch1")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    args = ap.parse_args()
    try:
        doc = json.load(open(args.out))
    except Exception:
        doc = {}
    doc["_note"] = ("HBM bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB x 1024; FETCH_SIZE doubled per the gfx950 note of "
                    "MI355X_MICROARCH.md), rocprofv3 --pmc, one counter per pass, level-1 geometry of the 384x1280 pyramid "
                    "(192x640, C=16) / the level-1 128->128 refiner layer; tools/pmc_traffic.py.  sources_sha = sha256[:16] of "
                    "the kernel's source files when the counters were collected; bench.py refuses entries that no longer match.")
    doc["collected"] = time.strftime("%Y-%m-%d %H:%M:%S")
    raw_lines = []
    for b in ([1] if args.config4 else (args.batch or [1])):
        entries = doc.setdefault(f"config4_batch{b}" if args.config4 else f"batch{b}", {})
        for name, (cmd, kernels, sources) in (TARGETS_CONFIG4 if args.config4 else TARGETS).items():
            if args.only and name not in args.only.split(","):
                continue
            if not all(os.path.isfile(os.path.join(ROOT, s)) for s in sources):
                continue
            vals = {}
            kname = None
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                acc = counter_pass(counter, cmd + ["--batch", str(b)], f"{name}_b{b}")
                match = [(k, v) for k, v in acc.items() if any(s in k for s in kernels)]
                if not match:
                    continue
                k, v = max(match, key=lambda kv: len(kv[1]))          # the kernel the driver launched over and over (the model step around it launches each kernel once or twice)
                kname = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                vals[counter] = sum(v) / len(v)
                raw_lines.append(f"batch {b:3d} {name:18s} {kname:44s} {counter:11s} launches {len(v):4d} mean {vals[counter]:14.1f} KiB")
            if len(vals) == 2:
                entries[name] = {"bytes": int(round((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)),
                                 "fetch_kib": round(vals["FETCH_SIZE"], 1), "write_kib": round(vals["WRITE_SIZE"], 1),
                                 "kernel": kname, "sources": sources, "sources_sha": sha(sources)}
                print(f"batch {b} {name}: {entries[name]['bytes'] / 1e6:.2f} MB / launch ({kname})", flush=True)
                if name.startswith("wino"):
                    # matrix-core occupancy of the MFMA kernels: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) against
                    # the kernel's GPU-active cycles (GRBM_GUI_ACTIVE, summed over the 8 XCDs) -- printed by bench.py next to roofline.frac
                    extra = {}
                    for counter in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
                        acc = counter_pass(counter, cmd + ["--batch", str(b)], f"{name}_b{b}")
                        match = [(k, v) for k, v in acc.items() if any(s in k for s in kernels)]
                        if match:
                            k, v = max(match, key=lambda kv: len(kv[1]))
                            extra[counter] = sum(v) / len(v)
                            raw_lines.append(f"batch {b:3d} {name:18s} {kname:44s} {counter:24s} launches {len(v):4d} mean {extra[counter]:14.1f}")
                    if len(extra) == 2 and extra["GRBM_GUI_ACTIVE"] > 0:
                        entries[name]["mfma_busy_cycles"] = extra["SQ_VALU_MFMA_BUSY_CYCLES"]
                        entries[name]["gui_active_cycles_sum_xcd"] = extra["GRBM_GUI_ACTIVE"]
                        entries[name]["mfma_busy_frac"] = round(extra["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * extra["GRBM_GUI_ACTIVE"] / 8.0), 4)
    json.dump(doc, open(args.out, "w"), indent=1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "pmc_traffic_rows.txt"), "a") as fh:
        fh.write("\n".join(raw_lines) + "\n")
    print("\n".join(raw_lines))


if __name__ == "__main__":
    main()
