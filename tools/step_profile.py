"""Where in a step are only sub-chip kernels in flight?  Takes the densest window of a rocprofv3 kernel trace (graph replays),
finds the step boundaries (the first kernel of an encoder batch, enc0_rgb_total_kernel, starts a step) and prints, per 100-us bin of the step, the
share of time covered by at least one kernel of >= THR workgroups, averaged over the steps in the window."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 128
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"])), r["Kernel_Name"]) for r in rows)
heads = [e[0] for e in ev if "enc0_stats_kernel" in e[3]] or [e[0] for e in ev if "enc0_rgb_total_kernel" in e[3]] or [e[0] for e in ev if "enc_head_conv_kernel" in e[3]]   # first kernel of an encoder batch
if len(heads) < 4:
    sys.exit("step_profile: no encoder-batch head kernels (enc0_rgb_total_kernel / enc_head_conv_kernel) in the trace -- nothing to profile")
# steps = consecutive head launches with a regular spacing (the timed graph replays): keep gaps within 20 % of the median
# head kernels per step: 2 when each encoder batch takes its own DINL statistics, 1 with network.encoder_stats_up_front --
# counted against the once-per-step metrics_finalize_kernel unless given
n_fin = sum(1 for e in ev if "metrics_finalize_kernel" in e[3])
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else (max(1, round(len(heads) / n_fin)) if n_fin else 2)
heads = heads[len(heads) % per_step::per_step][-13:]             # the last 12 steps: the timed graph replays
gaps = [b - a for a, b in zip(heads, heads[1:])]
med = sorted(gaps)[len(gaps) // 2]
steps = [(a, b) for a, b in zip(heads, heads[1:]) if abs((b - a) - med) < 0.2 * med]
print(f"{len(steps)} steps of ~{med / 1e3:.0f} us")
BIN = 100_000
nb = int(med // BIN) + 1
cover = [0.0] * nb
names = [collections.Counter() for _ in range(nb)]
for (s0, s1) in steps:
    big = sorted((max(s, s0), min(e, s1)) for s, e, wg, _ in ev if wg >= thr and e > s0 and s < s1)
    merged = []
    for s, e in big:
        if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
        else: merged.append([s, e])
    for s, e in merged:
        for k in range(int((s - s0) // BIN), min(nb, int((e - s0) // BIN) + 1)):
            lo, hi = s0 + k * BIN, s0 + (k + 1) * BIN
            cover[k] += max(0, min(e, hi) - max(s, lo))
    for s, e, wg, n in ev:
        if wg < thr and e > s0 and s < s1:
            k = min(nb - 1, int((max(s, s0) - s0) // BIN))
            names[k][n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:28]] += 1
for k in range(nb):
    c = cover[k] / (len(steps) * BIN)
    top = ", ".join(f"{n} x{v / len(steps):.1f}" for n, v in names[k].most_common(3))
    print(f"  {k * 100:5d}-{(k + 1) * 100:5d} us  big-kernel coverage {100 * c:5.1f} %  {'#' * int(20 * c):20s} small: {top}")
