"""Steady-state analysis of a rocprofv3 ``*_kernel_trace.csv``: finds the repeating
launch pattern at the end of the trace (the timed steps of bench.py), and reports,
for ONE step: per-kernel time, launch count, GPU-busy time, idle gaps and span."""
import csv
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "m4d::") if ("Args)" in name or "m4d" in name or "_kernel" in name and "anonymous" in name) else name
    if name.startswith("_ZN2ck") or "ck::" in name:
        mm = re.search(r"(kernel_[a-z_0-9]+)", name)
        return "ck::" + (mm.group(1) if mm else "kernel")
    name = re.sub(r"\(.*$", "", name)
    return name[:100]


def main(path, out=None):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    n = len(names)
    period = None
    for trail in range(0, 4):          # a few trailing launches (e.g. the final loss read-back) may follow the last step
        m = n - trail
        for P in range(40, m // 2):
            if names[m - P:m] == names[m - 2 * P:m - P]:
                period = P
                break
        if period is not None:
            rows, names, n = rows[:m], names[:m], m
            break
    lines = []
    if period is None:
        # fuzzy fallback (e.g. MIOpen picks a different split for one call): candidate periods are the
        # distances to earlier occurrences of the last kernel; accept >= 97 % position-wise agreement
        last = names[-1]
        for pos in range(n - 41, n // 2 - 1, -1):
            if names[pos] != last:
                continue
            P = n - 1 - pos
            if 2 * P > n:
                break
            same = sum(1 for a, b in zip(names[n - P:], names[n - 2 * P:n - P]) if a == b)
            if same >= 0.97 * P:
                period = P
                lines.append(f"approximately repeating pattern ({same}/{P} launches identical to the previous period)")
                break
    if period is None:
        lines.append("no repeating pattern found")
        period = min(n, 2000)
    step = rows[n - period:]
    t0 = int(step[0]["Start_Timestamp"])
    t1 = int(step[-1]["End_Timestamp"])
    prev_end = int(rows[n - period - 1]["End_Timestamp"]) if n > period else t0
    span = t1 - prev_end
    agg = OrderedDict()
    busy = 0
    for r in step:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        busy += d
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0])
        a[0] += 1
        a[1] += d
    lines.append(f"one steady-state step = {period} kernel launches, span {span / 1e3:.1f} us, GPU busy {busy / 1e3:.1f} us "
                 f"({100 * busy / span:.1f}%), idle {(span - busy) / 1e3:.1f} us")
    lines.append(f"{'kernel':102s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'%busy':>6s}")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:102s} {c:6d} {t / 1e3:10.1f} {t / c / 1e3:9.2f} {100 * t / busy:6.2f}")
    gaps = []
    pe = prev_end
    for r in step:
        gaps.append(max(int(r["Start_Timestamp"]) - pe, 0))
        pe = max(pe, int(r["End_Timestamp"]))
    big = sorted(range(len(gaps)), key=lambda i: -gaps[i])[:12]
    lines.append("largest gaps (us) [index in window: previous kernel -> next kernel]:")
    for i in sorted(big):
        prevn = short(step[i - 1]["Kernel_Name"]) if i > 0 else "(before window)"
        lines.append(f"  {gaps[i] / 1e3:9.1f}  [{i:5d}] {prevn[:60]} -> {short(step[i]['Kernel_Name'])[:60]}")
    gaps.sort()
    ng = len(gaps)
    lines.append(f"inter-kernel gaps: mean {sum(gaps) / ng / 1e3:.2f} us, median {gaps[ng // 2] / 1e3:.2f} us, "
                 f"p90 {gaps[int(ng * 0.9)] / 1e3:.2f} us, max {gaps[-1] / 1e3:.1f} us, sum {sum(gaps) / 1e3:.1f} us")
    durs = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
    lines.append(f"kernel durations: median {durs[ng // 2] / 1e3:.2f} us, <5us: {sum(d < 5000 for d in durs)}, "
                 f"<10us: {sum(d < 10000 for d in durs)}, >=100us: {sum(d >= 100000 for d in durs)}")
    m4d = sum(t for k, (c, t) in agg.items() if k.startswith("m4d::"))
    lines.append(f"hand-written (m4d::) kernels: {m4d / 1e3:.1f} us = {100 * m4d / busy:.1f}% of busy")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
