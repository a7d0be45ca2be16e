"""Condense a rocprofv3 ``*_kernel_stats.csv`` into a short table (kernel names
shortened, sorted by total time), marking the hand-written libm4depth_hip kernels."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN\d+_GLOBAL__N_1\d+(\w+?)(I[LbE]|E)", name)
    if "GLOBAL__N_1" in name and any(k in name for k in ("dscv", "sncv", "level_", "normalize", "resize", "converter", "interp", "backproject", "reproject")):
        mm = re.search(r"N_1\d+([a-z_0-9]+?kernel)(ILi(\d+)E|ILb(\d)E)?", name)
        if mm:
            return "m4d::" + mm.group(1) + (f"<{mm.group(3) or mm.group(4)}>" if mm.group(2) else "")
    if name.startswith("_ZN2ck"):
        mm = re.search(r"(kernel_\w+?)I", name)
        return "ck::" + (mm.group(1) if mm else "kernel")
    return name[:110]


def main(path, top=30):
    rows = list(csv.DictReader(open(path)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    total = sum(v[1] for v in agg.values())
    items = sorted(agg.items(), key=lambda kv: -kv[1][1])
    print(f"{'kernel':112s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'%':>6s}")
    for k, (c, t) in items[:top]:
        print(f"{k:112s} {c:7d} {t / 1e6:10.3f} {t / c / 1e3:9.2f} {100 * t / total:6.2f}")
    print("--- hand-written kernels (libm4depth_hip.so) ---")
    for k, (c, t) in items:
        if k.startswith("m4d::"):
            print(f"{k:112s} {c:7d} {t / 1e6:10.3f} {t / c / 1e3:9.2f} {100 * t / total:6.2f}")
    print(f"total kernel time {total / 1e6:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30)
