"""rocprofv3 durations of the roofline kernel (level-1 128->128 refiner layer = conv3x3_wino4_kernel on a 480-tile x
2-group grid of 512-thread workgroups) from a kernel trace of `bench.py --steps 10 --warmup 3`, to set beside the HIP-event average bench.py
prints.  conv3x3_wino4_kernel serves several wide layers of levels 1-2, so the --stats average of the NAME mixes them;
this keeps the launches with the level-1 Cout = 128 grid (960 workgroups: the 64->128 and the 128->128 layer), separates
the two layers by duration and isolates the last 15 launches of the 128->128 layer = bench.py's eager kernel-timing pass
(5 steps x 3 full frames, one stream).

    python tools/conv_roofline_check.py <kernel_trace.csv> [workgroups=960] [split_us=165]"""
import csv
import sys

NAME = "conv3x3_wino4_kernel"
wgs = int(sys.argv[2]) if len(sys.argv) > 2 else 960
split = float(sys.argv[3]) if len(sys.argv) > 3 else 165.0
rows = [r for r in csv.DictReader(open(sys.argv[1])) if NAME in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def workgroups(r):
    gx = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
    wx = int(r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or 256)
    return gx // max(wx, 1)


print(f"{NAME} launches: {len(rows)}")
sel = [r for r in rows if workgroups(r) == wgs]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel]
big = [x for x in d if x > split]
small = [x for x in d if x <= split]
print(f"  on the {wgs}-workgroup grid (level 1, Cout = 128): {len(d)}")
if big:
    print(f"  128->128 layer (> {split:.0f} us): n={len(big)}  mean {sum(big) / len(big):.2f} us  min {min(big):.2f}  max {max(big):.2f}"
          "   [graph replays overlap frames on several streams: individual launches stretch]")
if small:
    print(f"   64->128 layer (<= {split:.0f} us): n={len(small)}  mean {sum(small) / len(small):.2f} us")
last = big[-15:]
if last:
    print(f"  128->128 layer, last {len(last)} launches (the eager, single-stream kernel-timing pass bench.py brackets with HIP events): "
          f"mean {sum(last) / len(last):.2f} us  min {min(last):.2f}  max {max(last):.2f}")
