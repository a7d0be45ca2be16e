"""Per (kernel, grid) launch durations of a rocprofv3 --kernel-trace csv: count, mean, median, min, max in microseconds and
the share of the summed kernel time -- the grid (workgroups x, y) tells the layers / levels of one template instance apart
(level-1 refiner tail: 960 x b, fused level-1 front: 480 b, ...).  usage: trace_table.py <kernel_trace.csv> [min_share_%] [name filter]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
min_share = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
flt = sys.argv[3] if len(sys.argv) > 3 else ""
groups = defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if flt and flt not in name:
        continue
    wg = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
    groups[(name, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
total = sum(sum(v) for v in groups.values())
if not groups or total <= 0:
    sys.exit("trace_table: no kernel rows in " + sys.argv[1])
print(f"{'kernel':58s} {'grid':>11s} {'launches':>8s} {'mean us':>9s} {'median':>9s} {'min':>8s} {'max':>8s} {'share':>6s}")
for (name, wg), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    share = 100 * sum(v) / total
    if share < min_share:
        continue
    v = sorted(v)
    print(f"{name[:58]:58s} {wg[0]:>6d}x{wg[1]:<4d} {len(v):8d} {sum(v) / len(v):9.2f} {v[len(v) // 2]:9.2f} {v[0]:8.2f} {v[-1]:8.2f} {share:5.1f}%")
print(f"total kernel time {total / 1e3:.3f} ms in {sum(len(v) for v in groups.values())} launches")
