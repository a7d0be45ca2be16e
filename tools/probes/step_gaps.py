"""Per step of a rocprofv3 kernel trace (a step ends with metrics_finalize_kernel): span first kernel -> last kernel, idle gap to the
next step's first kernel, and the largest idle stretches INSIDE the step (no kernel of any queue in flight)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "metrics_finalize_kernel" in r["Kernel_Name"]]
for k in range(len(ends) - 6, len(ends) - 1):
    seg = rows[ends[k] + 1: ends[k + 1] + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    prev_end = int(rows[ends[k]]["End_Timestamp"])
    span = (int(seg[-1]["End_Timestamp"]) - t0) / 1e3
    # idle stretches inside
    cur_end, idle = t0, []
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s > cur_end:
            idle.append(((s - cur_end) / 1e3, (cur_end - t0) / 1e3, r["Kernel_Name"].split("(")[-2][-30:] if False else r["Kernel_Name"][:60]))
        cur_end = max(cur_end, e)
    idle.sort(reverse=True)
    print(f"step {k}: gap before {(t0 - prev_end) / 1e3:7.1f} us, span {span:8.1f} us, idle inside {sum(i[0] for i in idle):7.1f} us; largest: "
          + "; ".join(f"{d:.1f} us at {at:.0f} before {nm.replace('(anonymous namespace)::', '')[:28]}" for d, at, nm in idle[:5]))
