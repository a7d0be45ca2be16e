"""Absolute timeline of one steady-state step of a rocprofv3 kernel trace: start / end (us from the step's first kernel),
duration, workgroups, kernel -- for the small kernels of the coarse-level chains (to see real gaps between dependent launches)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
heads = [i for i, r in enumerate(rows) if "enc0_stats_kernel" in r["Kernel_Name"] or "enc0_rgb_total_kernel" in r["Kernel_Name"] or "enc_head_conv_kernel" in r["Kernel_Name"]]
# a step = two encoder batches; take the third-last complete step
n_fin = sum(1 for r in rows if "metrics_finalize_kernel" in r["Kernel_Name"])
per_step = max(1, round(len(heads) / n_fin)) if n_fin else 2          # head kernels per step (1 with the statistics up front)
i0, i1 = heads[-(3 * per_step + 1)], heads[-(2 * per_step + 1)]
t0 = int(rows[i0]["Start_Timestamp"])
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if s > lim: break
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]))
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); nm = re.sub(r"\(.*$", "", nm.replace("void ", ""))[:48]
    print(f"{s:8.1f} -> {e:8.1f}  dur {e - s:6.1f}  wg {wg:5d}  {nm}")
