"""One steady-state step of a rocprofv3 kernel trace with the HSA queue each kernel was dispatched on: start / end (us from the
step's first kernel), queue, workgroups, kernel.  Kernels that share a queue run in submission order whatever the graph's
dependencies say -- this is how to see a branch of the captured graph waiting behind another one."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("columns:", [c for c in rows[0].keys()], file=sys.stderr)
# a step ends with the metric pass's finalisation (one per step, launched after the graph): the window is one whole step
ends = [i for i, r in enumerate(rows) if "metrics_finalize_kernel" in r["Kernel_Name"]]
i0, i1 = ends[-4] + 1, ends[-3] + 1
t0 = int(rows[i0]["Start_Timestamp"])
qs = {}
for r in rows[i0:i1]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]))
    nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); nm = re.sub(r"\(.*$", "", nm.replace("void ", ""))[:44]
    q = r.get("Queue_Id", "?")
    qi = qs.setdefault(q, len(qs))
    st = r.get("Stream_Id", "")
    print(f"{s:8.1f} -> {e:8.1f}  dur {e - s:6.1f}  q{qi} s{st:>3s}  wg {wg:5d}  {nm}")
print("queues:", qs, file=sys.stderr)
