"""Timeline of the last N kernel launches of a rocprofv3 kernel trace (multi-stream graphs: launches overlap,
so per-kernel sums say little): start / end relative to the first listed launch, queue id, short name."""
import csv, re, sys
path, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
with open(out, "w") as fh:
    for r in rows:
        nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
        nm = re.sub(r"\(.*$", "", nm)[:60]
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        fh.write(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  q{r['Queue_Id']:>3}  grid {r['Grid_Size_X']:>7}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}  {nm}\n")
