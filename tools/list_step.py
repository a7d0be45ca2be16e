"""Lists every launch of the last steady-state step of a rocprofv3 kernel trace, in order:
index, duration, grid size, kernel (to see which individual launches a latency-bound step is made of)."""
import csv
import re
import sys


def main(path, out):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    n = len(names)
    period = None
    for trail in range(0, 400):
        m = n - trail
        for P in range(40, m // 2):
            if names[m - P:m] == names[m - 2 * P:m - P]:
                period = P
                break
        if period:
            rows = rows[:m]
            break
    if not period:
        print("no period")
        return
    step = rows[-period:]
    t_prev = int(rows[-period - 1]["End_Timestamp"])
    with open(out, "w") as fh:
        for i, r in enumerate(step):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            nm = re.sub(r"\(.*$", "", nm.replace("void ", ""))[:70]
            grid = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else r.get("Grid_Size", "?")
            fh.write(f"{i:4d} gap {(s - t_prev) / 1e3:7.2f} us  dur {(e - s) / 1e3:8.2f} us  grid {grid:>16}  {nm}\n")
            t_prev = e


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
