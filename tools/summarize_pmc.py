"""Per-kernel averages of a rocprofv3 counter_collection.csv (hand-written kernels only)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name", "")
    if "anonymous namespace" not in name:
        continue
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in d.items():
        print(f"{k:34s} {c:24s} launches {len(v):4d}  mean {sum(v) / len(v):16.1f}")
