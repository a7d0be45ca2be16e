"""GPU-busy fraction of the densest <ms> milliseconds of a rocprofv3 kernel trace (graph replays of bench.py run with
--no-kernel-timing): union of the kernel intervals / window, the sum of kernel durations / window (average number of
kernels in flight), and the time with exactly one kernel in flight whose grid is smaller than the chip."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ms = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"])), r["Kernel_Name"]) for r in rows)
# the window of <ms> milliseconds holding the most launches (the timed graph replays), found by a sliding count
starts = [e[0] for e in ev]
w = int(ms * 1e6)
best, best_i, j = -1, 0, 0
for i in range(len(starts)):
    while starts[j] < starts[i] - w: j += 1
    if i - j > best: best, best_i = i - j, i
t_end = starts[best_i]
t0 = t_end - w
ev = [(s, min(e, t_end), wg, n) for s, e, wg, n in ev if e > t0 and s < t_end]
span = t_end - t0
busy = 0; cur_s, cur_e = None, None
for s, e, _, _ in ev:
    s = max(s, t0)
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - max(s, t0) for s, e, _, _ in ev)
# sweep: time with only small kernels in flight (every running kernel has < 256 workgroups)
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 256
pts = []
for i, (s, e, wg, _) in enumerate(ev):
    pts.append((max(s, t0), 1, wg, i)); pts.append((e, -1, wg, i))
pts.sort()
small_only = 0; n_big = 0; last = t0
active = set()
import collections, re
blame = collections.Counter()
for tm, d, wg, i in pts:
    if active and n_big == 0:
        small_only += tm - last
        for k in active: blame[re.sub(r"\(.*$", "", ev[k][3].replace("(anonymous namespace)::", "").replace("void ", ""))[:60]] += (tm - last) / len(active)
    last = tm
    if d > 0: active.add(i)
    else: active.discard(i)
    if wg >= thr: n_big += d
for k, v in blame.most_common(14): print(f"   {v / 1e3:9.1f} us  {k}")
print(f"window {ms:.0f} ms: {len(ev)} launches; busy (union) {100 * busy / span:.1f} %; sum of durations / window {tot / span:.2f}; "
      f"only sub-chip kernels (< {thr} workgroups) in flight {100 * small_only / span:.1f} %; idle {100 * (span - busy) / span:.1f} %")
