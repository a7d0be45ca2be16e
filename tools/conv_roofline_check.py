"""rocprofv3 durations of the roofline kernel (level-1 128->128 MFMA convolution) from a kernel trace of
`bench.py --steps 10 --warmup 3`, to set beside the HIP-event average bench.py prints.  The template instance
conv3x3_mfma_kernel<4,3,1,...> serves both wide level-1 layers (64->128 and 128->128, same grid), so the
--stats average of the NAME mixes them; this separates the two by duration and isolates the last 15 launches
of the 128->128 layer = bench.py's eager kernel-timing pass (5 steps x 3 full frames, one stream)."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "conv3x3_mfma_kernel<4, 3, 1" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
big = [x for x in d if x > 250]
small = [x for x in d if x <= 250]
print("conv3x3_mfma_kernel<4,3,1,...> launches:", len(d))
print(f"  128->128 layer (> 250 us): n={len(big)}  mean {sum(big) / len(big):.2f} us  min {min(big):.2f}  max {max(big):.2f}"
      "   [graph replays overlap frames on several streams: individual launches stretch]")
print(f"   64->128 layer (<= 250 us): n={len(small)}  mean {sum(small) / len(small):.2f} us")
last = big[-15:]
print(f"  128->128 layer, last 15 launches (the eager, single-stream kernel-timing pass bench.py brackets with HIP events): "
      f"mean {sum(last) / len(last):.2f} us  min {min(last):.2f}  max {max(last):.2f}")
