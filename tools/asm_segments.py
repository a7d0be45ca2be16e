"""Instruction mix of a kernel's ISA (hipcc -save-temps .s file) between consecutive s_barrier instructions."""
import sys
path, name = sys.argv[1], sys.argv[2]
s = open(path).read()
i = s.index(name + ":")
j = s.index(".Lfunc_end", i)
segs, cur = [], []
for l in s[i:j].split("\n"):
    l = l.strip()
    if not l or l.startswith(";") or (l.startswith(".") and not l.startswith(".LBB")):
        continue
    cur.append(l)
    if l.startswith("s_barrier"):
        segs.append(cur); cur = []
segs.append(cur)
for k, sg in enumerate(segs):
    cnt = lambda f: sum(1 for l in sg if f(l))
    print(k, len(sg), "mfma", cnt(lambda l: l.startswith("v_mfma")), "valu", cnt(lambda l: l.startswith("v_") and not l.startswith("v_mfma")),
          "ds", cnt(lambda l: l.startswith("ds_")), "waitcnt", cnt(lambda l: l.startswith("s_waitcnt")),
          "accvgpr", cnt(lambda l: l.startswith("v_accvgpr")), "vmem", cnt(lambda l: "load" in l and ("global" in l or "buffer" in l)),
          "labels", [l for l in sg if l.startswith(".LBB")][:4])
