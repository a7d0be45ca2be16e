import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import m4depth_amd as M
from m4depth_amd import synthetic as S
from helpers import to_dev
dev = torch.device("cuda:0")
L, H, Wd, T, b = 4, 64, 128, 3, 2
W = S.init_weights(L, seed=5)
samples, cam = S.make_sequence(b, T, H, Wd, seed=99)
model = M.M4Depth(nbre_levels=L); model.load_numpy_weights(W, dev)
ds, dc = to_dev(samples, dev), to_dev(cam, dev)
a = model([ds, dc])["depth"].clone(); ea = [[{k: v.clone() for k, v in l.items()} for l in st] for st in model.last_estimates]
model.reset_state()
b2 = model([ds, dc])["depth"].clone()
print("seq vs seq equal:", torch.equal(a, b2), float((a - b2).abs().max()))
model.reset_state()
ests = []
for s in ds:
    last = model([[s], dc])["depth"]
    ests.append([{k: v.clone() for k, v in l.items()} for l in model.last_estimates[0]])
print("seq vs stream equal:", torch.equal(a, last), float((a - last).abs().max()))
for t in range(T):
    for l in range(L):
        for k in ("depth", "parallax", "other"):
            d = float((ea[t][l][k] - ests[t][l][k]).abs().max())
            if d != 0: print("t", t, "lvl", l, k, "maxdiff", d)
# encoder determinism
f1 = model.encoder(ds[1]["RGB_im"]); f2 = model.encoder(ds[1]["RGB_im"])
print("encoder repeat equal:", all(torch.equal(x, y) for x, y in zip(f1, f2)))
fa = [model.encoder(s["RGB_im"]) for s in ds]
print("encoder list vs single:", all(torch.equal(x, y) for x, y in zip(fa[1], f1)))
