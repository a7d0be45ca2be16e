import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)
import m4depth_amd as M
from m4depth_amd import synthetic as S
model = M.M4Depth(nbre_levels=6); model.load_numpy_weights(S.init_weights(6), dev)
def rep(fn, n=4):
    ys = [fn() for _ in range(n)]
    return all(torch.equal(ys[0], y) for y in ys[1:])
for (H, Wd, b) in [(64, 128, 2), (384, 1280, 1)]:
    x = torch.rand(b, H, Wd, 3, device=dev)
    enc = model.encoder
    fm = x
    for i in range(6):
        if min(fm.shape[1:3]) < 2: break
        c1 = enc.conv_layers_s1[i]; c2 = enc.conv_layers_s2[i]
        d1 = rep(lambda: c1(fm)); t = c1(fm)
        if i == 0:
            dd = rep(lambda: enc.dn_layers[0](t)); t = enc.dn_layers[0](t)
            print(H, "lvl", i, "DINL det", dd)
        t = F.leaky_relu(t, 0.1)
        d2 = rep(lambda: c2(t)); fm2 = F.leaky_relu(c2(t), 0.1)
        print(H, "lvl", i, "in", tuple(fm.shape), "s1 det", d1, "s2 det", d2, flush=True)
        fm = fm2
    # refiner convs at each level
    for lvl in model.d_estimator.levels:
        h, w = H >> lvl.lvl_depth, Wd >> lvl.lvl_depth
        if min(h, w) < 2: continue
        xi = torch.randn(b, h, w, lvl.f_in, device=dev)
        convs = list(lvl.disp_refiner.prep_conv_layers) + list(lvl.disp_refiner.est_d_conv_layers)
        res = []
        for cv in convs:
            res.append(rep(lambda: cv(xi))); xi = F.leaky_relu(cv(xi), 0.1)
        print(H, "refiner lvl", lvl.lvl_depth, (h, w), res, flush=True)
