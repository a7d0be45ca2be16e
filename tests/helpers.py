"""Shared helpers for the parity tests."""
import numpy as np

F = np.float32


def camera_np(b, h, w):
    return {"f": np.tile(np.array([[0.5 * w, 0.5 * h]], F), [b, 1]),
            "c": np.tile(np.array([[0.5 * w, 0.5 * h]], F), [b, 1])}


def motion_np(rng, b, quat=True, t_scale=(1.0, 1.0, 1.0)):
    aa = rng.normal(0.0, 0.02, [b, 3])
    if quat:
        ang = np.linalg.norm(aa, axis=1, keepdims=True)
        rot = np.concatenate([np.cos(ang / 2), aa / np.maximum(ang, 1e-12) * np.sin(ang / 2)], axis=1).astype(F)
    else:
        rot = aa.astype(F)
    trans = (rng.normal([0.0, 0.0, 0.3], 0.05, [b, 3]) * np.array(t_scale)).astype(F)
    return rot, trans


class torch_convolutions_on_cpu:
    """Host-logic tests only (no GPU): patches a plain-torch stand-in over ``_Conv3x3SameTF.forward`` -- which in the
    product has no CPU form and raises -- so that the layer wiring of the encoder / refiner modules, ``load_hwio`` and the
    TF 'SAME' ``same_pads`` rule can be checked against the oracle on CPU tensors.  Test infrastructure, not a fallback."""

    def __enter__(self):
        import torch.nn.functional as TF
        from m4depth_amd import network as N
        self._cls, self._old = N._Conv3x3SameTF, N._Conv3x3SameTF.forward

        def forward(conv, x_nhwc, slope=None, final=True):
            assert not x_nhwc.is_cuda
            x = x_nhwc.permute(0, 3, 1, 2)
            (pt, pb), (pl, pr) = conv.same_pads(*x.shape[2:])
            y = TF.conv2d(TF.pad(x, (pl, pr, pt, pb)), conv.weight, conv.bias, conv.stride, 0).permute(0, 2, 3, 1).contiguous()
            return y if slope is None else TF.leaky_relu(y, slope)
        N._Conv3x3SameTF.forward = forward
        self._dn_old = N.DomainNormalization.forward

        def dn_forward(dn, f_map, slope=1.0):           # the same for DomainNormalization (GPU-only in the product)
            assert not f_map.is_cuda
            if dn.scale is None:
                dn._build(f_map.shape[-1], f_map.device)
            from m4depth_amd.training import dinl_autograd
            out = dinl_autograd(dn, f_map)
            return out if slope == 1.0 else TF.leaky_relu(out, slope)
        N.DomainNormalization.forward = dn_forward
        return self

    def __exit__(self, *exc):
        from m4depth_amd import network as N
        self._cls.forward = self._old
        N.DomainNormalization.forward = self._dn_old
        return False


def to_dev(x, dev):
    import torch
    if isinstance(x, dict):
        return {k: to_dev(v, dev) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [to_dev(v, dev) for v in x]
    if isinstance(x, np.ndarray):
        if x.dtype == np.bool_:
            return torch.from_numpy(x)          # new_traj stays on the host: it steers control flow
        return torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    return x


def npy(t):
    return t.detach().cpu().numpy()


def rel_err(a, b, floor=1e-30):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a, F)
    b = np.ascontiguousarray(b, F)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    if not same.all():
        bad = np.argwhere(~same)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} / {a.size} elements differ bitwise; first at {i}: {a[i]!r} vs {b[i]!r}")


def make_fake_dataset(root, kind, n_traj=2, n_frames=7, size=(48, 64), seed=0):
    """A miniature dataset in the on-disk format of the reference's dataloaders: JPEG frames, the
    dataset's ground-truth encoding, tab-separated trajectory csv files.  Returns (db_path, records_path)."""
    import os
    from PIL import Image
    rng = np.random.default_rng(seed)
    h, w = size
    db = os.path.join(root, "db")
    rec = os.path.join(root, "records")
    for t in range(n_traj):
        tdir = os.path.join(db, f"traj_{t}")
        os.makedirs(os.path.join(tdir, "color"), exist_ok=True)
        os.makedirs(os.path.join(tdir, "gt"), exist_ok=True)
        os.makedirs(os.path.join(rec, f"set_{t}"), exist_ok=True)
        cols = ["id", "camera_l", "disp" if kind == "midair" else "depth", "qw", "qx", "qy", "qz", "tx", "ty", "tz"]
        if kind == "kitti-raw":
            cols += ["fx", "fy", "cx", "cy"]
        lines = ["\t".join(cols)]
        for i in range(n_frames):
            img = (rng.random([h, w, 3]) * 255).astype(np.uint8)
            img[: h // 8, : w // 8] = 0                                # a black corner (TartanAir mask)
            cpath = os.path.join(f"traj_{t}", "color", f"{i:06d}.JPEG")
            Image.fromarray(img).save(os.path.join(db, cpath), format="JPEG", quality=95)
            depth = rng.uniform(2.0, 60.0, [h, w]).astype(np.float32)
            if kind == "midair":
                gpath = os.path.join(f"traj_{t}", "gt", f"{i:06d}.PNG")
                bits = (np.float32(512.0) / depth).astype(np.float16).view(np.uint16)
                Image.fromarray(bits).save(os.path.join(db, gpath), format="PNG")
            elif kind == "kitti-raw":
                gpath = os.path.join(f"traj_{t}", "gt", f"{i:06d}.png")
                sparse = np.where(rng.random([h, w]) > 0.7, depth, 0.0)
                Image.fromarray((sparse * 256.0).astype(np.uint16)).save(os.path.join(db, gpath), format="PNG")
            else:
                gpath = os.path.join(f"traj_{t}", "gt", f"{i:06d}.npy")
                np.save(os.path.join(db, gpath), depth)
            aa = rng.normal(0.0, 0.01, 3)
            ang = np.linalg.norm(aa)
            q = np.concatenate([[np.cos(ang / 2)], aa / max(ang, 1e-12) * np.sin(ang / 2)])
            tr = rng.normal([0.0, 0.0, 0.3], 0.05, 3)
            vals = [str(i), cpath, gpath] + [repr(float(v)) for v in q] + [repr(float(v)) for v in tr]
            if kind == "kitti-raw":
                vals += ["0.58", "1.92", "0.49", "0.51"]
            lines.append("\t".join(vals))
        with open(os.path.join(rec, f"set_{t}", f"traj_{t:04d}.csv"), "w") as fh:
            fh.write("\n".join(lines) + "\n")
    return db, rec


class hbm_pressure:
    """Streaming HBM traffic beside the kernels under test: ``queue(n)`` enqueues ``n`` device-to-device copies of ``mb`` MB on
    a side stream.  A kernel whose correctness leans on a memory latency (round 3: a hand-counted ``s_waitcnt vmcnt`` one
    DMA too loose in m4d_wino6.hip) passes on a quiet chip and fails beside this -- the regime of three concurrent
    batch-32 frame streams, and of the driver's box."""

    def __init__(self, dev, mb=512):
        import torch
        self.stream = torch.cuda.Stream(device=dev)
        n = mb * (1 << 20) // 4
        self.src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
        self.dst = torch.empty_like(self.src)
        torch.cuda.synchronize()

    def queue(self, n):
        import torch
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                self.dst.copy_(self.src)


def image_checksums(t):
    """One int64 per image of a [b, ...] float32 tensor: the sum of its bit patterns."""
    import torch
    return t.contiguous().view(torch.int32).reshape(t.shape[0], -1).sum(dim=1, dtype=torch.int64)


def assert_replicas_bitwise(t, uniq, what):
    """Images i and i + k * uniq of ``t`` ([b, ...], b a multiple of ``uniq``) carry the same bits; the message names the
    tensor, the replicas that differ and where the first one does."""
    import torch
    b = t.shape[0]
    want = t[:uniq].repeat(b // uniq, *([1] * (t.dim() - 1)))
    if torch.equal(t, want):
        return
    ne = (t.contiguous().view(torch.int32) != want.contiguous().view(torch.int32))
    bad = ne.reshape(b, -1).any(dim=1).nonzero().flatten().tolist()
    first = ne.nonzero()[0].tolist()
    raise AssertionError(f"{what}: replicas {bad[:8]}{'...' if len(bad) > 8 else ''} differ from their originals "
                         f"({int(ne.sum())} elements; first at {first})")
