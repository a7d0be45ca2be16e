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
