"""The C-ABI shared library loads on a GPU-less host and exports every symbol that
include/m4depth_hip.h declares; argument validation returns hipErrorInvalidValue
without touching a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(name="m4depth_hip.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(m4d_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from m4depth_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 19
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/m4depth_hip.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == syms, "ctypes binding and header disagree"
    assert _lib.lib.m4d_abi_version() == _lib.ABI_VERSION == 6
    assert "gfx950" in _lib.build_info()
    # the experiments header: its symbols are exported by an EXPERIMENTS=1 build and ONLY by it, and the binding knows them all
    exp = header_symbols("m4depth_hip_experiments.h")
    assert sorted(_lib.EXPERIMENT_SYMBOLS) == exp and not set(exp) & set(syms)
    assert _lib.has_experiments == ("+experiments" in _lib.build_info())
    for s in exp:
        assert hasattr(raw, s) == _lib.has_experiments, f"{s}: experiments symbol {'missing from' if _lib.has_experiments else 'present in'} this build"


def test_product_library_exports_only_the_header(tmp_path):
    """Every m4d_* symbol the shared library exports is declared in include/m4depth_hip.h (or, in an experiments build, in
    include/m4depth_hip_experiments.h): no undeclared entry points, no experiment kernels behind global setters in the
    product library."""
    import subprocess
    from m4depth_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("m4d_")})
    allowed = set(header_symbols())
    if _lib.has_experiments:
        allowed |= set(header_symbols("m4depth_hip_experiments.h"))
    extra = [e for e in exported if e not in allowed]
    assert not extra, f"exported but not declared: {extra}"


def test_argument_validation_without_gpu():
    from m4depth_amd._lib import lib
    dims = (ctypes.c_int * 6)(1, 4, 4, 1, 1, 3)
    assert lib.m4d_backproject_fwd(None, None, dims, None, None) == 1
    assert lib.m4d_dense_image_warp(None, None, 1, 4, 4, 1, None, None, None) == 1
    assert lib.m4d_sncv_fwd(None, None, 1, 4, 4, 4, 1, 1, 1, None, 9, None) == 1
    assert lib.m4d_normalize_cuts(None, 1, 4, 4, 4, 1, None, None) == 1
    assert lib.m4d_bias_act(None, None, 4, 4, 0.1, None, None) == 1
    # the staggered first round is a validated per-launch argument (ABI 6), not library state
    assert lib.m4d_conv3x3_wino6_bias_act_ks(None, None, None, 1, 16, 16, 32, 64, 64, 0.1, None, 0, 0, 0, None) == 1
    assert not hasattr(ctypes.CDLL(__import__("m4depth_amd")._lib.LIB_PATH), "m4d_wino6_set_stagger")
    fake = ctypes.c_void_p(4096)           # never dereferenced: validation fails first
    assert lib.m4d_dense_image_warp(fake, fake, 1, 1, 4, 1, fake, None, None) == 1          # H < 2
    assert lib.m4d_sncv_fwd(fake, fake, 1, 4, 4, 6, 1, 1, 4, fake, 36, None) == 1            # cuts do not divide C
    assert lib.m4d_dscv_fwd(fake, fake, fake, fake, fake, 5, fake, fake, fake, 1, 4, 4, 4, 1, 1, 0,
                            fake, 3, None, None, 0, 1.0, None, None) == 1                    # rot_c not in {3,4}


def test_missing_library_is_loud(monkeypatch, tmp_path):
    import importlib
    from m4depth_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU / PyTorch fallback"):
        _lib._load()


def test_cpu_tensor_is_rejected():
    import torch
    import m4depth_amd as M
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        M.dense_image_warp(torch.zeros(1, 4, 4, 1), torch.zeros(1, 4, 4, 2))


def test_launch_tape_bookkeeping_without_gpu():
    """m4d_tape_begin / _end / _length / _free (csrc/m4d_tape.hip): an empty recording is a valid tape of length 0, nested
    recordings are refused, unknown ids are errors -- no device involved."""
    from m4depth_amd import _lib
    from m4depth_amd._lib import lib
    if not _lib.has_experiments:
        pytest.skip("the launch tape is an experiment: make EXPERIMENTS=1 (include/m4depth_hip_experiments.h)")
    t = lib.m4d_tape_begin()
    assert t >= 0
    assert lib.m4d_tape_begin() == -1                       # this thread is already recording
    assert lib.m4d_tape_replay(t, None) != 0                # a tape cannot be replayed while it records
    assert lib.m4d_tape_end() == 0 and lib.m4d_tape_end() == -1
    assert lib.m4d_tape_length(t) == 0 and lib.m4d_tape_length(t + 1000) == -1
    assert lib.m4d_tape_free(t) == 0 and lib.m4d_tape_free(t) != 0 and lib.m4d_tape_length(t) == -1
