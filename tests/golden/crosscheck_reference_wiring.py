"""Cross-check of the oracle's WIRING against the reference's own Python source, in the build container only.

What this is.  The reference is Python on TensorFlow 2.7, which is not installed here, so nothing of it can run as shipped and
the parity grade stays "unpinned".  But `/root/reference/utils/depth_operations.py` and `utils/dense_image_warp.py` are plain
Python that only CALLS ~60 TensorFlow array ops.  This script imports those two files UNMODIFIED from `/root/reference` with a
minimal numpy-backed module named `tensorflow` in `sys.modules` (the stand-in below: reshape / concat / stack / transpose /
slicing / tile / pad / gather / elementwise float32 arithmetic / clip / cast / meshgrid ..., each a one-line numpy call) and
compares what the reference's functions return with `oracle/m4depth_oracle.py` on the committed golden inputs and on seeded
random inputs.

What it can and cannot show.  It checks every reshape, transposition, axis, concatenation order, sign, clamp and operand the
reference's source spells out -- e.g. the cut-major channel order of the DSCV (`depth_operations.py:278`) against the
displacement-major order of the SNCV (`:297-311`), `(proj + delta) - start` (`:264`), query = grid + flow
(`dense_image_warp.py:244`), the `[..., r:r+1]` style slices of `reproject` -- because those are executed from the reference's
text.  It does NOT pin TensorFlow's internal arithmetic: where the result depends on how TF evaluates an op internally the
stand-in takes the SAME choice the oracle documents as [UNPINNED] (matmul = products summed sequentially over k, one rounding per
operation, no FMA; reduce_mean = sequential sum in index order then one divide; the float16 reduce_mean of the DSCV = float16
products accumulated in float32, divided, rounded to float16 once; tf.norm = sqrt of the sequential sum of squares).  So a
bit-for-bit match here says "the restatement wires the same operands through the same operations in the same order as the
reference's source", nothing about TF's kernels.  A stand-in library pins nothing; DESIGN.md section 2 says so.

Round 6 extension: `m4depth_network.py` (FeaturePyramid incl. DomainNormalization, DispRefiner, DepthEstimatorLevel.call with its
temporal memory, DepthEstimatorPyramid.call, M4Depth.call) and `metrics.py` run the same way, with a dozen more ops and a
three-class stand-in for `tensorflow.keras` (Layer = build-once-then-call with add_weight; Conv2D = the oracle's own
conv2d_same, so the convolution ARITHMETIC is shared and what is compared is everything around it; metrics.Mean = total / count):
a 3-level model steps through a reset frame and two full frames with the oracle's weights, every level's depth / parallax /
other of every frame, the final depth and the seven metrics compared with oracle.M4Depth / oracle.metrics_batch.

Nothing of this travels: the script reads /root/reference (absent on the GPU box), is not imported by any `-m gpu` test, by
smoke() or by bench.py; `tests/test_oracle.py::test_reference_wiring_crosscheck` runs it when /root/reference exists and
skips otherwise.  Run by hand:  python tests/golden/crosscheck_reference_wiring.py  -> prints the report (also kept as
tests/golden/crosscheck_reference_wiring.txt)."""
import contextlib
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("M4D_REFERENCE_ROOT", "/root/reference")
F = np.float32


# ----------------------------------------------------------------------------------------------------------------------------
# the numpy-backed stand-in for the handful of TensorFlow calls the two reference files make
# ----------------------------------------------------------------------------------------------------------------------------
class _Shape:
    def __init__(self, s):
        self._s = list(s)

    def as_list(self):
        return list(self._s)

    def __str__(self):
        return str(tuple(self._s))


def _np(x, dtype=None):
    """Anything the reference hands to an op -> ndarray (python floats become float32, as in TF)."""
    if isinstance(x, T):
        a = x.a
    elif isinstance(x, (list, tuple)):
        a = np.stack([_np(e) for e in x]) if len(x) and any(isinstance(e, (T, list, tuple, np.ndarray)) for e in x) else np.asarray(x)
    else:
        a = np.asarray(x)
    if a.dtype == np.float64 and not isinstance(x, (T, np.ndarray)):
        a = a.astype(F)
    if a.dtype == np.int64 and not isinstance(x, (T, np.ndarray)):
        a = a.astype(np.int32)
    return a if dtype is None else a.astype(dtype)


def _like(other, x):
    """A python scalar / list operand takes the tensor operand's dtype (TF's weak typing of constants)."""
    if isinstance(x, T):
        return x.a
    a = _np(x)
    if a.dtype != other.dtype and not isinstance(x, np.ndarray):
        a = a.astype(other.dtype)
    return a


def _matmul(a, b):
    """[..., m, k] @ [..., k, n]: products summed SEQUENTIALLY over k, one float32 rounding per operation (the oracle's
    [UNPINNED] choice for TF's batched matmul)."""
    a, b = _np(a), _np(b)
    out = a[..., :, 0:1] * b[..., 0:1, :]
    for k in range(1, a.shape[-1]):
        out = out + a[..., :, k:k + 1] * b[..., k:k + 1, :]
    return T(out)


class T:
    """A tensor: an ndarray with TF's method names.  Every arithmetic operator is ONE numpy operation (one rounding)."""
    __array_priority__ = 1000

    def __init__(self, a):
        self.a = a.a if isinstance(a, T) else np.asarray(a)

    dtype = property(lambda self: self.a.dtype)
    shape = property(lambda self: tuple(self.a.shape))

    def get_shape(self):
        return _Shape(self.a.shape)

    def numpy(self):
        return self.a

    def __getitem__(self, idx):
        return T(self.a[idx])

    def __len__(self):
        return len(self.a)

    def __iter__(self):
        return (T(v) for v in self.a)

    def __index__(self):
        return int(self.a)

    def __int__(self):
        return int(self.a)

    def __float__(self):
        return float(self.a)

    def __neg__(self):
        return T(-self.a)

    def __add__(self, o):
        return T(self.a + _like(self.a, o))

    def __radd__(self, o):
        return T(_like(self.a, o) + self.a)

    def __sub__(self, o):
        return T(self.a - _like(self.a, o))

    def __rsub__(self, o):
        return T(_like(self.a, o) - self.a)

    def __mul__(self, o):
        return T(self.a * _like(self.a, o))

    def __rmul__(self, o):
        return T(_like(self.a, o) * self.a)

    def __truediv__(self, o):
        return T(self.a / _like(self.a, o))

    def __rtruediv__(self, o):
        return T(_like(self.a, o) / self.a)

    def __pow__(self, p):
        if p == 2:
            return T(self.a * self.a)                 # x ** 2 = x * x (one rounding)
        raise NotImplementedError(f"pow {p}")

    def __matmul__(self, o):
        return _matmul(self, o)

    def __bool__(self):
        return bool(self.a)

    def copy(self):
        return T(self.a.copy())


class _Var(T):
    """tf.Variable / Layer.add_weight: a tensor with assign()."""

    def __init__(self, a):
        self.a = np.array(_np(a))

    def assign(self, v):
        self.a = np.array(_np(v), dtype=self.a.dtype).reshape(self.a.shape)
        return self

    def assign_add(self, v):
        self.a = self.a + np.asarray(_np(v), self.a.dtype)
        return self


def _reduce_mean(x, axis=None):
    a = _np(x)
    ax = axis % a.ndim
    n = a.shape[ax]
    acc_t = F                                         # float16 input (the DSCV): float32 accumulation, one rounding to half
    acc = np.take(a, 0, axis=ax).astype(acc_t)
    for i in range(1, n):
        acc = acc + np.take(a, i, axis=ax).astype(acc_t)
    return T((acc / acc_t(n)).astype(a.dtype))


def _norm(x, axis=None):
    a = _np(x)
    ax = axis % a.ndim
    acc = np.take(a, 0, axis=ax) * np.take(a, 0, axis=ax)
    for i in range(1, a.shape[ax]):
        v = np.take(a, i, axis=ax)
        acc = acc + v * v
    return T(np.sqrt(acc))


def _shape_arg(s):
    s = _np(s)
    return [int(v) for v in np.atleast_1d(s)]


def _slice(x, begin, size):
    a = _np(x)
    idx = tuple(slice(b, None if s == -1 else b + s) for b, s in zip(begin, size))
    return T(a[idx])


def _cast(x, dtype):
    return T(_np(x).astype(np.dtype(dtype)))


def _range(*args, dtype=None):
    vals = [int(v) if isinstance(v, T) else v for v in args]
    a = np.arange(*vals)
    if dtype is not None:
        a = a.astype(np.dtype(dtype))
    elif a.dtype == np.float64:
        a = a.astype(F)
    elif a.dtype == np.int64:
        a = a.astype(np.int32)
    return T(a)


def _meshgrid(a, b):
    gx, gy = np.meshgrid(_np(a), _np(b))              # TF's default indexing is 'xy', as numpy's
    return T(gx), T(gy)


@contextlib.contextmanager
def _scope(*_a, **_k):
    yield


def _axes(axis, ndim):
    if axis is None:
        return tuple(range(ndim))
    return tuple(a % ndim for a in (axis if isinstance(axis, (list, tuple)) else [axis]))


def _mean_axes(x, axis=None, keepdims=False, name=None):
    """tf.math.reduce_mean over several axes (DomainNormalization): numpy's float32 mean -- the oracle's [UNPINNED] choice."""
    a = _np(x)
    return T(a.mean(axis=_axes(axis, a.ndim), keepdims=keepdims, dtype=F))


def _variance(x, axis=None, keepdims=False, name=None):
    a = _np(x)
    m = a.mean(axis=_axes(axis, a.ndim), keepdims=True, dtype=F)
    return T(((a - m) * (a - m)).mean(axis=_axes(axis, a.ndim), keepdims=keepdims, dtype=F))


def _l2_normalize(x, axis=-1, epsilon=1e-12):
    a = _np(x)
    ss = (a * a).sum(axis=axis, keepdims=True, dtype=F)
    return T(a * (F(1.0) / np.sqrt(np.maximum(ss, F(epsilon)))))        # x * rsqrt(max(sum x^2, eps))


def _linalg_normalize(x, axis=-1):
    """(x / norm, norm), norm = sqrt of the SEQUENTIAL sum of squares along ``axis`` (the oracle's [UNPINNED] choice)."""
    a = _np(x)
    nrm = np.expand_dims(_norm(a, axis).a, axis)
    with np.errstate(invalid="ignore", divide="ignore"):
        return T(a / nrm), T(nrm)


def _resize_bilinear_v1(x, size, align_corners=False):
    """tf.compat.v1.image.resize_bilinear, legacy coordinates: src = dst * (in / out), lower = floor, upper = min(ceil, in - 1),
    top + (bottom - top) * ylerp with top = tl + (tr - tl) * xlerp -- the legacy kernel's expressions, float32."""
    a = _np(x)
    b, ih, iw, c = a.shape
    oh, ow = int(size[0]), int(size[1])

    def axis(o_n, i_n):
        src = np.arange(o_n, dtype=F) * F(F(i_n) / F(o_n))
        lo = np.floor(src)
        return lo.astype(np.int64), np.minimum(np.ceil(src).astype(np.int64), i_n - 1), (src - lo).astype(F)
    ylo, yhi, yl = axis(oh, ih)
    xlo, xhi, xl = axis(ow, iw)
    tl, tr = a[:, ylo][:, :, xlo], a[:, ylo][:, :, xhi]
    bl, br = a[:, yhi][:, :, xlo], a[:, yhi][:, :, xhi]
    xl4, yl4 = xl.reshape(1, 1, -1, 1), yl.reshape(1, -1, 1, 1)
    top = tl + (tr - tl) * xl4
    bot = bl + (br - bl) * xl4
    return T((top + (bot - top) * yl4).astype(F))


def _resize_nearest(x, size, method=None):
    """tf.image.resize(..., NEAREST_NEIGHBOR) of TF2 (half-pixel centres): src = min(floor((dst + 0.5) * in / out), in - 1)."""
    a = _np(x)
    ih, iw = a.shape[1:3]
    oh, ow = int(size[0]), int(size[1])
    iy = np.minimum(np.floor((np.arange(oh, dtype=F) + F(0.5)) * F(F(ih) / F(oh))).astype(np.int64), ih - 1)
    ix = np.minimum(np.floor((np.arange(ow, dtype=F) + F(0.5)) * F(F(iw) / F(ow))).astype(np.int64), iw - 1)
    return T(a[:, iy][:, :, ix])


def _sum_all(x, axis=None):
    """tf.reduce_sum over everything (metrics.py): float64 accumulation, one rounding (the oracle's [UNPINNED] choice)."""
    if axis is not None:
        raise NotImplementedError
    return T(F(_np(x).astype(np.float64).sum()))


class _Layer:
    """tensorflow.keras.layers.Layer: build(input shape) once, then call()."""

    def __init__(self, trainable=True, name=None, **_k):
        self._built = False
        self.trainable = trainable

    def add_weight(self, name=None, shape=None, dtype="float32", initializer=None, trainable=True, **_k):
        return _Var(initializer(list(shape), np.dtype(dtype)))

    def add_loss(self, *_a, **_k):
        pass

    def build(self, input_shape):
        pass

    def __call__(self, *args, **kw):
        if not self._built:
            self.build(list(args[0].shape) if args and isinstance(args[0], T) else None)
            self._built = True
        return self.call(*args, **kw)


class _Conv2D(_Layer):
    """tensorflow.keras.layers.Conv2D(filters, 3, strides, padding='same'): kernel [3,3,Cin,Cout] / bias assigned from the
    oracle's weight dict; the arithmetic IS the oracle's conv2d_same (shared on purpose: the convolution's summation order is
    not what this script checks)."""

    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", **_k):
        super().__init__()
        assert kernel_size == 3 and padding == "same"
        self.filters, self.stride = filters, int(strides[0])
        self.kernel = self.bias = None

    def call(self, x):
        from oracle import m4depth_oracle as O
        return T(O.conv2d_same(_np(x), self.kernel, self.bias, self.stride))


class _Mean(_Layer):
    """tensorflow.keras.metrics.Mean."""

    def __init__(self, name=None, **_k):
        super().__init__()
        self.name, self.total, self.count = name, F(0.0), 0

    def update_state(self, value, sample_weight=None):
        self.total = F(self.total + F(_np(value)))
        self.count += 1

    def result(self):
        return T(F(self.total / F(max(self.count, 1))))


def build_tensorflow_stub():
    tf = types.ModuleType("tensorflow")
    tf.float16 = tf.half = np.float16
    tf.float32 = np.float32
    tf.float64 = np.float64
    tf.int32 = np.int32
    tf.function = lambda fn=None, **_k: fn if fn is not None else (lambda f: f)
    tf.name_scope = _scope
    tf.identity = lambda x, name=None: T(x)
    tf.stop_gradient = lambda x: T(x)
    tf.convert_to_tensor = lambda x, dtype=None: T(_np(x, dtype))
    tf.reshape = lambda x, shape, name=None: T(_np(x).reshape(_shape_arg(shape)))
    tf.concat = lambda vals, axis, name=None: T(np.concatenate([_np(v) for v in vals], axis=axis))
    tf.stack = lambda vals, axis=0, name=None: T(np.stack([_np(v) for v in vals], axis=axis))
    tf.unstack = lambda x, axis=0: [T(v) for v in np.moveaxis(_np(x), axis, 0)]
    tf.split = lambda x, num_or_size_splits, axis=0: [T(v) for v in np.split(_np(x), num_or_size_splits, axis=axis)]
    tf.expand_dims = lambda x, axis, name=None: T(np.expand_dims(_np(x), axis))
    tf.squeeze = lambda x, axis=None: T(np.squeeze(_np(x), axis=axis))
    tf.transpose = lambda x, perm=None: T(np.transpose(_np(x), perm))
    tf.reverse = lambda x, axis: T(np.flip(_np(x), axis=tuple(axis)))
    tf.tile = lambda x, multiples: T(np.tile(_np(x), _shape_arg(multiples)))
    tf.pad = lambda x, paddings: T(np.pad(_np(x), [tuple(p) for p in paddings]))
    tf.slice = _slice
    tf.ones = lambda shape, dtype=np.float32: T(np.ones(_shape_arg(shape), np.dtype(dtype)))
    tf.range = _range
    tf.meshgrid = _meshgrid
    tf.shape = lambda input=None, **_k: T(np.asarray(_np(input).shape, np.int32))
    tf.cast = _cast
    tf.sqrt = lambda x: T(np.sqrt(_np(x)))
    tf.divide = lambda a, b: T(_np(a) / _like(_np(a), b))
    tf.multiply = lambda a, b: T(_np(a) * _like(_np(a), b))
    tf.clip_by_value = lambda x, lo, hi: T(np.minimum(np.maximum(_np(x), _like(_np(x), lo)), _like(_np(x), hi)))
    tf.reduce_mean = _reduce_mean
    tf.norm = _norm
    tf.load_op_library = lambda path: (_ for _ in ()).throw(RuntimeError("no custom op in the stand-in"))
    tf.linalg = types.SimpleNamespace(matmul=_matmul)
    tf.nn = types.SimpleNamespace(leaky_relu=lambda x, alpha=0.2, name=None: T(np.where(_np(x) > 0, _np(x), _np(x) * _np(x).dtype.type(alpha))))
    tf.image = types.SimpleNamespace()                # (resize_bilinear: wrap_feature_block is dead code in the reference)
    tf.compat = types.SimpleNamespace(v1=types.SimpleNamespace(name_scope=_scope, image=types.SimpleNamespace(resize_bilinear=_resize_bilinear_v1)))
    # ---- m4depth_network.py / metrics.py
    tf.zeros = lambda shape, dtype=np.float32: T(np.zeros(_shape_arg(shape), np.dtype(dtype)))
    tf.zeros_initializer = lambda: (lambda shape, dtype=np.float32: np.zeros(shape, np.dtype(dtype)))
    tf.ones_initializer = lambda: (lambda shape, dtype=np.float32: np.ones(shape, np.dtype(dtype)))
    tf.Variable = lambda initial_value=None, trainable=False, **_k: _Var(initial_value)
    tf.exp = lambda x: T(np.exp(_np(x)))
    tf.square = lambda x: T(_np(x) * _np(x))
    tf.maximum = lambda a, b: T(np.maximum(_np(a), _like(_np(a), b)))
    tf.greater = lambda a, b: T(_np(a) > _like(_np(a), b))
    tf.reduce_sum = _sum_all
    tf.math = types.SimpleNamespace(
        reduce_mean=_mean_axes, reduce_variance=_variance, l2_normalize=_l2_normalize, log=lambda x: T(np.log(_np(x))),
        abs=lambda x: T(np.abs(_np(x))), squared_difference=lambda a, b: T((_np(a) - _np(b)) * (_np(a) - _np(b))),
        multiply_no_nan=lambda x, y: T(np.where(_np(y) != 0, _np(x), F(0.0)).astype(F)),
        less=lambda a, b: T(_np(a) < _like(_np(a), b)))
    tf.linalg.normalize = _linalg_normalize
    tf.image.resize = _resize_nearest
    tf.image.ResizeMethod = types.SimpleNamespace(NEAREST_NEIGHBOR="nearest")
    tf.GradientTape = _scope
    tf.summary = types.SimpleNamespace(image=lambda *a, **k: None)
    keras = types.ModuleType("tensorflow.keras")
    keras.layers = types.SimpleNamespace(Layer=_Layer, Conv2D=_Conv2D)
    keras.models = types.SimpleNamespace(Model=_Layer)
    keras.metrics = types.SimpleNamespace(Mean=_Mean)
    keras.initializers = types.SimpleNamespace(HeNormal=lambda: None)
    keras.regularizers = types.SimpleNamespace(L1=lambda l1=0.0: None, L2=lambda l2=0.0: (lambda w: 0.0))
    tf.keras = keras

    # tensorflow.python.{framework,ops}.* as dense_image_warp.py imports them (the TF-addons origin of that file)
    ops = types.ModuleType("tensorflow.python.framework.ops")
    ops.name_scope = _scope
    ops.convert_to_tensor = lambda x: T(_np(x))
    ops.control_dependencies = _scope
    ops.RegisterGradient = lambda name: (lambda fn: fn)
    constant_op = types.ModuleType("tensorflow.python.framework.constant_op")
    constant_op.constant = lambda v, dtype=None: T(np.asarray(v, np.dtype(dtype) if dtype is not None else F))
    dtypes = types.ModuleType("tensorflow.python.framework.dtypes")
    dtypes.int32 = np.int32
    array_ops = types.ModuleType("tensorflow.python.ops.array_ops")
    array_ops.shape = lambda x: [int(v) for v in _np(x).shape]
    array_ops.unstack = tf.unstack
    array_ops.expand_dims = tf.expand_dims
    array_ops.reshape = tf.reshape
    array_ops.stack = tf.stack
    array_ops.meshgrid = _meshgrid
    array_ops.gather = lambda params, indices: T(_np(params)[_np(indices)])
    math_ops = types.ModuleType("tensorflow.python.ops.math_ops")
    math_ops.cast = _cast
    math_ops.minimum = lambda a, b: T(np.minimum(_np(a), _np(b)))
    math_ops.maximum = lambda a, b: T(np.maximum(_np(a), _np(b)))
    math_ops.floor = lambda x: T(np.floor(_np(x)))
    math_ops.range = _range
    check_ops = types.ModuleType("tensorflow.python.ops.check_ops")
    python = types.ModuleType("tensorflow.python")
    framework = types.ModuleType("tensorflow.python.framework")
    opsmod = types.ModuleType("tensorflow.python.ops")
    framework.constant_op, framework.dtypes, framework.ops = constant_op, dtypes, ops
    opsmod.array_ops, opsmod.check_ops, opsmod.math_ops = array_ops, check_ops, math_ops
    python.framework, python.ops = framework, opsmod
    tf.python = python
    return {"tensorflow": tf, "tensorflow.keras": keras, "tensorflow.python": python, "tensorflow.python.framework": framework,
            "tensorflow.python.framework.constant_op": constant_op, "tensorflow.python.framework.dtypes": dtypes,
            "tensorflow.python.framework.ops": ops, "tensorflow.python.ops": opsmod,
            "tensorflow.python.ops.array_ops": array_ops, "tensorflow.python.ops.check_ops": check_ops,
            "tensorflow.python.ops.math_ops": math_ops}


def load_reference(network=False):
    """(depth_operations module, dense_image_warp module[, m4depth_network module, metrics module]) of the reference, imported
    unmodified under the stand-in."""
    if not os.path.isfile(os.path.join(REF, "utils", "depth_operations.py")):
        raise FileNotFoundError(REF)
    saved = {k: sys.modules.get(k) for k in list(build_tensorflow_stub()) + ["utils", "utils.dense_image_warp", "utils.depth_operations",
                                                                          "m4depth_network", "metrics"]}
    sys.modules.update(build_tensorflow_stub())
    try:
        pkg = types.ModuleType("utils")
        pkg.__path__ = [os.path.join(REF, "utils")]
        sys.modules["utils"] = pkg
        mods = []
        for name in ("dense_image_warp", "depth_operations"):
            spec = importlib.util.spec_from_file_location(f"utils.{name}", os.path.join(REF, "utils", f"{name}.py"))
            m = importlib.util.module_from_spec(spec)
            sys.modules[f"utils.{name}"] = m
            with contextlib.redirect_stdout(open(os.devnull, "w")):      # ("Could not import cuda Backproject Module ...")
                spec.loader.exec_module(m)
            if name == "dense_image_warp":
                pkg.dense_image_warp = m.dense_image_warp                # `from utils import dense_image_warp` = the function
            mods.append(m)
        for name in ("depth_operations",):                               # `from utils.depth_operations import *` of the network file
            pkg.__dict__.update({k: v for k, v in mods[1].__dict__.items() if not k.startswith("_")})
        if network:
            for name in ("m4depth_network", "metrics"):
                spec = importlib.util.spec_from_file_location(name, os.path.join(REF, f"{name}.py"))
                m = importlib.util.module_from_spec(spec)
                sys.modules[name] = m
                spec.loader.exec_module(m)
                mods.append(m)
            return mods[1], mods[0], mods[2], mods[3]
        return mods[1], mods[0]
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


# ----------------------------------------------------------------------------------------------------------------------------
# the comparison
# ----------------------------------------------------------------------------------------------------------------------------
def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view({2: np.uint16, 4: np.uint32, 8: np.uint64}[a.dtype.itemsize])


def _cmp(name, got, want, rows, tol=0.0):
    got = got.a if isinstance(got, T) else np.asarray(got)
    want = np.asarray(want)
    ok_shape = tuple(got.shape) == tuple(want.shape)
    if not ok_shape:
        rows.append((name, False, f"SHAPE {got.shape} vs {want.shape}"))
        return
    same = np.array_equal(_bits(got.astype(want.dtype)), _bits(want)) if got.dtype.kind == want.dtype.kind else np.array_equal(got, want)
    if same:
        rows.append((name, True, f"bit-identical  {tuple(want.shape)} {want.dtype}"))
        return
    both_nan = np.isnan(got.astype(np.float64)) & np.isnan(want.astype(np.float64))
    err = np.abs(got.astype(np.float64) - want.astype(np.float64)) / np.maximum(np.abs(want.astype(np.float64)), 1e-30)
    err[both_nan] = 0.0
    mx = float(np.nanmax(err))
    rows.append((name, mx <= tol, f"max rel. difference {mx:.3e} over {int((err > 0).sum())} of {err.size} elements (tolerance {tol:g})"))


def run_network(rows):
    """m4depth_network.py + metrics.py of the reference under the stand-in against oracle.M4Depth / oracle.metrics_batch: a
    3-level model, batch 2, one reset frame + two full frames, the oracle's He-normal weights."""
    sys.path.insert(0, ROOT)
    from oracle import m4depth_oracle as O
    from m4depth_amd import synthetic as S
    with contextlib.redirect_stdout(open(os.devnull, "w")):              # (the reference prints "Seq sample ..." while tracing)
        _, _, N, MT = load_reference(network=True)
        L, b, Tn, H, Wd = 3, 2, 3, 32, 48
        Wts = S.init_weights(L, seed=77)
        samples, cam = S.make_sequence(b, Tn, H, Wd, seed=78)
        model = N.M4Depth(nbre_levels=L)
        enc = model.encoder
        for i in range(L):
            enc.conv_layers_s1[i].kernel, enc.conv_layers_s1[i].bias = Wts[f"enc.s1.{i}.kernel"], Wts[f"enc.s1.{i}.bias"]
            enc.conv_layers_s2[i].kernel, enc.conv_layers_s2[i].bias = Wts[f"enc.s2.{i}.kernel"], Wts[f"enc.s2.{i}.bias"]
        for i, lvl in enumerate(model.d_estimator.levels):
            convs = list(lvl.disp_refiner.prep_conv_layers) + list(lvl.disp_refiner.est_d_conv_layers)
            for j, cv in enumerate(convs):
                cv.kernel, cv.bias = Wts[f"lvl.{i + 1}.conv.{j}.kernel"], Wts[f"lvl.{i + 1}.conv.{j}.bias"]
        # DomainNormalization's scale / bias are created by its build(): run the encoder once, then overwrite them
        tsamples = [{k: T(v) for k, v in s.items()} for s in samples]
        tcam = {k: T(v) for k, v in cam.items()}
        enc(tsamples[0]["RGB_im"])
        enc.dn_layers[0].scale.assign(Wts["enc.dn.0.scale"].reshape(1, 1, 1, -1))
        enc.dn_layers[0].bias.assign(Wts["enc.dn.0.bias"].reshape(1, 1, 1, -1))
        # the reference's own sequence handling: M4Depth.call on the list of frames (inference: the levels keep their memory)
        ref_out = model([tsamples, tcam], training=False)
        # ... and once more frame by frame through DepthEstimatorPyramid to get at every level's estimate
        model2 = N.M4Depth(nbre_levels=L)
        for a_, b_ in zip(model.encoder.conv_layers_s1 + model.encoder.conv_layers_s2, model2.encoder.conv_layers_s1 + model2.encoder.conv_layers_s2):
            b_.kernel, b_.bias = a_.kernel, a_.bias
        for la, lb in zip(model.d_estimator.levels, model2.d_estimator.levels):
            for a_, b_ in zip(list(la.disp_refiner.prep_conv_layers) + list(la.disp_refiner.est_d_conv_layers),
                              list(lb.disp_refiner.prep_conv_layers) + list(lb.disp_refiner.est_d_conv_layers)):
                b_.kernel, b_.bias = a_.kernel, a_.bias
        model2.encoder(tsamples[0]["RGB_im"])
        model2.encoder.dn_layers[0].scale.assign(Wts["enc.dn.0.scale"].reshape(1, 1, 1, -1))
        model2.encoder.dn_layers[0].bias.assign(Wts["enc.dn.0.bias"].reshape(1, 1, 1, -1))
        pyrs = [model2.encoder(s["RGB_im"]) for s in tsamples]
        ref_seq = model2.d_estimator(pyrs, tsamples, tcam, False)
        o_out, o_seq = O.M4Depth(Wts, L)(samples, cam)
        opyr = O.feature_pyramid(samples[1]["RGB_im"], Wts, L)
    for l in range(L):
        _cmp(f"FeaturePyramid level {l + 1} (incl. DomainNormalization at level 0)", pyrs[1][l], opyr[l], rows)
    for t in range(Tn):
        for l in range(L):
            for key in ("depth", "parallax", "other"):
                _cmp(f"DepthEstimatorPyramid frame {t} level {l + 1} {key}", ref_seq[t][l][key], o_seq[t][l][key], rows)
    _cmp("M4Depth.call depth (nearest x2 of the finest level)", ref_out["depth"], o_out["depth"], rows)
    # metrics.py through the clipping of test_step (m4depth_network.py:462-470)
    gt = np.clip(samples[-1]["depth"], 0.0, 80.0).astype(F)
    est = np.clip(o_out["depth"], 0.001, 80.0).astype(F)
    mets = [MT.AbsRelError(), MT.SqRelError(), MT.RootMeanSquaredError(), MT.RootMeanSquaredLogError(),
            MT.ThresholdRelError(1), MT.ThresholdRelError(2), MT.ThresholdRelError(3)]
    for m in mets:
        m.update_state(T(gt), T(est))
    want = O.metrics_batch(samples[-1]["depth"], o_out["depth"])
    for m, w_ in zip(mets, want):
        _cmp(f"metrics.py {m.name}", np.asarray(m.result().a, F).reshape(1), np.asarray(w_, F).reshape(1), rows, tol=2e-6)


def run(verbose=True):
    sys.path.insert(0, ROOT)
    from oracle import m4depth_oracle as O
    R, W = load_reference()
    rows = []
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "ops.npz")))
    cvg = dict(np.load(os.path.join(ROOT, "tests", "golden", "cost_volumes.npz")))

    def cam_t(cam):
        return {k: T(v) for k, v in cam.items()}

    rng = np.random.default_rng(606)

    def motion(b, quat=True):
        aa = rng.normal(0.0, 0.02, [b, 3])
        if quat:
            ang = np.linalg.norm(aa, axis=1, keepdims=True)
            rot = np.concatenate([np.cos(ang / 2), aa / np.maximum(ang, 1e-12) * np.sin(ang / 2)], axis=1).astype(F)
        else:
            rot = aa.astype(F)
        return rot, rng.normal([0.05, -0.03, 0.3], 0.05, [b, 3]).astype(F)

    def camera(b, h, w):
        return {"f": np.tile(np.array([[0.5 * w, 0.55 * h]], F), [b, 1]), "c": np.tile(np.array([[0.48 * w, 0.52 * h]], F), [b, 1])}

    # ---- dense_image_warp (the python path: backproject.so is absent, dense_image_warp.py:44-61) on the golden vectors
    out = W.dense_image_warp(T(g["warp_img"]), T(g["warp_flow"]))
    _cmp("dense_image_warp(golden ops.npz) vs golden output", out, g["warp_out"], rows)
    _cmp("dense_image_warp vs oracle.dense_image_warp", out, O.dense_image_warp(g["warp_img"], g["warp_flow"]), rows)

    for quat in (True, False):
        tag = "quaternion" if quat else "small-angle"
        b, h, w = 2, 7, 9
        rot, trans = motion(b, quat)
        cam = camera(b, h, w)
        _cmp(f"get_rot_mat [{tag}]", R.get_rot_mat(T(rot)), O.get_rot_mat(rot), rows)
        m = rng.standard_normal([b, h, w, 3]).astype(F)
        co, me = R.get_coords_2d(T(m), cam_t(cam))
        oc, om = O.get_coords_2d(b, h, w, cam)
        _cmp(f"get_coords_2d coords [{tag}]", co[..., 0] if co.a.ndim == 5 else co, oc.reshape(np.asarray(co.a[..., 0] if co.a.ndim == 5 else co.a).shape), rows)
        _cmp(f"get_coords_2d mesh [{tag}]", me, np.broadcast_to(om, me.a.shape), rows)
        depth = (2.0 + 30.0 * rng.random([b, h, w, 1])).astype(F)
        para = (0.3 + 5.0 * rng.random([b, h, w, 1])).astype(F)
        _cmp(f"parallax2depth [{tag}]", R.parallax2depth(T(para), T(rot), T(trans), cam_t(cam)), O.parallax2depth(para, rot, trans, cam), rows)
        _cmp(f"depth2parallax [{tag}]", R.depth2parallax(T(depth), T(rot), T(trans), cam_t(cam)), O.depth2parallax(depth, rot, trans, cam), rows)
        _cmp(f"prev_d2para [{tag}]", R.prev_d2para(T(depth), T(rot), T(trans), cam_t(cam)), O.prev_d2para(depth, rot, trans, cam), rows)
        rw, (ra, rb) = R.reproject(T(m), T(depth), T(rot), T(trans), cam_t(cam))
        ow, (oa, ob) = O.reproject(m, depth, rot, trans, cam)
        _cmp(f"reproject warped map [{tag}]", rw, ow, rows, tol=0.0)
        _cmp(f"reproject aux 0 (proj - rot coords) [{tag}]", ra, oa, rows)
        _cmp(f"reproject aux 1 (rot coords) [{tag}]", rb, ob, rows)
        _cmp(f"recompute_depth [{tag}]", R.recompute_depth(T(depth), T(rot), T(trans), cam_t(cam)), O.recompute_depth(depth, rot, trans, cam), rows)
    _cmp("tile_in_batch", R.tile_in_batch(T(g["warp_img"]), 5), O.tile_in_batch(g["warp_img"], 5), rows)

    # ---- the two cost volumes: golden inputs where the file has them, + seeded geometries (1 / 2 / 4 cuts, ranges 4 and 2)
    for (b, h, w, C, k, r, quat) in ((1, 6, 7, 16, 1, 4, True), (2, 5, 8, 32, 2, 4, False), (1, 6, 6, 96, 4, 2, True)):
        rot, trans = motion(b, quat)
        cam = camera(b, h, w)
        c1 = O.normalize_cuts(rng.standard_normal([b, h, w, C]).astype(F), k)
        c2 = O.normalize_cuts(rng.standard_normal([b, h, w, C]).astype(F), k)
        disp = (0.2 + 6.0 * rng.random([b, h, w, 1])).astype(F)
        dpt = (0.2 + 6.0 * rng.random([b, h, w, 1])).astype(F)
        rcv, rpd = R.get_parallax_sweeping_cv(T(c1), T(c2), T(dpt), T(disp), T(rot), T(trans), cam_t(cam), r, k)
        ocv, opd = O.get_parallax_sweeping_cv(c1, c2, dpt, disp, rot, trans, cam, r, k)
        _cmp(f"DSCV cost volume  C={C} cuts={k} range={r}", rcv, ocv, rows)
        _cmp(f"DSCV warped previous parallax  C={C} cuts={k} range={r}", rpd, opd, rows)
        for dil in (1, 2):
            _cmp(f"SNCV cost_volume  C={C} cuts={k} range=3 dilation={dil}",
                 R.cost_volume(T(c1), T(c2), 3, dilation_rate=dil, nbre_cuts=k), O.cost_volume(c1, c2, 3, dilation_rate=dil, nbre_cuts=k), rows)
    run_network(rows)
    if verbose:
        for name, ok, msg in rows:
            print(f"{'ok  ' if ok else 'FAIL'}  {name:70s} {msg}")
        print(f"{sum(ok for _, ok, _ in rows)} of {len(rows)} comparisons agree")
    return rows


if __name__ == "__main__":
    sys.exit(0 if all(ok for _, ok, _ in run()) else 1)
