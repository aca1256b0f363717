"""A numpy-backed stand-in for the handful of TensorFlow-1.x symbols the reference touches  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Why it exists.  TensorFlow cannot be installed here (no wheel, no network), so the reference's arithmetic cannot run.  Its
PYTHON can: `/root/reference/{hyperparams,modules,networks,train,synthesize,data_load}.py` are imported UNMODIFIED with this
module registered as `tensorflow` (`oracle/run_reference.py`), and then the reference's own source decides every structural
fact of the path -- layer order, scope counters and variable names, kernel sizes / dilations / paddings, which tensor is split
where, the mask expression, the decoder-input shift (`train.py:51`) and the driver loop (`synthesize.py:45-57`).  What this file
restates is only the per-op semantics of TensorFlow (graph mode: placeholders, `Session.run` with feeds of ANY tensor,
variables by scope name, `Saver.restore`), written from TF 1.x's documented behaviour; where TF's own formula is known it is
followed literally (e.g. `tf.nn.batch_normalization`: `x * inv + (beta - mean * inv)` with `inv = rsqrt(var + eps) * gamma`).
So: **the reference's structure is pinned, TensorFlow's kernels are still a restatement** (DESIGN section 5).

Only tests/, tests/golden/make_golden_from_reference.py and oracle/run_reference.py import this.

Scope: every symbol reachable from `Graph(mode="synthesize")` and `synthesize()`, and (round 5) from `Graph(num, mode="train")` up to
the losses and the learning rate (`train.py:82-116`, `utils.py:134-145`): `sigmoid_cross_entropy_with_logits`, `pad(constant_values)`,
`minimum`, `**`, `clip_by_value`, inert `tf.summary.*`, and an `AdamOptimizer` whose `compute_gradients` / `apply_gradients` build nodes that
REFUSE to be evaluated (TensorFlow's autodiff and Adam kernel are not restated here: `oracle/train_ref.py`'s gradients are pinned by finite
differences THROUGH this graph's loss, tests/test_reference_pin.py).  The input queue (`data_load.get_batch`) still raises: the training pin
replaces it by placeholders.  `DROPOUT_HOOK` lets a test decide what `tf.layers.dropout(training=True)` does (TF's random stream cannot be
reproduced): it is called at graph construction with (variable scope, rate) and returns the mask function applied at run time.

Graph model: a `Tensor` is a lazy node (function + inputs + static shape + dtype).  `Session.run(fetches, feed_dict)` evaluates
the ancestors of the fetches iteratively; a fed tensor (placeholder or not: `synthesize.py:57` feeds `g.Y`) cuts the graph.
`FLOAT` selects the arithmetic every `tf.float32` resolves to: numpy float32 (TF's) or float64 (to tell structure from rounding).
"""
import re
import sys
import types

import numpy as np

FLOAT = np.float32          # what tf.float32 means; set with set_float()
RUN_LOG = []                # every Session.run: dict(fetches=[names], feeds={name: array}, results=[arrays])
LOG_RUNS = False


def set_float(dt):
    global FLOAT
    FLOAT = np.dtype(dt).type


class DType:
    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "tf." + self.name


float32 = DType("float32")
float64 = DType("float64")
int32 = DType("int32")
int64 = DType("int64")
bool_ = DType("bool")
string = DType("string")


def _np(dt):
    if isinstance(dt, DType):
        return {"float32": FLOAT, "float64": np.float64, "int32": np.int32, "int64": np.int64, "bool": np.bool_, "string": np.object_}[dt.name]
    return np.dtype(dt).type


# ----------------------------------------------------------------------------------------------- graph / scopes / variables
class _Graph:
    def __init__(self):
        self.scope = []                 # variable-scope stack
        self.variables = {}             # op name -> Variable (creation order kept: dict)
        self.trainable = []
        self.layer_names = {}           # (scope, base) -> count, for tf.layers default names (conv1d, conv1d_1, ...)
        self.values = {}                # op name -> ndarray


_G = _Graph()


def reset_default_graph():
    global _G
    _G = _Graph()
    del RUN_LOG[:]
    del DROPOUT_CALLS[:]
    del SUMMARIES[:]


def get_default_graph():
    return _G


class GraphKeys:
    TRAINABLE_VARIABLES = "trainable_variables"
    GLOBAL_VARIABLES = "variables"


class variable_scope:
    """tf.variable_scope(name_or_scope, default_name=None, values=None, reuse=None)."""

    def __init__(self, name_or_scope, default_name=None, values=None, reuse=None):
        self.name = name_or_scope if name_or_scope is not None else default_name
        self.reuse = reuse

    def __enter__(self):
        _G.scope.append(self.name)
        return self

    def __exit__(self, *a):
        _G.scope.pop()
        return False


def _scope_name():
    return "/".join(_G.scope)


def _full(name):
    s = _scope_name()
    return s + "/" + name if s else name


class TensorShape:
    def __init__(self, dims):
        self.dims = None if dims is None else list(dims)

    def as_list(self):
        if self.dims is None:
            raise ValueError("as_list() is not defined on an unknown TensorShape.")
        return list(self.dims)

    def __getitem__(self, i):
        return self.dims[i]

    def __len__(self):
        return len(self.dims)

    @property
    def ndims(self):
        return None if self.dims is None else len(self.dims)


def _bshape(a, b):
    """Static broadcast shape (None = unknown)."""
    if a is None or b is None:
        return None
    n = max(len(a), len(b))
    a = [1] * (n - len(a)) + list(a)
    b = [1] * (n - len(b)) + list(b)
    out = []
    for x, y in zip(a, b):
        if x == 1:
            out.append(y)
        elif y == 1:
            out.append(x)
        elif x is None:
            out.append(y)
        else:
            out.append(x)
    return out


class Tensor:
    _count = 0

    def __init__(self, fn, inputs, shape, dtype, name=None):
        self.fn, self.inputs, self.shape_, self.dtype = fn, list(inputs), (None if shape is None else list(shape)), dtype
        Tensor._count += 1
        self.name = (name or "op") + "_%d:0" % Tensor._count

    def get_shape(self):
        return TensorShape(self.shape_)

    @property
    def shape(self):
        return TensorShape(self.shape_)

    # arithmetic the reference writes with Python operators (modules.py:193, networks.py:140,146)
    def __add__(self, o):
        return _binary(np.add, self, o)

    def __radd__(self, o):
        return _binary(np.add, o, self)

    def __sub__(self, o):
        return _binary(np.subtract, self, o)

    def __rsub__(self, o):
        return _binary(np.subtract, o, self)

    def __mul__(self, o):
        return _binary(np.multiply, self, o)

    def __rmul__(self, o):
        return _binary(np.multiply, o, self)

    def __truediv__(self, o):
        return _binary(np.true_divide, self, o)

    def __neg__(self):
        return Tensor(lambda x: -x, [self], self.shape_, self.dtype, "neg")

    def __pow__(self, o):
        return _binary(np.power, self, o)

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        shp = None
        if self.shape_ is not None:
            shp = []
            dims = list(self.shape_)
            for k, s in enumerate(idx):
                d = dims[k]
                if isinstance(s, slice):
                    shp.append(None if d is None else len(range(*s.indices(d))))
                elif s is None or s is Ellipsis:
                    shp = None
                    break
                # an integer index drops the axis
            if shp is not None:
                shp += dims[len(idx):]
        return Tensor(lambda x: x[idx], [self], shp, self.dtype, "strided_slice")

    __hash__ = object.__hash__

    def __bool__(self):
        raise TypeError("Using a `tf.Tensor` as a Python `bool` is not allowed.")


def _const(v, dtype=None):
    if isinstance(v, Tensor):
        return v
    if isinstance(v, (bool, np.bool_)):
        arr = np.asarray(v)
    elif isinstance(v, (int, float)) and dtype is None:
        return v                                   # stays a weak Python scalar, like TF's implicit conversion to the other operand's dtype
    else:
        arr = np.asarray(v, dtype=dtype)
    return Tensor(lambda: arr, [], list(arr.shape), arr.dtype.type, "Const")


def _binary(op, a, b):
    a, b = _const(a), _const(b)
    ts = [x for x in (a, b) if isinstance(x, Tensor)]
    dt = ts[0].dtype
    if isinstance(a, Tensor) and isinstance(b, Tensor):
        return Tensor(lambda x, y: op(x, y), [a, b], _bshape(a.shape_, b.shape_), dt, op.__name__)
    if isinstance(a, Tensor):
        s = dt(b)                                  # convert_to_tensor(python scalar, dtype=other operand's dtype)
        return Tensor(lambda x: op(x, s), [a], a.shape_, dt, op.__name__)
    s = dt(a)
    return Tensor(lambda y: op(s, y), [b], b.shape_, dt, op.__name__)


class Variable(Tensor):
    def __init__(self, initial_value=None, name=None, trainable=True, dtype=None, shape=None, initializer=None, _full_name=None):
        opname = _full_name or _full(name)
        if opname in _G.variables:
            raise ValueError("Variable %s already exists, disallowed. Did you mean to set reuse=True?" % opname)
        if initial_value is not None:
            arr = np.asarray(initial_value)
            if arr.dtype == np.int64:
                arr = arr.astype(np.int32)         # tf.Variable(0) is int32
            shape = arr.shape
            dtype = arr.dtype.type
            self._init = lambda: arr.copy()
        else:
            dtype = _np(dtype if dtype is not None else float32)
            shp = tuple(int(s) for s in shape)
            self._init = lambda: np.asarray(initializer(shp, dtype), dtype=dtype)
        Tensor.__init__(self, lambda: _G.values[opname], [], list(shape), dtype, opname)
        self.name = opname + ":0"
        self.op_name = opname
        self.trainable = trainable
        _G.variables[opname] = self
        if trainable:
            _G.trainable.append(self)

    class _Op:
        def __init__(self, n):
            self.name = n

    @property
    def op(self):
        return Variable._Op(self.op_name)


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True):
    return Variable(name=name, shape=shape, dtype=dtype, initializer=initializer, trainable=trainable)


def get_collection(key, scope=None):
    vs = _G.trainable if key == GraphKeys.TRAINABLE_VARIABLES else list(_G.variables.values())
    if scope is None:
        return list(vs)
    return [v for v in vs if re.match(scope, v.name)]          # TF filters with re.match on the name


def global_variables_initializer():
    def init():
        for n, v in _G.variables.items():
            _G.values[n] = v._init()
        return None
    return Tensor(init, [], [], None, "init")


# initialisers (only used by global_variables_initializer; the parity runs overwrite every variable through Saver.restore)
_INIT_RNG = np.random.default_rng(0)


def _truncated_normal(shape, std, dtype):
    x = _INIT_RNG.standard_normal(shape)
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = _INIT_RNG.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2
    return (x * std).astype(dtype)


def truncated_normal_initializer(mean=0.0, stddev=1.0):
    return lambda shape, dtype: mean + _truncated_normal(shape, stddev, dtype)


def constant_initializer(value=0.0):
    return lambda shape, dtype: np.full(shape, value, dtype)


def zeros_initializer():
    return lambda shape, dtype: np.zeros(shape, dtype)


def ones_initializer():
    return lambda shape, dtype: np.ones(shape, dtype)


def _variance_scaling_initializer(factor=2.0, mode="FAN_IN", uniform=False):
    """tf.contrib.layers.variance_scaling_initializer defaults: truncated normal, stddev sqrt(1.3 * factor / fan_in)."""
    def init(shape, dtype):
        fan_in = float(shape[-2]) if len(shape) > 1 else float(shape[-1])
        for d in shape[:-2]:
            fan_in *= float(d)
        return _truncated_normal(shape, np.sqrt(1.3 * factor / fan_in), dtype)
    return init


# ----------------------------------------------------------------------------------------------- plain ops
def placeholder(dtype, shape=None, name=None):
    def missing():
        raise ValueError("You must feed a value for placeholder tensor")
    return Tensor(missing, [], None if shape is None else list(shape), _np(dtype), name or "Placeholder")


def zeros(shape, dtype=float32, name=None):
    dt = _np(dtype)
    shp = tuple(shape)
    return Tensor(lambda: np.zeros(shp, dt), [], list(shp), dt, "zeros")


def ones(shape, dtype=float32, name=None):
    dt = _np(dtype)
    shp = tuple(shape)
    return Tensor(lambda: np.ones(shp, dt), [], list(shp), dt, "ones")


def zeros_like(t):
    return Tensor(lambda x: np.zeros_like(x), [t], t.shape_, t.dtype, "zeros_like")


def ones_like(t):
    return Tensor(lambda x: np.ones_like(x), [t], t.shape_, t.dtype, "ones_like")


def convert_to_tensor(v, dtype=None):
    return _const(np.asarray(v), None if dtype is None else _np(dtype))


def concat(values, axis, name=None):
    values = [_const(np.asarray(v)) if not isinstance(v, Tensor) else v for v in values]
    shp = None
    if all(v.shape_ is not None for v in values):
        shp = list(values[0].shape_)
        ax = axis % len(shp)
        tot = 0
        for v in values:
            tot = None if (tot is None or v.shape_[ax] is None) else tot + v.shape_[ax]
        for k in range(len(shp)):                       # the other axes: any known size wins
            if k != ax:
                for v in values:
                    if shp[k] is None:
                        shp[k] = v.shape_[k]
        shp[ax] = tot
    return Tensor(lambda *xs: np.concatenate(xs, axis), values, shp, values[0].dtype, "concat")


def pad(t, paddings, mode="CONSTANT", constant_values=0, name=None):
    pw = [tuple(p) for p in paddings]
    shp = None if t.shape_ is None else [None if d is None else d + a + b for d, (a, b) in zip(t.shape_, pw)]
    return Tensor(lambda x: np.pad(x, pw, mode="constant", constant_values=constant_values), [t], shp, t.dtype, "Pad")


def split(value, num_or_size_splits, axis=0, name=None):
    n = int(num_or_size_splits)
    shp = None
    if value.shape_ is not None:
        shp = list(value.shape_)
        ax = axis % len(shp)
        if shp[ax] is not None:
            if shp[ax] % n:
                raise ValueError("Dimension size must be evenly divisible by %d but is %d" % (n, shp[ax]))
            shp[ax] //= n
    outs = []
    for k in range(n):
        outs.append(Tensor(lambda x, k=k: np.split(x, n, axis)[k], [value], shp, value.dtype, "split"))
    return outs


def expand_dims(t, axis, name=None):
    shp = None
    if t.shape_ is not None:
        shp = list(t.shape_)
        shp.insert(axis if axis >= 0 else len(shp) + 1 + axis, 1)
    return Tensor(lambda x: np.expand_dims(x, axis), [t], shp, t.dtype, "ExpandDims")


def squeeze(t, axis=None, name=None):
    shp = None
    if t.shape_ is not None and axis is not None:
        axes = [axis] if isinstance(axis, int) else list(axis)
        shp = [d for k, d in enumerate(t.shape_) if k not in [a % len(t.shape_) for a in axes]]
    ax = axis if axis is None or isinstance(axis, int) else tuple(axis)
    return Tensor(lambda x: np.squeeze(x, ax), [t], shp, t.dtype, "Squeeze")


def transpose(t, perm=None, name=None):
    shp = None if t.shape_ is None else ([t.shape_[p] for p in perm] if perm is not None else t.shape_[::-1])
    return Tensor(lambda x: np.transpose(x, perm), [t], shp, t.dtype, "transpose")


def tile(t, multiples, name=None):
    shp = None if t.shape_ is None else [None if d is None else d * m for d, m in zip(t.shape_, multiples)]
    return Tensor(lambda x: np.tile(x, multiples), [t], shp, t.dtype, "Tile")


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    def f(x, y):
        if transpose_a:
            x = np.swapaxes(x, -1, -2)
        if transpose_b:
            y = np.swapaxes(y, -1, -2)
        return np.matmul(x, y)
    shp = None
    if a.shape_ is not None and b.shape_ is not None:
        m = a.shape_[-1] if transpose_a else a.shape_[-2]
        n = b.shape_[-2] if transpose_b else b.shape_[-1]
        shp = list(a.shape_[:-2]) + [m, n]
    return Tensor(f, [a, b], shp, a.dtype, "MatMul")


def rsqrt(x, name=None):
    x = x if isinstance(x, Tensor) else _const(np.asarray(x))
    return Tensor(lambda v: (1.0 / np.sqrt(v)).astype(v.dtype), [x], x.shape_, x.dtype, "Rsqrt")


def to_float(x, name=None):
    if not isinstance(x, Tensor):
        arr = np.asarray(x, dtype=FLOAT)
        return Tensor(lambda: arr, [], list(arr.shape), FLOAT, "ToFloat")
    dt = FLOAT
    return Tensor(lambda v: v.astype(dt), [x], x.shape_, dt, "ToFloat")


def cast(x, dtype, name=None):
    dt = _np(dtype)
    return Tensor(lambda v: v.astype(dt), [x], x.shape_, dt, "Cast")


def sequence_mask(lengths, maxlen=None, dtype=bool_, name=None):
    """mask[i, j] = j < lengths[i]  (negative lengths give an all-False row)."""
    lengths = lengths if isinstance(lengths, Tensor) else _const(np.asarray(lengths))
    m = int(maxlen)
    shp = None if lengths.shape_ is None else list(lengths.shape_) + [m]
    return Tensor(lambda l: np.arange(m)[None, :] < np.asarray(l)[..., None], [lengths], shp, np.bool_, "SequenceMask")


def logical_or(a, b, name=None):
    return Tensor(np.logical_or, [a, b], _bshape(a.shape_, b.shape_), np.bool_, "LogicalOr")


def equal(a, b, name=None):
    if not isinstance(b, Tensor):
        bv = b
        return Tensor(lambda x: x == bv, [a], a.shape_, np.bool_, "Equal")
    return Tensor(lambda x, y: x == y, [a, b], _bshape(a.shape_, b.shape_), np.bool_, "Equal")


def not_equal(a, b, name=None):
    e = equal(a, b)
    return Tensor(np.logical_not, [e], e.shape_, np.bool_, "NotEqual")


def where(condition, x=None, y=None, name=None):
    def f(c, a, b):
        if c.shape != a.shape or a.shape != b.shape:            # tf.where(cond, x, y) wants equal shapes (no broadcasting in TF 1.x)
            raise ValueError("tf.where: shapes differ: %s %s %s" % (c.shape, a.shape, b.shape))
        return np.where(c, a, b)
    return Tensor(f, [condition, x, y], x.shape_, x.dtype, "Select")


def argmax(t, axis=None, name=None, output_type=int64):
    dt = _np(output_type)
    shp = None if t.shape_ is None else [d for k, d in enumerate(t.shape_) if k != axis % len(t.shape_)]
    return Tensor(lambda x: np.argmax(x, axis).astype(dt), [t], shp, dt, "ArgMax")     # first index on ties, like TF's kernel


def reduce_mean(t, axis=None, keep_dims=False, name=None):
    return Tensor(lambda x: np.mean(x, axis=axis, keepdims=keep_dims), [t], None, t.dtype, "Mean")


def reduce_sum(t, axis=None, keep_dims=False, name=None):
    return Tensor(lambda x: np.sum(x, axis=axis, keepdims=keep_dims), [t], None, t.dtype, "Sum")


def abs(t, name=None):                                                                   # noqa: A001
    return Tensor(np.abs, [t], t.shape_, t.dtype, "Abs")


def minimum(a, b, name=None):
    return _binary(np.minimum, a, b)


def maximum(a, b, name=None):
    return _binary(np.maximum, a, b)


def clip_by_value(t, clip_value_min, clip_value_max, name=None):
    return Tensor(lambda x: np.clip(x, clip_value_min, clip_value_max), [t], t.shape_, t.dtype, "clip_by_value")


# ----------------------------------------------------------------------------------------------- tf.nn
def _sigmoid(x, name=None):
    one = None

    def f(v):
        return (1.0 / (1.0 + np.exp(-v))).astype(v.dtype)
    del one
    return Tensor(f, [x], x.shape_, x.dtype, name or "Sigmoid")


def _relu(x, name=None):
    return Tensor(lambda v: np.maximum(v, 0), [x], x.shape_, x.dtype, name or "Relu")


def _softmax(x, axis=-1, name=None):
    def f(v):
        e = np.exp(v - v.max(axis=-1, keepdims=True))           # Eigen's softmax: exp(logits - max) / sum
        return e / e.sum(axis=-1, keepdims=True)
    return Tensor(f, [x], x.shape_, x.dtype, "Softmax")


def _embedding_lookup(params, ids, name=None):
    shp = None if (ids.shape_ is None or params.shape_ is None) else list(ids.shape_) + list(params.shape_[1:])
    return Tensor(lambda p, i: p[i], [params, ids], shp, params.dtype, "embedding_lookup")


def _moments(x, axes, keep_dims=False):
    """tf.nn.moments: mean, then mean of squared differences from it (biased)."""
    def fm(v):
        return v.mean(axis=tuple(axes), keepdims=keep_dims)

    def fv(v):
        m = v.mean(axis=tuple(axes), keepdims=True)
        return np.mean(np.square(v - m), axis=tuple(axes), keepdims=keep_dims)
    return Tensor(fm, [x], None, x.dtype, "mean"), Tensor(fv, [x], None, x.dtype, "variance")


def _batch_normalization(x, mean, variance, offset, scale, variance_epsilon):
    """tf.nn.batch_normalization, literally: inv = rsqrt(var + eps) * scale;  x * inv + (offset - mean * inv)."""
    def f(v, m, var, off, sc):
        inv = (1.0 / np.sqrt(var + v.dtype.type(variance_epsilon))).astype(v.dtype)
        inv = inv * sc
        return v * inv + (off - m * inv)
    return Tensor(f, [x, mean, variance, offset, scale], x.shape_, x.dtype, "batchnorm")


def _sigmoid_cross_entropy_with_logits(_sentinel=None, labels=None, logits=None, name=None):
    """tf.nn.sigmoid_cross_entropy_with_logits, the documented stable form:  max(x, 0) - x * z + log(1 + exp(-abs(x)))  (train.py:90,108)."""
    if _sentinel is not None:
        raise ValueError("Only call `sigmoid_cross_entropy_with_logits` with named arguments (labels=..., logits=...)")

    def f(x, z):
        if x.shape != z.shape:
            raise ValueError("logits and labels must have the same shape (%s vs %s)" % (x.shape, z.shape))
        return np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))
    return Tensor(f, [logits, labels], logits.shape_, logits.dtype, "logistic_loss")


nn = types.ModuleType("tensorflow.nn")
nn.sigmoid, nn.relu, nn.softmax = _sigmoid, _relu, _softmax
nn.embedding_lookup, nn.moments, nn.batch_normalization = _embedding_lookup, _moments, _batch_normalization
nn.sigmoid_cross_entropy_with_logits = _sigmoid_cross_entropy_with_logits
sigmoid = _sigmoid


# ----------------------------------------------------------------------------------------------- tf.layers
def _layer_scope(base):
    """tf.layers gives a layer the first free name of `base`, `base_1`, ... inside the current variable scope."""
    key = (_scope_name(), base)
    n = _G.layer_names.get(key, 0)
    _G.layer_names[key] = n + 1
    return base if n == 0 else "%s_%d" % (base, n)


def _conv1d(inputs, filters, kernel_size, strides=1, padding="valid", dilation_rate=1, activation=None, use_bias=True,
            kernel_initializer=None, bias_initializer=None, name=None, reuse=None):
    """tf.layers.conv1d: kernel (k, Cin, Cout), bias (Cout,); cross-correlation (no flip);
    'same' = (k-1)*dilation zeros in total, the smaller half on the left; 'valid' = none."""
    k = int(kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size)
    rate = int(dilation_rate[0] if isinstance(dilation_rate, (tuple, list)) else dilation_rate)
    cin = inputs.shape_[-1]
    if cin is None:
        raise ValueError("The channel dimension of the inputs should be defined. Found `None`.")
    pad_mode = padding.lower()
    if pad_mode not in ("same", "valid"):
        raise ValueError("padding " + padding)
    with variable_scope(name or _layer_scope("conv1d")):
        kernel = get_variable("kernel", [k, cin, filters], float32, kernel_initializer or _variance_scaling_initializer(1.0))
        bias = get_variable("bias", [filters], float32, bias_initializer or zeros_initializer()) if use_bias else None

    def f(x, w, *b):
        total = (k - 1) * rate
        if pad_mode == "same":
            pl = total // 2
            x = np.pad(x, ((0, 0), (pl, total - pl), (0, 0)))
        tout = x.shape[1] - total
        if tout <= 0:
            raise ValueError("conv1d: input shorter than the dilated kernel")
        y = None
        for j in range(k):
            t = np.matmul(x[:, j * rate: j * rate + tout, :], w[j])
            y = t if y is None else y + t
        if b:
            y = y + b[0]                                       # nn.bias_add
        return y
    shp = None if inputs.shape_ is None else [inputs.shape_[0], inputs.shape_[1] if pad_mode == "same" else
                                              (None if inputs.shape_[1] is None else inputs.shape_[1] - (k - 1) * rate), filters]
    out = Tensor(f, [inputs, kernel] + ([bias] if use_bias else []), shp, inputs.dtype, "conv1d")
    return activation(out) if activation is not None else out


def _conv2d_transpose(inputs, filters, kernel_size, strides=(1, 1), padding="valid", activation=None, use_bias=True,
                      kernel_initializer=None, bias_initializer=None, name=None, reuse=None):
    """tf.layers.conv2d_transpose, NHWC, kernel (kh, kw, Cout, Cin) = the gradient with respect to its input of the forward
    convolution  fwd: (B, Hout, Wout, Cout) -> (B, H, W, Cin)  with that kernel, stride and padding.  Output size 'same': in * stride.
    Forward SAME padding along an axis of output length n_in (our input) from length n_out (our output):
    total = max((n_in - 1) * s + k - n_out, 0), the smaller half in front.  Then  out[s*t + j - front] += in[t] . W[j]."""
    kh, kw = kernel_size
    sh, sw = strides
    cin = inputs.shape_[-1]
    pad_mode = padding.lower()
    with variable_scope(name or _layer_scope("conv2d_transpose")):
        kernel = get_variable("kernel", [kh, kw, filters, cin], float32, kernel_initializer or _variance_scaling_initializer(1.0))
        bias = get_variable("bias", [filters], float32, bias_initializer or zeros_initializer()) if use_bias else None

    def axis_geom(n_in, k, s):
        if pad_mode == "same":
            n_out = n_in * s
            total = max((n_in - 1) * s + k - n_out, 0)
            return n_out, total // 2
        return (n_in - 1) * s + k, 0                                                   # valid (max(k - s, 0) extra; k >= s here)

    def f(x, w, *b):
        B, H, W_, _ = x.shape
        Ho, fh = axis_geom(H, kh, sh)
        Wo, fw = axis_geom(W_, kw, sw)
        y = np.zeros((B, Ho, Wo, filters), x.dtype)
        for jh in range(kh):
            for jw in range(kw):
                contrib = np.matmul(x, w[jh, jw].T)                                    # (B,H,W,Cout):  sum_i in[.., i] * W[jh,jw,o,i]
                for th in range(H):
                    oh = sh * th + jh - fh
                    if not 0 <= oh < Ho:
                        continue
                    ow = sw * np.arange(W_) + jw - fw
                    ok = (ow >= 0) & (ow < Wo)
                    y[:, oh, ow[ok], :] += contrib[:, th, ok, :]
        if b:
            y = y + b[0]
        return y
    s = inputs.shape_
    shp = None if s is None else [s[0], None if s[1] is None else axis_geom(s[1], kh, sh)[0],
                                  None if s[2] is None else axis_geom(s[2], kw, sw)[0], filters]
    out = Tensor(f, [inputs, kernel] + ([bias] if use_bias else []), shp, inputs.dtype, "conv2d_transpose")
    return activation(out) if activation is not None else out


_DROPOUT_RNG = np.random.default_rng(0)
DROPOUT_HOOK = None         # test hook: f(variable scope at the call, rate) -> (x -> dropped x); None: a numpy random mask
DROPOUT_CALLS = []          # (variable scope, rate, training) of every tf.layers.dropout call, in graph-construction order


def _dropout(inputs, rate=0.5, noise_shape=None, seed=None, training=False, name=None):
    """tf.layers.dropout: identity unless training; then keep with probability 1 - rate and scale by 1 / (1 - rate)."""
    DROPOUT_CALLS.append((_scope_name(), rate, bool(training)))
    if not training or rate == 0:
        return Tensor(lambda x: x, [inputs], inputs.shape_, inputs.dtype, "dropout_identity")
    if DROPOUT_HOOK is not None:
        return Tensor(DROPOUT_HOOK(_scope_name(), rate), [inputs], inputs.shape_, inputs.dtype, "dropout_hook")

    def f(x):
        keep = _DROPOUT_RNG.random(x.shape) >= rate
        return (x / x.dtype.type(1.0 - rate)) * keep
    return Tensor(f, [inputs], inputs.shape_, inputs.dtype, "dropout")


def _dense(inputs, units, activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None, name=None):
    cin = inputs.shape_[-1]
    with variable_scope(name or _layer_scope("dense")):
        kernel = get_variable("kernel", [cin, units], float32, kernel_initializer or _variance_scaling_initializer(1.0))
        bias = get_variable("bias", [units], float32, bias_initializer or zeros_initializer())
    out = Tensor(lambda x, w, b: np.matmul(x, w) + b, [inputs, kernel, bias], list(inputs.shape_[:-1]) + [units], inputs.dtype, "dense")
    return activation(out) if activation is not None else out


layers = types.ModuleType("tensorflow.layers")
layers.conv1d, layers.conv2d_transpose, layers.dropout, layers.dense = _conv1d, _conv2d_transpose, _dropout, _dense


# ----------------------------------------------------------------------------------------------- tf.contrib
def _layer_norm(inputs, center=True, scale=True, activation_fn=None, reuse=None, variables_collections=None,
                outputs_collections=None, trainable=True, begin_norm_axis=1, begin_params_axis=-1, scope=None):
    """tf.contrib.layers.layer_norm: beta (zeros) / gamma (ones) of shape inputs.shape[begin_params_axis:], moments over
    axes [begin_norm_axis, rank) with keep_dims, tf.nn.batch_normalization with variance_epsilon = 1e-12."""
    shp = inputs.shape_
    rank = len(shp)
    if begin_norm_axis < 0:
        begin_norm_axis = rank + begin_norm_axis
    params_shape = shp[begin_params_axis:]
    if any(d is None for d in params_shape):
        raise ValueError("Inputs: shape(inputs)[%s:] is not fully defined: %s" % (begin_params_axis, shp))
    with variable_scope(scope, "LayerNorm", [inputs], reuse=reuse):
        beta = get_variable("beta", params_shape, float32, zeros_initializer())
        gamma = get_variable("gamma", params_shape, float32, ones_initializer())
    mean, variance = _moments(inputs, list(range(begin_norm_axis, rank)), keep_dims=True)
    out = _batch_normalization(inputs, mean, variance, beta, gamma, 1e-12)
    return activation_fn(out) if activation_fn is not None else out


def _not_on_synthesis_path(*a, **k):
    raise NotImplementedError("training-only TensorFlow symbol; the synthesis path (SURVEY section 8) never calls it")


contrib = types.ModuleType("tensorflow.contrib")
contrib.layers = types.ModuleType("tensorflow.contrib.layers")
contrib.layers.layer_norm = _layer_norm
contrib.layers.variance_scaling_initializer = _variance_scaling_initializer
contrib.training = types.ModuleType("tensorflow.contrib.training")
contrib.training.bucket_by_sequence_length = _not_on_synthesis_path


# ----------------------------------------------------------------------------------------------- session / saver
class Session:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def run(self, fetches, feed_dict=None):
        single = not isinstance(fetches, (list, tuple))
        fl = [fetches] if single else list(fetches)
        memo = {}
        for t, v in (feed_dict or {}).items():
            arr = np.asarray(v)
            if t.dtype is not None and arr.dtype != t.dtype:
                arr = arr.astype(t.dtype)                                  # a feed is converted to the tensor's dtype
            if t.shape_ is not None:
                if arr.ndim != len(t.shape_) or any(d is not None and d != s for d, s in zip(t.shape_, arr.shape)):
                    raise ValueError("Cannot feed value of shape %s for Tensor %s, which has shape %s" % (arr.shape, t.name, t.shape_))
            memo[id(t)] = arr
        for root in fl:                                                    # iterative post-order evaluation (the graph is ~700 nodes deep)
            stack = [root]
            while stack:
                n = stack[-1]
                if id(n) in memo:
                    stack.pop()
                    continue
                todo = [i for i in n.inputs if id(i) not in memo]
                if todo:
                    stack.extend(todo)
                    continue
                memo[id(n)] = n.fn(*[memo[id(i)] for i in n.inputs])
                stack.pop()
        res = [memo[id(t)] for t in fl]
        if LOG_RUNS:
            RUN_LOG.append(dict(fetches=[t.name for t in fl], fetch_tensors=fl,
                                feeds={t.name: memo[id(t)] for t in (feed_dict or {})}, feed_tensors=list((feed_dict or {}).keys()),
                                results=res))
        return res[0] if single else res


CHECKPOINTS = {}            # directory -> {variable op name: ndarray}; what tf.train.latest_checkpoint / Saver.restore see
RESTORED = []               # (checkpoint path, variable op name) for every variable a Saver restored


def register_checkpoint(directory, mapping):
    CHECKPOINTS[directory.rstrip("/")] = mapping


def _latest_checkpoint(checkpoint_dir, latest_filename=None):
    d = checkpoint_dir.rstrip("/")
    return d + "/model_gs_shim" if d in CHECKPOINTS else None


class _Saver:
    def __init__(self, var_list=None, **k):
        self.var_list = list(_G.variables.values()) if var_list is None else list(var_list)

    def restore(self, sess, save_path):
        if save_path is None:
            raise ValueError("Can't load save_path when it is None.")
        src = CHECKPOINTS[save_path.rsplit("/", 1)[0]]
        for v in self.var_list:
            n = v.op.name
            if n not in src:
                raise KeyError("NotFoundError: Key %s not found in checkpoint" % n)
            a = np.asarray(src[n])
            if tuple(a.shape) != tuple(v.shape_):
                raise ValueError("Assign requires shapes of both tensors to match: %s %s vs %s" % (n, a.shape, v.shape_))
            _G.values[n] = a.astype(v.dtype)
            RESTORED.append((save_path, n))

    def save(self, *a, **k):
        _not_on_synthesis_path()


def _refuses(what):
    def f(*a):
        raise NotImplementedError(what + ": TensorFlow's autodiff / optimizer kernels are not restated by the shim")
    return f


class _AdamOptimizer:
    """tf.train.AdamOptimizer as train.py:115-125 BUILDS it: `compute_gradients(loss)` yields one (gradient, variable) pair per trainable variable,
    `apply_gradients` one op.  The nodes exist (the reference's graph construction runs to its last line) and refuse evaluation."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8, **k):
        self.learning_rate, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon

    def compute_gradients(self, loss, var_list=None):
        vs = list(_G.trainable) if var_list is None else list(var_list)
        return [(Tensor(_refuses("gradient of the loss w.r.t. " + v.op_name), [loss, v], v.shape_, v.dtype, "gradients"), v) for v in vs]

    def apply_gradients(self, grads_and_vars, global_step=None, name=None):
        gv = list(grads_and_vars)
        return Tensor(_refuses("train_op"), [g for g, _ in gv], [], None, "Adam")


train = types.ModuleType("tensorflow.train")
train.Saver, train.latest_checkpoint = _Saver, _latest_checkpoint
train.AdamOptimizer = _AdamOptimizer
train.Supervisor = train.slice_input_producer = _not_on_synthesis_path
SUMMARIES = []              # (kind, tag) of every tf.summary.* call: inert, but the reference's list of what it reports is visible to a test
summary = types.ModuleType("tensorflow.summary")
summary.scalar = lambda tag, t, *a, **k: SUMMARIES.append(("scalar", tag))
summary.image = lambda tag, t, *a, **k: SUMMARIES.append(("image", tag))
summary.merge_all = lambda *a, **k: list(SUMMARIES)
decode_raw = py_func = device = _not_on_synthesis_path


# ----------------------------------------------------------------------------------------------- installation
def install():
    """Register this module (and stubs for the plotting / audio packages `utils.py` imports at module level) so that the
    reference's files import unmodified.  Returns the names installed."""
    me = sys.modules[__name__]
    installed = []
    for name, mod in (("tensorflow", me), ("tensorflow.nn", nn), ("tensorflow.layers", layers), ("tensorflow.contrib", contrib),
                      ("tensorflow.contrib.layers", contrib.layers), ("tensorflow.train", train), ("tensorflow.summary", summary)):
        sys.modules[name] = mod
        installed.append(name)
    if "matplotlib" not in sys.modules:
        try:
            import matplotlib  # noqa: F401
        except ImportError:
            mpl = types.ModuleType("matplotlib")
            mpl.use = lambda *a, **k: None
            plt = types.ModuleType("matplotlib.pyplot")
            mpl.pyplot = plt
            sys.modules["matplotlib"], sys.modules["matplotlib.pyplot"] = mpl, plt
            installed += ["matplotlib", "matplotlib.pyplot"]
    return installed
