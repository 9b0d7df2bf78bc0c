"""A torch-backed stand-in for the few dozen TensorFlow-1 / Keras symbols that the reference's MODEL code uses
(src/train/src/model.py and the graph assembly in src/train/train-model.py:117-231), so that the reference's own
graph-construction code can be EXECUTED in the build container (TensorFlow is not installable here) to pin
oracle/restate_model.py.  Build-container tool only: never imported by the product, the tests or the oracle.

Eager semantics: every op computes immediately on float64 torch tensors; placeholders take their value from
`tensorflow.FEEDS`; variables take their value from `tensorflow.VARIABLE_PROVIDER(full_name, shape)` (so a seeded
weight set can be injected by variable name, and unknown names fail loudly); every variable read is recorded in
`tensorflow.VARIABLES`.  Variable scoping follows TF1 (`variable_scope` prefixes, Keras layer name uniquification
`conv2d`, `conv2d_1`, ...).
"""
import contextlib
import types

import numpy as np
import torch

__version__ = "1.15.4-shim"
DT = torch.float64
float32 = "float32"
bool = "bool"  # noqa: A001

FEEDS = []                 # values handed to tf.placeholder / placeholder_with_default in creation order (None = use default)
VARIABLE_PROVIDER = None   # callable(full_name, shape) -> ndarray
VARIABLES = {}             # full_name -> ndarray actually used
_SCOPE = []
_LAYER_UIDS = {}


class Dim(int):
    @property
    def value(self):
        return int(self)


class TensorShape:
    def __init__(self, dims):
        self.dims = [None if d is None else Dim(d) for d in (dims.dims if isinstance(dims, TensorShape) else dims)]

    def as_list(self):
        return [None if d is None else int(d) for d in self.dims]

    @property
    def ndims(self):
        return len(self.dims)

    def __len__(self):
        return len(self.dims)

    def __iter__(self):
        return iter(self.dims)

    def __getitem__(self, i):
        return TensorShape(self.dims[i]) if isinstance(i, slice) else self.dims[i]

    def __repr__(self):
        return f"TensorShape({self.as_list()})"


def _raw(x):
    if isinstance(x, Tensor):
        return x.t
    if isinstance(x, torch.Tensor):
        return x.to(DT) if x.is_floating_point() else x
    if isinstance(x, (list, tuple)) and any(isinstance(v, Tensor) for v in x):
        return torch.stack([_raw(v) for v in x])
    return torch.as_tensor(np.asarray(x), dtype=DT if np.asarray(x).dtype.kind == "f" else None)


class Tensor:
    def __init__(self, t, name=None):
        self.t = t if isinstance(t, torch.Tensor) else _raw(t)
        self.name = name

    shape = property(lambda s: TensorShape(list(s.t.shape)))

    def get_shape(self):
        return self.shape

    def numpy(self):
        return self.t.detach().cpu().numpy()

    def __getitem__(self, idx):
        return Tensor(self.t[idx])

    def _b(self, o, f):
        return Tensor(f(self.t, _raw(o)))

    __add__ = lambda s, o: s._b(o, lambda a, b: a + b)
    __radd__ = __add__
    __sub__ = lambda s, o: s._b(o, lambda a, b: a - b)
    __rsub__ = lambda s, o: s._b(o, lambda a, b: b - a)
    __mul__ = lambda s, o: s._b(o, lambda a, b: a * b)
    __rmul__ = __mul__
    __truediv__ = lambda s, o: s._b(o, lambda a, b: a / b)
    __rtruediv__ = lambda s, o: s._b(o, lambda a, b: b / a)
    __neg__ = lambda s: Tensor(-s.t)
    __pow__ = lambda s, o: s._b(o, lambda a, b: a ** b)

    def assign(self, value):
        self.t = _raw(value).clone()
        return self


# ---- scopes / variables -----------------------------------------------------------------------------------------
@contextlib.contextmanager
def variable_scope(name, reuse=None, **kw):
    _SCOPE.append(str(name))
    try:
        yield
    finally:
        _SCOPE.pop()


def _full(name):
    return "/".join(_SCOPE + [name])


def _provide(name, shape, initializer=None):
    full = _full(name)
    shape = [int(d) for d in (shape.as_list() if isinstance(shape, TensorShape) else shape)]
    if VARIABLE_PROVIDER is None:
        raise RuntimeError("tensorflow shim: set VARIABLE_PROVIDER before building the graph")
    val = np.asarray(VARIABLE_PROVIDER(full, shape), dtype=np.float64)
    if list(val.shape) != shape:
        raise ValueError(f"variable {full}: provider returned shape {val.shape}, graph asks for {shape}")
    VARIABLES[full] = val
    return Tensor(torch.as_tensor(val, dtype=DT), name=full)


def get_variable(name, shape=None, initializer=None, **kw):
    return _provide(name, shape, initializer)


def Variable(initial_value=None, dtype=None, name=None, **kw):
    init = initial_value() if callable(initial_value) else initial_value
    return _provide(name, list(_raw(init).shape))


def placeholder(dtype=None, shape=None, name=None):
    val = FEEDS.pop(0) if FEEDS else None
    if val is None:                                      # never fed on the inference path (labels, loss schedule scalars)
        val = np.zeros([1 if d is None else int(d) for d in (shape or ())])
    return Tensor(_raw(val), name=name)


def placeholder_with_default(default, shape=None, name=None):
    val = FEEDS.pop(0) if FEEDS else None
    return Tensor(_raw(default if val is None else val), name=name)


def constant(v, dtype=None, **kw):
    return Tensor(_raw(v))


def constant_initializer(v):
    return ("constant", v)


# ---- ops ----------------------------------------------------------------------------------------------------------
def pad(x, paddings, mode="CONSTANT"):
    t = _raw(x)
    p = [tuple(int(v) for v in q) for q in paddings]
    if mode.upper() == "REFLECT":
        assert t.dim() == 4 and p[0] == (0, 0) and p[3] == (0, 0)
        u = torch.nn.functional.pad(t.permute(0, 3, 1, 2), (p[2][0], p[2][1], p[1][0], p[1][1]), mode="reflect")
        return Tensor(u.permute(0, 2, 3, 1))
    flat = []
    for q in reversed(p):
        flat += [q[0], q[1]]
    return Tensor(torch.nn.functional.pad(t, flat))


def _conv2d(x, w, padding, strides=1):
    """NHWC x HWIO, TF padding semantics (SAME with stride 1 = symmetric zero pad for odd kernels)."""
    t, k = _raw(x).permute(0, 3, 1, 2), _raw(w).permute(3, 2, 0, 1)
    kh, kw = k.shape[2], k.shape[3]
    if str(padding).upper() == "SAME":
        assert strides in (1, (1, 1), [1, 1]) and kh % 2 == 1 and kw % 2 == 1
        y = torch.nn.functional.conv2d(t, k, padding=(kh // 2, kw // 2))
    else:
        y = torch.nn.functional.conv2d(t, k)
    return Tensor(y.permute(0, 2, 3, 1))


def split(value, num_or_size_splits, axis=0):
    t = _raw(value)
    return [Tensor(c) for c in torch.chunk(t, int(num_or_size_splits), dim=axis)]


def concat(values, axis):
    return Tensor(torch.cat([_raw(v) for v in values], dim=axis))


def stack(values, axis=0):
    return Tensor(torch.stack([_raw(v).to(DT) for v in values], dim=axis))


def transpose(x, perm):
    return Tensor(_raw(x).permute(*perm))


def reshape(x, shape):
    return Tensor(_raw(x).reshape([int(s) for s in shape]))


def sigmoid(x):
    return Tensor(torch.sigmoid(_raw(x)))


def tanh(x):
    return Tensor(torch.tanh(_raw(x)))


def sqrt(x):
    return Tensor(torch.sqrt(_raw(x)))


def floor(x):
    return Tensor(torch.floor(_raw(x)))


def sign(x):
    return Tensor(torch.sign(_raw(x)))


def zeros(shape, dtype=None):
    return Tensor(torch.zeros([int(s) for s in shape], dtype=DT))


def ones(shape, dtype=None):
    return Tensor(torch.ones([int(s) for s in shape], dtype=DT))


def shape(x):
    return Tensor(torch.as_tensor(list(_raw(x).shape)))


def size(x):
    return Tensor(torch.as_tensor(_raw(x).numel()))


def to_float(x):
    return Tensor(_raw(x).to(DT))


def reduce_sum(x, axis=None, keep_dims=False, keepdims=False):
    t = _raw(x)
    return Tensor(t.sum() if axis is None else t.sum(dim=axis, keepdim=keep_dims or keepdims))


def clip_by_value(x, lo, hi):
    return Tensor(torch.clamp(_raw(x), lo, hi))


def random_uniform(shape, minval=0, maxval=1, dtype=None):
    s = [int(v) for v in (_raw(shape).tolist() if isinstance(shape, Tensor) else shape)]
    return Tensor(torch.rand(s, dtype=DT) * (maxval - minval) + minval)


def equal(a, b):
    return Tensor(_raw(a) == _raw(b))


def logical_not(a):
    return Tensor(~_raw(a).bool())


def logical_or(a, b):
    return Tensor(_raw(a).bool() | _raw(b).bool())


def cond(pred, true_fn=None, false_fn=None):
    return true_fn() if builtins_bool(_raw(pred).all()) else false_fn()


def builtins_bool(v):
    return True if v else False


def is_variable_initialized(v):
    return True


class _Math(types.SimpleNamespace):
    @staticmethod
    def reduce_mean(x, axis=None, keepdims=False):
        return Tensor(_raw(x).mean(dim=tuple(axis), keepdim=keepdims))


math = _Math()


class _NN(types.SimpleNamespace):
    @staticmethod
    def convolution(input, filter, padding, data_format=None, **kw):   # noqa: A002
        assert data_format is None
        return _conv2d(input, filter, padding)

    @staticmethod
    def moments(x, axes, keep_dims=False):
        t = _raw(x)
        mean = t.mean(dim=tuple(axes), keepdim=True)
        var = ((t - mean) ** 2).mean(dim=tuple(axes), keepdim=True)
        if not keep_dims:
            mean, var = mean.squeeze(), var.squeeze()
        return Tensor(mean), Tensor(var)

    @staticmethod
    def weighted_moments(x, axes, frequency_weights, keep_dims=False):
        t, w = _raw(x), _raw(frequency_weights)
        sw = w.sum(dim=tuple(axes), keepdim=True)
        mean = (t * w).sum(dim=tuple(axes), keepdim=True) / sw
        var = (w * (t - mean) ** 2).sum(dim=tuple(axes), keepdim=True) / sw
        if not keep_dims:
            mean, var = mean.squeeze(), var.squeeze()
        return Tensor(mean), Tensor(var)

    @staticmethod
    def swish(x):
        t = _raw(x)
        return Tensor(t * torch.sigmoid(t))

    @staticmethod
    def relu(x):
        return Tensor(torch.relu(_raw(x)))

    @staticmethod
    def sigmoid(x):
        return sigmoid(x)

    @staticmethod
    def max_pool(x, ksize, strides, padding):
        t = _raw(x).permute(0, 3, 1, 2)
        k = int(ksize[1])
        assert list(strides) == [1, 1, 1, 1] and padding == "SAME"
        lo, hi = (k - 1) // 2, k // 2
        t = torch.nn.functional.pad(t, (lo, hi, lo, hi), value=float("-inf"))
        return Tensor(torch.nn.functional.max_pool2d(t, k, stride=1).permute(0, 2, 3, 1))

    @staticmethod
    def bias_add(x, b):
        return Tensor(_raw(x) + _raw(b))

    @staticmethod
    def bidirectional_dynamic_rnn(cell_fw, cell_bw, inputs, sequence_length=None, dtype=None, **kw):
        """TF1 semantics for batch-major inputs [B, T, ...] with full-length sequences: the backward cell consumes the
        reversed sequence and its outputs are reversed back; returns ((out_fw, out_bw), (state_fw, state_bw))."""
        x = _raw(inputs)
        B, T = x.shape[0], x.shape[1]
        if sequence_length is not None:
            ln = _raw(sequence_length).reshape(-1)
            if not builtins_bool((ln == T).all()):
                raise ValueError(f"tensorflow shim: sequence_length {ln.tolist()} != time steps {T} (partial sequences not modelled)")
        res = []
        for direction, cell in (("fw", cell_fw), ("bw", cell_bw)):
            with variable_scope("bidirectional_rnn"), variable_scope(direction):
                state = Tensor(torch.zeros([B] + cell.state_size.as_list(), dtype=DT))
                outs = []
                order = range(T) if direction == "fw" else range(T - 1, -1, -1)
                for t in order:
                    out, state = cell(Tensor(x[:, t]), state)
                    outs.append(_raw(out))
                if direction == "bw":
                    outs = outs[::-1]
                res.append((Tensor(torch.stack(outs, dim=1)), state))
        return (res[0][0], res[1][0]), (res[0][1], res[1][1])


class _RNNCellBase:
    """tf.nn.rnn_cell.RNNCell: a Layer whose __call__(inputs, state, scope=None) runs call() inside a variable scope named
    after the class (`conv_gru_cell`), reused on later steps."""

    def __init__(self, _reuse=None, **kw):
        pass

    def __call__(self, inputs, state, scope=None):
        with variable_scope(scope or _snake(type(self).__name__)):
            return self.call(inputs, state)


def _snake(name):
    """keras.utils.generic_utils.to_snake_case: WSConv2D -> ws_conv2d, ConvGRUCell -> conv_gru_cell"""
    import re
    inter = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    return re.sub("([a-z])([A-Z])", r"\1_\2", inter).lower()


class _RnnCellNS(types.SimpleNamespace):
    RNNCell = _RNNCellBase
    LSTMStateTuple = tuple


_NN.rnn_cell = _RnnCellNS()
nn = _NN()


# ---- tf.layers / tf.keras -----------------------------------------------------------------------------------------------
class _Layers(types.SimpleNamespace):
    @staticmethod
    def conv2d(x, filters, kernel_size, kernel_initializer=None, strides=1, padding="valid", use_bias=True, trainable=True, **kw):
        k = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
        cin = _raw(x).shape[-1]
        with variable_scope("conv2d"):
            if isinstance(kernel_initializer, tuple) and kernel_initializer[0] == "constant":
                w = Tensor(torch.full((k[0], k[1], cin, filters), float(kernel_initializer[1]), dtype=DT))
                VARIABLES[_full("kernel")] = w.numpy()
            else:
                w = get_variable("kernel", [k[0], k[1], cin, filters])
            y = _conv2d(x, w, padding, strides)
            if use_bias:
                y = y + get_variable("bias", [filters])
        return y


layers = _Layers()


class KLayer:
    """Keras Layer: build(input_shape) once, call(inputs, ...); unnamed layers get the Keras unique name (snake class name,
    `_1`, `_2`, ... in creation order); weights live under that name inside the current variable scope."""

    def __init__(self, name=None, **kw):
        base = name or _snake(type(self).__name__)
        if name is None:
            n = _LAYER_UIDS.get(base, 0)
            _LAYER_UIDS[base] = n + 1
            base = base if n == 0 else f"{base}_{n}"
        self.name = base
        self.built = False
        self.input_spec = None

    def build(self, input_shape):
        self.built = True

    def _shape_of(self, inputs):
        if isinstance(inputs, (list, tuple)):
            return [i.shape for i in inputs]
        return inputs.shape

    def __call__(self, inputs, *a, **kw):
        if not self.built:
            self.build(self._shape_of(inputs))
            self.built = True
        return self.call(inputs, *a, **kw)

    apply = __call__


class InputSpec:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Conv2D(KLayer):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", activation=None, use_bias=True,
                 kernel_initializer=None, bias_initializer=None, kernel_regularizer=None, name=None, **kw):
        super().__init__(name=name)
        self.filters = filters
        self.kernel_size = kernel_size if isinstance(kernel_size, (tuple, list)) else (kernel_size, kernel_size)
        self.padding, self.activation, self.use_bias = padding, activation, use_bias

    def build(self, input_shape):
        cin = input_shape.as_list()[-1]
        with variable_scope(self.name):
            self.kernel = get_variable("kernel", [self.kernel_size[0], self.kernel_size[1], cin, self.filters])
            self.bias = get_variable("bias", [self.filters]) if self.use_bias else None
        self.built = True

    def call(self, inputs):
        y = _conv2d(inputs, self.kernel, self.padding)
        if self.bias is not None:
            y = y + self.bias
        if self.activation == "sigmoid":
            y = sigmoid(y)
        elif self.activation is not None:
            raise NotImplementedError(self.activation)
        return y


class Multiply(KLayer):
    def call(self, inputs):
        return inputs[0] * inputs[1]


class Add(KLayer):
    def call(self, inputs):
        return inputs[0] + inputs[1]


class MaxPool2D(KLayer):
    def __init__(self, pool_size=(2, 2), **kw):
        super().__init__(**kw)

    def call(self, x):
        t = _raw(x).permute(0, 3, 1, 2)
        return Tensor(torch.nn.functional.max_pool2d(t, 2).permute(0, 2, 3, 1))


class UpSampling2D(KLayer):
    def __init__(self, size=(2, 2), interpolation="nearest", **kw):
        super().__init__(**kw)
        assert tuple(size) == (2, 2) and interpolation == "nearest"

    def call(self, x):
        t = _raw(x)
        return Tensor(t.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2))


class Cropping2D(KLayer):
    def __init__(self, cropping, **kw):
        super().__init__(**kw)
        self.c = int(cropping)

    def call(self, x):
        c = self.c
        return Tensor(_raw(x)[:, c:-c, c:-c, :])


class _NotUsed(KLayer):
    def __init__(self, *a, **kw):
        raise NotImplementedError(f"tensorflow shim: {type(self).__name__} is not on the inference path")


class Lambda(_NotUsed):
    pass


class Dense(_NotUsed):
    pass


class ELU(_NotUsed):
    pass


class TimeDistributed(_NotUsed):
    pass


class _Initializers(types.SimpleNamespace):
    @staticmethod
    def he_normal():
        return ("he_normal",)

    @staticmethod
    def Ones():
        return ("ones",)


class _KBackend(types.SimpleNamespace):
    @staticmethod
    def std(x, axis=None, keepdims=False):
        t = _raw(x)
        return Tensor(t.std(dim=tuple(axis), keepdim=keepdims, unbiased=False))

    @staticmethod
    def learning_phase():
        return False

    @staticmethod
    def int_shape(x):
        return tuple(x.shape.as_list())

    @staticmethod
    def mean(x, axis=None):
        return Tensor(_raw(x).mean(dim=tuple(axis)))


class _KerasLayersNS(types.SimpleNamespace):
    pass


keras = types.SimpleNamespace(
    layers=_KerasLayersNS(Conv2D=Conv2D, Layer=KLayer, UpSampling2D=UpSampling2D, MaxPool2D=MaxPool2D, Cropping2D=Cropping2D),
    initializers=_Initializers(), backend=_KBackend())


class _Contrib(types.SimpleNamespace):
    pass


contrib = _Contrib(layers=types.SimpleNamespace(l2_regularizer=lambda s: None, layer_norm=None),
                   rnn=types.SimpleNamespace(LSTMStateTuple=tuple))


initializers = types.SimpleNamespace(orthogonal=lambda *a, **k: ("orthogonal",), he_normal=_Initializers.he_normal)


class _Logging(types.SimpleNamespace):
    ERROR = 0

    @staticmethod
    def set_verbosity(v):
        pass


logging = _Logging()


def disable_v2_behavior():
    pass


def reset():
    """forget scopes, layer name counters, recorded variables and pending feeds"""
    _SCOPE.clear()
    _LAYER_UIDS.clear()
    VARIABLES.clear()
    del FEEDS[:]
