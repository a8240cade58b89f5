"""Minimal EAGER stand-in for the TensorFlow-1.x API surface that the reference's WaveNet files use
(wavenet_vocoder/models/{wavenet,modules,mixture,gaussian}.py, wavenet_vocoder/util.py), backed by torch on CPU.

TEST INFRASTRUCTURE ONLY (like everything under oracle/).  TensorFlow 1.x cannot be installed here (no network), so the
reference's graph code cannot run as is.  This module lets the reference's OWN SOURCE FILES execute unmodified: every
`tf.*` call they make is served by the 1:1 torch op with TensorFlow's documented semantics (kernel layouts, SAME/VALID
padding arithmetic, channels_first/last, batch_to_space_nd, TensorArray/while_loop as a Python loop ...).  What is being
pinned with it is the reference's COMPOSITION -- which op, in which order, with which transposes, paddings, splits,
scalings and queue updates -- i.e. exactly the part a hand-written restatement can get wrong.  Each primitive below is
small enough to be checked against the TF documentation by eye, and tests/test_oracle_golden.py cross-checks the
convolution primitives against independent naive loops.

Only oracle/gen_golden_stack.py uses this (to write tests/golden/stack_*.npz in the container that has /root/reference).
"""
import contextlib
import math
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

_STATE = types.SimpleNamespace(variables={}, gen=torch.Generator().manual_seed(20240917), uniform_draws=[], normal_draws=[],
                               scope=[], dropout_masks=[])


def reset(seed=20240917):
    _STATE.variables.clear(); _STATE.gen = torch.Generator().manual_seed(seed)
    _STATE.uniform_draws.clear(); _STATE.normal_draws.clear(); _STATE.scope.clear(); _STATE.dropout_masks.clear()


def variables():
    return _STATE.variables


def _t(x, dtype=None):
    if isinstance(x, torch.Tensor):
        return x if dtype is None else x.to(dtype)
    return torch.as_tensor(np.asarray(x), dtype=dtype)


# ------------------------------------------------------------------ initialisers / variables
def zeros_initializer():
    return lambda shape, dtype=None: torch.zeros(tuple(shape), dtype=torch.float32)


def constant_initializer(value, dtype=None):
    def init(shape, dtype=None):
        v = np.asarray(value, dtype=np.float32)
        if v.size == 1:
            return torch.full(tuple(shape), float(v.reshape(-1)[0]))
        assert v.size == int(np.prod(shape)), (v.shape, shape)
        return torch.from_numpy(v.reshape(-1).copy()).reshape(tuple(shape))       # TF fills in flat (row-major) order
    return init


def truncated_normal_initializer(mean=0.0, stddev=1.0):
    return lambda shape, dtype=None: torch.fmod(torch.randn(tuple(shape), generator=_STATE.gen), 2.0) * stddev + mean


def _glorot_uniform(shape):
    shape = tuple(int(s) for s in shape)
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=_STATE.gen) * 2 - 1) * lim


class _Shape(tuple):
    """tf.TensorShape-like view of a variable's shape (WeightNorm.build reads kernel.shape.ndims, modules.py:149)."""
    @property
    def ndims(self):
        return len(self)

    def as_list(self):
        return list(self)


class _Var(torch.Tensor):
    """A variable: a torch tensor whose .shape also answers .ndims."""
    @property
    def shape(self):
        return _Shape(torch.Tensor.size(self))


def _make_variable(name, shape, initializer):
    """Variables are shared by name (tf.AUTO_REUSE-like): the synthesis-mode model instance reuses the training one's."""
    if name in _STATE.variables:
        v = _STATE.variables[name]
        assert tuple(v.shape) == tuple(int(s) for s in shape), (name, v.shape, shape)
        return v
    v = (initializer(shape) if initializer is not None else _glorot_uniform(shape)).float().contiguous().as_subclass(_Var)
    _STATE.variables[name] = v
    return v


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True):
    return _make_variable(name, shape, initializer)


@contextlib.contextmanager
def variable_scope(name, *a, **k):
    _STATE.scope.append(name)
    try:
        yield types.SimpleNamespace(name=name)
    finally:
        _STATE.scope.pop()


@contextlib.contextmanager
def _noop_ctx(*a, **k):
    yield None


class TensorShape:
    def __init__(self, s):
        self._s = [int(d) for d in (s._s if isinstance(s, TensorShape) else s)]

    def as_list(self):
        return list(self._s)


# ------------------------------------------------------------------ layers
class Layer:
    """tf.layers.Layer / keras base layer: build on first call, then call()."""

    def __init__(self, trainable=True, name=None, **kwargs):
        self.name = name
        self.built = False
        self.trainable = trainable

    def add_variable(self, name, shape, initializer=None, dtype=None, trainable=True):
        return _make_variable('%s/%s' % (self.name, name), shape, initializer)

    def build(self, input_shape=None):
        self.built = True

    def call(self, inputs, *a, **k):
        raise NotImplementedError

    def __call__(self, inputs, *args, **kwargs):
        if not self.built:
            self.build(TensorShape(tuple(inputs.shape)))
            self.built = True
        return self.call(inputs, *args, **kwargs)

    def _track_checkpointable(self, *a, **k):
        pass


class Wrapper(Layer):
    """tf.keras.layers.Wrapper"""

    def __init__(self, layer, **kwargs):
        self.layer = layer
        super().__init__(**kwargs)

    def build(self, input_shape=None):
        self.built = True


def _tuple(v, n):
    return tuple(v) if isinstance(v, (tuple, list)) else (v,) * n


class _ConvBase(Layer):
    RANK = 1

    def __init__(self, filters, kernel_size, strides=1, padding='valid', data_format='channels_last', dilation_rate=1,
                 activation=None, use_bias=True, kernel_initializer=None, bias_initializer=None, kernel_regularizer=None,
                 bias_regularizer=None, activity_regularizer=None, kernel_constraint=None, bias_constraint=None,
                 trainable=True, name=None, **kwargs):
        super().__init__(trainable=trainable, name=name)
        self.filters = filters
        self.kernel_size = _tuple(kernel_size, self.RANK)
        self.strides = _tuple(strides, self.RANK)
        self.padding = padding
        self.data_format = data_format
        self.dilation_rate = _tuple(dilation_rate, self.RANK)
        self.activation = activation
        self.use_bias = use_bias
        self.kernel_initializer = kernel_initializer
        self.bias_initializer = bias_initializer or zeros_initializer()
        self.kernel = None
        self.bias = None

    def _in_channels(self, input_shape):
        s = TensorShape(input_shape).as_list()
        return s[1] if self.data_format == 'channels_first' else s[-1]


def _same_pad(k, d=1):
    """TF 'SAME' at stride 1: total = d*(k-1); before = total // 2, after = total - before."""
    total = d * (k - 1)
    return total // 2, total - total // 2


class Conv1D(_ConvBase):
    """tf.layers.Conv1D: kernel [k, in, filters]; cross-correlation; 'valid' or 'same'."""
    RANK = 1

    def build(self, input_shape):
        cin = self._in_channels(input_shape)
        self.kernel = self.add_variable('kernel', (self.kernel_size[0], cin, self.filters), self.kernel_initializer)
        if self.use_bias:
            self.bias = self.add_variable('bias', (self.filters,), self.bias_initializer)
        self.built = True

    def call(self, inputs):
        x = inputs if self.data_format == 'channels_first' else inputs.permute(0, 2, 1)
        if self.padding.lower() == 'same':
            x = F.pad(x, _same_pad(self.kernel_size[0], self.dilation_rate[0]))
        y = F.conv1d(x, self.kernel.permute(2, 1, 0).contiguous(), self.bias if self.use_bias else None,
                     stride=self.strides[0], dilation=self.dilation_rate[0])
        y = y if self.data_format == 'channels_first' else y.permute(0, 2, 1)
        return self.activation(y) if self.activation is not None else y


class Conv2D(_ConvBase):
    """tf.layers.Conv2D: kernel [kh, kw, in, filters]; stride 1 only (all the reference uses)."""
    RANK = 2

    def build(self, input_shape):
        cin = self._in_channels(input_shape)
        self.kernel = self.add_variable('kernel', self.kernel_size + (cin, self.filters), self.kernel_initializer)
        if self.use_bias:
            self.bias = self.add_variable('bias', (self.filters,), self.bias_initializer)
        self.built = True

    def call(self, inputs):
        assert self.strides == (1, 1)
        x = inputs if self.data_format == 'channels_first' else inputs.permute(0, 3, 1, 2)      # NCHW
        if self.padding.lower() == 'same':
            pt, pb = _same_pad(self.kernel_size[0]); pl, pr = _same_pad(self.kernel_size[1])
            x = F.pad(x, (pl, pr, pt, pb))
        y = F.conv2d(x, self.kernel.permute(3, 2, 0, 1).contiguous(), self.bias if self.use_bias else None)
        y = y if self.data_format == 'channels_first' else y.permute(0, 2, 3, 1)
        return self.activation(y) if self.activation is not None else y


class Conv2DTranspose(_ConvBase):
    """tf.layers.Conv2DTranspose: kernel [kh, kw, filters(out), in].  'same': output = input * stride; the full transposed
    output (in-1)*s + k is cropped by (k - s) split as floor/ceil exactly like the gradient of a SAME forward conv."""
    RANK = 2

    def build(self, input_shape):
        cin = self._in_channels(input_shape)
        self.kernel = self.add_variable('kernel', self.kernel_size + (self.filters, cin), self.kernel_initializer)
        if self.use_bias:
            self.bias = self.add_variable('bias', (self.filters,), self.bias_initializer)
        self.built = True

    def call(self, inputs):
        x = inputs if self.data_format == 'channels_first' else inputs.permute(0, 3, 1, 2)
        w = self.kernel.permute(3, 2, 0, 1).contiguous()                                       # [in, out, kh, kw]
        y = F.conv_transpose2d(x, w, self.bias if self.use_bias else None, stride=self.strides)
        if self.padding.lower() == 'same':
            outs = []
            for ax, (k, s, n) in enumerate(zip(self.kernel_size, self.strides, x.shape[2:])):
                want = n * s
                extra = y.shape[2 + ax] - want           # = max(k - s, 0)
                lo = extra // 2
                outs.append((lo, lo + want))
            y = y[:, :, outs[0][0]:outs[0][1], outs[1][0]:outs[1][1]]
        y = y if self.data_format == 'channels_first' else y.permute(0, 2, 3, 1)
        return self.activation(y) if self.activation is not None else y


class InputSpec:
    def __init__(self, *a, **k):
        pass


# ------------------------------------------------------------------ tensor ops
def shape(x):
    return tuple(int(d) for d in x.shape)


def rank(x):
    return x.dim()


def reshape(x, s):
    return _t(x).reshape(tuple(int(d) for d in s))


def transpose(x, perm):
    return x.permute(*perm)


def expand_dims(x, axis):
    if isinstance(axis, (list, tuple)):
        assert len(axis) == 1
        axis = axis[0]
    return x.unsqueeze(int(axis))


def squeeze(x, axis=None):
    if axis is None:
        return x.squeeze()
    for a in sorted([a % x.dim() for a in (axis if isinstance(axis, (list, tuple)) else [axis])], reverse=True):
        assert x.shape[a] == 1
        x = x.squeeze(a)
    return x


def split(x, num_or_size_splits, axis=0):
    n = num_or_size_splits
    if isinstance(n, int):
        assert x.shape[axis] % n == 0
        return list(torch.split(x, x.shape[axis] // n, dim=axis))
    return list(torch.split(x, list(n), dim=axis))


def concat(values, axis=0, name=None):          # tf.concat(values, axis); wavenet.py:574 calls it with keywords
    return torch.cat(list(values), dim=axis)


def pad(x, paddings):
    p = np.asarray(paddings).reshape(-1, 2)
    flat = []
    for before, after in reversed(p.tolist()):
        flat += [int(before), int(after)]
    return F.pad(x, flat)


def tile(x, multiples):
    return x.repeat(*[int(m) for m in multiples])


def constant(v, dtype=None, name=None):
    return v if isinstance(v, (int, float)) else np.asarray(v)


def cast(x, dtype):
    return _t(x).to(dtype)


def identity(x, name=None):
    return x


def one_hot(indices, depth, dtype=None):
    return F.one_hot(_t(indices).long(), int(depth)).float()


def argmax(x, axis=None):
    return torch.argmax(x, dim=axis)


def ones(shape, dtype=None):
    return torch.ones(tuple(int(d) for d in shape), dtype=dtype or torch.float32)


def zeros(shape, dtype=None, name=None):
    return torch.zeros(tuple(int(d) for d in shape), dtype=dtype or torch.float32)


def reduce_sum(x, axis=None, keepdims=False):
    return x.sum() if axis is None else x.sum(dim=axis, keepdim=keepdims)


def reduce_mean(x, axis=None, keepdims=False):
    return x.mean() if axis is None else x.mean(dim=axis, keepdim=keepdims)


def reduce_max(x, axis=None, keepdims=False):
    return x.max() if axis is None else x.max(dim=axis, keepdim=keepdims).values


def maximum(a, b):
    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
        return max(a, b)
    a = _t(a); return torch.maximum(a, _t(b, a.dtype))


def minimum(a, b):
    if isinstance(a, (int, float)) and isinstance(b, (int, float)):
        return min(a, b)
    a = _t(a); return torch.minimum(a, _t(b, a.dtype))


def less(a, b):
    return a < b


def equal(a, b):
    return a == b


def cond(pred, true_fn, false_fn):
    return true_fn() if bool(pred) else false_fn()


def assert_equal(a, b, message=None):
    ok = bool(torch.as_tensor(np.asarray(a) == np.asarray(b)).all()) if not isinstance(a, torch.Tensor) else bool((a == b).all())
    if not ok:
        raise AssertionError(message or 'tf.assert_equal failed: %r != %r' % (a, b))


def matmul(a, b):
    return a @ b


def batch_to_space_nd(x, block_shape, crops):
    """block_shape [M], crops [[0,0]] (all the reference uses): batch = prod(block) * b';  out[b', i*block + j, ...] = in[j*b' + b', i, ...]"""
    assert len(block_shape) == 1 and np.asarray(crops).tolist() == [[0, 0]]
    r = int(block_shape[0]); n = x.shape[0] // r
    y = x.reshape((r, n) + tuple(x.shape[1:]))                  # [j, b', i, rest]
    y = y.permute(1, 2, 0, *range(3, y.dim()))                  # [b', i, j, rest]
    return y.reshape((n, x.shape[1] * r) + tuple(x.shape[2:]))


def resize_images(images, size, method=1):
    """NHWC nearest-neighbour (method=1), integer scale factors, align_corners=False: out[i] = in[floor(i * in/out)]."""
    assert method == 1
    H, W = int(size[0]), int(size[1])
    h, w = images.shape[1], images.shape[2]
    ih = torch.arange(H) * h // H; iw = torch.arange(W) * w // W
    return images[:, ih][:, :, iw]


def dropout(x, rate=0.5, training=False):
    """tf.layers.dropout -> tf.nn.dropout(x, keep_prob = 1 - rate): binary = floor(keep_prob + U[0,1)), out = x / keep_prob * binary.
    TensorFlow's RNG stream cannot be reproduced, so the keep masks are drawn from the stand-in's generator and RECORDED in call
    order (_STATE.dropout_masks): the golden files carry them and the oracle is handed the same masks -- what the reference's code
    decides (WHERE the op sits, on which tensor, and that the kept values are scaled by 1 / (1 - rate)) is executed, not restated."""
    if not training or rate == 0:
        return x
    keep = 1.0 - float(rate)
    binary = torch.floor(keep + torch.rand(x.shape, generator=_STATE.gen))
    _STATE.dropout_masks.append(binary.clone())
    return x / keep * binary


def random_uniform(s, minval=0.0, maxval=1.0, dtype=None):
    u = torch.rand(tuple(int(d) for d in s), generator=_STATE.gen) * (maxval - minval) + minval
    _STATE.uniform_draws.append(u)
    return u


class _Normal:
    def __init__(self, loc, scale, allow_nan_stats=True):
        self.loc, self.scale = loc, scale

    def cdf(self, x):      # tf.contrib.distributions.Normal.cdf == special_math.ndtr
        half_sqrt_2 = 0.5 * np.sqrt(2.0)
        w = ((x - self.loc) / self.scale) * half_sqrt_2
        z = torch.abs(w)
        y = torch.where(z < half_sqrt_2, 1.0 + torch.erf(w), torch.where(w > 0, 2.0 - torch.erfc(z), torch.erfc(z)))
        return 0.5 * y

    def sample(self):
        eps = torch.randn(self.loc.shape, generator=_STATE.gen)
        _STATE.normal_draws.append(eps)
        return self.loc + self.scale * eps


class TensorArray:
    def __init__(self, dtype=None, size=0, dynamic_size=True, **k):
        self._items = {}

    def write(self, i, v):
        self._items[int(i)] = v
        return self

    def stack(self):
        return torch.stack([self._items[i] for i in range(len(self._items))], dim=0)


def while_loop(cond_fn, body_fn, loop_vars, **k):
    vars_ = list(loop_vars)
    while bool(cond_fn(*vars_)):
        vars_ = list(body_fn(*vars_))
    return vars_


def sequence_mask(lengths, maxlen=None, dtype=None):
    lengths = _t(lengths).long().reshape(-1)
    maxlen = int(maxlen if maxlen is not None else lengths.max())
    m = torch.arange(maxlen)[None, :] < lengths[:, None]
    return m.to(dtype or torch.bool)


def count_nonzero(x, dtype=None):
    return (x != 0).sum().to(dtype or torch.int64)


def Print(x, data, *a, **k):
    return x


def install():
    """Register `tensorflow` (+ the few third-party modules the reference files import at module level) in sys.modules."""
    tf = types.ModuleType('tensorflow')
    for name, obj in globals().items():
        if not name.startswith('_') and name not in ('install', 'reset', 'variables', 'np', 'torch', 'F', 'sys', 'types', 'math', 'contextlib'):
            setattr(tf, name, obj)
    tf.float32, tf.int32, tf.int64, tf.bool = torch.float32, torch.int32, torch.int64, torch.bool
    tf.control_dependencies = _noop_ctx
    tf.device = _noop_ctx
    tf.log, tf.exp, tf.sqrt, tf.abs, tf.tanh, tf.sigmoid = torch.log, torch.exp, torch.sqrt, torch.abs, torch.tanh, torch.sigmoid
    tf.square = lambda x: x * x
    tf.where = lambda c, a, b: torch.where(c, _t(a), _t(b))
    tf.mod = lambda a, b: a % b
    tf.clip_by_value = lambda x, lo, hi: torch.clamp(x, lo, hi)
    tf.norm = lambda x, axis=None: torch.linalg.vector_norm(x) if axis is None else torch.linalg.vector_norm(x, dim=axis)
    tf.layers = types.SimpleNamespace(Layer=Layer, Conv1D=Conv1D, Conv2D=Conv2D, Conv2DTranspose=Conv2DTranspose, InputSpec=InputSpec, dropout=dropout)
    tf.keras = types.SimpleNamespace(layers=types.SimpleNamespace(Wrapper=Wrapper))
    tf.image = types.SimpleNamespace(resize_images=resize_images)
    tf.nn = types.SimpleNamespace(
        tanh=torch.tanh, sigmoid=torch.sigmoid, relu=lambda x, name=None: F.relu(x),
        leaky_relu=lambda features, alpha=0.2, name=None: F.leaky_relu(features, alpha),
        softmax=lambda x, axis=-1: torch.softmax(x, dim=axis), softplus=F.softplus,
        bias_add=lambda x, b: x + b, embedding_lookup=lambda table, ids: table[_t(ids).long()],
        log_softmax=lambda x, axis=-1: torch.log_softmax(x, dim=axis),
        l2_normalize=lambda x, axis=None, epsilon=1e-12: x * torch.rsqrt(torch.clamp((x * x).sum(dim=tuple(axis) if isinstance(axis, (list, tuple)) else axis, keepdim=True), min=epsilon)),
        moments=lambda x, axes: (x.mean(dim=tuple(axes)), x.var(dim=tuple(axes), unbiased=False)),
        softmax_cross_entropy_with_logits_v2=lambda logits, labels: -(labels * torch.log_softmax(logits, dim=-1)).sum(-1))
    tf.train = types.SimpleNamespace(replica_device_setter=lambda *a, **k: None)
    tf.contrib = types.SimpleNamespace(distributions=types.SimpleNamespace(Normal=_Normal),
                                       training=types.SimpleNamespace(HParams=lambda **kw: types.SimpleNamespace(**kw)))
    tf.GraphKeys = types.SimpleNamespace(UPDATE_OPS='update_ops')
    sys.modules['tensorflow'] = tf
    # module-level imports of the reference files that are irrelevant to the arithmetic
    for m in ('librosa', 'librosa.filters', 'librosa.display', 'librosa.core', 'keras', 'keras.utils'):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules['librosa'].filters = sys.modules['librosa.filters']; sys.modules['librosa'].display = sys.modules['librosa.display']
    npu = types.ModuleType('keras.utils.np_utils')
    npu.to_categorical = lambda y, num_classes=None: np.eye(int(num_classes if num_classes is not None else np.max(y) + 1), dtype=np.float32)[np.asarray(y).astype(int)]
    sys.modules['keras.utils.np_utils'] = npu
    sys.modules['keras.utils'].np_utils = npu; sys.modules['keras'].utils = sys.modules['keras.utils']
    torch.Tensor.get_shape = lambda self: tuple(self.shape)
    torch.Tensor.assign = lambda self, v: self.copy_(v)           # tf.Variable.assign (WeightNorm.build, modules.py:170)
    if not hasattr(torch.Size, 'ndims'):
        pass       # mixture.py:7,14 call x.get_shape() on tensors
    if not hasattr(np, 'int'):
        np.int = int          # removed in numpy >= 1.24; it always was the builtin
    return tf
