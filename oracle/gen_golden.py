"""Generate golden vectors by executing the REFERENCE's own source files in this container.

Run here (container with /root/reference mounted):   python oracle/gen_golden.py
Writes tests/golden/{mulaw,mol,gaussian}_golden.npz.  The GPU box has no /root/reference, so
only the committed .npz files travel.

What is executed from the reference, unmodified (loaded by path with importlib):
  * wavenet_vocoder/util.py     mulaw / inv_mulaw / mulaw_quantize / inv_mulaw_quantize on
    numpy arrays -- the file's numpy code path, no TensorFlow op involved.  (`np.int`, removed in
    numpy >= 1.24, is aliased to `int`, which is what it always was.)
  * wavenet_vocoder/models/mixture.py, gaussian.py -- their tf.* calls are served by the small
    eager stand-in below (`_TF`), which maps each op the two files use 1:1 onto the torch op with
    TF's documented semantics (tf.where = element-wise select, tf.nn.softplus/sigmoid = stable
    forms, reduce_* with keepdims, tf.maximum ...).  The *composition* (which branch, which
    clamp, which constant) is the reference's code, not ours.
TensorFlow itself is not installed (no network); conv stack / while_loop files are not run.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.environ.get('WN_GOLDEN_DIR') or os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden')      # WN_GOLDEN_DIR: write elsewhere (regen check of __graft_entry__.regen_golden)


class _Recorder:
    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.draws = []


_rec = _Recorder(1234)


class _Normal:
    def __init__(self, loc, scale, allow_nan_stats=True):
        self.loc, self.scale = loc, scale

    def cdf(self, x):
        # tf.contrib.distributions.Normal.cdf == special_math.ndtr((x - loc) / scale)
        # (tensorflow/python/ops/distributions/{normal,special_math}.py, TF 1.x; version unpinned)
        half_sqrt_2 = 0.5 * np.sqrt(2.0)
        w = ((x - self.loc) / self.scale) * half_sqrt_2
        z = torch.abs(w)
        y = torch.where(z < half_sqrt_2, 1.0 + torch.erf(w),
                        torch.where(w > 0, 2.0 - torch.erfc(z), torch.erfc(z)))
        return 0.5 * y

    def sample(self):
        eps = torch.randn(self.loc.shape, generator=_rec.g)
        _rec.draws.append(eps)
        return self.loc + self.scale * eps


class _CD:
    def __enter__(self): return self
    def __exit__(self, *a): return False


def _make_tf():
    tf = types.ModuleType('tensorflow')
    tf.float32 = torch.float32
    tf.int32 = torch.int32
    tf.reduce_max = lambda x, axis=None, keepdims=False: x.max(dim=axis, keepdim=keepdims).values
    tf.reduce_sum = lambda x, axis=None, keepdims=False: (x.sum() if axis is None else x.sum(dim=axis, keepdim=keepdims))
    tf.log = torch.log
    tf.exp = torch.exp
    tf.square = lambda x: x * x
    tf.control_dependencies = lambda deps: _CD()
    tf.assert_equal = lambda a, b, message=None: (_ for _ in ()).throw(AssertionError(message)) if not bool(torch.as_tensor(a == b).all()) else None
    tf.mod = lambda a, b: a % b
    tf.shape = lambda x: tuple(x.shape)
    tf.rank = lambda x: x.dim()
    tf.transpose = lambda x, perm: x.permute(*perm)
    tf.maximum = lambda a, b: torch.maximum(a, torch.as_tensor(b, dtype=a.dtype))
    tf.minimum = lambda a, b: torch.minimum(a, torch.as_tensor(b, dtype=a.dtype))
    tf.ones = lambda shape, dtype=torch.float32: torch.ones(tuple(int(s) for s in shape), dtype=dtype)
    tf.where = lambda c, a, b: torch.where(c, a, b)
    tf.expand_dims = lambda x, axis: x.unsqueeze(axis[0] if isinstance(axis, (list, tuple)) else axis)
    tf.squeeze = lambda x, axis=None: x.squeeze(axis[0] if isinstance(axis, (list, tuple)) else axis)
    tf.argmax = lambda x, axis: x.argmax(dim=axis)
    tf.one_hot = lambda idx, depth, dtype=torch.float32: torch.nn.functional.one_hot(idx, int(depth)).to(dtype)

    def random_uniform(shape, minval=0., maxval=1.):
        u = torch.rand(tuple(int(s) for s in shape), generator=_rec.g) * (maxval - minval) + minval
        _rec.draws.append(u)
        return u
    tf.random_uniform = random_uniform
    nn = types.ModuleType('tensorflow.nn')
    nn.sigmoid = torch.sigmoid
    nn.softplus = torch.nn.functional.softplus
    tf.nn = nn
    contrib = types.ModuleType('tensorflow.contrib')
    dist = types.ModuleType('tensorflow.contrib.distributions')
    dist.Normal = _Normal
    contrib.distributions = dist
    tf.contrib = contrib
    return tf


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    os.makedirs(OUT, exist_ok=True)
    if not hasattr(np, 'int'):
        np.int = int                                   # util.py:156 uses the removed alias
    torch.Tensor.get_shape = lambda self: tuple(self.shape)   # mixture.py:7,14
    sys.modules['tensorflow'] = _make_tf()
    # util.py imports these at module level for its plotting helpers only
    for n in ('librosa', 'librosa.display'):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules['librosa'].display = sys.modules['librosa.display']

    util = _load('ref_util', os.path.join(REF, 'wavenet_vocoder', 'util.py'))
    mixture = _load('ref_mixture', os.path.join(REF, 'wavenet_vocoder', 'models', 'mixture.py'))
    gaussian = _load('ref_gaussian', os.path.join(REF, 'wavenet_vocoder', 'models', 'gaussian.py'))

    rng = np.random.default_rng(5339)
    # ---- mu-law: dense sweep incl. exact bin edges, +-1, 0, denormals, and random audio-like data
    xs = np.concatenate([
        np.linspace(-1.0, 1.0, 20001),
        rng.uniform(-1, 1, 50000),
        np.array([0.0, -0.0, 1.0, -1.0, 1e-8, -1e-8, 0.999, -0.999, 0.5, -0.5]),
    ]).astype(np.float32)
    # exact decision boundaries of the quantiser, nudged one ulp either side
    edges = util.inv_mulaw(2 * np.arange(1, 255, dtype=np.float64) / 255 - 1).astype(np.float32)
    xs = np.concatenate([xs, edges, np.nextafter(edges, np.float32(2)), np.nextafter(edges, np.float32(-2))])
    q = util.mulaw_quantize(xs)
    np.savez_compressed(
        os.path.join(OUT, 'mulaw_golden.npz'),
        x=xs, mulaw=util.mulaw(xs), quantized=q.astype(np.int64),
        inv_q_all=util.inv_mulaw_quantize(np.arange(256)),          # full decode table
        inv_mulaw=util.inv_mulaw(np.linspace(-1, 1, 4097).astype(np.float32)),
        x64=xs.astype(np.float64), quantized64=util.mulaw_quantize(xs.astype(np.float64)).astype(np.int64),
        q0=np.int64(util.mulaw_quantize(0)), m0=np.float64(util.mulaw(0.0)))

    # ---- MoL loss (reduce=False) + sampler: covers every tf.where branch
    B, T, M = 3, 257, 10
    g = torch.Generator().manual_seed(5339)
    y_hat = torch.randn(B, 3 * M, T, generator=g)
    y_hat[:, 2 * M:, :] = y_hat[:, 2 * M:, :] * 4 - 6           # log-scales from wide (-2) to tiny (-30)
    y_hat[0, 2 * M:, :64] = -40.0                                 # below log_scale_min -> clamp branch
    y = (torch.rand(B, T, 1, generator=g) * 2 - 1)
    y[0, :8, 0] = -1.0; y[0, 8:16, 0] = 1.0; y[1, :8, 0] = -0.9995; y[1, 8:16, 0] = 0.9995
    outs = {}
    for nc, lsm in ((65536, float(np.log(1e-14))), (256, -7.0)):
        loss = mixture.discretized_mix_logistic_loss(y_hat, y, num_classes=nc, log_scale_min=lsm, reduce=False)
        red = mixture.discretized_mix_logistic_loss(y_hat, y, num_classes=nc, log_scale_min=lsm, reduce=True)
        outs['loss_%d' % nc] = loss.numpy(); outs['loss_sum_%d' % nc] = red.numpy()
    _rec.draws.clear()
    smp = mixture.sample_from_discretized_mix_logistic(y_hat, log_scale_min=float(np.log(1e-14)))
    u1, u2 = _rec.draws
    np.savez_compressed(os.path.join(OUT, 'mol_golden.npz'), y_hat=y_hat.numpy(), y=y.numpy(),
                        sample=smp.numpy(), u1=u1.numpy(), u2=u2.numpy(), **outs)

    # ---- Gaussian loss (pdf + cdf forms) + sampler
    y_hat2 = torch.randn(B, 2, T, generator=g)
    y_hat2[:, 1, :] = y_hat2[:, 1, :] * 3 - 5
    y_hat2[0, 1, :32] = -30.0
    outs = {}
    for lsm in (float(np.log(1e-7)), float(np.log(9.1188196e-4))):
        for cdf in (False, True):
            l = gaussian.gaussian_maximum_likelihood_estimation_loss(
                y_hat2, y, log_scale_min_gauss=lsm, num_classes=65536, use_cdf=cdf, reduce=False)
            outs['loss_cdf%d_%s' % (int(cdf), 'a' if lsm < -10 else 'b')] = l.numpy()
    _rec.draws.clear()
    smp = gaussian.sample_from_gaussian(y_hat2, log_scale_min_gauss=float(np.log(1e-7)))
    (eps,) = _rec.draws
    np.savez_compressed(os.path.join(OUT, 'gaussian_golden.npz'), y_hat=y_hat2.numpy(), y=y.numpy(),
                        sample=smp.numpy(), eps=eps.numpy(), **outs)
    print('golden vectors written to', os.path.normpath(OUT))


if __name__ == '__main__':
    main()
