"""Golden vectors for the optimiser row, produced by executing the REFERENCE'S OWN `WaveNet.add_loss` +
`WaveNet.add_optimizer` (wavenet.py:476-613, imported unmodified from /root/reference) on the eager TF-1 stand-in for three
training steps of a small model.

    python oracle/gen_golden_optim.py          (only in the container that has /root/reference)

What this pins, and what it cannot:
  * PINNED by execution of the reference's code: which gradients exist and in which order they meet the variables
    (`optimizer.compute_gradients(tower_loss[i])`), the tower average (expand_dims / concat / reduce_mean, :560-575), the clipping
    COMPOSITION -- per variable, `tf.clip_by_norm(g, wavenet_gradient_max_norm)` THEN `tf.clip_by_value(+-wavenet_gradient_max_value)`
    (:586-598), never a global norm --, Adam applied to the clipped gradients with the scheduled learning rate (:533-549, :600-602),
    the moving average taken AFTER the Adam update of the same step on the updated variables (:604-613), the learning-rate schedule.
  * NOT pinnable offline (TensorFlow is not installable here): the arithmetic inside the four TF library calls.  The stand-in
    implements them from TensorFlow 1.x's own documentation / kernels, stated here so that they can be checked against TF:
      tf.clip_by_norm(t, c)            t * c / max(||t||_2, c)                               (clip_ops.py: `t * clip_norm / maximum(l2norm, clip_norm)`)
      tf.train.AdamOptimizer           lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t);  m += (g - m)(1 - b1);  v += (g^2 - v)(1 - b2);
                                       var -= lr_t * m / (sqrt(v) + eps)                       (training_ops ApplyAdam, "epsilon hat" form; t counts from 1)
      ExponentialMovingAverage.apply   shadow -= (1 - decay) * (shadow - var), shadow initialised to the variable's initial value,
                                       no num_updates, no zero_debias                           (moving_averages.assign_moving_average)
      tf.train.exponential_decay       lr * decay_rate ^ (global_step / decay_steps)
    (1 - beta) factors are float32 subtractions, as in the float32 hyper-parameter tensors of the TF kernels.
Writes tests/golden/optim_golden.npz; tests/test_oracle_golden.py requires oracle.train_step (the function the device optimiser is
tested against) to reproduce parameters, moving averages, gradients and losses of all three steps.  TEST INFRASTRUCTURE.
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import gen_golden_stack as G  # noqa: E402
from oracle import tf1_shim as shim  # noqa: E402

OUT = os.environ.get('WN_GOLDEN_DIR') or os.path.join(ROOT, 'tests', 'golden')
_SLOTS = {}       # Adam slot variables + beta powers: graph variables in TF, so they outlive any optimizer OBJECT


class AdamOptimizer(object):
    """tf.train.AdamOptimizer stand-in (see the module docstring for the formula source)."""

    def __init__(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), float(beta1), float(beta2), float(epsilon)

    def compute_gradients(self, loss):
        names = list(shim.variables().keys())
        vs = [shim.variables()[n] for n in names]
        gs = torch.autograd.grad(loss, vs, allow_unused=True, retain_graph=True)
        return [(None if g is None else g.detach(), v) for g, v in zip(gs, vs)]

    def apply_gradients(self, grads_and_vars, global_step=None):
        f32 = np.float32
        st = _SLOTS.setdefault('powers', {'t': 0})
        st['t'] += 1
        t = st['t']
        lr_t = self.lr * math.sqrt(1 - self.b2 ** t) / (1 - self.b1 ** t)
        one_m_b1, one_m_b2 = float(f32(1) - f32(self.b1)), float(f32(1) - f32(self.b2))
        with torch.no_grad():
            for g, v in grads_and_vars:
                if g is None:
                    continue
                key = id(v)
                m, vv = _SLOTS.setdefault(key, (torch.zeros_like(v), torch.zeros_like(v)))
                m.add_((g - m) * one_m_b1)
                vv.add_((g * g - vv) * one_m_b2)
                v.sub_(lr_t * m / (torch.sqrt(vv) + self.eps))
            if global_step is not None:
                global_step.add_(1)
        return 'adam_optimize'


class ExponentialMovingAverage(object):
    def __init__(self, decay):
        self.decay = float(decay)
        self.shadow = {}

    def snapshot(self, variables):
        """TF initialises every shadow with the variable's INITIAL value when the variables are initialised."""
        for v in variables:
            self.shadow[id(v)] = v.detach().clone()

    def apply(self, variables):
        one_m = float(np.float32(1) - np.float32(self.decay))
        with torch.no_grad():
            for v in variables:
                s = self.shadow[id(v)]
                s.sub_(one_m * (s - v))
        return 'ema_apply'

    def average(self, v):
        return self.shadow[id(v)]


def install_optimizer_standins(tf):
    tf.train.AdamOptimizer = AdamOptimizer
    tf.train.ExponentialMovingAverage = ExponentialMovingAverage
    tf.train.exponential_decay = lambda lr, step, decay_steps, decay_rate, staircase=False, name=None: lr * decay_rate ** (float(step) / decay_steps)
    tf.clip_by_norm = lambda t, clip_norm: t * clip_norm / torch.clamp(torch.linalg.vector_norm(t), min=float(clip_norm))
    extra = dict(
        get_collection=lambda *a, **k: [],
        trainable_variables=lambda: list(shim.variables().values()),
        expand_dims=lambda x, axis: x.unsqueeze(axis[0] if isinstance(axis, (list, tuple)) else axis),
        concat=lambda values=None, axis=0, **k: torch.cat(list(values), dim=axis),
        reduce_mean=lambda x, axis=None, **k: x.mean() if axis is None else x.mean(dim=axis),
        cast=lambda x, dtype=None: (x.float() if dtype in (torch.float32, None) else x.to(dtype)) if torch.is_tensor(x) else float(x),
        maximum=lambda a, b: torch.maximum(torch.as_tensor(a, dtype=torch.float32), torch.as_tensor(b, dtype=torch.float32)),
        minimum=lambda a, b: torch.minimum(torch.as_tensor(a, dtype=torch.float32), torch.as_tensor(b, dtype=torch.float32)))
    for k, f in extra.items():          # only what the shim does not already serve (the forward / loss goldens pin its own versions)
        if not hasattr(tf, k):
            setattr(tf, k, f)

def main():
    if not os.path.isdir(os.path.join(G.REF, 'wavenet_vocoder')):
        raise SystemExit('needs /root/reference (run in the build container)')
    wn = G._import_reference()
    tf = sys.modules['tensorflow']
    install_optimizer_standins(tf)
    shim.reset(seed=777)
    _SLOTS.clear()
    hp = G._hparams(wavenet_dropout=0.0, NN_init=False)
    # optimiser hparams read by add_optimizer (hparams.py:301-327); clip thresholds low enough that BOTH clips bite
    hp.wavenet_num_gpus = 1; hp.tacotron_num_gpus = 1
    hp.wavenet_lr_schedule = 'exponential'; hp.wavenet_learning_rate = 1e-3; hp.wavenet_decay_rate = 0.5; hp.wavenet_decay_steps = 4
    hp.wavenet_warmup = 4000.0
    hp.wavenet_adam_beta1 = 0.9; hp.wavenet_adam_beta2 = 0.999; hp.wavenet_adam_epsilon = 1e-6; hp.wavenet_ema_decay = 0.9
    hp.wavenet_clip_gradients = True; hp.wavenet_gradient_max_norm = 0.02; hp.wavenet_gradient_max_value = 0.004
    gen = torch.Generator().manual_seed(11)
    B, Tc = 2, 5
    T = Tc * hp.hop_size
    wav = torch.rand(B, T, generator=gen) * 1.6 - 0.8
    c = torch.rand(B, hp.cin_channels, Tc, generator=gen)
    x = wav.view(B, 1, T)
    lengths = [T, T - 7]

    model = wn.WaveNet(hp, init=False)
    model.set_mode(True)
    model.step(x, c=c, g=None, softmax=False)                        # builds every variable
    for k, v in shim.variables().items():
        if k.endswith('/bias'):
            v.copy_((torch.rand(v.shape, generator=gen) * 2 - 1) * 0.1)
    names = list(shim.variables().keys())
    for v in shim.variables().values():
        v.requires_grad_(True)
    model.variables = list(shim.variables().values())                # wavenet.py:467 tf.trainable_variables()
    model.ema = tf.train.ExponentialMovingAverage(decay=hp.wavenet_ema_decay)      # wavenet.py:473
    model.ema.snapshot(model.variables)
    global_step = torch.zeros((), dtype=torch.int64)
    out = {'x': x.numpy(), 'c': c.numpy(), 'wav': wav.numpy(), 'lengths': np.array(lengths)}
    for k in names:
        out['p0/' + G._oracle_name(k)] = shim.variables()[k].detach().numpy().copy()
    nsteps = 3
    for s in range(nsteps):
        y_hat = model.step(x, c=c, g=None, softmax=False)
        model.is_training = True; model.is_evaluating = False
        model.tower_mask = [model.get_mask(torch.tensor(lengths, dtype=torch.int32), maxlen=T)]
        model.tower_y_hat_train = [y_hat]
        model.tower_y = [wav.view(B, T, 1)]
        model.add_loss()                                             # wavenet.py:476-519
        model.add_optimizer(global_step)                             # wavenet.py:522-613  (the update happens here: eager)
        out['loss/%d' % s] = np.float32(float(model.loss))
        out['lr/%d' % s] = np.float32(float(model.learning_rate))
        for k, g, cg in zip(names, model.gradients, [None] * len(names)):
            out['g%d/%s' % (s, G._oracle_name(k))] = (g if g is not None else torch.zeros_like(shim.variables()[k])).detach().numpy().copy()
        for k in names:
            v = shim.variables()[k]
            out['p%d/%s' % (s + 1, G._oracle_name(k))] = v.detach().numpy().copy()
            out['ema%d/%s' % (s + 1, G._oracle_name(k))] = model.ema.average(v).numpy().copy()
    assert int(global_step) == nsteps
    out['hparams_json'] = np.array(__import__('json').dumps({k: getattr(hp, k) for k in (
        'wavenet_learning_rate', 'wavenet_decay_rate', 'wavenet_decay_steps', 'wavenet_adam_beta1', 'wavenet_adam_beta2',
        'wavenet_adam_epsilon', 'wavenet_ema_decay', 'wavenet_gradient_max_norm', 'wavenet_gradient_max_value')}))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, 'optim_golden.npz'), **out)
    gn = {k: float(np.linalg.norm(v)) for k, v in out.items() if k.startswith('g0/')}
    print('optim_golden.npz: %d arrays; losses %s; lrs %s; %d of %d gradient tensors exceed the norm clip' % (
        len(out), [float(out['loss/%d' % s]) for s in range(nsteps)], [float(out['lr/%d' % s]) for s in range(nsteps)],
        sum(1 for v in gn.values() if v > hp.wavenet_gradient_max_norm), len(gn)))


if __name__ == '__main__':
    main()
