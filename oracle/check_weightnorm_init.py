"""Does the reference's data-dependent weight-normalisation initialisation change any variable?

The reference runs, on a fresh training with `wavenet_weight_normalization` (train.py:287-298):
    init_model, _ = model_train_mode(args, feeder, hparams, global_step, init=True)
    _ = sess.run(init_model.tower_y_hat)
i.e. ONE forward pass of a second WaveNet instance whose WeightNorm wrappers were built with init=True, so that their `call`
goes through `_data_dep_init` (modules.py:110-126).  That method computes per-channel moments of the layer output and then REBINDS
Python attributes of the wrapped layer object (`self.layer.g = self.layer.g * scale_init`, `self.layer.bias = -m_init * scale_init`)
-- it never calls `assign` on a variable, the rescaled tensors are not part of what `tower_y_hat` depends on (`x_init` was computed
before the rebinding, from the kernel formed at build time), and the instance is thrown away afterwards; the training model is a
different set of layer objects that shares only the VARIABLES (tf.AUTO_REUSE).

This script executes exactly that path with the reference's own wavenet.py / modules.py (imported unmodified from /root/reference)
on the eager TF-1 stand-in (oracle/tf1_shim.py) and records, in tests/golden/weightnorm_datadep_init.npz:
    raised / error           whether the init forward pass raised, and the message
    max_abs_change           max |variable_after - variable_before| over every variable (v, g, bias, ...) of the model
    n_variables, n_wrappers  how many variables / WeightNorm wrappers took part
    y_train_model_after_vs_before   the training model's forward on the same batch before / after the init pass
FINDING (recorded in the fixture): the pass cannot complete.  `_data_dep_init` takes the moments over `norm_axes` = the kernel's
leading axes [0, 1] of a channels-first activation [B, filters, T], gets [T] statistics and multiplies them into g [filters]:
a shape error at the first wrapped layer for every T != filters.  No variable is touched before the error.
tests/test_oracle_golden.py reads the fixture; DESIGN.md records the quirk.  TEST INFRASTRUCTURE; nothing here ships.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import gen_golden_stack as G  # noqa: E402
from oracle import tf1_shim as shim  # noqa: E402

OUT = os.environ.get('WN_GOLDEN_DIR') or os.path.join(ROOT, 'tests', 'golden')


def _wrappers(obj, seen, out):
    if id(obj) in seen or not hasattr(obj, '__dict__'):
        return
    seen.add(id(obj))
    if type(obj).__name__ == 'WeightNorm':
        out.append(obj)
    for vv in list(vars(obj).values()):
        for o in (vv if isinstance(vv, (list, tuple)) else [vv]):
            _wrappers(o, seen, out)


def main():
    if not os.path.isdir(os.path.join(G.REF, 'wavenet_vocoder')):
        raise SystemExit('needs /root/reference (run in the build container)')
    wn = G._import_reference()
    mods = sys.modules['wavenet_vocoder.models.modules']
    shim.reset(seed=4242)
    hp = G._hparams(wavenet_weight_normalization=True, NN_init=False, wavenet_init_scale=1.0)
    gen = torch.Generator().manual_seed(3)
    B, Tc = 2, 5
    T = Tc * hp.hop_size
    wav = torch.rand(B, T, generator=gen) * 1.6 - 0.8
    c = torch.rand(B, hp.cin_channels, Tc, generator=gen)
    x = wav.view(B, 1, T)

    # the training model first (creates + initialises every variable, like the graph the session initialises)
    model = wn.WaveNet(hp, init=False)
    model.set_mode(True)
    y_plain = model.step(x, c=c, g=None, softmax=False)
    before = {k: v.detach().clone() for k, v in shim.variables().items()}

    # count what _data_dep_init computes
    scales = []
    orig = mods.WeightNorm._data_dep_init

    def spy(self, inputs):
        g_before = self.layer.g
        out = orig(self, inputs)
        scales.append((torch.as_tensor(self.layer.g) / torch.as_tensor(g_before)).detach().reshape(-1))
        return out
    mods.WeightNorm._data_dep_init = spy
    error = ''
    y_init = None
    try:
        init_model = wn.WaveNet(hp, init=True)                      # train.py:290
        init_model.set_mode(True)
        try:
            y_init = init_model.step(x, c=c, g=None, softmax=False)     # == sess.run(init_model.tower_y_hat), train.py:297
        except RuntimeError as e:
            # tf.nn.moments(x_init, norm_axes) reduces the axes [0, 1] of the KERNEL's rank (modules.py:114, 157) on a channels-first
            # activation [B, filters, T]: the moments have shape [T], and `g [filters] * scale_init [T]` does not broadcast -- in
            # TensorFlow this is an InvalidArgumentError at session.run (T is a placeholder dimension), caught by train()'s blanket
            # `except Exception` (train.py:339-342), i.e. a FRESH weight-normalised training of the reference stops right there.
            error = str(e)
    finally:
        mods.WeightNorm._data_dep_init = orig
    after = shim.variables()
    assert set(after) == set(before), 'the init instance created new variables'
    change = max(float((after[k] - before[k]).abs().max()) for k in before)
    ws = []
    _wrappers(init_model, set(), ws)
    sc = torch.cat(scales) if scales else torch.zeros(1)
    # and the training model (its own layer objects, shared variables) computes what it computed before
    y_again = model.step(x, c=c, g=None, softmax=False)
    res = dict(max_abs_change=np.float64(change), n_variables=np.int64(len(before)), n_wrappers=np.int64(len(ws)),
               n_data_dep_init_calls=np.int64(len(scales)), raised=np.bool_(bool(error)), error=np.array(error),
               y_init_vs_plain=np.float64(float((y_init - y_plain).abs().max()) if y_init is not None else -1.0),
               y_train_model_after_vs_before=np.float64(float((y_again - y_plain).abs().max())),
               first_layer_filters=np.int64(hp.residual_channels), time_steps=np.int64(T),
               scale_init_range=np.array([float(sc.min()), float(sc.max())]))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, 'weightnorm_datadep_init.npz'), **res)
    print({k: (v.tolist() if hasattr(v, 'tolist') else v) for k, v in res.items()})


if __name__ == '__main__':
    main()
