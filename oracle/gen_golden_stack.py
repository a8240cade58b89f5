"""Golden vectors for the conv stack / upsample net / incremental loop, produced by executing the REFERENCE'S OWN
`wavenet_vocoder/models/{wavenet,modules,mixture,gaussian}.py` (unmodified, imported from /root/reference) on the eager
TF-1 stand-in of oracle/tf1_shim.py.

    python oracle/gen_golden_stack.py          (only in the container that has /root/reference)

Writes tests/golden/stack_<config>.npz with, per configuration:
    params/<oracle name>   the variables the reference model created (Glorot / NN-init kernels from the reference's own
                           initialiser code; biases replaced by small random values so that the bias paths carry signal)
    x, c[, g]              inputs
    y_hat                  WaveNet.step(x, c, g)                      (wavenet.py:650-721, teacher-forced batch forward)
    c_up                   self.upsampled_local_features              (wavenet.py:680-702)
    inc_raw, inc_out       WaveNet.incremental(...) raw network outputs per step and samples (wavenet.py:724-911), run
                           teacher-forced (test_inputs) and, for scalar inputs, free-running with the recorded sampler noise
tests/test_oracle_golden.py then requires oracle.step / oracle.upsample / oracle.incremental to reproduce them (fp32
rtol 1e-5): that pins the restatement to the reference's composition.  TEST INFRASTRUCTURE; nothing here ships.
"""
import importlib
import os
import re
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.environ.get('WN_GOLDEN_DIR') or os.path.join(ROOT, 'tests', 'golden')      # WN_GOLDEN_DIR: write elsewhere (regen check of __graft_entry__.regen_golden)

sys.path.insert(0, ROOT)
from oracle import tf1_shim as shim  # noqa: E402


def _hparams(**over):
    """The hparams keys wavenet.py / modules.py read (reference hparams.py:187-233, 323-324), small sizes."""
    d = dict(layers=4, stacks=2, residual_channels=16, gate_channels=32, skip_out_channels=16, out_channels=6, kernel_size=3,
             cin_channels=8, num_mels=8, gin_channels=-1, use_speaker_embedding=True, n_speakers=3, input_type='raw',
             quantize_channels=65536, use_bias=True, legacy=False, residual_legacy=False, wavenet_dropout=0.0,
             wavenet_weight_normalization=False, wavenet_init_scale=1.0, upsample_type='2D', upsample_scales=[2, 3],
             upsample_activation='Relu', leaky_alpha=0.4, freq_axis_kernel_size=3, NN_init=True, NN_scaler=0.3,
             hop_size=6, frame_shift_ms=None, sample_rate=22050, wavenet_swap_with_cpu=False,
             log_scale_min=float(np.log(1e-14)), log_scale_min_gauss=float(np.log(1e-7)), cdf_loss=False)
    d.update(over)
    d['hop_size'] = int(np.prod(d['upsample_scales']))
    return types.SimpleNamespace(**d)


CONFIGS = {
    'mol_2d': dict(),
    'mol_2d_legacy': dict(legacy=True, residual_legacy=True, upsample_activation='LeakyRelu', NN_init=False),
    'gauss_subpixel': dict(out_channels=2, upsample_type='SubPixel', upsample_scales=[3, 2], cdf_loss=True),
    'gauss_subpixel_nn_off': dict(out_channels=2, upsample_type='SubPixel', upsample_scales=[2, 2], NN_init=False, upsample_activation=None),
    'mol_resize': dict(upsample_type='Resize', upsample_scales=[2, 3], NN_init=False),
    'mol_resize_nninit': dict(upsample_type='Resize', upsample_scales=[3, 4]),
    'mol_1d': dict(upsample_type='1D', upsample_scales=[2, 2], NN_init=False),
    'mol_nearest': dict(upsample_type='NearestNeighbor', upsample_scales=[4]),
    'softmax': dict(input_type='mulaw-quantize', out_channels=32, quantize_channels=32, layers=6, stacks=3),
    # dropout ON (the benchmarked configuration trains with wavenet_dropout = 0.05): the stand-in records the keep masks it drew, so
    # the placement of the op (conv input only, modules.py:484) and the 1 / (1 - p) scaling are pinned by reference execution
    'mol_2d_dropout': dict(wavenet_dropout=0.05),
    'gauss_subpixel_legacy_dropout': dict(out_channels=2, upsample_type='SubPixel', upsample_scales=[3, 2], legacy=True, residual_legacy=True,
                                          wavenet_dropout=0.3),
    'mol_gin_embed': dict(gin_channels=4, use_speaker_embedding=True),
    'mol_gin_raw_nobias': dict(gin_channels=4, use_speaker_embedding=False, use_bias=False),
    # weight normalisation (modules.py:44-177): gains perturbed away from ||v|| so that the normalisation is visible
    'mol_weightnorm': dict(wavenet_weight_normalization=True, NN_init=False),
    'gauss_weightnorm_1d': dict(wavenet_weight_normalization=True, out_channels=2, upsample_type='1D', upsample_scales=[2, 2], NN_init=False, use_bias=False),
    # widths the HIP engine accepts (multiples of 64 / 16): tests/test_hip_reference_golden.py runs the DEVICE path on these
    'hip_mol_2d': dict(residual_channels=64, gate_channels=128, skip_out_channels=64, cin_channels=16, num_mels=16, out_channels=30,
                       upsample_scales=[4, 4], NN_init=False),
    'hip_gauss_subpixel_legacy': dict(residual_channels=64, gate_channels=128, skip_out_channels=64, cin_channels=16, num_mels=16, out_channels=2,
                                      upsample_type='SubPixel', upsample_scales=[4, 4], legacy=True, residual_legacy=True, NN_init=False),
    'hip_softmax_resize': dict(residual_channels=64, gate_channels=128, skip_out_channels=64, cin_channels=16, num_mels=16,
                               input_type='mulaw-quantize', out_channels=256, quantize_channels=256, upsample_type='Resize',
                               upsample_scales=[2, 8], upsample_activation='LeakyRelu', NN_init=False),
    'hip_mol_weightnorm': dict(residual_channels=64, gate_channels=128, skip_out_channels=64, cin_channels=16, num_mels=16, out_channels=30,
                               upsample_scales=[4, 4], NN_init=False, wavenet_weight_normalization=True),
    'hip_mol_gin_1d': dict(residual_channels=64, gate_channels=128, skip_out_channels=64, cin_channels=16, num_mels=16, out_channels=30,
                           upsample_type='1D', upsample_scales=[4, 4], gin_channels=16, use_speaker_embedding=True, n_speakers=4, NN_init=False),
}


def _import_reference():
    """Import the reference's files with /root/reference first on sys.path and this repo's same-named packages hidden."""
    shim.install()
    for m in list(sys.modules):
        if m.split('.')[0] in ('wavenet_vocoder', 'datasets', 'infolog', 'hparams'):
            del sys.modules[m]
    sys.path.insert(0, REF)
    # numpy >= 1.18 rejects expand_dims(axis > ndim); numpy 1.14 (the reference's era) appended the axis (modules.py:651)
    _ed = np.expand_dims
    np.expand_dims = lambda a, axis: _ed(a, min(axis, np.ndim(a))) if isinstance(axis, int) and axis > np.ndim(a) else _ed(a, axis)
    wn = importlib.import_module('wavenet_vocoder.models.wavenet')
    assert wn.__file__.startswith(REF), wn.__file__
    return wn


def _oracle_name(ref_name):
    """shim variable name ('<layer.name>/kernel') -> oracle / engine tensor name."""
    m = re.match(r'residual_block_(causal|cin|gin|skip|out)_conv_ResidualConv1DGLU_(\d+)/(kernel|bias|g)$', ref_name)
    if m:
        return 'ResidualConv1DGLU_%s/residual_block_%s_conv/%s' % (m.group(2), m.group(1), m.group(3))
    m = re.match(r'(input_convolution|final_convolution_[12])/(kernel|bias|g)$', ref_name)
    if m:
        return ref_name
    m = re.match(r'(ConvTranspose2D|ConvTranspose1D|ResizeConvolution|SubPixelConvolution)_layer_(\d+)/(kernel|bias|g)$', ref_name)
    if m:
        return 'local_conditioning_upsampling_%d/%s' % (int(m.group(2)) + 1, m.group(3))
    if ref_name == 'gc_embedding':
        return ref_name
    raise KeyError(ref_name)


def run_config(wn, name, over):
    shim.reset(seed=1000 + sum(map(ord, name)))
    hp = _hparams(**over)
    gen = torch.Generator().manual_seed(7 + len(name))
    B, Tc = (2, 5) if not name.startswith('hip_') else (2, 6)
    hop = hp.hop_size
    T = Tc * hop
    scalar = hp.input_type != 'mulaw-quantize'
    wav = torch.rand(B, T, generator=gen) * 1.6 - 0.8
    c = torch.rand(B, hp.cin_channels, Tc, generator=gen)
    g = None
    if hp.gin_channels > 0:
        g = (torch.randint(0, hp.n_speakers, (B, 1), generator=gen).int() if hp.use_speaker_embedding
             else torch.randn(B, hp.gin_channels, 1, generator=gen))
    if scalar:
        x = wav.view(B, 1, T)
    else:
        ids = torch.randint(0, hp.quantize_channels, (B, T), generator=gen)
        x = torch.nn.functional.one_hot(ids, hp.quantize_channels).float().permute(0, 2, 1).contiguous()

    # ---- training-mode instance: batch forward (channels_first convolutions)
    model = wn.WaveNet(hp, init=False)
    model.set_mode(True)
    y0 = model.step(x, c=c, g=g, softmax=False)                     # builds every variable
    for k, v in shim.variables().items():                           # reference biases start at zero: give them signal
        if k.endswith('/bias'):
            v.copy_((torch.rand(v.shape, generator=gen) * 2 - 1) * 0.1)
    wn_on = bool(getattr(hp, 'wavenet_weight_normalization', False))
    if wn_on:
        for k, v in shim.variables().items():                       # gains start at ||v|| (kernel == v): move them
            if k.endswith('/g'):
                v.mul_(torch.rand(v.shape, generator=gen) * 0.8 + 0.6)
        # in a TF graph `kernel = l2_normalize(v) * g` is re-evaluated on every run; the eager stand-in built it once
        def _wn_layers(obj, seen):
            if id(obj) in seen or not hasattr(obj, '__dict__'):
                return
            seen.add(id(obj))
            if type(obj).__name__ == 'WeightNorm':
                obj._compute_weights()
            for vv in list(vars(obj).values()):
                for o in (vv if isinstance(vv, (list, tuple)) else [vv]):
                    _wn_layers(o, seen)
        _wn_layers(model, set())
    shim._STATE.dropout_masks.clear()
    y_hat = model.step(x, c=c, g=g, softmax=False)
    drop_masks = [m.clone() for m in shim._STATE.dropout_masks]       # one [B, R, T] keep mask per residual layer, in layer order
    assert len(drop_masks) == (hp.layers if hp.wavenet_dropout > 0 else 0)
    c_up = model.upsampled_local_features
    assert y_hat.shape == (B, hp.out_channels, T) and not torch.equal(y0, y_hat)

    # ---- the masked training loss exactly as WaveNet.initialize/add_loss wire it (wavenet.py:272-283, 476-495, 632-638)
    lengths = [T, T - 7]
    model.is_training = True; model.is_evaluating = False
    hp.wavenet_num_gpus = 1
    model.tower_mask = [model.get_mask(torch.tensor(lengths, dtype=torch.int32), maxlen=T)]
    model.tower_y_hat_train = [y_hat]
    if scalar:
        model.tower_y = [wav.view(B, T, 1)]
    else:
        model.tower_y_hat_q = [y_hat.permute(0, 2, 1)]
        model.tower_y = [ids]
    model.add_loss()
    loss = torch.as_tensor(model.loss).reshape(1)

    # ---- synthesis-mode instance sharing the variables (channels_last convolutions, linearised weights, queues)
    model_s = wn.WaveNet(hp, init=False)
    model_s.set_mode(False)
    if scalar:
        init = torch.zeros(B, 1, 1)
        ti = wav.view(B, 1, T)                                      # [B, 1, T]: transposed inside (wavenet.py:753-755)
    else:
        init = torch.nn.functional.one_hot(torch.full((B, 1), hp.quantize_channels // 2), hp.quantize_channels).float()   # [B,1,Q]
        ti = ids.view(B, T, 1).int()
    kw = dict(c=c, g=g, time_length=T, softmax=False, quantize=True, log_scale_min=hp.log_scale_min, log_scale_min_gauss=hp.log_scale_min_gauss)
    res = {}
    shim._STATE.uniform_draws.clear(); shim._STATE.normal_draws.clear()
    if scalar and not wn_on:      # (with weight normalisation the reference's incremental path multiplies the RAW v: SURVEY C-7)
        out_tf = model_s.incremental(init, test_inputs=ti, **kw)
        res['inc_tf_raw'] = model_s.tower_y_hat_eval[0]              # [B, O, T]
        res['inc_tf_out'] = out_tf
        u_tf = [u.clone() for u in shim._STATE.uniform_draws]; n_tf = [e.clone() for e in shim._STATE.normal_draws]
        shim._STATE.uniform_draws.clear(); shim._STATE.normal_draws.clear()
        out_fr = model_s.incremental(init, test_inputs=None, **kw)
        res['inc_free_raw'] = model_s.tower_y_hat_eval[0]
        res['inc_free_out'] = out_fr
        u_fr = list(shim._STATE.uniform_draws); n_fr = list(shim._STATE.normal_draws)
        if hp.out_channels == 2:
            res['eps_tf'] = torch.stack([e.reshape(B) for e in n_tf]); res['eps_free'] = torch.stack([e.reshape(B) for e in n_fr])
        else:
            M = hp.out_channels // 3       # mixture.py:91 draws [B, T=1, M] then :104 [B, T=1] per step
            res['u1_tf'] = torch.stack([u.reshape(B, M) for u in u_tf[0::2]]); res['u2_tf'] = torch.stack([u.reshape(B) for u in u_tf[1::2]])
            res['u1_free'] = torch.stack([u.reshape(B, M) for u in u_fr[0::2]]); res['u2_free'] = torch.stack([u.reshape(B) for u in u_fr[1::2]])
    out = {'x': x, 'c': c, 'y_hat': y_hat, 'c_up': c_up, 'wav': wav, 'loss': loss.float(), 'lengths': torch.tensor(lengths)}
    if not scalar:
        out['ids'] = ids
    if g is not None:
        out['g'] = g
    if drop_masks:
        out['dropout_masks'] = torch.stack(drop_masks)                # [L, B, R, T]
    out.update(res)
    arrays = {k: v.detach().numpy() for k, v in out.items()}
    for k, v in shim.variables().items():
        arrays['params/' + _oracle_name(k)] = v.detach().numpy().copy()
    # SubPixelConvolution.build replaces its kernel attribute by W_0 tiled over the sub-pixel filters when NN_init is off
    # (modules.py:584-592): the EFFECTIVE kernel is what the layer multiplies with
    if not wn_on:
        ups = [l for l in getattr(model, 'upsample_conv', []) if hasattr(l, 'kernel') and l.kernel is not None]
        for i, l in enumerate(ups):
            arrays['params/local_conditioning_upsampling_%d/kernel' % (i + 1)] = l.kernel.detach().numpy().copy()
    arrays['hparams_keys'] = np.array(sorted(over.keys()))
    arrays['hparams_json'] = np.array(__import__('json').dumps(over))
    np.savez_compressed(os.path.join(OUT, 'stack_%s.npz' % name), **arrays)
    return {k: tuple(v.shape) for k, v in arrays.items() if not k.startswith('params/')}


def main():
    if not os.path.isdir(os.path.join(REF, 'wavenet_vocoder')):
        raise SystemExit('needs /root/reference (run in the build container)')
    wn = _import_reference()
    os.makedirs(OUT, exist_ok=True)
    for name, over in CONFIGS.items():
        info = run_config(wn, name, over)
        print('stack_%s.npz:' % name, info.get('y_hat'), 'inc' if 'inc_tf_raw' in info else '')


if __name__ == '__main__':
    main()
