"""Pin the oracle against golden vectors produced by the REFERENCE's own source files
(oracle/gen_golden.py executes util.py / mixture.py / gaussian.py from /root/reference)."""
import os

import numpy as np
import torch

from oracle import mulaw as M
from oracle import wavenet_oracle as O


def test_mulaw_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, 'mulaw_golden.npz'))
    x = g['x']
    assert np.array_equal(M.mulaw_quantize(x), g['quantized'])            # integer indices: bit-exact
    assert np.array_equal(M.mulaw_quantize(g['x64']), g['quantized64'])
    assert np.array_equal(M.mulaw(x), g['mulaw'])
    assert np.array_equal(M.inv_mulaw_quantize(np.arange(256)), g['inv_q_all'])
    assert np.array_equal(M.inv_mulaw(np.linspace(-1, 1, 4097).astype(np.float32)), g['inv_mulaw'])
    assert int(g['q0']) == 127 and int(M.mulaw_quantize(np.float64(0.0))) == 127   # wavenet.py:361,434
    assert float(g['m0']) == 0.0


def test_mulaw_roundtrip_properties():
    x = np.linspace(-1, 1, 100001)
    q = M.mulaw_quantize(x)
    assert q.min() == 0 and q.max() == 255
    assert np.all(np.diff(q) >= 0)                                         # monotone
    xr = M.inv_mulaw_quantize(q)
    # decode(encode(x)) stays inside the bin that contains x
    lo = M.inv_mulaw_quantize(np.maximum(q - 1, 0)); hi = M.inv_mulaw_quantize(np.minimum(q + 1, 255))
    assert np.all(xr >= lo - 1e-7) and np.all(xr <= hi + 1e-7)
    # decode lands on the bin's lower edge (truncating quantiser), so re-encoding is within one bin
    assert np.all(np.abs(M.mulaw_quantize(M.inv_mulaw_quantize(np.arange(256)).astype(np.float64)) - np.arange(256)) <= 1)


def test_mol_loss_and_sampler(golden_dir):
    g = np.load(os.path.join(golden_dir, 'mol_golden.npz'))
    y_hat, y = torch.from_numpy(g['y_hat']), torch.from_numpy(g['y'])
    for nc, lsm in ((65536, float(np.log(1e-14))), (256, -7.0)):
        l = O.discretized_mix_logistic_loss(y_hat, y, num_classes=nc, log_scale_min=lsm)
        np.testing.assert_allclose(l.numpy(), g['loss_%d' % nc], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(float(l.sum()), float(g['loss_sum_%d' % nc]), rtol=1e-6)
    s = O.sample_from_discretized_mix_logistic(y_hat, torch.from_numpy(g['u1']), torch.from_numpy(g['u2']),
                                               log_scale_min=float(np.log(1e-14)))
    np.testing.assert_allclose(s.numpy(), g['sample'], rtol=0, atol=1e-7)


def test_gaussian_loss_and_sampler(golden_dir):
    g = np.load(os.path.join(golden_dir, 'gaussian_golden.npz'))
    y_hat, y = torch.from_numpy(g['y_hat']), torch.from_numpy(g['y'])
    for tag, lsm in (('a', float(np.log(1e-7))), ('b', float(np.log(9.1188196e-4)))):
        for cdf in (False, True):
            l = O.gaussian_mle_loss(y_hat, y, lsm, 65536, cdf)
            np.testing.assert_allclose(l.numpy(), g['loss_cdf%d_%s' % (int(cdf), tag)], rtol=1e-6, atol=1e-6)
    s = O.sample_from_gaussian(y_hat, torch.from_numpy(g['eps']), float(np.log(1e-7)))
    np.testing.assert_allclose(s.numpy(), g['sample'], rtol=0, atol=1e-7)


# ---------------------------------------------------------------------------------------------------------------------
# Conv stack / upsample net / incremental loop: golden vectors produced by executing the reference's own wavenet.py and
# modules.py on the eager TF-1 stand-in (oracle/tf1_shim.py, oracle/gen_golden_stack.py).  The oracle must reproduce the
# reference's COMPOSITION of ops to fp32 round-off.
import glob
import json

import pytest

_STACK_DEFAULTS = dict(layers=4, stacks=2, residual_channels=16, gate_channels=32, skip_out_channels=16, out_channels=6, kernel_size=3,
                       cin_channels=8, gin_channels=-1, use_speaker_embedding=True, n_speakers=3, input_type='raw', quantize_channels=65536,
                       use_bias=True, legacy=False, residual_legacy=False, wavenet_dropout=0.0, upsample_type='2D', upsample_scales=[2, 3],
                       upsample_activation='Relu', leaky_alpha=0.4, freq_axis_kernel_size=3, NN_init=True, NN_scaler=0.3,
                       log_scale_min=float(np.log(1e-14)), log_scale_min_gauss=float(np.log(1e-7)), cdf_loss=False, wavenet_weight_normalization=False)
_STACK_FILES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stack_*.npz')))


def _load_stack(path):
    g = np.load(path)
    kw = dict(_STACK_DEFAULTS); kw.update(json.loads(str(g['hparams_json'])))
    cfg = O.OracleConfig(**{k: v for k, v in kw.items() if k in O.OracleConfig.__dataclass_fields__})
    params = {k[len('params/'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('params/')}
    return g, cfg, params


def test_reference_executed_goldens_exist():
    assert len(_STACK_FILES) >= 20 and any('dropout' in p for p in _STACK_FILES), 'run oracle/gen_golden_stack.py in the container that has /root/reference'


@pytest.mark.parametrize('path', _STACK_FILES, ids=[os.path.basename(p)[6:-4] for p in _STACK_FILES])
def test_stack_matches_reference_execution(path):
    g, cfg, params = _load_stack(path)
    # the oracle's parameter table has exactly the variables the reference model created (names and TF shapes)
    sh = O.param_shapes(cfg)
    assert set(sh) == set(params), (sorted(set(sh) ^ set(params)))
    for k in sh:
        assert tuple(params[k].shape) == tuple(sh[k]), (k, params[k].shape, sh[k])
    x, c = torch.from_numpy(g['x']), torch.from_numpy(g['c'])
    gg = None
    if 'g' in g.files:
        gg = torch.from_numpy(g['g'])
        gg = gg.reshape(-1) if cfg.use_speaker_embedding else gg.reshape(gg.shape[0], -1)
    c_up = O.upsample(params, cfg, c)
    np.testing.assert_allclose(c_up.numpy(), g['c_up'], rtol=1e-5, atol=1e-6)                 # wavenet.py:680-702
    masks = None
    if 'dropout_masks' in g.files:      # the keep masks the reference run drew (tf.layers.dropout executed on the stand-in, modules.py:484)
        assert cfg.wavenet_dropout > 0
        dm = torch.from_numpy(g['dropout_masks'])
        masks = [dm[l] for l in range(dm.shape[0])]
        assert len(masks) == cfg.layers and 0.0 < float(1.0 - dm.mean()) < 2.5 * cfg.wavenet_dropout + 0.05
        y_off = O.step(params, cfg, x, c, g=gg)
        assert not np.allclose(y_off.numpy(), g['y_hat'], rtol=1e-3, atol=1e-3)               # the masks matter
    y = O.step(params, cfg, x, c, g=gg, dropout_masks=masks)
    np.testing.assert_allclose(y.numpy(), g['y_hat'], rtol=2e-5, atol=2e-6)                   # wavenet.py:650-721
    # masked training loss as WaveNet.add_loss wires it (wavenet.py:476-495, 632-638; modules.py:781-836), ragged lengths
    y_t = torch.from_numpy(g['ids']).long() if 'ids' in g.files else torch.from_numpy(g['wav']).unsqueeze(-1)
    loss = O.training_loss(cfg, torch.from_numpy(g['y_hat']), y_t, [int(v) for v in g['lengths']])
    np.testing.assert_allclose(float(loss), float(g['loss'][0]), rtol=2e-6)
    if 'inc_tf_raw' not in g.files:
        return
    B, T = g['wav'].shape
    wav = torch.from_numpy(g['wav'])
    def noise(tag):
        if cfg.out_channels == 2:
            return {'eps': torch.from_numpy(g['eps_' + tag])}
        return {'u1': torch.from_numpy(g['u1_' + tag]), 'u2': torch.from_numpy(g['u2_' + tag])}
    for form in ('reference', 'ring'):
        # teacher-forced (wavenet.py:752-768, 877-878): raw outputs and the samples drawn from them
        out, raw = O.incremental(params, cfg, c, noise=noise('tf'), test_inputs=wav.unsqueeze(-1), formulation=form, g=gg)
        np.testing.assert_allclose(raw.numpy(), g['inc_tf_raw'], rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(out.numpy(), g['inc_tf_out'], rtol=0, atol=2e-5)
        # free-running: the sample feeds back (wavenet.py:853-880), so this also pins the recurrence
        out, raw = O.incremental(params, cfg, c, noise=noise('free'), test_inputs=None, formulation=form, g=gg)
        np.testing.assert_allclose(out.numpy(), g['inc_free_out'], rtol=0, atol=5e-5)
        np.testing.assert_allclose(raw.numpy(), g['inc_free_raw'], rtol=1e-4, atol=1e-4)


def test_shim_convolutions_against_naive_loops():
    """The TF stand-in's convolution primitives (the part of the golden pipeline that is ours, not the reference's) against
    direct index arithmetic written from the TensorFlow documentation."""
    from oracle import tf1_shim as S
    gen = torch.Generator().manual_seed(0)
    # Conv1D valid, dilation 2, channels_first: y[o,t] = b[o] + sum_{j,i} K[j,i,o] x[i, t + j*d]
    L = S.Conv1D(3, 3, dilation_rate=2, data_format='channels_first', name='t_c1'); S.reset()
    x = torch.randn(1, 2, 9, generator=gen); y = L(x); K, b = L.kernel, L.bias
    ref = torch.zeros(1, 3, 5)
    for o in range(3):
        for t in range(5):
            ref[0, o, t] = b[o] + sum(K[j, i, o] * x[0, i, t + 2 * j] for j in range(3) for i in range(2))
    assert torch.allclose(y, ref, atol=1e-5)
    # Conv2DTranspose SAME, stride (1,s), kernel (3,s), NCHW, kernel layout [kh,kw,out,in]: out[f', t*s+j] = sum_kf x[f'-kf+1, t] K[kf,j]
    s = 3
    L = S.Conv2DTranspose(1, (3, s), strides=(1, s), padding='same', data_format='channels_first', name='t_ct'); S.reset()
    x = torch.randn(1, 1, 4, 5, generator=gen); y = L(x); K = L.kernel[:, :, 0, 0]
    assert y.shape == (1, 1, 4, 5 * s)
    ref = torch.zeros(4, 5 * s)
    for f in range(4):
        for t in range(5):
            for j in range(s):
                ref[f, t * s + j] = sum(x[0, 0, f - kf + 1, t] * K[kf, j] for kf in range(3) if 0 <= f - kf + 1 < 4)
    assert torch.allclose(y[0, 0], ref, atol=1e-5)
    # Conv2D SAME with an EVEN kernel width (Resize upsampler): TF pads (k-1)//2 before, the rest after
    L = S.Conv2D(1, (3, 4), padding='same', data_format='channels_last', name='t_c2'); S.reset()
    x = torch.randn(1, 3, 6, 1, generator=gen); y = L(x); K = L.kernel[:, :, 0, 0]
    ref = torch.zeros(3, 6)
    for f in range(3):
        for t in range(6):
            ref[f, t] = sum(x[0, f + kf - 1, t + kt - 1, 0] * K[kf, kt] for kf in range(3) for kt in range(4)
                            if 0 <= f + kf - 1 < 3 and 0 <= t + kt - 1 < 6)
    assert torch.allclose(y[0, :, :, 0], ref, atol=1e-5)
    # batch_to_space_nd: out[b', i*r + j] = in[j*n + b', i]
    x = torch.arange(2 * 3 * 4).float().reshape(6, 4); y = S.batch_to_space_nd(x, [3], [[0, 0]])
    assert y.shape == (2, 12) and all(float(y[bp, i * 3 + j]) == float(x[j * 2 + bp, i]) for bp in range(2) for i in range(4) for j in range(3))


def test_reference_weightnorm_data_dependent_init_is_recorded(golden_dir):
    """the reference's data-dependent weight-norm initialisation (modules.py:110-126 driven by train.py:287-298),
    executed with the reference's own files on the TF stand-in by oracle/check_weightnorm_init.py.  Finding pinned by the fixture:
    the init forward pass raises a shape error in the first wrapped layer (moments over the kernel's leading axes of a channels-first
    activation have shape [T], g has [filters]) and no variable changes -- there is no initialisation arithmetic to reproduce; a
    weight-normalised model starts from g = ||v|| (WeightNorm.build, modules.py:164-170), which is what this tree does."""
    g = np.load(os.path.join(golden_dir, 'weightnorm_datadep_init.npz'))
    assert bool(g['raised']) and 'must match' in str(g['error'])
    assert int(g['time_steps']) != int(g['first_layer_filters'])
    assert float(g['max_abs_change']) == 0.0 and int(g['n_variables']) > 20 and int(g['n_wrappers']) > 10
    assert float(g['y_train_model_after_vs_before']) == 0.0
    # our initialiser for weight-normalised models: g = ||v|| over all axes but the last, kernel == v at step 0
    from oracle import wavenet_oracle as O
    cfg = O.OracleConfig(layers=2, stacks=1, residual_channels=8, gate_channels=16, skip_out_channels=8, cin_channels=4,
                         upsample_type='2D', upsample_scales=[2, 2], wavenet_weight_normalization=True)
    p = O.init_params(cfg, seed=1)
    eff = O.effective_params(p, cfg)
    for k, v in p.items():
        if k.endswith('/kernel'):
            assert torch.allclose(eff[k], v, atol=1e-6)


def test_train_step_reproduces_reference_add_optimizer(golden_dir):
    """three training steps of the reference's own WaveNet.add_loss + add_optimizer (wavenet.py:476-613), executed on
    the TF stand-in by oracle/gen_golden_optim.py with clip thresholds that bite, vs oracle.train_step -- the function the device
    optimiser (wn_norm2_kernel + wn_adam_kernel) is compared with in tests/test_hip_parity.py.  Pins the composition: the gradient
    set, per-variable clip_by_norm THEN clip_by_value, Adam on the clipped gradients at the scheduled rate, EMA after the update."""
    import json
    g = np.load(os.path.join(golden_dir, 'optim_golden.npz'))
    hpj = json.loads(str(g['hparams_json']))
    cfg = O.OracleConfig(layers=4, stacks=2, residual_channels=16, gate_channels=32, skip_out_channels=16, out_channels=6, cin_channels=8,
                         upsample_type='2D', upsample_scales=[2, 3], NN_init=False, NN_scaler=0.3, wavenet_dropout=0.0)
    names = [k[3:] for k in g.files if k.startswith('p0/')]
    assert set(names) == set(O.param_shapes(cfg))
    params = {k: torch.from_numpy(g['p0/' + k]) for k in O.param_shapes(cfg)}
    state = O.init_opt_state(params)
    x, c, wav, lengths = torch.from_numpy(g['x']), torch.from_numpy(g['c']), torch.from_numpy(g['wav']), [int(v) for v in g['lengths']]
    B, T = wav.shape
    lrk = dict(init_lr=hpj['wavenet_learning_rate'], schedule='exponential', decay_rate=hpj['wavenet_decay_rate'], decay_steps=hpj['wavenet_decay_steps'])
    adk = dict(beta1=hpj['wavenet_adam_beta1'], beta2=hpj['wavenet_adam_beta2'], eps=hpj['wavenet_adam_epsilon'], ema_decay=hpj['wavenet_ema_decay'])
    n_clipped = 0
    for s in range(3):
        assert abs(float(O.learning_rate(s, **lrk)) - float(g['lr/%d' % s])) <= 1e-9
        loss, grads, params, state = O.train_step(params, state, cfg, x, c, wav.view(B, T, 1), lengths, s, lr_kwargs=lrk,
                                                  max_norm=hpj['wavenet_gradient_max_norm'], max_value=hpj['wavenet_gradient_max_value'], adam_kwargs=adk)
        assert abs(float(loss) - float(g['loss/%d' % s])) <= 2e-6 * abs(float(g['loss/%d' % s]))
        for k in names:
            ref_g = torch.from_numpy(g['g%d/%s' % (s, k)])
            assert torch.allclose(grads[k], ref_g, rtol=2e-4, atol=2e-7), (s, k)             # same gradient set (before clipping)
            n_clipped += int(float(ref_g.norm()) > hpj['wavenet_gradient_max_norm']) + int(float(ref_g.abs().max()) > hpj['wavenet_gradient_max_value'])
            assert torch.allclose(params[k], torch.from_numpy(g['p%d/%s' % (s + 1, k)]), rtol=1e-5, atol=2e-7), (s, k)
            assert torch.allclose(state[k][2], torch.from_numpy(g['ema%d/%s' % (s + 1, k)]), rtol=1e-5, atol=2e-7), (s, k)
    assert n_clipped > 60                                                                  # both clips were active in the fixture
