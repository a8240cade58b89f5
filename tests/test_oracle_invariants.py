"""Internal invariants that pin the conv-stack part of the oracle: the reference
guarantees them by construction (zero queues == zero left padding; NN-init == repeat)."""
import math

import numpy as np
import pytest
import scipy.special
import scipy.stats
import torch

from oracle import wavenet_oracle as O


def small_cfg(**kw):
    base = dict(layers=4, stacks=2, residual_channels=16, gate_channels=32, skip_out_channels=16,
                out_channels=30, cin_channels=8, upsample_type='2D', upsample_scales=[4, 4], NN_scaler=0.3)
    base.update(kw)
    return O.OracleConfig(**base)


def test_receptive_field():
    assert O.receptive_field_size(24, 4, 3) == 505          # paper_hparams file shape
    assert O.receptive_field_size(24, 2, 3) == 16381        # BASELINE 2-stack shape
    assert O.receptive_field_size(20, 2, 3) == 4093
    assert O.receptive_field_size(8, 1, 3) == 511
    assert O.receptive_field_size(30, 3, 3) == 6139


def test_param_count_paper_shape():
    cfg = O.OracleConfig(layers=24, stacks=2)
    n = sum(int(np.prod(s)) for k, s in O.param_shapes(cfg).items() if not k.startswith('local'))
    assert n == 13676830


@pytest.mark.parametrize('ut,scales', [('2D', [4, 4]), ('SubPixel', [4, 4]), ('Resize', [3, 5]),
                                       ('1D', [4, 4]), ('NearestNeighbor', [4, 4])])
@pytest.mark.parametrize('it,oc,qc', [('raw', 30, 65536), ('raw', 2, 65536), ('mulaw-quantize', 256, 256)])
def test_batch_equals_incremental(ut, scales, it, oc, qc):
    torch.manual_seed(0)
    cfg = small_cfg(upsample_type=ut, upsample_scales=scales, input_type=it, out_channels=oc,
                    quantize_channels=qc, legacy=(ut == '2D'), residual_legacy=(ut == 'SubPixel'))
    P = O.init_params(cfg, bias_scale=0.1)
    B, Tc = 2, 3
    T = Tc * cfg.hop
    c = torch.rand(B, 8, Tc)
    if it == 'raw':
        y = torch.rand(B, T, 1) * 1.8 - 0.9
        x_shift = torch.cat([torch.zeros(B, 1, 1), y[:, :-1]], 1).transpose(1, 2)
        ti = y
    else:
        oh = torch.nn.functional.one_hot(torch.randint(0, 256, (B, T)), 256).float()
        x_shift = torch.cat([O.initial_input(cfg, B).unsqueeze(1), oh[:, :-1]], 1).transpose(1, 2)
        ti = oh
    yb = O.step(P, cfg, x_shift, c)
    noise = {'u1': torch.rand(T, B, 10) * .9 + .05, 'u2': torch.rand(T, B) * .9 + .05,
             'eps': torch.randn(T, B), 'gumbel_u': torch.rand(T, B, 256) * .9 + .05}
    o1, r1 = O.incremental(P, cfg, c, noise=noise, test_inputs=ti, formulation='reference')
    o2, r2 = O.incremental(P, cfg, c, noise=noise, test_inputs=ti, formulation='ring')
    assert torch.allclose(yb, r1, atol=1e-5)
    assert torch.equal(r1, r2) and torch.equal(o1, o2)


@pytest.mark.parametrize('ut,scales', [('2D', [5, 5, 11]), ('SubPixel', [11, 25]), ('1D', [4, 4]), ('Resize', [3, 5])])
def test_nn_init_upsample_is_scaled_repeat(ut, scales):
    cfg = small_cfg(upsample_type=ut, upsample_scales=scales)
    P = O.init_params(cfg)
    c = torch.rand(2, 8, 5)
    cu = O.upsample(P, cfg, c)
    assert torch.allclose(cu, torch.repeat_interleave(c, cfg.hop, dim=2) * 0.3, atol=1e-6)


def test_causality():
    cfg = small_cfg()
    P = O.init_params(cfg, bias_scale=0.1)
    T = 48
    x = torch.rand(1, 1, T)
    c = torch.rand(1, 8, 3)
    y0 = O.step(P, cfg, x, c)
    x2 = x.clone(); x2[0, 0, 20] += 1.0
    y1 = O.step(P, cfg, x2, c)
    assert torch.equal(y0[:, :, :20], y1[:, :, :20])
    assert not torch.equal(y0[:, :, 20], y1[:, :, 20])


def test_mol_one_component_closed_form():
    # M=1: loss = -log( sigmoid((y+D-mu)/s) - sigmoid((y-D-mu)/s) ) for interior y
    y = torch.tensor([[[0.1], [-0.3]]])
    mu, ls = 0.05, -3.0
    y_hat = torch.tensor([[[0.0, 0.0], [mu, mu], [ls, ls]]])
    l = O.discretized_mix_logistic_loss(y_hat, y, num_classes=256, log_scale_min=-7.0)
    D = 1 / 255.
    s = math.exp(ls)
    exp = [-math.log(scipy.special.expit((v + D - mu) / s) - scipy.special.expit((v - D - mu) / s)) for v in (0.1, -0.3)]
    np.testing.assert_allclose(l.flatten().numpy(), exp, rtol=1e-5)


def test_gaussian_pdf_closed_form():
    y = torch.tensor([[[0.2], [-0.5]]])
    y_hat = torch.tensor([[[0.1, 0.0], [-1.0, -2.0]]])
    l = O.gaussian_mle_loss(y_hat, y, -16.0, 65536, False)
    exp = [-scipy.stats.norm.logpdf(0.2, 0.1, math.exp(-1.0)), -scipy.stats.norm.logpdf(-0.5, 0.0, math.exp(-2.0))]
    np.testing.assert_allclose(l.flatten().numpy(), exp, rtol=1e-5)


def test_softmax_ce_denominator_is_count_nonzero():
    cfg = small_cfg(input_type='mulaw-quantize', out_channels=256, quantize_channels=256)
    y_hat = torch.randn(2, 256, 9)
    y = torch.randint(0, 256, (2, 9))
    l = O.training_loss(cfg, y_hat, y, [9, 5])
    logits = y_hat.transpose(1, 2)[:, :-1]
    ce = scipy.special.logsumexp(logits.numpy(), axis=-1) - np.take_along_axis(logits.numpy(), y[:, 1:, None].numpy(), -1)[..., 0]
    mask = np.array([[1] * 8, [1] * 4 + [0] * 4], np.float32)
    np.testing.assert_allclose(float(l), (ce * mask).sum() / 12, rtol=1e-5)


def test_autograd_vs_finite_differences():
    torch.manual_seed(1)
    cfg = small_cfg(layers=2, stacks=1, residual_channels=8, gate_channels=16, skip_out_channels=8,
                    log_scale_min=-7.0, quantize_channels=256)
    P = {k: v.double() for k, v in O.init_params(cfg, bias_scale=0.1).items()}
    T = 16
    y = (torch.rand(1, T, 1) * 1.6 - 0.8).double()
    x = y.transpose(1, 2)
    c = torch.rand(1, 8, 1).double()

    name = 'ResidualConv1DGLU_1/residual_block_causal_conv/kernel'

    def f(w):
        Q = dict(P); Q[name] = w
        return O.training_loss(cfg, O.step(Q, cfg, x, c), y, [T])
    w = P[name].clone().requires_grad_(True)
    assert torch.autograd.gradcheck(f, (w,), eps=1e-6, atol=1e-5, rtol=1e-3)


def test_tf_adam_and_clip_semantics():
    g = torch.tensor([300.0, -400.0])                       # ||g|| = 500 -> scaled to norm 100 -> (60,-80) -> clipped to +-5
    assert torch.allclose(O.clip_gradient(g), torch.tensor([5.0, -5.0]))
    g = torch.tensor([3.0, -4.0])
    assert torch.allclose(O.clip_gradient(g), g)
    p, m, v, e = O.adam_ema_update(torch.tensor([1.0]), torch.tensor([0.5]), torch.zeros(1), torch.zeros(1),
                                   torch.tensor([1.0]), step=1, lr=1e-3)
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    exp_p = 1.0 - lr_t * 0.05 / (math.sqrt(0.001 * 0.25) + 1e-6)
    assert abs(float(p) - exp_p) < 1e-7
    assert abs(float(e) - (1.0 - 1e-4 * (1.0 - exp_p))) < 1e-7
    assert abs(O.learning_rate(200000) - 5e-4) < 1e-12
    assert O.learning_rate(0, schedule='noam') == max(1e-3 * 4000 ** 0.5 * 4000 ** -1.5, 1e-4)


def test_global_conditioning_and_bias_free_layers_batch_equals_incremental():
    """The design invariant of the reference (batch step == teacher-forced incremental) must also hold with the
    hparam-gated options: global conditioning (embedding table or raw features) and use_bias=False."""
    for kw in (dict(gin_channels=8, use_speaker_embedding=True, n_speakers=3), dict(gin_channels=4, use_speaker_embedding=False, use_bias=False)):
        cfg = O.OracleConfig(layers=4, stacks=2, residual_channels=16, gate_channels=32, skip_out_channels=16, out_channels=6,
                             cin_channels=8, upsample_scales=[2, 2], wavenet_dropout=0.0, **kw)
        P = O.init_params(cfg, seed=3, bias_scale=0.1)
        assert ('gc_embedding' in P) == bool(kw.get('use_speaker_embedding'))
        assert any('gin_conv/kernel' in k for k in P)
        assert kw.get('use_bias', True) == any(k.endswith('residual_block_out_conv/bias') for k in P)
        B, Tc = 2, 5
        T = Tc * cfg.hop
        gen = torch.Generator().manual_seed(0)
        wav = torch.rand(B, T, generator=gen) * 1.6 - 0.8
        c = torch.rand(B, cfg.cin_channels, Tc, generator=gen)
        g = torch.tensor([2, 0]) if cfg.use_speaker_embedding else torch.randn(B, cfg.gin_channels, generator=gen)
        x_shift = torch.cat([torch.zeros(B, 1), wav[:, :-1]], 1).view(B, 1, T)
        yb = O.step(P, cfg, x_shift, c, g=g)
        noise = {'u1': torch.rand(T, B, 2, generator=gen) * 0.9 + 0.05, 'u2': torch.rand(T, B, generator=gen) * 0.9 + 0.05}
        _, raw = O.incremental(P, cfg, c, noise=noise, test_inputs=wav.unsqueeze(-1), g=g)
        assert torch.allclose(raw, yb, atol=2e-5), float((raw - yb).abs().max())
        # g matters: another speaker / feature vector changes the output
        g2 = torch.tensor([1, 1]) if cfg.use_speaker_embedding else g + 1.0
        assert float((O.step(P, cfg, x_shift, c, g=g2) - yb).abs().max()) > 1e-4
