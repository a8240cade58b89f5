"""Fast-WaveNet synthesis on the HIP path vs the oracle's incremental loop (reference formulation,
wavenet.py:724-911) -- teacher-forced and free-running, eager launches vs hipGraph replay."""
import numpy as np
import pytest
import torch

from hip_util import SMALL, make_hp, oracle_cfg, rel_err, synth_batch, upload_params
from oracle import mulaw as M
from oracle import wavenet_oracle as O

pytestmark = pytest.mark.gpu


def _setup(B, Tc, **kw):
    from wavenet_vocoder import _ext
    k = dict(SMALL); k.update(kw)
    hp = make_hp(**k)
    cfg = oracle_cfg(hp)
    T = Tc * cfg.hop
    eng = _ext.Engine(hp, B, T)
    params = O.init_params(cfg, seed=11, bias_scale=0.05)
    eng.pack_weights(upload_params(eng, params))
    wav, c = synth_batch(cfg, B, T, seed=3)
    return hp, cfg, eng, params, wav, c, T


def _noise(cfg, T, B, seed=0):
    g = torch.Generator().manual_seed(seed)
    if cfg.input_type == 'mulaw-quantize':
        u = torch.rand(T, B, cfg.quantize_channels, generator=g) * 0.98 + 0.01
        return u, {'gumbel_u': u}
    if cfg.out_channels == 2:
        e = torch.randn(T, B, generator=g)
        return e.unsqueeze(-1).contiguous(), {'eps': e}
    Mx = cfg.out_channels // 3
    u1 = torch.rand(T, B, Mx, generator=g) * 0.98 + 0.01
    u2 = torch.rand(T, B, generator=g) * 0.98 + 0.01
    return torch.cat([u1, u2.unsqueeze(-1)], -1).contiguous(), {'u1': u1, 'u2': u2}


@pytest.mark.parametrize('kw', [dict(), dict(out_channels=2, legacy=True, residual_legacy=True, upsample_type='SubPixel'),
                                dict(input_type='mulaw-quantize', out_channels=256, quantize_channels=256),
                                dict(gin_channels=16, use_speaker_embedding=True, n_speakers=4),          # per-stream gate bias
                                dict(gin_channels=8, use_speaker_embedding=False, use_bias=False, upsample_type='1D')])
def test_teacher_forced_matches_oracle(kw):
    B, Tc = 3, 6
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, **kw)
    nz_dev, nz_or = _noise(cfg, T, B)
    g = None
    if cfg.gin_channels > 0:
        gg = torch.Generator().manual_seed(5)
        g = (torch.randint(0, cfg.n_speakers, (B,), generator=gg).int() if cfg.use_speaker_embedding
             else torch.randn(B, cfg.gin_channels, generator=gg))
        eng.set_global_condition(g.cuda())
    if cfg.input_type == 'mulaw-quantize':
        ids = torch.from_numpy(M.mulaw_quantize(wav.numpy())).int()
        ti_dev = ids.cuda(); ti_or = torch.nn.functional.one_hot(ids.long(), 256).float()
        out = torch.empty(B, T, dtype=torch.int32, device='cuda')
    else:
        ti_dev = wav.contiguous().cuda(); ti_or = wav.unsqueeze(-1)
        out = torch.empty(B, T, device='cuda')
    raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, ti_dev, steps_per_graph=1)
    torch.cuda.synchronize()
    o_or, r_or = O.incremental(params, cfg, c, noise=nz_or, test_inputs=ti_or, formulation='reference', g=g)
    e = rel_err(raw.cpu(), r_or)
    print('\nteacher-forced raw rel err %.3e' % e)
    assert e < 1.4e-2                                  # measured 3.3 - 4.4e-3 (profiles/r2f_pytest_gpu_all_verbose.log)
    # sampler on the device's own raw outputs == device samples
    if cfg.input_type == 'mulaw-quantize':
        exp = torch.stack([O.sample_categorical(raw.cpu()[:, :, t], nz_or['gumbel_u'][t]) for t in range(T)], 1)
        assert torch.equal(out.cpu().long(), exp)                                   # bit-exact class ids
    elif cfg.out_channels == 2:
        exp = O.sample_from_gaussian(raw.cpu(), nz_or['eps'].t(), cfg.log_scale_min_gauss)
        assert torch.allclose(out.cpu(), exp, atol=2e-5)
    else:
        exp = O.sample_from_discretized_mix_logistic(raw.cpu(), nz_or['u1'].permute(1, 0, 2), nz_or['u2'].t(), cfg.log_scale_min)
        assert torch.allclose(out.cpu(), exp, atol=2e-5)
    # hipGraph replay == eager launches, bit for bit
    out2 = torch.empty_like(out); raw2 = torch.empty_like(raw)
    eng.synthesize(c.cuda(), nz_dev.cuda(), out2, raw2, ti_dev, steps_per_graph=8)
    torch.cuda.synchronize()
    assert torch.equal(raw, raw2) and torch.equal(out, out2)


def test_free_running_feedback_path():
    # free-running: the oracle teacher-forced with the DEVICE's samples must reproduce the device's raw outputs
    B, Tc = 2, 8
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc)
    nz_dev, nz_or = _noise(cfg, T, B, seed=4)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, None, steps_per_graph=16)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0
    _, r_or = O.incremental(params, cfg, c, noise=nz_or, test_inputs=out.cpu().unsqueeze(-1), formulation='ring')
    assert rel_err(raw.cpu(), r_or) < 1.4e-2


def test_incremental_equals_batch_forward_on_device():
    # size-independent property (the reference's own design invariant), both sides HIP:
    # synth raw[t] (teacher-forced) == train forward on the shifted input, at a larger size.
    B, Tc = 4, 40
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, layers=6, stacks=2)
    nz_dev, _ = _noise(cfg, T, B)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=32)
    x_shift = torch.cat([torch.zeros(B, 1), wav[:, :-1]], 1).view(B, 1, T).contiguous().cuda()
    yhat = torch.empty(B, cfg.out_channels, T, device='cuda')
    loss = torch.zeros(1, device='cuda')
    eng.train_fwd(x_shift, c.cuda(), wav.view(B, T, 1).contiguous().cuda(), torch.full((B,), T, dtype=torch.int32, device='cuda'), 0, loss, yhat)
    torch.cuda.synchronize()
    e = rel_err(raw, yhat)
    print('\nincremental vs batch (both HIP) rel err %.3e' % e)
    assert e < 5e-3                                    # measured 1.3e-3
