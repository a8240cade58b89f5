"""Persistent dataflow synthesis pipeline (csrc/wn_synth_pipe.hip; selected with steps_per_graph=0) vs the oracle's
incremental loop and vs the other HIP paths.  Tolerances as in test_hip_synth.py (bf16 operands, fp32 accumulation);
integer class ids are bit-exact given the device's own logits + identical Gumbel uniforms."""
import numpy as np
import pytest
import torch

from hip_util import rel_err
from oracle import mulaw as M
from oracle import wavenet_oracle as O
from test_hip_synth import _noise, _setup

pytestmark = pytest.mark.gpu

PAPER_WIDTH = dict(residual_channels=256, gate_channels=512, skip_out_channels=256, cin_channels=80, num_mels=80)


@pytest.mark.parametrize('kw', [dict(), dict(out_channels=2, legacy=True, residual_legacy=True, upsample_type='SubPixel'),
                                dict(input_type='mulaw-quantize', out_channels=256, quantize_channels=256)])
def test_pipe_teacher_forced_matches_oracle(kw):
    B, Tc = 3, 6
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, **kw)
    nz_dev, nz_or = _noise(cfg, T, B)
    if cfg.input_type == 'mulaw-quantize':
        ids = torch.from_numpy(M.mulaw_quantize(wav.numpy())).int()
        ti_dev = ids.cuda(); ti_or = torch.nn.functional.one_hot(ids.long(), 256).float()
        out = torch.empty(B, T, dtype=torch.int32, device='cuda')
    else:
        ti_dev = wav.contiguous().cuda(); ti_or = wav.unsqueeze(-1)
        out = torch.empty(B, T, device='cuda')
    raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, ti_dev, steps_per_graph=0)
    torch.cuda.synchronize()
    o_or, r_or = O.incremental(params, cfg, c, noise=nz_or, test_inputs=ti_or, formulation='reference')
    e = rel_err(raw.cpu(), r_or)
    print('\npipe teacher-forced raw rel err %.3e' % e)
    assert e < 3e-2
    if cfg.input_type == 'mulaw-quantize':
        exp = torch.stack([O.sample_categorical(raw.cpu()[:, :, t], nz_or['gumbel_u'][t]) for t in range(T)], 1)
        assert torch.equal(out.cpu().long(), exp)                                   # bit-exact class ids
    elif cfg.out_channels == 2:
        exp = O.sample_from_gaussian(raw.cpu(), nz_or['eps'].t(), cfg.log_scale_min_gauss)
        assert torch.allclose(out.cpu(), exp, atol=2e-5)
    else:
        exp = O.sample_from_discretized_mix_logistic(raw.cpu(), nz_or['u1'].permute(1, 0, 2), nz_or['u2'].t(), cfg.log_scale_min)
        assert torch.allclose(out.cpu(), exp, atol=2e-5)
    # the launch-per-layer path computes the same thing
    out2 = torch.empty_like(out); raw2 = torch.empty_like(raw)
    eng.synthesize(c.cuda(), nz_dev.cuda(), out2, raw2, ti_dev, steps_per_graph=8)
    torch.cuda.synchronize()
    assert rel_err(raw, raw2) < 2e-2


def test_pipe_free_running_feedback_path():
    B, Tc = 2, 8
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc)
    nz_dev, nz_or = _noise(cfg, T, B, seed=4)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, None, steps_per_graph=0)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0
    _, r_or = O.incremental(params, cfg, c, noise=nz_or, test_inputs=out.cpu().unsqueeze(-1), formulation='ring')
    assert rel_err(raw.cpu(), r_or) < 3e-2


@pytest.mark.parametrize('B,kw', [(4, dict(layers=6, stacks=2)), (8, dict(layers=8, stacks=2)),
                                  (2, dict(PAPER_WIDTH, layers=6, stacks=2)), (1, dict(PAPER_WIDTH, layers=24, stacks=2))])
def test_pipe_incremental_equals_batch_forward_on_device(B, kw):
    # size-independent property, both sides HIP: synth raw[t] (teacher-forced) == train forward on the
    # shifted input.  The paper-width cases put 8 CUs on every layer (the benchmark geometry, 24 layers = 193 workgroups).
    Tc = 40 if kw.get('residual_channels', 64) == 64 else 24
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, **kw)
    nz_dev, _ = _noise(cfg, T, B)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=0)
    x_shift = torch.cat([torch.zeros(B, 1), wav[:, :-1]], 1).view(B, 1, T).contiguous().cuda()
    yhat = torch.empty(B, cfg.out_channels, T, device='cuda')
    loss = torch.zeros(1, device='cuda')
    eng.train_fwd(x_shift, c.cuda(), wav.view(B, T, 1).contiguous().cuda(), torch.full((B,), T, dtype=torch.int32, device='cuda'), 0, loss, yhat)
    torch.cuda.synchronize()
    e = rel_err(raw, yhat)
    print('\npipe incremental vs batch (both HIP) B=%d rel err %.3e' % (B, e))
    assert e < 2e-2
