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
    assert e < 4e-3                                    # half storage (the default since round 5): measured 0.9 - 1.2e-3 (bf16 storage: 3.3 - 4.4e-3, tests/test_hip_round5.py)
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
    assert rel_err(raw, raw2) < 1.4e-2                 # (the launch-per-layer path keeps bf16 weights and queues)


def test_pipe_free_running_feedback_path():
    B, Tc = 2, 8
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc)
    nz_dev, nz_or = _noise(cfg, T, B, seed=4)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, None, steps_per_graph=0)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0
    _, r_or = O.incremental(params, cfg, c, noise=nz_or, test_inputs=out.cpu().unsqueeze(-1), formulation='ring')
    assert rel_err(raw.cpu(), r_or) < 1.4e-2


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
    assert e < 2.8e-2                                  # measured 3.8e-3 .. 9.3e-3 (24 layers: two independently rounded HIP paths)


# ---- C4 scale (BASELINE configs[3]): the 24-layer / 2-stack paper model, against the ORACLE (not against another HIP path) ------
PAPER_FULL = dict(PAPER_WIDTH, layers=24, stacks=2, out_channels=30, upsample_type='2D', upsample_scales=[5, 5, 11], hop_size=275,
                  legacy=False, residual_legacy=False, NN_scaler=0.1, log_scale_min=float(np.log(1e-14)))
# The pipeline stores weights, hand-offs and queues in IEEE half since round 5 (fp32 accumulation): its raw outputs are compared with the FP32
# oracle -- the reference's arithmetic (modules.py:273-303) -- directly.  Measured 1.15 - 1.2e-3 at this depth (bf16 storage: 8.7e-3 from the
# fp32 oracle, 9.7e-3 from the bf16-emulating one; both dtypes side by side: tests/test_hip_round5.py, profiles/r6g_parity_pipe_dtype.json).
TOL_RAW_FP32 = 4e-3


def _oracle_teacher_forced(params, cfg, wav, c, emulate):
    """SURVEY.md A.8 (wavenet.py:724-911 vs :650-721): teacher-forced incremental generation == the batch forward on the input
    shifted by one sample with the silence start frame (wavenet.py:433-445) in front.  Stream by stream (memory)."""
    B, T = wav.shape
    out = torch.empty(B, cfg.out_channels, T)
    with torch.no_grad():
        for b in range(B):
            xs = torch.cat([torch.zeros(1, 1), wav[b:b + 1, :-1]], 1).view(1, 1, T)
            out[b] = O.step(params, cfg, xs, c[b:b + 1], emulate_bf16=emulate)[0]
    return out


def test_pipe_c4_scale_teacher_forced_vs_oracle():
    """B = 8 streams x 22 000 steps (> receptive field 16 381 and > the 8192-slot ring of the d = 2048 layers, so every ring
    buffer wraps and every tap reads a slot written by this run), raw outputs of ALL streams vs the oracle."""
    B, Tc = 8, 80
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, **PAPER_FULL)
    assert T == 22000 and eng.receptive_field == 16381
    nz_dev, nz_or = _noise(cfg, T, B)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=0)
    torch.cuda.synchronize()
    raw = raw.cpu()
    r_fp = _oracle_teacher_forced(params, cfg, wav, c, False)
    per = [rel_err(raw[b], r_fp[b]) for b in range(B)]
    tail = [rel_err(raw[b, :, 17000:], r_fp[b, :, 17000:]) for b in range(B)]      # steps whose whole receptive field went through wrapped rings
    print('\npipe C4-scale teacher-forced vs the FP32 oracle: per stream ' + ' '.join('%.2e' % e for e in per))
    print('   last 5000 steps only: ' + ' '.join('%.2e' % e for e in tail))
    assert max(per) < TOL_RAW_FP32 and max(tail) < TOL_RAW_FP32
    # the sampler ran on those raw outputs (mixture.py:76-107)
    exp = O.sample_from_discretized_mix_logistic(raw, nz_or['u1'].permute(1, 0, 2), nz_or['u2'].t(), cfg.log_scale_min)
    assert torch.allclose(out.cpu(), exp, atol=2e-5)


def test_pipe_c4_scale_free_running_feedback():
    """Free-running generation at depth: the oracle, teacher-forced with the DEVICE's own samples, must reproduce the device's raw
    outputs (every sample fed back through the input convolution, wavenet.py:869-878)."""
    B, Tc = 1, 80
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, **PAPER_FULL)
    nz_dev, nz_or = _noise(cfg, T, B, seed=9)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, None, steps_per_graph=0)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0
    r_fp = _oracle_teacher_forced(params, cfg, out.cpu(), c, False)
    e = rel_err(raw.cpu(), r_fp)
    print('\npipe C4-scale free-running vs the FP32 oracle (fed the device samples): %.2e' % e)
    assert e < TOL_RAW_FP32
