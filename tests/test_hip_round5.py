"""Round-5 GPU tests: the persistent synthesis pipeline with IEEE-half storage (weights, hand-off granules, ring queues) next to the bf16
one, both against the FP32 oracle (the reference's loop is fp32: modules.py:273-303, wavenet.py:821-886) -- VERDICT round 4, item 5."""
import json
import os
import time

import numpy as np
import pytest
import torch

from hip_util import rel_err
from oracle import wavenet_oracle as O
from test_hip_bench_geometry import PAPER
from test_hip_synth import _noise, _setup

pytestmark = pytest.mark.gpu


def _run(eng, cfg, c, wav, nz_dev, B, T, f16):
    eng.pipeline_dtype(f16)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=0)        # warm-up run (slice images, residency)
    torch.cuda.synchronize(); eng.synth_check()
    t0 = time.time()
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=0)
    torch.cuda.synchronize(); eng.synth_check()
    dt = time.time() - t0
    assert eng.synth_path == 'pipeline'
    return out.cpu(), raw.cpu(), dt


def test_pipe_fp16_storage_vs_bf16_against_the_fp32_oracle():
    """C4's model (24 layers / 2 stacks, R = 256, 10-MoL), 2 streams x 22 000 teacher-forced steps (every ring wrapped), the pipeline in
    both storage types against the oracle's batch forward in FP32 (no rounding emulation): the distance to the reference's arithmetic is
    the price of the 16-bit type -- bf16 ~1e-2 (8 mantissa bits), half ~1e-3 (11) at the same speed."""
    B, Tc = 2, 80
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, **dict(PAPER, wavenet_dropout=0.0))
    assert T == 22000 and eng.pipeline_eligible(B)
    nz_dev, nz_or = _noise(cfg, T, B)
    with torch.no_grad():
        xs = torch.cat([torch.zeros(B, 1), wav[:, :-1]], 1).reshape(B, 1, T)
        r32 = O.step(params, cfg, xs, c)                                                        # fp32 oracle, teacher forced (== incremental)
    rec = {}
    for name, f16 in (('bf16', False), ('fp16', True)):
        out, raw, dt = _run(eng, cfg, c, wav, nz_dev, B, T, f16)
        per = [rel_err(raw[b], r32[b]) for b in range(B)]
        q = T // 4
        quarters = [rel_err(raw[:, :, i * q:(i + 1) * q], r32[:, :, i * q:(i + 1) * q]) for i in range(4)]
        exp = O.sample_from_discretized_mix_logistic(raw, nz_or['u1'].permute(1, 0, 2), nz_or['u2'].t(), cfg.log_scale_min)
        rec[name] = {'rel_l2_vs_fp32_oracle_per_stream': per, 'per_quarter': quarters, 'us_per_sample': dt / T * 1e6, 'samples_max_abs_vs_sampler_on_device_raw': float((out - exp).abs().max())}
        print('\npipeline storage %s: raw vs the FP32 oracle per stream %s (quarters %s); %.1f us per sample (B = %d)'
              % (name, ' '.join('%.2e' % e for e in per), ' '.join('%.2e' % e for e in quarters), dt / T * 1e6, B))
        assert torch.allclose(out, exp, atol=2e-5)
    d = os.environ.get('WN_PARITY_REPORT_DIR')
    if d:
        with open(os.path.join(d, 'parity_pipe_dtype.json'), 'w') as f:
            json.dump(rec, f, indent=1)
    assert max(rec['bf16']['rel_l2_vs_fp32_oracle_per_stream']) < 2.5e-2
    assert max(rec['fp16']['rel_l2_vs_fp32_oracle_per_stream']) < 4e-3                       # set from the first measurement (x <= 3)
    assert rec['fp16']['us_per_sample'] < 1.15 * rec['bf16']['us_per_sample']
    eng.close()


@pytest.mark.parametrize('kw', [dict(), dict(out_channels=2, legacy=True, residual_legacy=True, upsample_type='SubPixel'),
                                dict(input_type='mulaw-quantize', out_channels=256, quantize_channels=256),
                                dict(gin_channels=16, use_speaker_embedding=True, n_speakers=4)])
def test_pipe_fp16_small_models_match_the_fp32_oracle(kw):
    """The half-storage pipeline on the small test models (MoL / Gaussian / softmax heads, global conditioning), 3 streams, teacher forced,
    against the oracle's incremental loop in fp32."""
    from oracle import mulaw as M
    B, Tc = 3, 6
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, **kw)
    if not eng.pipeline_eligible(B):
        pytest.skip('model does not fit the pipeline')
    nz_dev, nz_or = _noise(cfg, T, B)
    g = None
    if cfg.gin_channels > 0:
        gg = torch.Generator().manual_seed(5)
        g = torch.randint(0, cfg.n_speakers, (B,), generator=gg).int()
        eng.set_global_condition(g.cuda())
    if cfg.input_type == 'mulaw-quantize':
        ids = torch.from_numpy(M.mulaw_quantize(wav.numpy())).int()
        ti_dev = ids.cuda(); ti_or = torch.nn.functional.one_hot(ids.long(), 256).float()
        out = torch.empty(B, T, dtype=torch.int32, device='cuda')
    else:
        ti_dev = wav.contiguous().cuda(); ti_or = wav.unsqueeze(-1)
        out = torch.empty(B, T, device='cuda')
    raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.pipeline_dtype(True)
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, ti_dev, steps_per_graph=0)
    torch.cuda.synchronize(); eng.synth_check()
    assert eng.synth_path == 'pipeline'
    with torch.no_grad():
        _, r_or = O.incremental(params, cfg, c, noise=nz_or, test_inputs=ti_or, g=g, formulation='ring')
    e = rel_err(raw.cpu(), r_or)
    print('\nfp16 pipeline %s: raw vs the fp32 incremental oracle %.2e' % (kw, e))
    assert e < 3e-3
    eng.close()


def test_pipe_fp16_overflow_is_reported_and_the_facade_falls_back_to_bf16():
    """IEEE half has 5 exponent bits: a model whose residual stream exceeds 65504 cannot ride the half pipeline.  The layer CUs test
    every value they queue (off the critical path) and raise the abort word; wn_synth_check names the layer; WaveNet.incremental
    (check = True: what Synthesizer calls) switches the context to bf16 storage, re-runs the batch and logs it."""
    from hip_util import upload_params
    from wavenet_vocoder.models.wavenet import WaveNet
    from wavenet_vocoder import _ext
    B, Tc = 2, 4
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc)
    big = {k: v.clone() for k, v in params.items()}
    for k in big:                                   # an input convolution that maps |x| <= 1 to ~1e5: the very first hand-off overflows
        if k.startswith('input_convolution') and k.endswith('kernel'):
            big[k] = big[k] * 0 + 2.0e5
    eng.pack_weights(upload_params(eng, big))
    nz_dev, _ = _noise(cfg, T, B)
    out = torch.empty(B, T, device='cuda')
    eng.pipeline_dtype(True)
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, None, wav.contiguous().cuda(), steps_per_graph=0)
    torch.cuda.synchronize()
    with pytest.raises(_ext.WnError, match='half-precision range'):
        eng.synth_check()
    eng.close()
    model = WaveNet(hp)
    model.build(B, T)
    model.params.copy_(upload_params(model.engine, big)); model._dirty = True
    logged = []
    import wavenet_vocoder.models.wavenet as W
    orig = W.log
    W.log = lambda m, **k: logged.append(m)
    try:
        got = model.incremental(None, c=c.cuda(), noise=nz_dev.cuda(), test_inputs=wav.cuda(), check=True)
    finally:
        W.log = orig
    assert any('bf16 pipeline storage' in m for m in logged), logged
    assert torch.isfinite(got).all() and model.engine.synth_path == 'pipeline' and model.synth_fallbacks == 1


def test_pipeline_instances_serve_the_default_synthesis_batch_side_by_side():
    """hparams.py: wavenet_synthesis_batch_size = 20 on hparams.py's own model width (R = 128: 4 CUs per layer).  A model whose L * P + 1 CUs fit the
    chip more than once runs a batch of more than 8 streams as several pipeline INSTANCES side by side (own CUs, mailboxes and queues each; runs
    of <= 8 streams cost the wall time of one).  Every stream against the FP32 oracle, the instance count, and a batch that is
    not divisible by the instance count (20 -> 7 + 7 + 6; 11 -> 6 + 5; 32 -> 11 + 11 + 10; 8 -> one run)."""
    for B, want_ni in ((20, 3), (11, 2), (32, 3), (8, 1)):
        hp, cfg, eng, params, wav, c, T = _setup(B, 6, layers=6, stacks=2, residual_channels=128, gate_channels=256, skip_out_channels=128)
        assert eng.pipeline_eligible(B)
        nz_dev, nz_or = _noise(cfg, T, B)
        out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
        eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=0)
        torch.cuda.synchronize(); eng.synth_check()
        assert eng.synth_path == 'pipeline' and eng.lib.wn_synth_last_instances(eng.h) == want_ni
        with torch.no_grad():
            _, r_or = O.incremental(params, cfg, c, noise=nz_or, test_inputs=wav.unsqueeze(-1), formulation='ring')
        per = [rel_err(raw.cpu()[b], r_or[b]) for b in range(B)]
        exp = O.sample_from_discretized_mix_logistic(raw.cpu(), nz_or['u1'].permute(1, 0, 2), nz_or['u2'].t(), cfg.log_scale_min)
        print('\n%d streams as %d pipeline instances: worst stream %.2e' % (B, want_ni, max(per)))
        assert max(per) < 4e-3 and torch.allclose(out.cpu(), exp, atol=2e-5)
        eng.close()
    # (hparams.py's own 20-layer model at hparams.py's own synthesis batch of 20 on three instances, against the oracle: tests/test_hip_round6.py)


@pytest.mark.parametrize('kw', [dict(), dict(gin_channels=16, use_speaker_embedding=True, n_speakers=4)])
def test_pipe_batched_premultiplication_matches_the_per_stream_form_and_the_oracle(kw):
    """VERDICT round 4, item 4 (several streams per iteration): on R = 256 models a layer CU no longer multiplies the past taps and the conditioning of
    a stream's next sample per stream (one 64 x (2R + C) matvec and three workgroup barriers each: 1.3 of its 3.7 us per stream); wave 3 parks the
    vectors and ONE [64 x K] x [K x streams] product per sample on the matrix cores serves all streams (wn_synth_pipe.hip pre_batch).  Same
    arithmetic up to the order of the fp32 sums: every stream against the FP32 oracle in both forms, the two forms against each other, the layer bias
    and the per-stream gate bias of global conditioning, dilation-1 layers (their tap t-d is the sample in flight), 1 / 5 / 17 / 20 / 24 streams
    (one and two 16-stream tiles, ragged), and 25 / 32 streams = more vectors than the freed tap-2 image holds: the rest is parked where W_out's was."""
    width = dict(residual_channels=256, gate_channels=512, skip_out_channels=256, cin_channels=80, num_mels=80)
    for B in (1, 5, 17, 20, 24, 25, 32):
        hp, cfg, eng, params, wav, c, T = _setup(B, 6, layers=6, stacks=2, **width, **kw)
        nz_dev, nz_or = _noise(cfg, T, B)
        g = None
        if kw:
            g = (torch.arange(B) % 4).to(torch.int32)
            eng.set_global_condition(g.cuda())
        raws = {}
        for bp in (1, 0):
            os.environ['WN_PIPE_BATCHPRE'] = str(bp); os.environ['WN_PIPE_INSTANCES'] = '1'      # (this 6-layer model would fit the chip three times: ONE run here)
            try:
                out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
                eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=0)
                torch.cuda.synchronize(); eng.synth_check()
            finally:
                os.environ.pop('WN_PIPE_BATCHPRE', None); os.environ.pop('WN_PIPE_INSTANCES', None)
            assert eng.lib.wn_synth_last_instances(eng.h) == 1
            assert eng.synth_path == 'pipeline' and eng.lib.wn_synth_last_batched(eng.h) == (1 if bp else 0)
            raws[bp] = raw.cpu()
            if bp and B in (5, 20, 32):      # the kernel with these widths as compile-time constants (what ran above) against the generic one: the same bits
                os.environ['WN_PIPE_SPEC'] = '0'; os.environ['WN_PIPE_INSTANCES'] = '1'
                try:
                    raw0 = torch.empty_like(raw)
                    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw0, wav.contiguous().cuda(), steps_per_graph=0)
                    torch.cuda.synchronize(); eng.synth_check()
                finally:
                    os.environ.pop('WN_PIPE_SPEC', None); os.environ.pop('WN_PIPE_INSTANCES', None)
                assert torch.equal(raw0.cpu(), raws[bp])
        with torch.no_grad():
            _, r_or = O.incremental(params, cfg, c, noise=nz_or, test_inputs=wav.unsqueeze(-1), formulation='ring', **({'g': g} if kw else {}))
        per = {bp: max(rel_err(raws[bp][b], r_or[b]) for b in range(B)) for bp in (1, 0)}
        both = max(rel_err(raws[1][b], raws[0][b]) for b in range(B))
        print('\n%d streams%s: batched %.2e / per-stream %.2e vs the FP32 oracle; batched vs per-stream %.2e' % (B, ' + global conditioning' if kw else '', per[1], per[0], both))
        assert per[1] < 4e-3 and per[0] < 4e-3 and both < 2e-3
        eng.close()


@pytest.mark.parametrize('kw', [dict(), dict(out_channels=2, legacy=True, residual_legacy=True, upsample_type='SubPixel'),
                                dict(gin_channels=16, use_speaker_embedding=True, n_speakers=4, layers=6, stacks=2)])
def test_pipe_kernel_specialised_for_the_default_model_widths(kw):
    """hparams.py's own widths (R = S = 128, gate 256: 4 CUs per layer) run the pipeline kernel that has them as compile-time constants (SPEC 2 in
    wn_synth_pipe.hip: a layer CU's time per stream is an instruction count; 32.7 -> 27.5 us per sample on the 20-layer model).  The
    parity tests of the generic kernel at those widths -- MoL, Gaussian / legacy, per-stream gate bias -- and the generic kernel itself (WN_PIPE_SPEC=0)
    on the same inputs: the two must agree to the last bit (same code, constants folded)."""
    import test_hip_round3 as T3
    import test_hip_synth_pipe as TP
    width = dict(residual_channels=128, gate_channels=256, skip_out_channels=128)
    if 'gin_channels' in kw:
        T3.test_pipe_global_conditioning_matches_oracle(dict(width, **kw))
    else:
        TP.test_pipe_teacher_forced_matches_oracle(dict(width, **kw))
    B, Tc = 3, 6
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, **dict(width, **kw))
    nz_dev, _ = _noise(cfg, T, B)
    if 'gin_channels' in kw:
        eng.set_global_condition(torch.tensor([0, 3, 1], dtype=torch.int32).cuda())
    raws = []
    for spec in ('1', '0'):
        os.environ['WN_PIPE_SPEC'] = spec
        try:
            out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
            eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=0)
            torch.cuda.synchronize(); eng.synth_check()
        finally:
            os.environ.pop('WN_PIPE_SPEC', None)
        assert eng.synth_path == 'pipeline'
        raws.append(raw.cpu())
    assert torch.equal(raws[0], raws[1])
    eng.close()
