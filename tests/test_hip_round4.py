"""Round-4 additions to the GPU parity suite (pytest -m gpu):

  * synthesis in the REFERENCE'S OWN ARITHMETIC (mi355_compute_dtype = 'fp32' honoured by wn_synthesize: fp32 weights read from the
    flat parameter buffer, fp32 ring queues, fp32 accumulation -- modules.py:273-303, wavenet.py:821-886): teacher-forced raw outputs
    vs the fp32 oracle and vs the reference-EXECUTED incremental goldens at 1e-4, and FREE-RUNNING samples equal to the reference
    execution's / the oracle's given the same noise (the bf16 paths can only be compared for the first few samples);
  * C4 parity at the batch bench.py times: 8 streams x 110 275 steps through the persistent pipeline, every stream against the oracle.
"""
import glob
import json
import os
import time

import numpy as np
import pytest
import torch

from hip_util import SMALL, make_hp, oracle_cfg, rel_err, synth_batch, upload_params
from oracle import mulaw as M
from oracle import wavenet_oracle as O
from test_hip_reference_golden import DEFAULTS, GOLD
from test_hip_round3 import PAPER
from test_hip_synth import _noise

pytestmark = pytest.mark.gpu

TOL_F32 = 1e-4            # north star: "outputs match the reference TF1 CPU path within a stated fp32 tolerance"; measured ~1e-6 (profiles/r5*_pytest_fp32_synth.log)


def _setup32(B, Tc, inference_only=False, **kw):
    from wavenet_vocoder import _ext
    k = dict(SMALL); k.update(kw); k['mi355_compute_dtype'] = 'fp32'
    hp = make_hp(**k)
    cfg = oracle_cfg(hp)
    T = Tc * cfg.hop
    eng = _ext.Engine(hp, B, T, inference_only=inference_only)
    params = O.init_params(cfg, seed=11, bias_scale=0.05)
    eng.pack_weights(upload_params(eng, params))
    wav, c = synth_batch(cfg, B, T, seed=3)
    return hp, cfg, eng, params, wav, c, T


@pytest.mark.parametrize('kw', [dict(), dict(out_channels=2, legacy=True, residual_legacy=True, upsample_type='SubPixel'),
                                dict(input_type='mulaw-quantize', out_channels=256, quantize_channels=256),
                                dict(gin_channels=16, use_speaker_embedding=True, n_speakers=4),
                                dict(gin_channels=8, use_speaker_embedding=False, use_bias=False, upsample_type='1D'),
                                dict(wavenet_weight_normalization=True, gate_channels=192, skip_out_channels=128)])
def test_fp32_synthesis_matches_fp32_oracle(kw):
    """Teacher-forced and free-running, every head / conditioning variant (and a gate width that is not a multiple of the 64-column
    workgroup: G / 2 = 96): raw outputs vs the oracle's incremental loop in the reference's formulation at the fp32 tolerance; free-running
    samples equal the oracle's given the same noise; hipGraph replay == eager launches bit for bit; B = 11 spans two stream groups."""
    B, Tc = 11, 6
    hp, cfg, eng, params, wav, c, T = _setup32(B, Tc, **kw)
    nz_dev, nz_or = _noise(cfg, T, B)
    g = None
    if cfg.gin_channels > 0:
        gg = torch.Generator().manual_seed(5)
        g = (torch.randint(0, cfg.n_speakers, (B,), generator=gg).int() if cfg.use_speaker_embedding
             else torch.randn(B, cfg.gin_channels, generator=gg))
        eng.set_global_condition(g.cuda())
    quant = cfg.input_type == 'mulaw-quantize'
    if quant:
        ids = torch.from_numpy(M.mulaw_quantize(wav.numpy())).int()
        ti_dev = ids.cuda(); ti_or = torch.nn.functional.one_hot(ids.long(), 256).float()
    else:
        ti_dev = wav.contiguous().cuda(); ti_or = wav.unsqueeze(-1)
    out = torch.empty(B, T, dtype=torch.int32 if quant else torch.float32, device='cuda')
    raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, ti_dev, steps_per_graph=1)
    torch.cuda.synchronize()
    assert eng.synth_path == 'graph-fp32' and not eng.pipeline_eligible(B)
    o_or, r_or = O.incremental(params, cfg, c, noise=nz_or, test_inputs=ti_or, formulation='reference', g=g)
    e = rel_err(raw.cpu(), r_or)
    # free-running: same noise, no teacher
    out_f = torch.empty_like(out); raw_f = torch.empty_like(raw)
    eng.synthesize(c.cuda(), nz_dev.cuda(), out_f, raw_f, None, steps_per_graph=8)
    out_g = torch.empty_like(out); raw_g = torch.empty_like(raw)
    eng.synthesize(c.cuda(), nz_dev.cuda(), out_g, raw_g, None, steps_per_graph=1)
    torch.cuda.synchronize()
    assert torch.equal(raw_f, raw_g) and torch.equal(out_f, out_g)                 # graph replay == eager, bit for bit
    of_or, rf_or = O.incremental(params, cfg, c, noise=nz_or, formulation='ring', g=g)
    e_f = rel_err(raw_f.cpu(), rf_or)
    if quant:
        same = float((out_f.cpu().long() == of_or.argmax(1)).float().mean())
        print('\nfp32 synthesis %s: teacher-forced raw %.2e, free-running raw %.2e, class ids equal %.4f' % (kw, e, e_f, same))
        assert same == 1.0
    else:
        d = float((out_f.cpu() - of_or.reshape(B, T)).abs().max())
        print('\nfp32 synthesis %s: teacher-forced raw %.2e, free-running raw %.2e, samples max |diff| %.2e' % (kw, e, e_f, d))
        assert d < 1e-4
    assert e < TOL_F32 and e_f < TOL_F32


def test_fp32_synthesis_context_grows_with_the_batch():
    """5 streams first (small queues and scratch), then the ABI's maximum of 32 on the same context (queues, per-step scratch and the
    captured graph are rebuilt) -- against the fp32 oracle; the first 5 streams of the wide run equal the narrow run bit for bit
    (streams are independent)."""
    B, Tc = 32, 3
    hp, cfg, eng, params, wav, c, T = _setup32(B, Tc)
    nz_dev, nz_or = _noise(cfg, T, B)
    out5 = torch.empty(5, T, device='cuda'); raw5 = torch.empty(5, cfg.out_channels, T, device='cuda')
    eng.synthesize(c[:5].contiguous().cuda(), nz_dev[:, :5].contiguous().cuda(), out5, raw5, wav[:5].contiguous().cuda(), steps_per_graph=4)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=4)
    torch.cuda.synchronize()
    assert eng.synth_path == 'graph-fp32'
    _, r_or = O.incremental(params, cfg, c, noise=nz_or, test_inputs=wav.unsqueeze(-1), formulation='reference')
    e = rel_err(raw.cpu(), r_or)
    print('\nfp32 synthesis, 32 streams after 5 on one context: teacher-forced raw rel err %.2e' % e)
    assert e < TOL_F32
    assert torch.equal(raw[:5], raw5) and torch.equal(out[:5], out5)


GOLD_INC = [p for p in GOLD if 'inc_tf_raw' in np.load(p).files]


@pytest.mark.parametrize('path', GOLD_INC, ids=[os.path.basename(p)[10:-4] for p in GOLD_INC])
def test_fp32_synthesis_matches_reference_executed_incremental(path):
    """The reference's own WaveNet.incremental, executed (oracle/gen_golden_stack.py): teacher-forced raw outputs AND the free-running
    generation -- samples and raw outputs -- with the reference's sampler noise.  Not via the oracle."""
    from wavenet_vocoder import _ext
    g = np.load(path)
    kw = dict(DEFAULTS); kw.update(json.loads(str(g['hparams_json'])))
    kw['hop_size'] = int(np.prod(kw['upsample_scales'])); kw['mi355_compute_dtype'] = 'fp32'
    hp = make_hp(**kw)
    B, T = g['wav'].shape
    eng = _ext.Engine(hp, B, T)
    ref_params = {k[len('params/'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('params/')}
    flat = torch.zeros(eng.n_params)
    for name, (shape, off) in eng.layout.items():
        flat[off:off + ref_params[name].numel()] = ref_params[name].reshape(-1)
    eng.pack_weights(flat.cuda())
    c = torch.from_numpy(g['c']).cuda()
    if 'g' in g.files:
        gg = torch.from_numpy(g['g'])
        eng.set_global_condition((gg.reshape(B).int() if hp.use_speaker_embedding else gg.reshape(B, -1).float()).cuda())

    def noise(tag):
        if hp.out_channels == 2:
            return torch.from_numpy(g['eps_' + tag]).unsqueeze(-1).contiguous().cuda()
        return torch.cat([torch.from_numpy(g['u1_' + tag]), torch.from_numpy(g['u2_' + tag]).unsqueeze(-1)], -1).contiguous().cuda()
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, hp.out_channels, T, device='cuda')
    eng.synthesize(c, noise('tf'), out, raw, torch.from_numpy(g['wav']).contiguous().cuda(), steps_per_graph=8)
    torch.cuda.synchronize()
    e_tf = rel_err(raw.cpu(), torch.from_numpy(g['inc_tf_raw']))
    eng.synthesize(c, noise('free'), out, raw, None, steps_per_graph=8)
    torch.cuda.synchronize()
    e_fr = rel_err(raw.cpu(), torch.from_numpy(g['inc_free_raw']))
    d_fr = float((out.cpu() - torch.from_numpy(g['inc_free_out']).reshape(B, T)).abs().max())
    print('\n[%s] fp32 synthesis vs reference execution: teacher-forced raw %.2e, free-running raw %.2e, free-running samples max |diff| %.2e'
          % (os.path.basename(path), e_tf, e_fr, d_fr))
    assert e_tf < TOL_F32 and e_fr < TOL_F32 and d_fr < 1e-4


def test_fp32_synthesis_c4_scale():
    """C4's model (24 layers / 2 stacks, R = 256, 10-MoL): 1 stream x 22 000 teacher-forced steps (> the receptive field 16 381, every ring
    wrapped) vs the fp32 oracle's batch forward on the shifted input, per quarter of the clip; then 2 200 FREE-RUNNING steps whose samples
    must equal the oracle's incremental loop given the same noise (>= 2 000 steps).  An inference-only context."""
    PAPER_FULL = dict(PAPER, wavenet_dropout=0.0)
    B, Tc = 1, 80
    hp, cfg, eng, params, wav, c, T = _setup32(B, Tc, inference_only=True, **PAPER_FULL)
    assert T == 22000
    nz_dev, nz_or = _noise(cfg, T, B)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    t0 = time.time()
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=50)
    torch.cuda.synchronize()
    dt = time.time() - t0
    raw = raw.cpu()
    with torch.no_grad():
        xs = torch.cat([torch.zeros(1, 1), wav[:, :-1]], 1).view(1, 1, T)
        r_or = O.step(params, cfg, xs, c)
    e = rel_err(raw, r_or)
    seg = [rel_err(raw[:, :, a:a + 5500], r_or[:, :, a:a + 5500]) for a in range(0, T, 5500)]
    print('\nfp32 synthesis, C4 model, 22 000 teacher-forced steps vs the fp32 oracle: %.2e (per quarter %s); %.1f s = %.0f us/step'
          % (e, ' '.join('%.2e' % s for s in seg), dt, dt / T * 1e6))
    assert e < TOL_F32 and max(seg) < TOL_F32
    exp = O.sample_from_discretized_mix_logistic(raw, nz_or['u1'].permute(1, 0, 2), nz_or['u2'].t(), cfg.log_scale_min)
    assert torch.allclose(out.cpu(), exp, atol=2e-5)
    # free-running
    Tf = 8 * cfg.hop
    cf = c[:, :, :8].contiguous()
    nzf_dev = nz_dev[:Tf].contiguous(); nzf_or = {k: v[:Tf] for k, v in nz_or.items()}
    outf = torch.empty(B, Tf, device='cuda'); rawf = torch.empty(B, cfg.out_channels, Tf, device='cuda')
    eng.synthesize(cf.cuda(), nzf_dev.cuda(), outf, rawf, None, steps_per_graph=50)
    torch.cuda.synchronize()
    with torch.no_grad():
        o_or, r_or2 = O.incremental(params, cfg, cf, noise=nzf_or, formulation='ring')
    d = float((outf.cpu() - o_or.reshape(B, Tf)).abs().max())
    print('   %d free-running steps: samples max |diff| vs the oracle %.2e, raw rel-L2 %.2e' % (Tf, d, rel_err(rawf.cpu(), r_or2)))
    assert Tf >= 2000 and d < 1e-4 and rel_err(rawf.cpu(), r_or2) < TOL_F32


@pytest.mark.parametrize('B', [20, 32])
def test_pipe_whole_batch_in_one_run(B):
    """hparams.py: wavenet_synthesis_batch_size = 20.  The persistent pipeline takes the whole batch in ONE run (streams follow each
    other through the layer ring; 256 B of LDS state per stream) instead of groups of 8: every stream against the oracle, and the
    facade (WaveNet.incremental, what Synthesizer calls) sends one run."""
    from test_hip_synth import _setup
    from wavenet_vocoder.models.wavenet import WaveNet
    Tc = 6
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, layers=6, stacks=2)
    assert eng.pipeline_eligible(B)
    nz_dev, nz_or = _noise(cfg, T, B)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=0)
    torch.cuda.synchronize(); eng.synth_check()
    assert eng.synth_path == 'pipeline'
    _, r_or = O.incremental(params, cfg, c, noise=nz_or, test_inputs=wav.unsqueeze(-1), formulation='ring')
    per = [rel_err(raw.cpu()[b], r_or[b]) for b in range(B)]
    print('\npipeline, %d streams in one run: worst stream %.2e' % (B, max(per)))
    assert max(per) < 4e-3                               # half storage (bf16 storage measured 6.1e-3)
    model = WaveNet(hp)
    model.build(B, T)
    model.params.copy_(upload_params(model.engine, params)); model._dirty = True
    logged = []
    import wavenet_vocoder.models.wavenet as W
    orig = W.log
    W.log = lambda m, **k: logged.append(m)
    try:
        got, raw2 = model.incremental(None, c=c.cuda(), noise=nz_dev.cuda(), test_inputs=wav.cuda(), return_raw=True, check=True)
    finally:
        W.log = orig
    assert any('(%d streams per run)' % B in m for m in logged), logged
    assert torch.equal(raw2, raw)


def test_pipe_c4_at_the_benched_batch_8x110275():
    """What bench.py's synthesis leg times -- 8 streams x 110 275 steps (5.0 s at 22.05 kHz) on the persistent pipeline -- against the
    oracle, EVERY stream, on the FINAL second of audio: by then every ring has wrapped 13 times and any drift over the 5 s would show.
    The oracle's batch forward (bf16-emulating, like the one-stream full-length test) runs on the window [T - 22 055 - receptive field, T):
    its outputs are exact from one receptive field in.  (All 8 streams over the full 5 s: profiles/r5a_parity_c4_b8_full.json, 9.74e-3 ..
    9.83e-3 per stream, worst second 9.86e-3 -- 230 s of oracle time, too long for the routine suite; one stream over the full length is
    test_c4_full_length_one_stream_vs_oracle.)"""
    from test_hip_synth import _setup
    PAPER_FULL = dict(PAPER, wavenet_dropout=0.0)
    B, Tc = 8, 401
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, **PAPER_FULL)
    assert T == 110275
    nz_dev, nz_or = _noise(cfg, T, B)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    t0 = time.time()
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=0)
    torch.cuda.synchronize(); eng.synth_check()
    dt = time.time() - t0
    assert eng.synth_path == 'pipeline'
    raw = raw.cpu()
    RF = eng.receptive_field
    sec = 22055
    fr0 = (T - sec - RF) // cfg.hop                    # first conditioning frame of the window (hop-aligned: the upsample net is per frame)
    w0 = fr0 * cfg.hop
    per = []
    t1 = time.time()
    with torch.no_grad():
        for b in range(B):
            xs = torch.cat([torch.zeros(1, 1), wav[b:b + 1, :-1]], 1)[:, w0:].reshape(1, 1, T - w0)
            r_em = O.step(params, cfg, xs, c[b:b + 1, :, fr0:])[0]          # FP32 oracle (half storage since round 5; bf16 storage: 9.7 - 9.9e-3 vs the emulating oracle, profiles/r5a_parity_c4_b8_full.json)
            per.append(rel_err(raw[b, :, T - sec:], r_em[:, T - sec - w0:]))
    rec = {'B': B, 'T': T, 'wall_s_device': dt, 'rtf_per_stream': dt / (T / 22050.0), 'rel_l2_final_second_per_stream': per, 'window_start': w0, 'oracle_seconds': time.time() - t1}
    print('\npipe C4 at the benched batch (8 x 110 275), final second of every stream: %s; device %.2f s (RTF %.2f), oracle %.0f s'
          % (' '.join('%.2e' % e for e in per), dt, rec['rtf_per_stream'], rec['oracle_seconds']))
    d = os.environ.get('WN_PARITY_REPORT_DIR')
    if d:
        with open(os.path.join(d, 'parity_c4_b8_final_second.json'), 'w') as f:
            json.dump(rec, f, indent=1)
    assert max(per) < 4e-3
    exp = O.sample_from_discretized_mix_logistic(raw, nz_or['u1'].permute(1, 0, 2), nz_or['u2'].t(), cfg.log_scale_min)
    assert torch.allclose(out.cpu(), exp, atol=2e-5)
