"""Round-6 GPU tests: parity AT THE SYNTHESIS GEOMETRIES bench.py TIMES ("Next round" item 1).

bench.py's `pipe_B16` / `pipe_B20` legs run the 24-layer paper model with 16 / 20 streams in ONE pipeline run on the default switches
(two head CUs, early requests from 18 streams, the kernel with the paper widths as compile-time constants, batched pre-multiplication)
and hparams.py's own 20-layer model with 20 streams on three side-by-side instances.  Until this round those code paths were checked
on 6-layer stand-ins only (tests/test_hip_round5.py, test_hip_round4.py); the 24-layer model at 2 and 8 streams.  Reference loop:
wavenet.py:724-911 (incremental), hparams.py:301-304 (wavenet_synthesis_batch_size = 20)."""
import json
import os
import time

import pytest
import torch

from hip_util import make_hp, oracle_cfg, rel_err, upload_params
from oracle import wavenet_oracle as O
from test_hip_bench_geometry import PAPER
from test_hip_synth import _noise, _setup

pytestmark = pytest.mark.gpu

PIPE_SWITCHES = ('WN_PIPE_BATCHPRE', 'WN_PIPE_SPEC', 'WN_PIPE_EARLY_FROM', 'WN_PIPE_INSTANCES', 'WN_PIPE_ABORT_EVERY', 'WN_PIPE_DTYPE', 'WN_PIPE_HEADS')


def _assert_default_switches():
    live = {k: os.environ[k] for k in PIPE_SWITCHES if k in os.environ}
    assert not live, 'this test pins the DEFAULT pipeline configuration; unset %s' % live


@pytest.mark.parametrize('B', [20, 16])
def test_pipe_paper_model_at_the_benched_stream_counts_vs_the_fp32_oracle(B):
    """The paper model (24 layers / 2 stacks, R = S = 256, 10-MoL: 193 CUs, SPEC 1 kernel) with 20 (and 16) streams in ONE pipeline run,
    27 500 teacher-forced steps (the d = 2048 rings -- 4097 rows -- wrap six times), default switches.  EVERY stream against the FP32 oracle on
    the final 5 500 samples: the oracle's batch forward runs on the window [T - 5 500 - receptive field, T) (hop aligned) and is exact from
    one receptive field in (the windowed scheme of test_pipe_c4_at_the_benched_batch_8x110275, which covers 8 streams x 110 275).  Bound
    4e-3 as in every half-storage pipeline test (measured at 2 / 8 streams: 1.2e-3)."""
    _assert_default_switches()
    Tc = 100
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, **dict(PAPER, wavenet_dropout=0.0))
    assert T == 27500 and eng.pipeline_eligible(B)
    nz_dev, nz_or = _noise(cfg, T, B)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    t0 = time.time()
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=0)
    torch.cuda.synchronize(); eng.synth_check()
    dt = time.time() - t0
    assert eng.synth_path == 'pipeline'
    assert eng.lib.wn_synth_last_instances(eng.h) == 1 and eng.lib.wn_synth_last_batched(eng.h) == 1
    raw = raw.cpu()
    RF = eng.receptive_field
    assert RF == 16381
    tail = 5500
    fr0 = (T - tail - RF) // cfg.hop
    w0 = fr0 * cfg.hop
    assert w0 > 0 and T - tail - w0 >= RF - 1
    per = []
    t1 = time.time()
    with torch.no_grad():
        for b0 in range(0, B, 4):           # four streams per oracle call (~0.5 GB per activation tensor)
            nb = min(4, B - b0)
            xs = torch.cat([torch.zeros(nb, 1), wav[b0:b0 + nb, :-1]], 1)[:, w0:].reshape(nb, 1, T - w0)
            r = O.step(params, cfg, xs, c[b0:b0 + nb, :, fr0:])
            per += [rel_err(raw[b0 + i, :, T - tail:], r[i, :, T - tail - w0:]) for i in range(nb)]
    rec = {'B': B, 'T': T, 'wall_s_device_incl_first_run_setup': dt, 'rel_l2_final_5500_per_stream': per, 'window_start': w0, 'oracle_seconds': time.time() - t1}
    print('\npaper model, %d streams x %d steps in one pipeline run (default switches), final %d samples of every stream vs the FP32 oracle: worst %.2e, best %.2e; oracle %.0f s'
          % (B, T, tail, max(per), min(per), rec['oracle_seconds']))
    d = os.environ.get('WN_PARITY_REPORT_DIR')
    if d:
        with open(os.path.join(d, 'parity_pipe_paper_b%d.json' % B), 'w') as f:
            json.dump(rec, f, indent=1)
    assert max(per) < 4e-3
    exp = O.sample_from_discretized_mix_logistic(raw, nz_or['u1'].permute(1, 0, 2), nz_or['u2'].t(), cfg.log_scale_min)
    assert torch.allclose(out.cpu(), exp, atol=2e-5)
    eng.close()


def test_pipe_default_hparams_model_20_streams_on_three_instances_vs_the_fp32_oracle():
    """hparams.py's OWN model (20 layers / 2 stacks, R = S = 128, gate 256, Gaussian head, SubPixel [11, 25]: 81 CUs, SPEC 2 kernel) at
    hparams.py's own wavenet_synthesis_batch_size = 20: three pipeline instances of 7 + 7 + 6 streams in one launch, 11 000 teacher-forced
    steps (receptive field 4 093: every ring wraps).  Every stream over the WHOLE clip against the FP32 oracle -- round 5 asserted
    `isfinite` here and took parity from a 6-layer stand-in."""
    import hparams as H
    from wavenet_vocoder import _ext
    _assert_default_switches()
    hp = H._build()
    hp.set_hparam('wavenet_dropout', 0.0)
    cfg = oracle_cfg(hp)
    assert (cfg.layers, cfg.residual_channels, cfg.gate_channels, cfg.out_channels) == (20, 128, 256, 2)
    B, Tc = int(hp.wavenet_synthesis_batch_size), 40
    assert B == 20
    T = Tc * cfg.hop
    eng = _ext.Engine(hp, B, T, inference_only=True)
    assert eng.pipeline_eligible(B)
    params = O.init_params(cfg, seed=11, bias_scale=0.05)
    eng.pack_weights(upload_params(eng, params))
    from hip_util import synth_batch
    wav, c = synth_batch(cfg, B, T, seed=3)
    nz_dev, nz_or = _noise(cfg, T, B)
    out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=0)
    torch.cuda.synchronize(); eng.synth_check()
    assert eng.synth_path == 'pipeline' and eng.lib.wn_synth_last_instances(eng.h) == 3
    raw = raw.cpu()
    with torch.no_grad():
        xs = torch.cat([torch.zeros(B, 1), wav[:, :-1]], 1).reshape(B, 1, T)
        r32 = O.step(params, cfg, xs, c)
    per = [rel_err(raw[b], r32[b]) for b in range(B)]
    last = [rel_err(raw[b, :, -2000:], r32[b, :, -2000:]) for b in range(B)]
    print('\nhparams.py model, 20 streams on 3 instances x %d steps vs the FP32 oracle: worst stream %.2e (final 2000 samples %.2e)' % (T, max(per), max(last)))
    d = os.environ.get('WN_PARITY_REPORT_DIR')
    if d:
        with open(os.path.join(d, 'parity_pipe_default_b20.json'), 'w') as f:
            json.dump({'B': B, 'T': T, 'rel_l2_per_stream': per, 'rel_l2_final_2000_per_stream': last}, f, indent=1)
    assert max(per) < 4e-3 and max(last) < 4e-3
    exp = O.sample_from_gaussian(raw, nz_or['eps'].t(), cfg.log_scale_min_gauss)
    assert torch.allclose(out.cpu(), exp, atol=2e-5)
    eng.close()


def test_pipeline_beside_a_competing_workload_completes_or_falls_back_cleanly():
    """VERDICT round 5 ("Engineering"): the persistent pipeline needs its L * P + heads workgroups CO-RESIDENT (193 of 256 CUs for the paper model, each with
    ~145 KB of LDS: a CU of its own).  What happens when something else holds CUs -- a training job or a second synthesiser on the same GPU?  Here: ~2 s of
    back-to-back 8192^3 bf16 matmuls on another stream (hipBLASLt workgroups on every CU) are enqueued, then the facade synthesises a teacher-forced clip of
    the 24-layer paper model through the pipeline.  Documented behaviour, both branches accepted and checked: (a) the pipeline's workgroups get their CUs as
    the competitor's workgroups retire and the run completes (slower), or (b) a hand-off spin hits its bound (~0.5 s), the run raises its abort word, and
    WaveNet.incremental(check = True) re-runs the batch once on the launch-per-layer path and logs it.  Never a hang, never an exception to the caller,
    never garbage: the raw outputs equal those of an undisturbed run within the storage type's tolerance."""
    from wavenet_vocoder.models.wavenet import WaveNet
    import wavenet_vocoder.models.wavenet as W
    B, Tc = 2, 8
    hp, cfg, eng, params, wav, c, T = _setup(B, Tc, **dict(PAPER, wavenet_dropout=0.0))
    eng.close()
    model = WaveNet(hp)
    model.build(B, T)
    model.params.copy_(upload_params(model.engine, params)); model._dirty = True
    nz_dev, _ = _noise(cfg, T, B)
    quiet, raw_quiet = model.incremental(None, c=c.cuda(), noise=nz_dev.cuda(), test_inputs=wav.cuda(), return_raw=True, check=True)
    assert model.engine.synth_path == 'pipeline' and getattr(model, 'synth_fallbacks', 0) == 0
    a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16); b = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):                   # (first calls: library heuristics, not the kernel's time)
            (a @ b)
        side.synchronize()
        t0 = time.time()
        for _ in range(8):
            (a @ b)
        side.synchronize()
        per = (time.time() - t0) / 8
        n = int(min(4000, max(200, 2.0 / max(per, 1e-5))))
    with torch.cuda.stream(side):
        for _ in range(n):
            (a @ b)
    logged = []
    orig = W.log
    W.log = lambda m, **k: logged.append(m)
    t1 = time.time()
    try:
        busy, raw_busy = model.incremental(None, c=c.cuda(), noise=nz_dev.cuda(), test_inputs=wav.cuda(), return_raw=True, check=True)
    finally:
        W.log = orig
    dt = time.time() - t1
    torch.cuda.synchronize()
    fell_back = getattr(model, 'synth_fallbacks', 0) == 1
    e = rel_err(raw_busy.cpu(), raw_quiet.cpu())
    print('\npipeline beside ~%.1f s of competing matmuls (%d x %.2f ms): %s in %.2f s; raw outputs vs the undisturbed run %.2e'
          % (n * per, n, per * 1e3, 'fell back to the launch-per-layer path' if fell_back else 'completed on the pipeline', dt, e))
    assert torch.isfinite(raw_busy).all() and dt < 120
    if fell_back:
        assert model.engine.synth_path == 'graph' and any('launch-per-layer' in m or 'graph path' in m or 're-run' in m for m in logged), logged
        assert e < 3e-2                      # bf16 storage on the fallback path vs half on the pipeline
    else:
        assert model.engine.synth_path == 'pipeline' and e < 1e-6      # same kernel, same inputs: the competitor only delays it


GEOMETRIES = [(1, 112, [112]), (1, 128, [128]), (3, 144, [144, 129, 2]), (5, 272, [272, 256, 255, 130, 16]), (2, 656, [656, 513]), (7, 64, [64, 63, 48, 33, 32, 17, 3])]


@pytest.mark.parametrize('name', ['paper_width_drop', 'wide', 'mol_2d', 'gauss_subpixel'])
def test_geometry_sweep_forward_and_backward(name):
    """Batch / time geometries around the tile boundaries of every tile-engine configuration (LDS-DMA 256- and 128-channel kernels, the round-1 kernel of
    the 64-channel models): one utterance, odd batches (the two-stream split takes 1 + 2, 2 + 3, 3 + 4 utterances), T one hop below / at / above the 128-row tile
    and the 64-row tile, ragged lengths down to 2 samples (a masked-mean loss over almost nothing), a time axis shorter than one tile.  y_hat against the
    bf16-emulating oracle, the device loss against the oracle's loss on the device's y_hat, every gradient against autograd -- tolerances of test_hip_parity."""
    import numpy as np
    from test_hip_parity import _run_fwd
    from hip_util import download_grads
    worst_y, worst_g = 0.0, 0.0
    for B, T, lengths in GEOMETRIES:
        r = _run_fwd(name, B=B, T=T, lengths=lengths)
        cfg, eng = r['cfg'], r['eng']
        assert r['T'] == T
        y_em = O.step(r['params'], cfg, r['x_or'], r['c'], dropout_masks=r['masks'], emulate_bf16=True, g=r['g'])
        ey = rel_err(r['yhat_dev'].cpu(), y_em)
        loss_same = float(O.training_loss(cfg, r['yhat_dev'].cpu(), r['y_or'], r['lengths']))
        ld = float(r['loss_dev'].item())
        assert np.isfinite(ld) and abs(ld - loss_same) <= 2e-4 * max(1.0, abs(loss_same)), (name, B, T, ld, loss_same)
        grads_dev = torch.empty(eng.n_params, device='cuda')
        eng.train_bwd(grads_dev)
        torch.cuda.synchronize()
        g_dev = download_grads(eng, grads_dev)
        leaf = {k: v.clone().requires_grad_(True) for k, v in r['params'].items()}
        y = O.step(leaf, cfg, r['x_or'], r['c'], dropout_masks=r['masks'], emulate_bf16=True, g=r['g'])
        loss = O.training_loss(cfg, y, r['y_or'], r['lengths'])
        gs = torch.autograd.grad(loss, list(leaf.values()), allow_unused=True)
        g_or = {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(leaf, gs)}
        eg = rel_err(torch.cat([g_dev[k].flatten() for k in g_or]), torch.cat([g_or[k].flatten() for k in g_or]))
        print('\n[%s] B = %d, T = %d, lengths %s: y_hat %.2e, all gradients %.2e, loss %.5f' % (name, B, T, lengths, ey, eg, ld))
        assert ey < 2.4e-2 and eg < 7e-3, (name, B, T, ey, eg)
        worst_y, worst_g = max(worst_y, ey), max(worst_g, eg)
        eng.close()
    print('[%s] worst over %d geometries: y_hat %.2e, gradients %.2e' % (name, len(GEOMETRIES), worst_y, worst_g))
