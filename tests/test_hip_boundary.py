"""C-ABI boundary behaviour on the device (pytest -m gpu): inference-only contexts (ABI v3), the device Philox noise stream of
wn_synthesize(noise = NULL), asynchronous synthesis + wn_synth_check, and the default synthesis path of the drop-in entry points
(the persistent pipeline whenever the model fits it)."""
import numpy as np
import pytest
import torch

from hip_util import SMALL, device_normal_noise, device_uniform_noise, make_hp, oracle_cfg, rel_err, synth_batch, upload_params
from oracle import wavenet_oracle as O

pytestmark = pytest.mark.gpu


def _engine(hp, B, T, **kw):
    from wavenet_vocoder import _ext
    return _ext.Engine(hp, B, T, **kw)


def test_device_noise_stream_is_philox_bit_exact():
    hp = make_hp(**SMALL)
    eng = _engine(hp, 2, 64)
    B, T, nps = 3, 37, eng.noise_per_step                     # 3*37*11 = 1221 floats: not a multiple of 4 (tail path)
    buf = torch.full((T * B * nps + 3,), -1.0, device='cuda')
    eng.fill_noise(buf[:T * B * nps].view(T, B, nps), B, T, seed=0x1234567890ABCDEF)
    torch.cuda.synchronize()
    got = buf.cpu().numpy()
    exp = device_uniform_noise(T * B * nps, 0x1234567890ABCDEF)
    assert np.array_equal(got[:T * B * nps], exp)                                 # bit-exact uniforms
    assert (got[T * B * nps:] == -1.0).all()                                      # nothing written past the buffer
    assert exp.min() >= np.float32(1e-5) and exp.max() <= np.float32(1.0) - np.float32(1e-5)     # mixture.py:91,104 range
    # Gaussian head: Box-Muller on the same words -> standard normal statistics
    kw = dict(SMALL); kw.update(out_channels=2, log_scale_min_gauss=float(np.log(1e-7)))
    eng2 = _engine(make_hp(**kw), 2, 64)
    z = torch.empty(16384, 2, 1, device='cuda')
    eng2.fill_noise(z, 2, 16384, seed=8)
    z = z.cpu().double().flatten()
    np.testing.assert_allclose(z.numpy(), device_normal_noise(32768, 8), rtol=0, atol=2e-5)       # same draws as the mirror (libm log / cos differ in the last bits)
    assert abs(float(z.mean())) < 0.03 and abs(float(z.std()) - 1.0) < 0.03                        # ... and standard normal
    assert abs(float((z ** 3).mean())) < 0.1 and abs(float((z ** 4).mean()) - 3.0) < 0.3


def test_inference_only_context_and_default_path():
    """cfg.inference_only: ~30x less workspace, training entry points refuse, synthesis results identical to a training context;
    noise = NULL (device stream) == the same run with wn_fill_noise's buffer; the default path is the pipeline."""
    from wavenet_vocoder import _ext
    kw = dict(SMALL); kw.update(layers=6, stacks=2)
    hp = make_hp(**kw)
    cfg = oracle_cfg(hp)
    B, Tc = 3, 12
    T = Tc * cfg.hop
    params = O.init_params(cfg, seed=11, bias_scale=0.05)
    full = _engine(hp, B, T)
    inf = _engine(hp, B, T, inference_only=True)
    ws_full = full.lib.wn_workspace_bytes(full.h); ws_inf = inf.lib.wn_workspace_bytes(inf.h)
    assert ws_inf * 10 < ws_full, (ws_inf, ws_full)
    flat = upload_params(full, params)
    full.pack_weights(flat); inf.pack_weights(flat)
    wav, c = synth_batch(cfg, B, T, seed=3)
    outs = []
    for eng in (full, inf):
        out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
        eng.synthesize(c.cuda(), None, out, raw, None, seed=99)          # device noise, default path, asynchronous
        torch.cuda.synchronize(); eng.synth_check()
        assert eng.synth_path == 'pipeline'
        outs.append((out.cpu(), raw.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # the same stream, explicit: fill_noise -> synthesize(noise)
    nz = torch.empty(T, B, inf.noise_per_step, device='cuda')
    inf.fill_noise(nz, B, T, seed=99)
    out2 = torch.empty(B, T, device='cuda'); raw2 = torch.empty(B, cfg.out_channels, T, device='cuda')
    inf.synthesize(c.cuda(), nz, out2, raw2, None)
    torch.cuda.synchronize(); inf.synth_check()
    assert torch.equal(out2.cpu(), outs[1][0]) and torch.equal(raw2.cpu(), outs[1][1])
    # ... and those samples are what the oracle's sampler draws from the device's raw outputs with that noise (mixture.py:76-107)
    Mx = cfg.out_channels // 3
    nzc = nz.cpu()
    exp = O.sample_from_discretized_mix_logistic(outs[1][1], nzc[:, :, :Mx].permute(1, 0, 2), nzc[:, :, Mx].t(), cfg.log_scale_min)
    assert torch.allclose(outs[1][0], exp, atol=2e-5)
    # a different seed gives a different utterance
    out3 = torch.empty(B, T, device='cuda')
    inf.synthesize(c.cuda(), None, out3, None, None, seed=100)
    torch.cuda.synchronize(); inf.synth_check()
    assert not torch.equal(out3.cpu(), outs[1][0])
    # training entry points refuse on the inference-only context, with a status code (no crash)
    x = wav.view(B, 1, T).contiguous().cuda(); y = wav.view(B, T, 1).contiguous().cuda()
    ln = torch.full((B,), T, dtype=torch.int32, device='cuda'); loss = torch.zeros(1, device='cuda')
    with pytest.raises(_ext.WnError) as ei:
        inf.train_fwd(x, c.cuda(), y, ln, 0, loss)
    assert ei.value.code == -5
    # capacity is fixed at creation
    with pytest.raises(_ext.WnError) as ei:
        inf.synthesize(torch.rand(B, cfg.cin_channels, Tc * 4, device='cuda'), None, torch.empty(B, T * 4, device='cuda'), None, None)
    assert ei.value.code == -2


def test_facade_incremental_uses_pipeline_and_chunks_large_batches():
    """WaveNet.incremental (what synthesize.py / Synthesizer.synthesize / the eval step call): default path = pipeline; a batch of
    more than 8 streams goes through it in groups, stream results independent of the grouping."""
    from wavenet_vocoder.models import create_model
    kw = dict(SMALL); kw.update(layers=6, stacks=2)
    hp = make_hp(**kw)
    cfg = oracle_cfg(hp)
    B, Tc = 11, 6
    T = Tc * cfg.hop
    model = create_model('WaveNet', hp)
    model.build(B, T, inference_only=True)
    c = torch.rand(B, cfg.cin_channels, Tc, generator=torch.Generator().manual_seed(1)).cuda()
    nz = torch.rand(T, B, model.engine.noise_per_step, generator=torch.Generator().manual_seed(2)).cuda() * 0.98 + 0.01
    out, raw = model.incremental(None, c=c, noise=nz, return_raw=True)
    torch.cuda.synchronize(); model.engine.synth_check()
    assert model.engine.synth_path == 'pipeline' and out.shape == (B, T)
    # stream 9 alone (its own noise column) reproduces its row of the grouped run
    out1, raw1 = model.incremental(None, c=c[9:10].contiguous(), noise=nz[:, 9:10].contiguous(), return_raw=True)
    torch.cuda.synchronize()
    assert torch.equal(out1[0], out[9]) and torch.equal(raw1[0], raw[9])
    assert model.upsampled_local_features.shape == (1, cfg.cin_channels, T)
