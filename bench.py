#!/usr/bin/env python3
"""Headline benchmark: WaveNet-vocoder TRAINING throughput (audio samples / s, whole job) on MI355X.

    python bench.py --gpus N --steps K --warmup W           (N>1: launched by torch.distributed.run)

One "step" = the full training step of the reference (wavenet_vocoder/train.py:303) on one batch per GPU:
pack weights -> forward (upsample net, 24 gated residual layers, head) -> masked MoL loss -> backward ->
[RCCL all-reduce(mean) of the flat fp32 gradient over ranks, wavenet.py:560-575] -> per-tensor clip +
TF-Adam + EMA.  Inputs are LJSpeech-shaped synthetic tensors already resident in HBM; weights are
random-init (Glorot) of the named architecture.

Workload (BASELINE.json configs[1], "C2"): paper_hparams WaveNet -- 24 layers / 2 stacks (BASELINE's
shape; --workload c2_4stack gives paper_hparams.py's 4-stack variant), R=256 G=512 S=256, 10-component
MoL, raw 16-bit scalar input, 80-mel local conditioning through the '2D' upsample net [5,5,11] (hop 275),
dropout 0.05, batch 8 x 11 000 samples per GPU, bf16 MFMA operands with fp32 accumulation.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (gate GEMM `wn_gemm_lds_kernel<2,2,4,2,32,3,EPI_GATE,1>`, one launch per
                  layer and step) timed live with HIP events on its launch stream inside the timed region;
  cpu_baseline -- the oracle (CPU restatement of the reference arithmetic, torch-CPU fp32 on all host
                  cores) running the same training step on a bounded sample of the same workload.
A short autoregressive-synthesis measurement (RTF at 22.05 kHz) is appended under "synthesis".
"""
import argparse
import json
import os
import sys
import time

# The step runs on 3 streams (two batch parts + the weight-gradient stream), data parallel adds the all-reduce stream(s): more than the
# runtime's default of 4 hardware queues, and streams that share a queue serialise (profiles/r2g_ab_batch_parts.txt).  Must be set before
# the HIP runtime initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'tacotron-2_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (hparams overrides on top of paper_hparams, B per GPU, T)
    'c2': (dict(stacks=2), 8, 11000),
    'c2_4stack': (dict(), 8, 11000),
    'c2_fp32': (dict(stacks=2, mi355_compute_dtype='fp32'), 8, 11000),      # C2 in the fp32 training mode (the reference's arithmetic; csrc/wn_f32.hip, vector-ALU SGEMM)
    'default_hparams': (None, 8, 11000),          # hparams.py defaults (Gaussian, R=128, SubPixel)
    'c5_stress': (dict(out_channels=2, residual_channels=512, gate_channels=1024, skip_out_channels=512, layers=30, stacks=3,
                       legacy=True, residual_legacy=True, upsample_type='SubPixel', upsample_scales=[15, 20], hop_size=300,
                       sample_rate=24000, cdf_loss=False), 8, 12000),
}


def build_hparams(workload):
    import hparams as base
    import paper_hparams as paper
    over, B, T = WORKLOADS[workload]
    if over is None:
        hp = base._build()
    else:
        o = dict(paper.PAPER_OVERRIDES); o.update(over)
        hp = base._build(o)
    return hp, B, T


def mac_per_sample(hp):
    """MAC_fwd = Cin*R + L(k R G + C G + G/2 S + G/2 R) + S S + S O."""
    R, G, S, O, C, L = hp.residual_channels, hp.gate_channels, hp.skip_out_channels, hp.out_channels, hp.cin_channels, hp.layers
    cin = 1 if hp.input_type != 'mulaw-quantize' else hp.quantize_channels
    return cin * R + L * (3 * R * G + C * G + (G // 2) * S + (G // 2) * R) + S * S + S * O


def alg_bytes_per_sample(hp, e=2):
    """bytes_train = 3 * e * [L (2R + C + 2S) + Cin + O] + 2 e L (G + G/2)."""
    R, G, S, O, C, L = hp.residual_channels, hp.gate_channels, hp.skip_out_channels, hp.out_channels, hp.cin_channels, hp.layers
    cin = 1 if hp.input_type != 'mulaw-quantize' else hp.quantize_channels
    return 3 * e * (L * (2 * R + C + 2 * S) + cin + O) + 2 * e * L * (G + G // 2)


def load_traffic(workload, B, T, path=None):
    """HBM-side traffic (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 PMC in separate passes, gfx950 FETCH_SIZE calibration) of the training step
    from the COMMITTED summary tools/pmc_summary.py --traffic wrote (profiles/traffic.json): per step over all kernels, and per launch
    of the dominant kernel.  None when the summary was measured on another workload / batch geometry (nothing is scaled or guessed)."""
    path = path or os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        with open(path) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    if t.get('workload') != workload or (int(t.get('batch', 0)), int(t.get('time', 0))) != (B, T):
        return None
    return {'bytes_per_step': t['bytes_per_step'], 'fetch_x2_bytes_per_step': t['fetch_x2_bytes_per_step'], 'write_bytes_per_step': t['write_bytes_per_step'],
            'gate_bytes_per_launch': t.get('gate_bytes_per_launch'), 'gate_launches_per_step': t.get('gate_launches_per_step'),
            'source': 'profiles/traffic.json <- %s (%s)' % (' + '.join(t.get('sources', [])), t.get('tag', ''))}


def synthetic_batch(hp, B, T, seed, device):
    g = torch.Generator().manual_seed(seed)
    hop = int(np.prod(hp.upsample_scales))
    t = torch.arange(T).float()
    f = torch.rand(B, 1, generator=g) * 320 + 80
    wav = (0.3 * torch.sin(2 * np.pi * f * t[None] / hp.sample_rate) + 0.1 * torch.randn(B, T, generator=g)).clamp(-0.999, 0.999)
    c = torch.rand(B, hp.cin_channels, T // hop, generator=g)
    x = wav.view(B, 1, T).contiguous().to(device)
    y = wav.view(B, T, 1).contiguous().to(device)
    lengths = torch.full((B,), T, dtype=torch.int32, device=device)
    return x, c.to(device), y, lengths, wav, c


def _log(msg):
    print('[bench %7.1fs] %s' % (time.time() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.time()


def cpu_baseline(hp, seconds_budget=20.0):
    """Oracle training step (fwd + loss + autograd bwd + clip + TF-Adam + EMA) on a bounded sample:
    the same architecture, batch 1 x (8 frames = 2200 samples), repeated until ~budget seconds.
    Threads: all host cores up to 64 (the bounded sample is too small to feed more)."""
    from oracle import wavenet_oracle as O
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    cfg = O.OracleConfig.from_hparams(hp)
    params = O.init_params(cfg, seed=5339)
    state = O.init_opt_state(params)
    B, Tc = 1, 8
    T = Tc * cfg.hop
    g = torch.Generator().manual_seed(0)
    wav = (torch.rand(B, T, generator=g) * 1.6 - 0.8)
    c = torch.rand(B, cfg.cin_channels, Tc, generator=g)
    masks = [(torch.rand(B, cfg.residual_channels, T, generator=g) >= cfg.wavenet_dropout).float() for _ in range(cfg.layers)]
    O.train_step(params, state, cfg, wav.view(B, 1, T), c, wav.view(B, T, 1), [T], 0, dropout_masks=masks)   # warm-up
    n, t0 = 0, time.time()
    while True:
        _, _, params, state = O.train_step(params, state, cfg, wav.view(B, 1, T), c, wav.view(B, T, 1), [T], n, dropout_masks=masks)
        n += 1
        if time.time() - t0 > seconds_budget or n >= 50:
            break
    dt = time.time() - t0
    return {'value': B * T * n / dt, 'unit': 'audio_samples/s', 'cores': cores, 'host_cores_total': os.cpu_count(), 'kind': 'port',
            'sample': 'oracle train step (fwd+loss+autograd bwd+clip+TF-Adam+EMA), same architecture, batch %dx%d samples, %d steps in %.1f s, torch-CPU fp32 %d threads'
                      % (B, T, n, dt, cores)}


def cpu_synth_baseline(hp, steps=2200, seconds_budget=15.0):
    """Oracle incremental loop (SURVEY 8d 'how the reference CPU path is timed', item 2) on a bounded sample: one stream of the same
    architecture, `steps` samples (>= 2000) or until the budget runs out, RTF extrapolated from the per-step time.  Two formulations:
    the reference's own ([B, 2d+1, R] queues rebuilt by slice + concat every step, modules.py:285-288) and O(1) ring buffers (what
    the HIP path does), to separate the algorithmic from the hardware gain.  The per-step work is ~150 small matvecs, so a few threads."""
    from oracle import wavenet_oracle as O
    cores = min(os.cpu_count() or 1, 8)
    torch.set_num_threads(cores)
    cfg = O.OracleConfig.from_hparams(hp)
    params = O.init_params(cfg, seed=5339)
    g = torch.Generator().manual_seed(0)
    Tc = max(2, -(-steps // cfg.hop))
    c = torch.rand(1, cfg.cin_channels, Tc, generator=g)
    out = {'unit': 'audio_samples/s', 'cores': cores, 'host_cores_total': os.cpu_count(), 'kind': 'port', 'sample_rate': hp.sample_rate}
    with torch.no_grad():
        for form in ('reference', 'ring'):
            # time in slices so that a slow host stops at the budget instead of running all the steps
            done, t0 = 0, time.time()
            for T in (64, steps):
                if T == steps and done:
                    per = (time.time() - t0) / done
                    T = int(max(256, min(steps, seconds_budget / max(per, 1e-9))))
                    done, t0 = 0, time.time()
                if cfg.out_channels == 2:
                    noise = {'eps': torch.randn(T, 1, generator=g)}
                elif cfg.scalar_input:
                    noise = {'u1': torch.rand(T, 1, cfg.out_channels // 3, generator=g) * 0.98 + 0.01, 'u2': torch.rand(T, 1, generator=g) * 0.98 + 0.01}
                else:
                    noise = {'gumbel_u': torch.rand(T, 1, cfg.quantize_channels, generator=g) * 0.98 + 0.01}
                O.incremental(params, cfg, c, T=T, noise=noise, formulation=form)
                done = T
            dt = time.time() - t0
            out[form] = {'steps': done, 'wall_s': dt, 'samples_per_s': done / dt, 'rtf_extrapolated': hp.sample_rate * dt / done}
    out['value'] = out['reference']['samples_per_s']
    out['sample'] = ('oracle incremental loop, same architecture, 1 stream x %d (reference queues) / %d (ring buffers) samples, torch-CPU fp32 %d threads; '
                     'RTF = extrapolated wall / audio time at %d Hz' % (out['reference']['steps'], out['ring']['steps'], cores, hp.sample_rate))
    return out


def other_workload_subprocess(key, device_index=0, hard_timeout=240):
    import subprocess
    code = ('import sys, json, torch; sys.path.insert(0, %r); import bench; torch.cuda.set_device(%d); '
            'print("OTHERWL" + json.dumps(bench.measure_other_workload(%r, torch.device("cuda", %d))))' % (ROOT, device_index, key, device_index))
    try:
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=hard_timeout)
        for line in r.stdout.splitlines():
            if line.startswith('OTHERWL'):
                return json.loads(line[len('OTHERWL'):])
        return {'error': 'failed: ' + r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {'error': 'timed out after %d s' % hard_timeout}


def cpu_full_batch(hp, B=None, T=None):
    """The oracle on the WHOLE bench batch (not the bounded sample of cpu_baseline): forward + loss + autograd backward utterance by
    utterance (the masked-mean loss of the batch is the length-weighted mean of the per-utterance losses, so the gradients add up),
    then ONE clip + TF-Adam + EMA update -- the full training step on all host cores.  Run only on request (--cpu-full-batch): about a
    minute with 64 threads (the bounded sample's rate: 2.4 k samples/s), far more on a small host; the caller bounds it (600 s)."""
    from collections import OrderedDict
    from oracle import wavenet_oracle as O
    B, T = B or 8, T or 11000
    cores = min(os.cpu_count() or 1, 64)      # (256 intra-op threads on these layer-sized convolutions are slower than 64: the first run with all cores did not finish in 13 min)
    torch.set_num_threads(cores)
    cfg = O.OracleConfig.from_hparams(hp)
    T = T // cfg.hop * cfg.hop
    params = O.init_params(cfg, seed=5339)
    state = O.init_opt_state(params)
    g = torch.Generator().manual_seed(0)
    wav = torch.rand(B, T, generator=g) * 1.6 - 0.8
    c = torch.rand(B, cfg.cin_channels, T // cfg.hop, generator=g)
    t0 = time.time()
    total = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
    for b in range(B):
        masks = [(torch.rand(1, cfg.residual_channels, T, generator=g) >= cfg.wavenet_dropout).float() for _ in range(cfg.layers)]
        leaf = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in params.items())
        y_hat = O.step(leaf, cfg, wav[b:b + 1].view(1, 1, T), c[b:b + 1], dropout_masks=masks)
        loss = O.training_loss(cfg, y_hat, wav[b:b + 1].view(1, T, 1), [T])
        gr = torch.autograd.grad(loss, list(leaf.values()), allow_unused=True)
        for k, gk in zip(leaf, gr):
            if gk is not None:
                total[k] += gk / B
    lr = O.learning_rate(0)
    for k, pk in params.items():
        m, v, e = state[k]
        O.adam_ema_update(pk, O.clip_gradient(total[k]), m, v, e, 1, lr)
    dt = time.time() - t0
    return {'value': B * T / dt, 'unit': 'audio_samples/s', 'cores': cores, 'host_cores_total': os.cpu_count(), 'kind': 'port', 'wall_s': dt,
            'sample': 'oracle training step (fwd + loss + autograd bwd per utterance, one clip + TF-Adam + EMA) on the FULL %d x %d batch, torch-CPU fp32 %d threads, run here' % (B, T, cores)}


def cpu_baseline_subprocess(workload, hard_timeout=150, fn='cpu_baseline'):
    """Run the CPU leg in a child process so that a pathological host (thread oversubscription) can never
    take the GPU number down with it."""
    import subprocess
    code = ('import sys, json; sys.path.insert(0, %r); import bench; hp, B, T = bench.build_hparams(%r); '
            'print("CPUBASE" + json.dumps(bench.%s(hp%s)))' % (ROOT, workload, fn, ', B, T' if fn == 'cpu_full_batch' else ''))
    try:
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=hard_timeout)
        for line in r.stdout.splitlines():
            if line.startswith('CPUBASE'):
                return json.loads(line[len('CPUBASE'):])
        return {'value': None, 'unit': 'audio_samples/s', 'cores': None, 'kind': 'port', 'sample': 'failed: ' + r.stderr[-200:]}
    except subprocess.TimeoutExpired:
        return {'value': None, 'unit': 'audio_samples/s', 'cores': None, 'kind': 'port', 'sample': 'timed out after %d s' % hard_timeout}


def synth_latency_budget(hp, scfg, B, measured_us):
    """Latency budget of ONE generated sample on the persistent pipeline (replaces round 5's `weight_reread_GBps`, which over-counted LDS
    reads under the batched pre-multiplication).  A sample is a chain of L + 1 dependent hand-offs (layer l -> l + 1 ... -> head -> layer 0):
        budget = L * (hop + layer critical path) + cross-XCD surcharge * (hops that change XCD) + head service + head -> layer-0 hop
                 + per-stream service * max(0, streams of one instance - knee)
    with the hop priced by the chip, not by this kernel: MI355X_MICROARCH.md "handoff-1to1" (one producer -> one consumer, data-tagged
    granules <= 4 KB, idle endpoints: 0.8 us; cross-XCD + 0.1 ... 0.3) -- the layer's P CUs gather P x 688 B, the same class.  The other
    terms are this pipeline's own stage traces: critical path through a layer CU (rebuild x, current-tap matvec, gate, out matvec: 0.2 us with
    the widths as compile-time constants, profiles/r8h), head (two convolutions + sampler + publish: 2.05 us, profiles/r4c_pipe_trace_b1.txt),
    per-stream service past the knee (1.7 us on the paper model, knee 12 streams; hparams.py's widths 0.6 us, knee 8: profiles/r8h, r8j).
    frac = budget / measured: 1.0 = nothing left but the chip's hand-off price."""
    L = int(hp.layers)
    P = int(hp.gate_channels) // 64
    spx = (L + 7) // 8
    paper = int(hp.residual_channels) >= 256
    hop, xcd_extra, layer_cp, head = 0.8, 0.2, 0.2, 2.05
    per_stream, knee = (1.7, 12) if paper else (0.6, 8)
    bi = int(scfg.get('streams_per_instance') or B)
    cross = max(0, -(-L // spx) - 1) + 1                    # layer groups on different XCDs + the way back to layer 0's XCD
    budget = L * (hop + layer_cp) + cross * xcd_extra + head + hop + per_stream * max(0, bi - knee)
    return {'hops': L + 1, 'hop_us': hop, 'cross_xcd_hops': cross, 'cross_xcd_extra_us': xcd_extra, 'layer_critical_path_us': layer_cp, 'head_us': head,
            'per_stream_us_past_knee': per_stream, 'knee_streams': knee, 'streams_per_instance': bi, 'cus_per_layer': P,
            'budget_us': budget, 'measured_us': measured_us, 'frac': budget / measured_us if measured_us else None,
            'source': 'MI355X_MICROARCH.md price list (handoff-1to1) + profiles/r4c_pipe_trace_b1.txt, r8h_pipe_batch_scaling.txt, r8j_pipe_default_model_spec2.txt'}


def synthesis_summary(res):
    """The RTF half of BASELINE's metric in a dozen numbers, LAST key of the JSON line (the driver keeps the tail of long lines)."""
    syn = res.get('synthesis') or {}
    out = {'sample_rate': None, 'deadline_us': None}
    for b in (1, 8, 16, 20):
        leg = syn.get('pipe_B%d' % b) or {}
        if 'rtf_per_stream' in leg:
            out['rtf_b%d' % b] = round(leg['rtf_per_stream'], 4); out['us_per_step_b%d' % b] = round(leg['us_per_step'], 2)
            out['deadline_us'] = round(leg['deadline_us'], 2)
            out['budget_frac_b%d' % b] = round((leg.get('latency_budget') or {}).get('frac') or 0.0, 3)
    leg = syn.get('pipe_B20') or syn.get('pipe_B8') or {}
    out['storage'] = leg.get('pipeline_storage'); out['path'] = leg.get('path'); out['real_time_b20'] = (syn.get('pipe_B20') or {}).get('real_time')
    out['switches_live'] = leg.get('switches_live')
    g = syn.get('graph_B1') or {}
    if 'rtf_per_stream' in g:
        out['rtf_graph_path_b1'] = round(g['rtf_per_stream'], 3)
    d = ((res.get('other_workloads') or {}).get('default_hparams') or {}).get('synthesis_pipeline') or {}
    if 'rtf_per_stream' in (d.get('pipe_B20') or {}):
        out['default_hparams_rtf_b20'] = round(d['pipe_B20']['rtf_per_stream'], 4)
    if 'synthesis_parity_gate' in res:
        out['parity_gate'] = res['synthesis_parity_gate'].get('ok')
    out.pop('sample_rate')
    return out


def measure_synthesis(hp, eng_params_flat, device, seconds=5.0, batches=(1, 8), modes=('pipe', 'graph')):
    """Autoregressive synthesis RTF at hp.sample_rate (BASELINE configs[3]: batch {1, 8} x 5 s from fixed mel conditioning).
    'pipe' = the persistent dataflow pipeline (steps_per_graph=0), timed on the full 5 s clip; 'graph' = the
    launch-per-layer hipGraph path, timed on 0.25 s (it is ~10x slower) for comparison."""
    from wavenet_vocoder import _ext
    hop = int(np.prod(hp.upsample_scales))
    out = {}
    for mode, secs, spg in (('pipe', seconds, 0), ('graph', min(seconds, 0.25), int(hp.mi355_steps_per_graph) or 16)):
        if mode not in modes:
            continue
        Tc = max(2, int(round(secs * hp.sample_rate / hop)))
        T = Tc * hop
        # the pipeline also at hparams.py's wavenet_synthesis_batch_size (20): ONE run, streams pipelined through the layer ring
        # (+ 16 streams: the knee of the curve -- a run costs the wall time of one stream up to ~12 streams, then ~1.7 us per stream)
        extra = tuple(b for b in (16, int(hp.wavenet_synthesis_batch_size)) if b not in batches) if mode == 'pipe' else ()
        for B in tuple(batches) + extra:
            _log('synthesis %s B=%d T=%d' % (mode, B, T))
            eng = _ext.Engine(hp, B, T, inference_only=True)       # synthesis-only context: pre-sized, ~1.5 KB of HBM per (stream x sample)
            eng.pack_weights(eng_params_flat)
            if mode == 'pipe' and not eng.pipeline_eligible(B):   # (would silently time the ~10x slower launch-per-layer path on the full clip)
                out['%s_B%d' % (mode, B)] = {'skipped': 'not pipeline-eligible at this batch'}
                eng.close()
                continue
            c = torch.rand(B, hp.cin_channels, Tc, device=device)
            samples = torch.empty(B, T, device=device)
            # sampling noise is drawn on the device (noise = None: Philox keyed by the seed); nothing is uploaded
            if mode == 'graph':
                eng.synthesize(c, None, samples, None, None, steps_per_graph=spg, seed=1)      # warm-up + graph build
            else:
                cw = c[:, :, :8].contiguous(); Tw = 8 * hop                            # short warm-up (weight slices, LDS images)
                eng.synthesize(cw, None, torch.empty(B, Tw, device=device), None, None, steps_per_graph=0, seed=1)
            torch.cuda.synchronize(); eng.synth_check()
            t0 = time.time()
            eng.synthesize(c, None, samples, None, None, steps_per_graph=spg, seed=2)
            torch.cuda.synchronize()
            dt = time.time() - t0
            eng.synth_check()
            scfg = eng.synth_config()                       # how the library configured the run it just timed (wn_synth_last_config), not what the environment asked for
            us = dt / T * 1e6
            out['%s_B%d' % (mode, B)] = {'seconds_of_audio_per_stream': T / hp.sample_rate, 'wall_s': dt, 'path': eng.synth_path,
                                        'instances': scfg.get('instances'), 'pipeline_storage': ('fp16' if scfg.get('half_storage') else 'bf16') if mode == 'pipe' else None,
                                        # 1: the layer CUs multiply every stream's past taps / conditioning in ONE matrix product per sample (R = 256 models, every eligible run of <= 32 streams)
                                        'batched_premultiplication': scfg.get('batched_premultiplication'), 'run_config': scfg,
                                        'switches_live': {k: v for k, v in sorted(os.environ.items()) if k.startswith('WN_PIPE_') or k in ('WN_SYNTH_MODE',)},
                                        'real_time': bool(us <= 1e6 / hp.sample_rate),
                                        'rtf_per_stream': dt / (T / hp.sample_rate), 'aggregate_samples_per_s': B * T / dt,
                                        'us_per_step': us, 'deadline_us': 1e6 / hp.sample_rate,
                                        # synthesis roofline: latency-bound -- a budget of dependent hand-offs, not a bandwidth
                                        'latency_budget': synth_latency_budget(hp, scfg, B, us) if mode == 'pipe' else None,
                                        'stream_tflops': 2.0 * mac_per_sample(hp) * B * T / dt / 1e12,
                                        'workspace_MB': eng.lib.wn_workspace_bytes(eng.h) / 1e6,
                                        'finite': bool(torch.isfinite(samples).all().item())}
            eng.close()
    return out


def synthesis_parity_gate(hp, eng_params_flat, device, B=None, frames=2, tol=4e-3):
    """a parity gate in front of the timed synthesis legs, on the switches that are live NOW.  No oracle here (only the
    cpu_baseline legs may touch oracle/): the persistent pipeline at hparams.py's synthesis batch against the PRODUCT's own fp32 synthesis
    mode (csrc/wn_synth_f32.hip: fp32 weights, queues and accumulation, pinned to the oracle at 2e-7 ... 6e-7 by tests/test_hip_round4.py) on
    the same weights, conditioning, teacher-forced inputs and noise: `frames` conditioning frames of every stream, raw outputs, rel-L2 per
    stream.  Bound 4e-3 = the bound of every half-storage pipeline test (measured 1.2e-3 ... 2.1e-3).  The full-length checks -- 20 streams x
    27 500 steps, 8 x 110 275 -- are tests/test_hip_round6.py / test_hip_round4.py; this gate catches a switch combination no test has seen."""
    import copy
    from wavenet_vocoder import _ext
    B = B or int(hp.wavenet_synthesis_batch_size)
    hop = int(np.prod(hp.upsample_scales)); T = frames * hop
    g = torch.Generator().manual_seed(77)
    c = torch.rand(B, hp.cin_channels, frames, generator=g).to(device)
    wav = (torch.rand(B, T, generator=g) * 1.6 - 0.8).to(device)
    eng = _ext.Engine(hp, B, T, inference_only=True)
    if hp.input_type == 'mulaw-quantize' or not eng.pipeline_eligible(B):
        eng.close()
        return {'ok': None, 'skipped': 'model not pipeline-eligible at B = %d (or class-id input)' % B}
    nps = eng.noise_per_step
    noise = (torch.randn(T, B, nps, generator=g) if hp.out_channels == 2 else torch.rand(T, B, nps, generator=g) * 0.98 + 0.01).to(device)
    raws = {}
    hp32 = copy.deepcopy(hp); hp32.set_hparam('mi355_compute_dtype', 'fp32')
    for name, h in (('pipeline', hp), ('fp32', hp32)):
        e = eng if name == 'pipeline' else _ext.Engine(h, B, T, inference_only=True)
        e.pack_weights(eng_params_flat)
        out = torch.empty(B, T, device=device); raw = torch.empty(B, hp.out_channels, T, device=device)
        e.synthesize(c, noise, out, raw, wav, steps_per_graph=0 if name == 'pipeline' else 16)
        torch.cuda.synchronize(); e.synth_check()
        raws[name] = (raw.double().cpu(), e.synth_path, e.synth_config() if name == 'pipeline' else None)
        e.close()
    a, b = raws['pipeline'][0], raws['fp32'][0]
    per = [float((a[i] - b[i]).norm() / (b[i].norm() + 1e-30)) for i in range(B)]
    return {'ok': bool(max(per) < tol and raws['pipeline'][1] == 'pipeline' and raws['fp32'][1] == 'graph-fp32'), 'tolerance': tol, 'streams': B, 'steps': T,
            'rel_l2_vs_fp32_mode_worst_stream': max(per), 'rel_l2_vs_fp32_mode_best_stream': min(per), 'paths': [raws['pipeline'][1], raws['fp32'][1]],
            'run_config': raws['pipeline'][2], 'what': 'pipeline (default switches as live) vs the product fp32 synthesis mode, teacher forced, same noise; see docstring'}


def measure_other_workload(key, device, steps=10, warmup=3):
    """A short run of another BASELINE workload through the same engine (c5_stress = configs[4], default_hparams = the shape that
    really is HBM-bound), same step function as the headline: ms/step plus its own whole-step HBM / MFMA fractions and the gate
    GEMM's pure-kernel fraction.  Reported next to the headline, never as `value`."""
    from wavenet_vocoder import _ext
    from wavenet_vocoder.models.modules import initialize_parameters
    hp, B, T = build_hparams(key)
    if key.endswith('_fp32'):      # an accuracy mode on the vector ALU (~2 orders slower): three steps say what it costs
        steps, warmup = 3, 1
    hop = int(np.prod(hp.upsample_scales)); T = T // hop * hop
    eng = _ext.Engine(hp, B, T, grad_buckets=1)
    flat = initialize_parameters(hp, eng.layout).to(device)
    grads = torch.zeros_like(flat); m, v, ema = torch.zeros_like(flat), torch.zeros_like(flat), flat.clone()
    loss = torch.zeros(1, device=device)
    x, c, y, lengths, _, _ = synthetic_batch(hp, B, T, seed=5339, device=device)

    def step(i):
        eng.pack_weights(flat)
        eng.train_fwd(x, c, y, lengths, 1000 + i, loss)
        eng.train_bwd(grads)
        lr = _ext.learning_rate(hp.wavenet_lr_schedule, hp.wavenet_learning_rate, i, hp.wavenet_decay_rate, hp.wavenet_decay_steps, hp.wavenet_warmup)
        eng.optim_step(flat, grads, m, v, ema, lr, i)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    eng.profile(True)
    t0 = time.time()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    k_ms, k_n = eng.profile_kernel_result()
    k_mhz, _ = eng.profile_kernel_clock()
    rows = eng.profile_rows_per_launch() or B * T
    eng.profile(False)
    R, G, C = hp.residual_channels, hp.gate_channels, hp.cin_channels
    value = B * T / dt
    gate_tf = (2.0 * G * (3 * R + C) * rows / (k_ms / k_n * 1e-3) / 1e12) if k_n else None
    f32 = getattr(hp, 'mi355_compute_dtype', 'bf16') == 'fp32'
    peak_tf = 157.3 if f32 else 2500.0
    e_act = 4 if f32 else 2
    out = {'workload_key': key, 'dtype': 'f32' if getattr(hp, 'mi355_compute_dtype', 'bf16') == 'fp32' else 'bf16', 'steps': steps, 'warmup': warmup, 'batch': B, 'time': T, 'layers': hp.layers, 'stacks': hp.stacks,
           'R': R, 'G': G, 'S': hp.skip_out_channels, 'out_channels': hp.out_channels, 'params': int(eng.n_params),
           'ms_per_step': dt * 1e3, 'value': value, 'unit': 'audio_samples/s', 'final_loss': float(loss.item()),
           'train_tflops_algorithmic': 6.0 * mac_per_sample(hp) * value / 1e12,
           # the matrix-pipe peak of the dtype the workload computes in: 2500 TFLOP/s dense bf16; 157.3 TFLOP/s fp32 (v_mfma_f32_32x32x2_f32)
           'mfma_peak_TFLOPs': peak_tf,
           'mfma_whole_step_frac': 6.0 * mac_per_sample(hp) * value / 1e12 / peak_tf,
           'hbm_whole_step_frac': alg_bytes_per_sample(hp, e_act) * value / 8e12, 'alg_bytes_per_sample': alg_bytes_per_sample(hp, e_act),
           'gate_kernel': {'achieved_TFLOPs': gate_tf, 'frac_of_peak': (gate_tf / peak_tf) if gate_tf else None, 'launches_timed': int(k_n),
                           'rows_per_launch': rows, 'sclk_in_kernel_mhz': k_mhz or None, 'timing': 'in-kernel stamps (pure kernel time), live in the two-stream step'},
           'bound': 'hbm' if alg_bytes_per_sample(hp, e_act) * peak_tf * 1e12 > 6.0 * mac_per_sample(hp) * 8e12 else 'mfma'}
    # measured fabric traffic of this workload's step where a committed PMC summary exists (profiles/traffic_<key>.json): the counter ratio next to the
    # algorithmic fraction, and the rate the memory system really ran at (6.3 TB/s is what a streaming copy reaches on this part)
    tr = load_traffic(key, B, T, path=os.path.join(ROOT, 'profiles', 'traffic_%s.json' % key))
    if tr:
        out['traffic_per_step'] = tr['bytes_per_step']; out['traffic_over_algorithmic'] = tr['bytes_per_step'] / (alg_bytes_per_sample(hp, e_act) * B * T)
        out['fabric_GBps_at_this_step_time'] = tr['bytes_per_step'] / dt / 1e9; out['traffic_source'] = tr['source'].replace('profiles/traffic.json', 'profiles/traffic_%s.json' % key)
    eng.close()
    if key == 'c5_stress':
        # synthesis at this width (R = S = 512 > the LDS-resident pipeline's 384): the launch-per-layer hipGraph path, 0.1 s of audio
        try:
            syn = measure_synthesis(hp, flat, device, seconds=0.1, batches=(1, 8), modes=('graph',))
            out['synthesis_graph_path'] = {k: {kk: v[kk] for kk in ('rtf_per_stream', 'us_per_step', 'deadline_us', 'path', 'seconds_of_audio_per_stream')} for k, v in syn.items()}
        except Exception as e:
            out['synthesis_graph_path'] = {'error': str(e)[:200]}
    if key == 'default_hparams':
        # hparams.py's own model at hparams.py's own synthesis batch (wavenet_synthesis_batch_size = 20): its 81 CUs fit the chip three times, so the batch runs as
        # three pipeline instances of 7 + 7 + 6 streams side by side in one launch; 2 s of audio per stream
        try:
            syn = measure_synthesis(hp, flat, device, seconds=2.0, batches=(8,), modes=('pipe',))      # (+ wavenet_synthesis_batch_size = 20)
            out['synthesis_pipeline'] = {k: {kk: v.get(kk) for kk in ('rtf_per_stream', 'us_per_step', 'deadline_us', 'real_time', 'path', 'instances', 'batched_premultiplication', 'seconds_of_audio_per_stream', 'aggregate_samples_per_s', 'run_config', 'latency_budget', 'switches_live')} for k, v in syn.items()}
        except Exception as e:
            out['synthesis_pipeline'] = {'error': str(e)[:200]}
    del flat, grads, m, v, ema
    torch.cuda.empty_cache()
    return out


def measure_with_feeder(hp, B, T, device, one_step, steps=30, n_utt=40):
    """K steps of the same step function with every batch coming from wavenet_vocoder.feeder.Feeder over a synthetic LJSpeech-shaped
    dataset written to a temp dir (utterances of 3-8 s, float32 audio + [-4, 4] mels): disk -> producer thread -> pinned -> H2D."""
    import shutil
    import tempfile
    from wavenet_vocoder.feeder import Feeder
    hop = int(np.prod(hp.upsample_scales))
    tmp = tempfile.mkdtemp(prefix='wn_bench_feeder_')
    try:
        rng = np.random.RandomState(5339)
        os.makedirs(os.path.join(tmp, 'audio')); os.makedirs(os.path.join(tmp, 'mels'))
        lines = []
        for i in range(n_utt):
            frames = int(rng.randint(3 * hp.sample_rate // hop, 8 * hp.sample_rate // hop))
            t = np.arange(frames * hop, dtype=np.float32)
            wav = np.clip(0.3 * np.sin(2 * np.pi * rng.uniform(80, 400) * t / hp.sample_rate) + 0.1 * rng.randn(frames * hop), -0.999, 0.999).astype(np.float32)
            mel = rng.uniform(-hp.max_abs_value, hp.max_abs_value, size=(frames, hp.num_mels)).astype(np.float32)
            a, m = os.path.join(tmp, 'audio', 'audio-%03d.npy' % i), os.path.join(tmp, 'mels', 'mel-%03d.npy' % i)
            np.save(a, wav); np.save(m, mel)
            lines.append('|'.join([a, m, m, '<no_g>', 'synthetic %d' % i]))
        meta = os.path.join(tmp, 'map.txt')
        with open(meta, 'w') as f:
            f.write('\n'.join(lines) + '\n')
        import copy
        fhp = copy.deepcopy(hp)
        fhp.parse('wavenet_batch_size=%d,max_time_steps=%d,wavenet_test_batches=1' % (B, T))

        class _Coord(object):
            stop = False

            def should_stop(self):
                return self.stop
        coord = _Coord()
        fd = Feeder(coord, meta, tmp, fhp, device=device)
        fd.start_threads()
        try:
            def fed_step(i):
                bx, by, bl, bc, _ = fd.next_train_batch()
                one_step(70000 + i, batch=(bx, bc, by, bl))
            for i in range(5):
                fed_step(i)
            torch.cuda.synchronize()
            t0 = time.time()
            for i in range(steps):
                fed_step(5 + i)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / steps
        finally:
            coord.stop = True
        return {'what': '%d untimed-by-contract steps, every batch from wavenet_vocoder.feeder.Feeder: %d .npy utterances (3-8 s) on local disk -> producer thread '
                        '(bucketing, hop-aligned crop to %d, mel clip + [0,1]) -> pinned host tensors -> non_blocking H2D -> the same step' % (steps, n_utt, T),
                'steps': steps, 'ms_per_step': dt * 1e3, 'value': B * T / dt, 'unit': 'audio_samples/s'}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class SmiSampler:
    """rocm-smi power / clock samples while a block runs (one subprocess call per ~second, in a thread).  Best effort: any failure
    just leaves the lists empty."""

    def __init__(self, device_index=0):
        import threading
        self.idx, self.power, self.sclk, self._stop = device_index, [], [], threading.Event()
        self._t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        import subprocess
        while not self._stop.is_set():
            try:
                r = subprocess.run(['rocm-smi', '-d', str(self.idx), '--showpower', '--showclocks'], capture_output=True, text=True, timeout=10)
                m = re.search(r'(?:Average|Current Socket) Graphics Package Power \(W\):\s*([0-9.]+)', r.stdout)
                if m:
                    self.power.append(float(m.group(1)))
                m = re.search(r'sclk clock level:.*?\((\d+)Mhz\)', r.stdout)
                if m:
                    self.sclk.append(float(m.group(1)))
            except Exception:
                pass
            self._stop.wait(0.5)

    def __enter__(self):
        if self.idx >= 0:           # (rank 0 only: eight ranks polling rocm-smi would only add noise)
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self.idx >= 0:
            self._t.join(timeout=15)

    def summary(self):
        out = {'samples': len(self.power)}
        if self.power:
            out.update(power_w_mean=float(np.mean(self.power)), power_w_max=float(np.max(self.power)))
        if self.sclk:
            out.update(sclk_mhz_mean=float(np.mean(self.sclk)), sclk_mhz_min=float(np.min(self.sclk)))
        return out


class _DryEngine(object):
    """Stand-in for the HIP engine in --dry-run: the bucket interface allreduce_mean_buckets_ walks, nothing else."""

    def __init__(self, n_params, buckets=3):
        self.n_params = n_params
        per = -(-n_params // buckets)
        self._b = [(o, min(per, n_params - o)) for o in range(0, n_params, per)]

    def grad_buckets(self):
        return self._b

    def wait_bucket(self, i, stream):
        return None


def gather_per_rank(dist, rank, world, ms_per_step, timer):
    """N > 1: every rank's own step time and exchange spans, collected on rank 0 (`per_rank` of the JSON line).  The driver computes
    scaling from its per-N runs; this says WHERE an N-rank step spends its exchange: per gradient bucket the span from its ready event
    to the end of its all-reduce on the communication stream, and how long the caller's stream stood at the join in front of the
    optimiser (0 = the exchange hid under the backward)."""
    mine = {'rank': rank, 'ms_per_step': ms_per_step, 'exchange': timer.summary() if timer is not None else None,
            'host': os.uname().nodename, 'local_rank': int(os.environ.get('LOCAL_RANK', '0'))}
    if world == 1:
        return [mine]
    out = [None] * world
    dist.all_gather_object(out, mine)
    return out if rank == 0 else None


def rccl_topology_excerpt(debug_dir, limit=60):
    """NCCL_DEBUG=INFO lines of the ranks' RCCL initialisation that describe the topology it built (rings / trees / channels, transport per
    peer: xGMI P2P vs SHM vs NET), for the JSON line of an N > 1 run; the full logs stay in `debug_dir`."""
    import glob
    import re
    keep = re.compile(r'(Channel \d+|Ring \d+|Trees?|via P2P|via SHM|via NET|xGMI|XGMI|nRanks|comm 0x.* rank|Connected all|NCCL_|RCCL|topology|busId)')
    lines = []
    for f in sorted(glob.glob(os.path.join(debug_dir, '*'))):
        try:
            for ln in open(f, errors='replace'):
                if keep.search(ln):
                    lines.append(os.path.basename(f) + ': ' + ln.strip()[:200])
        except OSError:
            pass
    return {'dir': debug_dir, 'files': len(glob.glob(os.path.join(debug_dir, '*'))), 'lines_matched': len(lines), 'excerpt': lines[:limit]}


def dry_run(args):
    """Launch / rendezvous rehearsal on CPU (gloo): proves that `python bench.py --gpus N` starts N ranks that find each other and walk
    the product's bucketed tower-gradient mean (wavenet_vocoder.parallel) in step.  Prints ONE line marked dry_run; value is null."""
    import torch.distributed as dist
    from wavenet_vocoder.parallel import allreduce_mean_buckets_
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    from wavenet_vocoder import launch as _launch
    _launch.init_process_group(backend='gloo', rank=rank, world_size=world)
    ones = torch.ones(1)
    dist.all_reduce(ones)
    from wavenet_vocoder.parallel import ExchangeTimer
    eng = _DryEngine(1000)
    flat = torch.zeros(eng.n_params)
    dist.barrier()
    t0 = time.time()
    ok = True
    timer = ExchangeTimer()
    for i in range(args.warmup + args.steps):
        grads = torch.full((eng.n_params,), float(rank + 1 + i))
        allreduce_mean_buckets_(eng, grads, timer=timer)
        want = sum(r + 1 + i for r in range(world)) / world               # the tower mean every rank must hold
        ok = ok and bool(torch.allclose(grads, torch.full_like(grads, want)))
        flat -= 0.1 * grads
    dist.barrier()
    dt = time.time() - t0
    per_rank = gather_per_rank(dist, rank, world, dt / (args.warmup + args.steps) * 1e3, timer)
    # the N = 1 leg of the real run, rehearsed: rank 0 alone walks the same loop with the exchange off while the others wait at a barrier
    # (`n1_reference` / `scaling_vs_n1` of the JSON line: weak scaling, so N ranks at the N = 1 step time read N.0)
    n1_wall = None
    if rank == 0:
        t1 = time.time()
        f1 = torch.zeros(eng.n_params)
        for i in range(args.warmup + args.steps):
            f1 -= 0.1 * torch.full((eng.n_params,), float(1 + i))
        n1_wall = max(time.time() - t1, 1e-9)
    dist.barrier()
    cs = flat.sum().reshape(1).double()
    lo, hi = cs.clone(), cs.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({'metric': 'wavenet_train_audio_samples_per_sec', 'value': None, 'unit': 'audio_samples/s', 'dry_run': True,
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': None,
                          'collective': {'backend': 'gloo', 'world_size': dist.get_world_size(), 'ranks_counted_by_allreduce': int(round(ones.item())),
                                         'self_launched': os.environ.get('WN_SELF_LAUNCHED') == '1'},
                          'tower_mean_correct': ok, 'replicas_identical': bool(lo.item() == hi.item()), 'wall_s': dt, 'per_rank': per_rank,
                          'n1_reference': {'what': 'rank 0 alone, same loop, exchange off (rehearsal: CPU tensors, not a measurement)', 'wall_s': n1_wall},
                          'scaling_vs_n1': (world * n1_wall / dt) if n1_wall else None,
                          'config': {'workload_key': args.workload},
                          'note': 'launch rehearsal on CPU: no kernel ran, nothing here is a measurement'}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-synth', action='store_true')
    ap.add_argument('--no-other-workloads', action='store_true', help='skip the 10-step runs of c5_stress / default_hparams appended under other_workloads')
    ap.add_argument('--no-feeder', action='store_true', help='skip the untimed block of steps fed through the on-disk feeder (feeder -> pinned -> H2D)')
    ap.add_argument('--cpu-full-batch', action='store_true', help='also RUN the oracle training step on the whole bench batch on the host cores (64 threads; bounded at 600 s)')
    ap.add_argument('--no-exclusive', action='store_true', help='skip the untimed single-stream pass (keeps a rocprofv3 trace to the timed configuration)')
    ap.add_argument('--sustained', type=int, default=100, help='steps per block of the untimed-by-contract sustained measurement (3 blocks after the timed region; 0 = off)')
    ap.add_argument('--emulate-allreduce-gbps', type=float, default=0.0,
                    help='single-GPU model of the data-parallel exchange: after the backward, every gradient bucket occupies the communication stream for '
                         'bytes / (this many GB/s) (a spin kernel gated on the bucket event, like the RCCL call would be); use with --grad-buckets 1 / 3')
    ap.add_argument('--grad-buckets', type=int, default=None, help='override wn_config.grad_buckets (default: 3 under torch.distributed with > 1 rank, else 1)')
    ap.add_argument('--batch', type=int, default=None, help='override per-GPU batch (debug)')
    ap.add_argument('--time', type=int, default=None, help='override T (debug)')
    ap.add_argument('--force-dist', action='store_true', help='take the multi-rank code path (process group, bucketed RCCL exchange, barriers) even with one rank: '
                    'a single-GPU rehearsal of what torch.distributed.run --nproc-per-node N executes')
    ap.add_argument('--dry-run', action='store_true', help='launch / rendezvous rehearsal without a GPU: the same self-launch, the same bucketed exchange walk over gloo '
                    'with a stand-in engine; prints a line marked dry_run with value null (never a measurement)')
    args = ap.parse_args()

    # `python bench.py --gpus N` with no launcher around it starts the N ranks itself (one process per GPU, the environment
    # torch.distributed.run would set); under a launcher (WORLD_SIZE / RANK present) this process IS one of the ranks.
    from wavenet_vocoder import launch
    if args.gpus > 1 and not launch.launched():
        if not args.dry_run:
            launch.require_gpus(args.gpus)
        _log('no launcher found: starting %d ranks (one per GPU) on 127.0.0.1' % args.gpus)
        raise SystemExit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    if args.dry_run:
        return dry_run(args)

    # stdout carries exactly ONE line (the result): native libraries that write to fd 1 (RCCL's version banner at the first
    # collective, rocm tools) are sent to stderr for the whole run; the JSON line goes to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback exists for the product path)')
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch `python bench.py --gpus %d` (starts its own ranks) or torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus, args.gpus))
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit('rank %d has no GPU: %d visible' % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        from wavenet_vocoder import launch as _launch
        if world > 1 and os.environ.get('WN_BENCH_RCCL_DEBUG', '1') != '0':
            # the first multi-GPU run explains itself: RCCL's own account of the topology it built goes to side files (one per process),
            # an excerpt into the JSON line (collective.rccl_topology)
            rccl_dir = os.environ.get('WN_BENCH_RCCL_DEBUG_DIR') or os.path.join(ROOT, 'bench_rccl_debug_n%d' % world)
            if rank == 0:
                os.makedirs(rccl_dir, exist_ok=True)
            os.environ.setdefault('NCCL_DEBUG', 'INFO')
            os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH,ENV')
            os.environ.setdefault('NCCL_DEBUG_FILE', os.path.join(rccl_dir, 'rccl.%h.%p.log'))
        if world > 1:
            _launch.init_process_group(backend='nccl', device_id=device)
        else:
            _launch.init_process_group(backend='nccl', device_id=device, rank=0, world_size=1)
    collective = None
    if use_dist:
        # count the ranks with a real collective: the number RCCL actually connected, not the number the environment promised
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        collective = {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'ranks_counted_by_allreduce': int(round(ones.item())),
                      'self_launched': os.environ.get('WN_SELF_LAUNCHED') == '1', 'gpus_visible': torch.cuda.device_count()}
        if collective['ranks_counted_by_allreduce'] != args.gpus and not (args.force_dist and world == 1):
            raise SystemExit('all-reduce counted %d ranks, --gpus %d' % (collective['ranks_counted_by_allreduce'], args.gpus))
        _log('process group: %s' % json.dumps(collective))

    from wavenet_vocoder import _ext
    from wavenet_vocoder.models.modules import initialize_parameters
    from wavenet_vocoder.parallel import allreduce_mean_buckets_
    hp, B, T = build_hparams(args.workload)
    B = args.batch or B
    T = args.time or T
    hop = int(np.prod(hp.upsample_scales))
    T = T // hop * hop
    eng = _ext.Engine(hp, B, T, grad_buckets=args.grad_buckets if args.grad_buckets is not None else (3 if args.force_dist and world == 1 else None))
    flat = initialize_parameters(hp, eng.layout).to(device)
    assert flat.numel() == eng.n_params
    if use_dist:
        dist.broadcast(flat, 0)
    grads = torch.zeros_like(flat)
    m, v, ema = torch.zeros_like(flat), torch.zeros_like(flat), flat.clone()
    loss = torch.zeros(1, device=device)
    x, c, y, lengths, _, _ = synthetic_batch(hp, B, T, seed=5339 + rank, device=device)   # disjoint utterances per rank

    emu_stream = torch.cuda.Stream(device=device) if args.emulate_allreduce_gbps > 0 else None
    emu_clock_hz = 100e6      # torch.cuda._sleep counts s_memrealtime-like ticks?  calibrated below
    if emu_stream is not None:
        # calibrate torch.cuda._sleep (cycles -> seconds) with device events, after a warm-up call
        torch.cuda._sleep(1_000_000); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(50_000_000); e1.record(); torch.cuda.synchronize()
        emu_clock_hz = 50_000_000 / (e0.elapsed_time(e1) * 1e-3)
        _log('emulated all-reduce: _sleep runs at %.1f MHz; %.2f ms for the %.1f MB gradient at %.0f GB/s'
             % (emu_clock_hz / 1e6, eng.n_params * 4 / (args.emulate_allreduce_gbps * 1e9) * 1e3, eng.n_params * 4 / 1e6, args.emulate_allreduce_gbps))

    def emulated_allreduce():
        """What allreduce_mean_buckets_ does, with the RCCL call replaced by a spin of bytes / bandwidth on the communication stream."""
        cur = torch.cuda.current_stream(device)
        for i_b, (off, n) in enumerate(eng.grad_buckets()):
            eng.wait_bucket(i_b, emu_stream)
            with torch.cuda.stream(emu_stream):
                torch.cuda._sleep(int(n * 4 / (args.emulate_allreduce_gbps * 1e9) * emu_clock_hz))
        cur.wait_stream(emu_stream)

    def one_step(i, exchange=True, batch=None):
        bx, bc, by, bl = batch if batch is not None else (x, c, y, lengths)
        eng.pack_weights(flat)
        eng.train_fwd(bx, bc, by, bl, 1000 + i, loss)
        eng.train_bwd(grads)
        if emu_stream is not None:
            emulated_allreduce()
        if exchange:
            allreduce_mean_buckets_(eng, grads, single_rank_ok=args.force_dist)          # per gradient bucket on a side stream, under the rest of the backward
        lr = _ext.learning_rate(hp.wavenet_lr_schedule, hp.wavenet_learning_rate, i, hp.wavenet_decay_rate, hp.wavenet_decay_steps, hp.wavenet_warmup)
        eng.optim_step(flat, grads, m, v, ema, lr, i)

    _log('engine + buffers ready; warm-up')
    for i in range(args.warmup):
        one_step(i)
        torch.cuda.synchronize(); _log('warm-up step %d done' % i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    eng.profile(True)
    t0 = time.time()
    for i in range(args.steps):
        one_step(args.warmup + i)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.time() - t0
    _log('timed region done: %.1f ms/step' % (dt / args.steps * 1e3))
    prof_ms, prof_n = eng.profile_result()
    kern_ms, kern_n = eng.profile_kernel_result()
    kern_mhz, kern_mhz_n = eng.profile_kernel_clock()        # mean shader clock inside the timed gate launches (in-kernel cycle counter / wall clock)
    rows_launch = eng.profile_rows_per_launch()
    eng.profile(False)
    # ---- sustained view (>= 20 warm-up, >= 100 timed steps, median of 3): the contract's timed region above may be a
    # fraction of a second (power and clocks have not settled); this block is NOT the headline `value`, it is reported next to it
    sustained = None
    if args.sustained > 0:
        blocks = []
        with SmiSampler(local_rank if rank == 0 else -1) as smi:
            step_i = args.warmup + args.steps
            for _ in range(3):
                torch.cuda.synchronize()
                if use_dist:
                    dist.barrier()
                tb = time.time()
                for _i in range(args.sustained):
                    one_step(step_i); step_i += 1
                torch.cuda.synchronize()
                if use_dist:
                    dist.barrier()
                blocks.append((time.time() - tb) / args.sustained * 1e3)
        med = float(np.median(blocks))
        sm = smi.summary()
        sustained = {'what': '3 blocks x %d steps right after the timed region, same step function; median block' % args.sustained,
                     'ms_per_step_blocks': blocks, 'ms_per_step': med, 'value': world * B * T / (med * 1e-3), 'unit': 'audio_samples/s', 'smi': sm,
                     # energy as a design axis: joules one training step costs at the sustained power draw, and per algorithmic byte / flop
                     'joules_per_step': (sm['power_w_mean'] * med * 1e-3) if sm.get('power_w_mean') else None,
                     'picojoules_per_alg_flop': (sm['power_w_mean'] * med * 1e-3 / (6.0 * mac_per_sample(hp) * B * T) * 1e12) if sm.get('power_w_mean') else None}
        _log('sustained: %s ms/step' % ', '.join('%.2f' % b for b in blocks))
    # untimed extra: the same step fed by the PRODUCT's feeder (wavenet_vocoder/feeder.py, reference feeder.py:266-340): .npy utterances
    # on disk -> background thread (length bucketing, hop-aligned crop, mel normalisation) -> pinned host batch -> non_blocking H2D ->
    # step.  `value` keeps the contract's resident inputs; this says what the input pipeline costs next to it.
    with_feeder = None
    if not args.no_feeder and world == 1 and not args.workload.endswith('_fp32'):
        try:
            with_feeder = measure_with_feeder(hp, B, T, device, one_step, steps=max(args.steps, 30))
            _log('with feeder: %.2f ms/step' % with_feeder['ms_per_step'])
        except Exception as e:          # a reported extra: never lose the JSON line to it
            with_feeder = {'error': str(e)[:300]}
    # untimed extra: the device timeline of ONE step from the engine's in-kernel stamps (no profiler: rocprofv3 slows the host's enqueue
    # enough to change which stream runs ahead) -- forward / backward chain / weight-gradient tail and how long two launches overlap
    device_timeline = None
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import devtrace
        torch.cuda.synchronize()
        eng.trace_arm(6)                              # the 6th step from here: the host is ahead of the device again by then
        for _i in range(9):
            one_step(args.warmup + args.steps + _i)
        device_timeline = devtrace.summarise(eng.trace_read())
        if device_timeline:
            device_timeline['what'] = ('one un-profiled step of the same loop, {first workgroup start, last workgroup end} stamps of every tile-engine / grouped '
                                       'weight-gradient launch on the 100 MHz wall clock (wn_trace_arm / wn_trace_read, tools/devtrace.py)')
            _log('device timeline: forward %.2f ms, backward chain %.2f ms, weight-gradient tail %.2f ms' % tuple((device_timeline[k] or 0) / 1e3 for k in ('forward_us', 'backward_chain_us', 'weight_gradient_tail_us')))
    except Exception as e:          # a reported extra: never lose the JSON line to it
        device_timeline = {'error': str(e)[:200]}
    # untimed extra: host time to ENQUEUE one step (the four C-ABI calls + the exchange, no device wait inside) next to the device time
    # of that step: the margin by which host launch cost hides behind the GPU (what a hipGraph capture of the step could remove)
    host_enqueue = None
    if not args.no_exclusive:          # (--no-exclusive keeps a profiler's view to the timed configuration: no extra steps)
        hs, ds = [], []
        for i in range(10):
            torch.cuda.synchronize()
            th = time.time(); one_step(args.warmup + args.steps + 5000 + i); te = time.time()
            torch.cuda.synchronize(); td = time.time()
            hs.append((te - th) * 1e3); ds.append((td - th) * 1e3)
        host_enqueue = {'what': 'median over 10 untimed steps: host wall time of the enqueue calls / of the whole step, device idle at the start',
                        'host_ms': float(np.median(hs)), 'step_ms': float(np.median(ds))}
        _log('host enqueue %.2f ms of a %.2f ms step' % (host_enqueue['host_ms'], host_enqueue['step_ms']))
    # untimed extra: the same kernel with the GPU to itself (whole batch on one stream), for the kernel-quality view
    excl_ms, excl_n, excl_rows, exk_ms, exk_n, exk_mhz = 0.0, 0, 0, 0.0, 0, 0.0
    if not args.no_exclusive:
        eng.set_batch_parts(1)
        one_step(args.warmup + args.steps)
        eng.profile(True)
        for i in range(2):
            one_step(args.warmup + args.steps + 1 + i)
        excl_ms, excl_n = eng.profile_result()
        exk_ms, exk_n = eng.profile_kernel_result()
        exk_mhz, _ = eng.profile_kernel_clock()
        excl_rows = eng.profile_rows_per_launch()
        eng.profile(False)
        eng.set_batch_parts(0)
    final_loss = float(loss.item())
    # untimed extra under the multi-rank code path: 10 more steps with the exchange instrumented (device events per bucket and at the
    # join), every rank's figures gathered on rank 0
    per_rank = None
    if use_dist:
        from wavenet_vocoder.parallel import ExchangeTimer
        xt = ExchangeTimer()
        torch.cuda.synchronize(); dist.barrier()

        def timed_exchange_step(i):
            eng.pack_weights(flat)
            eng.train_fwd(x, c, y, lengths, 1000 + i, loss)
            eng.train_bwd(grads)
            allreduce_mean_buckets_(eng, grads, single_rank_ok=args.force_dist, timer=xt)
            eng.optim_step(flat, grads, m, v, ema, 1e-4, i)
        for i in range(10):
            timed_exchange_step(70000 + i)
        torch.cuda.synchronize()
        per_rank = gather_per_rank(dist, rank, world, dt / args.steps * 1e3, xt)
        if collective is not None and world > 1 and os.environ.get('NCCL_DEBUG_FILE'):
            dist.barrier()
            if rank == 0:
                collective['rccl_topology'] = rccl_topology_excerpt(os.path.dirname(os.environ['NCCL_DEBUG_FILE']))
    if use_dist:
        tmax = torch.tensor([dt], device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # untimed extra under N > 1 ranks: the same K steps on rank 0 ALONE with the exchange switched off (the other ranks idle at a
    # barrier), i.e. this box's own N = 1 figure next to the N-rank one.  The driver computes scaling from its own N = 1 run; this is
    # the builder-side cross-check that both numbers come from one process on one box.
    n1_reference = None
    if use_dist and world > 1:
        torch.cuda.synchronize(); dist.barrier()
        if rank == 0:
            for i in range(2):
                one_step(90000 + i, exchange=False)
            torch.cuda.synchronize()
            t1 = time.time()
            for i in range(args.steps):
                one_step(90010 + i, exchange=False)
            torch.cuda.synchronize()
            d1 = time.time() - t1
            n1_reference = {'what': 'rank 0 alone, same %d steps, gradient exchange off, other ranks idle at a barrier' % args.steps,
                            'ms_per_step': d1 / args.steps * 1e3, 'value': B * T * args.steps / d1, 'unit': 'audio_samples/s'}
        dist.barrier()

    if rank == 0:
        samples = world * B * T * args.steps
        value = samples / dt
        mac = mac_per_sample(hp)
        # dominant kernel: gate GEMM.  Algorithmic flops per launch = 2 * G * (3R + C) * B*T  (SURVEY 8d per-sample x units/launch)
        R, G, C = hp.residual_channels, hp.gate_channels, hp.cin_channels
        rows_launch = rows_launch or B * T       # the layer chain runs per half-batch on two streams
        flops_launch = 2.0 * G * (3 * R + C) * rows_launch
        # two clocks on the same launches: the kernel's OWN duration from in-kernel stamps (first workgroup start .. last workgroup end;
        # what rocprofv3's kernel trace reports, profiles/*kernel_stats.csv) and the HIP-event bracket on the launch stream, which also
        # counts the wait for CU slots behind the other half-batch's kernels.  `achieved` / `frac` use the kernel's own duration.
        avg_ev_s = (prof_ms / max(prof_n, 1)) * 1e-3
        avg_s = (kern_ms / max(kern_n, 1)) * 1e-3 if kern_n else avg_ev_s
        achieved = flops_launch / avg_s / 1e12 if (kern_n or prof_n) else None
        achieved_ev = flops_launch / avg_ev_s / 1e12 if prof_n else None
        peak = 2500.0
        traffic = load_traffic(args.workload, B, T)
        unprof = sustained['ms_per_step'] if sustained else None
        ex_avg = (exk_ms / max(exk_n, 1)) if exk_n else (excl_ms / max(excl_n, 1))
        res = {
            'metric': 'wavenet_train_audio_samples_per_sec', 'value': value, 'unit': 'audio_samples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'C2 paper_hparams WaveNet 24L/%d-stack R256 G512 S256 10-MoL raw16, 2D upsample [5,5,11], B=%d x T=%d per GPU, dropout %.2f'
                                   % (hp.stacks, B, T, hp.wavenet_dropout) if args.workload.startswith('c2') else
                                   ('C5 Gaussian raw 24 kHz WaveNet %dL/%d-stack R%d G%d S%d, SubPixel %s, legacy scalings, B=%d x T=%d per GPU, dropout %.2f (BASELINE configs[4])'
                                    % (hp.layers, hp.stacks, hp.residual_channels, hp.gate_channels, hp.skip_out_channels, list(hp.upsample_scales), B, T, hp.wavenet_dropout)
                                    if args.workload == 'c5_stress' else args.workload),
                       'workload_key': args.workload, 'global_batch': world * B, 'seq_len': T, 'parallelism': 'dp%d' % world,
                       'layers': hp.layers, 'stacks': hp.stacks, 'params': int(eng.n_params)},
            'samples_per_sec_per_gpu': value / world,
            'train_tflops_algorithmic': 6.0 * mac * value / 1e12,
            'final_loss': final_loss,
            'roofline': {'bound': 'mfma', 'kernel': 'wn_gemm_lds_kernel<2,2,4,2,32,3,EPI_GATE,1,3> (dilated conv + cond GEMM + gate, fwd)',
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': (achieved / peak) if achieved else None,
                         # power as a bound: the step draws ~1.35 kW and the shader clock sits well under the 2400 MHz the 2.5 PFLOP/s peak assumes.  sclk_in_kernel_mhz =
                         # workgroup 0 of every timed launch reads the shader-cycle counter over the 100 MHz wall clock; peak_at_clock = peak x sclk / 2400
                         'sclk_in_kernel_mhz': kern_mhz or None, 'peak_at_clock': (peak * kern_mhz / 2400.0) if kern_mhz else None,
                         'frac_of_peak_at_clock': (achieved / (peak * kern_mhz / 2400.0)) if (achieved and kern_mhz) else None,
                         'frac_incl_queue_wait': (achieved_ev / peak) if achieved_ev else None,
                         'timing': 'achieved / frac: in-kernel start..end stamps of every timed launch (the kernel duration rocprofv3 reports); '
                                   'frac_incl_queue_wait: HIP events around the same launches on their stream (adds the wait behind the other stream)',
                         'traffic': (traffic['gate_bytes_per_launch'] * rows_launch / (B * T / 2.0)) if (traffic and traffic.get('gate_bytes_per_launch')) else None,
                         'traffic_source': (traffic['source'] + ': 2 x FETCH_SIZE + WRITE_SIZE per gate launch of a half batch') if traffic else None,
                         'launches_timed': int(kern_n or prof_n), 'avg_launch_ms': avg_s * 1e3 if (kern_n or prof_n) else None,
                         'avg_launch_ms_event_bracket': avg_ev_s * 1e3 if prof_n else None,
                         'alg_flops_per_launch': flops_launch, 'alg_bytes_per_launch': float(rows_launch) * (2 * R + 2 * C + 2 * G),
                         'rows_per_launch': rows_launch,
                         'profiling_cost': {'what': 'the timed region records 2 events per gate launch and the kernel stamps its start / end; the sustained blocks '
                                                    'below run the same steps without either', 'ms_per_step_profiled': dt / args.steps * 1e3, 'ms_per_step_unprofiled': unprof},
                         'note': 'each launch covers one half-batch and shares the GPU with the HBM-bound out-conv launches of the other half-batch on a second stream'},
            'roofline_exclusive': {'what': 'same kernel, whole batch on one stream (no concurrent kernels), 2 untimed steps after the timed region; in-kernel stamps',
                                   'avg_launch_ms': ex_avg, 'avg_launch_ms_event_bracket': excl_ms / max(excl_n, 1), 'rows_per_launch': excl_rows,
                                   'sclk_in_kernel_mhz': exk_mhz or None, 'peak_at_clock': (peak * exk_mhz / 2400.0) if exk_mhz else None,
                                   'achieved': (2.0 * G * (3 * R + C) * excl_rows / (ex_avg * 1e-3) / 1e12) if excl_n else None,
                                   'frac': (2.0 * G * (3 * R + C) * excl_rows / (ex_avg * 1e-3) / 1e12 / peak) if excl_n else None},
            # whole-step view asked for by the north star: SURVEY 8d algorithmic HBM bytes per audio sample (bf16) x samples/s vs 8 TB/s
            'hbm_roofline_whole_step': {'alg_bytes_per_sample': alg_bytes_per_sample(hp), 'achieved_GBps': alg_bytes_per_sample(hp) * value / world / 1e9,
                                        'peak_GBps': 8000.0, 'frac': alg_bytes_per_sample(hp) * value / world / 8e12,
                                        'alg_bytes_per_step': alg_bytes_per_sample(hp) * B * T,
                                        'traffic_per_step': traffic['bytes_per_step'] if traffic else None,
                                        'traffic_fetch_x2_per_step': traffic['fetch_x2_bytes_per_step'] if traffic else None,
                                        'traffic_write_per_step': traffic['write_bytes_per_step'] if traffic else None,
                                        'traffic_source': (traffic['source'] + ': sum over ALL kernels of 2 x FETCH_SIZE + WRITE_SIZE per step') if traffic else None},
            'mfma_whole_step_frac': 6.0 * mac * value / world / 1e12 / peak,
            'sustained': sustained, 'with_feeder': with_feeder, 'host_enqueue': host_enqueue, 'device_timeline': device_timeline,
            'grad_buckets': [list(b) for b in eng.grad_buckets()], 'force_dist': bool(args.force_dist),
            'collective': collective, 'per_rank': per_rank, 'n1_reference': n1_reference,
            'scaling_vs_n1': (value / n1_reference['value']) if n1_reference else None,      # speed-up factor over this box's own 1-rank run
            'emulated_allreduce': ({'gbps': args.emulate_allreduce_gbps, 'bytes': int(eng.n_params) * 4,
                                    'serial_ms': int(eng.n_params) * 4 / (args.emulate_allreduce_gbps * 1e9) * 1e3,
                                    'what': 'single-GPU model: each gradient bucket occupies the communication stream for bytes / bandwidth once its event fired'}
                                   if args.emulate_allreduce_gbps > 0 else None),
        }
        if not args.no_synth and world == 1:      # replicas-only path: measured on one GPU, not while the other ranks wait
            _log('synthesis measurement ...')
            try:
                res['synthesis_parity_gate'] = synthesis_parity_gate(hp, flat, device)
                _log('synthesis parity gate: %s' % json.dumps({k: res['synthesis_parity_gate'].get(k) for k in ('ok', 'rel_l2_vs_fp32_mode_worst_stream', 'streams', 'steps')}))
            except Exception as e:
                res['synthesis_parity_gate'] = {'ok': None, 'error': str(e)[:300]}
            try:
                res['synthesis'] = measure_synthesis(hp, flat, device)
            except Exception as e:          # never lose the training number to a synthesis problem
                res['synthesis'] = {'error': str(e)[:300]}
        if world == 1 and not args.no_other_workloads and args.workload == 'c2':
            # In a FRESH process each: streams are multiplexed onto a few hardware queues in creation order, and by now this process has
            # created and destroyed a dozen (engine side streams, synthesis contexts); an engine created here can get its two part
            # streams on ONE queue and lose their overlap (measured: hparams.py defaults 6.46 ms/step in-process vs 3.77 in a new
            # process, profiles/r4f_bench_default_flags.json vs r4f_other_workloads.json).  The headline engine is closed first.
            eng.close()
            res['other_workloads'] = {}
            for key in ('default_hparams', 'c5_stress', 'c2_fp32'):
                _log('other workload %s ...' % key)
                res['other_workloads'][key] = other_workload_subprocess(key, local_rank)
        if world == 1 and not args.no_cpu_baseline:
            _log('cpu baseline (oracle) ...')
            res['cpu_baseline'] = cpu_baseline_subprocess(args.workload)
            if args.cpu_full_batch:          # measured here or absent: never a number quoted from a file
                _log('cpu baseline on the full batch (oracle, ~1 min on 256 cores) ...')
                res['cpu_baseline']['full_batch'] = cpu_baseline_subprocess(args.workload, hard_timeout=600, fn='cpu_full_batch')
            if not args.no_synth:
                _log('cpu baseline, synthesis (oracle incremental loop) ...')
                try:
                    res['cpu_baseline_synthesis'] = cpu_baseline_subprocess(args.workload, fn='cpu_synth_baseline')
                except Exception as e:      # a reported extra: never lose the JSON line to it
                    res['cpu_baseline_synthesis'] = {'value': None, 'sample': 'failed: ' + str(e)[:200]}
        else:
            res['cpu_baseline'] = None
        res['synthesis_summary'] = synthesis_summary(res)      # LAST key on purpose
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(res) + '\n').encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
