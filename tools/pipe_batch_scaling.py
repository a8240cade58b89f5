#!/usr/bin/env python3
"""Per-sample time of the persistent synthesis pipeline against the number of streams in ONE run (wavenet.py:237-239 splits a batch
over towers; here the streams of a run follow each other through the layer ring): up to which batch does a run cost the wall time of
one stream, and what does hparams.py's wavenet_synthesis_batch_size = 20 cost?   python tools/pipe_batch_scaling.py [seconds [workload[,workload] [B,B,...]]]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tacotron-2_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from wavenet_vocoder import _ext  # noqa: E402
from wavenet_vocoder.models.modules import initialize_parameters  # noqa: E402


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    dev = torch.device('cuda', 0)
    out = {}
    keys = sys.argv[2].split(',') if len(sys.argv) > 2 else ('default_hparams', 'c2')
    Bs = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (1, 4, 8, 10, 12, 16, 20, 24, 32)
    for key in keys:
        hp, _, _ = bench.build_hparams(key)
        hop = int(np.prod(hp.upsample_scales))
        Tc = max(2, int(round(secs * hp.sample_rate / hop))); T = Tc * hop
        rows = {}
        flat = None
        for B in Bs:
            eng = _ext.Engine(hp, B, T, inference_only=True)
            if flat is None:
                flat = initialize_parameters(hp, eng.layout).to(dev)
            eng.pack_weights(flat)
            if not eng.pipeline_eligible(B):
                rows[B] = None; eng.close(); continue
            c = torch.rand(B, hp.cin_channels, Tc, device=dev)
            samples = torch.empty(B, T, device=dev)
            eng.synthesize(c[:, :, :8].contiguous(), None, torch.empty(B, 8 * hop, device=dev), None, None, steps_per_graph=0, seed=1)
            torch.cuda.synchronize()
            try:
                eng.synth_check()
            except Exception as e:      # noqa: BLE001
                print('%s B=%d (warm-up): %s' % (key, B, e), file=sys.stderr, flush=True)
                rows[B] = None; eng.close(); continue
            t0 = time.time()
            eng.synthesize(c, None, samples, None, None, steps_per_graph=0, seed=2)
            torch.cuda.synchronize(); dt = time.time() - t0
            try:
                eng.synth_check()
            except Exception as e:      # noqa: BLE001
                print('%s B=%d: %s' % (key, B, e), file=sys.stderr, flush=True)
                rows[B] = None; eng.close(); continue
            rows[B] = {'instances': int(eng.lib.wn_synth_last_instances(eng.h)), 'batched_premultiplication': int(eng.lib.wn_synth_last_batched(eng.h)), 'us_per_step': dt / T * 1e6, 'rtf_per_stream': dt / (T / hp.sample_rate), 'aggregate_samples_per_s': B * T / dt, 'finite': bool(torch.isfinite(samples).all())}
            eng.close()
        out[key] = rows
        print(key, {b: (r and (r['instances'], round(r['us_per_step'], 1))) for b, r in rows.items()}, file=sys.stderr, flush=True)
        base = rows[min(Bs, key=lambda b: abs(b - 8))]['us_per_step']
        print('%s (R = %d, %d layers): ' % (key, hp.residual_channels, hp.layers) + '  '.join('B=%d: %.1f us (%.2fx)' % (b, r['us_per_step'], r['us_per_step'] / base) if r else 'B=%d: -' % b for b, r in rows.items()))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
