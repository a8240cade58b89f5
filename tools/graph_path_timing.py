#!/usr/bin/env python3
"""Per-sample time of the launch-per-layer hipGraph synthesis path (wn_synth.hip) for the paper model and at C5's width (the model the
LDS-resident pipeline cannot hold), B = 1 / 8.   python tools/graph_path_timing.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tacotron-2_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from wavenet_vocoder import _ext  # noqa: E402
from wavenet_vocoder.models.modules import initialize_parameters  # noqa: E402

dev = torch.device('cuda', 0)
out = {}
for key in ('c2', 'c5_stress'):
    hp, _, _ = bench.build_hparams(key)
    eng = _ext.Engine(hp, 1, 11000, inference_only=True)
    flat = initialize_parameters(hp, eng.layout).to(dev)
    eng.close()
    r = bench.measure_synthesis(hp, flat, dev, seconds=0.25, batches=(1, 8), modes=('graph',))
    out[key] = {k: {kk: v[kk] for kk in ('us_per_step', 'rtf_per_stream', 'path', 'finite')} for k, v in r.items()}
    print(key, ' '.join('%s: %.1f us per sample (RTF %.2f)' % (k, v['us_per_step'], v['rtf_per_stream']) for k, v in out[key].items()), flush=True)
print(json.dumps(out))
