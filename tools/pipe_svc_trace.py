#!/usr/bin/env python3
"""Where a layer CU's service time per stream goes (WN_PIPE_SVC_TRACE=1; stderr) at a few batch sizes, C2 model.  Needs the diagnostic build of the
library (the stamp sites are compiled out of the product: they cost SGPRs):
    python tacotron-2_amd/csrc/build.py --pipe-svc && python tools/pipe_svc_trace.py [B ...] ; python tacotron-2_amd/csrc/build.py --force"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tacotron-2_amd')):
    sys.path.insert(0, p)
os.environ['WN_PIPE_SVC_TRACE'] = '1'
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from wavenet_vocoder import _ext  # noqa: E402
from wavenet_vocoder.models.modules import initialize_parameters  # noqa: E402

hp, _, _ = bench.build_hparams('c2')
hop = int(np.prod(hp.upsample_scales))
Tc = 16; T = Tc * hop
dev = torch.device('cuda', 0)
for B in [int(x) for x in sys.argv[1:]] or [8, 20]:
    eng = _ext.Engine(hp, B, T, inference_only=True)
    eng.pack_weights(initialize_parameters(hp, eng.layout).to(dev))
    c = torch.rand(B, hp.cin_channels, Tc, device=dev); out = torch.empty(B, T, device=dev)
    eng.synthesize(c, None, out, None, None, steps_per_graph=0, seed=1); torch.cuda.synchronize(); eng.synth_check()
    t0 = time.time()
    eng.synthesize(c, None, out, None, None, steps_per_graph=0, seed=2); torch.cuda.synchronize(); dt = time.time() - t0
    eng.synth_check()
    print('B=%d: %.1f us per sample' % (B, dt / T * 1e6), flush=True)
    eng.close()
