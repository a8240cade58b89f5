#!/usr/bin/env python3
"""Ad-hoc sweep of synthesis geometries (streams x frames) through every path against the oracle's incremental loop (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tacotron-2_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from hip_util import rel_err
from oracle import wavenet_oracle as O
from test_hip_synth import _noise, _setup

bad = 0
for kw in (dict(), dict(out_channels=2, legacy=True, residual_legacy=True, upsample_type='SubPixel'), dict(residual_channels=128, gate_channels=256, skip_out_channels=128, layers=6, stacks=3)):
    for B in (1, 2, 7, 9, 13, 31):
        for Tc in (1, 2, 5):
            hp, cfg, eng, params, wav, c, T = _setup(B, Tc, **kw)
            nz_dev, nz_or = _noise(cfg, T, B)
            with torch.no_grad():
                _, r_or = O.incremental(params, cfg, c, noise=nz_or, test_inputs=wav.unsqueeze(-1), formulation='ring')
            for spg, name in ((0, 'pipe'), (1, 'eager'), (3, 'graph3')):
                if spg == 0 and not eng.pipeline_eligible(B):
                    continue
                out = torch.empty(B, T, device='cuda'); raw = torch.empty(B, cfg.out_channels, T, device='cuda')
                try:
                    eng.synthesize(c.cuda(), nz_dev.cuda(), out, raw, wav.contiguous().cuda(), steps_per_graph=spg)
                    torch.cuda.synchronize(); eng.synth_check()
                    e = rel_err(raw.cpu(), r_or)
                    ok = e < (4e-3 if spg == 0 else 1.4e-2)
                except Exception as ex:      # noqa: BLE001
                    e, ok = str(ex)[:80], False
                if not ok:
                    bad += 1
                print('%s B=%2d Tc=%d %-6s path=%-9s %s %s' % (sorted(kw.items())[:2], B, Tc, name, eng.synth_path, e, '' if ok else '  <-- FAIL'), flush=True)
            eng.close()
print('failures:', bad)
