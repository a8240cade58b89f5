import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tacotron-2_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch, numpy as np
import test_hip_parity as P
from oracle import wavenet_oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else 'mol_2d'
r = P._run_fwd(name)
cfg, eng, B, T = r['cfg'], r['eng'], r['B'], r['T']
y_em, aux = O.step(r['params'], cfg, r['x_or'], r['c'], dropout_masks=r['masks'], emulate_bf16=True, return_aux=True)
rows = B * T
for l in range(cfg.layers):
    U = eng.debug_copy('U', l, rows, cfg.gate_channels // 2).cpu().view(B, T, -1).permute(0, 2, 1)
    ref = aux['u'][l]
    d = (U - ref).abs()
    print('U%d rel %.3e  max abs %.3e' % (l, P.rel_err(U, ref), d.max().item()))
    bad = (d > 0.02)
    if bad.any():
        idx = bad.nonzero()
        print('  bad count', idx.shape[0], 'of', d.numel())
        print('  bad b:', torch.unique(idx[:, 0]).tolist())
        ch = torch.unique(idx[:, 1]); print('  bad channels (%d):' % len(ch), ch.tolist()[:64])
        tt = torch.unique(idx[:, 2]); print('  bad t (%d): min %d max %d' % (len(tt), tt.min(), tt.max()), tt.tolist()[:40])
        i = idx[0]; print('  example', i.tolist(), U[i[0], i[1], i[2]].item(), ref[i[0], i[1], i[2]].item())
yh = r['yhat_dev'].cpu()
d = (yh - y_em).abs(); print('yhat rel %.3e max %.3e' % (P.rel_err(yh, y_em), d.max().item()))
bad = (d > 0.05).nonzero()
if len(bad):
    print(' yhat bad count', len(bad), 'channels', torch.unique(bad[:, 1]).tolist(), 't range', bad[:, 2].min().item(), bad[:, 2].max().item())
