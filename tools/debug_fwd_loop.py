import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tacotron-2_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch, numpy as np
import test_hip_parity as P
from oracle import wavenet_oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else 'mol_2d'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ref = None
for it in range(N):
    r = P._run_fwd(name)
    cfg, eng, B, T = r['cfg'], r['eng'], r['B'], r['T']
    if ref is None:
        ref = O.step(r['params'], cfg, r['x_or'], r['c'], dropout_masks=r['masks'], emulate_bf16=True, return_aux=True)
    y_em, aux = ref
    rows = B * T
    errs = []
    for l in range(cfg.layers):
        U = eng.debug_copy('U', l, rows, cfg.gate_channels // 2).cpu().view(B, T, -1).permute(0, 2, 1)
        errs.append(P.rel_err(U, aux['u'][l]))
    yh = r['yhat_dev'].cpu()
    print('iter %d: U errs %s yhat %.2e' % (it, ' '.join('%.1e' % e for e in errs), P.rel_err(yh, y_em)), flush=True)
    if max(errs) > 2e-2:
        l = int(np.argmax(errs))
        U = eng.debug_copy('U', l, rows, cfg.gate_channels // 2).cpu().view(B, T, -1).permute(0, 2, 1)
        d = (U - aux['u'][l]).abs(); idx = (d > 0.02).nonzero()
        print('  layer', l, 'bad count', idx.shape[0], 'of', d.numel(), 'b', torch.unique(idx[:, 0]).tolist())
        ch = torch.unique(idx[:, 1]); print('  bad channels (%d):' % len(ch), ch.tolist())
        tt = torch.unique(idx[:, 2]); print('  bad t (%d): min %d max %d' % (len(tt), tt.min(), tt.max()), tt.tolist()[:48])
    eng.close()
