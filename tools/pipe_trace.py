import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tacotron-2_amd')
import torch, numpy as np, bench
from wavenet_vocoder import _ext
from wavenet_vocoder.models.modules import initialize_parameters
hp, _, _ = bench.build_hparams('c2')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
hop = 275; Tc = 8; T = Tc * hop
eng = _ext.Engine(hp, B, T)
flat = initialize_parameters(hp, eng.layout).cuda()
eng.pack_weights(flat)
c = torch.rand(B, 80, Tc, device='cuda'); noise = torch.rand(T, B, eng.noise_per_step, device='cuda') * 0.98 + 0.01
out = torch.empty(B, T, device='cuda')
for i in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    eng.synthesize(c, noise, out, None, None, steps_per_graph=0)
    torch.cuda.synchronize(); print('B=%d T=%d: %.2f us/step' % (B, T, (time.time() - t0) / T * 1e6))
