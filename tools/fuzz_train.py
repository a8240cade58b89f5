#!/usr/bin/env python3
"""Ad-hoc sweep of hparams combinations the committed parity configurations do not pair (run on the GPU box): forward, loss and all gradients vs the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tacotron-2_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
import test_hip_parity as TP
from hip_util import download_grads, rel_err
from oracle import wavenet_oracle as O

EXTRA = {
    'one_upsample_layer': dict(upsample_scales=[16]),
    'three_upsample_layers': dict(upsample_scales=[2, 2, 4]),
    'four_upsample_layers_subpixel': dict(upsample_scales=[2, 2, 2, 2], upsample_type='SubPixel', out_channels=2, log_scale_min_gauss=float(np.log(1e-7))),
    'freq_kernel_5': dict(freq_axis_kernel_size=5),
    'freq_kernel_1_leaky': dict(freq_axis_kernel_size=1, upsample_activation='LeakyRelu', leaky_alpha=0.1),
    'cin32_resize': dict(cin_channels=32, num_mels=32, upsample_type='Resize', upsample_scales=[4, 4]),
    'no_upsample_activation_1d': dict(upsample_type='1D', upsample_activation='None'),
    'layers12_stacks4_drop30': dict(layers=12, stacks=4, wavenet_dropout=0.3),
    'layers5_stacks5': dict(layers=5, stacks=5),
    'mulaw_input_scalar': dict(input_type='mulaw'),
    'mol_3_components': dict(out_channels=9),
    'softmax_legacy_gin': dict(input_type='mulaw-quantize', out_channels=256, quantize_channels=256, legacy=True, residual_legacy=True, gin_channels=8, use_speaker_embedding=True, n_speakers=2),
    'wide_skip_narrow_res': dict(residual_channels=64, gate_channels=256, skip_out_channels=128),
    'r128_g512_s256_wnorm': dict(residual_channels=128, gate_channels=512, skip_out_channels=256, wavenet_weight_normalization=True, layers=4),
    'r256_g512_s128': dict(residual_channels=256, gate_channels=512, skip_out_channels=128, cin_channels=80, num_mels=80, layers=4),
    'r384_g768_s384': dict(residual_channels=384, gate_channels=768, skip_out_channels=384, layers=3, stacks=1),
}
TP.CONFIGS.update(EXTRA)
bad = 0
for name in EXTRA:
    try:
        r = TP._run_fwd(name, B=3, T=400, lengths=None)
        cfg, eng = r['cfg'], r['eng']
        y_em = O.step(r['params'], cfg, r['x_or'], r['c'], dropout_masks=r['masks'], emulate_bf16=True, g=r['g'])
        ey = rel_err(r['yhat_dev'].cpu(), y_em)
        loss_same = float(O.training_loss(cfg, r['yhat_dev'].cpu(), r['y_or'], r['lengths'])); ld = float(r['loss_dev'].item())
        grads_dev = torch.empty(eng.n_params, device='cuda'); eng.train_bwd(grads_dev); torch.cuda.synchronize()
        g_dev = download_grads(eng, grads_dev)
        leaf = {k: v.clone().requires_grad_(True) for k, v in r['params'].items()}
        y = O.step(leaf, cfg, r['x_or'], r['c'], dropout_masks=r['masks'], emulate_bf16=True, g=r['g'])
        gs = torch.autograd.grad(O.training_loss(cfg, y, r['y_or'], r['lengths']), list(leaf.values()), allow_unused=True)
        g_or = {k: (g if g is not None else torch.zeros_like(leaf[k])) for k, g in zip(leaf, gs)}
        eg = rel_err(torch.cat([g_dev[k].flatten() for k in g_or]), torch.cat([g_or[k].flatten() for k in g_or]))
        worst = max(((float((g_dev[k] - g_or[k]).norm()) / (float(g_or[k].norm()) + 1e-12), k) for k in g_or if float(g_or[k].norm()) > 1e-6))
        soft = cfg.input_type == 'mulaw-quantize'
        ok = ey < 2.4e-2 and abs(ld - loss_same) <= 2e-4 * max(1.0, abs(loss_same)) and eg < (4.5e-2 if soft else 7e-3)
        print('%-32s y_hat %.2e  loss dev %.5f vs %.5f  grads %.2e  worst tensor %.2e (%s)%s' % (name, ey, ld, loss_same, eg, worst[0], worst[1][-50:], '' if ok else '   <-- FAIL'), flush=True)
        eng.close()
    except Exception as ex:      # noqa: BLE001
        ok = False
        print('%-32s EXCEPTION %s   <-- FAIL' % (name, str(ex)[:200]), flush=True)
    bad += not ok
print('failures:', bad)
