"""Parameter construction for the WaveNet stack.

In the reference this file holds the TF layer classes (CausalConv1D, Conv1D1x1, ResidualConv1DGLU, the
upsamplers, the masked losses: modules.py:184-836).  Here the arithmetic of all of them lives in the HIP
library (csrc/); what remains on the host is *which tensors exist and how they start*:

  * conv kernels: Glorot-uniform, biases zero -- tf.layers' defaults when kernel_initializer=None
    (reference modules.py:195-196, 206-224);
  * upsample kernels: the checkerboard-free "nearest-neighbour" initialisation when hparams.NN_init
    (reference modules.py:642-654, 685-695, 724-733, 761-770), scaled by NN_scaler**(1/n_layers).
"""
import math

import numpy as np
import torch


def receptive_field_size(total_layers, num_cycles, kernel_size, dilation=lambda x: 2 ** x):
    assert total_layers % num_cycles == 0
    layers_per_cycle = total_layers // num_cycles
    return (kernel_size - 1) * sum(dilation(i % layers_per_cycle) for i in range(total_layers)) + 1


def _glorot_uniform(shape, gen, transposed=False):
    """fan_in / fan_out as TF computes them for conv kernels: receptive field x channels."""
    if len(shape) < 2:
        raise ValueError(shape)
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    a, b = shape[-2] * rf, shape[-1] * rf
    fan_in, fan_out = (b, a) if transposed else (a, b)     # Conv2DTranspose kernels are [kh,kw,out,in]
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen) * 2.0 - 1.0) * limit


def nn_upsample_kernel(upsample_type, shape, stride, scale):
    """Initial kernel making the layer a (scaled) nearest-neighbour upsampler."""
    k = np.zeros(shape, dtype=np.float32)
    if upsample_type == '2D':                       # [fk, s, 1, 1]: centre frequency row, every phase
        fk, s = shape[0], shape[1]
        k[fk // 2, :, 0, 0] = 1.0 / max(s // stride, 1) if s % 2 == 0 else 1.0
    elif upsample_type == 'Resize':                 # [fk, s, 1, 1]: centre tap(s)
        fk, s = shape[0], shape[1]
        taps = [s // 2 - 1, s // 2] if s % 2 == 0 else [s // 2]
        for j in taps:
            k[fk // 2, j, 0, 0] = 1.0 / max(s // stride, 1) if s % 2 == 0 else 1.0
    elif upsample_type == 'SubPixel':               # [fk, 3, 1, s]: centre tap replicated over the s sub-pixel filters
        fk = shape[0]
        k[fk // 2, 1, 0, :] = 1.0
    elif upsample_type == '1D':                     # [1, s, C, C]: identity over channels for every phase
        s, C = shape[1], shape[2]
        eye = np.eye(C, dtype=np.float32)
        for j in range(s):
            k[0, j] = eye
    else:
        raise ValueError(upsample_type)
    return torch.from_numpy(k * scale)


def initialize_parameters(hparams, layout, seed=None):
    """layout: OrderedDict name -> (shape, offset) from the engine.  Returns a flat fp32 CPU tensor."""
    gen = torch.Generator().manual_seed(hparams.wavenet_random_seed if seed is None else seed)
    total = max(off + int(np.prod(shape)) for shape, off in layout.values())
    total = (total + 7) // 8 * 8
    flat = torch.zeros(total, dtype=torch.float32)
    n_up = len(hparams.upsample_scales)
    for name, (shape, off) in layout.items():
        n = int(np.prod(shape))
        if name.endswith('/bias') or name.endswith('/g'):
            continue
        if name == 'gc_embedding':                           # tf.truncated_normal_initializer(0, 0.1) (reference modules.py:13-17)
            t = torch.fmod(torch.randn(shape, generator=gen), 2.0) * 0.1
        elif name.startswith('local_conditioning_upsampling_'):
            i = int(name.split('/')[0].rsplit('_', 1)[1]) - 1
            if hparams.NN_init:
                t = nn_upsample_kernel(hparams.upsample_type, shape, hparams.upsample_scales[i],
                                       float(hparams.NN_scaler) ** (1.0 / n_up))
            else:
                t = _glorot_uniform(shape, gen, transposed=hparams.upsample_type in ('1D', '2D'))
                if hparams.upsample_type == 'SubPixel':      # all sub-pixel filters start equal (ICNR), modules.py:584-592
                    t = t[..., :1].expand(*shape).clone()
        else:
            t = _glorot_uniform(shape, gen)
        flat[off:off + n] = t.reshape(-1)
    # weight normalisation gains: g = ||v|| over all axes but the last, so that g * v / ||v|| == v at the start
    # (WeightNorm.build assigns _init_norm(v) to g: reference modules.py:104-108, 161-171)
    for name, (shape, off) in layout.items():
        if name.endswith('/g'):
            kshape, koff = layout[name[:-1] + 'kernel']
            v = flat[koff:koff + int(np.prod(kshape))].reshape(-1, kshape[-1])
            flat[off:off + shape[0]] = v.norm(dim=0)
    return flat
