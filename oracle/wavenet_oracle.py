"""WaveNet-vocoder hot path, torch-CPU fp32 restatement (TEST INFRASTRUCTURE ONLY).

Restates the arithmetic of the reference (all paths relative to
/root/reference/wavenet_vocoder/):
  models/wavenet.py   WaveNet.__init__ :89-208, step :650-721, incremental :724-911,
                      add_loss :476-519, add_optimizer :522-629, get_mask :632-638
  models/modules.py   CausalConv1D :184-333, Conv1D1x1 :336-389, ResidualConv1DGLU :392-521,
                      upsamplers :524-770, masked losses :781-836
  models/mixture.py   :5-107        models/gaussian.py :5-52
Parameters are kept in the reference's TensorFlow layouts ([k, in, out] conv kernels,
[kh, kw, out, in] transposed-conv kernels, [kh, kw, in, out] conv2d kernels) under
names that mirror the reference's variable scopes.  Randomness (dropout masks,
sampling noise) is always an explicit input so the HIP path can be compared on
identical draws.

``emulate_bf16=True`` rounds to bfloat16 at exactly the points where the HIP path
stores bf16 (weights, layer inputs, conditioning, gate outputs, head hidden
activations); the contraction itself stays fp32 like an MFMA with fp32 accumulate.
That variant is used for tight kernel-logic checks; the plain fp32 variant is the
reference arithmetic.

Pinned (see oracle/__init__.py): step / upsample / training_loss / incremental reproduce, to fp32 round-off, the outputs
of the reference's own wavenet.py + modules.py executed on the eager TF-1 stand-in (tests/golden/stack_*.npz); train_step
reproduces three steps of the reference's own add_loss + add_optimizer executed there (tests/golden/optim_golden.npz: gradient
set, tower mean, per-variable clip_by_norm -> clip_by_value, Adam on the clipped gradients, EMA after the update, LR schedule).
The arithmetic INSIDE TensorFlow's library calls (ApplyAdam, clip_by_norm, assign_moving_average, exponential_decay) is restated
from TF 1.x's documented formulas (oracle/gen_golden_optim.py lists them) -- that part stays parity unpinned: TF is not installable.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F

SQRT_HALF = float(np.sqrt(0.5))  # modules.py:518, wavenet.py:715 (np.sqrt(0.5) folded to fp32)


@dataclass
class OracleConfig:
    layers: int = 24
    stacks: int = 4
    residual_channels: int = 256
    gate_channels: int = 512
    skip_out_channels: int = 256
    out_channels: int = 30
    kernel_size: int = 3
    cin_channels: int = 80
    input_type: str = 'raw'                # raw | mulaw | mulaw-quantize
    quantize_channels: int = 65536
    use_bias: bool = True
    legacy: bool = False
    residual_legacy: bool = False
    log_scale_min: float = float(np.log(1e-14))
    log_scale_min_gauss: float = float(np.log(1e-7))
    cdf_loss: bool = False
    upsample_type: str = '2D'              # 1D | 2D | Resize | SubPixel | NearestNeighbor
    upsample_activation: Optional[str] = 'Relu'   # Relu | LeakyRelu | None
    upsample_scales: List[int] = field(default_factory=lambda: [5, 5, 11])
    freq_axis_kernel_size: int = 3
    leaky_alpha: float = 0.4
    NN_init: bool = True
    NN_scaler: float = 0.1
    wavenet_dropout: float = 0.05
    gin_channels: int = -1                 # hparams.py:228: <= 0 disables global conditioning
    use_speaker_embedding: bool = True     # hparams.py:229
    n_speakers: int = 5                    # hparams.py:230
    wavenet_weight_normalization: bool = False   # hparams.py:323

    @property
    def scalar_input(self):
        return self.input_type in ('raw', 'mulaw')

    @property
    def in_channels(self):
        return 1 if self.scalar_input else self.quantize_channels

    @property
    def hop(self):
        return int(np.prod(self.upsample_scales))

    def dilations(self):
        per = self.layers // self.stacks
        return [2 ** (l % per) for l in range(self.layers)]   # wavenet.py:125

    @staticmethod
    def from_hparams(hp):
        kw = {}
        for f in OracleConfig.__dataclass_fields__:
            if hasattr(hp, f):
                kw[f] = getattr(hp, f)
        kw['upsample_scales'] = list(kw.get('upsample_scales', [5, 5, 11]))
        if kw.get('upsample_activation') == 'None':      # `--hparams upsample_activation=None` arrives as a string (the product's parser accepts that spelling)
            kw['upsample_activation'] = None
        return OracleConfig(**kw)


def receptive_field_size(total_layers, num_cycles, kernel_size):
    """wavenet.py:54-71"""
    assert total_layers % num_cycles == 0
    per = total_layers // num_cycles
    return (kernel_size - 1) * sum(2 ** (i % per) for i in range(total_layers)) + 1


# ----------------------------------------------------------------------------- parameters
def param_shapes(cfg: OracleConfig):
    """name -> shape in the reference's TF layout, in creation order."""
    R, G, S, O, C = (cfg.residual_channels, cfg.gate_channels, cfg.skip_out_channels,
                     cfg.out_channels, cfg.cin_channels)
    k = cfg.kernel_size
    sh = OrderedDict()
    sh['input_convolution/kernel'] = (1, cfg.in_channels, R)
    sh['input_convolution/bias'] = (R,)
    for l in range(cfg.layers):
        p = 'ResidualConv1DGLU_%d/' % l
        ub = cfg.use_bias                     # hparams.py:189: only the convolutions inside ResidualConv1DGLU (modules.py:399-446)
        sh[p + 'residual_block_causal_conv/kernel'] = (k, R, G)
        if ub:
            sh[p + 'residual_block_causal_conv/bias'] = (G,)
        if C > 0:
            sh[p + 'residual_block_cin_conv/kernel'] = (1, C, G)
            if ub:
                sh[p + 'residual_block_cin_conv/bias'] = (G,)
        if cfg.gin_channels > 0:              # modules.py:426-432
            sh[p + 'residual_block_gin_conv/kernel'] = (1, cfg.gin_channels, G)
            if ub:
                sh[p + 'residual_block_gin_conv/bias'] = (G,)
        sh[p + 'residual_block_skip_conv/kernel'] = (1, G // 2, S)
        if ub:
            sh[p + 'residual_block_skip_conv/bias'] = (S,)
        sh[p + 'residual_block_out_conv/kernel'] = (1, G // 2, R)
        if ub:
            sh[p + 'residual_block_out_conv/bias'] = (R,)
    sh['final_convolution_1/kernel'] = (1, S, S)
    sh['final_convolution_1/bias'] = (S,)
    sh['final_convolution_2/kernel'] = (1, S, O)
    sh['final_convolution_2/bias'] = (O,)
    if cfg.gin_channels > 0 and cfg.use_speaker_embedding:          # modules.py:13-17, wavenet.py:152-158
        sh['gc_embedding'] = (cfg.n_speakers, cfg.gin_channels)
    if C > 0 and cfg.upsample_type != 'NearestNeighbor':
        fk = cfg.freq_axis_kernel_size
        for i, s in enumerate(cfg.upsample_scales):
            p = 'local_conditioning_upsampling_%d/' % (i + 1)
            if cfg.upsample_type == '2D':          # modules.py:736 Conv2DTranspose [kh,kw,out,in]
                sh[p + 'kernel'] = (fk, s, 1, 1); sh[p + 'bias'] = (1,)
            elif cfg.upsample_type == '1D':        # modules.py:697 Conv2DTranspose [1,s,out,in]
                sh[p + 'kernel'] = (1, s, C, C); sh[p + 'bias'] = (C,)
            elif cfg.upsample_type == 'Resize':    # modules.py:657 Conv2D [kh,kw,in,out]
                sh[p + 'kernel'] = (fk, s, 1, 1); sh[p + 'bias'] = (1,)
            elif cfg.upsample_type == 'SubPixel':  # modules.py:539 Conv2D [kh,3,in=1,out=s]
                sh[p + 'kernel'] = (fk, 3, 1, s); sh[p + 'bias'] = (s,)
            else:
                raise ValueError(cfg.upsample_type)
    return weightnorm_param_shapes(cfg, sh)


def weightnorm_param_shapes(cfg, sh):
    """modules.py:152-166: every wrapped convolution gets g [kernel.shape[-1]] next to its kernel (= v)."""
    if not cfg.wavenet_weight_normalization:
        return sh
    out = OrderedDict()
    for k, v in sh.items():
        out[k] = v
        if k.endswith('/kernel'):
            out[k[:-6] + 'g'] = (v[-1],)
    return out


def effective_params(params, cfg):
    """kernel = l2_normalize(v, all axes but the last) * g   (WeightNorm._compute_weights, modules.py:98-103)."""
    if not cfg.wavenet_weight_normalization or not any(k.endswith('/g') for k in params):
        return params                                     # (already effective)
    out = OrderedDict()
    for k, v in params.items():
        if k.endswith('/g'):
            continue
        if k.endswith('/kernel'):
            g = params[k[:-6] + 'g']
            axes = tuple(range(v.dim() - 1))
            v = v * torch.rsqrt(torch.clamp((v * v).sum(dim=axes, keepdim=True), min=1e-12)) * g
        out[k] = v
    return out


def _nn_init_kernel(cfg, i, s):
    """NN-init upsample kernels: modules.py:642-654 (SubPixel), :685-695 (Resize),
    :724-733 (1D), :761-770 (2D)."""
    n = len(cfg.upsample_scales)
    scale = float(cfg.NN_scaler) ** (1.0 / n)
    fk = cfg.freq_axis_kernel_size
    t = cfg.upsample_type
    if t == '2D':
        ks = (fk, s)
        overlap = ks[1] // s
        kern = np.zeros(ks, np.float32)
        kern[ks[0] // 2, :] = (1.0 / max(overlap, 1.0)) if ks[1] % 2 == 0 else 1.0
        return (kern * scale).reshape(fk, s, 1, 1)
    if t == 'Resize':
        ks = (fk, s)
        overlap = ks[1] // s
        kern = np.zeros(ks, np.float32)
        js = [ks[1] // 2 - 1, ks[1] // 2] if ks[1] % 2 == 0 else [ks[1] // 2]
        for j in js:
            kern[ks[0] // 2, j] = (1.0 / max(overlap, 1.0)) if ks[1] % 2 == 0 else 1.0
        return (kern * scale).reshape(fk, s, 1, 1)
    if t == 'SubPixel':
        ks = (fk, 3)
        overlap = ks[1] // s
        kern = np.zeros(ks, np.float32)
        js = [ks[1] // 2 - 1, ks[1] // 2] if ks[1] % 2 == 0 else [ks[1] // 2]
        for j in js:
            kern[ks[0] // 2, j] = (1.0 / max(overlap, 1.0)) if ks[1] % 2 == 0 else 1.0
        kern = np.tile(kern[:, :, None, None], [1, 1, 1, s])
        return kern * scale
    if t == '1D':
        C = cfg.cin_channels
        overlap = float(s // s)
        kern = np.eye(C, dtype=np.float32).reshape(1, 1, C, C)
        kern = np.tile(kern, [1, s, 1, 1])
        if s % 2 == 0:
            kern = kern / max(overlap, 1.0)
        return kern * scale
    raise ValueError(t)


def init_params(cfg: OracleConfig, seed=5339, bias_scale=0.0):
    """Glorot-uniform conv kernels / zero biases (tf.layers default, modules.py:195-196),
    NN-init upsample kernels when cfg.NN_init.  ``bias_scale`` > 0 gives random biases so
    that tests exercise the bias paths (the reference's initial biases are zero)."""
    g = torch.Generator().manual_seed(seed)
    params = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        if name.endswith('/g'):
            continue                                      # set from the kernels below
        if name.endswith('bias'):
            t = torch.zeros(shape)
            if bias_scale > 0:
                t = (torch.rand(shape, generator=g) * 2 - 1) * bias_scale
        elif name == 'gc_embedding':                      # tf.truncated_normal_initializer(0, 0.1), modules.py:16-17
            t = torch.fmod(torch.randn(shape, generator=g), 2.0) * 0.1
        elif name.startswith('local_conditioning_upsampling') and cfg.NN_init:
            i = int(name.split('/')[0].rsplit('_', 1)[1]) - 1
            t = torch.from_numpy(np.ascontiguousarray(_nn_init_kernel(cfg, i, cfg.upsample_scales[i])))
        else:
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            if name.startswith('local_conditioning_upsampling') and cfg.upsample_type in ('2D', '1D'):
                fan_in, fan_out = shape[-1] * rf, shape[-2] * rf      # [kh,kw,out,in]
            else:
                fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            t = (torch.rand(shape, generator=g) * 2 - 1) * lim
        params[name] = t.float().contiguous()
    if cfg.wavenet_weight_normalization:                  # g = ||v|| (modules.py:104-108, 170)
        ordered = OrderedDict()
        for name in param_shapes(cfg):
            if name.endswith('/g'):
                v = params[name[:-1] + 'kernel']
                ordered[name] = v.reshape(-1, v.shape[-1]).norm(dim=0)
            else:
                ordered[name] = params[name]
        params = ordered
    return params


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


# ----------------------------------------------------------------------------- upsample net
def _same_pad_lr(k):
    left = (k - 1) // 2
    return left, k - 1 - left


def _act(cfg, x):
    if cfg.upsample_activation == 'Relu':
        return F.relu(x)
    if cfg.upsample_activation == 'LeakyRelu':
        return F.leaky_relu(x, cfg.leaky_alpha)
    assert cfg.upsample_activation is None
    return x


def upsample(params, cfg: OracleConfig, c):
    """c [B, C, Tc] -> [B, C, Tc*hop].  wavenet.py:680-702 / :781-803, modules.py:524-770."""
    params = effective_params(params, cfg)
    B, C, Tc = c.shape
    t = cfg.upsample_type
    if t == 'NearestNeighbor':                     # modules.py:524-536, wavenet.py:165-167
        return torch.repeat_interleave(c, cfg.hop, dim=2)
    x = c
    for i, s in enumerate(cfg.upsample_scales):
        K = params['local_conditioning_upsampling_%d/kernel' % (i + 1)]
        b = params['local_conditioning_upsampling_%d/bias' % (i + 1)]
        if t == '2D':      # NCHW [B,1,C,T]; Conv2DTranspose k=(fk,s) stride (1,s) SAME
            fk = K.shape[0]
            w = K.permute(3, 2, 0, 1).contiguous()          # [in,out,kh,kw]
            x = F.conv_transpose2d(x.unsqueeze(1), w, b, stride=(1, s), padding=((fk - 1) // 2, 0)).squeeze(1)
        elif t == '1D':    # NCHW [B,C,1,T]; Conv2DTranspose k=(1,s) stride (1,s)
            w = K.permute(3, 2, 0, 1).contiguous()
            x = F.conv_transpose2d(x.unsqueeze(2), w, b, stride=(1, s)).squeeze(2)
        elif t == 'SubPixel':  # NHWC [B,C(freq),T,1]; Conv2D 1->s k=(fk,3) SAME, then periodic shuffle
            fk = K.shape[0]
            w = K.permute(3, 2, 0, 1).contiguous()          # [out=s,in=1,kh,kw]
            pl, pr = _same_pad_lr(3)
            pt, pb = _same_pad_lr(fk)
            y = F.conv2d(F.pad(x.unsqueeze(1), (pl, pr, pt, pb)), w, b)   # [B,s,C,T]
            # PS (modules.py:604-640, r1=1): out[b,f,t*s+j] = conv[b,f,t,j]
            x = y.permute(0, 2, 3, 1).reshape(B, C, -1)
        elif t == 'Resize':    # NN resize x s along time then Conv2D 1->1 k=(fk,s) SAME
            fk = K.shape[0]
            w = K.permute(3, 2, 0, 1).contiguous()
            up = torch.repeat_interleave(x, s, dim=2)
            pl, pr = _same_pad_lr(s)
            pt, pb = _same_pad_lr(fk)
            x = F.conv2d(F.pad(up.unsqueeze(1), (pl, pr, pt, pb)), w, b).squeeze(1)
        else:
            raise ValueError(t)
        x = _act(cfg, x)
    return x


# ----------------------------------------------------------------------------- batch forward
def global_features(params, cfg, g):
    """wavenet.py:669-678: speaker ids [B] -> embedding rows (modules.py:19-21), or the given [B, gin] features."""
    if cfg.gin_channels <= 0 or g is None:
        return None
    if cfg.use_speaker_embedding:
        return params['gc_embedding'][torch.as_tensor(g).long().reshape(-1)]
    return torch.as_tensor(g).float().reshape(-1, cfg.gin_channels)


def _conv1x1(x, K, b):
    """x [B,Cin,T], K TF [1,Cin,Cout]."""
    y = torch.einsum('bit,io->bot', x, K[0])
    return y if b is None else y + b[None, :, None]


def glu_layer(P, params, cfg: OracleConfig, l, h, cu, mask, q, gvec=None):
    """One ResidualConv1DGLU.step (modules.py:471-521) of layer l on its input h [B,R,T]: returns (next layer input, skip
    contribution, gate output u).  P = kernels as the contraction sees them (bf16-rounded when emulating), params = fp32 biases,
    mask = {0,1} dropout mask [B,R,T] or None, q = rounding applied where the HIP path stores bf16 (identity for the fp32 oracle)."""
    k = cfg.kernel_size
    d = cfg.dilations()[l]
    keep = 1.0 - cfg.wavenet_dropout
    p = 'ResidualConv1DGLU_%d/' % l
    residual = h
    xin = h
    if mask is not None:
        xin = q(h * mask / keep)                                             # tf.layers.dropout on the conv INPUT only (modules.py:484)
    W = P[p + 'residual_block_causal_conv/kernel']                           # [k,R,G]
    z = F.conv1d(F.pad(xin, ((k - 1) * d, 0)), W.permute(2, 1, 0).contiguous(),
                 params.get(p + 'residual_block_causal_conv/bias'), dilation=d)   # modules.py:306-320
    if cu is not None:                                                       # modules.py:497-501
        z = z + _conv1x1(cu, P[p + 'residual_block_cin_conv/kernel'],
                         params.get(p + 'residual_block_cin_conv/bias'))
    if gvec is not None:                                                     # modules.py:503-508: g broadcast over time
        zg = gvec @ params[p + 'residual_block_gin_conv/kernel'][0]
        bg = params.get(p + 'residual_block_gin_conv/bias')
        z = z + (zg if bg is None else zg + bg)[:, :, None]
    a, b = z.chunk(2, dim=1)                                                 # modules.py:494
    u = q(torch.tanh(a) * torch.sigmoid(b))                                  # modules.py:510
    s = _conv1x1(u, P[p + 'residual_block_skip_conv/kernel'], params.get(p + 'residual_block_skip_conv/bias'))
    o = _conv1x1(u, P[p + 'residual_block_out_conv/kernel'], params.get(p + 'residual_block_out_conv/bias'))
    h = (o + residual) * SQRT_HALF if cfg.residual_legacy else (o + residual)   # modules.py:517-520
    return q(h), s, u


def contraction_params(params, cfg, emulate_bf16):
    """(P, params_effective): the kernels as the HIP contractions see them -- bf16 MFMA operands for every conv except the
    (fp32, K = Cin) input conv, the fp32 upsample kernels and the fp32 global-conditioning matvec."""
    params = effective_params(params, cfg)
    q = bf16_round if emulate_bf16 else (lambda t: t)
    P = {k: (q(v) if k.endswith('kernel') and not k.startswith(('local_conditioning', 'input_convolution')) and 'gin_conv' not in k else v)
         for k, v in params.items()}
    return P, params


def step(params, cfg: OracleConfig, x, c, dropout_masks=None, emulate_bf16=False, return_aux=False, g=None):
    """Teacher-forced parallel forward.  wavenet.py:650-721.

    x [B, Cin, T] fp32, c [B, C, Tc] fp32 -> y_hat [B, O, T].
    dropout_masks: None (dropout off) or list (per layer) of {0,1} float tensors [B, R, T];
      the layer input is x*mask/(1-p) (tf.layers.dropout, modules.py:484); the residual
      path uses the un-dropped x (modules.py:483, 517-520).
    """
    q = bf16_round if emulate_bf16 else (lambda t: t)
    P, params = contraction_params(params, cfg, emulate_bf16)
    gvec = global_features(params, cfg, g)                                   # [B, gin] or None (wavenet.py:669-678)
    aux = {}
    cu = None
    if c is not None:
        cu = upsample(params, cfg, c)
        assert cu.shape[-1] == x.shape[-1], (cu.shape, x.shape)     # wavenet.py:699
        aux['c_up'] = cu
        cu = q(cu)
    h = q(_conv1x1(x, P['input_convolution/kernel'], params['input_convolution/bias']))  # wavenet.py:705
    skips = None
    aux['layer_in'] = []
    aux['u'] = []
    for l, d in enumerate(cfg.dilations()):
        aux['layer_in'].append(h)
        h, s, u = glu_layer(P, params, cfg, l, h, cu, None if dropout_masks is None else dropout_masks[l], q, gvec)
        aux['u'].append(u)
        if skips is None:                                                    # wavenet.py:706-715
            skips = s
        else:
            skips = skips + s
            if cfg.legacy:
                skips = skips * SQRT_HALF
    aux['skips'] = skips
    y = q(F.relu(skips))                                                     # wavenet.py:136-149, 718-719
    y = _conv1x1(y, P['final_convolution_1/kernel'], params['final_convolution_1/bias'])
    y = q(F.relu(y))
    y = _conv1x1(y, P['final_convolution_2/kernel'], params['final_convolution_2/bias'])
    return (y, aux) if return_aux else y


# ----------------------------------------------------------------------------- losses
def _log_sum_exp(x):
    """mixture.py:5-10"""
    m = x.max(dim=-1).values
    m2 = x.max(dim=-1, keepdim=True).values
    return m + torch.log(torch.sum(torch.exp(x - m2), dim=-1))


def _log_prob_from_logits(x):
    """mixture.py:12-16"""
    m = x.max(dim=-1, keepdim=True).values
    return x - m - torch.log(torch.sum(torch.exp(x - m), dim=-1, keepdim=True))


def discretized_mix_logistic_loss(y_hat, y, num_classes=256, log_scale_min=-7.0):
    """mixture.py:18-74 with reduce=False.  y_hat [B,3M,T], y [B,T,1] -> [B,T,1]."""
    assert y_hat.shape[1] % 3 == 0
    M = y_hat.shape[1] // 3
    yh = y_hat.transpose(1, 2)
    logit_probs = yh[:, :, :M]
    means = yh[:, :, M:2 * M]
    log_scales = torch.clamp(yh[:, :, 2 * M:3 * M], min=log_scale_min)
    y = y * torch.ones(1, 1, M)
    centered_y = y - means
    inv_stdv = torch.exp(-log_scales)
    plus_in = inv_stdv * (centered_y + 1. / (num_classes - 1))
    cdf_plus = torch.sigmoid(plus_in)
    min_in = inv_stdv * (centered_y - 1. / (num_classes - 1))
    cdf_min = torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)
    log_one_minus_cdf_min = -F.softplus(min_in)
    cdf_delta = cdf_plus - cdf_min
    mid_in = inv_stdv * centered_y
    log_pdf_mid = mid_in - log_scales - 2. * F.softplus(mid_in)
    log_probs = torch.where(y < -0.999, log_cdf_plus,
                 torch.where(y > 0.999, log_one_minus_cdf_min,
                  torch.where(cdf_delta > 1e-5,
                              torch.log(torch.clamp(cdf_delta, min=1e-12)),
                              log_pdf_mid - float(np.log((num_classes - 1) / 2)))))
    log_probs = log_probs + _log_prob_from_logits(logit_probs)
    return -_log_sum_exp(log_probs).unsqueeze(-1)


def tf_ndtr(x):
    """Normal CDF exactly as TF 1.x computes it (special_math._ndtr): piecewise erf / erfc."""
    half_sqrt_2 = 0.5 * float(np.sqrt(2.0))
    w = x * half_sqrt_2
    z = torch.abs(w)
    y = torch.where(z < half_sqrt_2, 1.0 + torch.erf(w),
                    torch.where(w > 0, 2.0 - torch.erfc(z), torch.erfc(z)))
    return 0.5 * y


def gaussian_mle_loss(y_hat, y, log_scale_min_gauss, num_classes, use_cdf):
    """gaussian.py:5-37 with reduce=False.  y_hat [B,2,T], y [B,T,1] -> [B,T,1]."""
    assert y_hat.shape[1] == 2
    yh = y_hat.transpose(1, 2)
    mean = yh[:, :, 0]
    log_scale = torch.clamp(yh[:, :, 1], min=log_scale_min_gauss)
    yv = y.squeeze(-1)
    if use_cdf:
        scale = torch.exp(log_scale)          # tf.contrib.distributions.Normal(loc, scale).cdf, gaussian.py:22-26
        cdf_plus = tf_ndtr((yv + 1. / (num_classes - 1) - mean) / scale)
        cdf_min = tf_ndtr((yv - 1. / (num_classes - 1) - mean) / scale)
        log_prob = torch.log(torch.clamp(cdf_plus - cdf_min, min=1e-12))
    else:
        log_prob = -0.5 * (float(np.log(2. * np.pi)) + 2. * log_scale
                           + (yv - mean) ** 2 * torch.exp(-2. * log_scale))
    return -log_prob.unsqueeze(-1)


def sequence_mask(lengths, max_len):
    """util.py:165-171"""
    return (torch.arange(max_len)[None, :] < torch.as_tensor(lengths)[:, None]).float()


def training_loss(cfg: OracleConfig, y_hat, y, lengths):
    """wavenet.py:476-495 + get_mask :632-638.

    y_hat [B,O,T]; y: [B,T,1] float (scalar input) or [B,T] int (mulaw-quantize);
    prediction at t is scored against sample t+1, mask[:, 1:].
    """
    B, O, T = y_hat.shape
    mask = sequence_mask(lengths, T)[:, 1:]
    if cfg.input_type == 'mulaw-quantize':                  # modules.py:781-798
        logits = y_hat.transpose(1, 2)[:, :-1, :]
        tgt = y[:, 1:].long()
        losses = torch.logsumexp(logits, dim=-1) - logits.gather(-1, tgt.unsqueeze(-1)).squeeze(-1)
        masked = losses * mask
        return masked.sum() / torch.count_nonzero(masked).float()
    if cfg.out_channels == 2:                               # modules.py:819-836
        losses = gaussian_mle_loss(y_hat[:, :, :-1], y[:, 1:, :], cfg.log_scale_min_gauss,
                                   cfg.quantize_channels, cfg.cdf_loss)
    else:                                                   # modules.py:800-817
        losses = discretized_mix_logistic_loss(y_hat[:, :, :-1], y[:, 1:, :],
                                               num_classes=cfg.quantize_channels,
                                               log_scale_min=cfg.log_scale_min)
    m = mask.unsqueeze(-1)
    return (losses * m).sum() / m.sum()


# ----------------------------------------------------------------------------- samplers
def sample_from_discretized_mix_logistic(y, u1, u2, log_scale_min=-7.):
    """mixture.py:76-107.  y [B,3M,T]; u1 [B,T,M], u2 [B,T] uniforms in (1e-5, 1-1e-5)."""
    M = y.shape[1] // 3
    yt = y.transpose(1, 2)
    logit_probs = yt[:, :, :M]
    temp = logit_probs - torch.log(-torch.log(u1))
    argmax = temp.argmax(dim=-1)
    one_hot = F.one_hot(argmax, M).float()
    means = (yt[:, :, M:2 * M] * one_hot).sum(-1)
    log_scales = torch.clamp((yt[:, :, 2 * M:3 * M] * one_hot).sum(-1), min=log_scale_min)
    x = means + torch.exp(log_scales) * (torch.log(u2) - torch.log(1 - u2))
    return torch.clamp(x, -1., 1.)


def sample_from_gaussian(y, eps, log_scale_min_gauss):
    """gaussian.py:39-52.  y [B,2,T]; eps [B,T] standard normal draws."""
    yt = y.transpose(1, 2)
    mean = yt[:, :, 0]
    log_scale = torch.clamp(yt[:, :, 1], min=log_scale_min_gauss)
    return torch.clamp(mean + torch.exp(log_scale) * eps, -1., 1.)


def sample_categorical(logits, gumbel_u):
    """wavenet.py:861-867: tf.multinomial(logits, 1) == argmax(logits + Gumbel noise).
    logits [B,Q]; gumbel_u [B,Q] uniforms in (0,1).  Returns int64 [B]."""
    return (logits - torch.log(-torch.log(gumbel_u))).argmax(dim=-1)


# ----------------------------------------------------------------------------- incremental
def initial_input(cfg: OracleConfig, B):
    """wavenet.py:433-445: silence start frame."""
    if cfg.input_type == 'mulaw-quantize':
        from .mulaw import mulaw_quantize
        x0 = torch.zeros(B, cfg.quantize_channels)
        x0[:, int(mulaw_quantize(np.float64(0.0)))] = 1.0
        return x0
    return torch.zeros(B, 1)      # mulaw(0.0) == 0.0 == raw silence


def incremental(params, cfg: OracleConfig, c, T=None, noise=None, test_inputs=None,
                formulation='reference', g=None):
    """Fast-WaveNet autoregressive generation.  wavenet.py:724-911, modules.py:273-303.

    c [B,C,Tc] (already transposed as :427).  noise: dict with
        MoL:      'u1' [T,B,M], 'u2' [T,B]
        Gaussian: 'eps' [T,B]
        softmax:  'gumbel_u' [T,B,Q]
    test_inputs: None or [B,T,Cin] teacher-forcing inputs (:752-768, :877-878).
    formulation='reference': [B,2d+1,R] queues rebuilt by slice+concat each step (:285-288,
      :815-816); 'ring': O(1) ring buffers (what the HIP path does) -- identical results.
    Returns (outputs [B, Cout, T] as :911, raw [B, O, T] as :904-908).
    """
    params = effective_params(params, cfg)      # (the reference would use the raw v here: SURVEY appendix C-7; fixed)
    B = c.shape[0]
    cu = upsample(params, cfg, c)            # [B,C,Tup]
    if T is None:
        T = cu.shape[-1]
    R = cfg.residual_channels
    k = cfg.kernel_size
    dil = cfg.dilations()
    Wf = params['input_convolution/kernel'][0]
    bf = params['input_convolution/bias']
    lw = []
    for l in range(cfg.layers):
        p = 'ResidualConv1DGLU_%d/' % l
        lw.append(dict(
            Wlin=params[p + 'residual_block_causal_conv/kernel'].reshape(-1, cfg.gate_channels),  # modules.py:251
            b=params.get(p + 'residual_block_causal_conv/bias', 0.0),
            Wc=params[p + 'residual_block_cin_conv/kernel'][0], bc=params.get(p + 'residual_block_cin_conv/bias', 0.0),
            Wg=params[p + 'residual_block_gin_conv/kernel'][0] if cfg.gin_channels > 0 else None,
            bg=params.get(p + 'residual_block_gin_conv/bias', 0.0),
            Ws=params[p + 'residual_block_skip_conv/kernel'][0], bs=params.get(p + 'residual_block_skip_conv/bias', 0.0),
            Wo=params[p + 'residual_block_out_conv/kernel'][0], bo=params.get(p + 'residual_block_out_conv/bias', 0.0)))
    gvec = global_features(params, cfg, g)                                     # wavenet.py:766-777
    W1, b1 = params['final_convolution_1/kernel'][0], params['final_convolution_1/bias']
    W2, b2 = params['final_convolution_2/kernel'][0], params['final_convolution_2/bias']
    if formulation == 'reference':
        queues = [torch.zeros(B, k + (k - 1) * (d - 1), R) for d in dil]       # wavenet.py:815
    else:
        rings = [torch.zeros(B, 2 * d + 1, R) for d in dil]
    cur = initial_input(cfg, B)
    outs, raws = [], []
    for t in range(T):
        ct = cu[:, :, t]                                                       # wavenet.py:823
        x = cur @ Wf + bf                                                      # :826
        skips = None
        for l, d in enumerate(dil):
            w = lw[l]
            if formulation == 'reference':
                qd = torch.cat([queues[l][:, 1:, :], x.unsqueeze(1)], dim=1)   # modules.py:285-288
                queues[l] = qd
                taps = qd[:, 0::d, :] if d > 1 else qd                         # modules.py:291-292
            else:
                n = 2 * d + 1
                rings[l][:, t % n, :] = x
                taps = torch.stack([rings[l][:, (t - 2 * d) % n, :], rings[l][:, (t - d) % n, :], x], dim=1)
            z = taps.reshape(B, -1) @ w['Wlin'] + w['b']                        # modules.py:295-297
            z = z + (ct @ w['Wc'] + w['bc'])
            if gvec is not None:
                z = z + (gvec @ w['Wg'] + w['bg'])                              # modules.py:503-508
            a, b = z.chunk(2, dim=-1)
            u = torch.tanh(a) * torch.sigmoid(b)
            s = u @ w['Ws'] + w['bs']
            o = u @ w['Wo'] + w['bo']
            x = (o + x) * SQRT_HALF if cfg.residual_legacy else (o + x)
            if cfg.legacy:                                                      # wavenet.py:833-836
                skips = s if skips is None else (skips + s) * SQRT_HALF
            else:
                skips = s if skips is None else (skips + s)
        y = F.relu(skips) @ W1 + b1
        y = F.relu(y) @ W2 + b2                                                 # [B,O]
        raws.append(y)
        if cfg.scalar_input:
            yb = y.unsqueeze(-1)                                                # [B,O,1]
            if cfg.out_channels == 2:
                smp = sample_from_gaussian(yb, noise['eps'][t].reshape(B, 1), cfg.log_scale_min_gauss)
            else:
                smp = sample_from_discretized_mix_logistic(
                    yb, noise['u1'][t].reshape(B, 1, -1), noise['u2'][t].reshape(B, 1), cfg.log_scale_min)
            out = smp.reshape(B, 1)
            nxt = out
        else:
            idx = sample_categorical(y, noise['gumbel_u'][t])
            out = F.one_hot(idx, cfg.quantize_channels).float()
            nxt = out
        outs.append(out)
        if test_inputs is not None:
            nxt = test_inputs[:, t, :]                                          # :877-878
        cur = nxt
    return torch.stack(outs, dim=-1), torch.stack(raws, dim=-1)


# ----------------------------------------------------------------------------- optimiser
def learning_rate(step, init_lr=1e-3, schedule='exponential', decay_rate=0.5, decay_steps=200000,
                  warmup=4000.0):
    """wavenet.py:615-629."""
    if schedule == 'noam':
        s = float(step + 1)
        return max(init_lr * warmup ** 0.5 * min(s * warmup ** -1.5, s ** -0.5), 1e-4)
    return init_lr * decay_rate ** (step / decay_steps)


def clip_gradient(g, max_norm=100.0, max_value=5.0):
    """wavenet.py:586-593: per-tensor tf.clip_by_norm then tf.clip_by_value."""
    n = torch.sqrt((g * g).sum())
    g1 = g * max_norm / torch.maximum(n, torch.tensor(max_norm))
    return torch.clamp(g1, -max_value, max_value)


def adam_ema_update(p, g, m, v, ema, step, lr, beta1=0.9, beta2=0.999, eps=1e-6, ema_decay=0.9999):
    """TF-1 AdamOptimizer (epsilon-hat form) + ExponentialMovingAverage.apply, wavenet.py:549,
    :601-613.  ``step`` is the 1-based update count t.  Returns new (p, m, v, ema)."""
    # TF's ApplyAdam / assign_moving_average kernels work on float32 hyper-parameter tensors: the
    # (1 - beta) factors are float32 subtractions (1 - 0.999f != 0.001)
    f32 = np.float32
    b1, b2, dec = f32(beta1), f32(beta2), f32(ema_decay)
    one_m_b1, one_m_b2, one_m_dec = float(f32(1) - b1), float(f32(1) - b2), float(f32(1) - dec)
    lr_t = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    m = float(b1) * m + one_m_b1 * g
    v = float(b2) * v + one_m_b2 * g * g
    p = p - lr_t * m / (torch.sqrt(v) + eps)
    ema = ema - one_m_dec * (ema - p)
    return p, m, v, ema


def train_step(params, opt_state, cfg: OracleConfig, x, c, y, lengths, step_idx, dropout_masks=None,
               lr_kwargs=None, clip=True, max_norm=100.0, max_value=5.0, world_grads=None, adam_kwargs=None):
    """One full training step: fwd + loss + autograd bwd + clip + TF-Adam + EMA.
    opt_state: dict name -> (m, v, ema).  ``world_grads``: optional list of per-"tower" grad dicts
    to average with (wavenet.py:560-575)."""
    leaf = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in params.items())
    y_hat = step(leaf, cfg, x, c, dropout_masks=dropout_masks)
    loss = training_loss(cfg, y_hat, y, lengths)
    grads = torch.autograd.grad(loss, list(leaf.values()), allow_unused=True)
    grads = OrderedDict((k, (g if g is not None else torch.zeros_like(params[k]))) for k, g in zip(leaf, grads))
    if world_grads is not None:
        n = len(world_grads) + 1
        for k in grads:
            grads[k] = (grads[k] + sum(w[k] for w in world_grads)) / n
    lr = learning_rate(step_idx, **(lr_kwargs or {}))
    new_p, new_s = OrderedDict(), {}
    for k, p in params.items():
        g = clip_gradient(grads[k], max_norm, max_value) if clip else grads[k]
        m, v, e = opt_state[k]
        np_, m, v, e = adam_ema_update(p, g, m, v, e, step_idx + 1, lr, **(adam_kwargs or {}))
        new_p[k] = np_
        new_s[k] = (m, v, e)
    return loss.detach(), grads, new_p, new_s


def init_opt_state(params):
    """Adam slots start at zero; EMA shadow starts at the variable's value (TF semantics)."""
    return {k: (torch.zeros_like(v), torch.zeros_like(v), v.clone()) for k, v in params.items()}
