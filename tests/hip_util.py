"""Shared helpers of the GPU parity tests: hparams builders, oracle <-> engine parameter transfer, and the
numpy mirror of the device dropout-mask hash (csrc/wn_common.h: wn_mix32 / wn_drop_word / wn_layer_key)."""
import numpy as np
import torch

import hparams as hparams_mod
from oracle import wavenet_oracle as O


def make_hp(**kw):
    hp = hparams_mod._build()
    for k, v in kw.items():
        hp.set_hparam(k, v)
    return hp


SMALL = dict(layers=4, stacks=2, residual_channels=64, gate_channels=128, skip_out_channels=64, cin_channels=16,
             num_mels=16, upsample_type='2D', upsample_scales=[4, 4], hop_size=16, out_channels=30,
             input_type='raw', quantize_channels=65536, legacy=False, residual_legacy=False, wavenet_dropout=0.0,
             log_scale_min=float(np.log(1e-14)), NN_scaler=0.3)


def oracle_cfg(hp):
    return O.OracleConfig.from_hparams(hp)


def upload_params(engine, params):
    """oracle name->tensor dict -> flat fp32 CUDA buffer in the engine's layout."""
    flat = torch.zeros(engine.n_params, dtype=torch.float32)
    for name, (shape, off) in engine.layout.items():
        t = params[name]
        assert tuple(t.shape) == tuple(shape), (name, t.shape, shape)
        flat[off:off + t.numel()] = t.reshape(-1)
    return flat.cuda()


def download_grads(engine, flat):
    flat = flat.cpu()
    return {name: flat[off:off + int(np.prod(shape))].view(*shape).clone() for name, (shape, off) in engine.layout.items()}


# ---- dropout mask mirror ---------------------------------------------------------------------
def _mix32(x):
    x = x.astype(np.uint64) & 0xffffffff
    x ^= x >> 16; x = (x * 0x85ebca6b) & 0xffffffff
    x ^= x >> 13; x = (x * 0xc2b2ae35) & 0xffffffff
    x ^= x >> 16
    return x


def layer_key(seed, layer):
    k = (seed * 0x9E3779B97F4A7C15 + (layer + 1) * 0xD1B54A32D192ED03) & 0xffffffffffffffff
    k ^= k >> 29
    return k & 0xffffffff, (k >> 32) & 0xffffffff


def dropout_mask(seed, layer, rows, R, p):
    """{0,1} float mask [rows, R] identical to the device's (element e = row*R + r)."""
    lo, hi = layer_key(seed, layer)
    e = np.arange(rows * R, dtype=np.uint64)
    a = _mix32((e >> 2) ^ lo)                                  # one double hash per QUAD of elements (wn_drop_quad)
    b = _mix32((a + hi) & 0xffffffff)
    w = np.where(((e >> 1) & 1) == 1, a ^ (((b << 16) | (b >> 16)) & 0xffffffff), b)
    bits = np.where((e & 1) == 1, w >> 16, w & 0xffff)
    thresh = int(np.rint(np.float32(p) * np.float32(65536.0)))
    return (bits >= thresh).astype(np.float32).reshape(rows, R)


def dropout_mask_rows(seed, layer, row0, nrows, R, p):
    """rows [row0, row0 + nrows) of the device mask (element e = row*R + r, absolute rows); in-place uint32 arithmetic
    (wraps like the device's), ~0.1 s per 11000 x 256 layer."""
    assert (row0 * R) % 4 == 0 and (nrows * R) % 4 == 0
    lo, hi = layer_key(seed, layer)
    a = np.arange((row0 * R) >> 2, ((row0 + nrows) * R) >> 2, dtype=np.uint32)

    def mix(x):
        x ^= (x >> np.uint32(16)); x *= np.uint32(0x85ebca6b)
        x ^= (x >> np.uint32(13)); x *= np.uint32(0xc2b2ae35)
        x ^= (x >> np.uint32(16))
        return x
    a ^= np.uint32(lo); a = mix(a)
    b = a + np.uint32(hi); b = mix(b)
    w1 = a ^ ((b << np.uint32(16)) | (b >> np.uint32(16)))
    thresh = int(np.rint(np.float32(p) * np.float32(65536.0)))
    out = np.empty((nrows * R // 4, 4), dtype=np.float32)
    out[:, 0] = (b & np.uint32(0xffff)) >= thresh
    out[:, 1] = (b >> np.uint32(16)) >= thresh
    out[:, 2] = (w1 & np.uint32(0xffff)) >= thresh
    out[:, 3] = (w1 >> np.uint32(16)) >= thresh
    return out.reshape(nrows, R)


def oracle_masks(seed, cfg, B, T):
    """per-layer masks in the oracle's [B, R, T] layout."""
    out = []
    for l in range(cfg.layers):
        m = dropout_mask(seed, l, B * T, cfg.residual_channels, cfg.wavenet_dropout)
        out.append(torch.from_numpy(m).view(B, T, cfg.residual_channels).permute(0, 2, 1).contiguous())
    return out


def rel_err(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def synth_batch(cfg, B, T, seed=0):
    """LJSpeech-shaped synthetic tensors as the feeder emits them (feeder.py:266-340)."""
    g = torch.Generator().manual_seed(seed)
    Tc = T // cfg.hop
    t = torch.arange(T).float()
    f = torch.rand(B, 1, generator=g) * 320 + 80
    wav = 0.3 * torch.sin(2 * np.pi * f * t[None] / 22050.0) + 0.1 * torch.randn(B, T, generator=g)
    wav = wav.clamp(-0.999, 0.999)
    c = torch.rand(B, cfg.cin_channels, Tc, generator=g)
    return wav, c


# ---- device noise stream mirror (csrc/wn_loss.hip: wn_philox4x32_10 / wn_u01 / wn_noise_kernel) -----------------------------
def philox4x32_10(counter, key):
    """counter: uint64 array (group indices), key: 64-bit seed -> uint32 [n, 4] (Philox4x32-10, counter words (lo, hi, 0, 0))."""
    c = [(counter & 0xffffffff).astype(np.uint64), (counter >> 32).astype(np.uint64), np.zeros_like(counter, dtype=np.uint64), np.zeros_like(counter, dtype=np.uint64)]
    k0, k1 = int(key) & 0xffffffff, (int(key) >> 32) & 0xffffffff
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        n0 = (p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)
        n1 = p1 & np.uint64(0xffffffff)
        n2 = (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)
        n3 = p0 & np.uint64(0xffffffff)
        c = [n0, n1, n2, n3]
        k0 = (k0 + 0x9E3779B9) & 0xffffffff; k1 = (k1 + 0xBB67AE85) & 0xffffffff
    return np.stack(c, axis=1).astype(np.uint32)


def device_uniform_noise(n, seed):
    """The first n elements of the device's uniform noise stream (float32, bit-exact)."""
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    w = philox4x32_10(g, seed).reshape(-1)[:n]
    u01 = ((w >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    return np.minimum(np.maximum(u01, np.float32(1e-5)), np.float32(1.0) - np.float32(1e-5))


def device_normal_noise(n, seed):
    """Mirror of the device's Gaussian stream (Box-Muller on the word pairs (w0, w1), (w2, w3) of each Philox group), float64."""
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    w = philox4x32_10(g, seed)
    u = ((w >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    u = u.astype(np.float64)
    r0, r1 = np.sqrt(-2 * np.log(u[:, 0])), np.sqrt(-2 * np.log(u[:, 2]))
    p0, p1 = 2 * np.pi * u[:, 1], 2 * np.pi * u[:, 3]
    return np.stack([r0 * np.cos(p0), r0 * np.sin(p0), r1 * np.cos(p1), r1 * np.sin(p1)], 1).reshape(-1)[:n]
