"""Host-side helpers of the WaveNet path: input-type predicates, the mu-law codec on numpy arrays AND on
device tensors (one entry point per reference function, dispatching like wavenet_vocoder/util.py:131-163 did
between numpy and TF), sequence masks and diagnostic plots.

The device branch calls the HIP kernels (wn_mulaw* in include/wavenet_mi355.h); the numpy branch is used by
the feeder / preprocessing on host arrays.  mu is fixed to 255 exactly as the reference does (util.py:48)."""
import numpy as np

_MU = 255


def _assert_valid_input_type(s):
    assert s == 'mulaw-quantize' or s == 'mulaw' or s == 'raw'


def is_mulaw_quantize(s):
    _assert_valid_input_type(s)
    return s == 'mulaw-quantize'


def is_mulaw(s):
    _assert_valid_input_type(s)
    return s == 'mulaw'


def is_raw(s):
    _assert_valid_input_type(s)
    return s == 'raw'


def is_scalar_input(s):
    return is_raw(s) or is_mulaw(s)


def _is_device_tensor(x):
    try:
        import torch
        return isinstance(x, torch.Tensor) and x.is_cuda
    except ImportError:
        return False


def _to_numpy(x):
    try:
        import torch
        if isinstance(x, torch.Tensor):
            return x.detach().cpu().numpy(), True
    except ImportError:
        pass
    return (x if np.isscalar(x) else np.asarray(x)), False


def mulaw(x, mu=256):
    """f(x) = sign(x) ln(1 + 255|x|) / ln(256)."""
    if _is_device_tensor(x):
        from wavenet_vocoder import _ext
        return _ext.mulaw(x.float().contiguous())
    a, was_t = _to_numpy(x)
    y = np.sign(a) * np.log1p(_MU * np.abs(a)) / np.log1p(_MU)
    return y


def inv_mulaw(y, mu=256):
    """f^-1(y) = sign(y) (256^|y| - 1) / 255."""
    if _is_device_tensor(y):
        from wavenet_vocoder import _ext
        return _ext.inv_mulaw(y.float().contiguous())
    a, _ = _to_numpy(y)
    return np.sign(a) * (1.0 / _MU) * ((1.0 + _MU) ** np.abs(a) - 1.0)


def mulaw_quantize(x, mu=256):
    """Companding + truncating quantiser to {0..255}."""
    if _is_device_tensor(x):
        from wavenet_vocoder import _ext
        return _ext.mulaw_quantize(x.float().contiguous())
    a, _ = _to_numpy(x)
    q = (mulaw(a) + 1) / 2 * _MU
    return int(q) if np.isscalar(a) or np.ndim(a) == 0 else q.astype(np.int64)


def inv_mulaw_quantize(y, mu=256):
    if _is_device_tensor(y):
        import torch
        from wavenet_vocoder import _ext
        return _ext.inv_mulaw_quantize(y.to(torch.int32).contiguous())
    a, _ = _to_numpy(y)
    a = np.float32(a) if np.isscalar(a) else np.asarray(a).astype(np.float32)
    return inv_mulaw(2 * a / _MU - 1)


def sequence_mask(input_lengths, max_len=None, expand=True):
    """float mask [B, max_len(, 1)]: 1 where t < length."""
    import torch
    lengths = torch.as_tensor(input_lengths)
    if max_len is None:
        max_len = int(lengths.max())
    m = (torch.arange(max_len, device=lengths.device)[None, :] < lengths[:, None]).float()
    return m.unsqueeze(-1) if expand else m


def _pyplot():
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    return plt


def waveplot(path, y_hat, y_target, hparams, title=None):
    """Target vs prediction wave plot (librosa.display is not a dependency here: plain matplotlib)."""
    plt = _pyplot()
    sr = hparams.sample_rate
    fig = plt.figure(figsize=(12, 4))
    for i, (sig, lab) in enumerate([(y_target, 'Target waveform'), (y_hat, 'Predicted waveform')]):
        if sig is None:
            continue
        ax = plt.subplot(3 if y_target is not None else 1, 1, i + 1 if y_target is not None else 1)
        t = np.arange(len(sig)) / float(sr)
        ax.plot(t, sig, linewidth=0.5)
        ax.set_title(lab)
    if y_target is not None:
        ax = plt.subplot(3, 1, 3)
        n = min(len(y_hat), len(y_target))
        t = np.arange(n) / float(sr)
        ax.plot(t, y_target[:n], linewidth=0.5, label='target', alpha=0.5)
        ax.plot(t, y_hat[:n], linewidth=0.5, label='prediction', color='red', alpha=0.5)
        ax.legend(loc='upper right')
    if title is not None:
        fig.suptitle(title, fontsize=10)
    plt.tight_layout()
    plt.savefig(path, format='png')
    plt.close()


def plot_spectrogram(pred_spectrogram, path, title=None, split_title=False, target_spectrogram=None, max_len=None, auto_aspect=False):
    plt = _pyplot()
    if max_len is not None:
        pred_spectrogram = pred_spectrogram[:max_len]
        if target_spectrogram is not None:
            target_spectrogram = target_spectrogram[:max_len]
    fig = plt.figure(figsize=(10, 8))
    if title is not None:
        fig.text(0.5, 0.18, title, horizontalalignment='center', fontsize=16)
    if target_spectrogram is not None:
        ax1 = fig.add_subplot(311)
        ax2 = fig.add_subplot(312)
        im = ax1.imshow(np.rot90(target_spectrogram), aspect='auto' if auto_aspect else None, interpolation='none')
        ax1.set_title('Target Mel-Spectrogram')
        fig.colorbar(mappable=im, shrink=0.65, orientation='horizontal', ax=ax1)
        ax2.set_title('Predicted Mel-Spectrogram')
    else:
        ax2 = fig.add_subplot(211)
    im = ax2.imshow(np.rot90(pred_spectrogram), aspect='auto' if auto_aspect else None, interpolation='none')
    fig.colorbar(mappable=im, shrink=0.65, orientation='horizontal', ax=ax2)
    plt.tight_layout()
    plt.savefig(path, format='png')
    plt.close()
