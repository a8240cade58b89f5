"""Audio helpers the WaveNet path needs (reference datasets/audio.py:17-20, 54-59).  The mel / Griffin-Lim
front-end of the reference belongs to the Tacotron model and to dataset preprocessing: out of scope."""
import numpy as np
from scipy.io import wavfile


def get_hop_size(hparams):
    hop_size = hparams.hop_size
    if hop_size is None:
        assert hparams.frame_shift_ms is not None
        hop_size = int(hparams.frame_shift_ms / 1000 * hparams.sample_rate)
    return hop_size


def save_wavenet_wav(wav, path, sr, inv_preemphasize=None, k=None):
    """Peak-normalise to int16 and write.  Like the reference (audio.py:17-20) the inverse pre-emphasis
    arguments are accepted and ignored; unlike it, the caller's array is not modified in place."""
    wav = np.asarray(wav, dtype=np.float32)
    out = wav * (32767 / max(0.01, float(np.max(np.abs(wav))) if wav.size else 0.01))
    wavfile.write(path, sr, out.astype(np.int16))


# ---- reconstruction mel (reference datasets/audio.py:62-68, 169-173, 222-259): only used for the diagnostic plot
# "Local Condition vs Reconst. Mel-Spectrogram" of train.py:114,154 / synthesizer.py:115.  The reference calls librosa
# (librosa.stft, librosa.filters.mel); librosa is not a dependency here, so its published algorithm is restated on numpy:
# centred frames (zero padding, pad_mode='constant'), periodic Hann window of win_size centred in n_fft, rfft; Slaney-scale
# triangular filters with area ("slaney") normalisation.
_mel_basis_cache = {}


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0; min_log_mel = min_log_hz / f_sp; logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0; min_log_mel = min_log_hz / f_sp; logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def _build_mel_basis(hparams):
    """== librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) (htk=False, norm='slaney'): [num_mels, 1 + n_fft // 2]."""
    assert hparams.fmax <= hparams.sample_rate // 2
    key = (hparams.sample_rate, hparams.n_fft, hparams.num_mels, hparams.fmin, hparams.fmax)
    if key not in _mel_basis_cache:
        n_mels = hparams.num_mels
        fftfreqs = np.linspace(0.0, hparams.sample_rate / 2.0, 1 + hparams.n_fft // 2)
        mel_f = _mel_to_hz(np.linspace(_hz_to_mel(hparams.fmin), _hz_to_mel(hparams.fmax), n_mels + 2))
        fdiff = np.diff(mel_f)
        ramps = mel_f[:, None] - fftfreqs[None, :]
        lower = -ramps[:-2] / fdiff[:-1, None]
        upper = ramps[2:] / fdiff[1:, None]
        weights = np.maximum(0.0, np.minimum(lower, upper))
        weights *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
        _mel_basis_cache[key] = weights.astype(np.float32)
    return _mel_basis_cache[key]


def _stft(y, hparams):
    """== librosa.stft(y, n_fft, hop_length, win_length, pad_mode='constant') -> complex [1 + n_fft // 2, frames]."""
    if getattr(hparams, 'use_lws', False):
        raise NotImplementedError('use_lws: the lws package is not available')
    from scipy.signal import get_window
    n_fft, hop, win = hparams.n_fft, get_hop_size(hparams), hparams.win_size
    w = get_window('hann', win, fftbins=True)
    lpad = (n_fft - win) // 2
    w = np.pad(w, (lpad, n_fft - win - lpad))
    y = np.pad(np.asarray(y, dtype=np.float64), n_fft // 2, mode='constant')
    n_frames = 1 + (len(y) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    return np.fft.rfft(y[idx] * w[None, :], axis=1).T


def _amp_to_db(x, hparams):
    min_level = np.exp(hparams.min_level_db / 20 * np.log(10))
    return 20 * np.log10(np.maximum(min_level, x))


def _normalize(S, hparams):
    m, lo = hparams.max_abs_value, hparams.min_level_db
    if hparams.allow_clipping_in_normalization:
        if hparams.symmetric_mels:
            return np.clip((2 * m) * ((S - lo) / (-lo)) - m, -m, m)
        return np.clip(m * ((S - lo) / (-lo)), 0, m)
    assert S.max() <= 0 and S.min() - lo >= 0
    if hparams.symmetric_mels:
        return (2 * m) * ((S - lo) / (-lo)) - m
    return m * ((S - lo) / (-lo))


def melspectrogram(wav, hparams):
    """Reference datasets/audio.py:62-68 -> [num_mels, frames]."""
    D = _stft(wav, hparams)
    S = _amp_to_db(np.dot(_build_mel_basis(hparams), np.abs(D) ** hparams.magnitude_power), hparams) - hparams.ref_level_db
    if hparams.signal_normalization:
        return _normalize(S, hparams)
    return S
