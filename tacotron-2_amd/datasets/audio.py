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
