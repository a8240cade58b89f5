"""mu-law codec + sequence mask, numpy restatement (test infrastructure only).

Follows /root/reference/wavenet_vocoder/util.py:
  mulaw              util.py:30-49   (mu is hard-overridden to 255, :48)
  inv_mulaw          util.py:52-68   (:67)
  mulaw_quantize     util.py:71-102  (:99; truncation toward zero via astype(int), :152-156)
  inv_mulaw_quantize util.py:105-129 (:127)
  sequence_mask      util.py:165-171
All arithmetic is float64 when given python floats / float64 arrays and follows
numpy promotion rules for float32 arrays exactly like the reference's numpy path.
"""
import numpy as np

MU = 255  # util.py:48,67,99,127 -- the `mu` argument is ignored by the reference


def mulaw(x):
    x = np.asarray(x)
    return np.sign(x) * np.log1p(MU * np.abs(x)) / np.log1p(MU)


def inv_mulaw(y):
    y = np.asarray(y)
    return np.sign(y) * (1.0 / MU) * ((1.0 + MU) ** np.abs(y) - 1.0)


def mulaw_quantize(x):
    y = mulaw(x)
    # scale [-1, 1] to [0, mu]; numpy astype(int) truncates toward zero (util.py:102,156)
    return ((y + 1) / 2 * MU).astype(np.int64)


def inv_mulaw_quantize(q):
    q = np.asarray(q)
    y = 2 * q.astype(np.float32) / MU - 1  # util.py:128 (_asfloat -> float32, :163)
    return inv_mulaw(y)


def sequence_mask(lengths, max_len=None):
    """util.py:165-171 (tf.sequence_mask): mask[b, t] = t < lengths[b], float32."""
    lengths = np.asarray(lengths)
    if max_len is None:
        max_len = int(lengths.max())
    return (np.arange(max_len)[None, :] < lengths[:, None]).astype(np.float32)
