"""Golden vectors for the HOST-side data formats, produced by executing the reference's own `wavenet_vocoder/feeder.py`
(batch assembly), `models/wavenet.py` (learning-rate schedules) and `datasets/audio.py` (wav writer) in this container
(TF symbols served by oracle/tf1_shim.py; the methods exercised are numpy code).  Writes tests/golden/host_golden.npz.
TEST INFRASTRUCTURE: tests/test_host_cpu.py compares tacotron-2_amd/'s mirror against these."""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
OUT = os.environ.get('WN_GOLDEN_DIR') or os.path.join(ROOT, 'tests', 'golden')      # WN_GOLDEN_DIR: write elsewhere (regen check of __graft_entry__.regen_golden)
sys.path.insert(0, ROOT)
from oracle import tf1_shim as shim  # noqa: E402


def main():
    if not os.path.isdir(os.path.join(REF, 'wavenet_vocoder')):
        raise SystemExit('needs /root/reference')
    tf = shim.install()
    tf.train.exponential_decay = lambda lr, step, decay_steps, decay_rate, staircase=False, name=None: lr * decay_rate ** (float(step) / decay_steps)
    for m in list(sys.modules):
        if m.split('.')[0] in ('wavenet_vocoder', 'datasets', 'infolog', 'hparams'):
            del sys.modules[m]
    sys.path.insert(0, REF)
    feeder = importlib.import_module('wavenet_vocoder.feeder')
    audio = importlib.import_module('datasets.audio')
    wn = importlib.import_module('wavenet_vocoder.models.wavenet')
    assert feeder.__file__.startswith(REF)
    out = {}

    # ---- feeder._prepare_batch (feeder.py:266-349) on fixed examples; no crop needed (max_time_steps large)
    rng = np.random.RandomState(3)
    hop, mels = 16, 8
    for tag, itype in (('raw', 'raw'), ('mulawq', 'mulaw-quantize')):
        hp = types.SimpleNamespace(wavenet_num_gpus=1, input_type=itype, quantize_channels=256, max_time_sec=None, max_time_steps=4096,
                                   hop_size=hop, frame_shift_ms=None, sample_rate=22050, symmetric_mels=True, max_abs_value=4.0,
                                   clip_for_wavenet=True, normalize_for_wavenet=True, cin_channels=mels, gin_channels=4)
        F = object.__new__(feeder.Feeder)
        F._hparams = hp; F.local_condition = True; F.global_condition = True
        examples = []
        for i, frames in enumerate((9, 5, 12, 7)):                       # distinct lengths: rows can be matched after the shuffle
            x = (rng.randint(0, 256, size=frames * hop).astype(np.int16) if itype == 'mulaw-quantize'
                 else rng.uniform(-0.9, 0.9, size=frames * hop).astype(np.float32))
            c = rng.uniform(-5, 5, size=(frames, mels)).astype(np.float32)     # beyond [-4, 4]: exercises clip_for_wavenet
            examples.append((x, c, str(i % 3), len(x)))
            out['%s_ex%d_x' % (tag, i)] = x; out['%s_ex%d_c' % (tag, i)] = c; out['%s_ex%d_g' % (tag, i)] = np.int32(i % 3)
        np.random.seed(0)
        inputs, targets, lengths, cb, gb = F._prepare_batch(list(examples))
        order = np.argsort(lengths)
        out[tag + '_inputs'] = inputs[order]; out[tag + '_targets'] = targets[order]; out[tag + '_lengths'] = lengths[order]
        out[tag + '_c'] = cb[order]; out[tag + '_g'] = gb[order]
    # crop path (feeder.py:368-387): only its invariants are recorded (np.random stream differs between implementations)
    hp.max_time_steps = 100                                              # -> max_steps 96, 6 frames
    hp.input_type = 'raw'
    x = rng.uniform(-0.9, 0.9, size=20 * hop).astype(np.float32); c = rng.uniform(-4, 4, size=(20, mels)).astype(np.float32)
    np.random.seed(1)
    crops = [F._adjust_time_resolution([(x, c, '0', len(x))], True, F._limit_time())[0] for _ in range(64)]
    out['crop_x_len'] = np.array([len(b[0]) for b in crops]); out['crop_c_len'] = np.array([len(b[1]) for b in crops])
    starts = np.array([int(np.where(np.all(c == b[1][0], axis=1))[0][0]) for b in crops])
    out['crop_start_min'] = starts.min(); out['crop_start_max'] = starts.max()

    # ---- learning-rate schedules (wavenet.py:615-629)
    steps = np.array([0, 1, 10, 3999, 4000, 4001, 100000, 200000, 400000, 1000000])
    dummy = object.__new__(wn.WaveNet)
    out['lr_steps'] = steps
    out['lr_noam'] = np.array([float(dummy._noam_learning_rate_decay(1e-3, torch.tensor(int(s)), 4000.0)) for s in steps])
    out['lr_exp'] = np.array([float(dummy._exponential_learning_rate_decay(1e-3, int(s), 0.5, 200000)) for s in steps])

    # ---- wav writer (datasets/audio.py:17-20) and hop size (:54-59)
    w = (0.25 * np.sin(np.arange(800) / 7.0)).astype(np.float32)
    path = os.path.join(OUT, '_tmp_ref.wav')
    audio.save_wavenet_wav(w.copy(), path, sr=22050, inv_preemphasize=True, k=0.97)
    from scipy.io import wavfile
    sr, data = wavfile.read(path); os.remove(path)
    out['wav_in'] = w; out['wav_int16'] = data; out['wav_sr'] = np.int32(sr)
    out['hop_from_ms'] = np.int32(audio.get_hop_size(types.SimpleNamespace(hop_size=None, frame_shift_ms=12.5, sample_rate=22050)))
    np.savez_compressed(os.path.join(OUT, 'host_golden.npz'), **out)
    print('host_golden.npz:', len(out), 'arrays; crop starts', out['crop_start_min'], '..', out['crop_start_max'], 'x_len', set(out['crop_x_len'].tolist()))


if __name__ == '__main__':
    main()
