"""Synthesis front-end: same class, methods, file names and mel preparation as the reference's
``wavenet_vocoder/synthesizer.py`` (load / synthesize), running the HIP Fast-WaveNet loop."""
import os

import numpy as np
import torch

from datasets.audio import get_hop_size, melspectrogram, save_wavenet_wav
from infolog import log
from wavenet_vocoder import util
from wavenet_vocoder.models import create_model


def _interp(feats, in_range):
    return (feats - in_range[0]) / (in_range[1] - in_range[0])


def _pad_inputs(x, maxlen, _pad=0):
    return np.pad(x, [(0, maxlen - len(x)), (0, 0)], mode='constant', constant_values=_pad)


class Synthesizer(object):
    def load(self, checkpoint_path, hparams, model_name='WaveNet'):
        log('Constructing model: {}'.format(model_name))
        self._hparams = hparams
        self.local_conditions, self.global_conditions = self._check_conditions()
        self.synth_debug = bool(hparams.wavenet_synth_debug)
        self.model = create_model(model_name, hparams)
        self._state = None
        self._tf_prefix = None
        if checkpoint_path is not None:
            log('Loading checkpoint: {}'.format(checkpoint_path))
            if os.path.exists(checkpoint_path + '.index'):
                # a checkpoint written by the reference (TensorFlow tensor bundle): read by name once the layout is known
                self._tf_prefix = checkpoint_path
            else:
                self._state = torch.load(checkpoint_path, map_location='cpu')
        self._capacity = (0, 0)

    def _ensure_capacity(self, batch, time_steps):
        if batch <= self._capacity[0] and time_steps <= self._capacity[1]:
            return
        if self.model.engine is not None:
            self.model.engine.close()
            self.model.engine = None
        cap = (max(batch, self._capacity[0]), max(time_steps, self._capacity[1]))
        # synthesis-only engine: no saved-activation / backward workspace (that is ~45 KB of HBM per (stream x sample) at the paper
        # shape, i.e. ~180 GB for the default 20 x 9 s batch; what synthesis needs is ~1.5 KB) and every buffer pre-sized
        self.model.build(cap[0], cap[1], inference_only=True)
        if self._tf_prefix is not None:
            from wavenet_vocoder.tf_checkpoint import load_reference_checkpoint
            flat, step, missing = load_reference_checkpoint(self._tf_prefix, self.model.engine.layout)
            if missing:
                raise RuntimeError('TensorFlow checkpoint {} lacks {} of the model\'s tensors, e.g. {}'.format(self._tf_prefix, len(missing), missing[:3]))
            p = torch.from_numpy(flat)
            self._state = {'params': p, 'ema': p.clone(), 'adam_m': torch.zeros_like(p), 'adam_v': torch.zeros_like(p), 'global_step': step or 0}
            self._tf_prefix = None
        if self._state is not None:
            self.model.load_state_dict(self._state)
            self.model.use_ema_weights() if getattr(self._hparams, 'mi355_synthesize_with_ema', False) else None
        self._capacity = (cap[0], self.model.max_time)

    def synthesize(self, mel_spectrograms, speaker_ids, basenames, out_dir, log_dir):
        hparams = self._hparams
        if self.synth_debug:
            assert len(hparams.wavenet_debug_mels) == len(hparams.wavenet_debug_wavs)
            mel_spectrograms = [np.load(mel_file) for mel_file in hparams.wavenet_debug_mels]
        hop = get_hop_size(hparams)
        audio_lengths = [len(x) * hop for x in mel_spectrograms]
        maxlen = max(len(x) for x in mel_spectrograms)
        T2_output_range = (-hparams.max_abs_value, hparams.max_abs_value) if hparams.symmetric_mels else (0, hparams.max_abs_value)
        if hparams.clip_for_wavenet:
            mel_spectrograms = [np.clip(x, T2_output_range[0], T2_output_range[1]) for x in mel_spectrograms]
        c_batch = np.stack([_pad_inputs(x, maxlen, _pad=T2_output_range[0]) for x in mel_spectrograms]).astype(np.float32)
        if hparams.normalize_for_wavenet:
            c_batch = _interp(c_batch, T2_output_range).astype(np.float32)
        # global condition: int32 speaker ids [B, 1] (reference synthesizer.py:72)
        g = None
        if self.global_conditions:
            if speaker_ids is None:
                raise RuntimeError('Please provide speaker ids (--speaker_id) to a globally conditioned WaveNet')
            g = torch.from_numpy(np.asarray(speaker_ids, dtype=np.int32).reshape(len(c_batch), 1))
        self._ensure_capacity(len(c_batch), maxlen * hop)
        dev = self.model.device
        test_inputs = None
        if self.synth_debug:
            test_wavs = [np.load(w).reshape(-1) for w in hparams.wavenet_debug_wavs]
            T = maxlen * hop
            test_inputs = torch.from_numpy(np.stack([np.pad(w[:T], (0, max(0, T - len(w)))) for w in test_wavs]).astype(np.float32)).to(dev)
        # c: [B, Tc, num_mels] like the reference's placeholder; the model transposes it (wavenet.py:427)
        self.model.initialize(None, torch.from_numpy(c_batch).to(dev), None if g is None else g.to(dev), None, test_inputs=test_inputs)
        torch.cuda.synchronize()
        self.model.engine.synth_check()           # the generation is enqueued asynchronously: a pipeline hand-off timeout surfaces here
        generated = self.model.tower_y_hat[0].float().cpu().numpy()
        feats = self.model.tower_synth_upsampled_local_features[0].cpu().numpy()
        generated_wavs = [w[:length] for w, length in zip(generated, audio_lengths)]
        upsampled_features = [f[:, :length] for f, length in zip(feats, audio_lengths)]
        audio_filenames = []
        for i, (wav, feat, input_mel) in enumerate(zip(generated_wavs, upsampled_features, mel_spectrograms)):
            audio_filename = os.path.join(out_dir, 'wavenet-audio-{}.wav'.format(basenames[i]))
            save_wavenet_wav(wav, audio_filename, sr=hparams.sample_rate, inv_preemphasize=hparams.preemphasize, k=hparams.preemphasis)
            audio_filenames.append(audio_filename)
            if log_dir is not None:
                try:
                    # generated-audio mel vs the conditioning mel (reference synthesizer.py:113-117)
                    util.plot_spectrogram(melspectrogram(wav.astype(np.float64), hparams).T,
                                          os.path.join(log_dir, 'wavenet-mel-spectrogram-{}.png'.format(basenames[i])),
                                          title='Local Condition vs Reconstructed Audio Mel-Spectrogram analysis', target_spectrogram=input_mel)
                    util.plot_spectrogram(feat.T, os.path.join(log_dir, 'wavenet-upsampled_features-{}.png'.format(basenames[i])),
                                          title='Upmsampled Local Condition features', auto_aspect=True)
                    util.waveplot(os.path.join(log_dir, 'wavenet-waveplot-{}.png'.format(basenames[i])), wav, None, hparams,
                                  title='WaveNet generated Waveform.')
                except Exception as e:
                    log('plotting skipped: {}'.format(e))
        return audio_filenames

    def _check_conditions(self):
        return self._hparams.cin_channels > 0, self._hparams.gin_channels > 0
