"""Batch synthesis entry point (reference ``wavenet_vocoder/synthesize.py``): lists mel .npy files (or reads the
Tacotron evaluation map.txt), synthesises them in chunks of ``wavenet_synthesis_batch_size`` and writes
``wavenet_<output_dir>/wavs/{wavenet-audio-*.wav, map.txt}`` and ``.../plots``."""
import os

import numpy as np
from tqdm import tqdm

from hparams import hparams_debug_string
from infolog import log
from wavenet_vocoder.synthesizer import Synthesizer
from wavenet_vocoder.train import get_checkpoint_state


def run_synthesis(args, checkpoint_path, output_dir, hparams):
    log_dir = os.path.join(output_dir, 'plots')
    wav_dir = os.path.join(output_dir, 'wavs')
    log(hparams_debug_string())
    synth = Synthesizer()
    synth.load(checkpoint_path, hparams)
    if args.model == 'Tacotron-2':
        with open(os.path.join(args.mels_dir, 'map.txt'), encoding='utf-8') as f:
            metadata = np.array([line.strip().split('|') for line in f])
        speaker_ids, mel_files, texts = metadata[:, 2], metadata[:, 1], metadata[:, 0]
        speaker_ids = None if (speaker_ids == '<no_g>').all() else speaker_ids
    else:
        mel_files = sorted(os.path.join(args.mels_dir, f) for f in os.listdir(args.mels_dir) if f.split('.')[-1] == 'npy')
        speaker_ids = None if args.speaker_id is None else args.speaker_id.replace(' ', '').split(',')
        if speaker_ids is not None:
            assert len(speaker_ids) == len(mel_files)
        texts = None
    log('Starting synthesis! (this will take a while..)')
    os.makedirs(log_dir, exist_ok=True)
    os.makedirs(wav_dir, exist_ok=True)
    bs = hparams.wavenet_synthesis_batch_size
    chunks = [mel_files[i:i + bs] for i in range(0, len(mel_files), bs)]
    with open(os.path.join(wav_dir, 'map.txt'), 'w') as file:
        for i, mel_batch in enumerate(tqdm(chunks)):
            mel_spectros = [np.load(mel_file) for mel_file in mel_batch]
            basenames = [os.path.basename(mel_file).replace('.npy', '') for mel_file in mel_batch]
            speaker_id_batch = None if speaker_ids is None else speaker_ids[i * bs:(i + 1) * bs]
            audio_files = synth.synthesize(mel_spectros, speaker_id_batch, basenames, wav_dir, log_dir)
            speaker_logs = ['<no_g>'] * len(mel_batch) if speaker_id_batch is None else speaker_id_batch
            for j, mel_file in enumerate(mel_batch):
                if texts is None:
                    file.write('{}|{}|{}\n'.format(mel_file, audio_files[j], speaker_logs[j]))
                else:
                    file.write('{}|{}|{}|{}\n'.format(texts[i * bs + j], mel_file, audio_files[j], speaker_logs[j]))
    log('synthesized audio waveforms at {}'.format(wav_dir))


def wavenet_synthesize(args, hparams, checkpoint):
    output_dir = 'wavenet_' + args.output_dir
    checkpoint_path = get_checkpoint_state(checkpoint)
    if checkpoint_path is None or not (os.path.exists(checkpoint_path) or os.path.exists(checkpoint_path + '.index')):
        raise RuntimeError('Failed to load checkpoint at {}'.format(checkpoint))
    log('loaded model at {}'.format(checkpoint_path))
    run_synthesis(args, checkpoint_path, output_dir, hparams)
