"""Batch synthesis driver behind ``synthesize.py --model WaveNet | Tacotron-2`` (reference ``wavenet_vocoder/synthesize.py:69``
``wavenet_synthesize(args, hparams, checkpoint)``).  Inputs: every ``*.npy`` mel file of ``args.mels_dir`` (optionally one speaker id per
file in ``args.speaker_id``), or -- Tacotron-2 mode -- the ``text|mel|speaker`` rows of the Tacotron evaluation ``map.txt`` in that
directory.  Outputs, under ``wavenet_<output_dir>/``: ``wavs/wavenet-audio-<mel>.wav``, ``plots/``, and ``wavs/map.txt`` with one
``[text|]mel|wav|speaker`` row per utterance (the reference's format strings have one placeholder too few there, SURVEY appendix C-11:
its rows lose the last column; these keep it)."""
import os
from collections import namedtuple

import numpy as np
from tqdm import tqdm

from hparams import hparams_debug_string
from infolog import log
from wavenet_vocoder.synthesizer import Synthesizer
from wavenet_vocoder.train import get_checkpoint_state

Utterance = namedtuple('Utterance', 'text mel_path speaker')      # text None outside Tacotron-2 mode; speaker None without global conditioning


def _utterances(args):
    """The work list, in the order the reference walks it (sorted file names / map.txt row order)."""
    if args.model == 'Tacotron-2':
        with open(os.path.join(args.mels_dir, 'map.txt'), encoding='utf-8') as f:
            rows = [line.strip().split('|') for line in f if line.strip()]
        no_g = all(r[2] == '<no_g>' for r in rows)
        return [Utterance(r[0], r[1], None if no_g else r[2]) for r in rows]
    paths = sorted(os.path.join(args.mels_dir, name) for name in os.listdir(args.mels_dir) if name.rsplit('.', 1)[-1] == 'npy')
    speakers = [None] * len(paths)
    if args.speaker_id is not None:
        speakers = args.speaker_id.replace(' ', '').split(',')
        assert len(speakers) == len(paths), 'one speaker id per mel file'
    return [Utterance(None, p, s) for p, s in zip(paths, speakers)]


def run_synthesis(args, checkpoint_path, output_dir, hparams):
    plot_dir, wav_dir = os.path.join(output_dir, 'plots'), os.path.join(output_dir, 'wavs')
    log(hparams_debug_string())
    synth = Synthesizer()
    synth.load(checkpoint_path, hparams)
    work = _utterances(args)
    log('Starting synthesis! (this will take a while..)')
    for d in (plot_dir, wav_dir):
        os.makedirs(d, exist_ok=True)
    step = int(hparams.wavenet_synthesis_batch_size)
    with open(os.path.join(wav_dir, 'map.txt'), 'w') as index:
        for start in tqdm(range(0, len(work), step)):
            batch = work[start:start + step]
            mels = [np.load(u.mel_path) for u in batch]
            names = [os.path.basename(u.mel_path).replace('.npy', '') for u in batch]
            speakers = None if batch[0].speaker is None else [u.speaker for u in batch]
            wavs = synth.synthesize(mels, speakers, names, wav_dir, plot_dir)
            for u, wav in zip(batch, wavs):
                cols = ([] if u.text is None else [u.text]) + [u.mel_path, wav, '<no_g>' if u.speaker is None else u.speaker]
                index.write('|'.join(str(c) for c in cols) + '\n')
    log('synthesized audio waveforms at {}'.format(wav_dir))


def wavenet_synthesize(args, hparams, checkpoint):
    checkpoint_path = get_checkpoint_state(checkpoint)
    if checkpoint_path is None or not (os.path.exists(checkpoint_path) or os.path.exists(checkpoint_path + '.index')):
        raise RuntimeError('Failed to load checkpoint at {}'.format(checkpoint))
    log('loaded model at {}'.format(checkpoint_path))
    run_synthesis(args, checkpoint_path, 'wavenet_' + args.output_dir, hparams)
