"""Synthesis CLI -- same flags as the reference's top-level synthesize.py; ``--model WaveNet`` synthesises audio from
mel .npy files in --mels_dir with the checkpoint under logs-<name>/wave_<--checkpoint>."""
import argparse
import os

from hparams import hparams
from infolog import log


def prepare_run(args):
    modified_hp = hparams.parse(args.hparams)
    run_name = args.name or args.wavenet_name or args.model
    wave_checkpoint = os.path.join('logs-' + run_name, 'wave_' + args.checkpoint)
    return wave_checkpoint, modified_hp


def main():
    accepted_modes = ['eval', 'synthesis', 'live']
    parser = argparse.ArgumentParser()
    parser.add_argument('--checkpoint', default='pretrained/', help='Path to model checkpoint')
    parser.add_argument('--hparams', default='', help='Hyperparameter overrides as a comma-separated list of name=value pairs')
    parser.add_argument('--name', help='Name of logging directory if the two models were trained together.')
    parser.add_argument('--tacotron_name', help='Name of logging directory of Tacotron. If trained separately')
    parser.add_argument('--wavenet_name', help='Name of logging directory of WaveNet. If trained separately')
    parser.add_argument('--model', default='Tacotron-2')
    parser.add_argument('--input_dir', default='training_data/', help='folder to contain inputs sentences/targets')
    parser.add_argument('--mels_dir', default='tacotron_output/eval/', help='folder to contain mels to synthesize audio from using the Wavenet')
    parser.add_argument('--output_dir', default='output/', help='folder to contain synthesized mel spectrograms')
    parser.add_argument('--mode', default='eval', help='mode of run: can be one of {}'.format(accepted_modes))
    parser.add_argument('--GTA', default='True', help='Ground truth aligned synthesis, defaults to True, only considered in synthesis mode')
    parser.add_argument('--text_list', default='', help='Text file contains list of texts to be synthesized. Valid if mode=eval')
    parser.add_argument('--speaker_id', default=None, help='Defines the speakers ids to use when running standalone Wavenet on a folder of mels.')
    args = parser.parse_args()

    if args.model not in ('Tacotron', 'WaveNet', 'Tacotron-2'):
        raise ValueError('please enter a valid model to synthesize with')
    if args.mode not in accepted_modes:
        raise ValueError('accepted modes are: {}, found {}'.format(accepted_modes, args.mode))
    if args.mode == 'live' and args.model == 'Wavenet':
        raise RuntimeError('Wavenet vocoder cannot be tested live due to its slow generation. Live only works with Tacotron!')
    if args.model != 'WaveNet':
        raise NotImplementedError('--model {}: the Tacotron feature-prediction model is out of scope of this tree; '
                                  'run it with the reference to produce mels, then --model WaveNet --mels_dir <dir>'.format(args.model))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit('synthesize.py needs an MI355X: the HIP library is the only compute path (no CPU fallback)')
    wave_checkpoint, hp = prepare_run(args)
    from wavenet_vocoder.synthesize import wavenet_synthesize
    wavenet_synthesize(args, hp, wave_checkpoint)


if __name__ == '__main__':
    main()
