"""Training CLI -- same flags as the reference's top-level train.py.  Only ``--model WaveNet`` is built in this
tree (the Tacotron feature-prediction model is out of scope); ``Tacotron`` / ``Tacotron-2`` raise a clear error.

Single GPU:   python train.py --model WaveNet
N GPUs:       python train.py --model WaveNet --hparams wavenet_num_gpus=N      (the reference's single command, hparams.py:37: this
              process starts the N ranks itself, one per GPU -- wavenet_vocoder/launch.py)
     or:      python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 train.py --model WaveNet
"""
import argparse
import os
from time import sleep

# the training step uses 3 HIP streams (+ the all-reduce streams under torch.distributed): more than the runtime's default of 4 hardware
# queues once data parallel; streams sharing a queue serialise.  Must be set before the HIP runtime initialises (first torch.cuda call).
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import infolog
from hparams import hparams
from infolog import log

log = infolog.log


def prepare_run(args):
    modified_hp = hparams.parse(args.hparams)
    run_name = args.name or args.model
    log_dir = os.path.join(args.base_dir, 'logs-{}'.format(run_name))
    os.makedirs(log_dir, exist_ok=True)
    infolog.init(os.path.join(log_dir, 'Terminal_train_log'), run_name, args.slack_url)
    return log_dir, modified_hp


def _self_launch_if_asked(args):
    """``wavenet_num_gpus = N > 1`` with no launcher around this process: start the N ranks (this same command line) and return their
    exit code; None when this process is itself a rank (or a single-GPU run)."""
    import sys
    from wavenet_vocoder import launch
    n = int(hparams.parse(args.hparams).wavenet_num_gpus)
    if n <= 1 or launch.launched():
        if launch.launched() and n > 1 and int(os.environ['WORLD_SIZE']) != n:
            raise SystemExit('wavenet_num_gpus={} but the launcher started {} ranks'.format(n, os.environ['WORLD_SIZE']))
        return None
    launch.require_gpus(n)
    print('wavenet_num_gpus={}: starting {} ranks, one per GPU'.format(n, n), flush=True)
    return launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], n)


def _init_distributed():
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    infolog.set_rank(int(os.environ.get('RANK', '0')))
    if not torch.cuda.is_available():
        raise SystemExit('train.py needs an MI355X: the HIP library is the only compute path (no CPU fallback)')
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        from wavenet_vocoder import launch as _launch
        _launch.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
    return world


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--base_dir', default='')
    parser.add_argument('--hparams', default='', help='Hyperparameter overrides as a comma-separated list of name=value pairs')
    parser.add_argument('--tacotron_input', default='training_data/train.txt')
    parser.add_argument('--wavenet_input', default='tacotron_output/gta/map.txt')
    parser.add_argument('--name', help='Name of logging directory.')
    parser.add_argument('--model', default='Tacotron-2')
    parser.add_argument('--input_dir', default='training_data', help='folder to contain inputs sentences/targets')
    parser.add_argument('--output_dir', default='output', help='folder to contain synthesized mel spectrograms')
    parser.add_argument('--mode', default='synthesis', help='mode for synthesis of tacotron after training')
    parser.add_argument('--GTA', default='True', help='Ground truth aligned synthesis, defaults to True, only considered in Tacotron synthesis mode')
    parser.add_argument('--restore', type=bool, default=True, help='Set this to False to do a fresh training')
    parser.add_argument('--summary_interval', type=int, default=250, help='Steps between running summary ops')
    parser.add_argument('--embedding_interval', type=int, default=5000, help='Steps between updating embeddings projection visualization')
    parser.add_argument('--checkpoint_interval', type=int, default=2500, help='Steps between writing checkpoints')
    parser.add_argument('--eval_interval', type=int, default=5000, help='Steps between eval on test data')
    parser.add_argument('--tacotron_train_steps', type=int, default=55000, help='total number of tacotron training steps')
    parser.add_argument('--wavenet_train_steps', type=int, default=500000, help='total number of wavenet training steps')
    parser.add_argument('--tf_log_level', type=int, default=1, help='accepted for compatibility; unused')
    parser.add_argument('--slack_url', default=None, help='accepted for compatibility; unused (no network)')
    args = parser.parse_args()

    accepted_models = ['Tacotron', 'WaveNet', 'Tacotron-2']
    if args.model not in accepted_models:
        raise ValueError('please enter a valid model to train: {}'.format(accepted_models))
    if args.model != 'WaveNet':
        raise NotImplementedError('--model {}: the Tacotron feature-prediction model is out of scope of this tree; '
                                  'train it with the reference and pass its GTA map.txt via --wavenet_input, then run --model WaveNet'.format(args.model))
    rc = _self_launch_if_asked(args)
    if rc is not None:
        raise SystemExit(rc)
    _init_distributed()
    log_dir, hp = prepare_run(args)
    from wavenet_vocoder.train import wavenet_train
    checkpoint = wavenet_train(args, log_dir, hp, args.wavenet_input)
    if checkpoint is None:
        raise SystemExit('Error occured while training Wavenet, Exiting!')


if __name__ == '__main__':
    main()
