"""WaveNet training driver: same entry point, directories, log lines and intervals as the reference's
``wavenet_vocoder/train.py`` (``wavenet_train(args, log_dir, hparams, input_path)``), on PyTorch-ROCm + the HIP
engine, one process per GPU (launch N ranks with ``python -m torch.distributed.run``; rank 0 writes files).

Checkpoints: ``<log_dir>/wave_pretrained/wavenet_model.ckpt-<step>.pt`` + a ``checkpoint`` index file naming the
newest one (what tf.train.get_checkpoint_state read).  Unlike the reference's shadow saver -- which stored the
raw variables under EMA names and dropped the optimiser slots -- the file holds the parameters,
the true EMA, Adam m/v and the global step.
"""
import json
import os
import time
import traceback

import numpy as np
import torch

import infolog
from datasets.audio import melspectrogram, save_wavenet_wav
from hparams import hparams_debug_string
from wavenet_vocoder import util
from wavenet_vocoder.feeder import Feeder, SyntheticFeeder, _interp
from wavenet_vocoder.models import create_model

log = infolog.log


class ValueWindow(object):
    """Sliding mean of the last N values (reference tacotron/utils/__init__.py)."""

    def __init__(self, window_size=100):
        self._window_size = window_size
        self._values = []

    def append(self, x):
        self._values = self._values[-(self._window_size - 1):] + [x]

    @property
    def sum(self):
        return sum(self._values)

    @property
    def count(self):
        return len(self._values)

    @property
    def average(self):
        return self.sum / max(1, self.count)

    def reset(self):
        self._values = []


class _Coordinator(object):
    def __init__(self):
        self._stop = False

    def should_stop(self):
        return self._stop

    def request_stop(self, e=None):
        self._stop = True


class _LateScalars(object):
    """Device scalars read by the host one (or more) steps late: ``push`` copies a small device vector to a pinned slot behind the
    work already enqueued and records an event; ``pop_ready(keep)`` waits for -- and returns -- all but the ``keep`` newest."""

    def __init__(self, width, device):
        self._pending = []
        self._width = width
        self._device = device
        self._pin = torch.cuda.is_available()

    def push(self, tag, vec):
        host = torch.empty(self._width, dtype=torch.float32, pin_memory=self._pin)
        host.copy_(vec, non_blocking=True)
        ev = None
        if vec.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
        self._pending.append((tag, host, ev))

    def pop_ready(self, keep=1):
        out = []
        while len(self._pending) > keep:
            tag, host, ev = self._pending.pop(0)
            if ev is not None:
                ev.synchronize()
            out.append((tag, tuple(float(v) for v in host)))
        return out

    def drain(self):
        return self.pop_ready(0)


def time_string():
    from datetime import datetime
    return datetime.now().strftime('%Y-%m-%d %H:%M')


def _dist():
    d = torch.distributed
    return d if (d.is_available() and d.is_initialized()) else None


def _rank():
    d = _dist()
    return d.get_rank() if d else 0


def save_checkpoint(model, save_dir, checkpoint_path, max_to_keep=20):
    step = model.global_step
    path = '{}-{}.pt'.format(checkpoint_path, step)
    torch.save(model.state_dict(), path)
    index = os.path.join(save_dir, 'checkpoint')
    hist = []
    if os.path.exists(index):
        with open(index) as f:
            try:
                hist = json.load(f).get('all_model_checkpoint_paths', [])
            except ValueError:
                hist = []          # the reference's TensorFlow text-proto index (get_checkpoint_state reads it): start a new history
    hist.append(os.path.basename(path))
    for old in hist[:-max_to_keep]:
        p = os.path.join(save_dir, old)
        if os.path.exists(p):
            os.remove(p)
    hist = hist[-max_to_keep:]
    with open(index, 'w') as f:
        json.dump({'model_checkpoint_path': os.path.basename(path), 'all_model_checkpoint_paths': hist}, f)
    log('Saved checkpoint {}'.format(path))
    return path


def get_checkpoint_state(save_dir):
    index = os.path.join(save_dir, 'checkpoint')
    if not os.path.exists(index):
        return None
    with open(index) as f:
        text = f.read()
    try:
        name = json.loads(text).get('model_checkpoint_path')
    except ValueError:
        # the reference's (TensorFlow) index file: text proto `model_checkpoint_path: "wavenet_model.ckpt-1000"`
        import re
        m = re.search(r'^model_checkpoint_path:\s*"([^"]+)"', text, re.M)
        name = m.group(1) if m else None
    return os.path.join(save_dir, name) if name else None


def _scalars_writer(tensorboard_dir):
    path = os.path.join(tensorboard_dir, 'scalars.jsonl')      # tensorboard is not a dependency: JSON lines instead
    return open(path, 'a')


def add_embedding_stats(tensorboard_dir, embedding_names, paths_to_meta, tables, step):
    """Speaker-embedding projector (reference train.py:26-39): TensorBoard's projector plugin reads ``projector_config.pbtxt``
    from the event directory.  The reference points it at the TF checkpoint (``model_checkpoint_path`` + ``tensor_name``); there
    is no TF checkpoint here, so every table is written as a TSV next to the config and referenced by ``tensor_path`` (the
    plugin's documented alternative, also what projector.tensorflow.org loads), with the same metadata file."""
    entries = []
    for name, meta, table in zip(embedding_names, paths_to_meta, tables):
        t = table.detach().float().cpu().numpy() if torch.is_tensor(table) else np.asarray(table, dtype=np.float32)
        fname = '{}-{}.tsv'.format(name.replace('/', '_').replace(':', '_'), step)
        np.savetxt(os.path.join(tensorboard_dir, fname), t.reshape(t.shape[0], -1), fmt='%.7g', delimiter='\t')
        entries.append('embeddings {{\n  tensor_name: "{}"\n  tensor_path: "{}"\n  metadata_path: "{}"\n}}\n'.format(name, fname, meta))
    with open(os.path.join(tensorboard_dir, 'projector_config.pbtxt'), 'w', encoding='utf-8') as f:
        f.write(''.join(entries))


def _plot_reconstruction_mel(wav, input_mel, path, hparams, title):
    """Mel of the generated audio next to the conditioning it was generated from (reference train.py:110-116, 150-156):
    both should agree on the low-frequency content.  ``input_mel``: [cin, Tc] or [Tc, cin]."""
    T2_output_range = (-hparams.max_abs_value, hparams.max_abs_value) if hparams.symmetric_mels else (0, hparams.max_abs_value)
    generated_mel = _interp(melspectrogram(np.asarray(wav, dtype=np.float64), hparams).T, T2_output_range)
    m = input_mel.detach().float().cpu().numpy() if torch.is_tensor(input_mel) else np.asarray(input_mel)
    if m.ndim == 2 and m.shape[0] == hparams.cin_channels and m.shape[1] != hparams.cin_channels:
        m = m.T
    util.plot_spectrogram(generated_mel, path, title=title, target_spectrogram=m)


def save_log(model, batch, step, plot_dir, wav_dir, hparams, model_name):
    """Predicted-vs-target wav + plots for item 0 of the current batch (reference train.py:128-162)."""
    log('\nSaving intermediate states at step {}'.format(step))
    x, y, lengths, c, g = batch
    idx = 0
    length = int(lengths[idx])
    y_hat = model.step(x[idx:idx + 1], c[idx:idx + 1], g=None if g is None else g[idx:idx + 1])
    T = y_hat.shape[-1]
    nps = model.engine.noise_per_step
    if util.is_mulaw_quantize(hparams.input_type):
        from wavenet_vocoder import _ext
        pred = util.inv_mulaw_quantize(_ext.argmax_channels(y_hat)).reshape(-1)
        target = util.inv_mulaw_quantize(y[idx].reshape(-1))
    else:
        noise = torch.randn(T, 1, nps, device=y_hat.device) if hparams.out_channels == 2 else torch.rand(T, 1, nps, device=y_hat.device) * (1 - 2e-5) + 1e-5
        out = torch.empty(1, T, device=y_hat.device)
        model.engine.sample(y_hat, noise, out)
        pred, target = out.reshape(-1), y[idx].reshape(-1).float()
        if util.is_mulaw(hparams.input_type):
            pred, target = util.inv_mulaw(pred), util.inv_mulaw(target)
    pred = pred[:length].cpu().numpy(); target = target[:length].cpu().numpy()
    save_wavenet_wav(pred, os.path.join(wav_dir, 'step-{}-pred.wav'.format(step)), sr=hparams.sample_rate)
    save_wavenet_wav(target, os.path.join(wav_dir, 'step-{}-real.wav'.format(step)), sr=hparams.sample_rate)
    feats = torch.empty(1, hparams.cin_channels, T, device=y_hat.device)
    model.engine.upsampled_features(feats)
    try:
        util.waveplot(os.path.join(plot_dir, 'step-{}-waveplot.png'.format(step)), pred, target, hparams,
                      title='{}, {}, step={}'.format(model_name, time_string(), step))
        util.plot_spectrogram(feats[0].cpu().numpy().T, os.path.join(plot_dir, 'step-{}-upsampled-features.png'.format(step)),
                              title='Upsampled Local Condition features, step={}'.format(step), auto_aspect=True)
        _plot_reconstruction_mel(pred, c[idx], os.path.join(plot_dir, 'step-{}-reconstruction-mel-spectrogram.png'.format(step)), hparams,
                                 'Local Condition vs Reconst. Mel-Spectrogram, step={}'.format(step))
    except Exception as e:      # plotting must never kill a training run
        log('plotting skipped: {}'.format(e))


def eval_step(model, batch, step, plot_dir, wav_dir, scalars, hparams, model_name):
    """Full-utterance autoregressive generation of item 0, teacher-forced unless wavenet_natural_eval
    (reference train.py:89-126, wavenet.py:342-405)."""
    start_time = time.time()
    x, y, lengths, c, g = batch
    model.initialize(y, c, g, lengths)
    torch.cuda.synchronize()
    y_hat = model.tower_y_hat[0].cpu().numpy()
    y_target = model.tower_y_target[0].float().cpu().numpy()
    loss = float(model.eval_loss.item())
    duration = time.time() - start_time
    log('Time Evaluation: Generation of {} audio frames took {:.3f} sec ({:.3f} frames/sec)'.format(len(y_target), duration, len(y_target) / duration))
    save_wavenet_wav(y_hat, os.path.join(wav_dir, 'step-{}-pred.wav'.format(step)), sr=hparams.sample_rate)
    save_wavenet_wav(y_target, os.path.join(wav_dir, 'step-{}-real.wav'.format(step)), sr=hparams.sample_rate)
    try:
        util.waveplot(os.path.join(plot_dir, 'step-{}-waveplot.png'.format(step)), y_hat, y_target, hparams,
                      title='{}, {}, step={}, loss={:.5f}'.format(model_name, time_string(), step, loss))
        _plot_reconstruction_mel(y_hat, model.tower_eval_c[0], os.path.join(plot_dir, 'step-{}-reconstruction-mel-spectrogram.png'.format(step)), hparams,
                                 'Local Condition vs Reconst. Mel-Spectrogram, step={}, loss={:.5f}'.format(step, loss))
        feats = model.tower_eval_upsampled_local_features[0]
        util.plot_spectrogram(feats.float().cpu().numpy().T, os.path.join(plot_dir, 'step-{}-upsampled-features.png'.format(step)),
                              title='Upsampled Local Condition features, step={}, loss={:.5f}'.format(step, loss), auto_aspect=True)
    except Exception as e:
        log('plotting skipped: {}'.format(e))
    log('Eval loss for global step {}: {:.3f}'.format(step, loss))
    scalars.write(json.dumps({'step': step, 'Wavenet_eval_model/eval_stats/wavenet_eval_loss': loss}) + '\n'); scalars.flush()
    return loss


def train(log_dir, args, hparams, input_path):
    save_dir = os.path.join(log_dir, 'wave_pretrained')
    plot_dir = os.path.join(log_dir, 'plots')
    wav_dir = os.path.join(log_dir, 'wavs')
    eval_dir = os.path.join(log_dir, 'eval-dir')
    eval_plot_dir = os.path.join(eval_dir, 'plots')
    eval_wav_dir = os.path.join(eval_dir, 'wavs')
    tensorboard_dir = os.path.join(log_dir, 'wavenet_events')
    meta_folder = os.path.join(log_dir, 'metas')
    rank = _rank()
    infolog.set_rank(rank)            # rank 0 owns the terminal and Terminal_train_log; the others speak only through all_ranks=True
    if rank == 0:
        for d in (save_dir, plot_dir, wav_dir, eval_dir, eval_plot_dir, eval_wav_dir, tensorboard_dir, meta_folder):
            os.makedirs(d, exist_ok=True)
    if _dist():
        _dist().barrier()
    checkpoint_path = os.path.join(save_dir, 'wavenet_model.ckpt')
    input_path = os.path.join(args.base_dir, input_path)
    log('Checkpoint_path: {}'.format(checkpoint_path))
    log('Loading training data from: {}'.format(input_path))
    log('Using model: {}'.format(args.model))
    log(hparams_debug_string())

    torch.manual_seed(hparams.wavenet_random_seed)
    coord = _Coordinator()
    world = _dist().get_world_size() if _dist() else 1
    if getattr(hparams, 'mi355_synthetic_data', False) or not os.path.exists(input_path):
        if not getattr(hparams, 'mi355_synthetic_data', False):
            log('No metadata at {}: training on LJSpeech-shaped synthetic tensors'.format(input_path))
        feeder = SyntheticFeeder(hparams, hparams.wavenet_batch_size // world, hparams.max_time_steps)
    else:
        feeder = Feeder(coord, input_path, args.base_dir, hparams)

    model = create_model(args.model if args.model != 'Tacotron-2' else 'WaveNet', hparams)
    hop = hparams.hop_size
    max_t = hparams.max_time_steps if hparams.max_time_sec is None else int(hparams.max_time_sec * hparams.sample_rate)
    eval_max_t = int(getattr(args, 'eval_max_time', 0) or max_t)
    model.build(max(hparams.wavenet_batch_size // world, 1), max(max_t, eval_max_t))

    # speaker-embedding metadata for the projector (reference train.py:233-244)
    if getattr(hparams, 'speakers_path', None) is not None:
        speaker_embedding_meta = hparams.speakers_path
    else:
        speaker_embedding_meta = os.path.join(meta_folder, 'SpeakerEmbeddings.tsv')
        if rank == 0 and not os.path.isfile(speaker_embedding_meta):
            with open(speaker_embedding_meta, 'w', encoding='utf-8') as f:
                for speaker in hparams.speakers:
                    f.write('{}\n'.format(speaker))
        speaker_embedding_meta = speaker_embedding_meta.replace(log_dir, '..')

    step = 0
    time_window = ValueWindow(100)
    loss_window = ValueWindow(100)
    log('Wavenet training set to a maximum of {} steps'.format(args.wavenet_train_steps))
    scalars = _scalars_writer(tensorboard_dir) if rank == 0 else None
    try:
        if args.restore:
            ckpt = get_checkpoint_state(save_dir)
            if ckpt and os.path.exists(ckpt):
                log('Loading checkpoint {}'.format(ckpt), slack=True)
                model.load_state_dict(torch.load(ckpt, map_location='cpu'))
                step = model.global_step
            elif ckpt and os.path.exists(ckpt + '.index'):
                # a checkpoint written by the reference (TensorFlow tensor bundle, train.py:67-87): parameters by name; its Adam slots
                # are not stored under names this optimiser could use, so the moments restart from zero
                from wavenet_vocoder.tf_checkpoint import load_reference_checkpoint
                log('Loading TensorFlow checkpoint {}'.format(ckpt), slack=True)
                flat, tf_step, missing = load_reference_checkpoint(ckpt, model.engine.layout)
                if missing:
                    raise RuntimeError('TensorFlow checkpoint {} lacks {} of the model\'s tensors, e.g. {}'.format(ckpt, len(missing), missing[:3]))
                p = torch.from_numpy(flat)
                model.load_state_dict({'params': p, 'ema': p.clone(), 'adam_m': torch.zeros_like(p), 'adam_v': torch.zeros_like(p), 'global_step': tf_step or 0})
                step = model.global_step
            else:
                log('No model to load at {}'.format(save_dir), slack=True)
        else:
            log('Starting new training!', slack=True)
        feeder.start_threads(None)

        # The host reads the loss ONE STEP LATE: step k's loss travels to pinned memory behind step k's kernels and is looked at after
        # step k+1 has been enqueued, so the device never waits for the host (the reference's session.run returns the loss of the step
        # it ran: one host round trip per step, which here would expose the ~1 ms of enqueue time every step).  On steps that WRITE
        # something (summary, checkpoint, eval, the last step) the read is not late: the NaN / > 100 guard (train.py:307-309) runs on
        # the step's own loss before anything is saved, so `--restore` can never resume from an exploded checkpoint -- one host sync
        # per interval.  Data parallel: the loss the host sees is the tower mean (identical on every rank, so the guard trips on all
        # of them in the same iteration) and the same small collective carries a "some rank's feeder failed" count, so that every
        # rank leaves the loop at the SAME step instead of hanging in an all-reduce.
        late = _LateScalars(2, model.device)
        dp = _dist() if (_dist() and world > 1) else None
        flag_const = (torch.zeros(1, device=model.device), torch.ones(1, device=model.device))     # (no H2D copy inside the loop)

        def agree(local_error, what):
            """One tiny MAX all-reduce: did ANY rank fail in the block just left?  Every rank raises together (the failing one its own
            exception), nobody is left waiting inside the next collective.  Also the barrier that keeps the other ranks out of the next
            step's all-reduce while rank 0 writes files."""
            if dp is not None:
                f = flag_const[1 if local_error is not None else 0].clone()
                dp.all_reduce(f, op=dp.ReduceOp.MAX)
                if float(f.item()) > 0 and local_error is None:
                    raise RuntimeError('another rank failed while {}: stopping every rank'.format(what))
            if local_error is not None:
                raise local_error

        # the first batch, before any step is enqueued: a rank whose feeder cannot deliver it has no last good batch to keep its
        # collectives matched with, so the ranks agree here once
        first_error, batch = None, None
        if step < args.wavenet_train_steps:
            try:
                batch = feeder.next_train_batch()
            except RuntimeError as e:
                first_error = e
            agree(first_error, 'fetching the first batch')
        last_batch = batch
        feeder_error = None
        newest = (step, float('nan'))                # (step, loss) of the newest loss the host has looked at
        while not coord.should_stop() and step < args.wavenet_train_steps:
            start_time = time.time()
            if batch is None and feeder_error is None:
                try:
                    batch = feeder.next_train_batch()
                    last_batch = batch
                except RuntimeError as e:
                    if dp is None:
                        raise
                    feeder_error = e
            if feeder_error is not None:
                # keep the collectives matched with the last good batch and raise the flag; every rank reads the summed flag at the same
                # late read (one step later, or in this very step when it is a writing step) and leaves there
                batch = last_batch
            x, y, lengths, c, g = batch
            model.initialize(y, c, g, lengths, x=x)
            loss_t = model.add_loss(flags=flag_const[1 if feeder_error is not None else 0])      # data parallel: ONE 2-float all-reduce (loss mean, flag count)
            step = model.add_optimizer(step)
            embed = (hparams.gin_channels > 0 and model.embedding_table is not None
                     and (step % args.embedding_interval == 0 or step == args.wavenet_train_steps or step == 1))
            writes = (step % args.summary_interval == 0 or step % args.checkpoint_interval == 0 or step % args.eval_interval == 0
                      or step >= args.wavenet_train_steps or embed)              # the same decision on every rank
            late.push(step, torch.cat([loss_t.reshape(1).float(), model.reduced_flags.reshape(1).float()]))
            ready = late.pop_ready(keep=0 if writes else 1)
            time_window.append(time.time() - start_time)
            for pstep, (loss, bad) in ready:
                if bad > 0:
                    if feeder_error is not None:
                        raise feeder_error
                    raise RuntimeError('the feeder of another rank failed at step {}: stopping every rank'.format(pstep))
                loss_window.append(loss)
                newest = (pstep, loss)
                message = 'Step {:7d} [{:.3f} sec/step, loss={:.5f}, avg_loss={:.5f}]'.format(pstep, time_window.average, loss, loss_window.average)
                log(message, end='\r', slack=(pstep % args.checkpoint_interval == 0))
                if np.isnan(loss) or loss > 100:
                    log('Loss exploded to {:.5f} at step {}'.format(loss, pstep))
                    raise Exception('Loss exploded')

            if dp is not None and step % args.checkpoint_interval == 0:
                from wavenet_vocoder.parallel import assert_replicas_in_sync
                assert_replicas_in_sync(model.params, what='parameters at step {}'.format(step))

            block_error = None
            try:
                if step % args.summary_interval == 0 and rank == 0:
                    log('\nWriting summary at step {}'.format(step))
                    gmax = float(model.grads.abs().max().item())
                    n_samples = int(lengths.sum().item()) * world
                    scalars.write(json.dumps({'step': newest[0], 'wavenet_loss': newest[1], 'wavenet_learning_rate': model.learning_rate,
                                              'wavenet_max_gradient_norm': gmax,
                                              'audio_samples_per_sec': n_samples / max(time_window.average, 1e-9)}) + '\n')
                    scalars.flush()

                if (step % args.checkpoint_interval == 0 or step == args.wavenet_train_steps) and rank == 0:
                    save_log(model, batch, step, plot_dir, wav_dir, hparams=hparams, model_name=args.model)
                    save_checkpoint(model, save_dir, checkpoint_path)

                if step % args.eval_interval == 0 and rank == 0:
                    log('\nEvaluating at step {}'.format(step))
                    eval_step(model, feeder.next_eval_batch(), step, eval_plot_dir, eval_wav_dir, scalars, hparams=hparams, model_name=args.model)
                if embed and rank == 0:
                    log('\nSaving Model Speaker Embeddings visualization..')
                    add_embedding_stats(tensorboard_dir, ['WaveNet_model/inference/gc_embedding'], [speaker_embedding_meta], [model.embedding_table], step)
                    log('WaveNet Speaker embeddings have been updated on tensorboard!')
            except Exception as e:
                if dp is None:
                    raise
                block_error = e
            if dp is not None and writes:
                # rank 0 wrote logs / a checkpoint / ran the eval step: the others wait HERE, not inside the next all-reduce -- and
                # learn here if it failed (a failing eval feeder on rank 0 used to leave them waiting at a barrier)
                agree(block_error, 'writing logs / checkpoint / evaluating at step {}'.format(step))
            batch = None

        if coord.should_stop() and step < args.wavenet_train_steps:
            raise RuntimeError('training stopped by the coordinator at step {} of {}'.format(step, args.wavenet_train_steps))
        log('Wavenet training complete after {} global steps'.format(args.wavenet_train_steps), slack=True)
        return save_dir
    except Exception as e:
        log('Exiting due to exception: {}'.format(e), slack=True, all_ranks=True)
        traceback.print_exc()
        coord.request_stop(e)


def wavenet_train(args, log_dir, hparams, input_path):
    return train(log_dir, args, hparams, input_path)
