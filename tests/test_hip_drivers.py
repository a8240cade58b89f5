"""End-to-end run of the reference-shaped drivers on the device: wavenet_train on a small on-disk dataset in the reference's
formats (map.txt + audio-*.npy + mel-*.npy, feeder.py:241-257), checkpoint + restore, then wavenet_synthesize from the
checkpoint (train.py:345, synthesize.py:69, synthesizer.py:46) -- same directories and file names as the reference."""
import glob
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dataset(root, n=24, hop=16, mels=16, speakers=None):
    os.makedirs(os.path.join(root, 'audio'), exist_ok=True); os.makedirs(os.path.join(root, 'mels'), exist_ok=True)
    rng = np.random.RandomState(0)
    lines = []
    for i in range(n):
        frames = int(rng.randint(24, 48))
        t = np.arange(frames * hop)
        wav = (0.4 * np.sin(2 * np.pi * (100 + 7 * i) * t / 22050.0) + 0.02 * rng.randn(frames * hop)).astype(np.float32)
        mel = rng.uniform(-4, 4, size=(frames, mels)).astype(np.float32)
        np.save(os.path.join(root, 'audio', 'audio-%03d.npy' % i), wav)
        np.save(os.path.join(root, 'mels', 'mel-%03d.npy' % i), mel)
        sp = '<no_g>' if speakers is None else str(i % speakers)
        lines.append('audio/audio-%03d.npy|mels/mel-%03d.npy|mels/mel-%03d.npy|%s|%d|%d|text' % (i, i, i, sp, len(wav), frames))
    with open(os.path.join(root, 'map.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')
    return 'map.txt'


@pytest.mark.parametrize('gin', [False, True])
def test_train_checkpoint_restore_synthesize(tmp_path, gin):
    import hparams as H
    from wavenet_vocoder.train import wavenet_train, get_checkpoint_state
    from wavenet_vocoder.synthesize import wavenet_synthesize
    root = str(tmp_path)
    meta = _dataset(root, speakers=3 if gin else None)
    hp = H._build()
    hp.parse('layers=4,stacks=2,residual_channels=64,gate_channels=128,skip_out_channels=64,cin_channels=16,num_mels=16,out_channels=30,'
             'hop_size=16,upsample_scales=[4,4],max_time_steps=512,wavenet_batch_size=4,wavenet_test_batches=1,wavenet_synthesis_batch_size=2,'
             'wavenet_learning_rate=1e-3,wavenet_dropout=0.05' + (',gin_channels=8,n_speakers=3,use_speaker_embedding=True' if gin else ''))
    log_dir = os.path.join(root, 'logs-WaveNet')
    os.makedirs(log_dir, exist_ok=True)
    args = types.SimpleNamespace(base_dir=root, model='WaveNet', restore=False, wavenet_train_steps=6, checkpoint_interval=3,
                                 summary_interval=2, eval_interval=6, embedding_interval=100, eval_max_time=0)
    save_dir = wavenet_train(args, log_dir, hp, meta)
    assert save_dir is not None and os.path.isdir(save_dir), 'the driver returns None when training raised'
    ckpt = get_checkpoint_state(save_dir)
    assert ckpt.endswith('wavenet_model.ckpt-6.pt') and os.path.exists(ckpt)
    for f in ('wavs/step-3-pred.wav', 'wavs/step-3-real.wav', 'wavs/step-6-pred.wav', 'eval-dir/wavs/step-6-pred.wav', 'wavenet_events/scalars.jsonl',
              'plots/step-3-waveplot.png', 'plots/step-3-upsampled-features.png', 'plots/step-3-reconstruction-mel-spectrogram.png',
              'eval-dir/plots/step-6-waveplot.png', 'eval-dir/plots/step-6-reconstruction-mel-spectrogram.png', 'eval-dir/plots/step-6-upsampled-features.png'):
        assert os.path.exists(os.path.join(log_dir, f)), f
    if gin:     # speaker-embedding projector (reference train.py:26-39, 327-334): config + table + metadata
        cfgp = open(os.path.join(log_dir, 'wavenet_events', 'projector_config.pbtxt')).read()
        assert 'gc_embedding' in cfgp and 'metadata_path: "../metas/SpeakerEmbeddings.tsv"' in cfgp
        table = np.loadtxt(os.path.join(log_dir, 'wavenet_events', [l.split('"')[1] for l in cfgp.split('\n') if 'tensor_path' in l][0]), delimiter='\t')
        assert table.shape == (3, 8)
        assert len(open(os.path.join(log_dir, 'metas', 'SpeakerEmbeddings.tsv')).read().split()) == len(hp.speakers)
    rows = [json.loads(l) for l in open(os.path.join(log_dir, 'wavenet_events', 'scalars.jsonl'))]
    losses = [r['wavenet_loss'] for r in rows if 'wavenet_loss' in r]
    assert len(losses) == 3 and all(np.isfinite(losses))
    assert any('Wavenet_eval_model/eval_stats/wavenet_eval_loss' in r for r in rows)
    # restore and continue: the step counter and the optimiser state come back
    args.restore = True; args.wavenet_train_steps = 8
    assert wavenet_train(args, log_dir, hp, meta) == save_dir
    assert get_checkpoint_state(save_dir).endswith('wavenet_model.ckpt-8.pt')
    sd = torch.load(get_checkpoint_state(save_dir), map_location='cpu')
    assert int(sd['global_step']) == 8 and float(sd['adam_v'].abs().sum()) > 0 and not torch.equal(sd['params'], sd['ema'])
    # synthesis from mel .npy files (synthesize.py:69): wavs + map.txt in wavenet_<output_dir>
    mels_dir = os.path.join(root, 'mels_in'); os.makedirs(mels_dir)
    for i in range(3):
        np.save(os.path.join(mels_dir, 'mel-%d.npy' % i), np.load(os.path.join(root, 'mels', 'mel-%03d.npy' % i))[:6])
    cwd = os.getcwd(); os.chdir(root)
    try:
        sargs = types.SimpleNamespace(model='WaveNet', mels_dir=mels_dir, output_dir='output/', speaker_id='0,1,2' if gin else None)
        wavenet_synthesize(sargs, hp, save_dir)
    finally:
        os.chdir(cwd)
    wavs = sorted(glob.glob(os.path.join(root, 'wavenet_output', 'wavs', '*.wav')))
    assert len(wavs) == 3
    plots = os.listdir(os.path.join(root, 'wavenet_output', 'plots'))
    assert sum(p.startswith('wavenet-mel-spectrogram-') for p in plots) == 3 and sum(p.startswith('wavenet-waveplot-') for p in plots) == 3
    lines = open(os.path.join(root, 'wavenet_output', 'wavs', 'map.txt')).read().strip().split('\n')
    assert len(lines) == 3 and all(len(l.split('|')) == 3 for l in lines)
    from scipy.io import wavfile
    sr, data = wavfile.read(wavs[0])
    assert sr == hp.sample_rate and len(data) == 6 * 16 and np.abs(data).max() > 0


def test_feeder_prefetch_keeps_batches_and_errors_in_order(tmp_path):
    """The GPU path of Feeder.next_train_batch copies the NEXT batch to the device on a copy stream while the current step runs
    (round 4).  Same batches, in the same order, as the plain path of a second feeder over the same files (CPU tensors); device
    tensors equal their pinned sources; and a broken utterance still surfaces as 'feeder thread failed' on the call that would have
    returned its batch -- after every good batch before it, not one call early."""
    import hparams as H
    from wavenet_vocoder import feeder as F
    from wavenet_vocoder.train import _Coordinator
    root = str(tmp_path)
    meta = os.path.join(root, _dataset(root, n=24))
    hp = H._build()
    hp.parse('hop_size=16,num_mels=16,cin_channels=16,upsample_scales=[4,4],max_time_steps=512,wavenet_batch_size=4,wavenet_test_batches=1')
    co = [_Coordinator(), _Coordinator()]
    fd_gpu = F.Feeder(co[0], meta, root, hp)
    fd_cpu = F.Feeder(co[1], meta, root, hp, device=torch.device('cpu'))
    fd_gpu.start_threads(); fd_cpu.start_threads()
    try:
        for _ in range(70):                       # more than one 64-batch group: the order rng advances identically in both
            a, b = fd_gpu.next_train_batch(), fd_cpu.next_train_batch()
            torch.cuda.synchronize()
            for ta, tb in zip(a, b):
                assert (ta is None) == (tb is None)
                if ta is not None:
                    assert ta.is_cuda and torch.equal(ta.cpu(), tb)
    finally:
        for c in co:
            c.request_stop()
    # a broken utterance: every batch the plain path delivers before the error is delivered by the prefetching path too
    bad = os.path.join(root, F.Feeder(None, meta, root, hp, device=torch.device('cpu'))._train_meta[3][0])      # an utterance of the TRAIN split
    np.save(bad, np.load(bad)[:-2])               # audio / mel length mismatch
    counts = []
    for dev in (None, torch.device('cpu')):
        c = _Coordinator()
        fd = F.Feeder(c, meta, root, hp, device=dev)
        fd.start_threads()
        n = 0
        try:
            with pytest.raises(RuntimeError, match='feeder thread failed'):
                for _ in range(300):
                    fd.next_train_batch(); n += 1
        finally:
            c.request_stop()
        counts.append(n)
    assert counts[0] == counts[1] and counts[0] < 300, counts
