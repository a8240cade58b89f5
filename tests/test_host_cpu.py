"""CPU-only tests of the host layer: hparams surface, C-ABI library (loads + exports every declared symbol),
feeder alignment (port of the reference's test_wavenet_feeder.py onto synthetic .npy fixtures), numpy mu-law
branch of util.py against the reference-generated golden vectors, and the data-parallel glue under gloo."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hparams_surface_and_parse():
    import hparams as H
    import paper_hparams as P
    hp = H._build()
    # reference defaults (hparams.py) and the paper file's overrides
    assert (hp.layers, hp.stacks, hp.residual_channels, hp.gate_channels, hp.out_channels) == (20, 2, 128, 256, 2)
    assert hp.upsample_type == 'SubPixel' and hp.upsample_scales == [11, 25] and hp.hop_size == 275
    assert (P.hparams.layers, P.hparams.stacks, P.hparams.out_channels, P.hparams.upsample_scales) == (24, 4, 30, [5, 5, 11])
    assert abs(hp.log_scale_min - np.log(1e-14)) < 1e-12 and hp.wavenet_adam_epsilon == 1e-6
    hp.parse('layers=8,stacks=1,upsample_scales=[16,16],input_type=mulaw-quantize,wavenet_dropout=0.1,legacy=False,max_time_sec=0.5')
    assert hp.layers == 8 and hp.upsample_scales == [16, 16] and hp.input_type == 'mulaw-quantize'
    assert hp.wavenet_dropout == 0.1 and hp.legacy is False and hp.max_time_sec == 0.5
    with pytest.raises(ValueError):
        hp.parse('no_such_key=1')
    with pytest.raises(ValueError):
        hp.parse('upsample_scales=5')
    assert 'layers: 8' not in H.hparams_debug_string() or True
    assert H.hparams_debug_string().startswith('Hyperparameters:')


def test_library_loads_and_exports_every_declared_symbol():
    sys.path.insert(0, os.path.join(ROOT, 'tacotron-2_amd', 'csrc'))
    import build as B
    lib_path = B.build(verbose=False)
    assert os.path.exists(lib_path)
    from wavenet_vocoder import _ext
    lib = _ext.load_library()
    header = open(os.path.join(ROOT, 'include', 'wavenet_mi355.h')).read()
    declared = set(re.findall(r'\b(wn_[a-z0-9_]+)\s*\(', header)) - {'wn_ctx'}
    assert len(declared) >= 25
    for sym in sorted(declared):
        assert hasattr(lib, sym), 'symbol %s declared in include/wavenet_mi355.h is not exported' % sym
    assert set(_ext.exported_symbols()) <= declared | {'wn_debug_copy'}
    # host-only entry points work without a GPU
    assert abs(_ext.learning_rate('exponential', 1e-3, 200000) - 5e-4) < 1e-9
    assert abs(_ext.learning_rate('noam', 1e-3, 0) - max(1e-3 * 4000 ** 0.5 * 4000 ** -1.5, 1e-4)) < 1e-9
    assert lib.wn_dominant_kernel_name().decode().startswith('wn_gemm_lds_kernel')


def test_engine_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import hparams as H
    from wavenet_vocoder import _ext
    with pytest.raises(_ext.WnError) as ei:
        _ext.Engine(H._build(), 1, 275)
    assert ei.value.code == -3            # WN_E_HIP: no device -> no silent CPU path


def test_util_numpy_mulaw_matches_reference_golden(golden_dir):
    from wavenet_vocoder import util
    g = np.load(os.path.join(golden_dir, 'mulaw_golden.npz'))
    assert np.array_equal(util.mulaw_quantize(g['x']), g['quantized'])
    assert np.array_equal(util.mulaw(g['x']), g['mulaw'])
    assert np.array_equal(util.inv_mulaw_quantize(np.arange(256)), g['inv_q_all'])
    assert util.mulaw_quantize(0.0) == 127
    assert util.is_scalar_input('raw') and util.is_scalar_input('mulaw') and not util.is_scalar_input('mulaw-quantize')
    with pytest.raises(AssertionError):
        util.is_raw('pcm')


def _write_dataset(tmp, n=24, hop=16, num_mels=16, seed=0):
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(tmp, 'audio')); os.makedirs(os.path.join(tmp, 'mels'))
    lines = []
    for i in range(n):
        frames = int(rng.randint(20, 90))
        wav = rng.uniform(-0.9, 0.9, size=frames * hop).astype(np.float32)
        mel = rng.uniform(-4, 4, size=(frames, num_mels)).astype(np.float32)
        a, m = os.path.join(tmp, 'audio', 'audio-%03d.npy' % i), os.path.join(tmp, 'mels', 'mel-%03d.npy' % i)
        np.save(a, wav); np.save(m, mel)
        lines.append('|'.join([a, m, m, '<no_g>', 'text %d' % i]))
    meta = os.path.join(tmp, 'map.txt')
    with open(meta, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    return meta


def test_feeder_alignment_and_batch_layout(tmp_path):
    """Port of the reference's test_wavenet_feeder.py: after cropping, len(audio) == len(mel) * hop for every
    example, and the batch tensors have the layouts the model consumes."""
    import hparams as H
    from wavenet_vocoder.feeder import Feeder, _limit_time
    hp = H._build()
    hp.parse('hop_size=16,num_mels=16,cin_channels=16,upsample_scales=[4,4],max_time_steps=500,wavenet_batch_size=4,wavenet_test_batches=1')
    meta = _write_dataset(str(tmp_path))
    fd = Feeder(None, meta, str(tmp_path), hp, device=torch.device('cpu'))
    assert len(fd._test_meta) == 4 and len(fd._train_meta) == 20
    groups = fd._next_group(train=True)
    assert len(groups) == 64 and all(len(b) == 4 for b in groups)
    for b in groups[:8]:
        lim = _limit_time(list(b), hp, np.random.RandomState(0))
        for x, c, g, l in lim:
            assert len(x) % len(c) == 0 and len(x) // len(c) == 16 and len(x) <= 500 - 500 % 16
        inputs, targets, lengths, c, g = fd._prepare_batch(b)
        B, _, T = inputs.shape
        assert inputs.shape == (4, 1, T) and targets.shape == (4, T, 1) and c.shape == (4, 16, T // 16)
        assert lengths.dtype == np.int32 and T == lengths.max() and T % 16 == 0
        assert c.min() >= 0.0 and c.max() <= 1.0                      # normalize_for_wavenet -> [0,1]
        assert np.array_equal(inputs[:, 0, :], targets[:, :, 0])      # inputs and targets are the same waveform
    # mulaw-quantize: class ids, padded with the silence class
    hp.parse('input_type=mulaw-quantize,quantize_channels=256,out_channels=256')
    for p in os.listdir(os.path.join(str(tmp_path), 'audio')):
        f = os.path.join(str(tmp_path), 'audio', p)
        from wavenet_vocoder.util import mulaw_quantize
        np.save(f, mulaw_quantize(np.load(f)).astype(np.int16))
    fd2 = Feeder(None, meta, str(tmp_path), hp, device=torch.device('cpu'))
    inputs, targets, lengths, c, g = fd2._prepare_batch(fd2._next_group(train=True)[0])
    assert inputs.dtype == np.int32 and inputs.ndim == 2 and inputs.min() >= 0 and inputs.max() <= 255


def test_parameter_initialisation_nn_upsample_property():
    """NN-init upsample kernels make the (oracle) upsample net a scaled nearest-neighbour repeat."""
    import hparams as H
    from oracle import wavenet_oracle as O
    from wavenet_vocoder.models.modules import initialize_parameters, receptive_field_size
    assert receptive_field_size(24, 4, 3) == 505 and receptive_field_size(24, 2, 3) == 16381
    for ut, scales in (('2D', [5, 5, 11]), ('SubPixel', [11, 25])):
        hp = H._build(); hp.parse('upsample_type=%s,upsample_scales=[%s],cin_channels=16,num_mels=16,NN_scaler=0.3' % (ut, ','.join(map(str, scales))))
        cfg = O.OracleConfig.from_hparams(hp)
        shapes = O.param_shapes(cfg)
        off, layout = 0, {}
        for k, s in shapes.items():
            layout[k] = (tuple(s), off); off += (int(np.prod(s)) + 7) // 8 * 8
        flat = initialize_parameters(hp, layout, seed=1)
        params = {k: flat[o:o + int(np.prod(s))].view(*s) for k, (s, o) in layout.items()}
        c = torch.rand(2, 16, 3)
        cu = O.upsample(params, cfg, c)
        assert torch.allclose(cu, torch.repeat_interleave(c, cfg.hop, dim=2) * 0.3, atol=1e-6)
        k = params['ResidualConv1DGLU_0/residual_block_causal_conv/kernel']
        lim = np.sqrt(6.0 / (3 * cfg.residual_channels + 3 * cfg.gate_channels))
        assert float(k.abs().max()) <= lim + 1e-7 and float(k.std()) > 0.4 * lim
        assert float(params['ResidualConv1DGLU_0/residual_block_causal_conv/bias'].abs().max()) == 0.0


_DP_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tacotron-2_amd'))
import torch, torch.distributed as dist
from oracle import wavenet_oracle as O
from wavenet_vocoder.parallel import allreduce_mean_, shard_batch, rank, world_size
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=int(sys.argv[1]), world_size=2)
cfg = O.OracleConfig(layers=2, stacks=1, residual_channels=8, gate_channels=16, skip_out_channels=8, cin_channels=4,
                     upsample_type='2D', upsample_scales=[2, 2], wavenet_dropout=0.0)
params = O.init_params(cfg, seed=3, bias_scale=0.05)
g = torch.Generator().manual_seed(0)
B, T = 4, 32
wav = torch.rand(B, T, generator=g) * 1.6 - 0.8
c = torch.rand(B, 4, T // 4, generator=g)
idx = shard_batch(list(range(B)))                      # rank r: utterances [2r, 2r+1]
def tower_grads(ids):
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    y = O.step(leaf, cfg, wav[ids].view(len(ids), 1, T), c[ids])
    loss = O.training_loss(cfg, y, wav[ids].view(len(ids), T, 1), [T] * len(ids))
    gs = torch.autograd.grad(loss, list(leaf.values()), allow_unused=True)
    return loss.detach(), torch.cat([(gg if gg is not None else torch.zeros_like(v)).flatten() for gg, v in zip(gs, leaf.values())])
loss, flat = tower_grads(idx)
allreduce_mean_(flat)                                   # the product's tower-gradient mean
l0, g0 = tower_grads([0, 1]); l1, g1 = tower_grads([2, 3])
ref = (g0 + g1) / 2                                     # wavenet.py:564-575: mean over towers of per-tower grads
assert torch.allclose(flat, ref, atol=1e-7), float((flat - ref).abs().max())
lt = loss.clone(); dist.all_reduce(lt); lt /= 2
assert abs(float(lt) - float((l0 + l1) / 2)) < 1e-6    # reported loss = mean of tower losses (wavenet.py:515-516)
assert world_size() == 2 and idx == [2 * rank(), 2 * rank() + 1]
dist.barrier(); dist.destroy_process_group()
print('rank ok')
'''


def test_data_parallel_gradient_mean_gloo(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    script = tmp_path / 'dp_worker.py'
    script.write_text(_DP_WORKER % {'root': ROOT, 'port': port})
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and 'rank ok' in o, o[-2000:]


_BUCKET_WORKER = r'''
import sys, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/tacotron-2_amd')
import torch.distributed as dist
from wavenet_vocoder.parallel import allreduce_mean_buckets_, allreduce_mean_
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=int(sys.argv[1]), world_size=2)
class FakeEngine:                       # the engine's bucket interface (wn_bwd_num_buckets / _range / _wait_bucket) without a GPU
    def __init__(self, buckets): self.buckets = buckets; self.waited = []
    def grad_buckets(self): return list(self.buckets)
    def wait_bucket(self, i, stream): self.waited.append(i)
n = 1000
# the shape of the real table: top layers (+ head) first, ..., [0, lowest layers), tail; disjoint, covering, NOT in address order
buckets = [(600, 300), (400, 200), (200, 200), (0, 200), (900, 100)]
g = torch.Generator().manual_seed(int(sys.argv[1]) + 1)
mine = torch.randn(n, generator=g)
other = torch.randn(n, generator=torch.Generator().manual_seed(2 - int(sys.argv[1])))
flat = mine.clone()
eng = FakeEngine(buckets)
allreduce_mean_buckets_(eng, flat)
assert eng.waited == [0, 1, 2, 3, 4], eng.waited                       # every bucket gated on ITS event, in completion order
assert torch.allclose(flat, (mine + other) / 2, atol=1e-7)
ref = mine.clone(); allreduce_mean_(ref)
assert torch.equal(flat, ref)                                          # identical to the single flat collective
try:
    allreduce_mean_buckets_(FakeEngine(buckets[:-1]), mine.clone()); raise SystemExit('a hole in the bucket table went unnoticed')
except RuntimeError: pass
dist.barrier(); dist.destroy_process_group()
print('rank ok')
'''


def test_bucketed_gradient_allreduce_gloo(tmp_path):
    """The product's bucket walk (WaveNet.add_optimizer -> parallel.allreduce_mean_buckets_) on two gloo ranks with a stand-in for
    the engine's bucket table: same result as one flat all-reduce, one wait per bucket, holes in the table are detected."""
    port = 31500 + (os.getpid() % 2000)
    script = tmp_path / 'bucket_worker.py'
    script.write_text(_BUCKET_WORKER % {'root': ROOT, 'port': port})
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and 'rank ok' in o, o[-2000:]


_PRODUCT_DP_WORKER = r'''
import sys, types, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/tacotron-2_amd')
import torch.distributed as dist
import hparams as H
from wavenet_vocoder.models.wavenet import WaveNet
from wavenet_vocoder.parallel import assert_replicas_in_sync, param_checksum
rank = int(sys.argv[1])
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=rank, world_size=2)
N = 4096
class FakeEngine:
    """The engine calls WaveNet.add_optimizer makes, on CPU tensors: a rank-dependent "backward", the bucket table, and clip -> Adam
    -> EMA written with deterministic torch ops (what wn_optim_step does on the device, atomic-free since round 3)."""
    n_params = N
    def __init__(self): self.step_seen = 0
    def train_bwd(self, grads):
        g = torch.Generator().manual_seed(1000 * self.step_seen + rank)
        grads.copy_(torch.randn(N, generator=g) * (10.0 if self.step_seen == 1 else 0.1))      # step 1: norms above the clip threshold
        self.step_seen += 1
    def grad_buckets(self): return [(2048, 2048), (1024, 1024), (0, 1024)]
    def wait_bucket(self, i, stream): pass
    def optim_step(self, p, g, m, v, ema, lr, step):
        for lo in range(0, N, 512):                       # "variables" of 512 floats: per-variable clip_by_norm, clip_by_value
            gs = g[lo:lo + 512]
            gs = gs * 1.0 / torch.clamp(gs.norm(), min=1.0)
            gs = gs.clamp(-0.5, 0.5)
            m[lo:lo + 512].mul_(0.9).add_(gs, alpha=0.1); v[lo:lo + 512].mul_(0.999).addcmul_(gs, gs, value=0.001)
            p[lo:lo + 512].sub_(lr * m[lo:lo + 512] / (v[lo:lo + 512].sqrt() + 1e-8))
        ema.sub_((1 - 0.9999) * (ema - p))
hp = H._build()
model = WaveNet(hp)
model.engine = FakeEngine()
g0 = torch.Generator().manual_seed(7 + rank)              # replicas START different: build() broadcasts rank 0's parameters
model.params = torch.randn(N, generator=g0)
dist.broadcast(model.params, 0)
model.grads = torch.zeros(N); model.adam_m = torch.zeros(N); model.adam_v = torch.zeros(N); model.ema_params = model.params.clone()
model._dist = dist; model._world = 2; model._have_fwd = True
for step in range(3):
    model._have_fwd = True
    assert model.add_optimizer(step) == step + 1
    assert_replicas_in_sync(model.params, 'parameters'); assert_replicas_in_sync(model.adam_v, 'adam v'); assert_replicas_in_sync(model.ema_params, 'ema')
other = [torch.zeros(N), torch.zeros(N)]
dist.all_gather(other, model.params)
assert torch.equal(other[0], other[1])                    # bit-identical replicas after three steps
# the guard notices a single-ulp drift on one rank -- on BOTH ranks
if rank == 1:
    model.params.view(torch.int32)[17] += 1
try:
    assert_replicas_in_sync(model.params, 'parameters'); raise SystemExit('drift went unnoticed on rank %%d' %% rank)
except RuntimeError as e:
    assert 'diverged' in str(e)
dist.barrier(); dist.destroy_process_group()
print('rank ok')
'''


def test_product_add_optimizer_keeps_replicas_bit_identical_gloo(tmp_path):
    """wavenet.py:553-613 as this tree runs it data parallel: the PRODUCT's WaveNet.add_optimizer (backward -> bucketed tower mean ->
    clip / Adam / EMA) on two gloo ranks with a stand-in engine, three steps (one with norms above the clip threshold): parameters,
    Adam slots and EMA stay bit-identical across ranks, and the checksum guard (parallel.assert_replicas_in_sync, called by the
    training loop at every checkpoint interval) raises on every rank when one replica drifts by a single ulp."""
    port = 33500 + (os.getpid() % 2000)
    script = tmp_path / 'product_dp_worker.py'
    script.write_text(_PRODUCT_DP_WORKER % {'root': ROOT, 'port': port})
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and 'rank ok' in o, o[-3000:]


_TRAIN_LOOP_WORKER = r'''
import sys, types, os, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + '/tacotron-2_amd')
import torch.distributed as dist
rank, fail_at = int(sys.argv[1]), int(sys.argv[2])
explode_at = int(sys.argv[4]) if len(sys.argv) > 4 else 0
eval_fails = len(sys.argv) > 5 and sys.argv[5] == 'evalfail'
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=rank, world_size=2)
import hparams as H
from wavenet_vocoder import train as T
N = 256
class FakeModel:
    """What wavenet_vocoder.train.train() touches of WaveNet, on CPU tensors, with the product's collectives (loss mean, gradient mean)."""
    def __init__(self, hp): self.device = torch.device('cpu'); self.global_step = 0; self.learning_rate = 1e-3; self.embedding_table = None; self.steps = []
    def build(self, B, T):
        self.params = torch.randn(N, generator=torch.Generator().manual_seed(3 + rank)); dist.broadcast(self.params, 0)
        self.grads = torch.zeros(N); return self
    def initialize(self, y, c, g, lengths, x=None): self._x = x
    def add_loss(self, flags=None):
        from wavenet_vocoder.parallel import allreduce_loss_and_flags
        l = self._x.mean().reshape(1) * (float('nan') if (explode_at and self.global_step + 1 == explode_at and rank == 1) else 1.0)
        vec = allreduce_loss_and_flags(l, flags)            # the product's ONE small collective: tower-mean loss + flag count
        self.reduced_flags = vec[1:]; return vec[:1]
    def add_optimizer(self, step):
        self.grads.copy_(torch.full((N,), float(self._x.mean()))); dist.all_reduce(self.grads); self.grads /= 2
        self.params -= 0.01 * self.grads; self.global_step = step + 1; self.steps.append(self.global_step); return self.global_step
    def state_dict(self): return {'params': self.params.clone(), 'global_step': self.global_step}
class FakeFeeder:
    def __init__(self, hp, B, T): self.i = 0; self.test_steps = 1
    def start_threads(self, session=None): pass
    def next_train_batch(self):
        self.i += 1
        if rank == 1 and fail_at and self.i >= fail_at: raise RuntimeError('feeder thread failed: missing file (injected)')
        x = torch.full((2, 1, 8), 0.1 * self.i + rank)
        return x, x.view(2, 8, 1), torch.full((2,), 8, dtype=torch.int32), torch.zeros(2, 4, 2), None
    def next_eval_batch(self): return self.next_train_batch()
model_box = []
T.create_model = lambda name, hp: (model_box.append(FakeModel(hp)) or model_box[-1])
T.SyntheticFeeder = FakeFeeder
T.save_log = lambda *a, **k: None
def _eval(*a, **k):
    if eval_fails: raise RuntimeError('eval feeder failed (injected, rank 0 only)')
    return 0.0
T.eval_step = _eval
hp = H._build(); hp.parse('mi355_synthetic_data=True,wavenet_batch_size=4,max_time_steps=8,hop_size=4,upsample_scales=[2,2]')
args = types.SimpleNamespace(base_dir=sys.argv[3], model='WaveNet', restore=False, wavenet_train_steps=7, checkpoint_interval=2, summary_interval=3,
                             eval_interval=4, embedding_interval=100, eval_max_time=0)
log_dir = os.path.join(sys.argv[3], 'logs'); os.makedirs(log_dir, exist_ok=True)
ret = T.wavenet_train(args, log_dir, hp, 'no_such_map.txt')
m = model_box[0]
print('RESULT rank=%%d ret=%%s steps=%%d' %% (rank, 'ok' if ret else 'none', len(m.steps)))
both = [torch.zeros(N), torch.zeros(N)]
dist.all_gather(both, m.params)
assert torch.equal(both[0], both[1]), 'replicas differ after the loop'
dist.barrier(); dist.destroy_process_group()
print('rank ok')
'''


def _run_train_loop_workers(tmp_path, fail_at, extra=()):
    port = 35500 + (os.getpid() % 2000) + fail_at + 11 * len(extra)
    script = tmp_path / 'train_loop_worker.py'
    script.write_text(_TRAIN_LOOP_WORKER % {'root': ROOT, 'port': port})
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(fail_at), str(tmp_path)] + list(extra), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    res = []
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and 'rank ok' in o, o[-3000:]
        res.append(re.search(r'RESULT rank=(\d) ret=(\w+) steps=(\d+)', o).groups())
    return res, outs


def test_training_loop_never_checkpoints_an_exploded_step_gloo(tmp_path):
    """ADVICE round 3 (reference train.py:307-309 raises BEFORE saver.save): rank 1's loss turns NaN at step 4 -- a checkpoint-interval
    step.  The tower-mean loss is NaN on both ranks, the guard reads the step's OWN loss before anything is written, both ranks stop
    after 4 steps and no wavenet_model.ckpt-4 exists (the index still names step 2)."""
    res, outs = _run_train_loop_workers(tmp_path, 0, extra=['4'])
    assert {int(r[2]) for r in res} == {4} and all(r[1] == 'none' for r in res), res
    d = os.path.join(str(tmp_path), 'logs', 'wave_pretrained')
    assert os.path.exists(os.path.join(d, 'wavenet_model.ckpt-2.pt')) and not os.path.exists(os.path.join(d, 'wavenet_model.ckpt-4.pt'))
    assert 'ckpt-2' in open(os.path.join(d, 'checkpoint')).read()
    assert 'Loss exploded to' in outs[0] and 'Loss exploded to' not in outs[1] and '[rank 1] Exiting due to exception' in outs[1]          # rank 0 owns the terminal (infolog.set_rank)


def test_training_loop_eval_failure_on_rank0_stops_every_rank_gloo(tmp_path):
    """rank 0's eval step raises at step 4 while rank 1 waits for it: the post-block agreement (one MAX all-reduce in
    place of the barrier) tells rank 1, both leave after 4 steps."""
    res, outs = _run_train_loop_workers(tmp_path, 0, extra=['0', 'evalfail'])
    assert {int(r[2]) for r in res} == {4} and all(r[1] == 'none' for r in res), res
    assert 'another rank failed' in outs[1]


@pytest.mark.parametrize('fail_at', [0, 4, 5, 7])
def test_training_loop_data_parallel_control_flow_gloo(tmp_path, fail_at):
    """The PRODUCT training loop (wavenet_vocoder/train.py, mirror of the reference's train.py:345) on two gloo ranks with a stand-in
    model / feeder: losses read one step late, the replica checksum guard at every checkpoint interval, rank-0-only logging behind
    barriers -- and a feeder failure on ONE rank (at batch 4; at the last batch): the failing rank keeps its collectives matched with
    its last good batch, every rank sees the flag at the same late read and leaves the loop after the SAME number of steps (nobody is
    left inside an all-reduce), and the driver returns None like the reference's does after an exception (train.py:340-343)."""
    res, _ = _run_train_loop_workers(tmp_path, fail_at)
    steps = {int(r[2]) for r in res}
    assert len(steps) == 1, res                                  # both ranks ran the same number of optimiser steps
    if fail_at == 0:
        assert steps == {7} and all(r[1] == 'ok' for r in res)
        assert os.path.exists(os.path.join(str(tmp_path), 'logs', 'wave_pretrained', 'wavenet_model.ckpt-6.pt'))
    else:
        # the failing step + the one the others had already enqueued -- unless the failing step is a WRITING step (4: checkpoint + eval
        # interval, 7: the last), whose scalars are read at once
        want = fail_at if fail_at in (4, 7) else fail_at + 1
        assert steps == {want} and all(r[1] == 'none' for r in res), res


def test_late_scalars_are_read_one_step_behind():
    """The training loop's host-side reads (train._LateScalars): the loss of step k is looked at after step k + 1 was enqueued."""
    from wavenet_vocoder.train import _LateScalars
    late = _LateScalars(2, torch.device('cpu'))
    seen = []
    for k in range(1, 5):
        late.push(k, torch.tensor([float(k) * 0.5, 0.0]))
        seen += late.pop_ready(keep=1 if k < 4 else 0)
    assert [s for s, _ in seen] == [1, 2, 3, 4] and [v[0] for _, v in seen] == [0.5, 1.0, 1.5, 2.0]
    late.push(9, torch.tensor([float('nan'), 1.0]))
    (tag, (loss, bad)), = late.drain()
    assert tag == 9 and np.isnan(loss) and bad == 1.0


def test_host_formats_match_reference_execution(golden_dir, tmp_path):
    """Batch assembly, learning-rate schedules, wav writer and hop size against golden vectors produced by executing the
    reference's own feeder.py / wavenet.py / datasets/audio.py (oracle/gen_golden_host.py)."""
    import hparams as H
    from wavenet_vocoder import _ext
    from wavenet_vocoder.feeder import Feeder, _limit_time
    from datasets import audio
    g = np.load(os.path.join(golden_dir, 'host_golden.npz'))
    for tag, itype in (('raw', 'raw'), ('mulawq', 'mulaw-quantize')):
        hp = H._build()
        hp.parse('hop_size=16,num_mels=8,cin_channels=8,upsample_scales=[4,4],max_time_steps=4096,gin_channels=4,input_type=%s,quantize_channels=256' % itype)
        fd = object.__new__(Feeder)
        fd._hparams = hp; fd.local_condition = True; fd.global_condition = True; fd._rng = np.random.RandomState(5)
        fd._spec_pad = -hp.max_abs_value if hp.symmetric_mels else 0.
        ex = [(g['%s_ex%d_x' % (tag, i)], g['%s_ex%d_c' % (tag, i)], str(int(g['%s_ex%d_g' % (tag, i)])), len(g['%s_ex%d_x' % (tag, i)])) for i in range(4)]
        inputs, targets, lengths, c, gb = fd._prepare_batch(ex)
        order = np.argsort(lengths)
        inputs, targets, lengths, c, gb = inputs[order], targets[order], lengths[order], c[order], gb[order]
        assert np.array_equal(lengths, g[tag + '_lengths']) and lengths.dtype == np.int32
        np.testing.assert_allclose(c, g[tag + '_c'], rtol=0, atol=1e-7)              # clip -> pad with range min -> [0,1] -> [B, mels, Tc]
        assert np.array_equal(gb, g[tag + '_g']) and gb.dtype == np.int32 and gb.shape == (4, 1)
        if itype == 'raw':
            assert np.array_equal(inputs, g[tag + '_inputs']) and np.array_equal(targets, g[tag + '_targets'])
        else:
            # the reference feeds one-hot [B, Q, T] (all-zero columns in the padding); the device boundary takes the class ids
            ref_ids = g[tag + '_inputs'].argmax(axis=1)
            for b, n in enumerate(lengths):
                assert np.array_equal(inputs[b, :n], ref_ids[b, :n]) and np.array_equal(targets[b, :n], g[tag + '_targets'][b, :n, 0])
                assert not g[tag + '_inputs'][b, :, n:].any()
    # hop-aligned random crop (feeder.py:368-387): same lengths, same range of start frames
    hp = H._build(); hp.parse('hop_size=16,num_mels=8,cin_channels=8,upsample_scales=[4,4],max_time_steps=100')
    rng = np.random.RandomState(0)
    x = np.arange(20 * 16, dtype=np.float32); cc = np.repeat(np.arange(20, dtype=np.float32)[:, None], 8, axis=1)
    starts = set()
    for _ in range(400):
        (xo, co, _, ln), = _limit_time([(x, cc, None, len(x))], hp, rng)
        assert len(xo) == int(g['crop_x_len'][0]) and len(co) == int(g['crop_c_len'][0]) and xo[0] == co[0, 0] * 16
        starts.add(int(co[0, 0]))
    assert min(starts) == int(g['crop_start_min']) and max(starts) == int(g['crop_start_max'])
    # learning-rate schedules (wavenet.py:615-629)
    for s, a, b in zip(g['lr_steps'], g['lr_noam'], g['lr_exp']):
        assert abs(_ext.learning_rate('noam', 1e-3, int(s), 0.5, 200000, 4000.0) - a) <= 1e-6 * a
        assert abs(_ext.learning_rate('exponential', 1e-3, int(s), 0.5, 200000, 4000.0) - b) <= 1e-6 * b
    # wav writer (datasets/audio.py:17-20): peak-normalised int16, inverse pre-emphasis NOT applied
    p = os.path.join(str(tmp_path), 'w.wav')
    audio.save_wavenet_wav(g['wav_in'].copy(), p, sr=22050, inv_preemphasize=True, k=0.97)
    from scipy.io import wavfile
    sr, data = wavfile.read(p)
    assert sr == int(g['wav_sr']) and np.array_equal(data, g['wav_int16'])
    hp2 = H._build(); hp2.hop_size = None; hp2.frame_shift_ms = 12.5
    assert audio.get_hop_size(hp2) == int(g['hop_from_ms'])


def test_tf_checkpoint_bundle_reader_round_trip(tmp_path):
    """TF-1 tensor-bundle reader: LevelDB-table index with prefix-compressed keys over several blocks, raw data
    shard, reference variable names (EMA shadows, enclosing scopes) -> engine tensor names."""
    from wavenet_vocoder import tf_checkpoint as C
    rng = np.random.RandomState(0)
    scope = 'WaveNet_model/inference/'
    ema = '/ExponentialMovingAverage'
    tensors, expect = {}, {}
    for l in range(3):
        for kind, shape in (('causal', (3, 8, 16)), ('cin', (1, 4, 16)), ('skip', (1, 8, 8)), ('out', (1, 8, 8))):
            for leaf, shp in (('kernel', shape), ('bias', (shape[-1],))):
                name = '%sResidualConv1DGLU_%d/residual_block_%s_conv_ResidualConv1DGLU_%d/%s%s' % (scope, l, kind, l, leaf, ema)
                tensors[name] = rng.randn(*shp).astype(np.float32)
                expect['ResidualConv1DGLU_%d/residual_block_%s_conv/%s' % (l, kind, leaf)] = tensors[name]
    for nm, shp in (('input_convolution/input_convolution/kernel', (1, 1, 8)), ('input_convolution/input_convolution/bias', (8,)),
                    ('skip_convolutions/final_convolution_1/kernel', (1, 8, 8)), ('skip_convolutions/final_convolution_2/bias', (6,)),
                    ('local_conditioning_upsampling_1/ConvTranspose2D_layer_0/kernel', (3, 4, 1, 1)),
                    ('local_conditioning_upsampling_2/ConvTranspose2D_layer_1/bias', (1,))):
        tensors[scope + nm + ema] = rng.randn(*shp).astype(np.float32)
    expect.update({'input_convolution/kernel': tensors[scope + 'input_convolution/input_convolution/kernel' + ema],
                   'input_convolution/bias': tensors[scope + 'input_convolution/input_convolution/bias' + ema],
                   'final_convolution_1/kernel': tensors[scope + 'skip_convolutions/final_convolution_1/kernel' + ema],
                   'final_convolution_2/bias': tensors[scope + 'skip_convolutions/final_convolution_2/bias' + ema],
                   'local_conditioning_upsampling_1/kernel': tensors[scope + 'local_conditioning_upsampling_1/ConvTranspose2D_layer_0/kernel' + ema],
                   'local_conditioning_upsampling_2/bias': tensors[scope + 'local_conditioning_upsampling_2/ConvTranspose2D_layer_1/bias' + ema]})
    tensors['global_step'] = np.array(123456, dtype=np.int64)
    tensors['WaveNet_model/beta1_power'] = np.array(0.5, dtype=np.float32)          # optimiser junk: ignored by the name map
    prefix = os.path.join(str(tmp_path), 'wavenet_model.ckpt-123456')
    C.write_bundle(prefix, tensors, entries_per_block=5)
    header, entries = C.read_index(prefix + '.index')
    assert header['num_shards'] == 1 and set(entries) == set(tensors)
    back = C.load_checkpoint(prefix)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and np.array_equal(back[k], v), k
    # into an engine-style layout (name -> (shape, offset))
    layout, off = {}, 0
    for k, v in expect.items():
        layout[k] = (tuple(v.shape), off); off += (v.size + 7) // 8 * 8
    layout['final_convolution_2/kernel'] = ((1, 8, 6), off)                          # not in the checkpoint: reported as missing
    flat, step, missing = C.load_reference_checkpoint(prefix, layout)
    assert step == 123456 and missing == ['final_convolution_2/kernel']
    for k, v in expect.items():
        shape, o = layout[k]
        assert np.array_equal(flat[o:o + v.size].reshape(shape), v), k
    assert C.crc32c(b'123456789') == 0xe3069283                                        # CRC-32C check value
    # a file that is not a table is rejected loudly
    with open(os.path.join(str(tmp_path), 'junk.index'), 'wb') as f:
        f.write(b'\x00' * 100)
    with pytest.raises(ValueError):
        C.read_index(os.path.join(str(tmp_path), 'junk.index'))


def test_tf_checkpoint_state_file_is_understood(tmp_path):
    from wavenet_vocoder.train import get_checkpoint_state
    d = str(tmp_path)
    with open(os.path.join(d, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "wavenet_model.ckpt-2000"\nall_model_checkpoint_paths: "wavenet_model.ckpt-1000"\nall_model_checkpoint_paths: "wavenet_model.ckpt-2000"\n')
    assert get_checkpoint_state(d) == os.path.join(d, 'wavenet_model.ckpt-2000')


def test_static_isa_audit_no_scratch_and_two_workgroups_per_cu():
    """Regression guard on the compiled gfx950 code objects (no GPU needed): no kernel may spill to scratch, and the
    production 256x128 tile kernels must stay within the VGPR / LDS budget that lets two 8-wave workgroups share a CU
    (the overlap of one workgroup's epilogue with the other's main loop relies on it)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('isa_audit', os.path.join(ROOT, 'tools', 'isa_audit.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = mod.audit()
    assert len(rows) > 40
    # (the fused d x | d z pair kernel keeps ONE 64-bit value in scratch across its main loop -- stored before the loop, reloaded after
    # it, nothing inside: allowed up to 16 bytes; everything else must be spill-free.  SGPR spills go to VGPR lanes, not memory.)
    bad = [r['name'] for r in rows if (r['scratch'] or r['vspill'] or r['dyn_stack'])
           and not (r['name'].startswith('wn_fused_pair_kernel') and int(r['scratch']) <= 16 and int(r['vspill']) <= 2 and not r['dyn_stack'])]
    assert not bad, 'kernels with scratch / spills: %s' % bad
    prod = [r for r in rows if r['name'].startswith('wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3,') or r['name'].startswith('wn_wgrad_lds_kernel')
            or r['name'].startswith('wn_fused_pair_kernel')]
    assert len(prod) >= 8
    for r in prod:
        if re.match(r'wn_wgrad_lds_kernel<3, [23]>', r['name']):
            # multi-A weight-gradient workgroups: alone on their CU by design (128 - 192 accumulator registers, <= 120 KiB ring)
            assert r['vgpr'] <= 256 and r['lds'] <= 160 * 1024, (r['name'], r['vgpr'], r['lds'])
            continue
        assert r['vgpr'] <= 128, (r['name'], r['vgpr'])            # 512 / 128 = 4 waves per SIMD = two 8-wave workgroups per CU
        assert 2 * r['lds'] <= 160 * 1024, (r['name'], r['lds'])   # two residents in the 160 KB LDS


def test_isa_audit_unwaited_prefetch_registers_stay_untouched():
    """wn_synth_pipe.hip issues its ring-tap prefetches as inline-asm loads with the wait many instructions later; hipcc
    does not track loads inside inline asm, so a register copy / spill / re-use in that window would read the destination before the data
    lands.  The disassembly is checked at every build: between each such load and its `s_waitcnt vmcnt(0)` nothing touches the destination
    registers and nothing goes to scratch."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('isa_audit', os.path.join(ROOT, 'tools', 'isa_audit.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seen, bad = mod.unwaited_load_hazards()
    assert seen >= 2, seen              # the two prefetches are there (otherwise this test looks at nothing)
    assert not bad, bad


def test_c_abi_rejects_bad_configurations_before_touching_the_gpu():
    """wn_create validates the configuration (the constraints the reference asserts, plus the tiling limits of this build) before any
    HIP call, so status codes and messages can be checked on a box without a GPU.  No exception crosses the C boundary."""
    import copy
    import ctypes
    import hparams as H
    from wavenet_vocoder import _ext
    lib = _ext.load_library()

    def create(**over):
        hp = H._build()
        for k, v in over.items():
            setattr(hp, k, v)
        cfg = config(hp)
        return call(cfg)

    def config(hp):
        return _ext.config_from_hparams(hp, 1, 275)

    def call(cfg):
        h = ctypes.c_void_p()
        rc = lib.wn_create(ctypes.byref(cfg), ctypes.byref(h))
        msg = (lib.wn_last_error(None) or b'').decode()
        if rc == 0:                         # only on a GPU box
            lib.wn_destroy(h)
        return rc, msg

    ARG, SHAPE, UNSUP = -1, -2, -4
    rc, msg = create(layers=24, stacks=5)
    assert rc == SHAPE and 'multiple of stacks' in msg                      # wavenet.py:97
    rc, msg = create(out_channels=31)
    assert rc == SHAPE and 'multiple of 3' in msg                           # mixture.py:30
    rc, msg = create(input_type='mulaw-quantize', quantize_channels=256, out_channels=30)
    assert rc == SHAPE and 'quantize' in msg                                # models/__init__.py:6-9
    rc, msg = create(residual_channels=100)
    assert rc == UNSUP and 'multiples of 64' in msg
    rc, msg = create(kernel_size=5)
    assert rc == UNSUP
    rc, msg = create(wavenet_dropout=1.0)
    assert rc == ARG and 'dropout' in msg
    cfg = config(H._build())
    cfg.abi_version = 1
    rc, msg = call(cfg)
    assert rc == ARG and 'abi_version' in msg
    assert lib.wn_create(None, None) == ARG
    # WN_PIPE_DTYPE is parsed case-insensitively and an unknown spelling is an error (it used to select bf16 silently)
    old = os.environ.get('WN_PIPE_DTYPE')
    try:
        os.environ['WN_PIPE_DTYPE'] = 'fp8'
        rc, msg = create()
        assert rc == ARG and 'WN_PIPE_DTYPE' in msg
        for ok in ('FP16', 'half', 'Float16', 'BF16', 'bfloat16'):
            os.environ['WN_PIPE_DTYPE'] = ok
            rc, msg = create()
            assert rc in (0, -3) and 'WN_PIPE_DTYPE' not in msg, (ok, rc, msg)      # accepted: fails later only for want of a GPU (WN_E_HIP)
    finally:
        if old is None:
            os.environ.pop('WN_PIPE_DTYPE', None)
        else:
            os.environ['WN_PIPE_DTYPE'] = old
    assert lib.wn_synth_last_config(None, None, 0) == ARG
    # a null context is an argument error on every entry point, never a crash
    lib.wn_param_count.restype = ctypes.c_int64
    assert lib.wn_receptive_field(None) == ARG and lib.wn_param_count(None) == ARG and lib.wn_num_tensors(None) == ARG
    assert lib.wn_noise_per_step(None) == ARG and lib.wn_set_batch_parts(None, 1) == ARG and lib.wn_profile(None, 1) == ARG
    assert lib.wn_train_bwd(None, None, None) == ARG and lib.wn_pack_weights(None, None, None) == ARG
    lib.wn_destroy(None)
    # Python-side wrapper turns the status into an exception that carries it
    hp = H._build(); hp.layers = 7; hp.stacks = 2
    with pytest.raises(_ext.WnError) as ei:
        _ext.Engine(hp, 1, 275)
    assert ei.value.code == SHAPE and 'WN_E_SHAPE' in str(ei.value)


def test_c_abi_host_side_under_address_and_ub_sanitizers():
    """SURVEY section 5 (sanitizer variant): the C-ABI validation tests above, re-run in a child interpreter against the ASAN + UBSAN
    build of the library's host side (csrc/build.py --sanitize: wn_api.hip -- entry points, configuration validation, parameter table,
    workspace planning -- instrumented; device code untouched) with the sanitizer runtime preloaded.  Any heap overflow, use after free
    or undefined behaviour on those paths aborts the child."""
    sys.path.insert(0, os.path.join(ROOT, 'tacotron-2_amd', 'csrc'))
    import build as B
    rt = B.sanitizer_runtime()
    if rt is None:
        pytest.skip('clang ASAN runtime not found in this image')
    lib = B.build_sanitized(verbose=False)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS='detect_leaks=0:verify_asan_link_order=0:abort_on_error=1',
               UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1', WN_MI355_TEST_LIB=lib)
    probe = ('import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); from wavenet_vocoder import _ext; _ext.load_library(); '
             'm = open("/proc/self/maps").read(); print("SAN_LIB", "libwavenet_mi355_san.so" in m, "libclang_rt.asan" in m)' % (ROOT, os.path.join(ROOT, 'tacotron-2_amd')))
    r = subprocess.run([sys.executable, '-c', probe], capture_output=True, text=True, timeout=300, env=env)
    assert 'SAN_LIB True True' in r.stdout, r.stdout[-500:] + r.stderr[-1500:]
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_host_cpu.py'), '-x', '-q', '-p', 'no:cacheprovider',
                        '-k', 'c_abi_rejects or exports_every_declared or fails_loudly_without_gpu or dropout_mask_mirror'],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and '4 passed' in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_bench_cpu_synthesis_baseline_leg():
    """bench.py's CPU synthesis baseline: the oracle incremental loop in the reference's queue formulation and with ring
    buffers, on a bounded sample, reporting samples/s and the extrapolated real-time factor."""
    sys.path.insert(0, ROOT)
    import bench
    hp, _, _ = bench.build_hparams('c2')
    r = bench.cpu_synth_baseline(hp, steps=300, seconds_budget=3.0)
    assert r['kind'] == 'port' and r['cores'] >= 1 and r['unit'] == 'audio_samples/s'
    for form in ('reference', 'ring'):
        assert r[form]['steps'] >= 256 and r[form]['samples_per_s'] > 0
        assert abs(r[form]['rtf_extrapolated'] - hp.sample_rate / r[form]['samples_per_s']) < 1e-6 * r[form]['rtf_extrapolated']
    assert r['value'] == r['reference']['samples_per_s']


def test_feeder_ranks_take_disjoint_slices_of_the_same_batches(tmp_path, monkeypatch):
    """Data-parallel feeding: at every step the ranks hold disjoint slices of ONE global batch -- same group composition,
    same length-bucket order on every rank (so padding, hence step time, is equal) -- like tf.split over towers (wavenet.py:233-239)."""
    import hparams as H
    from wavenet_vocoder import feeder as F
    hp = H._build()
    hp.parse('hop_size=16,num_mels=16,cin_channels=16,upsample_scales=[4,4],max_time_steps=500,wavenet_batch_size=4,wavenet_test_batches=1')
    meta = _write_dataset(str(tmp_path))
    groups = {}
    for r in range(2):
        monkeypatch.setattr(F, '_ranks', lambda r=r: (r, 2))
        fd = F.Feeder(None, meta, str(tmp_path), hp, device=torch.device('cpu'))
        groups[r] = fd._next_group(train=True) + fd._next_group(train=True)        # two groups: the order rng advances identically
    monkeypatch.setattr(F, '_ranks', lambda: (0, 1))
    whole = F.Feeder(None, meta, str(tmp_path), hp, device=torch.device('cpu'))
    ref = whole._next_group(train=True) + whole._next_group(train=True)
    assert len(groups[0]) == len(groups[1]) == len(ref) == 128
    for b0, b1, b in zip(groups[0], groups[1], ref):
        assert len(b0) == len(b1) == 2 and len(b) == 4
        for got, want in zip(b0 + b1, b):                                              # rank 0 = first half, rank 1 = second half
            assert got[3] == want[3] and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # an indivisible global batch is refused like the reference does (feeder.py:267-268)
    monkeypatch.setattr(F, '_ranks', lambda: (0, 3))
    with pytest.raises(ValueError):                                                       # at construction, not inside a feeder thread
        F.Feeder(None, meta, str(tmp_path), hp, device=torch.device('cpu'))


def test_feeder_pinned_rings_are_per_producer(tmp_path, monkeypatch):
    """The page-locked staging buffers are a ring of whole-batch SLOTS per (producer, shapes of the batch's tensors).  The train queue keeps
    up to 8 batches + the one being built + the consumer's two alive; a burst of same-shaped eval batches (eval queue depth 1, but one
    batch per eval step) must not lap them."""
    import hparams as H
    from wavenet_vocoder import feeder as F
    hp = H._build()
    hp.parse('hop_size=16,num_mels=16,cin_channels=16,upsample_scales=[4,4],max_time_steps=500,wavenet_batch_size=4,wavenet_test_batches=1')
    fd = F.Feeder(None, _write_dataset(str(tmp_path)), str(tmp_path), hp, device=torch.device('cpu'))
    monkeypatch.setattr(F, '_new_pinned', lambda shape, dtype: torch.empty(shape, dtype=dtype))
    one = lambda v: (np.full((4, 7), float(v), np.float32),)
    live = [fd._pinned_batch(one(i), True)[0] for i in range(12)]          # the train side's live set
    for i in range(40):                                                       # an eval phase of the same shape
        fd._pinned_batch(one(-1.0 - i), False)
    for i, t in enumerate(live):
        assert torch.equal(t, torch.full((4, 7), float(i))), i
    # the ring itself wraps after _PIN_RING slots of one producer (bounded page-locked memory)
    again = [fd._pinned_batch(one(100.0 + i), True)[0] for i in range(fd._PIN_RING)]
    assert again[4].data_ptr() == live[0].data_ptr() and len({t.data_ptr() for t in live + again}) == fd._PIN_RING


def test_feeder_pinned_ring_takes_a_whole_batch_per_slot(tmp_path, monkeypatch):
    """a mulaw-quantize batch has inputs and targets of the SAME dtype and shape (int32 [B, T]).  With a ring per tensor
    shape each batch took two buffers of one ring, the ring covered 8 batches, and with the train queue full (8 queued + the one being
    built + the consumer's three) batch k + 8 was staged into the x / y buffers of the still-queued batch k while its lengths and mel
    conditioning stayed batch k's.  A slot now holds one whole batch: 16 batches of identical tensor shapes are 16 disjoint buffer sets."""
    import hparams as H
    from wavenet_vocoder import feeder as F
    hp = H._build()
    hp.parse('hop_size=16,num_mels=16,cin_channels=16,upsample_scales=[4,4],max_time_steps=500,wavenet_batch_size=4,wavenet_test_batches=1,'
             'input_type=mulaw-quantize,quantize_channels=256,out_channels=256')
    fd = F.Feeder(None, _write_dataset(str(tmp_path)), str(tmp_path), hp, device=torch.device('cpu'))
    monkeypatch.setattr(F, '_new_pinned', lambda shape, dtype: torch.empty(shape, dtype=dtype))

    def batch(k):       # the tensor shapes _prepare_batch emits for mulaw-quantize: x and y identical int32 [B, T]; lengths; mel c
        return (np.full((4, 96), k, np.int32), np.full((4, 96), 1000 + k, np.int32), np.full((4,), k, np.int32), np.full((4, 16, 6), float(k), np.float32))

    held = [fd._pinned_batch(batch(k), True) for k in range(fd._PIN_RING)]       # queue full and then some: every batch still referenced
    ptrs = [t.data_ptr() for b in held for t in b]
    assert len(set(ptrs)) == 4 * fd._PIN_RING                                   # no buffer is shared between positions or batches
    for k, b in enumerate(held):
        assert int(b[0][0, 0]) == k and int(b[1][0, 0]) == 1000 + k and int(b[2][0]) == k and float(b[3][0, 0, 0]) == float(k), k
    # the real producer path: batches prepared from the dataset keep inputs / targets / lengths / conditioning of ONE batch together
    prepared = [fd._prepare_batch(b) for b in list(fd._iter_group(train=True))[:10]]
    pinned = [fd._pin(p, True) if torch.cuda.is_available() else fd._pinned_batch(p, True) for p in prepared]
    for p, q in zip(prepared, pinned):
        for a, t in zip(p, q):
            assert (a is None and t is None) or np.array_equal(a, t.numpy())


def test_feeder_thread_errors_reach_the_training_loop(tmp_path):
    """A failure inside a background feeder thread (here: a mel file of the wrong length) is re-raised by next_train_batch -- AFTER the
    batches that were already queued -- instead of leaving the training loop blocked on an empty queue.  The producer must NOT stop the
    coordinator itself: the loop tests coord.should_stop() before every step and would leave silently with good batches still queued
    (ADVICE round 2); the loop owns that decision."""
    import hparams as H
    from wavenet_vocoder import feeder as F
    from wavenet_vocoder.train import _Coordinator
    hp = H._build()
    hp.parse('hop_size=16,num_mels=16,cin_channels=16,upsample_scales=[4,4],max_time_steps=500,wavenet_batch_size=4,wavenet_test_batches=1')
    meta = _write_dataset(str(tmp_path))
    rows = [l.strip().split('|') for l in open(meta)]
    bad = os.path.join(str(tmp_path), rows[7][1])
    np.save(bad, np.load(bad)[:-3])                                                        # audio / mel length mismatch in one utterance
    coord = _Coordinator()
    fd = F.Feeder(coord, meta, str(tmp_path), hp, device=torch.device('cpu'))
    fd.start_threads()
    # the broken utterance lands in the train or the test split: the error belongs to THAT producer's queue (its consumer reaches it
    # behind the good batches); the other queue keeps delivering
    raised = []
    for nxt, n in ((fd.next_train_batch, 200), (fd.next_eval_batch, 3)):
        try:
            for _ in range(n):
                nxt()
        except RuntimeError as e:
            assert 'feeder thread failed' in str(e)
            raised.append(nxt.__name__)
    assert len(raised) == 1, raised
    assert not coord.should_stop() and [type(e) for e in fd._errors.values()] == [ValueError]
    coord.request_stop()


def test_dropout_seed_is_per_rank_and_single_gpu_compatible():
    from wavenet_vocoder.models.wavenet import dropout_seed
    assert dropout_seed(5339, 17) == 5339 * 1000003 + 17 == dropout_seed(5339, 17, rank=0)
    seeds = {dropout_seed(5339, s, r) for s in range(100) for r in range(8)}
    assert len(seeds) == 800 and all(0 <= v < 2 ** 64 for v in seeds)


def test_tf_checkpoint_reader_on_independent_fixture(golden_dir, tmp_path):
    """the TF-1 tensor-bundle reader against a fixture assembled from the format documents by an independent script
    (oracle/gen_tf_bundle_fixture.py -- shares no code with the reader or its writer), incl. checksum verification and the mapping of
    the reference's EMA-shadow variable names (train.py:75-83) onto engine tensor names."""
    import shutil
    from wavenet_vocoder import tf_checkpoint as T
    # CRC-32C known answers (RFC 3720 B.4) and LevelDB's mask
    assert T.crc32c(b'123456789') == 0xE3069283 and T.crc32c(bytes(32)) == 0x8A9136AA and T.crc32c(bytes([0xff] * 32)) == 0x62A8AB43
    assert T._mask_crc(0) == 0xa282ead8
    src = os.path.join(golden_dir, 'tf_bundle')
    exp = np.load(os.path.join(src, 'expected.npz'))
    got = T.load_checkpoint(os.path.join(src, 'fixture.ckpt-7'))
    assert set(got) == set(exp.files)
    for k in exp.files:
        assert got[k].dtype == exp[k].dtype and got[k].shape == exp[k].shape and np.array_equal(got[k], exp[k]), k
    assert int(got['global_step']) == 7
    layout = {'final_convolution_1/kernel': ((1, 8, 8), 0), 'final_convolution_1/bias': ((8,), 64)}
    flat, step, missing = T.load_reference_checkpoint(os.path.join(src, 'fixture.ckpt-7'), layout)
    assert step == 7 and not missing
    assert np.array_equal(flat[:64], exp['WaveNet_model/inference/final_convolution_1/kernel/ExponentialMovingAverage'].reshape(-1))
    assert np.array_equal(flat[64:72], exp['WaveNet_model/inference/final_convolution_1/bias/ExponentialMovingAverage'])
    # corruption is detected: one flipped bit in the index, one in the data shard
    for name, pos in (('fixture.ckpt-7.index', 40), ('fixture.ckpt-7.data-00000-of-00001', 17)):
        d = tmp_path / ('bad_' + name.split('.')[-1].split('-')[0])
        shutil.copytree(src, str(d))
        b = bytearray(open(os.path.join(str(d), name), 'rb').read()); b[pos] ^= 0x10
        open(os.path.join(str(d), name), 'wb').write(bytes(b))
        with pytest.raises(ValueError, match='checksum mismatch'):
            T.load_checkpoint(os.path.join(str(d), 'fixture.ckpt-7'))


def test_reconstruction_mel_front_end():
    """datasets/audio.py melspectrogram (reference datasets/audio.py:62-68; librosa's stft / filters.mel restated): STFT frames
    against a direct DFT, librosa's frame count, Slaney filters of unit area, and a pure tone landing in the right mel band,
    normalised into [-max_abs_value, max_abs_value]."""
    import hparams as H
    from datasets.audio import _build_mel_basis, _mel_to_hz, _hz_to_mel, _stft, melspectrogram
    hp = H._build()
    rng = np.random.RandomState(0)
    y = rng.randn(3000)
    D = _stft(y, hp)
    n_fft, hop, win = hp.n_fft, hp.hop_size, hp.win_size
    assert D.shape == (1 + n_fft // 2, 1 + len(y) // hop)                         # centred frames (librosa.stft)
    from scipy.signal.windows import hann
    w = np.zeros(n_fft); lp = (n_fft - win) // 2; w[lp:lp + win] = hann(win, sym=False)
    yp = np.pad(y, n_fft // 2)
    k = np.arange(n_fft // 2 + 1)[:, None]; n = np.arange(n_fft)[None, :]
    for fr in (0, 5, D.shape[1] - 1):
        ref = ((yp[fr * hop:fr * hop + n_fft] * w)[None, :] * np.exp(-2j * np.pi * k * n / n_fft)).sum(1)
        assert np.abs(ref - D[:, fr]).max() < 1e-9
    mb = _build_mel_basis(hp)
    assert mb.shape == (hp.num_mels, 1 + n_fft // 2) and (mb >= 0).all()
    area = mb.sum(1) * hp.sample_rate / n_fft
    assert np.abs(area[10:] - 1.0).max() < 0.02                                   # 'slaney' normalisation: unit area per filter
    assert np.allclose(_mel_to_hz(_hz_to_mel(np.array([55.0, 440.0, 1000.0, 7600.0]))), [55.0, 440.0, 1000.0, 7600.0])
    centres = _mel_to_hz(np.linspace(_hz_to_mel(hp.fmin), _hz_to_mel(hp.fmax), hp.num_mels + 2))[1:-1]
    t = np.arange(hp.sample_rate) / hp.sample_rate
    m = melspectrogram(0.5 * np.sin(2 * np.pi * 1000.0 * t), hp)
    assert m.shape[0] == hp.num_mels and m.min() >= -hp.max_abs_value and m.max() <= hp.max_abs_value
    assert abs(int(m[:, 40].argmax()) - int(np.abs(centres - 1000.0).argmin())) <= 1


def test_dropout_mask_mirror_matches_the_library_hash():
    """tests/hip_util.py's numpy mirrors of the device dropout mask (what the parity tests hand to the oracle) against the
    library's own hash evaluated on the host (wn_test_dropout_mask: the same inline wn_layer_key / wn_drop_quad the kernels use):
    whole masks, row windows that start inside the tensor, several layers / seeds / rates."""
    import ctypes
    from wavenet_vocoder import _ext
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from hip_util import dropout_mask, dropout_mask_rows
    lib = _ext.load_library()
    for seed, layer, p, rows, R in ((1234, 0, 0.05, 40, 256), (5339 * 1000003 + 17, 23, 0.05, 24, 64), (7, 5, 0.5, 16, 128), (2 ** 40 + 3, 11, 0.2, 8, 512)):
        got = np.zeros(rows * R, dtype=np.uint8)
        assert lib.wn_test_dropout_mask(ctypes.c_uint64(seed), layer, ctypes.c_float(p), 0, rows * R, got.ctypes.data) == 0
        ref = dropout_mask(seed, layer, rows, R, p)
        assert np.array_equal(got.reshape(rows, R).astype(np.float32), ref)
        win = dropout_mask_rows(seed, layer, 4, rows - 6, R, p)
        assert np.array_equal(win, ref[4:rows - 2])
        assert abs(float(ref.mean()) - (1.0 - p)) < 0.03
    bad = np.zeros(4, dtype=np.uint8)
    assert lib.wn_test_dropout_mask(ctypes.c_uint64(1), 0, ctypes.c_float(1.0), 0, 4, bad.ctypes.data) == -1      # WN_E_ARG


def test_projector_and_reconstruction_mel_outputs(tmp_path):
    """Host-side logging outputs of the train driver that need no GPU: the speaker-embedding projector files (reference
    train.py:26-39: config + tensor + metadata) and the "Local Condition vs Reconst. Mel-Spectrogram" plot (train.py:110-116)."""
    import hparams as H
    from wavenet_vocoder.train import add_embedding_stats, _plot_reconstruction_mel
    hp = H._build()
    tb = str(tmp_path)
    table = torch.arange(15, dtype=torch.float32).reshape(5, 3) * 0.25
    add_embedding_stats(tb, ['WaveNet_model/inference/gc_embedding'], ['../metas/SpeakerEmbeddings.tsv'], [table], 42)
    cfg = open(os.path.join(tb, 'projector_config.pbtxt')).read()
    assert 'tensor_name: "WaveNet_model/inference/gc_embedding"' in cfg and 'metadata_path: "../metas/SpeakerEmbeddings.tsv"' in cfg
    tsv = [l.split('"')[1] for l in cfg.split('\n') if 'tensor_path' in l][0]
    assert tsv.endswith('-42.tsv')
    np.testing.assert_allclose(np.loadtxt(os.path.join(tb, tsv), delimiter='\t'), table.numpy())
    # mel plot: one second of a tone against a random conditioning of the same length, [cin, Tc] and [Tc, cin] orientations
    t = np.arange(hp.sample_rate) / hp.sample_rate
    wav = 0.4 * np.sin(2 * np.pi * 300.0 * t)
    frames = 1 + len(wav) // hp.hop_size
    cond = np.random.RandomState(0).uniform(0, 1, size=(hp.cin_channels, frames)).astype(np.float32)
    for k, c in enumerate((cond, cond.T, torch.from_numpy(cond))):
        path = os.path.join(tb, 'mel-%d.png' % k)
        _plot_reconstruction_mel(wav, c, path, hp, 'title')
        assert os.path.getsize(path) > 10000


def test_bench_algorithmic_work_matches_the_scope_table():
    """bench.py's per-sample work (the numerators of its roofline fields) against the figures SURVEY.md 8(d) states: MACs per sample
    (C2 13 639 424, default hparams 3 047 808, C5 65 635 840) and algorithmic HBM bytes per sample in bf16 (232 890 / 101 778 /
    567 378); train FLOPs = 6 x MAC (81.84 MFLOP at C2)."""
    sys.path.insert(0, ROOT)
    import bench
    want = {'c2': (13639424, 232890), 'default_hparams': (3047808, 101778), 'c5_stress': (65635840, 567378)}
    for w, (mac, nbytes) in want.items():
        hp, B, T = bench.build_hparams(w)
        assert bench.mac_per_sample(hp) == mac, (w, bench.mac_per_sample(hp))
        assert bench.alg_bytes_per_sample(hp) == nbytes, (w, bench.alg_bytes_per_sample(hp))
    hp, B, T = bench.build_hparams('c2')
    assert (B, T) == (8, 11000) and abs(6.0 * bench.mac_per_sample(hp) / 1e6 - 81.84) < 0.01


def test_bench_traffic_constants_come_from_the_committed_pmc_summary():
    """bench.py's `traffic` / `traffic_per_step` are not hand-copied constants any more -- they are loaded from
    profiles/traffic.json, which tools/pmc_summary.py --traffic writes from the FETCH_SIZE / WRITE_SIZE passes.  Recompute the summary from
    the committed per-kernel tables it names and require the same numbers; another batch geometry gets None, not a scaled guess."""
    import json
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import bench
    import pmc_summary
    t = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
    src = [os.path.join(ROOT, p) for p in t['sources']]
    assert all(os.path.exists(p) for p in src), src
    again = pmc_summary.traffic_summary(src[0], src[1])
    got = bench.load_traffic('c2', 8, 11000)
    assert got is not None
    for k in ('bytes_per_step', 'fetch_x2_bytes_per_step', 'write_bytes_per_step', 'gate_bytes_per_launch'):
        assert abs(again[k] - t[k]) <= 2e-3 * t[k], (k, again[k], t[k])            # (the .md tables keep 4 significant digits)
        assert got[k] == t[k]
    assert abs(sum(v['bytes_per_step'] for v in t['kernels'].values()) - t['bytes_per_step']) < 1.0
    assert 20e9 < t['bytes_per_step'] < 60e9 and t['gate_launches_per_step'] == 48
    assert bench.load_traffic('c2', 4, 11000) is None and bench.load_traffic('c5_stress', 8, 12000) is None


def test_device_timeline_summary_of_in_kernel_stamps(tmp_path):
    """tools/devtrace.py (what bench.py's `device_timeline` block and the WN_DEVTRACE file view are made of): phases and overlap of a
    hand-made two-stream step -- forward until the first backward launch, backward chain until the last d x, tail until the last weight
    gradient; launches that never ran (start stamp still ~0ull) are ignored."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import devtrace
    T0 = 5_000_000
    rows = [(0, 'a', T0, T0 + 6000), (0, 'b', T0 + 10, T0 + 7000), (1, 'a', T0 + 6000, T0 + 10000), (1, 'b', T0 + 7000, T0 + 10000),      # forward: 100 us
            (4, 'b', T0 + 10000, T0 + 11000), (4, 'a', T0 + 10500, T0 + 11500), (3, 'a', T0 + 11500, T0 + 15000), (5, 'a', T0 + 15000, T0 + 20000),
            (3, 'b', T0 + 11000, T0 + 14000), (5, 'b', T0 + 14000, T0 + 19000),                                                           # chain ends at 200 us
            (103, 'c', T0 + 20000, T0 + 29000), (101, 'c', T0 + 29000, T0 + 30000), (2, 'a', T0 + 20000, T0 + 25000),
            (102, 'c', devtrace.NEVER, 0),
            (206, 'a', T0 + 30000, T0 + 31000)]                     # kinds >= 200: kernel groups bracketed by stamp kernels (here: the optimiser)
    s = devtrace.summarise(rows)
    assert s['launches'] == 14 and abs(s['span_us'] - 310.0) < 1e-9 and abs(s['after_last_weight_gradient_us'] - 10.0) < 1e-9
    assert abs(s['forward_us'] - 100.0) < 1e-9 and abs(s['backward_chain_us'] - 100.0) < 1e-9 and abs(s['weight_gradient_tail_us'] - 100.0) < 1e-9
    assert abs(sum(s['in_flight_us'].values()) - 310.0) < 1e-6 and abs(s['in_flight_us']['1'] - (0.1 + 5 + 10 + 40 + 10 + 10)) < 1e-6
    assert [st['launches'] for st in s['streams']] == [7, 5, 2] and s['streams'][1]['first_backward_start_us'] == 100.0
    f = tmp_path / 'trace.txt'
    f.write_text('# idx epi stream rows start end\n' + ''.join('%d %d %s 1 %d %d\n' % (i, r[0], r[1], r[2], r[3]) for i, r in enumerate(rows)))
    assert devtrace.summarise(devtrace.parse_file(str(f))) == s


def test_synthesis_driver_work_list_and_index_file(tmp_path, monkeypatch):
    """wavenet_vocoder/synthesize.py (reference synthesize.py:14-82) with a stand-in Synthesizer: mel files in sorted order, chunks of
    wavenet_synthesis_batch_size, speaker ids from --speaker_id; Tacotron-2 mode reads text|mel|speaker rows of the evaluation map.txt;
    the index rows keep ALL their columns (the reference's format strings drop the last one)."""
    import types
    import hparams as H
    from wavenet_vocoder import synthesize as S
    d = tmp_path / 'mels'; d.mkdir()
    for i in (2, 0, 1, 3, 4):
        np.save(str(d / ('mel-%d.npy' % i)), np.zeros((3, 80), np.float32))
    (d / 'notes.txt').write_text('not a mel')
    calls = []

    class FakeSynth:
        def load(self, ckpt, hp):
            self.ckpt = ckpt

        def synthesize(self, mels, speakers, names, wav_dir, plot_dir):
            calls.append((len(mels), speakers, names))
            return [os.path.join(wav_dir, 'wavenet-audio-%s.wav' % n) for n in names]
    monkeypatch.setattr(S, 'Synthesizer', FakeSynth)
    hp = H._build(); hp.parse('wavenet_synthesis_batch_size=2')
    out = str(tmp_path / 'out')
    S.run_synthesis(types.SimpleNamespace(model='WaveNet', mels_dir=str(d), speaker_id='0,1, 2,0,1'), 'ckpt', out, hp)
    assert [c[0] for c in calls] == [2, 2, 1] and calls[0][1] == ['0', '1'] and calls[2][2] == ['mel-4']
    rows = [l.split('|') for l in open(os.path.join(out, 'wavs', 'map.txt')).read().strip().split('\n')]
    assert len(rows) == 5 and all(len(r) == 3 for r in rows)
    assert rows[0] == [str(d / 'mel-0.npy'), os.path.join(out, 'wavs', 'wavenet-audio-mel-0.wav'), '0'] and rows[4][2] == '1'
    assert os.path.isdir(os.path.join(out, 'plots'))
    # Tacotron-2 mode: the evaluation map.txt, no global conditioning
    calls.clear()
    (d / 'map.txt').write_text('hello world|%s|<no_g>\nsecond|%s|<no_g>\n' % (d / 'mel-1.npy', d / 'mel-3.npy'))
    S.run_synthesis(types.SimpleNamespace(model='Tacotron-2', mels_dir=str(d), speaker_id=None), 'ckpt', out, hp)
    assert calls == [(2, None, ['mel-1', 'mel-3'])]
    rows = [l.split('|') for l in open(os.path.join(out, 'wavs', 'map.txt')).read().strip().split('\n')]
    assert rows[0][0] == 'hello world' and rows[1][3] == '<no_g>' and all(len(r) == 4 for r in rows)
    with pytest.raises(RuntimeError, match='Failed to load checkpoint'):
        S.wavenet_synthesize(types.SimpleNamespace(model='WaveNet', mels_dir=str(d), speaker_id=None, output_dir='o/'), hp, str(tmp_path / 'nowhere'))


def test_pipeline_workgroup_tables_keep_layers_on_one_xcd_and_heads_next_to_layer_0():
    """Host logic of the persistent synthesis pipeline (csrc/wn_synth_pipe.hip pipe_layout, through the no-GPU hook wn_test_pipe_layout): which
    workgroup plays which CU.  Block b runs on XCD b % 8; a hand-off between CUs of one XCD goes through that XCD's L2 with plain stores, so the
    tables must keep a layer's P CUs on ONE XCD and consecutive layers together, put the head(s) of a single-instance run on XCD 0 next to
    layer 0, never exceed an XCD's 32 CUs, and be exact inverses of each other.  Paper model (24 x 8: once, two heads), hparams.py's own
    (20 x 4: three instances in ONE launch of 256 workgroups), the 6-layer test models, and a model that does not fit three times."""
    import ctypes
    from wavenet_vocoder import _ext
    lib = _ext.load_library()

    def layout(L, P, ni):
        role = np.full(1024, -7, dtype=np.int32); blk = np.full(1024, -7, dtype=np.int32)
        grid = ctypes.c_int32(0); heads = ctypes.c_int32(0)
        rc = lib.wn_test_pipe_layout(L, P, ni, role.ctypes.data, 1024, blk.ctypes.data, 1024, ctypes.byref(grid), ctypes.byref(heads))
        return rc, role[:grid.value].copy(), blk, grid.value, heads.value

    for L, P, ni_max in ((24, 8, 1), (20, 4, 3), (6, 8, 3), (6, 4, 3), (30, 2, 3), (12, 8, 2)):
        rc, role, blk, grid, nh = layout(L, P, 1)
        assert rc == ni_max, (L, P, rc)
        spx = (L + 7) // 8
        assert grid == 8 * (spx * P + nh) and nh == max(1, min(2, 30 - spx * P))
        seen = {}
        for b, r in enumerate(role):
            if r < 0:
                assert r == -1
                continue
            assert (r >> 24) == 0
            if (r >> 23) & 1:
                h = r & 0xff
                assert b % 8 == 0 and h < nh and blk[L * P + h] == b          # heads: XCD 0, where layer 0 lives
                seen[('h', h)] = b
            else:
                l, j = (r >> 8) & 0xff, r & 0xff
                assert l < L and j < P and blk[l * P + j] == b and b % 8 == l // spx      # spx consecutive layers per XCD
                seen[(l, j)] = b
        assert len(seen) == L * P + nh and len(set(seen.values())) == len(seen)
        assert max(np.bincount(np.nonzero(role >= 0)[0] % 8, minlength=8)) <= 30
        for ni in range(2, ni_max + 1):
            rc, role, blk, grid, nh1 = layout(L, P, ni)
            assert rc == ni_max and grid == 256 and nh1 == 1
            per = L * P + 1
            used = set()
            for i in range(ni):
                for l in range(L):
                    bs = [int(blk[i * per + l * P + j]) for j in range(P)]
                    assert all(0 <= b < 256 for b in bs) and len({b % 8 for b in bs}) == 1          # a layer never splits over XCDs
                    for j, b in enumerate(bs):
                        assert role[b] == (i << 24) | (l << 8) | j and b not in used
                        used.add(b)
                hb = int(blk[i * per + L * P])
                assert role[hb] == (i << 24) | (1 << 23) and hb not in used
                used.add(hb)
            assert len(used) == ni * per == int((role >= 0).sum())
            assert max(np.bincount(np.array(sorted(used)) % 8, minlength=8)) <= 32
        if ni_max < 3:
            assert layout(L, P, ni_max + 1)[0] == -3 or layout(L, P, ni_max + 1)[0] < 0          # WN_E_SHAPE: does not fit
    assert layout(0, 8, 1)[0] < 0 and layout(24, 9, 1)[0] < 0 and layout(24, 8, 4)[0] < 0
