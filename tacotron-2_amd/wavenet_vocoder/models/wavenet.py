"""WaveNet vocoder model -- host façade over the HIP engine.

Mirrors the public surface of the reference's ``wavenet_vocoder/models/wavenet.py:WaveNet`` (``initialize``,
``add_loss``, ``add_optimizer``, ``step``, ``incremental``, ``receptive_field``, ``variables``, the ``tower_*``
result lists) but is EAGER: where the reference builds TF graph nodes that ``session.run`` evaluates later,
these methods enqueue the HIP kernels immediately on the current stream and return device tensors.

Data-parallel training is one process per GPU (torch.distributed, backend "nccl" == RCCL): every rank owns
an identical replica, ``add_optimizer`` all-reduces the flat fp32 gradient (mean over ranks == the tower
average of wavenet.py:560-575) and every rank applies the same clip / Adam / EMA update.
"""
import numpy as np
import torch

from datasets import audio
from infolog import log
from wavenet_vocoder import _ext, util
from wavenet_vocoder.parallel import allreduce_loss_and_flags, allreduce_mean_buckets_
from wavenet_vocoder.util import is_mulaw, is_mulaw_quantize, is_scalar_input

from .modules import initialize_parameters, receptive_field_size


def dropout_seed(random_seed, global_step, rank=0):
    """64-bit key of a training step's dropout masks.  Every rank draws its own masks, like the reference's towers each own a
    tf.layers.dropout op (modules.py:484); rank 0 of any world size equals the single-GPU stream."""
    return (int(random_seed) * 1000003 + int(global_step) + int(rank) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF


class WaveNet(object):
    def __init__(self, hparams, init=False):
        self._hparams = hparams
        if self.local_conditioning_enabled():
            assert hparams.num_mels == hparams.cin_channels
        assert hparams.layers % hparams.stacks == 0
        if hparams.gin_channels > 0 and hparams.use_speaker_embedding:
            assert hparams.n_speakers is not None                                    # wavenet.py:154
        self.scalar_input = is_scalar_input(hparams.input_type)
        self.receptive_field = receptive_field_size(hparams.layers, hparams.stacks, hparams.kernel_size)
        self.embed_speakers = 'gc_embedding' if (hparams.gin_channels > 0 and hparams.use_speaker_embedding) else None
        self.engine = None
        self.is_training = False
        self.is_evaluating = False
        self.global_step = 0
        self._world = 1
        self._dist = None

    # ------------------------------------------------------------------ construction
    def build(self, max_batch, max_time, device=None, params=None, inference_only=False):
        """Allocate the engine (packed weights + workspace) and the flat fp32 parameter / optimiser buffers.  inference_only: a
        synthesis-only engine (no training workspace: ~1.5 KB instead of ~45 KB of HBM per (stream x sample))."""
        hp = self._hparams
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = device
        hop = audio.get_hop_size(hp)
        max_time = (int(max_time) + hop - 1) // hop * hop
        self.engine = _ext.Engine(hp, max_batch, max_time, inference_only=inference_only)
        self.inference_only = bool(inference_only)
        self.max_batch, self.max_time = max_batch, max_time
        if params is None:
            params = initialize_parameters(hp, self.engine.layout)
        self.params = params.to(device).contiguous()
        assert self.params.numel() == self.engine.n_params
        if inference_only:                                  # synthesis only: no gradient / Adam slots (4 x 55 MB at the paper shape)
            self.grads = self.adam_m = self.adam_v = torch.zeros(0, device=device)
        else:
            self.grads = torch.zeros_like(self.params)
            self.adam_m = torch.zeros_like(self.params)
            self.adam_v = torch.zeros_like(self.params)
        self.ema_params = self.params.clone()                # tf.train.ExponentialMovingAverage shadow (wavenet.py:473)
        self.variables = self.engine.views(self.params)      # name -> tensor view (TF layouts)
        self.gradients = None if inference_only else self.engine.views(self.grads)
        self._loss_dev = torch.zeros(1, device=device)
        self._dirty = True
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self._dist = torch.distributed
            self._world = self._dist.get_world_size()
            self._dist.broadcast(self.params, 0)
            self.ema_params.copy_(self.params)
        log('Initializing Wavenet model.  Dimensions: ')
        log('  Receptive Field:           ({} samples / {:.1f} ms)'.format(self.receptive_field, self.receptive_field / hp.sample_rate * 1000.))
        n = sum(int(np.prod(s)) for s, _ in self.engine.layout.values())
        log('  WaveNet Parameters:        {:.3f} Million.'.format(n / 1000000))
        return self

    def _ensure_packed(self):
        if self._dirty:
            self.engine.pack_weights(self.ema_params if getattr(self, '_pack_ema', False) else self.params)
            self._dirty = False

    def use_ema_weights(self, enable=True):
        """Synthesis / evaluation with the averaged weights (what the reference intended with its shadow saver, train.py:75-83):
        from now on the engine packs ``ema_params`` instead of ``params`` (training forwards included, so only switch it on for a
        synthesis-only model or switch it off again)."""
        self._pack_ema = bool(enable)
        self._dirty = True

    # ------------------------------------------------------------------ reference-shaped API
    def initialize(self, y, c, g, input_lengths, x=None, synthesis_length=None, test_inputs=None, split_infos=None):
        """Train (x given), eval (y given, x None) or synthesis (neither) -- wavenet.py:218-473."""
        hp = self._hparams
        self.is_training = x is not None
        self.is_evaluating = not self.is_training and y is not None
        self.tower_y_hat, self.tower_y_target, self.tower_synth_upsampled_local_features = [], [], []
        if self.is_training:
            B, T = int(x.shape[0]), int(x.shape[-1])
            if self.engine is None:
                self.build(B, T)
            self._ensure_packed()
            rank = self._dist.get_rank() if (self._dist is not None and self._world > 1) else 0
            self._seed = dropout_seed(hp.wavenet_random_seed, self.global_step, rank)
            self.tower_y, self.tower_input_lengths, self.tower_c = [y], [input_lengths], [c]
            self._y_hat_train = None
            self._set_global(g, B)
            self.engine.train_fwd(x.contiguous(), c.contiguous(), y.contiguous(), input_lengths, self._seed, self._loss_dev)
            self._have_fwd = True
            return
        if self.is_evaluating:
            # item 0 of the batch, teacher forced unless wavenet_natural_eval (wavenet.py:342-405)
            idx = 0
            length = int(input_lengths[idx])
            hop = audio.get_hop_size(hp)
            length = length // hop * hop
            y0 = y[idx].reshape(-1)[:length]
            c0 = c[idx:idx + 1, :, :length // hop].contiguous()
            ti = None if hp.wavenet_natural_eval else y0.reshape(1, -1).contiguous()
            g0 = None if g is None else torch.as_tensor(g)[idx:idx + 1]
            out, raw = self.incremental(None, c=c0, g=g0, time_length=length, test_inputs=ti, return_raw=True, check=True)
            tgt = y0.reshape(1, -1)
            ln = torch.tensor([length], dtype=torch.int32, device=raw.device)
            self.engine.loss(raw, tgt.contiguous(), ln, 0, self._loss_dev)          # no shift: wavenet.py:497-506
            self.eval_loss = self._loss_dev.clone()
            y_hat = out.reshape(-1)
            y_target = y0
            if is_mulaw_quantize(hp.input_type):
                y_hat = util.inv_mulaw_quantize(y_hat); y_target = util.inv_mulaw_quantize(y_target)
            elif is_mulaw(hp.input_type):
                y_hat = util.inv_mulaw(y_hat); y_target = util.inv_mulaw(y_target.float())
            self.tower_y_hat.append(y_hat)
            self.tower_y_target.append(y_target)
            self.tower_eval_c = [c0[0]]
            self.tower_eval_upsampled_local_features = [self.upsampled_local_features[0]]
            return
        # synthesis: c arrives [B, Tc, num_mels] like the reference's placeholder (wavenet.py:408-465)
        assert c is not None, 'local conditioning is required'
        if c.dim() != 3:
            raise ValueError('Expected 3 dimension shape [batch_size(1), time_length, {}] for local condition features but found {}'.format(
                hp.cin_channels, tuple(c.shape)))
        cT = c.transpose(1, 2).contiguous()
        out = self.incremental(None, c=cT, g=g, time_length=None, test_inputs=test_inputs, check=True)
        if is_mulaw_quantize(hp.input_type):
            y_hat = util.inv_mulaw_quantize(out)
        elif is_mulaw(hp.input_type):
            y_hat = util.inv_mulaw(out)
        else:
            y_hat = out
        self.tower_y_hat.append(y_hat)
        self.tower_synth_upsampled_local_features.append(self.upsampled_local_features)

    def add_loss(self, flags=None):
        """wavenet.py:476-519.  The masked loss is fused into the forward call; this exposes it.  Data parallel: the reported loss is
        the mean of the per-tower losses (wavenet.py:515-516; logging and the NaN guard only -- gradients never see it), and the
        training loop's per-step flags (``flags``: a float vector, e.g. "my feeder failed") ride in the SAME small all-reduce:
        ``self.reduced_flags`` = how many ranks raised each."""
        if self.is_training:
            self.tower_loss = [self._loss_dev]
            if flags is None:
                flags = self._loss_dev.new_zeros(0)
            if self._dist is not None and self._world > 1:
                vec = allreduce_loss_and_flags(self._loss_dev, flags)
                self.loss, self.reduced_flags = vec[:1], vec[1:]
            else:
                self.loss, self.reduced_flags = self._loss_dev, flags
            return self.loss
        if self.is_evaluating:
            return self.eval_loss
        raise RuntimeError('Model not in train/eval mode but computing loss: Where did this go wrong?')

    def learning_rate_at(self, global_step):
        hp = self._hparams
        return _ext.learning_rate(hp.wavenet_lr_schedule, hp.wavenet_learning_rate, global_step,
                                  hp.wavenet_decay_rate, hp.wavenet_decay_steps, hp.wavenet_warmup)

    def add_optimizer(self, global_step=None):
        """Backward + tower-gradient mean (RCCL all-reduce) + clip + Adam + EMA -- wavenet.py:522-613."""
        if not getattr(self, '_have_fwd', False):
            raise RuntimeError('add_optimizer needs a training-mode initialize() first')
        step = self.global_step if global_step is None else int(global_step)
        self.engine.train_bwd(self.grads)
        allreduce_mean_buckets_(self.engine, self.grads)      # per bucket, as soon as it is final: overlaps the rest of the backward
        self.learning_rate = self.learning_rate_at(step)
        self.engine.optim_step(self.params, self.grads, self.adam_m, self.adam_v, self.ema_params, self.learning_rate, step)
        self._dirty = True
        self._have_fwd = False
        self.global_step = step + 1
        self.optimize = True
        return self.global_step

    def get_mask(self, input_lengths, maxlen=None):
        expand = not is_mulaw_quantize(self._hparams.input_type)
        mask = util.sequence_mask(input_lengths, max_len=maxlen, expand=expand)
        return mask[:, 1:] if not expand else mask[:, 1:, :]

    def _set_global(self, g, B):
        """Hand the global conditioning of this batch to the engine (wavenet.py:669-678): speaker ids [B] / [B,1] when the
        model owns an embedding table, else features [B, gin_channels]."""
        if not self.global_conditioning_enabled():
            return
        if g is None:
            raise ValueError('global conditioning is enabled (gin_channels > 0) but no g was given')
        g = torch.as_tensor(g, device=self.device)
        if self.embed_speakers is not None:
            g = g.reshape(B).to(torch.int32)
        else:
            g = g.reshape(B, self._hparams.gin_channels).to(torch.float32)
        self.engine.set_global_condition(g.contiguous())

    @property
    def embedding_table(self):
        return self.variables['gc_embedding'] if self.embed_speakers is not None else None

    def has_speaker_embedding(self):
        return self.embed_speakers is not None

    def local_conditioning_enabled(self):
        return self._hparams.cin_channels > 0

    def global_conditioning_enabled(self):
        return self._hparams.gin_channels > 0

    def step(self, x, c=None, g=None, softmax=False):
        """Teacher-forced parallel forward: x [B,Cin,T] (or ids [B,T]), c [B,cin,Tc] -> [B,O,T] (wavenet.py:650-721)."""
        B, T = int(x.shape[0]), int(x.shape[-1])
        if self.engine is None:
            self.build(B, T)
        self._ensure_packed()
        y_hat = torch.empty(B, self._hparams.out_channels, T, device=x.device)
        lengths = torch.full((B,), T, dtype=torch.int32, device=x.device)
        dummy_y = x.reshape(B, T).contiguous() if not self.scalar_input else x.reshape(B, T, 1).contiguous()
        self._set_global(g, B)
        self.engine.train_fwd(x.contiguous(), c.contiguous(), dummy_y, lengths, 0, None, y_hat)
        self._have_fwd = False
        return torch.softmax(y_hat, dim=1) if softmax else y_hat

    def incremental(self, initial_input, c=None, g=None, time_length=100, test_inputs=None, softmax=True, quantize=True,
                    log_scale_min=-7.0, log_scale_min_gauss=-7.0, noise=None, return_raw=False, check=False):
        """Fast-WaveNet generation with ring-buffer queues: c [B,cin,Tc] -> samples [B,T] (wavenet.py:724-911).
        ``initial_input`` is accepted for signature parity; generation always starts from the reference's silence
        frame (wavenet.py:433-445).  ``noise`` [T,B,noise_per_step] may be supplied for reproducible draws.
        ``check``: wait for the generation and verify it; if the persistent pipeline gave up on a hand-off (a workgroup was not
        resident -- the reference's loop cannot fail this way) the batch is re-run ONCE on the launch-per-layer graph path."""
        hp = self._hparams
        B, Tc = int(c.shape[0]), int(c.shape[-1])
        hop = audio.get_hop_size(hp)
        T = Tc * hop
        if self.engine is None:
            self.build(B, T)
        self._ensure_packed()
        dev = c.device
        # noise None: drawn on the device (Philox keyed by (wavenet_random_seed, call counter)): U(1e-5, 1 - 1e-5) as mixture.py:91,104,
        # standard normal for the Gaussian head (gaussian.py:50), Gumbel uniforms for tf.multinomial (wavenet.py:865)
        self._synth_calls = getattr(self, '_synth_calls', 0) + 1
        seed = (int(hp.wavenet_random_seed) << 20) + self._synth_calls
        out = torch.empty(B, T, device=dev, dtype=torch.float32 if self.scalar_input else torch.int32)
        raw = torch.empty(B, hp.out_channels, T, device=dev) if return_raw else None
        ti = None
        if test_inputs is not None:
            ti = test_inputs.reshape(B, -1)[:, :T]
            ti = (ti.float() if self.scalar_input else ti.to(torch.int32)).contiguous()
            assert ti.shape[1] == T, 'teacher-forcing inputs must cover the whole synthesis length'
        spg = int(getattr(hp, 'mi355_steps_per_graph', 0))
        cc = c.contiguous().float()
        feats = torch.empty(B, hp.cin_channels, T, device=dev)
        gt = None
        if self.global_conditioning_enabled():
            if g is None:
                raise ValueError('global conditioning is enabled (gin_channels > 0) but no g was given')
            gt = torch.as_tensor(g, device=self.device).reshape(B, -1)

        def run(spg_):
            # The persistent pipeline pipelines the streams of a run through its layer ring: 10 streams at the wall time of one (34 - 37 us
            # per sample: real time at 22.05 kHz), every further stream + 3.6 us per sample (profiles/r5u_pipe_batch_scaling.txt) -- one run
            # of 20 streams (hparams.py: wavenet_synthesis_batch_size = 20) takes 2.0x the wall time of 8 where three groups of 8 took 3x.
            # So: the whole batch in ONE run when its per-stream LDS state fits (wn_synth_pipe_eligible), else groups of 8 (streams are
            # independent: wavenet.py:237-239 splits them over towers).  Models the pipeline does not fit take the launch-per-layer graph
            # path, whose time per step is nearly independent of the batch: up to 32 streams per run.
            group = min(B, 32)
            piped = False
            if spg_ <= 0:                   # the largest run of <= B streams the pipeline takes (an inference-only context is pre-sized: it never grows)
                g = group
                while g > 0 and not self.engine.pipeline_eligible(g):
                    g -= 1
                if g > 0:
                    piped, group = True, g
            for b0 in range(0, B, group):
                b1 = min(B, b0 + group)
                nz = None if noise is None else noise[:, b0:b1].contiguous()
                if gt is not None:
                    self._set_global(gt[b0:b1], b1 - b0)                                # wavenet.py:766-777
                self.engine.synthesize(cc[b0:b1].contiguous(), nz, out[b0:b1], None if raw is None else raw[b0:b1], None if ti is None else ti[b0:b1].contiguous(),
                                       steps_per_graph=spg_, seed=seed * 64 + b0 // group)
                self.engine.upsampled_features(feats[b0:b1])
            return group

        def attempt(spg_):
            grp = run(spg_)
            if check:
                torch.cuda.synchronize()
                self.engine.synth_check()
            return grp

        try:
            group = attempt(spg)
        except _ext.WnError as e:
            out_of_range = 'half-precision range' in str(e)
            if not check or self.engine.synth_path != 'pipeline' or not ('timed out' in str(e) or out_of_range):
                raise
            self.synth_fallbacks = getattr(self, 'synth_fallbacks', 0) + 1
            torch.cuda.synchronize()
            try:
                self.engine.synth_check()          # (a flag of a later group of the same batch may still be pending)
            except _ext.WnError:
                pass
            if out_of_range:                       # the residual stream of this model exceeds 65504 somewhere: bf16 storage (8 exponent bits) from now on
                log('WaveNet synthesis: {} -- re-running this batch with bf16 pipeline storage'.format(e))
                self.engine.pipeline_dtype(False)
                group = attempt(spg)
            else:
                log('WaveNet synthesis: {} -- re-running this batch on the launch-per-layer graph path'.format(e))
                group = attempt(32)
        if getattr(self, '_logged_synth_path', None) != self.engine.synth_path:
            self._logged_synth_path = self.engine.synth_path
            log('WaveNet synthesis path: {} ({} streams per run)'.format(self.engine.synth_path, group))
        self.upsampled_local_features = feats
        return (out, raw) if return_raw else out

    # ------------------------------------------------------------------ checkpoint state
    def state_dict(self):
        return {'params': self.params.detach().cpu(), 'ema': self.ema_params.detach().cpu(), 'adam_m': self.adam_m.detach().cpu(),
                'adam_v': self.adam_v.detach().cpu(), 'global_step': self.global_step,
                'layout': {k: (tuple(s), int(o)) for k, (s, o) in self.engine.layout.items()}}

    def load_state_dict(self, sd):
        for name, dst in (('params', self.params), ('ema', self.ema_params), ('adam_m', self.adam_m), ('adam_v', self.adam_v)):
            if getattr(self, 'inference_only', False) and name.startswith('adam'):
                continue
            src = sd[name]
            if src.numel() != dst.numel():
                raise ValueError('checkpoint tensor %s has %d elements, model expects %d' % (name, src.numel(), dst.numel()))
            dst.copy_(src.to(dst.device))
        self.global_step = int(sd.get('global_step', 0))
        self._dirty = True
