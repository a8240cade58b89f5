"""ctypes binding of libwavenet_mi355.so (C ABI: include/wavenet_mi355.h).

This is the ONLY compute path of the package: there is no CPU / eager fallback.  If the shared
library has not been built (``python tacotron-2_amd/csrc/build.py``) importing the engine raises.
PyTorch is used for device memory, streams and torch.distributed only.
"""
import ctypes
import os
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, '..', '..', 'csrc', 'libwavenet_mi355.so'))
if os.environ.get('WN_MI355_TEST_LIB'):      # test hook: the ASAN / UBSAN build of the host side (csrc/build.py --sanitize), never the product
    LIB_PATH = os.environ['WN_MI355_TEST_LIB']

WN_ABI_VERSION = 4
WN_MAX_UPSAMPLE = 8
INPUT_TYPES = {'raw': 0, 'mulaw': 1, 'mulaw-quantize': 2}
UPSAMPLE_TYPES = {'NearestNeighbor': 0, '2D': 1, 'SubPixel': 2, '1D': 3, 'Resize': 4}
ACTIVATIONS = {None: 0, 'None': 0, 'Relu': 1, 'LeakyRelu': 2}
LR_SCHEDULES = {'exponential': 0, 'noam': 1}
COMPUTE_DTYPES = {'bf16': 0, 'fp32': 1, 'float32': 1}      # wn_compute_dtype: 'fp32' = the reference's arithmetic for forward, loss and backward (csrc/wn_f32.hip)
STATUS = {0: 'WN_OK', -1: 'WN_E_ARG', -2: 'WN_E_SHAPE', -3: 'WN_E_HIP', -4: 'WN_E_UNSUPPORTED', -5: 'WN_E_STATE'}


class WnConfig(ctypes.Structure):
    _fields_ = [
        ('abi_version', ctypes.c_int32),
        ('layers', ctypes.c_int32), ('stacks', ctypes.c_int32),
        ('residual_channels', ctypes.c_int32), ('gate_channels', ctypes.c_int32),
        ('skip_out_channels', ctypes.c_int32), ('out_channels', ctypes.c_int32),
        ('kernel_size', ctypes.c_int32), ('cin_channels', ctypes.c_int32),
        ('input_type', ctypes.c_int32), ('quantize_channels', ctypes.c_int32),
        ('use_bias', ctypes.c_int32), ('legacy', ctypes.c_int32), ('residual_legacy', ctypes.c_int32),
        ('log_scale_min', ctypes.c_float), ('log_scale_min_gauss', ctypes.c_float),
        ('cdf_loss', ctypes.c_int32),
        ('upsample_type', ctypes.c_int32), ('upsample_activation', ctypes.c_int32),
        ('n_upsample', ctypes.c_int32), ('upsample_scales', ctypes.c_int32 * WN_MAX_UPSAMPLE),
        ('freq_axis_kernel_size', ctypes.c_int32), ('leaky_alpha', ctypes.c_float),
        ('dropout', ctypes.c_float), ('clip_gradients', ctypes.c_int32),
        ('gradient_max_norm', ctypes.c_float), ('gradient_max_value', ctypes.c_float),
        ('adam_beta1', ctypes.c_float), ('adam_beta2', ctypes.c_float),
        ('adam_epsilon', ctypes.c_float), ('ema_decay', ctypes.c_float),
        ('max_batch', ctypes.c_int32), ('max_time', ctypes.c_int32),
        ('gin_channels', ctypes.c_int32), ('use_speaker_embedding', ctypes.c_int32), ('n_speakers', ctypes.c_int32),
        ('weight_normalization', ctypes.c_int32),
        ('inference_only', ctypes.c_int32),
        ('grad_buckets', ctypes.c_int32),
        ('compute_dtype', ctypes.c_int32),
    ]


class WnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('%s: %s' % (STATUS.get(code, code), msg))
        self.code = code


_lib = None


def load_library():
    """dlopen the HIP library; raises (loudly) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libwavenet_mi355.so not found at %s -- the HIP extension is the only compute '
                           'path of this package; build it with `python tacotron-2_amd/csrc/build.py`' % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64: it must be the HIP runtime of the process (device memory and streams come from
    # torch), so import torch BEFORE the library is dlopen'ed -- loaded the other way round, /opt/rocm's runtime is mapped first
    # and the second runtime finds "no ROCm-capable device".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64
    sigs = {
        'wn_create': (ctypes.c_int, [ctypes.POINTER(WnConfig), ctypes.POINTER(vp)]),
        'wn_destroy': (None, [vp]),
        'wn_last_error': (ctypes.c_char_p, [vp]),
        'wn_receptive_field': (ctypes.c_int, [vp]),
        'wn_param_count': (i64, [vp]),
        'wn_num_tensors': (ctypes.c_int, [vp]),
        'wn_tensor_info': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i64)]),
        'wn_pack_weights': (ctypes.c_int, [vp, vp, vp]),
        'wn_train_fwd': (ctypes.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, u64, vp, vp, vp]),
        'wn_train_bwd': (ctypes.c_int, [vp, vp, vp]),
        'wn_bwd_num_buckets': (ctypes.c_int, [vp]),
        'wn_bwd_bucket_range': (ctypes.c_int, [vp, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]),
        'wn_bwd_wait_bucket': (ctypes.c_int, [vp, i32, vp]),
        'wn_get_upsampled_features': (ctypes.c_int, [vp, vp, vp]),
        'wn_optim_step': (ctypes.c_int, [vp, vp, vp, vp, vp, vp, f32, i64, vp]),
        'wn_learning_rate': (f32, [i32, f32, i64, f32, i64, f32]),
        'wn_synthesize': (ctypes.c_int, [vp, vp, i32, i32, vp, u64, vp, vp, vp, i32, vp]),
        'wn_noise_per_step': (ctypes.c_int, [vp]),
        'wn_fill_noise': (ctypes.c_int, [vp, vp, i32, i32, u64, vp]),
        'wn_synth_check': (ctypes.c_int, [vp]),
        'wn_synth_last_path': (ctypes.c_int, [vp]),
        'wn_test_gemm8p_mask': (ctypes.c_int, [vp]),
        'wn_test_pipe_layout': (ctypes.c_int, [i32, i32, i32, vp, i32, vp, i32, vp, vp]),
        'wn_synth_pipe_dtype': (ctypes.c_int, [vp, i32]),
        'wn_synth_last_instances': (ctypes.c_int, [vp]),
        'wn_synth_last_config': (ctypes.c_int, [vp, ctypes.POINTER(i32), i32]),
        'wn_synth_last_batched': (ctypes.c_int, [vp]),
        'wn_synth_pipe_eligible': (ctypes.c_int, [vp, i32]),
        'wn_sample': (ctypes.c_int, [vp, vp, i32, i32, vp, vp, vp]),
        'wn_mulaw': (ctypes.c_int, [vp, vp, i64, vp]),
        'wn_inv_mulaw': (ctypes.c_int, [vp, vp, i64, vp]),
        'wn_mulaw_quantize': (ctypes.c_int, [vp, vp, i64, vp]),
        'wn_inv_mulaw_quantize': (ctypes.c_int, [vp, vp, i64, vp]),
        'wn_argmax_channels': (ctypes.c_int, [vp, vp, i32, i32, i32, vp]),
        'wn_loss': (ctypes.c_int, [vp, vp, vp, vp, i32, i32, i32, vp, vp]),
        'wn_profile': (ctypes.c_int, [vp, i32]),
        'wn_profile_result': (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64)]),
        'wn_profile_kernel_result': (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64)]),
        'wn_profile_kernel_clock': (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64)]),
        'wn_profile_rows_per_launch': (i64, [vp]),
        'wn_set_batch_parts': (ctypes.c_int, [vp, i32]),
        'wn_debug_copy': (ctypes.c_int, [vp, ctypes.c_char_p, i32, vp, i64, vp]),
        'wn_trace_arm': (ctypes.c_int, [vp, i32]),
        'wn_trace_read': (ctypes.c_int, [vp, i32, ctypes.POINTER(i32), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
        'wn_set_global_condition': (ctypes.c_int, [vp, vp, i32, vp]),
        'wn_workspace_bytes': (i64, [vp]),
        'wn_dominant_kernel_name': (ctypes.c_char_p, []),
        'wn_test_dropout_mask': (ctypes.c_int, [ctypes.c_uint64, i32, ctypes.c_float, i64, i64, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)            # AttributeError here == ABI symbol missing
        fn.restype = res
        fn.argtypes = args
    lib._wn_symbols = list(sigs)
    _lib = lib
    return lib


def exported_symbols():
    return list(load_library()._wn_symbols)


def config_from_hparams(hp, max_batch, max_time, inference_only=False, grad_buckets=None):
    """hparams (reference keys, hparams.py:187-233, 309-327) -> wn_config."""
    cfg = WnConfig()
    cfg.abi_version = WN_ABI_VERSION
    for k in ('layers', 'stacks', 'residual_channels', 'gate_channels', 'skip_out_channels', 'out_channels',
              'kernel_size', 'cin_channels', 'quantize_channels', 'freq_axis_kernel_size'):
        setattr(cfg, k, int(getattr(hp, k)))
    if hp.input_type not in INPUT_TYPES:
        raise AssertionError('input_type must be one of raw / mulaw / mulaw-quantize')   # util.py:10-11
    cfg.input_type = INPUT_TYPES[hp.input_type]
    cfg.use_bias = int(bool(hp.use_bias))
    cfg.legacy = int(bool(hp.legacy))
    cfg.residual_legacy = int(bool(hp.residual_legacy))
    cfg.log_scale_min = float(hp.log_scale_min)
    cfg.log_scale_min_gauss = float(hp.log_scale_min_gauss)
    cfg.cdf_loss = int(bool(hp.cdf_loss))
    if hp.upsample_type not in UPSAMPLE_TYPES:
        raise ValueError('unknown upsample_type %r' % (hp.upsample_type,))
    cfg.upsample_type = UPSAMPLE_TYPES[hp.upsample_type]
    cfg.upsample_activation = ACTIVATIONS[hp.upsample_activation]
    scales = list(hp.upsample_scales)
    if hp.upsample_type == 'NearestNeighbor':
        from datasets.audio import get_hop_size
        scales = [get_hop_size(hp)]
    cfg.n_upsample = len(scales)
    for i, s in enumerate(scales):
        cfg.upsample_scales[i] = int(s)
    cfg.leaky_alpha = float(hp.leaky_alpha)
    cfg.dropout = float(hp.wavenet_dropout)
    cfg.clip_gradients = int(bool(hp.wavenet_clip_gradients))
    cfg.gradient_max_norm = float(hp.wavenet_gradient_max_norm)
    cfg.gradient_max_value = float(hp.wavenet_gradient_max_value)
    cfg.adam_beta1 = float(hp.wavenet_adam_beta1)
    cfg.adam_beta2 = float(hp.wavenet_adam_beta2)
    cfg.adam_epsilon = float(hp.wavenet_adam_epsilon)
    cfg.ema_decay = float(hp.wavenet_ema_decay)
    cfg.max_batch = int(max_batch)
    cfg.max_time = int(max_time)
    cfg.gin_channels = int(getattr(hp, 'gin_channels', -1))                       # hparams.py:228-230
    cfg.use_speaker_embedding = int(bool(getattr(hp, 'use_speaker_embedding', True))) if cfg.gin_channels > 0 else 0
    cfg.n_speakers = int(getattr(hp, 'n_speakers', 0) or 0)
    cfg.weight_normalization = int(bool(getattr(hp, 'wavenet_weight_normalization', False)))      # hparams.py:323
    cfg.inference_only = int(bool(inference_only))
    if grad_buckets is None:      # data parallel: 3 pieces (two early layer groups + the rest) overlap the all-reduce with the backward; single GPU: 1
        import torch
        dp = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        grad_buckets = int(getattr(hp, 'mi355_grad_buckets', 3)) if dp else 1
    cfg.grad_buckets = int(grad_buckets)
    dt = str(getattr(hp, 'mi355_compute_dtype', 'bf16'))
    if dt not in COMPUTE_DTYPES:
        raise ValueError("mi355_compute_dtype must be 'bf16' or 'fp32' (got %r)" % (dt,))
    cfg.compute_dtype = COMPUTE_DTYPES[dt]
    return cfg


def _ptr(t):
    return ctypes.c_void_p(0) if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check(t, dtype, name):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError('%s must be a CUDA(HIP) tensor' % name)
    if t.dtype != dtype:
        raise TypeError('%s must have dtype %s (got %s)' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)
    return t


class Engine:
    """One wn_ctx: owns packed weights + workspace on the current device."""

    def __init__(self, hp, max_batch, max_time, inference_only=False, grad_buckets=None):
        """inference_only: synthesis-only context (no training workspace, every synthesis buffer pre-sized: wn_config.inference_only).
        grad_buckets: pieces of the flat gradient train_bwd completes early (None: 3 under torch.distributed with > 1 rank, else 1)."""
        self.lib = load_library()
        self.cfg = config_from_hparams(hp, max_batch, max_time, inference_only, grad_buckets)
        h = ctypes.c_void_p()
        rc = self.lib.wn_create(ctypes.byref(self.cfg), ctypes.byref(h))
        if rc != 0:
            raise WnError(rc, (self.lib.wn_last_error(None) or b'').decode())
        self.h = h
        self.hop = int(np.prod([self.cfg.upsample_scales[i] for i in range(self.cfg.n_upsample)]))
        self.n_params = int(self.lib.wn_param_count(self.h))
        self.layout = OrderedDict()
        name = ctypes.create_string_buffer(128)
        shape = (ctypes.c_int32 * 4)()
        ndim = ctypes.c_int32()
        off = ctypes.c_int64()
        for i in range(self.lib.wn_num_tensors(self.h)):
            self._ok(self.lib.wn_tensor_info(self.h, i, name, shape, ctypes.byref(ndim), ctypes.byref(off)))
            self.layout[name.value.decode()] = (tuple(shape[k] for k in range(ndim.value)), int(off.value))

    def set_global_condition(self, g):
        """g: int32 speaker ids [B] (use_speaker_embedding) or float32 [B, gin_channels]; applies to the next forward / synthesis
        (wavenet.py:669-678, 766-777)."""
        import torch
        g = g.contiguous()
        want = torch.int32 if self.cfg.use_speaker_embedding else torch.float32
        _check(g, want, 'g')
        self._ok(self.lib.wn_set_global_condition(self.h, _ptr(g), int(g.shape[0]), _stream()))

    def close(self):
        if getattr(self, 'h', None):
            self.lib.wn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ok(self, rc):
        if rc != 0:
            raise WnError(rc, (self.lib.wn_last_error(self.h) or b'').decode())

    # ---- parameter helpers
    def views(self, flat):
        """name -> view into the flat fp32 buffer (TF layouts)."""
        out = OrderedDict()
        for k, (shape, off) in self.layout.items():
            out[k] = flat[off:off + int(np.prod(shape))].view(*shape)
        return out

    @property
    def receptive_field(self):
        return int(self.lib.wn_receptive_field(self.h))

    @property
    def noise_per_step(self):
        return int(self.lib.wn_noise_per_step(self.h))

    # ---- hot path
    def pack_weights(self, params):
        import torch
        _check(params, torch.float32, 'params')
        self._ok(self.lib.wn_pack_weights(self.h, _ptr(params), _stream()))

    def train_fwd(self, x, c, y, lengths, dropout_seed, loss_out, y_hat_out=None):
        import torch
        B, T = int(lengths.shape[0]), int(x.shape[-1])
        Tc = int(c.shape[-1])
        _check(c, torch.float32, 'c'); _check(lengths, torch.int32, 'lengths')
        self._ok(self.lib.wn_train_fwd(self.h, _ptr(x), _ptr(c), _ptr(y), _ptr(lengths), B, T, Tc,
                                       ctypes.c_uint64(int(dropout_seed) & (2 ** 64 - 1)), _ptr(loss_out), _ptr(y_hat_out), _stream()))

    def train_bwd(self, grads):
        import torch
        _check(grads, torch.float32, 'grads')
        self._ok(self.lib.wn_train_bwd(self.h, _ptr(grads), _stream()))

    def grad_buckets(self):
        """[(offset, count)] of the pieces in which train_bwd completes the flat gradient buffer (top layers first)."""
        out = []
        off, cnt = ctypes.c_int64(), ctypes.c_int64()
        for i in range(int(self.lib.wn_bwd_num_buckets(self.h))):
            self._ok(self.lib.wn_bwd_bucket_range(self.h, i, ctypes.byref(off), ctypes.byref(cnt)))
            out.append((int(off.value), int(cnt.value)))
        return out

    def wait_bucket(self, i, stream):
        """Order `stream` (a torch.cuda.Stream) after bucket i of the last train_bwd."""
        self._ok(self.lib.wn_bwd_wait_bucket(self.h, int(i), ctypes.c_void_p(stream.cuda_stream)))

    def optim_step(self, params, grads, m, v, ema, lr, step):
        self._ok(self.lib.wn_optim_step(self.h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(ema),
                                        ctypes.c_float(lr), ctypes.c_int64(step), _stream()))

    def upsampled_features(self, out):
        self._ok(self.lib.wn_get_upsampled_features(self.h, _ptr(out), _stream()))

    def synthesize(self, c, noise, out_samples, out_raw=None, test_inputs=None, steps_per_graph=0, seed=0):
        """Enqueue the generation of T = Tc*hop samples for B streams (asynchronous: call synth_check() after synchronising).
        noise None: drawn on the device from `seed` (Philox4x32-10; fill_noise gives the same stream).  steps_per_graph <= 0: the
        persistent pipeline when the model fits it, else the launch-per-layer hipGraph path; > 0: that path with this many steps
        per captured graph."""
        B, Tc = int(c.shape[0]), int(c.shape[-1])
        self._ok(self.lib.wn_synthesize(self.h, _ptr(c), B, Tc, _ptr(noise), ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), _ptr(test_inputs),
                                        _ptr(out_samples), _ptr(out_raw), int(steps_per_graph), _stream()))

    def fill_noise(self, noise, B, T, seed):
        """The device noise stream of synthesize(noise=None, seed): float32 [T, B, noise_per_step]."""
        import torch
        _check(noise, torch.float32, 'noise')
        self._ok(self.lib.wn_fill_noise(self.h, _ptr(noise), int(B), int(T), ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), _stream()))
        return noise

    def synth_check(self):
        """Wait for the last synthesize of this engine and raise if the pipeline gave up on a hand-off."""
        self._ok(self.lib.wn_synth_check(self.h))

    def pipeline_eligible(self, B):
        """Would synthesize(steps_per_graph <= 0) run B streams on the persistent pipeline?"""
        rc = int(self.lib.wn_synth_pipe_eligible(self.h, int(B)))
        if rc < 0:
            self._ok(rc)
        return rc == 1

    def pipeline_dtype(self, half):
        """16-bit storage type of the persistent pipeline for the next runs: True = IEEE half (default), False = bf16."""
        self._ok(self.lib.wn_synth_pipe_dtype(self.h, 1 if half else 0))

    def synth_config(self):
        """How the last synthesize() ran, as the library configured it (wn_synth_last_config)."""
        v = (ctypes.c_int32 * 10)()
        n = int(self.lib.wn_synth_last_config(self.h, v, 10))
        if n < 0:
            self._ok(n)
        keys = ('path', 'instances', 'batched_premultiplication', 'kernel_spec', 'half_storage', 'head_cus', 'early_from', 'abort_every', 'workgroups', 'streams_per_instance')
        d = dict(zip(keys, (int(v[i]) for i in range(n))))
        d['path'] = {0: None, 1: 'graph', 2: 'pipeline', 3: 'graph-fp32'}.get(d.get('path'))
        return d

    @property
    def synth_path(self):
        return {0: None, 1: 'graph', 2: 'pipeline', 3: 'graph-fp32'}.get(int(self.lib.wn_synth_last_path(self.h)))

    def loss(self, y_hat, y, lengths, shift, loss_out):
        B, T = int(y_hat.shape[0]), int(y_hat.shape[-1])
        self._ok(self.lib.wn_loss(self.h, _ptr(y_hat), _ptr(y), _ptr(lengths), B, T, int(shift), _ptr(loss_out), _stream()))

    def profile(self, enable):
        self._ok(self.lib.wn_profile(self.h, int(bool(enable))))

    def profile_result(self):
        ms, n = ctypes.c_double(), ctypes.c_int64()
        self._ok(self.lib.wn_profile_result(self.h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def profile_kernel_result(self):
        """(total ms, launches) of the timed gate launches by their in-kernel start / end stamps (pure kernel time)."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        self._ok(self.lib.wn_profile_kernel_result(self.h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def profile_kernel_clock(self):
        """(MHz, launches): mean shader clock inside the timed gate launches (workgroup 0's cycle counter over the 100 MHz wall clock)."""
        mhz, n = ctypes.c_double(), ctypes.c_int64()
        self._ok(self.lib.wn_profile_kernel_clock(self.h, ctypes.byref(mhz), ctypes.byref(n)))
        return mhz.value, n.value

    def set_batch_parts(self, parts):
        self._ok(self.lib.wn_set_batch_parts(self.h, int(parts)))

    def profile_rows_per_launch(self):
        return int(self.lib.wn_profile_rows_per_launch(self.h))

    def trace_arm(self, steps_from_now=1):
        """Stamp every tile-engine / grouped weight-gradient launch of the `steps_from_now`-th next training step (wn_trace_arm)."""
        self._ok(self.lib.wn_trace_arm(self.h, int(steps_from_now)))

    def trace_read(self, cap=1024):
        """[(kind, stream, start_tick, end_tick)] of the armed step in enqueue order, 100 MHz ticks; [] if it has not completed.  Synchronises."""
        k = (ctypes.c_int32 * cap)(); st = (ctypes.c_uint64 * cap)(); t0 = (ctypes.c_uint64 * cap)(); t1 = (ctypes.c_uint64 * cap)()
        n = self.lib.wn_trace_read(self.h, cap, k, st, t0, t1)
        if n < 0:
            self._ok(n)
        return [(int(k[i]), int(st[i]), int(t0[i]), int(t1[i])) for i in range(n)]

    def debug_copy(self, name, layer, rows, cols):
        import torch
        out = torch.empty(rows, cols, dtype=torch.float32, device='cuda')
        self._ok(self.lib.wn_debug_copy(self.h, name.encode(), int(layer), _ptr(out), ctypes.c_int64(rows * cols), _stream()))
        return out

    def sample(self, y_hat, noise, out):
        B, T = int(y_hat.shape[0]), int(y_hat.shape[-1])
        self._ok(self.lib.wn_sample(self.h, _ptr(y_hat), B, T, _ptr(noise), _ptr(out), _stream()))


def learning_rate(schedule, init_lr, step, decay_rate=0.5, decay_steps=200000, warmup=4000.0):
    return float(load_library().wn_learning_rate(LR_SCHEDULES[schedule], init_lr, int(step), decay_rate, int(decay_steps), warmup))


# ---- mu-law codec on device tensors (wavenet_vocoder/util.py semantics, mu = 255)
def _ew(fn_name, src, dst):
    lib = load_library()
    rc = getattr(lib, fn_name)(_ptr(src), _ptr(dst), ctypes.c_int64(src.numel()), _stream())
    if rc != 0:
        raise WnError(rc, (lib.wn_last_error(None) or b'').decode())
    return dst


def mulaw(x):
    import torch
    return _ew('wn_mulaw', _check(x, torch.float32, 'x'), torch.empty_like(x))


def inv_mulaw(y):
    import torch
    return _ew('wn_inv_mulaw', _check(y, torch.float32, 'y'), torch.empty_like(y))


def mulaw_quantize(x):
    import torch
    return _ew('wn_mulaw_quantize', _check(x, torch.float32, 'x'), torch.empty(x.shape, dtype=torch.int32, device=x.device))


def inv_mulaw_quantize(q):
    import torch
    return _ew('wn_inv_mulaw_quantize', _check(q, torch.int32, 'q'), torch.empty(q.shape, dtype=torch.float32, device=q.device))


def argmax_channels(logits):
    import torch
    _check(logits, torch.float32, 'logits')
    B, Q, T = logits.shape
    out = torch.empty(B, T, dtype=torch.int32, device=logits.device)
    lib = load_library()
    rc = lib.wn_argmax_channels(_ptr(logits), _ptr(out), B, Q, T, _stream())
    if rc != 0:
        raise WnError(rc, (lib.wn_last_error(None) or b'').decode())
    return out
