/*
 * wavenet_mi355.h -- C ABI of libwavenet_mi355.so: the MI355X (gfx950) native WaveNet-vocoder
 * training / Fast-WaveNet synthesis hot path.
 *
 * The reference (Rayhane-mamah/Tacotron-2) has NO native / FFI boundary: its hot path is a chain of
 * TensorFlow-1 graph ops built by the Python class wavenet_vocoder/models/wavenet.py:WaveNet and run
 * with session.run (wavenet_vocoder/train.py:303, synthesizer.py:97).  This header is therefore the
 * boundary the reference *would* bind if it had one; each entry point names the reference code whose
 * arithmetic it replaces.  The Python mirror of the reference's class API that calls these functions
 * through ctypes lives in tacotron-2_amd/wavenet_vocoder/ (see INTEGRATION.md).
 *
 * Conventions
 *   - every function returns an int status: 0 = WN_OK, negative = error; wn_last_error() gives text.
 *     No C++ exception crosses this boundary.
 *   - all data pointers are DEVICE pointers (HBM) owned by the caller and borrowed for the call only,
 *     unless the parameter is documented "host".  Tensors are contiguous in the stated layout.
 *   - all work is enqueued asynchronously, ordered after everything already on the caller's stream (`void* stream` is a
 *     hipStream_t) and before whatever the caller enqueues next (ctx-owned side streams are forked and joined with events);
 *     no entry point of the drop-in surface synchronises the device (wn_synth_check and the WN_TEST_HOOKS say so where they do).
 *     One wn_ctx per (process, device); not thread-safe.
 *   - memory: wn_create sizes the workspace once for (max_batch, max_time).  Training contexts allocate their synthesis state on the
 *     first wn_synthesize (eval steps); inference-only contexts (cfg.inference_only) pre-size it and never allocate afterwards.
 *   - parameters, gradients and optimiser slots are single flat fp32 buffers; the tensors inside them
 *     keep the reference's TensorFlow layouts ([k,in,out] conv kernels ...) at the offsets reported by
 *     wn_tensor_info(), under names mirroring the reference's variable scopes.
 */
#ifndef WAVENET_MI355_H
#define WAVENET_MI355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WN_ABI_VERSION 4

enum wn_status {
    WN_OK = 0,
    WN_E_ARG = -1,          /* bad argument / null pointer            */
    WN_E_SHAPE = -2,        /* shape constraint violated              */
    WN_E_HIP = -3,          /* HIP runtime error (see wn_last_error)  */
    WN_E_UNSUPPORTED = -4,  /* valid reference config not built yet   */
    WN_E_STATE = -5         /* call order violated (e.g. bwd before fwd) */
};

enum wn_input_type { WN_INPUT_RAW = 0, WN_INPUT_MULAW = 1, WN_INPUT_MULAW_QUANTIZE = 2 }; /* hparams.py:187 */
enum wn_upsample_type {                                                                    /* hparams.py:219 */
    WN_UP_NEAREST = 0, WN_UP_2D = 1, WN_UP_SUBPIXEL = 2, WN_UP_1D = 3, WN_UP_RESIZE = 4
};
enum wn_activation { WN_ACT_NONE = 0, WN_ACT_RELU = 1, WN_ACT_LEAKY_RELU = 2 };             /* hparams.py:220 */
enum wn_lr_schedule { WN_LR_EXPONENTIAL = 0, WN_LR_NOAM = 1 };                              /* hparams.py:309 */
/* Arithmetic of training / the teacher-forced forward (WaveNet.step, add_loss, add_optimizer, evaluation):
 *   WN_COMPUTE_BF16  bf16 MFMA operands, fp32 accumulation (BASELINE configs[1]'s training dtype; the tuned path);
 *   WN_COMPUTE_F32   the reference's own arithmetic -- fp32 activations, fp32 weights, fp32 accumulation (modules.py:306-320,
 *                    wavenet.py:650-721) -- for wn_train_fwd AND wn_train_bwd (fp32 MFMA SGEMMs, ~13x slower: an accuracy /
 *                    validation mode; gradient buckets collapse to one), and for wn_synthesize: fp32 weights read from the flat
 *                    parameter buffer, fp32 ring queues, fp32 accumulation, precise tanh / exp (modules.py:273-303,
 *                    wavenet.py:821-886; launch-per-layer path only, far from real time; wn_synth_last_path() = 3). */
enum wn_compute_dtype { WN_COMPUTE_BF16 = 0, WN_COMPUTE_F32 = 1 };

#define WN_MAX_UPSAMPLE 8

/* Model + optimiser hyper-parameters: the hparams.py keys read by wavenet.py:89-208 / :522-629. */
typedef struct wn_config {
    int32_t abi_version;            /* = WN_ABI_VERSION */
    /* architecture (hparams.py:187-211) */
    int32_t layers, stacks;
    int32_t residual_channels, gate_channels, skip_out_channels, out_channels;
    int32_t kernel_size;            /* must be 3 */
    int32_t cin_channels;           /* == num_mels, multiple of 16 */
    int32_t input_type;             /* wn_input_type */
    int32_t quantize_channels;
    int32_t use_bias;
    int32_t legacy, residual_legacy;
    float   log_scale_min, log_scale_min_gauss;
    int32_t cdf_loss;
    /* upsample net (hparams.py:219-225) */
    int32_t upsample_type;          /* wn_upsample_type */
    int32_t upsample_activation;    /* wn_activation */
    int32_t n_upsample;
    int32_t upsample_scales[WN_MAX_UPSAMPLE];
    int32_t freq_axis_kernel_size;
    float   leaky_alpha;
    /* training (hparams.py:309-327) */
    float   dropout;                /* wavenet_dropout */
    int32_t clip_gradients;
    float   gradient_max_norm, gradient_max_value;
    float   adam_beta1, adam_beta2, adam_epsilon, ema_decay;
    /* capacity: workspace is sized once for these (288 GB HBM: be generous) */
    int32_t max_batch;              /* utterances per call                       */
    int32_t max_time;               /* samples per utterance (train or synth)    */
    /* global conditioning (hparams.py:228-230; wavenet.py:152-158, 669-678; modules.py:10-21, 426-432, 499-508) */
    int32_t gin_channels;           /* <= 0 disables                                                          */
    int32_t use_speaker_embedding;  /* 1: g = speaker ids looked up in the [n_speakers, gin_channels] table    */
    int32_t n_speakers;
    /* Salimans & Kingma weight normalisation of every convolution (hparams.py:323; modules.py:44-177): the exported tensors
     * become kernel (= v), g [last kernel axis], bias; kernels are used as g * v / ||v|| (norm over all axes but the last). */
    int32_t weight_normalization;
    /* 1: synthesis-only context.  Skips the saved-activation / backward workspace of training (~45 KB per (batch x time) row at the
     * paper shape; what remains -- conditioning, ring queues, mailboxes, the noise buffer -- is ~1.5 KB per row) and PRE-SIZES every
     * synthesis buffer for (max_batch, max_time), so wn_synthesize never allocates.  wn_train_* / wn_optim_step return WN_E_STATE.
     * A training context (0) can still synthesise (eval steps); it allocates its synthesis state on the first wn_synthesize. */
    int32_t inference_only;
    /* Data-parallel training: number of pieces in which wn_train_bwd completes the layer stack's gradients (wn_bwd_*bucket*), so that
     * the caller can all-reduce one piece while the next is computed.  <= 1 (single GPU): everything is final when the call ends and
     * the weight gradients run after the backward chain (measured 1.5 % faster than overlapping them when there is nothing to hide). */
    int32_t grad_buckets;
    int32_t compute_dtype;          /* enum wn_compute_dtype; hparams key mi355_compute_dtype: 'bf16' | 'fp32' */
} wn_config;

typedef struct wn_ctx wn_ctx;

/* ---- lifetime ------------------------------------------------------------------------------- */
int  wn_create(const wn_config* cfg, wn_ctx** out);     /* validates cfg (wavenet.py:94,97; models/__init__.py:6-9) */
void wn_destroy(wn_ctx* ctx);
const char* wn_last_error(const wn_ctx* ctx);           /* ctx may be NULL: error of the last failed wn_create */
int  wn_receptive_field(const wn_ctx* ctx);             /* wavenet.py:54-71; like every accessor below: WN_E_ARG for a NULL ctx */

/* ---- parameter table (host-side; replaces tf.trainable_variables(), wavenet.py:467) ---------- */
int64_t wn_param_count(const wn_ctx* ctx);              /* floats in the flat parameter buffer (incl. alignment pad) */
int     wn_num_tensors(const wn_ctx* ctx);
/* name: >=128 chars; shape: >=4 ints (TF layout) */
int     wn_tensor_info(const wn_ctx* ctx, int index, char* name, int32_t* shape, int32_t* ndim, int64_t* offset);

/* Re-pack the flat fp32 parameters into the bf16 MFMA-fragment-ordered copies the kernels read
 * (ctx-owned).  Call after every change of the parameters (i.e. after wn_optim_step). */
int wn_pack_weights(wn_ctx* ctx, const float* params, void* stream);

/* Global conditioning of the NEXT forward / synthesis call (wavenet.py:669-678 / :766-777): g = int32 speaker ids [B]
 * when cfg.use_speaker_embedding, else float [B, gin_channels].  Copied into the context (pointer borrowed for the call).
 * Required before wn_train_fwd / wn_synthesize when cfg.gin_channels > 0 ("g" must match the batch size). */
int wn_set_global_condition(wn_ctx* ctx, const void* g, int32_t B, void* stream);

/* ---- training: replaces WaveNet.step + add_loss (wavenet.py:650-721, 476-495) ----------------- */
/* x        scalar input: float [B,1,T]; mulaw-quantize: int32 class ids [B,T] (== the reference's one-hot
 *          [B,256,T], feeder.py:295-306, without materialising it)
 * c        float [B, cin, Tc], Tc*hop == T   (local conditioning, feeder.py:319-340)
 * y        targets: float [B,T,1] (scalar) or int32 [B,T] (mulaw-quantize)   (wavenet.py:488-495)
 * lengths  int32 [B]
 * dropout_seed  counter-based mask key for this step (tf.layers.dropout, modules.py:484); the same
 *          mask is regenerated in wn_train_bwd.  Dropout is disabled when cfg.dropout == 0.
 * loss_out float [1] device: masked mean loss of this batch (modules.py:798/817/836)
 * y_hat_out optional float [B, out_channels, T] (NULL to skip)                                    */
int wn_train_fwd(wn_ctx* ctx, const void* x, const float* c, const void* y, const int32_t* lengths,
                 int32_t B, int32_t T, int32_t Tc, uint64_t dropout_seed,
                 float* loss_out, float* y_hat_out, void* stream);

/* Backward of the last wn_train_fwd: writes d(loss)/d(param) for every tensor into the flat fp32
 * buffer `grads` (same layout as params; overwritten, not accumulated).
 * Replaces optimizer.compute_gradients (wavenet.py:557). */
int wn_train_bwd(wn_ctx* ctx, float* grads, void* stream);

/* Gradient buckets for data-parallel training (replaces the tower-gradient loop of wavenet.py:553-581, which averages variable by
 * variable after ALL gradients exist).  wn_train_bwd completes the flat gradient buffer in wn_bwd_num_buckets() contiguous
 * pieces, top layers first: the weight gradients of a bucket are computed on a ctx-owned low-priority stream while the serial
 * d z / d h chain is still working on the layers below.  wn_bwd_wait_bucket makes `stream` (e.g. the communication stream of the
 * caller's all-reduce) wait until bucket i of the LAST enqueued wn_train_bwd is final; the caller's own stream is ordered after
 * the whole buffer when wn_train_bwd returns, as before.  Ranges are floats into the flat buffer; the buckets are disjoint and
 * cover every tensor.  Models with weight normalisation or global conditioning report one bucket. */
int wn_bwd_num_buckets(const wn_ctx* ctx);
int wn_bwd_bucket_range(const wn_ctx* ctx, int32_t i, int64_t* offset, int64_t* count);
int wn_bwd_wait_bucket(wn_ctx* ctx, int32_t i, void* stream);

/* Stand-alone masked loss on [B,O,T] network outputs.  shift = 1: training alignment (prediction t vs sample
 * t+1, wavenet.py:488-495); shift = 0: evaluation of the incremental loop's raw outputs (wavenet.py:497-506).
 * Invalidates the saved backward state of the last wn_train_fwd. */
int wn_loss(wn_ctx* ctx, const float* y_hat, const void* y, const int32_t* lengths, int32_t B, int32_t T,
            int32_t shift, float* loss_out, void* stream);

/* Optional access to activations of the last forward (wavenet.py:702 upsampled_local_features):
 * float [B, cin, T]. */
int wn_get_upsampled_features(wn_ctx* ctx, float* out, void* stream);

/* Per-tensor clip_by_norm -> clip_by_value -> TF-Adam -> EMA, in place (wavenet.py:586-613).
 * `step` is the 0-based global step before this update; lr is the already-scheduled rate. */
int wn_optim_step(wn_ctx* ctx, float* params, const float* grads, float* adam_m, float* adam_v,
                  float* ema, float lr, int64_t step, void* stream);
/* wavenet.py:615-629 (host helper). */
float wn_learning_rate(int32_t schedule, float init_lr, int64_t step, float decay_rate,
                       int64_t decay_steps, float warmup_steps);

/* ---- synthesis: replaces WaveNet.incremental (wavenet.py:724-911) ----------------------------- */
/* Fast-WaveNet generation of T = Tc*hop samples for B streams with ring-buffer queues.
 * c           float [B, cin, Tc]  (already transposed as wavenet.py:427)
 * noise       float [T, B, noise_per_step]: MoL: M uniforms u1 then 1 uniform u2 (mixture.py:91,104);
 *             Gaussian: 1 standard-normal draw (gaussian.py:50); softmax: Q uniforms (Gumbel-max form of
 *             tf.multinomial, wavenet.py:865).  NULL => drawn on the device: Philox4x32-10 keyed by `seed`, counter = the element
 *             index of this [T, B, noise_per_step] layout (24-bit uniforms clamped to [1e-5, 1 - 1e-5], mixture.py:91,104; Gaussian draws by
 *             Box-Muller) into a ctx-owned buffer -- replaces tf.random_uniform / Normal.sample / tf.multinomial's own generator;
 *             wn_fill_noise exposes the same stream so that a run can be reproduced with an explicit buffer.
 * test_inputs optional teacher forcing (wavenet.py:877-878): float [B,T] (scalar) / int32 [B,T] ids.
 * out_samples float [B,T] (scalar types) or int32 [B,T] (class ids)   (wavenet.py:874, 897-911)
 * out_raw     optional float [B, out_channels, T] raw network outputs  (wavenet.py:847, 904-908)     */
int wn_synthesize(wn_ctx* ctx, const float* c, int32_t B, int32_t Tc, const float* noise,
                  uint64_t seed, const void* test_inputs, void* out_samples, float* out_raw,
                  int32_t steps_per_graph, void* stream);
int wn_noise_per_step(const wn_ctx* ctx);
/* The device noise stream of wn_synthesize(noise = NULL, seed): fills float [T, B, noise_per_step]. */
int wn_fill_noise(wn_ctx* ctx, float* noise, int32_t B, int32_t T, uint64_t seed, void* stream);
/* wn_synthesize enqueues and returns (no device synchronisation).  The persistent pipeline bounds every hand-off spin; if one times
 * out (a workgroup was not resident) the kernels leave early and raise a device flag.  wn_synth_check waits for the LAST
 * wn_synthesize of this context to finish and returns WN_E_HIP (+ wn_last_error) if that happened, WN_OK otherwise; the next
 * wn_synthesize on the context reports a pending failure too.  wn_synth_last_path: 0 none yet, 1 launch-per-layer hipGraph path,
 * 2 persistent pipeline, 3 fp32 launch-per-layer path (compute_dtype = WN_COMPUTE_F32). */
int wn_synth_check(wn_ctx* ctx);
int wn_synth_last_path(const wn_ctx* ctx);
/* 1 if wn_synthesize(steps_per_graph <= 0) would run B streams of this model on the persistent pipeline (all of a CU's weights must
 * fit its 160 KiB of LDS next to 256 B of state per stream: one CU per 32 gate pairs, <= 8 CUs per layer, R, S <= 384, B <= 32), 0 if
 * it would take the launch-per-layer hipGraph path.  Host helper: a caller sends the whole batch in one run when this says 1 for it.
 * Per generated sample on the paper model (R = S = 256, 24 layers; deadline 45.35 us at 22.05 kHz): 28 us for 1 ... 12 streams, ~1.7 us per
 * stream beyond -- 16: 36 us, 20: 42.5 us (real time), 24: 49 us; a model whose CUs fit the chip more than once is cut into several runs side by
 * side in the same launch (wn_synth_last_instances: hparams.py's default model serves 20 streams at 26 us per sample). */
int wn_synth_pipe_eligible(const wn_ctx* ctx, int32_t B);
/* 16-bit storage type of the persistent pipeline's weights, hand-off granules and ring queues for the NEXT runs of this context
 * (fp32 accumulation either way): 1 = IEEE half (the default: raw outputs 1.2e-3 from the reference's fp32 loop at C4's model, 34.8 us per
 * sample), 0 = bf16 (8.7e-3, 35.0 us; 8 exponent bits).  A half run whose residual stream leaves the half range (|x| > 65504) is
 * reported by wn_synth_check / the next wn_synthesize as WN_E_HIP ("left the half-precision range"): switch to bf16 and run again.
 * Environment at wn_create: WN_PIPE_DTYPE=fp16|bf16 (case-insensitive; also f16 / half / float16 / bfloat16; anything else fails wn_create). */
int wn_synth_pipe_dtype(wn_ctx* ctx, int32_t half);
/* How the LAST wn_synthesize of this context ran, as the library configured it (not as the environment asked): fills up to `cap` of
 * out[0] path (as wn_synth_last_path) and, for a pipeline run, out[1] instances, [2] batched pre-multiplication, [3] kernel specialisation
 * (0 generic, 1 paper widths R = S = 256, 2 hparams.py widths R = S = 128), [4] storage (1 IEEE half, 0 bf16), [5] head CUs, [6] early
 * requests from this many streams, [7] abort word tested every n streams (0: once per sample), [8] workgroups launched, [9] streams of the
 * largest instance; zeros for the other paths.  Returns the number of entries written (10) or WN_E_ARG.  bench.py records it next to
 * every timed synthesis leg. */
int wn_synth_last_config(const wn_ctx* ctx, int32_t* out, int32_t cap);
/* how many pipeline INSTANCES the last wn_synthesize ran side by side (1 for a run of <= 10 streams or a model whose CUs fit the chip once:
 * the paper model takes 193 of 256; hparams.py's default model 81: up to three instances of <= 10 streams each). */
int wn_synth_last_instances(const wn_ctx* ctx);
/* 1 if the last wn_synthesize ran the persistent pipeline with the BATCHED pre-multiplication (R = 256 models, any eligible run of <= 32 streams: the
 * past taps and conditioning of every stream's next sample are multiplied in one [64 x K] x [K x streams] matrix product per sample and CU
 * instead of one matvec per stream), 0 otherwise.  Environment: WN_PIPE_BATCHPRE=0 keeps the per-stream form (A/B switch). */
int wn_synth_last_batched(const wn_ctx* ctx);

/* Stand-alone samplers on [B,O,T] parameters (train-time log path, wavenet.py:302-325). */
int wn_sample(wn_ctx* ctx, const float* y_hat, int32_t B, int32_t T, const float* noise /*[T,B,nps]*/,
              void* out /* float [B,T] or int32 [B,T] */, void* stream);

/* ---- mu-law codec (wavenet_vocoder/util.py:30-129; mu fixed to 255 as the reference does) ------ */
int wn_mulaw(const float* x, float* y, int64_t n, void* stream);
int wn_inv_mulaw(const float* y, float* x, int64_t n, void* stream);
int wn_mulaw_quantize(const float* x, int32_t* q, int64_t n, void* stream);        /* bit-exact vs numpy */
int wn_inv_mulaw_quantize(const int32_t* q, float* x, int64_t n, void* stream);
/* argmax over channels of [B,Q,T] logits -> int32 [B,T] (first max wins, like tf.argmax) */
int wn_argmax_channels(const float* logits, int32_t* out, int32_t B, int32_t Q, int32_t T, void* stream);

/* ---- introspection ------------------------------------------------------------------------------ */
int64_t wn_workspace_bytes(const wn_ctx* ctx);
/* name of the kernel that dominates training time (bench.py's roofline line names it) */
const char* wn_dominant_kernel_name(void);

/* ================================================================================================
 * WN_TEST_HOOKS -- NOT part of the drop-in surface.  Entry points used only by tests/, bench.py and tools/ to look inside a
 * context; a reference-side binding has no use for them.  Compile callers with -DWN_NO_TEST_HOOKS to hide the declarations.
 * ================================================================================================ */
#ifndef WN_NO_TEST_HOOKS
/* copy an internal activation buffer of the last step to `out` as fp32 ("X","U","TS","DZ","R1","H2","DY","DSKIP","DPRE1","GX0",
 * "GX1","cbt" are bf16 [rows][channels]; "YHAT","DC","CUP" fp32). */
int wn_debug_copy(wn_ctx* ctx, const char* name, int32_t layer, float* out, int64_t n, void* stream);
/* live HIP-event timing of the dominant training kernel (the gate GEMM, one launch per layer and step), recorded on the launch
 * stream between wn_profile(ctx,1) and wn_profile_result -- which SYNCHRONISES on the recorded events (bench only). */
int wn_profile(wn_ctx* ctx, int32_t enable);
int wn_profile_result(wn_ctx* ctx, double* total_ms, int64_t* launches);
/* the same launches by IN-KERNEL stamps (first workgroup's start .. last workgroup's end on the 100 MHz wall clock): the kernel's own
 * duration, as rocprofv3's kernel trace reports it -- without the wait behind the other stream's kernels the event bracket includes */
int wn_profile_kernel_result(wn_ctx* ctx, double* total_ms, int64_t* launches);
/* mean shader clock (MHz) INSIDE the same launches: workgroup 0 of each reads the shader-cycle counter and the 100 MHz wall clock at its start
 * and end.  The step runs at the chip's power limit (~1.35 kW), so the matrix peak that applies is 2500 TFLOP/s x clock / 2400 MHz. */
int wn_profile_kernel_clock(wn_ctx* ctx, double* mhz, int64_t* launches);
/* time rows (utterances x samples) one timed launch processed: the layer chain runs per half-batch on two streams */
int64_t wn_profile_rows_per_launch(const wn_ctx* ctx);
/* Device timeline of ONE training step from in-kernel stamps, no profiler attached (a profiler slows the host's enqueue enough to change
 * which stream runs ahead).  wn_trace_arm: every tile-engine and grouped weight-gradient launch of the `steps_from_now`-th next
 * wn_train_fwd .. wn_train_bwd stamps {first workgroup's start, last workgroup's end} on the 100 MHz wall clock into its own slot.
 * wn_trace_read (SYNCHRONISES the device): copies up to `cap` launches in enqueue order -- kind (0 gate, 1 out / skip / head conv,
 * 2 fp32-output GEMM, 3 d z, 4 head mask GEMM, 5 d x, 100 + n: grouped weight gradient with n A tiles), stream handle, start / end
 * ticks -- and returns their number (0 while no armed step has completed; negative wn_status on error). */
int wn_trace_arm(wn_ctx* ctx, int32_t steps_from_now);
int wn_trace_read(wn_ctx* ctx, int32_t cap, int32_t* kind, uint64_t* stream, uint64_t* start_ticks, uint64_t* end_ticks);
/* A/B switch of the stream structure.  0: default (2 when the batch has >= 2 utterances), 1: whole batch on the caller's stream,
 * 2: two half-batches on two streams */
int wn_set_batch_parts(wn_ctx* ctx, int32_t parts);
/* the dropout keep-mask of `layer` for the step seed, evaluated ON THE HOST by the very functions the kernels inline
 * (wn_layer_key / wn_drop_quad): out[i] = 1 if element first + i of the [rows][R] layer input is kept, else 0.  No context, no GPU:
 * pins the numpy mirror the parity tests hand to the oracle. */
int wn_test_dropout_mask(uint64_t seed, int32_t layer, float p, int64_t first, int64_t n, uint8_t* out);
/* which launches of this context take the 8-phase kernel (csrc/wn_tile8p.h; WN_GEMM8P in the environment of wn_create -- an A/B switch,
 * default 0): bit 0 = the gate GEMM, bit 1 = d x.  A model that does not fit the kernel reports 0 whatever the switch says. */
int wn_test_gemm8p_mask(const wn_ctx* ctx);
/* the workgroup tables of the persistent synthesis pipeline for a model of L layers x P CUs per layer cut into ni instances (host logic only: no
 * context, no GPU): role[b] of workgroup b = layer << 8 | j, bit 23: a head (j = its index), bits 24-25: instance, -1: none; blk = the inverse
 * (ni = 1: [L * P + heads]; else ni x [L * P + 1]).  Block b runs on XCD b % 8: a layer's P CUs share an XCD, consecutive layers stay together.
 * Returns how many instances of the model the chip holds (1 ... 3), WN_E_SHAPE if ni do not fit. */
int wn_test_pipe_layout(int32_t L, int32_t P, int32_t ni, int32_t* role, int32_t cap_role, int32_t* blk, int32_t cap_blk, int32_t* grid, int32_t* heads);
#endif /* WN_NO_TEST_HOOKS */

#ifdef __cplusplus
}
#endif
#endif /* WAVENET_MI355_H */
