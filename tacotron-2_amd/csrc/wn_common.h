// Internal definitions shared by the HIP translation units of libwavenet_mi355.so.
// gfx950 (MI355X / CDNA4) only -- no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/wavenet_mi355.h"

typedef unsigned short bf16_t;                                    // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;      // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;      // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define WN_SQRT_HALF 0.70710678118654752440f

// ------------------------------------------------------------------------------------------------
// bf16 helpers (round-to-nearest-even, identical to torch's float->bfloat16).  On the device the conversion is the native fptrunc
// (hipcc selects v_cvt_pk_bf16_f32 on gfx950: ONE instruction per pair instead of ~14 integer ops for two software roundings --
// the fused epilogues convert 2-4 values per output element); same result for every non-NaN input, NaNs come back quiet.
__host__ __device__ inline bf16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(bf16_t, (__bf16)f);
#else
    union { float f; uint32_t u; } v; v.f = f;
    if ((v.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((v.u >> 16) | 0x40);   // NaN
    uint32_t r = v.u + 0x7fffu + ((v.u >> 16) & 1u);
    return (bf16_t)(r >> 16);
#endif
}
__host__ __device__ inline float bf2f(bf16_t h) {
    union { float f; uint32_t u; } v; v.u = ((uint32_t)h) << 16; return v.f;
}
__device__ inline uint32_t pack_bf2(float lo, float hi) {
    typedef __bf16 bf16x2_native_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_native_t __attribute__((ext_vector_type(2)));
    const f32x2_native_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_native_t));
}

// ------------------------------------------------------------------------------------------------
// Dropout mask: counter-based hash, 64 bits per QUAD of elements (16 bits each).
// element e (flat index in the [rows][R] layer input) of layer-key `key` is KEPT iff
// bits16(e) >= thresh16, thresh16 = round(p * 65536).  Quad q = e >> 2:  a = mix(q ^ key_lo),  b = mix(a + key_hi);
// elements 4q, 4q+1 take the low / high half of b, elements 4q+2, 4q+3 those of a ^ rotl(b, 16).  (Round 1 hashed every PAIR
// twice: 4 quarter-rate v_mul_lo_u32 per two elements in the out-conv and dx epilogues; this is 4 per four.)
// Spec mirrored in python (tests/hip_util.py) bit for bit.
__host__ __device__ inline uint32_t wn_mix32(uint32_t x) {          // murmur3 finaliser
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16; return x;
}
__host__ __device__ inline void wn_drop_quad(uint32_t key_lo, uint32_t key_hi, uint32_t quad_index, uint32_t& w0, uint32_t& w1) {
    const uint32_t a = wn_mix32(quad_index ^ key_lo), b = wn_mix32(a + key_hi);
    w0 = b; w1 = a ^ ((b << 16) | (b >> 16));
}
__host__ __device__ inline uint32_t wn_drop_word(uint32_t key_lo, uint32_t key_hi, uint32_t pair_index) {      // elements 2*pair_index, +1
    uint32_t w0, w1; wn_drop_quad(key_lo, key_hi, pair_index >> 1, w0, w1);
    return (pair_index & 1u) ? w1 : w0;
}
__host__ __device__ inline void wn_layer_key(uint64_t seed, int layer, uint32_t* lo, uint32_t* hi) {
    uint64_t k = seed * 0x9E3779B97F4A7C15ull + (uint64_t)(layer + 1) * 0xD1B54A32D192ED03ull;
    k ^= k >> 29;
    *lo = (uint32_t)k; *hi = (uint32_t)(k >> 32);
}

// ------------------------------------------------------------------------------------------------
struct WnTensor {
    std::string name;
    int32_t shape[4];
    int32_t ndim;
    int64_t offset;       // floats into the flat buffer
    int64_t numel;
};

struct WnLayerOffsets {   // offsets (floats) into the flat parameter buffer
    int64_t dil_k, cin_k, dil_b, cin_b, skip_k, out_k, skip_b, out_b;
    int64_t gin_k = -1, gin_b = -1;      // global-conditioning 1x1 conv [1, gin, G] (+ bias); -1 when gin_channels <= 0
};

// A-operand pack descriptor: Wpk[m][k] = scale * params[base + (k - k0) * stride_k + perm(m) * stride_m]
struct PackSeg { int64_t base; int32_t k0, nk, stride_k, stride_m; float scale; };

struct PackedW {          // one fragment-ordered bf16 matrix
    bf16_t* dev = nullptr;
    int32_t M = 0, K = 0;             // padded: M % 32 == 0, K % 16 == 0
    int32_t M_valid = 0;
    int32_t gate_interleave = 0;      // row permutation (see wn_pack.hip)
    int32_t GH = 0;
    int32_t kil = 0;                  // > 0: the three dilated taps are interleaved along K in blocks of `kil` channels (32: wn_gemm_lds_kernel's TAPS
                                      //      staging; 64: the K-tiles of wn_gemm8p_kernel)
                                      //      [tap0 blk0 | tap1 blk0 | tap2 blk0 | tap0 blk1 | ...] (tile engine: taps of one k-block back to back => L2 reuse)
    std::vector<PackSeg> segs;
    PackSeg* dev_segs = nullptr;
};

struct WnLayerPacks { PackedW w1, wo, ws, w2T, w1T; };

struct wn_ctx {
    wn_config cfg;
    std::string err;
    int L, R, G, GH, S, O, C, Cin, hop;
    int OP;                               // out_channels padded to 32
    int CP;                               // cin padded to 32 (M of the d_c GEMM)
    std::vector<int> dil;
    std::vector<float> skip_scale;        // c_l of the legacy skip recursion (wavenet.py:706-715)
    float res_scale;                      // sqrt(.5) if residual_legacy else 1
    std::vector<WnTensor> tensors;
    int64_t n_params = 0;                 // floats of the EFFECTIVE parameter buffer the kernels read (ctx-owned params_dev)
    // weight normalisation: the caller's flat buffers (params / grads / Adam slots / EMA) use the RAW layout `raw_tensors`
    // (kernel = v, g, bias, ...); params_dev holds g * v / ||v||.  Without it raw == effective.
    bool wnorm = false; int64_t n_raw = 0; std::vector<WnTensor> raw_tensors;
    struct WnMap { int64_t raw_off, eff_off, numel, g_off; int32_t cout, pad; };
    std::vector<WnMap> wmap; void* wmap_dev = nullptr; float* raw_dev = nullptr; float* deff = nullptr;
    WnLayerOffsets first;                 // dil_k = input kernel, dil_b = input bias
    std::vector<WnLayerOffsets> lay;
    int64_t fin1_k, fin1_b, fin2_k, fin2_b;
    std::vector<int64_t> up_k, up_b;
    // packed weights
    float* params_dev = nullptr;          // ctx-owned fp32 copy of the flat parameters taken at wn_pack_weights
    bool packed = false;
    int cup_final_idx = 0;                // CUP[cup_final_idx] = upsampled conditioning [B,C,T] fp32
    std::vector<WnLayerPacks> packs;
    PackedW wskip, wh1, wh2, wh2T, wh1T, wcT;
    void* pack_jobs_dev = nullptr; int pack_njobs = 0, pack_nblocks = 0;   // table of the single pack launch
    float* b1sum = nullptr;               // [L][G] dil bias + cin bias
    float* skip_bias_total = nullptr;     // [S]
    // use_bias=False (hparams.py:189): the residual layers have no bias variables.  Forward READS then go to a zero tail
    // behind the ctx-owned parameter copy (offset n_params .. n_params + zpad); gradient WRITES are skipped (lbias == false).
    bool lbias = true; int zpad = 0;
    // global conditioning: gin > 0.  gvec [maxB][gin] (embedded or given g), gids [maxB], gbias [L][maxB][G] =
    // b1sum + W_g^T g + b_g per utterance (the gate epilogue's bias then has an utterance stride), colsum [L][maxB][G] =
    // sum_t dz per utterance (backward), emb_off = embedding table [n_speakers][gin] or -1.
    int gin = 0; int64_t emb_off = -1; float* gvec = nullptr; int32_t* gids = nullptr; float* gbias = nullptr; float* colsum = nullptr;
    int gB = 0; bool have_g = false;
    int32_t* tensor_offsets_dev = nullptr; // [ntensors+1] for the optimiser
    float* norm2_dev = nullptr;           // [ntensors]
    int32_t* norm_spans_dev = nullptr; int32_t* norm_first_dev = nullptr; float* norm_part_dev = nullptr; int norm_nspans = 0;   // atomic-free clip norms
    // workspace
    char* ws = nullptr; size_t ws_bytes = 0;
    int maxB, maxT; int64_t NT;
    bf16_t* XD;                           // [L][NT][R] dropout-applied layer inputs (aliases X when dropout == 0)
    bf16_t *cbt, *X, *TS, *U, *R1, *H2, *DY, *DPRE1, *DSKIP, *DZ, *GX0, *GX1;
    float *YHAT, *DC, *CUP[WN_MAX_UPSAMPLE + 1], *DCUP[2];
#define WN_CS_MAXBLK 512                  // row blocks of wn_colsum2 (wn_frontend.hip); sizes cs_part (wn_api.hip)
#define WN_CS_SLOTS 2
    float* cs_part = nullptr;             // partial sums of wn_colsum2: WN_CS_SLOTS regions of WN_CS_MAXBLK x 2 x 1024 floats
    float* UPPART = nullptr; int64_t uppart_floats = 0;   // partial sums of the upsample-kernel gradients (two-stage, no atomics)
    void* XIN; float* CIN;                // ctx-owned copies of the step's x and c (pointers are borrowed per call)
    float* wg_partial = nullptr; size_t wg_partial_bytes = 0;   // split-K partial tiles of the grouped wgrad (wn_wgrad.h)
    bf16_t* GXall = nullptr;              // [L+1][NT][R] gradient wrt every layer input (kept for the grouped W_out wgrad)
#define WN_ZERO_PAGE_BYTES 4096
#define WN_PIPE_F16_DEFAULT true       // storage type of the persistent synthesis pipeline when WN_PIPE_DTYPE is not set: IEEE half (1.2e-3 vs bf16's 8.7e-3 from the fp32 loop, same speed)
    bf16_t* zero_page = nullptr;          // zeros: DMA source for out-of-range rows / k-steps (wn_gemm_lds_kernel: 16 B per lane from one address;
                                          // wn_gemm8p_kernel's SGPR-base form: base + 16 * lane, i.e. 1 KiB)
    int gemm8p = 0;                       // WN_GEMM8P at wn_create, bit 0: gate, bit 1: d x on the 8-phase kernel (wn_tile8p.h) where the model fits it; default 0: the 256 x 128 LDS-DMA ring kernel
    float* scal;                          // device scalars: [0]=loss sum [1]=denominator [2]=1/denominator [3]=count
    // state of the last forward
    int fB = 0, fT = 0, fTc = 0; uint64_t fseed = 0; bool have_fwd = false; bool have_loss = false;
    const void* fx = nullptr; const void* fy = nullptr; const int32_t* flen = nullptr; const float* fc = nullptr;
    // live profiling of the dominant kernel (bench.py roofline): event pairs around every gate-GEMM launch
    bool prof = false; std::vector<hipEvent_t> pev; size_t pev_used = 0;
    // WN_DEVTRACE=<file> (debug): in-kernel {first start, last end} stamps of EVERY tile-engine launch of one training step (the
    // WN_DEVTRACE_STEP-th wn_train_fwd, default 8) -- the device timeline of the chain without a profiler attached (rocprofv3 slows the
    // host's enqueue enough to change which stream runs ahead).  Written two steps later by wn_devtrace_poll.
    unsigned long long* trace_dev = nullptr; int trace_n = 0, trace_calls = 0, trace_state = 0, trace_arm_at = 0;      // state: 0 idle, 1 recording, 2 recorded
#define WN_TRACE_MAX 1024
    struct { int epi; void* st; int rows; } trace_tag[WN_TRACE_MAX];
    unsigned long long* kprof_dev = nullptr;      // [WN_KPROF_MAX][2] in-kernel {first start, last end} stamps of the timed gate launches
    unsigned long long* kclk_dev = nullptr;       // [WN_KPROF_MAX][2] {shader cycles, 100 MHz ticks} of workgroup 0 of the same launches (wn_profile_kernel_clock)
#define WN_KPROF_MAX 8192
    // batch parts: the serial layer chain of the two half-batches runs on two streams so that the MFMA/power-bound GEMMs of
    // one half overlap the HBM-bound kernels of the other (fwd: gate | out conv, bwd: dx | dgate); joined before the loss / wgrads
#define WN_MAX_PARTS 4
    hipStream_t st2 = nullptr;            // part 1 (and the "side" work of the backward tail)
    hipStream_t stp[WN_MAX_PARTS] = {};   // stp[k], k >= 2: further batch parts (stp[1] aliases st2)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_pjoin[WN_MAX_PARTS] = {}; int parts = 1; int parts_req = 0; int prof_rows = 0;
    // gradient buckets (wn_train.hip: wn_plan_buckets): weight gradients run bucket by bucket on a third, low-priority stream under
    // the serial backward chain; ev_bucket[k] = bucket k of the flat gradient is final (index WN_MAX_BUCKETS: the whole buffer)
#define WN_MAX_BUCKETS 8
    hipStream_t st3 = nullptr; hipEvent_t ev_chain[WN_MAX_PARTS][WN_MAX_BUCKETS] = {}; hipEvent_t ev_bucket[WN_MAX_BUCKETS + 2] = {}; hipEvent_t ev_w0 = nullptr;
    hipEvent_t ev_head[WN_MAX_PARTS] = {};   // d pre1 of a batch part exists (the head weight gradients may start under the chain)
    int nbuckets = 0, nbuckets_early = 0, nearly_live = 0; int bucket_lo[WN_MAX_BUCKETS + 2] = {}, bucket_hi[WN_MAX_BUCKETS + 2] = {};
    int64_t bucket_off[WN_MAX_BUCKETS + 2] = {}, bucket_cnt[WN_MAX_BUCKETS + 2] = {}; bool have_bwd = false;
    bool inference = false;               // cfg.inference_only: no training workspace, synthesis state pre-sized at wn_create
    float* noise_buf = nullptr; size_t noise_bytes = 0;      // device-drawn sampling noise [T][B][nps] (wn_synthesize with noise == NULL)
    int synth_path = 0;                   // 0 none, 1 graph, 2 pipeline, 3 fp32 graph (wn_synth_last_path)
    // synthesis state (lazy)
    struct Synth* synth = nullptr;
    void* synth32 = nullptr;              // fp32 synthesis state (wn_synth_f32.hip; cfg.compute_dtype = WN_COMPUTE_F32)
    void* pipe = nullptr;                 // persistent synthesis pipeline state (wn_synth_pipe.hip)
    bool pipe_f16 = false;                // persistent pipeline: IEEE-half weights / hand-offs / queues instead of bf16 (WN_PIPE_DTYPE=fp16|bf16 at wn_create, wn_synth_pipe_dtype)
    int synth_instances = 0;              // pipeline instances the last wn_synthesize ran side by side (wn_synth_last_instances)
    int synth_batchpre = 0;               // the last pipeline run multiplied every stream's past taps / conditioning in ONE matrix product per sample (wn_synth_last_batched)
#define WN_SYNTH_CFG_N 10
    int32_t synth_cfg[WN_SYNTH_CFG_N] = {};   // how the last pipeline run was configured (wn_synth_last_config): [1] instances [2] batched pre-multiplication [3] kernel specialisation
                                          // (0 generic, 1 paper widths, 2 hparams.py widths) [4] 1 = IEEE-half storage [5] head CUs [6] early requests from n streams [7] abort test every
                                          // n streams (0: once per sample) [8] workgroups launched [9] streams of the largest instance
    int pipe_cap = 0;                     // inference-only contexts: streams of ONE pipeline run the pre-sized buffers hold (0: pipeline not used / not limited)
    void* f32 = nullptr;                  // fp32-forward state (wn_f32.hip), allocated on the first forward of a cfg.compute_dtype = WN_COMPUTE_F32 context
    bool fwd_was_f32 = false;
    float* dy32_next = nullptr;           // the next wn_loss_run also writes d y_hat in fp32 here ([rows][ldDY]; fp32 training mode)
};

extern std::string g_create_err;

#define WN_FAIL(ctx, code, ...) do { char _b[512]; snprintf(_b, sizeof _b, __VA_ARGS__); \
    if (ctx) (ctx)->err = _b; else g_create_err = _b; return (code); } while (0)
#define WN_HIP(ctx, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) \
    WN_FAIL(ctx, WN_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)
#define WN_LAUNCH_CHECK(ctx) WN_HIP(ctx, hipGetLastError())

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- cross-TU entry points (host) ---------------------------------------------------------------
int wn_build_packs(wn_ctx* ctx);
int wn_launch_pack(wn_ctx* ctx, const float* params, hipStream_t st);
int wn_fwd_impl(wn_ctx* ctx, hipStream_t st, float* loss_out, float* y_hat_out);
int wn_bwd_impl(wn_ctx* ctx, float* grads, hipStream_t st);
int wn_optim_impl(wn_ctx* ctx, float* p, const float* g, float* m, float* v, float* ema, float lr, int64_t step, hipStream_t st);
int wn_synth_impl(wn_ctx* ctx, const float* c, int B, int Tc, const float* noise, uint64_t seed,
                  const void* test_inputs, void* out_samples, float* out_raw, int steps_per_graph, hipStream_t st);
void wn_synth_free(wn_ctx* ctx);
void wn_synth_f32_free(wn_ctx* ctx);
int wn_synth_f32_reserve(wn_ctx* ctx, int B);
int wn_synth_f32_impl(wn_ctx* ctx, const float* c, int B, int Tc, const float* noise, const void* test_inputs,
                      void* out_samples, float* out_raw, int steps_per_graph, hipStream_t st);      // fp32 weights / queues / accumulation
void wn_pipe_free(wn_ctx* ctx);
bool wn_pipe_eligible(const wn_ctx* ctx, int B);
int wn_pipe_reserve(wn_ctx* ctx, int B, int T);            // size every pipeline buffer for (B, T) (no-op when already large enough)
int wn_synth_reserve(wn_ctx* ctx);                        // state of the launch-per-layer graph path
int wn_pipe_check(wn_ctx* ctx, bool wait);                // pending abort flag of the last pipeline run -> WN_E_HIP
int wn_noise_reserve(wn_ctx* ctx, int B, int T);
int wn_fill_noise_impl(wn_ctx* ctx, float* noise, int B, int T, uint64_t seed, hipStream_t st);
int wn_pipe_synthesize(wn_ctx* ctx, const float* c, int B, int Tc, const float* noise, const void* test_inputs,
                       void* out_samples, float* out_raw, hipStream_t st);
extern "C" int wn_noise_per_step(const wn_ctx* c);
int wn_upsample_fwd(wn_ctx* ctx, const float* params_unused, const float* c, int B, int Tc, hipStream_t st);
int wn_weightnorm_apply(wn_ctx* ctx, const float* raw_params, hipStream_t st);     // raw (v, g, bias) -> params_dev (effective)
int wn_weightnorm_grad(wn_ctx* ctx, float* raw_grads, hipStream_t st);             // deff (effective grads) -> raw grads
int wn_gbias_fwd(wn_ctx* ctx, int B, hipStream_t st);                 // global-conditioning bias table of this batch
void wn_devtrace_poll(wn_ctx* c, hipStream_t st, bool step_start);
// Device timeline of the kernels that carry no in-kernel stamps (upsample net, input convolution, loss, column sums, optimiser ...): while
// a trace is being recorded, a one-thread stamp kernel in front of and behind the group on ITS stream writes the wall clock into the
// group's slot (start = the stream reached the group, end = its last kernel retired, each +- one dispatch).  Nothing is enqueued otherwise.
int wn_trace_scope_begin(wn_ctx* c, hipStream_t st, int kind);       // slot or -1
void wn_trace_scope_end(wn_ctx* c, hipStream_t st, int slot);
struct WnTraceScope {
    wn_ctx* c; hipStream_t st; int slot;
    WnTraceScope(wn_ctx* c_, hipStream_t st_, int kind) : c(c_), st(st_), slot(wn_trace_scope_begin(c_, st_, kind)) {}
    ~WnTraceScope() { if (slot >= 0) wn_trace_scope_end(c, st, slot); }
};
enum { WN_TR_UPSAMPLE_FWD = 201, WN_TR_INPUT_CONV = 202, WN_TR_LOSS = 203, WN_TR_COLSUM = 204, WN_TR_UPSAMPLE_BWD = 205, WN_TR_OPTIMISER = 206, WN_TR_INPUT_CONV_BWD = 207 };
int wn_gin_bwd(wn_ctx* ctx, float* grads, hipStream_t st, bool have_colsum = false);           // d W_g, d b_g, d embedding
int wn_colsum2(wn_ctx* c, const bf16_t* M, int ld, int ncols, int nvalid, const float* xw, int64_t rows, float* out_b, float* out_w, int slot, hipStream_t st);
size_t wn_wgrad_partial_need(wn_ctx* ctx);
void wn_plan_buckets(wn_ctx* ctx);
int wn_f32_forward(wn_ctx* ctx, hipStream_t st);                  // fp32-accurate forward into YHAT (wn_f32.hip)
void wn_f32_free(wn_ctx* ctx);
int wn_f32_backward(wn_ctx* ctx, float* grads, hipStream_t st);       // fp32 backward of the last fp32 forward (wn_f32.hip)
float* wn_f32_dy(wn_ctx* ctx);                                        // fp32 d y_hat buffer of the fp32 state (allocates it)
int wn_upsample_bwd(wn_ctx* c, const float* dc_final, float* grads, hipStream_t st);
int wn_loss_fwd_bwd(wn_ctx* c, float* loss_out, hipStream_t st);
const float* wn_f32_debug(const wn_ctx* ctx, const char* name, int layer);
int wn_sample_impl(wn_ctx* ctx, const float* y_hat, int B, int T, const float* noise, void* out, hipStream_t st);
