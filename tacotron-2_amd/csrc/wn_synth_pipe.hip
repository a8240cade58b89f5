// Fast-WaveNet autoregressive synthesis as ONE persistent, weight-stationary dataflow kernel
// (replaces WaveNet.incremental, wavenet.py:724-911, and the per-layer launches of wn_synth.hip).
//
// Why: the sample-to-sample dependency chain crosses every layer, so a step costs (number of dependent kernel
// boundaries) x ~1.5 us when layers are launches -- 2L+3 = 51 boundaries = 75 us before any arithmetic, against a
// real-time budget of 45.35 us per sample at 22.05 kHz.  Here nothing is launched per sample:
//   * every layer is owned by P = gate_channels/64 workgroups, one per CU (142 KiB of LDS each); CU (l, j) keeps in LDS,
//     for the whole utterance, the rows of [W_dil | W_cin] of ITS 32 tanh/sigmoid gate pairs (all 3 taps + conditioning)
//     and the COLUMNS of W_out / W_skip that multiply its 32 gate outputs.  27 MB of bf16 weights are read from HBM once;
//   * a step is a message that travels the ring  head -> layer 0 -> ... -> layer L-1 -> head.  CU (l, j) sums the P
//     partial vectors it receives into x_l(t), finishes z = z_past + W_tap2 x_l(t) (the taps t-d, t-2d and the
//     conditioning were pre-multiplied while the message was elsewhere), gates, and publishes ITS partial of
//     x_{l+1}(t) = rho (x_l + W_out u + b): one exchange per layer, no barrier anywhere;
//   * hand-off = data-tagged 16-byte granules {3 payload words, tag = t+1} written with ONE write-through (sc1) store
//     and polled with sc1 loads: no flag, no fence (tools/hop_probe.hip: 1.06 us per P=8 hop, 0.71 us for P=4);
//     single-buffered mailboxes are safe because sample t exists only after every CU consumed step t;
//   * the skip sum travels the same ring as P independent fp32 running sums (CU (l, j) -> (l+1, j)) and is reduced by
//     the head CU, which also holds the two head convolutions, samples (MoL / Gaussian / categorical, mixture.py:76-107,
//     gaussian.py:39-52, wavenet.py:861-867) and applies the input convolution of the next step;
//   * streams of a batch are independent messages that follow each other through the ring (pipelined, not batched):
//     B = 8 costs the same wall time per sample as B = 1;
//   * queues are ring buffers in HBM (one private copy per CU, 4d slots per layer, zero-initialised == the reference's
//     zero queues, wavenet.py:815-816), touched off the critical path only.
// Every spin loop is bounded; a timeout raises a device flag that makes all workgroups leave.
// Round 5: a run of more than ~12 streams is bound by a layer CU's service time per stream, and with ONE wave per SIMD that time is an
// instruction count.  The pre-multiplication of every stream's next sample is one matrix product per sample on the matrix cores (pre_stash / pre_batch),
// streams alternate between two head CUs, the mailbox copy nobody reads is not written, the publishing stores are builtins the compiler's waitcnt
// pass can count (st_buf16), the skip chain needs no barrier, and the paper model's / hparams.py's widths are compile-time constants (SPEC 1 / 2):
// 28 us per sample for 1 ... 12 streams of the paper model, 42.5 us at hparams.py's synthesis batch of 20 (real time at 22.05 kHz; was 72).
#include "wn_common.h"
#include <algorithm>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

#define PIPE_THREADS 256
#define PIPE_XG 128           // granule slots per x partial: 4 bf16 channels each (R/4 used; lane g polls g and g+64)
#define PIPE_SG 128           // granules per skip partial: 3 fp32 channels each
#define PIPE_SPIN_LIMIT 3000000
#define PIPE_ZS 68             // floats of z_past per stream: 64 rows + 4 (the batched pre-multiplication stores float4s: conflict-free at a 272-B pitch)

struct PipeArgs {
    int32_t L, P, R, G, GH, S, O, OP, C, Cin, B, T;
    float rho; int32_t mode, nps; float lsmin; int32_t start_id, spx;
    const char* slices; int64_t layer_slice_bytes, head_slice_off;
    int32_t off_w1c, off_w1p, off_wo, off_ws, off_zb, off_ob, layer_lds_static;      // byte offsets inside a layer slice / LDS image
    int32_t hoff_wh1, hoff_wh2, hoff_b1, hoff_b2, hoff_sb, hoff_win, hoff_bin, head_lds_static;
    u32x4* XM; u32x4* SM;            // mailboxes written with write-through (sc1) stores: visible to every XCD
    u32x4* XML; u32x4* SML;          // the same mailboxes written with plain stores: they live in the WRITER's XCD L2, readable (sc1 loads) by CUs of that XCD only
    int32_t* xcc_tab;                // [grid] XCC id + 1 of every workgroup (0 = not started)
    const int32_t* role_tab;         // [grid] role of every workgroup: layer << 8 | j, bit 23: head, -1: none (host-built: which XCD hosts which layers)
    const int32_t* block_tab;        // [L * P + NH] inverse: workgroup id of CU (layer, j); [L * P + h]: head h
    int32_t abort_every;             // 1: layer CUs test the abort word after every stream (rounds 2-4), 0: once per sample
    int32_t early_from;              // runs of at least this many streams request the NEXT stream's x granules and this stream's skip granule one stage early (see the sample loop)
    int32_t NH;                      // head CUs of a single-instance run: head h serves the streams s = h (mod NH) (instances side by side: one head each)
    int32_t ninst, iB[4], is0[4];    // pipeline instances side by side in ONE launch (round 5): instance i serves the streams [is0[i], is0[i] + iB[i]) of the
                                     // batch of B on its own CUs (role table: bits 24-25), mailboxes and ring queues (offsets linear in is0); ninst = 1: iB[0] = B
    bf16_t* ring; int64_t ring_unit; const bf16_t* cbt;      // ring_unit: ring elements of ONE stream over all layers and CUs (instance offset = ring_unit * is0)
    const float* noise; const void* test_inputs; void* out_samples; float* out_raw;
    const float* win_global; const float* bin_global;
    int32_t* abort_flag;
    const float* gbias;              // global conditioning: [L][B][G] gate bias per stream (b_dil + b_cin + W_g^T g_s + b_g), else null
    unsigned long long* trace; int32_t trace_t0, trace_n;     // optional timestamps (WN_PIPE_TRACE=1): [trace_n][2*(L+2)] of s_memrealtime
    unsigned long long* svc; int32_t svc_l, svc_s;            // optional (WN_PIPE_SVC_TRACE=1): [trace_n][16] stamps of ONE layer CU's iteration for one stream (where its service time goes)
    int64_t ring_off[32]; int64_t cin_b_off[32]; int32_t ring_mask[32]; int32_t dil[32];
};

// ---- granule I/O: 16 bytes, one write-through store / one L1-bypassing load ------------------------------------------
__device__ __forceinline__ void st_g16(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_g16_local(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u32x4 ld_g16(const u32x4* p) {
    u32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v;
}
// the same load WITHOUT the wait: the caller waits by hand (pf_wait) before touching the register
__device__ __forceinline__ void ld_g16_nowait(u32x4& v, const u32x4* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void pf_wait(u32x4& a, u32x4& b) { asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) :: "memory"); }
__device__ __forceinline__ void ld2_g16(const u32x4* p0, const u32x4* p1, u32x4& v0, u32x4& v1) {
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1) : "v"(p0), "v"(p1) : "memory");
}
// workgroup barrier for LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would wait ~1 us for the acknowledgement
// of the write-through granule stores that were just issued
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Polling loads as buffer loads with the sc1 cache policy (aux = 16): the compiler tracks them in vmcnt (no hand-written waits).
// (Keeping 3 polls in flight, one every ~1/3 round trip, was tried: the extra fabric traffic slows EVERY hop, 46 -> 52 us/step.)
__device__ __forceinline__ u32x4 poll_ld(__amdgpu_buffer_rsrc_t r, int byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
    asm volatile("" ::: "memory");          // keeps successive polls of the same address from being merged
    return v;
}
// (the base is wave-uniform by construction -- mailbox of (layer, stream), chosen by a per-wave flag -- but hipcc cannot prove it and wraps every
// load through the descriptor in a waterfall loop of readfirstlane / compare / branch: the descriptor is built from readfirstlane'd words)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t poll_rsrc(const void* base, int bytes) {
    const uint64_t b = (uint64_t)base;
    const uint64_t u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    return __builtin_amdgcn_make_buffer_rsrc((void*)u, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Publishing stores the COMPILER can count (fast path of the layer CUs).  vmcnt retires in order and counts stores: hipcc does not see a store inside
// inline asm, so every wait it emits for a later load (`vmcnt(number of younger loads)`) also waits for the acknowledgement of an older asm store -- ~1 us
// for a write-through one.  As builtins, issued unconditionally by every wave (a lane / wave / copy that has nothing to publish passes offset -1: beyond
// num_records, the hardware drops the write), they are part of the count on every path and a wait skips exactly the stores younger than what it needs.
__device__ __forceinline__ void st_buf16(__amdgpu_buffer_rsrc_t r, int byte_off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 0); }
__device__ __forceinline__ void st_buf16_wt(__amdgpu_buffer_rsrc_t r, int byte_off, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 17 /* sc0 sc1 */); }

__device__ __forceinline__ bool pipe_aborted(const int32_t* f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }
__device__ __forceinline__ void pipe_abort(int32_t* f, int code) { __hip_atomic_store(f, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// cross-lane moves on the DPP path (a few cycles) instead of __shfl (ds_bpermute: an LDS round trip each)
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
#define DPP_XOR1 0xB1            // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E            // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141    // lane i <- lane 7-i inside every group of 8 (reaches the other quad)
#define DPP_QUAD_BCAST(k) ((k) * 0x55)

__device__ __forceinline__ float dot8(const uint4 w, const uint4 x, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.x), __builtin_bit_cast(bf16x2_t, x.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.y), __builtin_bit_cast(bf16x2_t, x.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.z), __builtin_bit_cast(bf16x2_t, x.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.w), __builtin_bit_cast(bf16x2_t, x.w), acc, false);
    return acc;
}

// ---- the pipeline's 16-bit storage type (weights in LDS / registers, hand-off granules, ring queues): H = 0 bf16 (v_dot2_f32_bf16),
// H = 1 IEEE half (v_dot2_f32_f16: same LDS footprint, same rate, 3 more mantissa bits; fp32 accumulation either way).  The reference's loop
// is fp32 (modules.py:273-303); the distance to it is a dtype choice: measured in tests/test_hip_round5.py for both.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
template <int H> __device__ __forceinline__ uint16_t f2n(float f) {
    if constexpr (H) return __builtin_bit_cast(uint16_t, (_Float16)f); else return f2bf(f);
}
template <int H> __device__ __forceinline__ float n2f(uint16_t h) {
    if constexpr (H) return (float)__builtin_bit_cast(_Float16, h); else return bf2f(h);
}
template <int H> __device__ __forceinline__ uint32_t pack_n2(float lo, float hi) {
    if constexpr (H) { typedef float f32x2_t __attribute__((ext_vector_type(2))); const f32x2_t v = {lo, hi}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t)); }
    else return pack_bf2(lo, hi);
}
template <int H> __device__ __forceinline__ float dot8n(const uint4 w, const uint4 x, float acc) {
    if constexpr (H) {
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w.x), __builtin_bit_cast(f16x2_t, x.x), acc, false);
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w.y), __builtin_bit_cast(f16x2_t, x.y), acc, false);
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w.z), __builtin_bit_cast(f16x2_t, x.z), acc, false);
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, w.w), __builtin_bit_cast(f16x2_t, x.w), acc, false);
        return acc;
    } else return dot8(w, x, acc);
}
// 8 bf16 values (the upsampled conditioning the training path shares) -> the pipeline's storage type
template <int H> __device__ __forceinline__ uint4 cvt8_bf16(const uint4 v) {
    if constexpr (H) {
        const uint32_t in[4] = {v.x, v.y, v.z, v.w}; uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pack_n2<1>(bf2f((bf16_t)(in[i] & 0xffffu)), bf2f((bf16_t)(in[i] >> 16)));
        return make_uint4(o[0], o[1], o[2], o[3]);
    } else return v;
}

// matvec over k-chunks [kc0, kc1) of a [kchunk][rows][8] 16-bit LDS image against a 16-bit vector in LDS; this lane's row
template <int H> __device__ __forceinline__ float mv_rows(const char* W, int rows, int row, const char* vec, int kc0, int kc1) {
    float a0 = 0.0f, a1 = 0.0f;
    int kc = kc0;
    for (; kc + 1 < kc1; kc += 2) {
        a0 = dot8n<H>(*reinterpret_cast<const uint4*>(W + ((size_t)kc * rows + row) * 16), *reinterpret_cast<const uint4*>(vec + kc * 16), a0);
        a1 = dot8n<H>(*reinterpret_cast<const uint4*>(W + ((size_t)(kc + 1) * rows + row) * 16), *reinterpret_cast<const uint4*>(vec + (kc + 1) * 16), a1);
    }
    if (kc < kc1) a0 = dot8n<H>(*reinterpret_cast<const uint4*>(W + ((size_t)kc * rows + row) * 16), *reinterpret_cast<const uint4*>(vec + kc * 16), a0);
    return a0 + a1;
}

// ======================================================================================================================
// slice builder: fp32 parameters -> the bf16 LDS images of every CU (run once per wn_pack_weights + synthesis)
struct SliceJob { int64_t dst; int64_t src; int32_t rows, kchunks, stride_k, stride_row, kind; float scale; int32_t row_perm_base, row_perm_split, pad; };
// kind 0: bf16 image [kc][row][8], element (kc,row,e) = scale * params[src + (kc*8+e)*stride_k + rowsrc(row)*stride_row]
//         rowsrc(row) = row < split ? base + row : GHoff + base + (row - split)   (gate pairs; split = 0 -> identity + base)
// kind 1: fp32 vector of `rows` floats, element row = scale * params[src + rowsrc(row)]
__global__ void wn_pipe_slice_kernel(const float* __restrict__ params, char* __restrict__ slices, const SliceJob* __restrict__ jobs, const int* __restrict__ job_block0, int njobs, int GH, int f16) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (job_block0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const SliceJob jb = jobs[lo];
    const int64_t idx = (int64_t)(blockIdx.x - job_block0[lo]) * blockDim.x + threadIdx.x;
    auto rowsrc = [&](int row) { return jb.row_perm_split ? (row < jb.row_perm_split ? jb.row_perm_base + row : GH + jb.row_perm_base + (row - jb.row_perm_split)) : jb.row_perm_base + row; };
    if (jb.kind == 0) {
        const int64_t n = (int64_t)jb.kchunks * jb.rows * 8;
        if (idx >= n) return;
        const int e = (int)(idx & 7); const int64_t r2 = idx >> 3;
        const int row = (int)(r2 % jb.rows), kc = (int)(r2 / jb.rows);
        const float v = jb.scale * params[jb.src + (int64_t)(kc * 8 + e) * jb.stride_k + (int64_t)rowsrc(row) * jb.stride_row];
        reinterpret_cast<bf16_t*>(slices + jb.dst)[idx] = f16 ? f2n<1>(v) : f2bf(v);
    } else {
        if (idx >= jb.rows) return;
        reinterpret_cast<float*>(slices + jb.dst)[idx] = jb.scale * params[jb.src + rowsrc((int)idx)];
    }
}

// ======================================================================================================================
#ifdef WN_PIPE_SVC_BUILD      // diagnostic build only (csrc/build.py --pipe-svc): the eleven stamp sites cost SGPRs the kernel does not have
#define PIPE_SVC(k) do { if (a.svc && inst == 0 && l == a.svc_l && j == 0 && s == a.svc_s && tid == 0 && t >= a.trace_t0 && t < a.trace_t0 + a.trace_n) a.svc[(size_t)(t - a.trace_t0) * 16 + (k)] = wall_clock64(); } while (0)
#define PIPE_SVC_T(k) do { if (a.svc && inst == 0 && l == a.svc_l && j == 0 && tid == 0 && t >= a.trace_t0 && t < a.trace_t0 + a.trace_n) a.svc[(size_t)(t - a.trace_t0) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define PIPE_SVC(k) do { } while (0)
#define PIPE_SVC_T(k) do { } while (0)
#endif
template <int H, int MULTI, int BP, int SPEC>
__global__ __launch_bounds__(PIPE_THREADS) void wn_synth_pipe_kernel(const PipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // SPEC: the paper model's widths as compile-time constants (R = S = 256, 8 CUs per layer).  A layer CU runs ONE wave per SIMD, so its service time per stream
    // is an instruction count (~5 cycles per issued instruction): with the widths known the slow-path code, the granule loops, most exec masking and the
    // address multiplications disappear from the stream iteration.
    // (SPEC 2: hparams.py's own widths, R = S = 128 with 4 CUs per layer -- the generic code path, minus its loops and multiplications)
    const int R = SPEC == 1 ? 256 : SPEC == 2 ? 128 : a.R, S = SPEC == 1 ? 256 : SPEC == 2 ? 128 : a.S, P = SPEC == 1 ? 8 : SPEC == 2 ? 4 : a.P, T = a.T, C = a.C;
    // ---- role (host-built table: consecutive layers share an XCD -- block b runs on XCD b % 8 --; one instance: spx layers per XCD, the head on XCD 0)
    const int32_t role = a.role_tab[blockIdx.x];
    if (role < 0) return;
    const bool is_head = (role >> 23) & 1;
    const int layer = (role >> 8) & 0xff, j = role & 0xff;                     // (a head: j = its index)
    const int NH = MULTI ? 1 : a.NH;
    // ---- instance (MULTI: several independent pipelines in this launch; else compile-time instance 0 = the whole batch, nothing rebased)
    const int inst = MULTI ? (role >> 24) & 3 : 0;
    const int B = MULTI ? a.iB[inst] : a.B, s0 = MULTI ? a.is0[inst] : 0, Bn = a.B;
    const int64_t moff = MULTI ? (int64_t)(a.L + 1) * s0 * P : 0;                       // mailbox granule sets in front of this instance's
    u32x4* const XM_ = a.XM + moff * PIPE_XG; u32x4* const XML_ = a.XML + moff * PIPE_XG;
    u32x4* const SM_ = a.SM + moff * PIPE_SG; u32x4* const SML_ = a.SML + moff * PIPE_SG;
    bf16_t* const ring_ = a.ring + (MULTI ? a.ring_unit * s0 : 0);                       // a.ring_off[l] is per STREAM: x B for this instance's layer offset
    const bf16_t* const cbt_ = a.cbt + (MULTI ? (int64_t)s0 * T * C : 0);
    const char* const ti_ = a.test_inputs ? (const char*)a.test_inputs + (MULTI ? (int64_t)s0 * T * 4 : 0) : nullptr;
    char* const outs_ = (char*)a.out_samples + (MULTI ? (int64_t)s0 * T * 4 : 0);
    float* const outr_ = a.out_raw ? a.out_raw + (MULTI ? (int64_t)s0 * a.O * T : 0) : nullptr;
    const int32_t* const btab = a.block_tab + (MULTI ? inst * (a.L * P + 1) : 0);
#ifdef WN_PIPE_SVC_BUILD
    const bool trace_on = a.trace && inst == 0;
#else
    constexpr bool trace_on = false;          // (stage trace WN_PIPE_TRACE: diagnostic build only, like the service-time stamps -- its tests sat in every stream iteration)
#endif
    int32_t* const abortf = a.abort_flag;
    // ---- which XCD am I really on?  (block -> XCD = b % 8 is observed, not guaranteed: the table makes the fast path a pure
    // speed choice -- a hand-off whose two ends share an XCD uses the plain-store copy of the mailbox, served by that XCD's L2
    // in about half the round trip of the write-through copy)
    int my_xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(my_xcc)); my_xcc &= 0xf;
    if (threadIdx.x == 0) __hip_atomic_store(a.xcc_tab + blockIdx.x, my_xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    auto block_of = [&](int lay, int jj) { return btab[lay * P + jj]; };     // inverse of the role map
    auto xcc_of = [&](int block) {
        int v = 0, spins = 0;
        while ((v = __hip_atomic_load(a.xcc_tab + block, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > PIPE_SPIN_LIMIT) { pipe_abort(abortf, 50); break; }
        }
        return v - 1;
    };
    auto same_xcc = [&](int block) { return xcc_of(block) == my_xcc; };

    if (!is_head) {
        // ================================================================= layer CU (layer, j)
        const int l = layer;
        {   // weights -> LDS (one coalesced pass)
            const uint4* src = reinterpret_cast<const uint4*>(a.slices + (int64_t)(l * P + j) * a.layer_slice_bytes);
            uint4* dst = reinterpret_cast<uint4*>(lds);
            for (int i = tid; i < a.layer_lds_static / 16; i += PIPE_THREADS) dst[i] = src[i];
        }
        const char* W1c = lds + a.off_w1c; const char* W1p = lds + a.off_w1p; const char* Wo = lds + a.off_wo; const char* Ws = lds + a.off_ws;
        const float* zb = reinterpret_cast<const float*>(lds + a.off_zb); const float* ob = reinterpret_cast<const float*>(lds + a.off_ob);
        char* p = lds + a.layer_lds_static;
        bf16_t* xcur_b = reinterpret_cast<bf16_t*>(p); p += R * 2;
        float* xcur_f = reinterpret_cast<float*>(p); p += R * 4;
        float* psum = reinterpret_cast<float*>(p); p += 4 * R * 4;
        bf16_t* xwave = reinterpret_cast<bf16_t*>(p); p += 4 * R * 2;         // per-wave private copies of x_l(t) (fast path)
        float* zpart = reinterpret_cast<float*>(p); p += 4 * 64 * 4;
        bf16_t* ucur = reinterpret_cast<bf16_t*>(p); p += 64;
        bf16_t* outp = reinterpret_cast<bf16_t*>(p); p += ((R + 7) / 8 * 8 + 8) * 2;
        float* skp = reinterpret_cast<float*>(p); p += (S + 4) * 4;
        bf16_t* vec = reinterpret_cast<bf16_t*>(p); p += (2 * R + C) * 2;
        float* zpast = reinterpret_cast<float*>(p);                       // [B][PIPE_ZS]
        const int d = a.dil[l], mask = a.ring_mask[l];
        const int KP = (2 * R + C) / 8;                                    // k-chunks of the past-tap image
        const int nprod = (l == 0) ? 1 : P;
        bool loc0 = (wave < nprod);
        if (l == 0) { for (int h = 0; h < NH; ++h) loc0 = loc0 && same_xcc(btab[a.L * P + h]); }      // layer 0 reads what the heads publish: the XCD-local copy only if EVERY head shares the XCD
        else loc0 = loc0 && same_xcc(block_of(l - 1, wave));
        const bool loc1 = (wave + 4 < nprod) && same_xcc(block_of(l - 1, wave + 4));
        const bool loc_skip = (l > 0) && same_xcc(block_of(l - 1, j));
        const bool top = (l == a.L - 1);
        // Which copy of a mailbox do MY consumers read?  (A hand-off used to be written twice -- a plain store for consumers on this XCD, a write-through
        // store for the others -- and every consumer picks by the same XCC comparison made here: the copy nobody reads is not written.  A layer's P CUs
        // share an XCD by layout, so this halves the publishing stores; the write-through ones are the slow ones, ~1 us to acknowledge.)
        bool xpub_loc = false, xpub_rem = false, spub_loc = false, spub_rem = false;
        if (!top) {
            bool all = true, any = false;
            for (int jj = 0; jj < P; ++jj) { const bool q = same_xcc(block_of(l + 1, jj)); all = all && q; any = any || q; }
            xpub_loc = any; xpub_rem = !all;
            spub_loc = same_xcc(block_of(l + 1, j)); spub_rem = !spub_loc;
        } else {
            bool all = true, any = false;
            for (int h = 0; h < NH; ++h) { const bool q = same_xcc(btab[a.L * P + h]); all = all && q; any = any || q; }
            spub_loc = any; spub_rem = !all;
        }
        lds_barrier();

        // z_past for (s, tn): taps x(tn-2d), x(tn-d) from this CU's ring (zero before the utterance), conditioning c(s, tn)
        auto precompute = [&](int s, int tn, bool tap1_is_cur) {
            bf16_t* ringb = ring_ + a.ring_off[l] * B + ((int64_t)(j * B + s) * (mask + 1)) * R;
            // global conditioning (wavenet.py:766-777, modules.py:503-508): g is constant over time, so W_g^T g + b_g is a per-STREAM gate
            // bias; it replaces the layer's own bias vector here, off the critical path (the load flies under the matvec below)
            float gb = 0.0f;
            if (a.gbias && tid < 64) gb = a.gbias[((int64_t)l * Bn + s0 + s) * a.G + (tid < 32 ? 32 * j + tid : a.GH + 32 * j + (tid - 32))];
            for (int i = tid; i < KP; i += PIPE_THREADS) {
                uint4 v = make_uint4(0, 0, 0, 0);
                const int k = i * 8;
                if (k < R) { const int tau = tn - 2 * d; if (tau >= 0) v = __builtin_bit_cast(uint4, ld_g16(reinterpret_cast<const u32x4*>(ringb + (int64_t)(tau & mask) * R + k))); }
                else if (k < 2 * R) {
                    const int tau = tn - d;
                    if (tau >= 0) v = tap1_is_cur ? *reinterpret_cast<const uint4*>(xcur_b + (k - R))
                                                  : __builtin_bit_cast(uint4, ld_g16(reinterpret_cast<const u32x4*>(ringb + (int64_t)(tau & mask) * R + (k - R))));
                } else v = cvt8_bf16<H>(*reinterpret_cast<const uint4*>(cbt_ + ((int64_t)s * T + tn) * C + (k - 2 * R)));
                *reinterpret_cast<uint4*>(vec + k) = v;
            }
            lds_barrier();
            {
                const int per = (KP + 3) / 4, kc0 = wave * per, kc1 = min(KP, kc0 + per);
                zpart[wave * 64 + lane] = mv_rows<H>(W1p, 64, lane, reinterpret_cast<const char*>(vec), kc0, kc1);
            }
            lds_barrier();
            if (tid < 64) zpast[s * PIPE_ZS + tid] = zpart[tid] + zpart[64 + tid] + zpart[128 + tid] + zpart[192 + tid] + (a.gbias ? gb : zb[tid]);
            lds_barrier();
        };
        for (int s = 0; s < B; ++s) precompute(s, 0, false);
        // Inside the sample loop the pre-multiplication is split (round 4): its ring reads -- rows written >= d samples ago, HBM latency
        // once the streams' queues exceed L2 -- are REQUESTED by wave 3 right after this CU published its x partial and CONSUMED after the
        // skip chain, so their latency lies under the skip matvec and the wait for the previous layer's running sum instead of in front of
        // the matvec.  Wave 3 on purpose: vmcnt retires in order per WAVE, and waves 0 / 1 poll the skip granules in between (a poll
        // behind an older outstanding load would wait for it: the skip chain is the second latency-critical chain of the ring).
        const int KR = 2 * R / 8;                                            // 16-B chunks of the two past taps (KP <= 128: two chunks per lane of wave 3, wn_pipe_eligible)
        u32x4 pf0 = {0, 0, 0, 0}, pf1 = {0, 0, 0, 0};
        auto pre_issue = [&](int s, int tn, bool tap1_is_cur) {
            if (wave != 3) return;
            // wave-uniform parts as scalars, per-lane parts as 32-bit byte offsets (a ring of one stream is <= 8192 rows x R x 2 B; the conditioning of a
            // run is B x T x C x 2 B < 4 GB): a per-lane 64-bit multiply per candidate address cost this wave 0.35 us per stream
            const char* ringb = (const char*)(ring_ + a.ring_off[l] * B + ((int64_t)(j * B + s) * (mask + 1)) * R);
            const uint32_t row0 = (uint32_t)__builtin_amdgcn_readfirstlane(((tn - 2 * d) & mask) * R * 2), row1 = (uint32_t)__builtin_amdgcn_readfirstlane(((tn - d) & mask) * R * 2);
            const char* cb = (const char*)cbt_ + (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((s * T + tn) * C) * 2;
            const bool ok0 = tn - 2 * d >= 0, ok1 = tn - d >= 0 && !tap1_is_cur;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = lane + 64 * q, k = i * 8;
                u32x4& dst = q ? pf1 : pf0;
                dst = (u32x4){0, 0, 0, 0};
                // ONE load site (an address select, not exec-masked branches writing the same destination registers): ring row of a past tap, or --
                // round 5 -- the conditioning chunk c(s, tn) of the same pre-multiplication, which used to be read INSIDE pre_finish: a dependent
                // global load on the layer CU's service path (profiles/r6i_pipe_svc_trace.txt)
                const bool tap0 = k < R, is_tap = i < KR, is_cond = !is_tap && i < KP;
                const char* src = is_tap ? ringb + (tap0 ? row0 + (uint32_t)k * 2 : row1 + (uint32_t)(k - R) * 2) : cb + (uint32_t)(k - 2 * R) * 2;
                if ((is_tap && (tap0 ? ok0 : ok1)) || is_cond) ld_g16_nowait(dst, reinterpret_cast<const u32x4*>(src));
            }
        };
        auto pre_finish = [&](int s, int tn, bool tap1_is_cur) {
            float gb = 0.0f;
            if (a.gbias && tid < 64) gb = a.gbias[((int64_t)l * Bn + s0 + s) * a.G + (tid < 32 ? 32 * j + tid : a.GH + 32 * j + (tid - 32))];
            if (wave == 3) {
                pf_wait(pf0, pf1);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int i = lane + 64 * q, k = i * 8;
                    if (i < KR) {
                        uint4 v = __builtin_bit_cast(uint4, q ? pf1 : pf0);
                        if (k >= R && tap1_is_cur && tn - d >= 0) v = *reinterpret_cast<const uint4*>(xcur_b + (k - R));
                        *reinterpret_cast<uint4*>(vec + k) = v;
                    } else if (i < KP) *reinterpret_cast<uint4*>(vec + k) = cvt8_bf16<H>(__builtin_bit_cast(uint4, q ? pf1 : pf0));      // the conditioning chunks
                }
            }
            lds_barrier();
            {
                const int per = (KP + 3) / 4, kc0 = wave * per, kc1 = min(KP, kc0 + per);
                zpart[wave * 64 + lane] = mv_rows<H>(W1p, 64, lane, reinterpret_cast<const char*>(vec), kc0, kc1);
            }
            lds_barrier();
            if (tid < 64) zpast[s * PIPE_ZS + tid] = zpart[tid] + zpart[64 + tid] + zpart[128 + tid] + zpart[192 + tid] + (a.gbias ? gb : zb[tid]);
            lds_barrier();
        };

        // ---- fast path (R == 256): the critical-path weights live in REGISTERS for the whole utterance.
        //   z rows: wave w owns gate pairs 8w..8w+7; lane = (pair = lane>>3, k8 = lane&7: a 32-channel eighth of K) holds BOTH rows
        //   of its pair (tanh and sigmoid) over its eighth -> 2 x 4 k-chunks = 32 VGPRs; the 8 partial sums are combined with DPP
        //   (quad xor 1, xor 2, half-row mirror) and the gate is evaluated in the lane that owns the pair: no LDS round trip.
        //   out rows: thread r owns row r of W_out[:, my 32 columns] -> 4 k-chunks = 16 VGPRs.
        const bool fast = (R == 256);
        const bool fast_skip = fast;
        uint4 w1t[4], w1s[4], wor[4];
        const int pr = lane >> 3, k8 = lane & 7;
        const int zrow_t = 8 * wave + pr, zrow_s = 32 + 8 * wave + pr;
        if (fast) {
#pragma unroll
            for (int cix = 0; cix < 4; ++cix) {
                w1t[cix] = *reinterpret_cast<const uint4*>(W1c + ((size_t)(4 * k8 + cix) * 64 + zrow_t) * 16);
                w1s[cix] = *reinterpret_cast<const uint4*>(W1c + ((size_t)(4 * k8 + cix) * 64 + zrow_s) * 16);
            }
#pragma unroll
            for (int cix = 0; cix < 4; ++cix) wor[cix] = *reinterpret_cast<const uint4*>(Wo + ((size_t)cix * R + tid) * 16);
        }
        // ---- batched pre-multiplication (BP, round 5; fast path only).  A layer CU spent 1.3 of its 3.7 us per stream on the pre-multiplication of
        // that stream's next sample (a 64 x (2R + C) matvec behind three workgroup barriers, profiles/r6i_pipe_svc_trace.txt) -- work that does not
        // depend on the message in flight.  Here wave 3 only PARKS the stream's input vector (two ring rows + conditioning) in LDS, and once per
        // sample, after the last stream, the CU multiplies ALL parked vectors at once on the matrix cores: [64 rows x K] x [K x streams], wave w
        // owning rows 16w..16w+15 over the full K (no cross-wave sum), v_mfma_f32_16x16x32 with A straight from the [kc][row][8] weight image.
        // The vectors live where the register-resident tap-2 weights were read from (W1c's 32 KiB image is dead after the copy above).
        // parked vector = KP chunks of 16 B, zero-padded to whole passes of 16 chunks (a partial last pass multiplies weights -- or whatever finite image
        // follows them -- by zeros), at a pitch of 2 (mod 16) chunks: the 16-lane groups of a ds_read_b128 then hit 16 different 16-B slots
        const int KPAD = (KP + 15) / 16 * 16;
        const int VSTR = ((KPAD + 13) / 16 * 16 + 2) * 16;
        // vectors 0 ... VCAP1 - 1 live where the tap-2 image was (64 R x 2 bytes), the rest where W_out's was (32 R x 2 bytes): both are register-resident on this path
        const int VCAP1 = 64 * R * 2 / VSTR;
        auto vrow = [&](int sidx) -> char* { return sidx < VCAP1 ? lds + a.off_w1c + sidx * VSTR : lds + a.off_wo + (sidx - VCAP1) * VSTR; };
        if (BP) {
            lds_barrier();                                                    // every wave holds its W1c registers before the image is overwritten
            for (int i = tid; i < B * (KPAD - KP); i += PIPE_THREADS) *reinterpret_cast<uint4*>(vrow(i / (KPAD - KP)) + (KP + i % (KPAD - KP)) * 16) = make_uint4(0, 0, 0, 0);
        }
        auto pre_stash = [&](int s, int tn, bool tap1_is_cur) {
            if (wave != 3) return;
            pf_wait(pf0, pf1);
            char* vb = vrow(s);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int i = lane + 64 * q, k = i * 8;
                if (i < KR) {
                    uint4 v = __builtin_bit_cast(uint4, q ? pf1 : pf0);
                    if (k >= R && tap1_is_cur && tn - d >= 0) v = *reinterpret_cast<const uint4*>(xcur_b + (k - R));
                    *reinterpret_cast<uint4*>(vb + k * 2) = v;
                } else if (i < KP) *reinterpret_cast<uint4*>(vb + k * 2) = cvt8_bf16<H>(__builtin_bit_cast(uint4, q ? pf1 : pf0));
            }
        };
        auto pre_batch = [&](int t) {
            (void)t;
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            const int n = lane & 15, kq = lane >> 4;
            const bool two = B > 16;
            // per-stream gate bias (global conditioning) or the layer's own: D[m = 4 kq + i][n] is this lane's (row 16 wave + 4 kq + i, stream n / 16 + n)
            float bias0[4], bias1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * wave + 4 * kq + i;
                if (a.gbias) {
                    const int ch = row < 32 ? 32 * j + row : a.GH + 32 * j + (row - 32);
                    bias0[i] = n < B ? a.gbias[((int64_t)l * Bn + s0 + n) * a.G + ch] : 0.0f;
                    bias1[i] = two && 16 + n < B ? a.gbias[((int64_t)l * Bn + s0 + 16 + n) * a.G + ch] : 0.0f;      // (never a load whose value no path consumes: it would stay
                                                                                                                     // pending in the compiler's count and every later wait would drain the queue)
                } else bias0[i] = bias1[i] = zb[row];
            }
            lds_barrier();                                                    // wave 3 parked the last stream's vector
            PIPE_SVC_T(13);
            f32x4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            const char* wa = W1p + (size_t)(16 * wave + n) * 16;
            const char* vb0 = vrow(n); const char* vb1 = vrow(min(16 + n, two ? B - 1 : n));      // (lanes of a column past the last stream re-read a parked vector: their results are not stored)
            // four k-steps (16 chunks) per pass, all LDS reads of a pass issued before its first MFMA, addresses = one pointer per operand + immediates (a
            // k-step at a time with per-k-step address arithmetic and masking was issue-bound: 2.0 us per sample, profiles/r8c_pipe_svc_trace.txt)
            typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
            typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
            auto mm = [&](const uint4& A, const uint4& Bv, f32x4_t c) __attribute__((always_inline)) {
                if constexpr (H) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, A), __builtin_bit_cast(f16x8_t, Bv), c, 0, 0, 0);
                else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, A), __builtin_bit_cast(bf16x8_t, Bv), c, 0, 0, 0);
            };
            const char* pa = wa + kq * 1024; const char* pb0 = vb0 + kq * 16; const char* pb1 = vb1 + kq * 16;
            if (two) {
                for (int kc0 = 0; kc0 < KPAD; kc0 += 16, pa += 16 * 1024, pb0 += 256, pb1 += 256) {
                    uint4 A[4], B0[4], B1[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { A[u] = *reinterpret_cast<const uint4*>(pa + u * 4096); B0[u] = *reinterpret_cast<const uint4*>(pb0 + u * 64); B1[u] = *reinterpret_cast<const uint4*>(pb1 + u * 64); }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { acc0 = mm(A[u], B0[u], acc0); acc1 = mm(A[u], B1[u], acc1); }
                }
            } else {
                for (int kc0 = 0; kc0 < KPAD; kc0 += 16, pa += 16 * 1024, pb0 += 256) {
                    uint4 A[4], B0[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { A[u] = *reinterpret_cast<const uint4*>(pa + u * 4096); B0[u] = *reinterpret_cast<const uint4*>(pb0 + u * 64); }
#pragma unroll
                    for (int u = 0; u < 4; u += 2) { acc0 = mm(A[u], B0[u], acc0); acc1 = mm(A[u + 1], B0[u + 1], acc1); }      // two accumulation chains, summed below
                }
            }
            PIPE_SVC_T(14);
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(bias0[i]), "v"(bias1[i]));      // every bias load is consumed HERE on every path: one left pending in the
                                                                                               // compiler's count (a skipped exec-masked add) turns the next wait -- the top of every stream iteration -- into vmcnt(0)
            {
                const int row0 = 16 * wave + 4 * kq;
                if (!two) acc0 += acc1;
                if (n < B) *reinterpret_cast<float4*>(zpast + n * PIPE_ZS + row0) = make_float4(acc0[0] + bias0[0], acc0[1] + bias0[1], acc0[2] + bias0[2], acc0[3] + bias0[3]);
                if (two && 16 + n < B) *reinterpret_cast<float4*>(zpast + (16 + n) * PIPE_ZS + row0) = make_float4(acc1[0] + bias1[0], acc1[1] + bias1[1], acc1[2] + bias1[2], acc1[3] + bias1[3]);
            }
            lds_barrier();
        };
        const int NXG = R / 4;
        // Early requests (runs of >= a.early_from streams).  Such runs are bound by the service time per stream of the slowest CU, and what a CU polls
        // for was normally published long before it asks: the first-pass x granules of the NEXT stream (right after this stream's x partial left) and
        // this stream's skip granule (before the own skip matvec) are REQUESTED one stage early and only tested where they used to be polled -- a tag
        // that is not there yet falls back to the polling loop.  Shorter runs are latency-bound: an early poll finds nothing and slows somebody's
        // critical hop, so they keep one poll in flight.  (With the per-stream pre-multiplication and ONE head every CU waited for the slowest one
        // and this bought nothing: 72.0 vs 71.7 us per sample at 20 streams, profiles/r6n; it pays once those two are out of the way.)
        const bool many = B >= a.early_from;
        // who owns which granule of the running skip sum (3 fp32 channels each).  Fast path: every wave owns ceil(ng / 4) of them and computes ITS three
        // rows of W_skip[:, mine] u itself -- no exchange through LDS, no barrier in the skip chain (round 5; it was matvec by row -> barrier -> add by
        // granule -> barrier -> publish: 1.3 of a layer CU's 3.3 us per stream, profiles/r8d_pipe_svc_trace.txt).  Else: thread g owns granule g.
        const int ng = (S + 2) / 3, SKW = (ng + 3) / 4;
        const bool sk_en = fast_skip ? (lane < SKW && SKW * wave + lane < ng) : (tid < ng);
        const int sk_g3 = sk_en ? (fast_skip ? SKW * wave + lane : tid) : 0;
        u32x4 xp0 = {0, 0, 0, 0}, xp1 = {0, 0, 0, 0}, skpre = {0, 0, 0, 0}; bool xp_valid = false, skp_valid = false;
        auto x_request = [&](int s, int t) {
            xp_valid = false; skp_valid = false;
            const bool wrap = s + 1 >= B;
            if (!many || (wrap && t + 1 >= T)) return;
            const int sn = wrap ? 0 : s + 1;
            if (wave < nprod) {
                const __amdgpu_buffer_rsrc_t rn0 = poll_rsrc((loc0 ? XML_ : XM_) + ((int64_t)(l * B + sn) * P) * PIPE_XG, P * PIPE_XG * 16);
                const __amdgpu_buffer_rsrc_t rn1 = poll_rsrc((loc1 ? XML_ : XM_) + ((int64_t)(l * B + sn) * P) * PIPE_XG, P * PIPE_XG * 16);
                const bool two = wave + 4 < nprod;
                const int o0 = (wave * PIPE_XG + lane) * 16, o1 = two ? ((wave + 4) * PIPE_XG + lane) * 16 : o0;
                xp0 = poll_ld(rn0, o0); xp1 = poll_ld(two ? rn1 : rn0, o1);
                xp_valid = true;
            }
            // the NEXT stream's skip granule too (not this stream's a stage early: a load issued behind this stream's write-through x store returns
            // behind that store's acknowledgement -- vmcnt retires in order --, ~1 us on the CUs whose consumers sit on another XCD)
            if (l > 0) {
                const __amdgpu_buffer_rsrc_t rsn = poll_rsrc((loc_skip ? SML_ : SM_) + ((int64_t)(l * B + sn) * P + j) * PIPE_SG, PIPE_SG * 16);
                if (sk_en) skpre = poll_ld(rsn, sk_g3 * 16);
                skp_valid = true;
            }
        };
        for (int t = 0; t < T; ++t) {
            const uint32_t want = (uint32_t)(t + 1);
            for (int s = 0; s < B; ++s) {
                PIPE_SVC(0);
                const bool xp_have = xp_valid, skp_have = skp_valid;       // a request is good for exactly ONE iteration
                xp_valid = false; skp_valid = false;
                const u32x4 skcur = skpre;                                 // (this iteration's x_request overwrites skpre with the NEXT stream's granule before step 4 reads this one)
                // ---- 1. x_l(t) = sum of the partial vectors published by the previous stage (granule g = channels 4g..4g+3)
                {
                    const __amdgpu_buffer_rsrc_t rs0 = poll_rsrc((loc0 ? XML_ : XM_) + ((int64_t)(l * B + s) * P) * PIPE_XG, P * PIPE_XG * 16);
                    const __amdgpu_buffer_rsrc_t rs1 = poll_rsrc((loc1 ? XML_ : XM_) + ((int64_t)(l * B + s) * P) * PIPE_XG, P * PIPE_XG * 16);
                    for (int g = lane; g < NXG; g += 64) {
                        float v[4] = {0, 0, 0, 0};
                        const int p0 = wave, p1 = wave + 4;
                        if (p0 < nprod) {
                            const bool two = p1 < nprod;
                            const int o0 = (p0 * PIPE_XG + g) * 16, o1 = two ? (p1 * PIPE_XG + g) * 16 : o0;
                            u32x4 g0 = xp0, g1 = xp1;
                            bool have = false;
                            if (xp_have && g == lane) have = __all(g0.w == want && g1.w == want) != 0;      // requested during the previous stream's iteration
                            int spins = 0;
                            while (!have) {     // ONE poll in flight: more concurrent polls measurably slow every hop (fabric contention)
                                g0 = poll_ld(rs0, o0); g1 = poll_ld(two ? rs1 : rs0, o1);
                                if (__all(g0.w == want && g1.w == want)) break;
                                if (++spins > PIPE_SPIN_LIMIT) { pipe_abort(abortf, 100 + l); break; }
                                if ((spins & 255) == 0 && pipe_aborted(abortf)) break;
                            }
                            if (!two) g1 = (u32x4){0, 0, 0, 0};
                            v[0] = n2f<H>((bf16_t)(g0.x & 0xffff)) + n2f<H>((bf16_t)(g1.x & 0xffff)); v[1] = n2f<H>((bf16_t)(g0.x >> 16)) + n2f<H>((bf16_t)(g1.x >> 16));
                            v[2] = n2f<H>((bf16_t)(g0.y & 0xffff)) + n2f<H>((bf16_t)(g1.y & 0xffff)); v[3] = n2f<H>((bf16_t)(g0.y >> 16)) + n2f<H>((bf16_t)(g1.y >> 16));
                        }
                        *reinterpret_cast<float4*>(psum + wave * R + g * 4) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
                PIPE_SVC(1);
                lds_barrier();                                                                   // (A) the 4 waves' partial sums
                PIPE_SVC(2);
                if (trace_on && s == 0 && j == 0 && tid == 0 && t >= a.trace_t0 && t < a.trace_t0 + a.trace_n) a.trace[(size_t)(t - a.trace_t0) * 2 * (a.L + 2) + 2 * l] = wall_clock64();
                if (fast) {
                    // every wave rebuilds the full bf16 x_l(t) for itself (no second barrier): lane g -> channels 4g..4g+3
                    bf16_t* myx = xwave + wave * R;
                    {
                        const float4 q0 = *reinterpret_cast<const float4*>(psum + lane * 4), q1 = *reinterpret_cast<const float4*>(psum + R + lane * 4);
                        const float4 q2 = *reinterpret_cast<const float4*>(psum + 2 * R + lane * 4), q3 = *reinterpret_cast<const float4*>(psum + 3 * R + lane * 4);
                        const uint2 pk = make_uint2(pack_n2<H>(q0.x + q1.x + q2.x + q3.x, q0.y + q1.y + q2.y + q3.y), pack_n2<H>(q0.z + q1.z + q2.z + q3.z, q0.w + q1.w + q2.w + q3.w));
                        *reinterpret_cast<uint2*>(myx + lane * 4) = pk;
                        if (wave == 0) *reinterpret_cast<uint2*>(xcur_b + lane * 4) = pk;               // the copy the off-critical-path steps read
                    }
                    // ---- 2. z = z_past + W_tap2 x ; gate (modules.py:494-510), all inside the wave
                    float zt = 0.0f, zs = 0.0f;
#pragma unroll
                    for (int cix = 0; cix < 4; ++cix) {
                        const uint4 xv = *reinterpret_cast<const uint4*>(myx + (4 * k8 + cix) * 8);
                        zt = dot8n<H>(w1t[cix], xv, zt); zs = dot8n<H>(w1s[cix], xv, zs);
                    }
                    zt += dpp_f<DPP_XOR1>(zt); zs += dpp_f<DPP_XOR1>(zs);
                    zt += dpp_f<DPP_XOR2>(zt); zs += dpp_f<DPP_XOR2>(zs);
                    zt += dpp_f<DPP_HALF_MIRROR>(zt); zs += dpp_f<DPP_HALF_MIRROR>(zs);
                    if (k8 == 0) {
                        zt += zpast[s * PIPE_ZS + zrow_t]; zs += zpast[s * PIPE_ZS + zrow_s];
                        const float e = __expf(2.0f * zt);
                        ucur[8 * wave + pr] = f2n<H>((1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f)) * __builtin_amdgcn_rcpf(1.0f + __expf(-zs)));
                    }
                    lds_barrier();                                                                 // (B) the 32 gate outputs of this CU
                    PIPE_SVC(3);
                    // ---- 3. partial of x_{l+1}(t) = rho (W_out[:, mine] u_mine [+ x + b on CU 0]) -> granules (modules.py:512-521)
                    // (the top layer multiplies too and publishes nothing: both copies disabled -- the same instruction stream for every CU)
                    {
                        float o = 0.0f, o2 = 0.0f;
                        o = dot8n<H>(wor[0], *reinterpret_cast<const uint4*>(ucur), o); o2 = dot8n<H>(wor[1], *reinterpret_cast<const uint4*>(ucur + 8), o2);
                        o = dot8n<H>(wor[2], *reinterpret_cast<const uint4*>(ucur + 16), o); o2 = dot8n<H>(wor[3], *reinterpret_cast<const uint4*>(ucur + 24), o2);
                        o += o2;
                        if (j == 0) o += n2f<H>(myx[tid]) + ob[tid];
                        const uint32_t me = f2n<H>(o * a.rho);
                        const uint32_t n1 = dpp_u<DPP_QUAD_BCAST(1)>(me), n2 = dpp_u<DPP_QUAD_BCAST(2)>(me), n3 = dpp_u<DPP_QUAD_BCAST(3)>(me);
                        if (trace_on && s == 0 && j == 0 && tid == 0 && t >= a.trace_t0 && t < a.trace_t0 + a.trace_n) a.trace[(size_t)(t - a.trace_t0) * 2 * (a.L + 2) + 2 * l + 1] = wall_clock64();
                        const u32x4 g = {me | (n1 << 16), n2 | (n3 << 16), 0, want};
                        const int64_t gb = ((int64_t)((l + 1) * B + s) * P + j) * PIPE_XG;              // (layer L's slots exist and nobody reads them)
                        const int off = (lane & 3) == 0 ? (tid >> 2) * 16 : -1;
                        st_buf16(poll_rsrc(XML_ + gb, PIPE_XG * 16), xpub_loc ? off : -1, g);
                        st_buf16_wt(poll_rsrc(XM_ + gb, PIPE_XG * 16), xpub_rem ? off : -1, g);
                    }
                    if (tid < R) xcur_f[tid] = n2f<H>(myx[tid]);
                } else {
                    for (int r = tid; r < R; r += PIPE_THREADS) {
                        const bf16_t xb = f2n<H>(psum[r] + psum[R + r] + psum[2 * R + r] + psum[3 * R + r]);
                        xcur_b[r] = xb; xcur_f[r] = n2f<H>(xb);
                    }
                    lds_barrier();
                    // ---- 2. z = z_past + W_tap2 x ; gate (modules.py:494-510)
                    {
                        const int KC = R / 8, per = (KC + 3) / 4, kc0 = wave * per, kc1 = min(KC, kc0 + per);
                        zpart[wave * 64 + lane] = mv_rows<H>(W1c, 64, lane, reinterpret_cast<const char*>(xcur_b), kc0, kc1);
                    }
                    lds_barrier();
                    if (tid < 32) {
                        const float za = zpart[tid] + zpart[64 + tid] + zpart[128 + tid] + zpart[192 + tid] + zpast[s * PIPE_ZS + tid];
                        const float zs = zpart[32 + tid] + zpart[96 + tid] + zpart[160 + tid] + zpart[224 + tid] + zpast[s * PIPE_ZS + 32 + tid];
                        const float e = __expf(2.0f * za);
                        ucur[tid] = f2n<H>((1.0f - 2.0f / (e + 1.0f)) * (1.0f / (1.0f + __expf(-zs))));
                    }
                    lds_barrier();
                    // ---- 3. partial of x_{l+1}(t) = rho (W_out[:, mine] u_mine [+ x + b on CU 0])  -> granules (modules.py:512-521)
                    if (!top) {
                        for (int r = tid; r < R; r += PIPE_THREADS) {
                            float o = mv_rows<H>(Wo, R, r, reinterpret_cast<const char*>(ucur), 0, 4);
                            if (j == 0) o += xcur_f[r] + ob[r];
                            outp[r] = f2n<H>(o * a.rho);
                        }
                        lds_barrier();
                        for (int g = tid; g < NXG; g += PIPE_THREADS) {
                            const uint2 q = *reinterpret_cast<const uint2*>(outp + g * 4);
                            u32x4 gr = {q.x, q.y, 0, want};
                            const int64_t gi = ((int64_t)((l + 1) * B + s) * P + j) * PIPE_XG + g;
                            if (xpub_loc) st_g16_local(XML_ + gi, gr);
                            if (xpub_rem) st_g16(XM_ + gi, gr);
                        }
                    }
                }
                x_request(s, t);
                PIPE_SVC(4);
                if (!fast) lds_barrier();            // (the granules of the slow path pass through outp; the fast path shares nothing here: ucur is rewritten behind barrier A)
                PIPE_SVC(5);
                if (trace_on && (!fast || top) && s == 0 && j == 0 && tid == 0 && t >= a.trace_t0 && t < a.trace_t0 + a.trace_n) a.trace[(size_t)(t - a.trace_t0) * 2 * (a.L + 2) + 2 * l + 1] = wall_clock64();
                if (t + 1 < T) pre_issue(s, t + 1, d == 1);          // wave 3: ring reads of this stream's NEXT sample, consumed in step 5
                // ---- 4. skip chain: running sum of CU (l-1, j) + W_skip[:, mine] u_mine  -> CU (l+1, j) / head (wavenet.py:833-836)
                {
                    // own contribution first (the running sum of CU (l-1, j) is published ~1 us after its x partial: no point in
                    // polling early, and every useless poll slows somebody's critical hop)
                    // (-- except in runs of many streams, see `many`: requested during the previous stream's iteration)
                    const __amdgpu_buffer_rsrc_t rs = poll_rsrc((loc_skip ? SML_ : SM_) + ((int64_t)(l * B + s) * P + j) * PIPE_SG, PIPE_SG * 16);
                    const int64_t gb = ((int64_t)((l + 1) * B + s) * P + j) * PIPE_SG;
                    if (fast_skip) {
                        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
                        {
                            const int r0 = min(3 * sk_g3, S - 1), r1 = min(3 * sk_g3 + 1, S - 1), r2 = min(3 * sk_g3 + 2, S - 1);
#pragma unroll
                            for (int kc = 0; kc < 4; ++kc) {
                                const uint4 u = *reinterpret_cast<const uint4*>(ucur + kc * 8);
                                a0 = dot8n<H>(*reinterpret_cast<const uint4*>(Ws + ((size_t)kc * S + r0) * 16), u, a0);
                                a1 = dot8n<H>(*reinterpret_cast<const uint4*>(Ws + ((size_t)kc * S + r1) * 16), u, a1);
                                a2 = dot8n<H>(*reinterpret_cast<const uint4*>(Ws + ((size_t)kc * S + r2) * 16), u, a2);
                            }
                            if (3 * sk_g3 + 1 >= S) a1 = 0.0f;
                            if (3 * sk_g3 + 2 >= S) a2 = 0.0f;
                        }
                        PIPE_SVC(6);
                        if (l > 0 && sk_en) {
                            u32x4 g = skcur;
                            int spins = 0;
                            while (!(skp_have && g.w == want)) {
                                g = poll_ld(rs, sk_g3 * 16);
                                if (g.w == want) break;
                                __builtin_amdgcn_s_sleep(2);
                                if (++spins > PIPE_SPIN_LIMIT) { pipe_abort(abortf, 200 + l); break; }
                                if ((spins & 255) == 0 && pipe_aborted(abortf)) break;
                            }
                            a0 += __uint_as_float(g.x); a1 += __uint_as_float(g.y); a2 += __uint_as_float(g.z);
                        }
                        PIPE_SVC(7);
                        const u32x4 g = {__float_as_uint(a0), __float_as_uint(a1), __float_as_uint(a2), want};
                        st_buf16(poll_rsrc(SML_ + gb, PIPE_SG * 16), sk_en && spub_loc ? sk_g3 * 16 : -1, g);
                        st_buf16_wt(poll_rsrc(SM_ + gb, PIPE_SG * 16), sk_en && spub_rem ? sk_g3 * 16 : -1, g);
                    } else {
                        for (int r = tid; r < S; r += PIPE_THREADS) skp[r] = mv_rows<H>(Ws, S, r, reinterpret_cast<const char*>(ucur), 0, 4);
                        if (tid < 4) skp[S + tid] = 0.0f;
                        lds_barrier();
                        PIPE_SVC(6);
                        if (l > 0 && sk_en) {       // one poll per GRANULE (3 channels), by the first (S+2)/3 threads
                            u32x4 g = skcur;
                            int spins = 0;
                            while (!(skp_have && g.w == want)) {
                                g = poll_ld(rs, sk_g3 * 16);
                                if (g.w == want) break;
                                __builtin_amdgcn_s_sleep(2);
                                if (++spins > PIPE_SPIN_LIMIT) { pipe_abort(abortf, 200 + l); break; }
                                if ((spins & 255) == 0 && pipe_aborted(abortf)) break;
                            }
                            skp[sk_g3 * 3] += __uint_as_float(g.x); skp[sk_g3 * 3 + 1] += __uint_as_float(g.y); skp[sk_g3 * 3 + 2] += __uint_as_float(g.z);
                        }
                        lds_barrier();
                        PIPE_SVC(7);
                        const u32x4 g = {__float_as_uint(skp[sk_g3 * 3]), __float_as_uint(skp[sk_g3 * 3 + 1]), __float_as_uint(skp[sk_g3 * 3 + 2]), want};
                        st_buf16(poll_rsrc(SML_ + gb, PIPE_SG * 16), sk_en && spub_loc ? sk_g3 * 16 : -1, g);
                        st_buf16_wt(poll_rsrc(SM_ + gb, PIPE_SG * 16), sk_en && spub_rem ? sk_g3 * 16 : -1, g);
                    }
                }
                PIPE_SVC(8);
                // ---- 5. queue update (private ring) and the pre-multiplication for this stream's next step
                {
                    bf16_t* ringb = ring_ + a.ring_off[l] * B + ((int64_t)(j * B + s) * (mask + 1)) * R;
                    {   // (R / 8 <= 48 chunks: the first lanes of wave 0; the other waves issue the same store disabled)
                        const bool en = tid < R / 8;
                        st_buf16(poll_rsrc(ringb + (int64_t)(t & mask) * R, R * 2), en ? tid * 16 : -1, __builtin_bit_cast(u32x4, *reinterpret_cast<const uint4*>(xcur_b + (en ? tid * 8 : 0))));
                    }
                    PIPE_SVC(9);
                    if constexpr (H) {      // half has 5 exponent bits: a residual stream beyond 65504 became inf in the hand-off -- report it, do not synthesise garbage
                        for (int r = tid; r < R; r += PIPE_THREADS)      // (R <= 384 on the generic path: more channels than threads)
                            if (!(fabsf(xcur_f[r]) <= 65504.0f)) pipe_abort(abortf, 400 + l);
                    }
                    if (t + 1 < T) { if (BP) pre_stash(s, t + 1, d == 1); else pre_finish(s, t + 1, d == 1); }      // ring rows are read past this CU's L1 (sc1): slots are recycled
                }
                PIPE_SVC(10);
                if (a.abort_every && pipe_aborted(abortf)) return;
            }
            // (once per sample, not per stream unless a.abort_every (WN_PIPE_ABORT_EVERY=1, A/B): the flag is a device-scope load whose wait also drains the early requests above; every spin loop tests it
            // on its own every 256 polls)
            if (pipe_aborted(abortf)) return;
            PIPE_SVC_T(11);
            if (BP && t + 1 < T) pre_batch(t);                       // z_past of every stream's next sample in one matrix product
            PIPE_SVC_T(12);
        }
        return;
    }

    // ===================================================================== head CU
    {
        {
            const uint4* src = reinterpret_cast<const uint4*>(a.slices + a.head_slice_off);
            uint4* dst = reinterpret_cast<uint4*>(lds);
            for (int i = tid; i < a.head_lds_static / 16; i += PIPE_THREADS) dst[i] = src[i];
        }
        const char* Wh1 = lds + a.hoff_wh1; const char* Wh2 = lds + a.hoff_wh2;
        const float* b1 = reinterpret_cast<const float*>(lds + a.hoff_b1); const float* b2 = reinterpret_cast<const float*>(lds + a.hoff_b2);
        const float* sb = reinterpret_cast<const float*>(lds + a.hoff_sb);
        const float* win = reinterpret_cast<const float*>(lds + a.hoff_win); const float* bin = reinterpret_cast<const float*>(lds + a.hoff_bin);
        char* p = lds + a.head_lds_static;
        float* psum = reinterpret_cast<float*>(p); p += 4 * (S + 4) * 4;
        bf16_t* r1 = reinterpret_cast<bf16_t*>(p); p += S * 2;
        bf16_t* h2 = reinterpret_cast<bf16_t*>(p); p += S * 2;
        float* yraw = reinterpret_cast<float*>(p); p += a.OP * 4;
        bf16_t* outp = reinterpret_cast<bf16_t*>(p); p += ((R + 7) / 8 * 8 + 8) * 2;
        float* nxt_f = reinterpret_cast<float*>(p); p += 16;
        int* nxt_i = reinterpret_cast<int*>(p); p += 16;
        const int L = a.L, O = a.O, OP = a.OP, mode = a.mode;
        const bool hloc0 = (wave < P) && same_xcc(block_of(L - 1, wave)), hloc1 = (wave + 4 < P) && same_xcc(block_of(L - 1, wave + 4));
        // which copy of layer 0's mailbox is read?  CU (0, jj) takes the XCD-local one iff EVERY head shares its XCD (loc0 there): the same test, made here
        bool ipub_loc = false, ipub_rem = false;
        for (int jj = 0; jj < P; ++jj) {
            const int cx = xcc_of(block_of(0, jj));
            bool l0 = true;
            for (int h = 0; h < NH; ++h) l0 = l0 && xcc_of(btab[L * P + h]) == cx;
            ipub_loc = ipub_loc || l0; ipub_rem = ipub_rem || !l0;
        }
        lds_barrier();
        // fast head (S == 256, O <= 32): both head convolutions multiply from REGISTERS (128 + 16 VGPRs of weights per thread)
        const bool hfast = (S == 256 && OP == 32);
        uint4 wh1r[32], wh2r[4];
        if (hfast) {
#pragma unroll
            for (int kc = 0; kc < 32; ++kc) wh1r[kc] = *reinterpret_cast<const uint4*>(Wh1 + ((size_t)kc * 256 + tid) * 16);
#pragma unroll
            for (int cix = 0; cix < 4; ++cix) wh2r[cix] = *reinterpret_cast<const uint4*>(Wh2 + ((size_t)((tid & 7) * 4 + cix) * 32 + (tid >> 3)) * 16);
        }

        // x_0(tn) = input convolution of `value` (wavenet.py:826 / 433-445)  -> layer 0's mailbox
        auto publish_input = [&](int s, int tn) {
            for (int r = tid; r < R; r += PIPE_THREADS) {
                float v;
                if (mode == 2) v = a.win_global[(int64_t)nxt_i[0] * R + r] + a.bin_global[r];
                else v = win[r] * nxt_f[0] + bin[r];
                outp[r] = f2n<H>(v);
            }
            lds_barrier();
            for (int g = tid; g < R / 4; g += PIPE_THREADS) {
                const uint2 q = *reinterpret_cast<const uint2*>(outp + g * 4);
                u32x4 gr = {q.x, q.y, 0, (uint32_t)(tn + 1)};
                const int64_t gi = ((int64_t)(0 * B + s) * P + 0) * PIPE_XG + g;
                if (ipub_loc) st_g16_local(XML_ + gi, gr);
                if (ipub_rem) st_g16(XM_ + gi, gr);
            }
            lds_barrier();
        };
        if (tid == 0) { nxt_f[0] = 0.0f; nxt_i[0] = a.start_id; }
        lds_barrier();
        for (int s = j; s < B; s += NH) publish_input(s, 0);          // head j of NH: the streams s = j (mod NH)

        for (int t = 0; t < T; ++t) {
            const uint32_t want = (uint32_t)(t + 1);
            for (int s = j; s < B; s += NH) {
                // (prefetch, off the critical path) this step's noise and teacher-forcing value
                float nz_pre = 0.0f, ti_pre = 0.0f;       // nz_pre: lane i < M: -log(-log u1_i) (Gumbel); lane M: log u2 - log(1 - u2) (logistic)
                if (mode == 0 && wave == 0) {
                    const int M = O / 3;
                    if (lane < a.nps) { const float uu = a.noise[((int64_t)t * Bn + s0 + s) * a.nps + lane]; nz_pre = (lane < M) ? -logf(-logf(uu)) : logf(uu) - logf(1.0f - uu); }
                    if (ti_) ti_pre = ((const float*)ti_)[(int64_t)s * T + t];
                }
                // ---- total skip = sum of the P running sums that left the top layer (+ all skip biases), ReLU (wavenet.py:840)
                {
                    const __amdgpu_buffer_rsrc_t rsa = poll_rsrc((hloc0 ? SML_ : SM_) + ((int64_t)(L * B + s) * P) * PIPE_SG, P * PIPE_SG * 16);
                    const __amdgpu_buffer_rsrc_t rsb = poll_rsrc((hloc1 ? SML_ : SM_) + ((int64_t)(L * B + s) * P) * PIPE_SG, P * PIPE_SG * 16);
                    const int ng = (S + 2) / 3;
                    float v[2][3] = {{0, 0, 0}, {0, 0, 0}};
                    const bool h0 = lane < ng, h1 = lane + 64 < ng;
                    const int pa = wave, pb = wave + 4;
                    if (pa < P) {
                        // a lane without a granule / a missing second producer re-reads a valid slot (the tag test stays uniform)
                        const int oa0 = (pa * PIPE_SG + (h0 ? lane : 0)) * 16, oa1 = (pa * PIPE_SG + (h1 ? lane + 64 : 0)) * 16;
                        const int qb = (pb < P) ? pb : pa;
                        const int ob0 = (qb * PIPE_SG + (h0 ? lane : 0)) * 16, ob1 = (qb * PIPE_SG + (h1 ? lane + 64 : 0)) * 16;
                        struct Q { u32x4 a0, a1, b0, b1; };
                        auto issue = [&]() { Q q; q.a0 = poll_ld(rsa, oa0); q.a1 = poll_ld(rsa, oa1); q.b0 = poll_ld(pb < P ? rsb : rsa, ob0); q.b1 = poll_ld(pb < P ? rsb : rsa, ob1); return q; };
                        auto good = [&](const Q& q) { return __all(q.a0.w == want && q.a1.w == want && q.b0.w == want && q.b1.w == want) != 0; };
                        Q g;
                        int spins = 0;
                        for (;;) {
                            g = issue();
                            if (good(g)) break;
                            if (++spins > PIPE_SPIN_LIMIT) { pipe_abort(abortf, 300); break; }
                            if ((spins & 255) == 0 && pipe_aborted(abortf)) break;
                        }
                        const float fb = (pb < P) ? 1.0f : 0.0f;
                        if (h0) { v[0][0] = __uint_as_float(g.a0.x) + fb * __uint_as_float(g.b0.x); v[0][1] = __uint_as_float(g.a0.y) + fb * __uint_as_float(g.b0.y); v[0][2] = __uint_as_float(g.a0.z) + fb * __uint_as_float(g.b0.z); }
                        if (h1) { v[1][0] = __uint_as_float(g.a1.x) + fb * __uint_as_float(g.b1.x); v[1][1] = __uint_as_float(g.a1.y) + fb * __uint_as_float(g.b1.y); v[1][2] = __uint_as_float(g.a1.z) + fb * __uint_as_float(g.b1.z); }
                    }
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                        for (int e = 0; e < 3; ++e) { const int ch = (lane + 64 * hh) * 3 + e; if (ch < S) psum[wave * (S + 4) + ch] = v[hh][e]; }
                }
                lds_barrier();
                if (trace_on && s == 0 && tid == 0 && t >= a.trace_t0 && t < a.trace_t0 + a.trace_n) a.trace[(size_t)(t - a.trace_t0) * 2 * (a.L + 2) + 2 * L] = wall_clock64();
                for (int r = tid; r < S; r += PIPE_THREADS) {
                    float tot = sb[r];
                    for (int w = 0; w < 4 && w < P; ++w) tot += psum[w * (S + 4) + r];
                    r1[r] = f2n<H>(fmaxf(tot, 0.0f));
                }
                lds_barrier();
                // ---- head convolutions (wavenet.py:840-844)
                if (hfast) {
                    float h0 = 0.0f, h1 = 0.0f, h2a = 0.0f, h3 = 0.0f;
#pragma unroll
                    for (int kc = 0; kc < 32; kc += 4) {
                        h0 = dot8n<H>(wh1r[kc], *reinterpret_cast<const uint4*>(r1 + kc * 8), h0); h1 = dot8n<H>(wh1r[kc + 1], *reinterpret_cast<const uint4*>(r1 + (kc + 1) * 8), h1);
                        h2a = dot8n<H>(wh1r[kc + 2], *reinterpret_cast<const uint4*>(r1 + (kc + 2) * 8), h2a); h3 = dot8n<H>(wh1r[kc + 3], *reinterpret_cast<const uint4*>(r1 + (kc + 3) * 8), h3);
                    }
                    h2[tid] = f2n<H>(fmaxf((h0 + h1) + (h2a + h3) + b1[tid], 0.0f));
                } else {
                    for (int r = tid; r < S; r += PIPE_THREADS) h2[r] = f2n<H>(fmaxf(mv_rows<H>(Wh1, S, r, reinterpret_cast<const char*>(r1), 0, S / 8) + b1[r], 0.0f));
                }
                lds_barrier();
                if (hfast) {      // thread = (row = tid >> 3, k-eighth = tid & 7): 4 chunks each, summed with DPP
                    const int row = tid >> 3, ke = tid & 7;
                    float y = 0.0f;
#pragma unroll
                    for (int cix = 0; cix < 4; ++cix) y = dot8n<H>(wh2r[cix], *reinterpret_cast<const uint4*>(h2 + (ke * 4 + cix) * 8), y);
                    y += dpp_f<DPP_XOR1>(y); y += dpp_f<DPP_XOR2>(y); y += dpp_f<DPP_HALF_MIRROR>(y);
                    if (ke == 0) yraw[row] = (row < O) ? y + b2[row] : 0.0f;
                } else {
                    for (int r = tid; r < OP; r += PIPE_THREADS) yraw[r] = (r < O) ? mv_rows<H>(Wh2, OP, r, reinterpret_cast<const char*>(h2), 0, S / 8) + b2[r] : 0.0f;
                }
                lds_barrier();
                if (trace_on && s == 0 && tid == 0 && t >= a.trace_t0 && t < a.trace_t0 + a.trace_n) a.trace[(size_t)(t - a.trace_t0) * 2 * (a.L + 2) + 2 * L + 1] = wall_clock64();
                // ---- sample (wavenet.py:847-878); noise [T][B][nps] was prefetched while the message was in the stack
                if (mode == 0 && O / 3 <= 16) {
                    if (wave == 0) {         // mixture.py:76-107: Gumbel-max over the mixture logits (first maximum wins), then the logistic
                        const int M = O / 3;
                        float v = (lane < M) ? yraw[lane] + nz_pre : -INFINITY; int bi = lane;
#define ARGMAX_STEP(CTRL) { const float ov = dpp_f<CTRL>(v); const int oi = (int)dpp_u<CTRL>((uint32_t)bi); if (ov > v || (ov == v && oi < bi)) { v = ov; bi = oi; } }
                        ARGMAX_STEP(DPP_XOR1) ARGMAX_STEP(DPP_XOR2) ARGMAX_STEP(DPP_HALF_MIRROR) ARGMAX_STEP(0x140 /* row_mirror: the other half of the 16 */)
#undef ARGMAX_STEP
                        const float lgt = __shfl(nz_pre, M);
                        if (lane == 0) {
                            const float ls = fmaxf(yraw[2 * M + bi], a.lsmin);
                            float x = yraw[M + bi] + expf(ls) * lgt;
                            x = fminf(fmaxf(x, -1.0f), 1.0f);
                            ((float*)outs_)[(int64_t)s * T + t] = x;
                            nxt_f[0] = ti_ ? ti_pre : x;
                        }
                    }
                } else if (tid == 0) {
                    const float* nz = a.noise + ((int64_t)t * Bn + s0 + s) * a.nps;
                    if (mode == 2) {
                        float best = -INFINITY; int bi = 0;
                        for (int q = 0; q < O; ++q) { const float vv = yraw[q] - logf(-logf(nz[q])); if (vv > best) { best = vv; bi = q; } }
                        ((int32_t*)outs_)[(int64_t)s * T + t] = bi;
                        nxt_i[0] = ti_ ? ((const int32_t*)ti_)[(int64_t)s * T + t] : bi;
                    } else {
                        float x;
                        if (mode == 0) {
                            const int M = O / 3;
                            float best = -INFINITY; int bi = 0;
                            for (int i = 0; i < M; ++i) { const float vv = yraw[i] - logf(-logf(nz[i])); if (vv > best) { best = vv; bi = i; } }
                            const float ls = fmaxf(yraw[2 * M + bi], a.lsmin);
                            const float u = nz[M];
                            x = yraw[M + bi] + expf(ls) * (logf(u) - logf(1.0f - u));
                        } else x = yraw[0] + expf(fmaxf(yraw[1], a.lsmin)) * nz[0];
                        x = fminf(fmaxf(x, -1.0f), 1.0f);
                        ((float*)outs_)[(int64_t)s * T + t] = x;
                        nxt_f[0] = ti_ ? ((const float*)ti_)[(int64_t)s * T + t] : x;
                    }
                }
                if (outr_) for (int o = tid; o < O; o += PIPE_THREADS) outr_[((int64_t)s * O + o) * T + t] = yraw[o];
                lds_barrier();
                if (trace_on && s == 0 && tid == 0 && t >= a.trace_t0 && t < a.trace_t0 + a.trace_n) a.trace[(size_t)(t - a.trace_t0) * 2 * (a.L + 2) + 2 * L + 2] = wall_clock64();
                if (t + 1 < T) publish_input(s, t + 1);
                if (trace_on && s == 0 && tid == 0 && t >= a.trace_t0 && t < a.trace_t0 + a.trace_n) a.trace[(size_t)(t - a.trace_t0) * 2 * (a.L + 2) + 2 * L + 3] = wall_clock64();
            }
            if (pipe_aborted(abortf)) return;
        }
    }
}

// ======================================================================================================================
struct Pipe {
    int B = 0, T = 0, P = 0, spx = 0, grid = 0;
    char* slices = nullptr; int64_t layer_slice_bytes = 0, head_slice_off = 0, slices_bytes = 0;
    SliceJob* jobs_dev = nullptr; int* job_block0_dev = nullptr; int njobs = 0, nblocks = 0;
    u32x4* XM = nullptr; u32x4* SM = nullptr; u32x4* XML = nullptr; u32x4* SML = nullptr; size_t xm_bytes = 0, sm_bytes = 0;
    bf16_t* ring = nullptr; size_t ring_bytes = 0; int ring_B = 0;
    int32_t* abort_dev = nullptr;       // [0] flag of the running launch, [1] sticky OR of every run since the last wn_pipe_check, +256 B: XCC table
    int32_t* abort_host = nullptr;      // pinned: the abort flag of the last run lands here asynchronously (read by wn_pipe_check)
    bool pending = false;               // a run has been enqueued whose flag has not been inspected yet
    int test_aborts = 0;
    int layer_lds = 0, head_lds = 0;
    bool f16 = false;                   // 16-bit storage type of weights / hand-offs / queues: IEEE half instead of bf16 (wn_ctx::pipe_f16 at the run)
    PipeArgs proto;
    hipStream_t priv = nullptr; hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // several pipeline INSTANCES side by side (round 5): a model whose L * P + 1 CUs fit the chip more than once (hparams.py's defaults: 81 CUs, three
    // times by CU count, twice with the slack the dispatcher needs) serves a batch of more than 8 streams as NI independent runs of <= 8 where they fit -- the regime in which a run costs the wall time of ONE stream
#define PIPE_MAX_INST 3
    int ni_max = 1;                                    // how many instances the chip holds (layout below)
    int32_t* tabs_dev[PIPE_MAX_INST + 1] = {};         // [ni]: role table (256) + ni block tables (L * P + 1 each) of the ni-instance layout
    int grid_ni[PIPE_MAX_INST + 1] = {};
};

// Which workgroup plays which CU.  Block b runs on XCD b % 8 (observed; speed only): one instance keeps the scheme of rounds 2-4 (spx consecutive
// layers per XCD, the head on XCD 0, <= 30 CUs per XCD); several instances share ONE launch of 256 workgroups and are packed XCD by XCD, a layer's P CUs
// never split.  (One launch PER instance was built first and fails: where a kernel's ids land is only modulo-8 regular WITHIN a launch, so the
// instances' CU sets collide on some XCD, a launch stays partially resident and its hand-offs time out -- measured, profiles/r7f.)
// Heads of a single-instance run.  Past ~10 streams a run is bound by the slowest CU's service time per stream, and the head's (wait for the P running
// skip sums, two convolutions, the sampler, the input convolution of the next step: ~3.2 us) is longer than a layer CU's once that one multiplies its past
// taps in one batch per sample (profiles/r8a_pipe_svc_trace.txt): streams alternate between PIPE_HEADS head CUs, all on XCD 0 next to layer 0.
#define PIPE_HEADS 2
#define PIPE_EARLY_FROM 18         // streams per run from which the early x / skip requests are on (WN_PIPE_EARLY_FROM); measured on the paper model: 14 / 16 streams
                                   // 30.5 / 34.7 us per sample without, 32.0 / 35.8 with; 20 streams 43.8 without, 42.5 with (profiles/r8h_pipe_batch_scaling_ab.txt)
static int pipe_heads(int L, int P) { const int spx = (L + 7) / 8; return std::max(1, std::min(PIPE_HEADS, 30 - spx * P)); }
static bool pipe_layout(int L, int P, int ni, std::vector<int32_t>& role, std::vector<int32_t>& blk, int& grid) {
    if (ni == 1) {
        const int spx = (L + 7) / 8, nh = pipe_heads(L, P);
        blk.assign((size_t)L * P + nh, -1);
        grid = 8 * (spx * P + nh);
        role.assign(grid, -1);
        for (int b = 0; b < grid; ++b) {
            const int xcd = b & 7, slot = b >> 3;
            if (slot < spx * P) { const int layer = spx * xcd + slot / P, j = slot % P; if (layer < L) { role[b] = (layer << 8) | j; blk[layer * P + j] = b; } }
            else if (xcd == 0) { const int h = slot - spx * P; role[b] = (1 << 23) | h; blk[L * P + h] = b; }
        }
        return true;
    }
    const int per = L * P + 1;
    blk.assign((size_t)ni * per, -1);
    grid = 256; role.assign(grid, -1);           // ONE launch for all instances: every XCD gets exactly 32 ids, any CU of the XCD takes any of them
    int xcd = 0, used = 0;
    auto place = [&](int n) { if (used + n > 32) { ++xcd; used = 0; } const int k = used; used += n; return xcd < 8 ? k : -1; };
    for (int i = 0; i < ni; ++i) {
        for (int l = 0; l < L; ++l) {
            const int k = place(P); if (k < 0) return false;
            for (int j = 0; j < P; ++j) { const int b = 8 * (k + j) + xcd; role[b] = (i << 24) | (l << 8) | j; blk[(size_t)i * per + l * P + j] = b; }
        }
        const int k = place(1); if (k < 0) return false;
        role[8 * k + xcd] = (i << 24) | (1 << 23); blk[(size_t)i * per + L * P] = 8 * k + xcd;
    }
    return true;
}
static int pipe_ni_max(int L, int P) {
    std::vector<int32_t> r, b; int g;
    for (int ni = PIPE_MAX_INST; ni > 1; --ni) if (pipe_layout(L, P, ni, r, b, g)) return ni;
    return 1;
}
// instances a batch of B streams is cut into: runs of <= 8 streams cost the wall time of ONE stream (10: + 5 %, then + 3.6 us per stream)
static int pipe_instances(int L, int P, int B) {
    const char* e = getenv("WN_PIPE_INSTANCES"); const int env = e ? atoi(e) : 0;      // A/B switch: 1 = one run for the whole batch
    const int want = env > 0 ? env : (B + 7) / 8;
    return std::max(1, std::min(std::min(want, pipe_ni_max(L, P)), B));
}


// test hook (no context, no GPU): the role / block tables the persistent pipeline would launch with for a model of L layers and P CUs per layer cut into
// ni instances.  role [grid], blk [ni == 1 ? L * P + heads : ni * (L * P + 1)]; returns the instance count the chip holds (pipe_ni_max) or < 0.
#ifndef WN_NO_TEST_HOOKS
extern "C" int wn_test_pipe_layout(int32_t L, int32_t P, int32_t ni, int32_t* role, int32_t cap_role, int32_t* blk, int32_t cap_blk, int32_t* grid, int32_t* heads) {
    if (L <= 0 || L > 32 || P <= 0 || P > 8 || ni < 1 || ni > PIPE_MAX_INST || !role || !blk || !grid || !heads) return WN_E_ARG;
    std::vector<int32_t> r, b; int g = 0;
    if (!pipe_layout(L, P, ni, r, b, g)) return WN_E_SHAPE;
    if ((int)r.size() > cap_role || (int)b.size() > cap_blk) return WN_E_ARG;
    std::copy(r.begin(), r.end(), role); std::copy(b.begin(), b.end(), blk);
    *grid = g; *heads = ni == 1 ? pipe_heads(L, P) : 1;
    return pipe_ni_max(L, P);
}
#endif

void wn_pipe_free(wn_ctx* c) {
    Pipe* p = (Pipe*)c->pipe;
    if (!p) return;
    if (p->slices) hipFree(p->slices); if (p->jobs_dev) hipFree(p->jobs_dev); if (p->job_block0_dev) hipFree(p->job_block0_dev);
    if (p->XM) hipFree(p->XM); if (p->SM) hipFree(p->SM); if (p->XML) hipFree(p->XML); if (p->SML) hipFree(p->SML); if (p->ring) hipFree(p->ring); if (p->abort_dev) hipFree(p->abort_dev);
    if (p->priv) (void)hipStreamSynchronize(p->priv);
    if (p->abort_host) hipHostFree(p->abort_host);
    if (p->ev0) hipEventDestroy(p->ev0); if (p->ev1) hipEventDestroy(p->ev1); if (p->priv) hipStreamDestroy(p->priv);
    for (int i = 0; i <= PIPE_MAX_INST; ++i) if (p->tabs_dev[i]) hipFree(p->tabs_dev[i]);
    delete p; c->pipe = nullptr;
}

// can this model run on the persistent pipeline?  (one CU per 32 gate pairs, all of a CU's weights in 160 KiB of LDS)
bool wn_pipe_eligible(const wn_ctx* c, int B) {
    if (c->cfg.compute_dtype == WN_COMPUTE_F32) return false;          // the reference's arithmetic: wn_synth_f32.hip (fp32 weights and queues)
    const int R = c->R, S = c->S, C = c->C, GH = c->GH, L = c->L;
    if (GH % 32 || R % 8 || S % 8 || C % 8 || R > 512) return false;
    const int P = GH / 32;
    if (P > 8 || L > 32 || B > 32 || R > 384 || S > 384 || c->OP > 256 || (2 * R + C) / 8 > 128) return false;      // (B: 256 B of LDS per stream, checked below; beyond 10 streams a run costs + 3.6 us per stream and sample)
    const int spx = (L + 7) / 8;
    if (spx * P + 1 > 30) return false;                 // 32 CUs per XCD, keep slack
    const int64_t layer_static = 64LL * R * 2 + 64LL * (2 * R + C) * 2 + 32LL * R * 2 + 32LL * S * 2 + 256 + R * 4;
    const int ni = pipe_instances(L, P, B), Bi = (B + ni - 1) / ni;                  // streams of ONE run (z_past: 256 B of LDS each)
    const int64_t layer_dyn = R * 2 + R * 4 + 16LL * R + 8LL * R + 1024 + 64 + (R + 16) * 2 + (S + 4) * 4 + (2 * R + C) * 2 + 4LL * PIPE_ZS * Bi + 64;
    const int64_t head_static = (int64_t)S * S * 2 + (int64_t)c->OP * S * 2 + S * 4 * 2 + c->OP * 4 + R * 8 + 64;
    const int64_t head_dyn = 16LL * (S + 4) + S * 4 + c->OP * 4 + (R + 16) * 2 + 64;
    return layer_static + layer_dyn <= 160 * 1024 && head_static + head_dyn <= 160 * 1024;
}

static int pipe_build(wn_ctx* c, Pipe* p) {
    const int L = c->L, R = c->R, G = c->G, GH = c->GH, S = c->S, C = c->C, O = c->O, OP = c->OP;
    const int P = GH / 32;
    p->P = P; p->spx = (L + 7) / 8; p->grid = 8 * (p->spx * P + pipe_heads(L, P));
    PipeArgs& a = p->proto; memset(&a, 0, sizeof a);
    auto al = [](int64_t x) { return (x + 255) / 256 * 256; };
    int64_t o = 0;
    a.off_w1c = (int)o; o += 64LL * R * 2;
    a.off_w1p = (int)o; o += 64LL * (2 * R + C) * 2;
    a.off_wo = (int)o; o += 32LL * R * 2;
    a.off_ws = (int)o; o += 32LL * S * 2;
    a.off_zb = (int)o; o += 256;
    a.off_ob = (int)o; o += R * 4;
    a.layer_lds_static = (int)((o + 15) / 16 * 16);
    p->layer_slice_bytes = al(a.layer_lds_static);
    int64_t h = 0;
    a.hoff_wh1 = (int)h; h += (int64_t)S * S * 2;
    a.hoff_wh2 = (int)h; h += (int64_t)OP * S * 2;
    a.hoff_b1 = (int)h; h += S * 4;
    a.hoff_b2 = (int)h; h += OP * 4;
    a.hoff_sb = (int)h; h += S * 4;
    a.hoff_win = (int)h; h += R * 4;
    a.hoff_bin = (int)h; h += R * 4;
    a.head_lds_static = (int)((h + 15) / 16 * 16);
    p->head_slice_off = (int64_t)L * P * p->layer_slice_bytes;
    p->slices_bytes = p->head_slice_off + al(a.head_lds_static);
    WN_HIP(c, hipMalloc((void**)&p->slices, p->slices_bytes));
    WN_HIP(c, hipMemset(p->slices, 0, p->slices_bytes));
    p->layer_lds = a.layer_lds_static + (R * 2 + R * 4 + 16 * R + 8 * R + 1024 + 64 + (R + 16) * 2 + (S + 4) * 4 + (2 * R + C) * 2 + 64);      // + 256 B of z_past per stream, added at launch
    p->head_lds = a.head_lds_static + (16 * (S + 4) + S * 4 + OP * 4 + (R + 16) * 2 + 64);

    std::vector<SliceJob> jobs; std::vector<int> b0; int nblocks = 0;
    auto add = [&](SliceJob jb) { b0.push_back(nblocks); const int64_t n = jb.kind == 0 ? (int64_t)jb.kchunks * jb.rows * 8 : jb.rows; nblocks += cdiv(n, 256); jobs.push_back(jb); };
    auto img = [&](int64_t dst, int64_t src, int rows, int kchunks, int stride_k, int stride_row, float scale, int base, int split) {
        SliceJob jb; memset(&jb, 0, sizeof jb); jb.dst = dst; jb.src = src; jb.rows = rows; jb.kchunks = kchunks; jb.stride_k = stride_k; jb.stride_row = stride_row;
        jb.kind = 0; jb.scale = scale; jb.row_perm_base = base; jb.row_perm_split = split; add(jb); };
    auto vecj = [&](int64_t dst, int64_t src, int rows, float scale, int base, int split) {
        SliceJob jb; memset(&jb, 0, sizeof jb); jb.dst = dst; jb.src = src; jb.rows = rows; jb.kind = 1; jb.scale = scale; jb.row_perm_base = base; jb.row_perm_split = split; jb.stride_k = -1; add(jb); };
    for (int l = 0; l < L; ++l) {
        const WnLayerOffsets& lo = c->lay[l];
        for (int j = 0; j < P; ++j) {
            const int64_t base = (int64_t)(l * P + j) * p->layer_slice_bytes;
            // W1c: current-time tap (kernel index 2, modules.py:306-325); rows = [32 tanh | 32 sigmoid] channels of this CU
            img(base + a.off_w1c, lo.dil_k + 2LL * R * G, 64, R / 8, G, 1, 1.0f, 32 * j, 32);
            // W1p: taps t-2d (index 0), t-d (index 1), then the conditioning 1x1
            img(base + a.off_w1p, lo.dil_k, 64, R / 8, G, 1, 1.0f, 32 * j, 32);
            img(base + a.off_w1p + 64LL * R * 2, lo.dil_k + 1LL * R * G, 64, R / 8, G, 1, 1.0f, 32 * j, 32);
            img(base + a.off_w1p + 2 * 64LL * R * 2, lo.cin_k, 64, C / 8, G, 1, 1.0f, 32 * j, 32);
            // Wo / Ws: the 32 input columns (gate outputs) of this CU; out_k [GH][R], skip_k [GH][S]
            img(base + a.off_wo, lo.out_k + 32LL * j * R, R, 4, R, 1, 1.0f, 0, 0);
            img(base + a.off_ws, lo.skip_k + 32LL * j * S, S, 4, S, 1, c->skip_scale[l], 0, 0);
            vecj(base + a.off_zb, lo.dil_b, 64, 1.0f, 32 * j, 32);          // (+ cin_b: wn_pipe_fixup_kernel)
            vecj(base + a.off_ob, lo.out_b, R, 1.0f, 0, 0);
        }
    }
    const int64_t hb = p->head_slice_off;
    img(hb + a.hoff_wh1, c->fin1_k, S, S / 8, S, 1, 1.0f, 0, 0);
    // (the Wh2 image [kc][OP rows][8] and the summed skip bias are written by wn_pipe_fixup_kernel)
    vecj(hb + a.hoff_b1, c->fin1_b, S, 1.0f, 0, 0);
    vecj(hb + a.hoff_b2, c->fin2_b, O, 1.0f, 0, 0);
    if (c->Cin == 1) { vecj(hb + a.hoff_win, c->first.dil_k, R, 1.0f, 0, 0); vecj(hb + a.hoff_bin, c->first.dil_b, R, 1.0f, 0, 0); }
    p->njobs = (int)jobs.size(); p->nblocks = nblocks;
    WN_HIP(c, hipMalloc((void**)&p->jobs_dev, jobs.size() * sizeof(SliceJob)));
    WN_HIP(c, hipMemcpy(p->jobs_dev, jobs.data(), jobs.size() * sizeof(SliceJob), hipMemcpyHostToDevice));
    WN_HIP(c, hipMalloc((void**)&p->job_block0_dev, b0.size() * sizeof(int)));
    WN_HIP(c, hipMemcpy(p->job_block0_dev, b0.data(), b0.size() * sizeof(int), hipMemcpyHostToDevice));
    WN_HIP(c, hipMalloc((void**)&p->abort_dev, 256 + PIPE_MAX_INST * 4096));       // [0]: abort flag; +256: one XCC table per instance (grid <= 1024 entries)
    WN_HIP(c, hipMemset(p->abort_dev, 0, 256 + PIPE_MAX_INST * 4096));
    WN_HIP(c, hipHostMalloc((void**)&p->abort_host, 64, hipHostMallocDefault));
    *p->abort_host = 0;
    WN_HIP(c, hipStreamCreateWithFlags(&p->priv, hipStreamNonBlocking));
    WN_HIP(c, hipEventCreateWithFlags(&p->ev0, hipEventDisableTiming));
    WN_HIP(c, hipEventCreateWithFlags(&p->ev1, hipEventDisableTiming));
    p->ni_max = pipe_ni_max(L, P);
    for (int ni = 1; ni <= p->ni_max; ++ni) {
        std::vector<int32_t> role, blk; int grid = 0;
        if (!pipe_layout(L, P, ni, role, blk, grid)) WN_FAIL(c, WN_E_STATE, "pipeline layout for %d instances", ni);
        p->grid_ni[ni] = grid;
        std::vector<int32_t> all(role);                 // role table (grid) then the ni block tables (L * P + 1 each)
        all.insert(all.end(), blk.begin(), blk.end());
        WN_HIP(c, hipMalloc((void**)&p->tabs_dev[ni], all.size() * 4));
        WN_HIP(c, hipMemcpy(p->tabs_dev[ni], all.data(), all.size() * 4, hipMemcpyHostToDevice));
    }
    return WN_OK;
}

// small fix-up kernels of the slice images: z bias = dil_b + cin_b ; skip bias total ; Wh2 image with row pitch OP
__global__ void wn_pipe_fixup_kernel(const float* __restrict__ params, char* __restrict__ slices, const PipeArgs a, int64_t layer_slice_bytes,
                                     const float* __restrict__ skip_bias_total, int64_t fin2_k, int f16) {
    const int l = blockIdx.x / a.P, j = blockIdx.x % a.P;
    if (blockIdx.x < a.L * a.P) {
        float* zb = reinterpret_cast<float*>(slices + (int64_t)(l * a.P + j) * layer_slice_bytes + a.off_zb);
        for (int r = threadIdx.x; r < 64; r += blockDim.x) {
            const int ch = r < 32 ? 32 * j + r : a.GH + 32 * j + (r - 32);
            zb[r] += params[a.cin_b_off[l] + ch];
        }
    } else {
        float* sb = reinterpret_cast<float*>(slices + a.head_slice_off + a.hoff_sb);
        for (int r = threadIdx.x; r < a.S; r += blockDim.x) sb[r] = skip_bias_total[r];
        // Wh2 image [kc][OP rows][8]
        bf16_t* w = reinterpret_cast<bf16_t*>(slices + a.head_slice_off + a.hoff_wh2);
        const int n = (a.S / 8) * a.OP * 8;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int e = i & 7, row = (i >> 3) % a.OP, kc = (i >> 3) / a.OP;
            const float wv = row < a.O ? params[fin2_k + (int64_t)(kc * 8 + e) * a.O + row] : 0.0f;
            w[i] = f16 ? f2n<1>(wv) : f2bf(wv);
        }
    }
}

// the run's flag is folded into the sticky word that travels to the host: back-to-back runs (batches of more than 8 streams go
// through the pipeline in groups) each clear and overwrite the per-run flag, so a timeout in any group but the last would be lost
__global__ void wn_pipe_sticky_kernel(int32_t* f) { if (threadIdx.x == 0 && f[0] != 0 && f[1] == 0) f[1] = f[0]; }

// Size mailboxes and ring queues for B streams (they do not depend on T; the conditioning lives in the ctx workspace).  Called from
// wn_create on inference-only contexts, so that wn_synthesize never allocates there; training contexts get here on first use.
int wn_pipe_reserve(wn_ctx* c, int B, int T) {
    (void)T;
    Pipe* p = (Pipe*)c->pipe;
    int rc;
    if (!p) { p = new Pipe(); c->pipe = p; if ((rc = pipe_build(c, p))) return rc; }
    const int L = c->L, R = c->R, P = p->P;
    const size_t xm = (size_t)(L + 1) * B * P * PIPE_XG * 16, sm = (size_t)(L + 1) * B * P * PIPE_SG * 16;
    int64_t roff = 0;
    for (int l = 0; l < L; ++l) { int slots = 4; while (slots < 2 * c->dil[l] + 1) slots <<= 1; roff += (int64_t)P * B * slots * R; }
    const bool grow = xm > p->xm_bytes || sm > p->sm_bytes || (size_t)roff * 2 > p->ring_bytes;
    if (!grow) return WN_OK;
    if (c->inference && p->xm_bytes) WN_FAIL(c, WN_E_SHAPE, "synthesis batch %d exceeds the pre-sized pipeline of this inference-only context", B);
    if (p->priv) WN_HIP(c, hipStreamSynchronize(p->priv));          // (growing: nothing of ours may still read the old buffers)
    if (xm > p->xm_bytes) { if (p->XM) hipFree(p->XM); if (p->XML) hipFree(p->XML); WN_HIP(c, hipMalloc((void**)&p->XM, xm)); WN_HIP(c, hipMalloc((void**)&p->XML, xm)); p->xm_bytes = xm; }
    if (sm > p->sm_bytes) { if (p->SM) hipFree(p->SM); if (p->SML) hipFree(p->SML); WN_HIP(c, hipMalloc((void**)&p->SM, sm)); WN_HIP(c, hipMalloc((void**)&p->SML, sm)); p->sm_bytes = sm; }
    if ((size_t)roff * 2 > p->ring_bytes) { if (p->ring) hipFree(p->ring); WN_HIP(c, hipMalloc((void**)&p->ring, (size_t)roff * 2)); p->ring_bytes = (size_t)roff * 2; }
    return WN_OK;
}

// Abort flag of the last pipeline run.  wait = false: report it only if that run has already finished (no synchronisation);
// wait = true: wait for it (wn_synth_check).
int wn_pipe_check(wn_ctx* c, bool wait) {
    Pipe* p = (Pipe*)c->pipe;
    if (!p || !p->pending) return WN_OK;
    if (wait) WN_HIP(c, hipEventSynchronize(p->ev1));
    else if (hipEventQuery(p->ev1) != hipSuccess) return WN_OK;     // still running: nothing to report yet
    p->pending = false;
    const int32_t flag = *p->abort_host;
    if (flag != 0) {
        *p->abort_host = 0;
        (void)hipMemsetAsync(p->abort_dev + 1, 0, 4, p->priv);      // reported: the next runs start clean (ordered before them on the pipeline's stream)
        if (flag >= 400 && flag < 400 + 64)
            WN_FAIL(c, WN_E_HIP, "synthesis pipeline (fp16 storage): the residual stream of layer %d left the half-precision range (|x| > 65504); "
                                 "use the bf16 pipeline (WN_PIPE_DTYPE=bf16) for this model", flag - 400);
        WN_FAIL(c, WN_E_HIP, "synthesis pipeline timed out waiting for a hand-off (code %d): are all %d workgroups resident?", flag, p->grid);
    }
    return WN_OK;
}

int wn_pipe_synthesize(wn_ctx* c, const float* cin, int B, int Tc, const float* noise, const void* test_inputs,
                       void* out_samples, float* out_raw, hipStream_t caller_st) {
    const int T = Tc * c->hop, L = c->L, R = c->R;
    if ((int64_t)B * T > c->NT) WN_FAIL(c, WN_E_SHAPE, "synthesis B*T = %d*%d exceeds the workspace (max_batch*max_time = %lld)", B, T, (long long)c->NT);
    int rc;
    if ((rc = wn_pipe_reserve(c, B, T))) return rc;
    Pipe* p = (Pipe*)c->pipe;
    p->f16 = c->pipe_f16;
    hipStream_t st = p->priv;
    WN_HIP(c, hipEventRecord(p->ev0, caller_st));
    WN_HIP(c, hipStreamWaitEvent(st, p->ev0, 0));
    const int P = p->P;
    PipeArgs a = p->proto;
    a.L = L; a.P = P; a.R = R; a.G = c->G; a.GH = c->GH; a.S = c->S; a.O = c->O; a.OP = c->OP; a.C = c->C; a.Cin = c->Cin; a.B = B; a.T = T;
    a.rho = c->res_scale; a.mode = c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE ? 2 : (c->O == 2 ? 1 : 0);
    a.nps = wn_noise_per_step(c); a.lsmin = a.mode == 1 ? c->cfg.log_scale_min_gauss : c->cfg.log_scale_min; a.start_id = 127; a.spx = p->spx;
    a.slices = p->slices; a.layer_slice_bytes = p->layer_slice_bytes; a.head_slice_off = p->head_slice_off;
    // ---- slice images from the current parameters (cheap: 27 MB)
    {
        hipLaunchKernelGGL(wn_pipe_slice_kernel, dim3(p->nblocks), dim3(256), 0, st, c->params_dev, p->slices, p->jobs_dev, p->job_block0_dev, p->njobs, c->GH, p->f16 ? 1 : 0);
        WN_LAUNCH_CHECK(c);
        for (int l = 0; l < L; ++l) a.cin_b_off[l] = c->lay[l].cin_b;
        hipLaunchKernelGGL(wn_pipe_fixup_kernel, dim3(L * P + 1), dim3(256), 0, st, c->params_dev, p->slices, a, p->layer_slice_bytes, c->skip_bias_total, c->fin2_k, p->f16 ? 1 : 0);
        WN_LAUNCH_CHECK(c);
    }
    // ---- mailboxes, rings
    const size_t xm = (size_t)(L + 1) * B * P * PIPE_XG * 16, sm = (size_t)(L + 1) * B * P * PIPE_SG * 16;
    int64_t roff = 0;
    for (int l = 0; l < L; ++l) {
        int slots = 4; while (slots < 2 * c->dil[l] + 1) slots <<= 1;
        a.ring_mask[l] = slots - 1; a.dil[l] = c->dil[l]; a.ring_off[l] = roff;
        roff += (int64_t)P * B * slots * R;
    }
    WN_HIP(c, hipMemsetAsync(p->XM, 0, xm, st));
    WN_HIP(c, hipMemsetAsync(p->SM, 0, sm, st));
    WN_HIP(c, hipMemsetAsync(p->XML, 0, xm, st));
    WN_HIP(c, hipMemsetAsync(p->SML, 0, sm, st));
    WN_HIP(c, hipMemsetAsync(p->abort_dev, 0, 4, st));                     // the per-run flag (the sticky word [1] survives)
    WN_HIP(c, hipMemsetAsync(p->abort_dev + 64, 0, 4096, st));            // XCC table
    a.XM = p->XM; a.SM = p->SM; a.XML = p->XML; a.SML = p->SML; a.ring = p->ring; a.abort_flag = p->abort_dev; a.xcc_tab = p->abort_dev + 64;
    // ---- conditioning for the whole utterance (wavenet.py:781-803): cbt [B*T][C] bf16
    c->fB = B; c->fT = T; c->fTc = Tc;
    if ((rc = wn_upsample_fwd(c, nullptr, cin, B, Tc, st))) return rc;
    if (c->gin > 0) { if ((rc = wn_gbias_fwd(c, B, st))) return rc; a.gbias = c->gbias; }      // wavenet.py:766-777
    a.cbt = c->cbt; a.noise = noise; a.test_inputs = test_inputs; a.out_samples = out_samples; a.out_raw = out_raw;
    a.win_global = c->params_dev + c->first.dil_k; a.bin_global = c->params_dev + c->first.dil_b;
    unsigned long long* trace_dev = nullptr; const int trace_n = 32;
#ifdef WN_PIPE_SVC_BUILD
    const bool want_trace = getenv("WN_PIPE_TRACE") != nullptr, want_svc = getenv("WN_PIPE_SVC_TRACE") != nullptr;
#else
    const bool want_trace = false, want_svc = false;      // (the stamp sites exist in the diagnostic build only: csrc/build.py --pipe-svc)
    if (getenv("WN_PIPE_TRACE") || getenv("WN_PIPE_SVC_TRACE")) { static bool told = false; if (!told) { told = true; fprintf(stderr, "[pipe] WN_PIPE_TRACE / WN_PIPE_SVC_TRACE need the diagnostic build (python tacotron-2_amd/csrc/build.py --pipe-svc)\n"); } }
#endif
    if (want_trace && T > 600) {
        WN_HIP(c, hipMalloc((void**)&trace_dev, (size_t)trace_n * 2 * (L + 2) * 8));
        WN_HIP(c, hipMemsetAsync(trace_dev, 0, (size_t)trace_n * 2 * (L + 2) * 8, st));
        a.trace = trace_dev; a.trace_t0 = 500; a.trace_n = trace_n;
    }
    unsigned long long* svc_dev = nullptr;
    if (want_svc && T > 600) {
        WN_HIP(c, hipMalloc((void**)&svc_dev, (size_t)trace_n * 16 * 8));
        WN_HIP(c, hipMemsetAsync(svc_dev, 0, (size_t)trace_n * 16 * 8, st));
        a.svc = svc_dev; a.svc_l = L / 2; a.svc_s = B / 2; a.trace_t0 = 500; a.trace_n = trace_n;
    }
    // ---- ONE launch; ni > 1: several pipeline instances side by side, instance i serving the streams [is0, is0 + iB) on its own CUs (role table bits
    // 24-25), mailbox / queue regions (offsets linear in is0) -- runs of <= 8 streams cost the wall time of one stream
    const int ni = pipe_instances(L, P, B);
    const int per = L * P + 1, grid = p->grid_ni[ni];
    const int Bmax = (B + ni - 1) / ni;
    const int lds_bytes = std::max(p->layer_lds + 4 * PIPE_ZS * Bmax, p->head_lds);      // z_past [iB][64] fp32 per layer CU (wn_pipe_eligible checked that it fits)
    a.ninst = ni; a.ring_unit = 0; a.NH = ni == 1 ? pipe_heads(L, P) : 1;
    { const char* e = getenv("WN_PIPE_EARLY_FROM"); a.early_from = e ? atoi(e) : PIPE_EARLY_FROM; }
    { const char* e = getenv("WN_PIPE_ABORT_EVERY"); a.abort_every = e ? atoi(e) : 0; }
    for (int l = 0; l < L; ++l) { a.ring_off[l] = a.ring_unit; a.ring_unit += (int64_t)P * (a.ring_mask[l] + 1) * R; }      // per STREAM: the kernel multiplies by its instance's stream count
    for (int i = 0, s0 = 0; i < ni; ++i) { a.iB[i] = B / ni + (i < B % ni ? 1 : 0); a.is0[i] = s0; s0 += a.iB[i]; }
    a.role_tab = p->tabs_dev[ni]; a.block_tab = a.role_tab + grid;
    (void)per;
    {
        // batched pre-multiplication: fast-path models (R = 256: the parked vectors take the place of the register-resident tap-2 image) up to the
        // number of vectors that image holds; WN_PIPE_BATCHPRE=0: the per-stream pre-multiplication of rounds 2-4 (A/B switch)
        const char* bpe = getenv("WN_PIPE_BATCHPRE"); const bool bp_env = !bpe || atoi(bpe) != 0;
        const int kpad = ((2 * R + c->C) / 8 + 15) / 16 * 16, vstr = ((kpad + 13) / 16 * 16 + 2) * 16;
        const bool bp = bp_env && R == 256 && Bmax <= 32 && Bmax <= (64 * R * 2) / vstr + (32 * R * 2) / vstr;      // parked vectors: the tap-2 image's place, then W_out's
        typedef void (*kern_t)(const PipeArgs);
        const char* spe = getenv("WN_PIPE_SPEC");
        const int spec = (spe && atoi(spe) == 0) ? 0 : (R == 256 && c->S == 256 && P == 8) ? 1 : (R == 128 && c->S == 128 && P == 4) ? 2 : 0;
#define PK(h, m, b) {wn_synth_pipe_kernel<h, m, b, 0>, wn_synth_pipe_kernel<h, m, b, 1>, wn_synth_pipe_kernel<h, m, b, 2>}
        static const kern_t kerns[2][2][2][3] = {{{PK(0, 0, 0), PK(0, 0, 1)}, {PK(0, 1, 0), PK(0, 1, 1)}}, {{PK(1, 0, 0), PK(1, 0, 1)}, {PK(1, 1, 0), PK(1, 1, 1)}}};
#undef PK
        kern_t kern = kerns[p->f16 ? 1 : 0][ni > 1 ? 1 : 0][bp ? 1 : 0][spec];
        c->synth_batchpre = bp ? 1 : 0;
        const int32_t cfgv[WN_SYNTH_CFG_N] = {2, ni, bp ? 1 : 0, spec, p->f16 ? 1 : 0, a.NH, a.early_from, a.abort_every, grid, Bmax};
        for (int i = 0; i < WN_SYNTH_CFG_N; ++i) c->synth_cfg[i] = cfgv[i];
        WN_HIP(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(PIPE_THREADS), lds_bytes, st, a);
    }
    WN_LAUNCH_CHECK(c);
    c->synth_instances = ni;
    // the abort flag travels to pinned host memory behind the kernel; nobody waits for it here (wn_pipe_check / the next call read it)
    if (const char* e = getenv("WN_PIPE_TEST_ABORT")) {      // test hook: raise the flag as a timed-out hand-off would, until `n` runs of this context were flagged in total
        if (p->test_aborts < atoi(e)) { ++p->test_aborts; WN_HIP(c, hipMemsetD32Async((hipDeviceptr_t)p->abort_dev, 999, 1, st)); }
    }
    hipLaunchKernelGGL(wn_pipe_sticky_kernel, dim3(1), dim3(64), 0, st, p->abort_dev);
    WN_HIP(c, hipMemcpyAsync(p->abort_host, p->abort_dev + 1, 4, hipMemcpyDeviceToHost, st));
    p->pending = true; c->synth_path = 2;
    if (svc_dev) {        // where a layer CU's service time per stream goes (diagnostic mode: synchronises)
        WN_HIP(c, hipStreamSynchronize(st));
        std::vector<unsigned long long> h((size_t)trace_n * 16);
        hipMemcpy(h.data(), svc_dev, h.size() * 8, hipMemcpyDeviceToHost); hipFree(svc_dev);
        static const char* nm[10] = {"x poll", "barrier A", "x rebuild + z + gate + barrier B", "out matvec + publish x", "barrier", "skip own matvec + barrier", "skip poll + add + barrier",
                                     "skip publish", "ring store", "pre-multiplication of the next sample"};
        double d[10] = {0}; double tot = 0; int n = 0;
        for (int i = 0; i < trace_n; ++i) { const unsigned long long* r = &h[(size_t)i * 16]; if (!r[0] || !r[10]) continue; ++n; for (int k = 0; k < 10; ++k) d[k] += (double)(r[k + 1] - r[k]); tot += (double)(r[10] - r[0]); }
        fprintf(stderr, "[pipe svc] layer %d CU 0, stream %d of %d: %.2f us per stream iteration (mean of %d):", L / 2, B / 2, B, n ? tot / n / 100.0 : 0.0, n);
        for (int k = 0; k < 10; ++k) fprintf(stderr, " | %s %.2f", nm[k], n ? d[k] / n / 100.0 : 0.0);
        double bt = 0; int bn = 0;
        for (int i = 0; i < trace_n; ++i) { const unsigned long long* r = &h[(size_t)i * 16]; if (r[11] && r[12]) { bt += (double)(r[12] - r[11]); ++bn; } }
        double b1 = 0, b2 = 0, b3 = 0;
        for (int i = 0; i < trace_n; ++i) { const unsigned long long* r = &h[(size_t)i * 16]; if (r[11] && r[12] && r[13] && r[14]) { b1 += (double)(r[13] - r[11]); b2 += (double)(r[14] - r[13]); b3 += (double)(r[12] - r[14]); } }
        fprintf(stderr, " || once per sample: batched pre-multiplication %.2f (%s) = wait for the parked vectors %.2f + matrix product %.2f + z_past store and barrier %.2f\n", bn ? bt / bn / 100.0 : 0.0,
                c->synth_batchpre ? "on" : "off", bn ? b1 / bn / 100.0 : 0.0, bn ? b2 / bn / 100.0 : 0.0, bn ? b3 / bn / 100.0 : 0.0);
    }
    if (trace_dev) {      // per-stage latencies in units of the 100 MHz real-time counter (10 ns)   [diagnostic mode: synchronises]
        WN_HIP(c, hipStreamSynchronize(st));
        std::vector<unsigned long long> h((size_t)trace_n * 2 * (L + 2));
        hipMemcpy(h.data(), trace_dev, h.size() * 8, hipMemcpyDeviceToHost); hipFree(trace_dev);
        const int W = 2 * (L + 2);
        std::vector<double> hop(L + 1, 0.0), comp(L, 0.0); double hskip = 0, hconv = 0, hsamp = 0, hpub = 0, step = 0;
        for (int i = 0; i < trace_n; ++i) {
            const unsigned long long* r = &h[(size_t)i * W];
            for (int l = 0; l < L; ++l) { comp[l] += (double)(r[2 * l + 1] - r[2 * l]); if (l > 0) hop[l] += (double)(r[2 * l] - r[2 * (l - 1) + 1]); }
            hskip += (double)(r[2 * L] - r[2 * (L - 1) + 1]); hconv += (double)(r[2 * L + 1] - r[2 * L]); hsamp += (double)(r[2 * L + 2] - r[2 * L + 1]); hpub += (double)(r[2 * L + 3] - r[2 * L + 2]);
            if (i + 1 < trace_n) { hop[0] += (double)(h[(size_t)(i + 1) * W] - r[2 * L + 3]); step += (double)(h[(size_t)(i + 1) * W] - r[0]); }
        }
        fprintf(stderr, "[pipe trace] us: step %.2f | head: skip-wait %.2f conv %.2f sample %.2f publish %.2f | head->L0 hop %.2f\n", step / (trace_n - 1) / 100.0,
                hskip / trace_n / 100.0, hconv / trace_n / 100.0, hsamp / trace_n / 100.0, hpub / trace_n / 100.0, hop[0] / (trace_n - 1) / 100.0);
        fprintf(stderr, "[pipe trace] per layer (hop-in, compute) us:");
        for (int l = 0; l < L; ++l) fprintf(stderr, " %d:(%.2f,%.2f)", l, l ? hop[l] / trace_n / 100.0 : 0.0, comp[l] / trace_n / 100.0);
        fprintf(stderr, "\n[pipe trace] head us: skip-wait %.2f conv %.2f sample %.2f publish %.2f | head->L0 hop %.2f | step %.2f\n", hskip / trace_n / 100.0, hconv / trace_n / 100.0,
                hsamp / trace_n / 100.0, hpub / trace_n / 100.0, hop[0] / (trace_n - 1) / 100.0, step / (trace_n - 1) / 100.0);
    }
    WN_HIP(c, hipEventRecord(p->ev1, st));
    WN_HIP(c, hipStreamWaitEvent(caller_st, p->ev1, 0));
    return WN_OK;
}
