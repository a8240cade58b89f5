// Fast-WaveNet autoregressive synthesis in the REFERENCE'S OWN ARITHMETIC (cfg.compute_dtype = WN_COMPUTE_F32):
// WaveNet.incremental is fp32 throughout -- fp32 queues, fp32 `linearized_weights`, fp32 matmul (modules.py:273-303,
// wavenet.py:821-886).  The real-time paths (wn_synth_pipe.hip, wn_synth.hip) keep weights and queues in bf16; this one reads
// the fp32 parameters straight from the ctx-owned flat TF-layout buffer (no packing), keeps the Fast-WaveNet queues as fp32 ring
// buffers and accumulates every matvec in fp32 in a fixed order, with precise tanhf / expf.  A validation mode: speed is not the
// point (one launch per layer stage, captured in the same hipGraph scheme as the bf16 launch-per-layer path).
//
// Matvec shape: out[n][o] = sum_k W[k][o] * in[n][k], W row-major [K][ld] exactly as TensorFlow stores a [k, in, out] kernel
// (reshape(kernel, [k * in, out]) of modules.py:251 is this buffer, tap 0 first), so a wave reads 64 consecutive output columns
// of one k per load (coalesced 256 B), the input vectors of up to 8 streams sit in LDS and are read as broadcasts, the 4 waves of
// a workgroup split K and their partial sums are added in wave order.
#include "wn_common.h"
#include <stdlib.h>

#define F32S_NS 8        // streams per workgroup

struct SynthF32 {
    int capB = 0;                                   // streams the rings are sized for
    std::vector<float*> ring; std::vector<int> mask;
    float *ucur = nullptr, *skip_acc = nullptr, *h2 = nullptr, *yraw = nullptr;
    int32_t* t_dev = nullptr;
    hipStream_t priv = nullptr; hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipGraphExec_t gexec = nullptr; int g_steps = 0, g_B = 0, g_T = 0; const void* g_key[5] = {0, 0, 0, 0, 0};
};

// ---- the shared core: K split over the 4 waves, one output column per lane, F32S_NS streams per workgroup.
// `fill(k, s)` returns input element k of local stream s; `wcol(k)` the weight of (k, this lane's column `sel`).  NCOL columns per
// lane (the gate needs its tanh and sigmoid pre-activations side by side).  Returns the summed accumulators of (stream s, column c)
// for the streams this thread finalises: s = wave and wave + 4.
template <int NCOL, class Fill, class Wcol, class Epi>
__device__ __forceinline__ void f32s_matvec(float* smem, const int K, Fill fill, Wcol wcol, Epi epi) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < F32S_NS * K; i += 256) { const int s = i / K, k = i - s * K; smem[s * K + k] = fill(k, s); }
    __syncthreads();
    float acc[NCOL][F32S_NS];
#pragma unroll
    for (int c = 0; c < NCOL; ++c)
#pragma unroll
        for (int s = 0; s < F32S_NS; ++s) acc[c][s] = 0.0f;
    const int kq = (K + 3) / 4, k0 = wave * kq, k1 = min(K, k0 + kq);
#pragma unroll 8                                     // eight weight loads in flight per lane (same order of the additions)
    for (int k = k0; k < k1; ++k) {
        float w[NCOL];
#pragma unroll
        for (int c = 0; c < NCOL; ++c) w[c] = wcol(k, c);
#pragma unroll
        for (int s = 0; s < F32S_NS; ++s) {
            const float x = smem[s * K + k];
#pragma unroll
            for (int c = 0; c < NCOL; ++c) acc[c][s] = __builtin_fmaf(w[c], x, acc[c][s]);
        }
    }
    __syncthreads();                                 // the input vectors are dead: the same LDS carries the partial sums
    float* red = smem;                               // [wave][s][c][lane]
#pragma unroll
    for (int s = 0; s < F32S_NS; ++s)
#pragma unroll
        for (int c = 0; c < NCOL; ++c) red[((wave * F32S_NS + s) * NCOL + c) * 64 + lane] = acc[c][s];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < F32S_NS / 4; ++h) {
        const int s = wave + 4 * h;
        float v[NCOL];
#pragma unroll
        for (int c = 0; c < NCOL; ++c) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += red[((w * F32S_NS + s) * NCOL + c) * 64 + lane];      // fixed order: wave 0 .. 3
            v[c] = t;
        }
        epi(s, v);
    }
}
static inline size_t f32s_lds_bytes(int K, int ncol) { return 4 * (size_t)std::max(F32S_NS * K, 4 * F32S_NS * ncol * 64); }

// ---- z = [W_dil ; W_cin]^T [x(t-2d); x(t-d); x(t); c_t] + b -> u = tanh(a) * sigmoid(b)     (modules.py:273-303, 494-510)
__global__ __launch_bounds__(256) void wn_f32s_gate(const float* __restrict__ Wd, const float* __restrict__ Wc, int R, int C, int G, int GH,
                                                    const float* __restrict__ ring, int mask, int d, int SB,
                                                    const float* __restrict__ cup, int T, int B,
                                                    const float* __restrict__ bias, int bias_bstride, float* __restrict__ ucur,
                                                    const int32_t* __restrict__ t_dev) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, t = *t_dev, g = blockIdx.x * 64 + lane, n0 = blockIdx.y * F32S_NS, K = 3 * R + C;
    const int gc = min(g, GH - 1);
    f32s_matvec<2>(smem, K,
        [&](int k, int s) -> float {
            const int n = n0 + s;
            if (n >= B) return 0.0f;
            if (k < 3 * R) {
                const int j = k / R, i = k - j * R, tau = t - (2 - j) * d;            // kernel index 0 <-> x[t - 2d], 2 <-> x[t]
                return tau >= 0 ? ring[((size_t)(tau & mask) * SB + n) * R + i] : 0.0f;
            }
            return cup[((size_t)n * C + (k - 3 * R)) * T + t];
        },
        [&](int k, int c) -> float { const float* W = k < 3 * R ? Wd + (size_t)k * G : Wc + (size_t)(k - 3 * R) * G; return W[c * GH + gc]; },
        [&](int s, const float* v) {
            const int n = n0 + s;
            if (n >= B || g >= GH) return;
            const float* gb = bias + (size_t)n * bias_bstride;
            const float za = v[0] + gb[g], zb = v[1] + gb[GH + g];
            ucur[(size_t)n * GH + g] = tanhf(za) * (1.0f / (1.0f + expf(-zb)));
        });
}

// ---- x_next = (W_out^T u + b + x) * rho -> next layer's queue;  skips += c_l * W_skip^T u      (modules.py:512-521, wavenet.py:833-836)
__global__ __launch_bounds__(256) void wn_f32s_out(const float* __restrict__ Wo, const float* __restrict__ Ws, int R, int S, int GH,
                                                   const float* __restrict__ ucur, const float* __restrict__ out_bias, float rho,
                                                   const float* __restrict__ ring_cur, int mask_cur, float* __restrict__ ring_next, int mask_next, int SB,
                                                   float* __restrict__ skip_acc, float skip_scale, int first_layer, int B,
                                                   const int32_t* __restrict__ t_dev) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, t = *t_dev, n0 = blockIdx.y * F32S_NS;
    const int nrb = (R + 63) / 64;
    const bool is_skip = (int)blockIdx.x >= nrb;
    if (!is_skip && ring_next == nullptr) return;                                      // top layer: its residual output is never used
    const int o = (is_skip ? (int)blockIdx.x - nrb : (int)blockIdx.x) * 64 + lane;
    const int lim = is_skip ? S : R, oc = min(o, lim - 1);
    const float* const W = is_skip ? Ws : Wo;
    f32s_matvec<1>(smem, GH,
        [&](int k, int s) -> float { const int n = n0 + s; return n < B ? ucur[(size_t)n * GH + k] : 0.0f; },
        [&](int k, int) -> float { return W[(size_t)k * lim + oc]; },
        [&](int s, const float* v) {
            const int n = n0 + s;
            if (n >= B || o >= lim) return;
            if (is_skip) {
                float* p = skip_acc + (size_t)n * S + o;
                *p = first_layer ? skip_scale * v[0] : __builtin_fmaf(skip_scale, v[0], *p);
            } else {
                const float x = ring_cur[((size_t)(t & mask_cur) * SB + n) * R + o];
                ring_next[((size_t)(t & mask_next) * SB + n) * R + o] = (v[0] + out_bias[o] + x) * rho;
            }
        });
}

// ---- head: h2 = relu(W1^T relu(skips + b_skip) + b1);  y = W2^T h2 + b2     (wavenet.py:840-844)
__global__ __launch_bounds__(256) void wn_f32s_head(const float* __restrict__ W, int K, int M, const float* __restrict__ in, const float* __restrict__ in_bias,
                                                    int relu_in, const float* __restrict__ b, int relu_out, float* __restrict__ out, int ld_out, int B) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, o = blockIdx.x * 64 + lane, n0 = blockIdx.y * F32S_NS, oc = min(o, M - 1);
    f32s_matvec<1>(smem, K,
        [&](int k, int s) -> float {
            const int n = n0 + s;
            if (n >= B) return 0.0f;
            float x = in[(size_t)n * K + k] + (in_bias ? in_bias[k] : 0.0f);
            return relu_in ? fmaxf(x, 0.0f) : x;
        },
        [&](int k, int) -> float { return W[(size_t)k * M + oc]; },
        [&](int s, const float* v) {
            const int n = n0 + s;
            if (n >= B || o >= M) return;
            const float y = v[0] + b[o];
            out[(size_t)n * ld_out + o] = relu_out ? fmaxf(y, 0.0f) : y;
        });
}

// ---- sampler + bookkeeping + input convolution of the NEXT step     (wavenet.py:847-878, 826); mode 0 MoL, 1 Gaussian, 2 categorical
__global__ __launch_bounds__(256) void wn_f32s_sample(const float* __restrict__ yraw, int O, int OP, int mode, int nps, float lsmin,
                                                      const float* __restrict__ noise, const void* __restrict__ test_inputs,
                                                      void* __restrict__ out_samples, float* __restrict__ out_raw,
                                                      const float* __restrict__ Wf, const float* __restrict__ bf_, int R,
                                                      float* __restrict__ ring0, int mask0, int SB, int B, int T, int32_t* __restrict__ t_dev) {
    __shared__ float nxt_f[32];
    __shared__ int nxt_i[32];
    const int tid = threadIdx.x;
    const int t = *t_dev;
    if (tid < B) {
        const int n = tid;
        const float* p = yraw + (size_t)n * OP;
        const float* nz = noise + ((size_t)t * B + n) * nps;
        if (mode == 2) {
            float best = -INFINITY; int bi = 0;
            for (int q = 0; q < O; ++q) { const float v = p[q] - logf(-logf(nz[q])); if (v > best) { best = v; bi = q; } }
            ((int32_t*)out_samples)[(size_t)n * T + t] = bi;
            nxt_i[n] = test_inputs ? ((const int32_t*)test_inputs)[(size_t)n * T + t] : bi;
        } else {
            float x;
            if (mode == 0) {
                const int M = O / 3;
                float best = -INFINITY; int bi = 0;
                for (int i = 0; i < M; ++i) { const float v = p[i] - logf(-logf(nz[i])); if (v > best) { best = v; bi = i; } }
                const float ls = fmaxf(p[2 * M + bi], lsmin);
                const float u = nz[M];
                x = p[M + bi] + expf(ls) * (logf(u) - logf(1.0f - u));
            } else {
                x = p[0] + expf(fmaxf(p[1], lsmin)) * nz[0];
            }
            x = fminf(fmaxf(x, -1.0f), 1.0f);
            ((float*)out_samples)[(size_t)n * T + t] = x;
            nxt_f[n] = test_inputs ? ((const float*)test_inputs)[(size_t)n * T + t] : x;
        }
    }
    if (out_raw) for (int o = tid; o < B * O; o += 256) { const int n = o / O, oc = o - n * O; out_raw[((size_t)n * O + oc) * T + t] = yraw[(size_t)n * OP + oc]; }
    __syncthreads();
    for (int o = tid; o < B * R; o += 256) {          // input convolution of step t + 1 into queue 0
        const int n = o / R, r = o - n * R;
        ring0[((size_t)((t + 1) & mask0) * SB + n) * R + r] = (mode == 2) ? Wf[(size_t)nxt_i[n] * R + r] + bf_[r] : __builtin_fmaf(Wf[r], nxt_f[n], bf_[r]);
    }
    __syncthreads();
    if (tid == 0) *t_dev = t + 1;
}
// initial input (silence, wavenet.py:433-445) -> queue 0 slot 0; t = 0
__global__ void wn_f32s_init(const float* __restrict__ Wf, const float* __restrict__ bf_, int R, int mode, int start_id,
                             float* __restrict__ ring0, int B, int32_t* t_dev) {
    for (int o = threadIdx.x; o < B * R; o += blockDim.x) {
        const int n = o / R, r = o - n * R;
        ring0[(size_t)n * R + r] = (mode == 2) ? Wf[(size_t)start_id * R + r] + bf_[r] : bf_[r];      // x = 0 for raw / mulaw
    }
    if (threadIdx.x == 0) *t_dev = 0;
}

void wn_synth_f32_free(wn_ctx* c) {
    SynthF32* s = (SynthF32*)c->synth32;
    if (!s) return;
    for (float* p : s->ring) if (p) hipFree(p);
    for (float* p : {s->ucur, s->skip_acc, s->h2, s->yraw}) if (p) hipFree(p);
    if (s->t_dev) hipFree(s->t_dev);
    if (s->gexec) hipGraphExecDestroy(s->gexec);
    if (s->ev0) hipEventDestroy(s->ev0);
    if (s->ev1) hipEventDestroy(s->ev1);
    if (s->priv) { (void)hipStreamSynchronize(s->priv); hipStreamDestroy(s->priv); }
    delete s; c->synth32 = nullptr;
}

// queues for `B` streams (4d slots per layer: t & (4d - 1), reads reach back 2d), per-step scratch, the capture stream
int wn_synth_f32_reserve(wn_ctx* c, int B) {
    if (!c->synth32) c->synth32 = new SynthF32();
    SynthF32* s = (SynthF32*)c->synth32;
    const int L = c->L, R = c->R;
    if (s->capB < B) {
        if (c->inference && s->capB > 0) WN_FAIL(c, WN_E_SHAPE, "fp32 synthesis: %d streams exceed the %d this inference-only context was sized for", B, s->capB);
        if (s->capB > 0) {
            (void)hipDeviceSynchronize();
            for (float*& p : s->ring) { if (p) hipFree(p); p = nullptr; }
            for (float** p : {&s->ucur, &s->skip_acc, &s->h2, &s->yraw}) { if (*p) hipFree(*p); *p = nullptr; }
        }
        if (s->gexec) { hipGraphExecDestroy(s->gexec); s->gexec = nullptr; }
        s->ring.assign(L, nullptr); s->mask.assign(L, 0);
        for (int l = 0; l < L; ++l) {
            int slots = 4; while (slots < 4 * c->dil[l]) slots <<= 1;
            s->mask[l] = slots - 1;
            WN_HIP(c, hipMalloc((void**)&s->ring[l], (size_t)slots * B * R * 4));
        }
        // per-step scratch, one row per stream, sized with the queues (wn_synthesize admits at most 32 streams)
        WN_HIP(c, hipMalloc((void**)&s->ucur, (size_t)B * c->GH * 4));
        WN_HIP(c, hipMalloc((void**)&s->skip_acc, (size_t)B * c->S * 4));
        WN_HIP(c, hipMalloc((void**)&s->h2, (size_t)B * c->S * 4));
        WN_HIP(c, hipMalloc((void**)&s->yraw, (size_t)B * c->OP * 4));
        s->capB = B;
    }
    if (!s->t_dev) WN_HIP(c, hipMalloc((void**)&s->t_dev, 4));
    if (!s->priv) WN_HIP(c, hipStreamCreateWithFlags(&s->priv, hipStreamNonBlocking));
    if (!s->ev0) WN_HIP(c, hipEventCreateWithFlags(&s->ev0, hipEventDisableTiming));
    if (!s->ev1) WN_HIP(c, hipEventCreateWithFlags(&s->ev1, hipEventDisableTiming));
    return WN_OK;
}

static int f32s_enqueue_step(wn_ctx* c, SynthF32* s, int B, int T, const float* noise, const void* test_inputs, void* out_samples, float* out_raw, hipStream_t st) {
    const int L = c->L, R = c->R, G = c->G, GH = c->GH, S = c->S, C = c->C, SB = s->capB;
    const float* P = c->params_dev;
    const float* cup = c->CUP[c->cup_final_idx];
    const int ny = cdiv(B, F32S_NS);
    for (int l = 0; l < L; ++l) {
        hipLaunchKernelGGL(wn_f32s_gate, dim3(cdiv(GH, 64), ny), dim3(256), f32s_lds_bytes(3 * R + C, 2), st, P + c->lay[l].dil_k, P + c->lay[l].cin_k, R, C, G, GH,
                           s->ring[l], s->mask[l], c->dil[l], SB, cup, T, B,
                           c->gin > 0 ? c->gbias + (size_t)l * B * G : c->b1sum + (size_t)l * G, c->gin > 0 ? G : 0, s->ucur, s->t_dev);
        const bool top = (l == L - 1);
        hipLaunchKernelGGL(wn_f32s_out, dim3(cdiv(R, 64) + cdiv(S, 64), ny), dim3(256), f32s_lds_bytes(GH, 1), st, P + c->lay[l].out_k, P + c->lay[l].skip_k, R, S, GH,
                           s->ucur, P + c->lay[l].out_b, c->res_scale, s->ring[l], s->mask[l], top ? nullptr : s->ring[l + 1], top ? 0 : s->mask[l + 1], SB,
                           s->skip_acc, c->skip_scale[l], l == 0 ? 1 : 0, B, s->t_dev);
    }
    hipLaunchKernelGGL(wn_f32s_head, dim3(cdiv(S, 64), ny), dim3(256), f32s_lds_bytes(S, 1), st, P + c->fin1_k, S, S, s->skip_acc, c->skip_bias_total, 1,
                       P + c->fin1_b, 1, s->h2, S, B);
    hipLaunchKernelGGL(wn_f32s_head, dim3(cdiv(c->O, 64), ny), dim3(256), f32s_lds_bytes(S, 1), st, P + c->fin2_k, S, c->O, s->h2, (const float*)nullptr, 0,
                       P + c->fin2_b, 0, s->yraw, c->OP, B);
    const int mode = c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE ? 2 : (c->O == 2 ? 1 : 0);
    const float lsmin = mode == 1 ? c->cfg.log_scale_min_gauss : c->cfg.log_scale_min;
    hipLaunchKernelGGL(wn_f32s_sample, dim3(1), dim3(256), 0, st, s->yraw, c->O, c->OP, mode, wn_noise_per_step(c), lsmin, noise, test_inputs, out_samples, out_raw,
                       P + c->first.dil_k, P + c->first.dil_b, R, s->ring[0], s->mask[0], SB, B, T, s->t_dev);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}

int wn_synth_f32_impl(wn_ctx* c, const float* cin, int B, int Tc, const float* noise, const void* test_inputs,
                      void* out_samples, float* out_raw, int steps_per_graph, hipStream_t caller_st) {
    const int T = Tc * c->hop;
    if ((int64_t)B * T > c->NT) WN_FAIL(c, WN_E_SHAPE, "synthesis B*T = %d*%d exceeds the workspace (max_batch*max_time = %lld)", B, T, (long long)c->NT);
    if (c->gin > 0 && (!c->have_g || c->gB != B))
        WN_FAIL(c, WN_E_STATE, "global conditioning is enabled: call wn_set_global_condition with this batch (B=%d) first [wavenet.py:766-777]", B);
    if (3 * c->R + c->C > 2000) WN_FAIL(c, WN_E_SHAPE, "fp32 synthesis: 3 * residual_channels + cin_channels = %d input taps exceed the 64 KB LDS image of 8 streams", 3 * c->R + c->C);
    int rc = wn_synth_f32_reserve(c, B);
    if (rc) return rc;
    SynthF32* s = (SynthF32*)c->synth32;
    if (steps_per_graph <= 0) steps_per_graph = 32;
    c->synth_path = 3;
    hipStream_t st = s->priv;                          // ctx-owned stream (the caller's may be the legacy NULL stream, which cannot be captured)
    WN_HIP(c, hipEventRecord(s->ev0, caller_st));
    WN_HIP(c, hipStreamWaitEvent(st, s->ev0, 0));
    c->fB = B; c->fT = T; c->fTc = Tc;
    if ((rc = wn_upsample_fwd(c, nullptr, cin, B, Tc, st))) return rc;      // fp32 [B][C][T] (wavenet.py:781-803)
    if ((rc = wn_gbias_fwd(c, B, st))) return rc;
    const int R = c->R;
    for (int l = 0; l < c->L; ++l) WN_HIP(c, hipMemsetAsync(s->ring[l], 0, (size_t)(s->mask[l] + 1) * s->capB * R * 4, st));      // zero queues (wavenet.py:815-816)
    const int mode = c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE ? 2 : (c->O == 2 ? 1 : 0);
    hipLaunchKernelGGL(wn_f32s_init, dim3(1), dim3(256), 0, st, c->params_dev + c->first.dil_k, c->params_dev + c->first.dil_b, R, mode, 127, s->ring[0], B, s->t_dev);
    WN_LAUNCH_CHECK(c);
    int done = 0;
    if (steps_per_graph > 1 && T >= steps_per_graph) {
        const void* key[5] = {noise, test_inputs, out_samples, out_raw, cin};
        const bool reuse = s->gexec && s->g_steps == steps_per_graph && s->g_B == B && s->g_T == T && memcmp(key, s->g_key, sizeof key) == 0;
        if (!reuse) {
            if (s->gexec) { hipGraphExecDestroy(s->gexec); s->gexec = nullptr; }
            hipGraph_t graph;
            WN_HIP(c, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int i = 0; i < steps_per_graph; ++i) {
                rc = f32s_enqueue_step(c, s, B, T, noise, test_inputs, out_samples, out_raw, st);
                if (rc) { hipStreamEndCapture(st, &graph); return rc; }
            }
            WN_HIP(c, hipStreamEndCapture(st, &graph));
            WN_HIP(c, hipGraphInstantiate(&s->gexec, graph, nullptr, nullptr, 0));
            hipGraphDestroy(graph);
            s->g_steps = steps_per_graph; s->g_B = B; s->g_T = T; memcpy(s->g_key, key, sizeof key);
        }
        for (; done + steps_per_graph <= T; done += steps_per_graph) WN_HIP(c, hipGraphLaunch(s->gexec, st));
    }
    for (; done < T; ++done) { rc = f32s_enqueue_step(c, s, B, T, noise, test_inputs, out_samples, out_raw, st); if (rc) return rc; }
    WN_HIP(c, hipEventRecord(s->ev1, st));
    WN_HIP(c, hipStreamWaitEvent(caller_st, s->ev1, 0));
    return WN_OK;
}
