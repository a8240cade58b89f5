// add_optimizer (wavenet.py:522-613): per-variable clip-by-norm (atomic-free norms over a host-built span table), clip-by-value, TF-Adam, EMA --
// one fused pass over the flat parameter / gradient / slot buffers.
#include "wn_common.h"
#include <math.h>
#include <algorithm>

// =================================================================================== optimiser
// wavenet.py:586-613: per-tensor tf.clip_by_norm -> tf.clip_by_value -> tf.train.AdamOptimizer (epsilon-hat) -> EMA
__device__ __forceinline__ int find_tensor(const int32_t* __restrict__ offs, int nt, int64_t i) {
    int lo = 0, hi = nt - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (offs[mid] <= i) lo = mid; else hi = mid - 1; }
    return lo;
}
#define WN_NORM_SPAN 4096      // floats per wave
// Per-variable squared norms WITHOUT atomics: the replicas of a data-parallel job apply clip_by_norm to the SAME all-reduced gradient
// and must come out bit-identical, which a float-atomic accumulation order does not give (round 2 flushed one atomic per wave and
// tensor: whenever a norm exceeded the clip threshold the scale differed by ulps between ranks and nothing re-synchronised them).
// Stage 1: one wave per span of the host-built table (never crosses a tensor), fixed lane-strided order + butterfly -> part[span];
// stage 2: one wave per tensor sums its spans in a fixed order.
__global__ __launch_bounds__(256) void wn_norm2_span_kernel(const float* __restrict__ g, const int32_t* __restrict__ spans, int nspans, float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= nspans) return;
    const int i0 = spans[2 * w], i1 = spans[2 * w + 1];
    float s = 0.0f;          // (tensor offsets are multiples of 8 floats: every span starts 16-B aligned)
    for (int j = i0 + lane * 4; j < i1; j += 256) {
        if (j + 3 < i1) { const float4 v = *reinterpret_cast<const float4*>(g + j); s += v.x * v.x; s += v.y * v.y; s += v.z * v.z; s += v.w * v.w; }
        else for (int k = j; k < i1; ++k) s += g[k] * g[k];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) part[w] = s;
}
__global__ __launch_bounds__(256) void wn_norm2_tensor_kernel(const float* __restrict__ part, const int32_t* __restrict__ first, int nt, float* __restrict__ norm2) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= nt) return;
    float s = 0.0f;
    for (int i = first[t] + lane; i < first[t + 1]; i += 64) s += part[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) norm2[t] = s;
}
__global__ void wn_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                               float* __restrict__ ema, const int32_t* __restrict__ offs, int nt, int64_t n,
                               const float* __restrict__ norm2, int clip, float max_norm, float max_value,
                               float lr_t, float b1, float b2, float eps, float ema_decay) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    auto upd = [&](float gi, float& pi, float& mi, float& vi, float& ei, float cs) {
        if (clip) gi = fminf(fmaxf(gi * max_norm / cs, -max_value), max_value);      // tf.clip_by_norm: t * clip / max(norm, clip)
        mi = b1 * mi + (1.0f - b1) * gi;
        vi = b2 * vi + (1.0f - b2) * gi * gi;
        pi = pi - lr_t * mi / (sqrtf(vi) + eps);
        ei = ei - (1.0f - ema_decay) * (ei - pi);
    };
    auto clip_scale = [&](int64_t j) { return fmaxf(sqrtf(norm2[find_tensor(offs, nt, j)]), max_norm); };      // the denominator
    if (i + 3 < n) {
        float4 G = *reinterpret_cast<const float4*>(g + i), P = *reinterpret_cast<float4*>(p + i), M = *reinterpret_cast<float4*>(m + i);
        float4 V = *reinterpret_cast<float4*>(v + i), E = *reinterpret_cast<float4*>(ema + i);
        float c0 = 1.0f, c1 = 1.0f, c2 = 1.0f, c3 = 1.0f;
        if (clip) {
            const int t0 = find_tensor(offs, nt, i);
            const int64_t tend = (t0 + 1 < nt) ? (int64_t)offs[t0 + 1] : n;
            c0 = fmaxf(sqrtf(norm2[t0]), max_norm);
            if (i + 3 < tend) { c1 = c2 = c3 = c0; }
            else { c1 = clip_scale(i + 1); c2 = clip_scale(i + 2); c3 = clip_scale(i + 3); }
        }
        upd(G.x, P.x, M.x, V.x, E.x, c0); upd(G.y, P.y, M.y, V.y, E.y, c1); upd(G.z, P.z, M.z, V.z, E.z, c2); upd(G.w, P.w, M.w, V.w, E.w, c3);
        *reinterpret_cast<float4*>(p + i) = P; *reinterpret_cast<float4*>(m + i) = M; *reinterpret_cast<float4*>(v + i) = V; *reinterpret_cast<float4*>(ema + i) = E;
    } else {
        for (int64_t j = i; j < n; ++j) {
            float pi = p[j], mi = m[j], vi = v[j], ei = ema[j];
            upd(g[j], pi, mi, vi, ei, clip ? clip_scale(j) : 1.0f);
            p[j] = pi; m[j] = mi; v[j] = vi; ema[j] = ei;
        }
    }
}

int wn_optim_impl(wn_ctx* c, float* p, const float* g, float* m, float* v, float* ema, float lr, int64_t step, hipStream_t st) {
    const wn_config& h = c->cfg;
    const int nt = (int)c->raw_tensors.size();
    const int64_t n = c->n_raw;
    struct TraceDone { wn_ctx* c; ~TraceDone() { if (c->trace_state == 3) c->trace_state = 2; } } trace_done{c};      // (destroyed AFTER trace_scope: the end stamp is enqueued first)
    WnTraceScope trace_scope(c, st, WN_TR_OPTIMISER);
    if (h.clip_gradients) {
        hipLaunchKernelGGL(wn_norm2_span_kernel, dim3(cdiv(c->norm_nspans, 4)), dim3(256), 0, st, g, c->norm_spans_dev, c->norm_nspans, c->norm_part_dev);
        hipLaunchKernelGGL(wn_norm2_tensor_kernel, dim3(cdiv(nt, 4)), dim3(256), 0, st, c->norm_part_dev, c->norm_first_dev, nt, c->norm2_dev);
    }
    const double t = (double)(step + 1);
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)h.adam_beta2, t)) / (1.0 - pow((double)h.adam_beta1, t)));
    hipLaunchKernelGGL(wn_adam_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, st, p, g, m, v, ema, c->tensor_offsets_dev, nt, n, c->norm2_dev,
                       h.clip_gradients, h.gradient_max_norm, h.gradient_max_value, lr_t, h.adam_beta1, h.adam_beta2, h.adam_epsilon, h.ema_decay);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
