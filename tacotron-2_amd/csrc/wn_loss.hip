// Output heads: MoL / Gaussian / softmax losses with their gradients in one pass (mixture.py:18-74, gaussian.py:5-37, modules.py:781-836),
// the samplers (mixture.py:76-107, gaussian.py:39-52, wavenet.py:861-869), the device noise stream (Philox4x32-10) and the mu-law codec
// (util.py:30-129).
#include "wn_common.h"
#include "wn_mulaw_tables.h"
#include <math.h>
#include <algorithm>

// =================================================================================== losses
__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.0f) + log1pf(__expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// scal[0] = loss accumulator, [1] = denominator, [2] = 1/denominator, [3] = non-zero count (CE)
__global__ void wn_loss_prep(const int32_t* __restrict__ lengths, int B, int T, float* scal, int shift) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float d = 0.0f;
        for (int b = 0; b < B; ++b) { int l = min(lengths[b], T); d += (float)max(l - shift, 0); }   // sum(mask[:,1:]) wavenet.py:632-638
        scal[0] = 0.0f; scal[1] = d; scal[2] = d > 0.0f ? 1.0f / d : 0.0f; scal[3] = 0.0f;
    }
}

#define WN_MAX_MIX 16
// Discretised mixture of logistics, mixture.py:18-74 + modules.py:800-817, with its gradient.
// one thread per (b, t): prediction at t scored against y[t+1] (wavenet.py:494-495).
__global__ void wn_mol_loss(const float* __restrict__ yhat, const float* __restrict__ y, const int32_t* __restrict__ lengths,
                            bf16_t* __restrict__ dY, int ldDY, float* __restrict__ scal, int B, int T, int M,
                            float num_classes, float log_scale_min, int shift, float* __restrict__ dY32) {
    // dY32 (optional): the same gradient rows in fp32 (the fp32 backward of wn_f32.hip)
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float my = 0.0f;
    if (idx < (int64_t)B * T) {
        const int b = (int)(idx / T), t = (int)(idx - (int64_t)b * T);
        const bool valid = (t + shift < T) && (t + shift < lengths[b]);
        bf16_t* drow = dY + idx * ldDY;
        if (dY32) for (int o = 0; o < ldDY; ++o) dY32[idx * ldDY + o] = 0.0f;
        if (!valid) {
            for (int o = 0; o < ldDY; ++o) drow[o] = 0;
        } else {
            const float inv_den = scal[2];
            const float yv = y[(int64_t)b * T + t + shift];
            const float* yh = yhat + ((int64_t)b * 3 * M) * T + t;
            float logit[WN_MAX_MIX], lp[WN_MAX_MIX], dmu[WN_MAX_MIX], dls[WN_MAX_MIX];
            const float D = 1.0f / (num_classes - 1.0f);
            const float logbin = logf((num_classes - 1.0f) * 0.5f);
            float mx = -INFINITY;
            for (int i = 0; i < M; ++i) { logit[i] = yh[(int64_t)i * T]; mx = fmaxf(mx, logit[i]); }
            float se = 0.0f;
            for (int i = 0; i < M; ++i) se += __expf(logit[i] - mx);
            const float lse = logf(se);
            float mlp = -INFINITY;
            for (int i = 0; i < M; ++i) {
                const float mu = yh[(int64_t)(M + i) * T];
                const float lsr = yh[(int64_t)(2 * M + i) * T];
                const float ls = fmaxf(lsr, log_scale_min);
                const float cy = yv - mu, inv = __expf(-ls);
                const float p = inv * (cy + D), m = inv * (cy - D), mid = inv * cy;
                float l, gm, gs;          // log-prob and its derivatives wrt mu and ls
                if (yv < -0.999f) { l = p - softplusf(p); const float s = sigmoidf_(-p); gm = -inv * s; gs = -p * s; }
                else if (yv > 0.999f) { l = -softplusf(m); const float s = sigmoidf_(m); gm = inv * s; gs = m * s; }
                else {
                    const float sp = sigmoidf_(p), sm = sigmoidf_(m);
                    const float cd = sp - sm;
                    if (cd > 1e-5f) {
                        l = logf(fmaxf(cd, 1e-12f));
                        const float dp = sp * (1.0f - sp), dm = sm * (1.0f - sm);
                        gm = -inv * (dp - dm) / cd; gs = (-p * dp + m * dm) / cd;
                    } else {
                        const float q = 1.0f - 2.0f * sigmoidf_(mid);
                        l = mid - ls - 2.0f * softplusf(mid) - logbin;
                        gm = -inv * q; gs = -mid * q - 1.0f;
                    }
                }
                if (lsr < log_scale_min) gs = 0.0f;       // tf.maximum passes the gradient only where x >= min
                lp[i] = l + (logit[i] - mx - lse);
                dmu[i] = gm; dls[i] = gs;
                mlp = fmaxf(mlp, lp[i]);
            }
            float sw = 0.0f;
            for (int i = 0; i < M; ++i) sw += __expf(lp[i] - mlp);
            const float loss = -(mlp + logf(sw));
            my = loss;
            for (int i = 0; i < M; ++i) {
                const float w = __expf(lp[i] - mlp) / sw;              // responsibility
                const float pi = __expf(logit[i] - mx) / se;           // prior
                drow[i] = f2bf((pi - w) * inv_den);
                drow[M + i] = f2bf(-w * dmu[i] * inv_den);
                drow[2 * M + i] = f2bf(-w * dls[i] * inv_den);
                if (dY32) { float* d32 = dY32 + idx * ldDY; d32[i] = (pi - w) * inv_den; d32[M + i] = -w * dmu[i] * inv_den; d32[2 * M + i] = -w * dls[i] * inv_den; }
            }
            for (int o = 3 * M; o < ldDY; ++o) drow[o] = 0;
        }
    }
    // block reduce
    for (int o = 32; o > 0; o >>= 1) my += __shfl_down(my, o);
    __shared__ float part[8];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = my;
    __syncthreads();
    if (threadIdx.x == 0) { float s = 0.0f; for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += part[i]; unsafeAtomicAdd(&scal[0], s); }
}

__device__ __forceinline__ float ndtrf_(float x) {      // TF special_math._ndtr, piecewise erf/erfc
    const float hs2 = 0.70710678118654752440f;
    const float w = x * hs2, z = fabsf(w);
    const float y = (z < hs2) ? 1.0f + erff(w) : ((w > 0.0f) ? 2.0f - erfcf(z) : erfcf(z));
    return 0.5f * y;
}

// Gaussian MLE, gaussian.py:5-37 + modules.py:819-836, with its gradient.
__global__ void wn_gauss_loss(const float* __restrict__ yhat, const float* __restrict__ y, const int32_t* __restrict__ lengths,
                              bf16_t* __restrict__ dY, int ldDY, float* __restrict__ scal, int B, int T,
                              float num_classes, float log_scale_min, int use_cdf, int shift, float* __restrict__ dY32) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float my = 0.0f;
    if (idx < (int64_t)B * T) {
        const int b = (int)(idx / T), t = (int)(idx - (int64_t)b * T);
        const bool valid = (t + shift < T) && (t + shift < lengths[b]);
        bf16_t* drow = dY + idx * ldDY;
        float g0 = 0.0f, g1 = 0.0f;
        if (valid) {
            const float inv_den = scal[2];
            const float yv = y[(int64_t)b * T + t + shift];
            const float mu = yhat[((int64_t)b * 2) * T + t], lsr = yhat[((int64_t)b * 2 + 1) * T + t];
            const float ls = fmaxf(lsr, log_scale_min);
            float gm, gs, loss;
            if (use_cdf) {
                const float D = 1.0f / (num_classes - 1.0f);
                const float sc = __expf(ls);
                const float zp = (yv + D - mu) / sc, zm = (yv - D - mu) / sc;
                const float diff = ndtrf_(zp) - ndtrf_(zm);
                loss = -logf(fmaxf(diff, 1e-12f));
                if (diff >= 1e-12f) {
                    const float c0 = 0.3989422804014327f;
                    const float pp = c0 * __expf(-0.5f * zp * zp), pm = c0 * __expf(-0.5f * zm * zm);
                    gm = (pp - pm) / (sc * diff);            // d(-lp)/dmu
                    gs = (zp * pp - zm * pm) / diff;         // d(-lp)/dls
                } else { gm = 0.0f; gs = 0.0f; }
            } else {
                const float e2 = __expf(-2.0f * ls), dlt = yv - mu;
                loss = 0.5f * (1.8378770664093453f + 2.0f * ls + dlt * dlt * e2);
                gm = -dlt * e2; gs = 1.0f - dlt * dlt * e2;
            }
            if (lsr < log_scale_min) gs = 0.0f;
            my = loss; g0 = gm * inv_den; g1 = gs * inv_den;
        }
        drow[0] = f2bf(g0); drow[1] = f2bf(g1);
        for (int o = 2; o < ldDY; ++o) drow[o] = 0;
        if (dY32) { float* d32 = dY32 + idx * ldDY; d32[0] = g0; d32[1] = g1; for (int o = 2; o < ldDY; ++o) d32[o] = 0.0f; }
    }
    for (int o = 32; o > 0; o >>= 1) my += __shfl_down(my, o);
    __shared__ float part[8];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = my;
    __syncthreads();
    if (threadIdx.x == 0) { float s = 0.0f; for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += part[i]; unsafeAtomicAdd(&scal[0], s); }
}

// Masked softmax cross-entropy, modules.py:781-798 (denominator = count_nonzero(masked loss)).
// pass 0: per-element loss into `tmp`, sum and non-zero count; pass 1: gradients (needs the count).
__global__ void wn_ce_loss(const float* __restrict__ yhat, const int32_t* __restrict__ y, const int32_t* __restrict__ lengths,
                           bf16_t* __restrict__ dY, int ldDY, float* __restrict__ scal, float* __restrict__ tmp,
                           int B, int T, int Q, int pass, int shift, float* __restrict__ dY32) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float my = 0.0f, cnt = 0.0f;
    if (idx < (int64_t)B * T) {
        const int b = (int)(idx / T), t = (int)(idx - (int64_t)b * T);
        const bool valid = (t + shift < T) && (t + shift < lengths[b]);
        const float* yh = yhat + ((int64_t)b * Q) * T + t;
        if (pass == 0) {
            float l = 0.0f;
            if (valid) {
                float mx = -INFINITY;
                for (int q = 0; q < Q; ++q) mx = fmaxf(mx, yh[(int64_t)q * T]);
                float se = 0.0f;
                for (int q = 0; q < Q; ++q) se += __expf(yh[(int64_t)q * T] - mx);
                const int tgt = y[(int64_t)b * T + t + shift];
                l = mx + logf(se) - yh[(int64_t)tgt * T];
            }
            tmp[idx] = l; my = l; cnt = (l != 0.0f) ? 1.0f : 0.0f;
        } else {
            bf16_t* drow = dY + idx * ldDY;
            if (dY32) for (int q = 0; q < ldDY; ++q) dY32[idx * ldDY + q] = 0.0f;
            if (!valid) { for (int q = 0; q < ldDY; ++q) drow[q] = 0; }
            else {
                const float inv = 1.0f / scal[3];
                float mx = -INFINITY;
                for (int q = 0; q < Q; ++q) mx = fmaxf(mx, yh[(int64_t)q * T]);
                float se = 0.0f;
                for (int q = 0; q < Q; ++q) se += __expf(yh[(int64_t)q * T] - mx);
                const int tgt = y[(int64_t)b * T + t + shift];
                for (int q = 0; q < Q; ++q) {
                    const float gq = (__expf(yh[(int64_t)q * T] - mx) / se - (q == tgt ? 1.0f : 0.0f)) * inv;
                    drow[q] = f2bf(gq);
                    if (dY32) dY32[idx * ldDY + q] = gq;
                }
                for (int q = Q; q < ldDY; ++q) drow[q] = 0;
            }
        }
    }
    if (pass == 0) {
        for (int o = 32; o > 0; o >>= 1) { my += __shfl_down(my, o); cnt += __shfl_down(cnt, o); }
        __shared__ float part[16];
        if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = my; part[8 + (threadIdx.x >> 6)] = cnt; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.0f, n = 0.0f;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { s += part[i]; n += part[8 + i]; }
            unsafeAtomicAdd(&scal[0], s); unsafeAtomicAdd(&scal[3], n);
        }
    }
}

__global__ void wn_loss_finalize(float* scal, float* loss_out, int use_count) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float den = use_count ? scal[3] : scal[1];
        *loss_out = scal[0] / den;
    }
}

// shift = 1: training (prediction at t scored against sample t+1, wavenet.py:488-495);
// shift = 0: evaluation of the incremental loop's raw outputs (wavenet.py:497-506).
int wn_loss_run(wn_ctx* c, const float* yhat, const void* y, const int32_t* lengths, int B, int T, int shift, float* loss_out, hipStream_t st) {
    const int64_t n = (int64_t)B * T;
    if (n > c->NT) WN_FAIL(c, WN_E_SHAPE, "loss: B*T exceeds the workspace");
    const int ldDY = (c->O + 15) / 16 * 16;
    float* const dy32 = c->dy32_next; c->dy32_next = nullptr;      // set by the fp32 forward for ITS loss call only
    hipLaunchKernelGGL(wn_loss_prep, dim3(1), dim3(64), 0, st, lengths, B, T, c->scal, shift);
    if (c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE) {
        float* tmp = c->DC;      // scratch, free at this point of the step
        hipLaunchKernelGGL(wn_ce_loss, dim3(cdiv(n, 256)), dim3(256), 0, st, yhat, (const int32_t*)y, lengths, c->DY, ldDY, c->scal, tmp, B, T, c->O, 0, shift, dy32);
        hipLaunchKernelGGL(wn_ce_loss, dim3(cdiv(n, 256)), dim3(256), 0, st, yhat, (const int32_t*)y, lengths, c->DY, ldDY, c->scal, tmp, B, T, c->O, 1, shift, dy32);
        hipLaunchKernelGGL(wn_loss_finalize, dim3(1), dim3(64), 0, st, c->scal, loss_out, 1);
    } else if (c->O == 2) {
        hipLaunchKernelGGL(wn_gauss_loss, dim3(cdiv(n, 256)), dim3(256), 0, st, yhat, (const float*)y, lengths, c->DY, ldDY, c->scal, B, T,
                           (float)c->cfg.quantize_channels, c->cfg.log_scale_min_gauss, c->cfg.cdf_loss, shift, dy32);
        hipLaunchKernelGGL(wn_loss_finalize, dim3(1), dim3(64), 0, st, c->scal, loss_out, 0);
    } else {
        if (c->O / 3 > WN_MAX_MIX) WN_FAIL(c, WN_E_UNSUPPORTED, "more than %d mixture components", WN_MAX_MIX);
        hipLaunchKernelGGL(wn_mol_loss, dim3(cdiv(n, 256)), dim3(256), 0, st, yhat, (const float*)y, lengths, c->DY, ldDY, c->scal, B, T, c->O / 3,
                           (float)c->cfg.quantize_channels, c->cfg.log_scale_min, shift, dy32);
        hipLaunchKernelGGL(wn_loss_finalize, dim3(1), dim3(64), 0, st, c->scal, loss_out, 0);
    }
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
int wn_loss_fwd_bwd(wn_ctx* c, float* loss_out, hipStream_t st) {
    WnTraceScope trace_scope(c, st, WN_TR_LOSS);
    return wn_loss_run(c, c->YHAT, c->fy, c->flen, c->fB, c->fT, 1, loss_out, st);
}
extern "C" int wn_loss(wn_ctx* c, const float* y_hat, const void* y, const int32_t* lengths, int32_t B, int32_t T, int32_t shift, float* loss_out, void* stream) {
    if (!c || !y_hat || !y || !lengths || !loss_out) return WN_E_ARG;
    if (shift != 0 && shift != 1) WN_FAIL(c, WN_E_ARG, "shift must be 0 or 1");
    if (c->inference) WN_FAIL(c, WN_E_STATE, "wn_loss on an inference-only context (the loss gradient buffer is training workspace)");
    c->have_loss = false;      // DY is overwritten
    return wn_loss_run(c, y_hat, y, lengths, B, T, shift, loss_out, (hipStream_t)stream);
}

// =================================================================================== mu-law codec
// util.py:30-129.  The quantiser is evaluated through the exact float32 decision thresholds of the
// reference's numpy float32 path (wn_mulaw_tables.h, generated by oracle/gen_mulaw_tables.py from the
// reference's own util.py): bit-exact indices for every float32 input, independent of device log1p ULPs.
__global__ void wn_mulaw_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const float s = (v > 0.0f) ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
    y[i] = (float)((double)s * log1p(255.0 * fabs((double)v)) / 5.545177444479562);
}
__global__ void wn_inv_mulaw_kernel(const float* __restrict__ y, float* __restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = y[i];
    const float s = (v > 0.0f) ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
    x[i] = (float)((double)s * (1.0 / 255.0) * (pow(256.0, fabs((double)v)) - 1.0));
}
__device__ __forceinline__ int mulaw_q(float v) {
    // number of thresholds <= v  (thresholds ascending; NaN -> 0)
    int lo = 0, hi = 255;               // answer in [0,255]
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (v >= WN_MULAW_THRESH[mid - 1]) lo = mid; else hi = mid - 1; }
    return lo;
}
__global__ void wn_mulaw_quantize_kernel(const float* __restrict__ x, int32_t* __restrict__ q, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) q[i] = mulaw_q(x[i]);
}
__global__ void wn_inv_mulaw_quantize_kernel(const int32_t* __restrict__ q, float* __restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int v = min(max(q[i], 0), 255); x[i] = WN_MULAW_DECODE[v]; }
}
__global__ void wn_argmax_kernel(const float* __restrict__ logits, int32_t* __restrict__ out, int B, int Q, int T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T) return;
    const int b = (int)(i / T), t = (int)(i - (int64_t)b * T);
    const float* p = logits + ((int64_t)b * Q) * T + t;
    float best = p[0]; int bi = 0;
    for (int q = 1; q < Q; ++q) { const float v = p[(int64_t)q * T]; if (v > best) { best = v; bi = q; } }   // first max wins (tf.argmax)
    out[i] = bi;
}

#define EW_LAUNCH(kern, n, st, ...) do { if ((n) > 0) hipLaunchKernelGGL(kern, dim3(cdiv((n), 256)), dim3(256), 0, (hipStream_t)(st), __VA_ARGS__); \
    hipError_t _e = hipGetLastError(); if (_e != hipSuccess) { g_create_err = hipGetErrorString(_e); return WN_E_HIP; } return WN_OK; } while (0)

extern "C" int wn_mulaw(const float* x, float* y, int64_t n, void* st) { if (!x || !y || n < 0) return WN_E_ARG; EW_LAUNCH(wn_mulaw_kernel, n, st, x, y, n); }
extern "C" int wn_inv_mulaw(const float* y, float* x, int64_t n, void* st) { if (!x || !y || n < 0) return WN_E_ARG; EW_LAUNCH(wn_inv_mulaw_kernel, n, st, y, x, n); }
extern "C" int wn_mulaw_quantize(const float* x, int32_t* q, int64_t n, void* st) { if (!x || !q || n < 0) return WN_E_ARG; EW_LAUNCH(wn_mulaw_quantize_kernel, n, st, x, q, n); }
extern "C" int wn_inv_mulaw_quantize(const int32_t* q, float* x, int64_t n, void* st) { if (!x || !q || n < 0) return WN_E_ARG; EW_LAUNCH(wn_inv_mulaw_quantize_kernel, n, st, q, x, n); }
extern "C" int wn_argmax_channels(const float* l, int32_t* o, int32_t B, int32_t Q, int32_t T, void* st) {
    if (!l || !o || B <= 0 || Q <= 0 || T <= 0) return WN_E_ARG; EW_LAUNCH(wn_argmax_kernel, (int64_t)B * T, st, l, o, B, Q, T); }

// =================================================================================== samplers
// mixture.py:76-107, gaussian.py:39-52, wavenet.py:861-867; noise [T][B][nps] supplied by the caller.
__device__ __forceinline__ float sample_mol(const float* p, int64_t stride, int M, const float* nz, float log_scale_min) {
    float best = -INFINITY; int bi = 0;
    for (int i = 0; i < M; ++i) { const float v = p[(int64_t)i * stride] - logf(-logf(nz[i])); if (v > best) { best = v; bi = i; } }
    const float mu = p[(int64_t)(M + bi) * stride];
    const float ls = fmaxf(p[(int64_t)(2 * M + bi) * stride], log_scale_min);
    const float u = nz[M];
    const float x = mu + expf(ls) * (logf(u) - logf(1.0f - u));
    return fminf(fmaxf(x, -1.0f), 1.0f);
}
__device__ __forceinline__ float sample_gauss(const float* p, int64_t stride, const float* nz, float lsmin) {
    const float x = p[0] + expf(fmaxf(p[stride], lsmin)) * nz[0];
    return fminf(fmaxf(x, -1.0f), 1.0f);
}
__device__ __forceinline__ int sample_cat(const float* p, int64_t stride, int Q, const float* nz) {
    float best = -INFINITY; int bi = 0;
    for (int q = 0; q < Q; ++q) { const float v = p[(int64_t)q * stride] - logf(-logf(nz[q])); if (v > best) { best = v; bi = q; } }
    return bi;
}
__global__ void wn_sample_kernel(const float* __restrict__ yhat, const float* __restrict__ noise, void* __restrict__ out,
                                 int B, int T, int O, int mode, int nps, float lsmin) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T) return;
    const int b = (int)(i / T), t = (int)(i - (int64_t)b * T);
    const float* p = yhat + ((int64_t)b * O) * T + t;
    const float* nz = noise + ((int64_t)t * B + b) * nps;
    if (mode == 0) ((float*)out)[i] = sample_mol(p, T, O / 3, nz, lsmin);
    else if (mode == 1) ((float*)out)[i] = sample_gauss(p, T, nz, lsmin);
    else ((int32_t*)out)[i] = sample_cat(p, T, O, nz);
}
int wn_sample_impl(wn_ctx* c, const float* y_hat, int B, int T, const float* noise, void* out, hipStream_t st) {
    const int mode = c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE ? 2 : (c->O == 2 ? 1 : 0);
    const float lsmin = mode == 1 ? c->cfg.log_scale_min_gauss : c->cfg.log_scale_min;
    hipLaunchKernelGGL(wn_sample_kernel, dim3(cdiv((int64_t)B * T, 256)), dim3(256), 0, st, y_hat, noise, out, B, T, c->O, mode, wn_noise_per_step(c), lsmin);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}

// =================================================================================== device noise stream
// Sampling noise of wn_synthesize(noise = NULL): replaces tf.random_uniform (mixture.py:91,104), Normal.sample (gaussian.py:50) and
// tf.multinomial's generator (wavenet.py:865).  Philox4x32-10 (Salmon et al., SC'11), key = the 64-bit seed, counter = (group index,
// 0, 0): group g yields the four 32-bit words of elements 4g .. 4g+3 of the flat [T][B][nps] buffer, so a draw depends only on
// (seed, element index) -- any launch geometry reproduces it (tests/hip_util.py mirrors it in numpy, bit for bit).
__host__ __device__ inline void wn_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
// 24 random bits -> (0, 1) open at both ends, then clamped to the reference's range [1e-5, 1 - 1e-5] (mixture.py:91,104)
__host__ __device__ inline float wn_u01(uint32_t w) { return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__global__ void wn_noise_kernel(float* __restrict__ out, int64_t n, uint32_t k0, uint32_t k1, int gaussian) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g * 4 >= n) return;
    uint32_t w[4];
    wn_philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), k0, k1, w);
    float v[4];
    if (gaussian) {                                   // Box-Muller on the word pairs (w0, w1) and (w2, w3)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float r = sqrtf(-2.0f * logf(wn_u01(w[2 * h]))), ph = 6.28318530717958647692f * wn_u01(w[2 * h + 1]);
            v[2 * h] = r * cosf(ph); v[2 * h + 1] = r * sinf(ph);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(wn_u01(w[j]), 1e-5f), 1.0f - 1e-5f);     // exact ops only (no rounding: any compiler, and the numpy mirror, give the same bits)
    }
    if (g * 4 + 3 < n) *reinterpret_cast<float4*>(out + g * 4) = make_float4(v[0], v[1], v[2], v[3]);
    else for (int j = 0; j < 4 && g * 4 + j < n; ++j) out[g * 4 + j] = v[j];
}
int wn_fill_noise_impl(wn_ctx* c, float* noise, int B, int T, uint64_t seed, hipStream_t st) {
    const int64_t n = (int64_t)B * T * wn_noise_per_step(c);
    if ((reinterpret_cast<uintptr_t>(noise) & 15) != 0) WN_FAIL(c, WN_E_ARG, "noise buffer must be 16-byte aligned");
    const int gaussian = (c->cfg.input_type != WN_INPUT_MULAW_QUANTIZE && c->O == 2) ? 1 : 0;
    hipLaunchKernelGGL(wn_noise_kernel, dim3(cdiv((n + 3) / 4, 256)), dim3(256), 0, st, noise, n, (uint32_t)seed, (uint32_t)(seed >> 32), gaussian);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
