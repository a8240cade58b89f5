// Fast-WaveNet autoregressive synthesis (replaces WaveNet.incremental, wavenet.py:724-911).
//
// What differs from the reference's formulation:
//   * queues are RING BUFFERS of 4*d slots per layer, indexed t & (4d-1): O(1) per step instead of the
//     slice+concat rebuild of the whole [B,2d+1,R] queue (modules.py:285-288); unwritten slots are zero,
//     which is exactly the reference's zero-initialised queue (wavenet.py:815-816);
//   * per step and layer two tiny MFMA contractions replace ~20 TF ops: streams are the N dimension
//     (<= 32) of v_mfma_f32_32x32x16_bf16, weights are the same fragment-ordered bf16 packs the
//     training kernels use (one coalesced 1-KiB load per wave and k-step, L2-resident), the 4 waves of
//     a workgroup split K and reduce through LDS;
//   * the whole step (2L+3 kernels) is captured once into a hipGraph of `steps_per_graph` steps; all
//     kernels read the time index from device memory so the graph is replayed unchanged.
#include "wn_common.h"
#include <stdlib.h>

struct Synth {
    int B = 0, T = 0;
    std::vector<bf16_t*> ring; std::vector<int> mask;
    bf16_t* ucur = nullptr;        // [32][GH]
    float* skip_acc = nullptr;     // [32][S]
    bf16_t* h2 = nullptr;          // [32][S]
    float* yraw = nullptr;         // [32][OP]
    int32_t* t_dev = nullptr;
    hipStream_t priv = nullptr;    // capture / replay happens on a ctx-owned stream (the caller's may be the legacy
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // NULL stream, which cannot be captured); ordered by events
    bf16_t** ring_tab = nullptr;   // device table of ring pointers
    hipGraphExec_t gexec = nullptr; int g_steps = 0; int g_B = 0; const void* g_key[5] = {0, 0, 0, 0, 0}; int g_T = 0;
};

__device__ __forceinline__ f32x16_t zero16() { f32x16_t z; for (int i = 0; i < 16; ++i) z[i] = 0.0f; return z; }

// map (channel-in-tile ch, stream n) -> (lane, reg) of the 32x32 accumulator layout
__device__ __forceinline__ float acc_at(const float* red, int wave_stride, int nw, int tile, int ch, int n) {
    const int reg = ((ch >> 3) << 2) | (ch & 3), lane = n + 32 * ((ch >> 2) & 1);
    float s = 0.0f;
    for (int w = 0; w < nw; ++w) s += red[(size_t)w * wave_stride + (tile * 64 + lane) * 16 + reg];
    return s;
}

// Launch-per-layer kernels are LATENCY-bound (a few blocks per launch, one dependent chain of L2 reads per wave): round 5 splits K over NW waves
// (4 or 8 by the K of the launch) and issues EVERY fragment load of a wave up front (<= WN_SYN_MAXK k-steps per pass, unrolled and predicated:
// one L2 round trip per pass instead of one per k-step).  C5 width (G = 1024, K = 1616): 834 -> see profiles/ (DESIGN 3.4).
#define WN_SYN_MAXK 8
// ---- stage A: z = [W_dil | W_cin] [x(t-2d); x(t-d); x(t); c_t] + b  -> tanh * sigmoid -> u    (modules.py:273-303, 494-510)
template <int NW>
__global__ __launch_bounds__(NW * 64) void wn_synth_gate(const bf16_t* __restrict__ Apk, int ksteps, const bf16_t* __restrict__ ring, int mask,
                                                     int d, int R, const bf16_t* __restrict__ cbt, int C, int T, int B,
                                                     const float* __restrict__ bias, int bias_bstride, int GH, bf16_t* __restrict__ ucur,
                                                     const int32_t* __restrict__ t_dev, int kil) {
    __shared__ float red[NW * 2 * 64 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = *t_dev;
    const int n = lane & 31, h = lane >> 5;
    const int blk = blockIdx.x;
    f32x16_t acc0 = zero16(), acc1 = zero16();
    const bf16_t* A0 = Apk + ((size_t)(2 * blk) * ksteps * 64 + lane) * 8;
    const bf16_t* A1 = Apk + ((size_t)(2 * blk + 1) * ksteps * 64 + lane) * 8;
    for (int ks0 = wave; ks0 < ksteps; ks0 += NW * WN_SYN_MAXK) {
        uint4 bv[WN_SYN_MAXK], a0[WN_SYN_MAXK], a1[WN_SYN_MAXK];
#pragma unroll
        for (int i = 0; i < WN_SYN_MAXK; ++i) {
            const int ks = ks0 + i * NW;
            bv[i] = make_uint4(0, 0, 0, 0); a0[i] = make_uint4(0, 0, 0, 0); a1[i] = make_uint4(0, 0, 0, 0);
            if (ks < ksteps) {
                const int k0 = ks * 16 + h * 8;
                if (n < B) {
                    if (k0 < 3 * R) {
                        // K order of the pack: [tap0 | tap1 | tap2] or, interleaved in blocks of `kil` channels, [tap0 b0 | tap1 b0 | tap2 b0 | tap0 b1 | ...]
                        int j, r;
                        if (kil > 0) { const int kb = k0 / kil; j = kb % 3; r = (kb / 3) * kil + (k0 - kb * kil); }
                        else { j = k0 / R; r = k0 - j * R; }
                        const int tau = t - (2 - j) * d;
                        if (tau >= 0) bv[i] = *reinterpret_cast<const uint4*>(ring + ((size_t)(tau & mask) * 32 + n) * R + r);
                    } else {
                        bv[i] = *reinterpret_cast<const uint4*>(cbt + ((size_t)n * T + t) * C + (k0 - 3 * R));
                    }
                }
                a0[i] = *reinterpret_cast<const uint4*>(A0 + (size_t)ks * 512);
                a1[i] = *reinterpret_cast<const uint4*>(A1 + (size_t)ks * 512);
            }
        }
#pragma unroll
        for (int i = 0; i < WN_SYN_MAXK; ++i) {      // (k-steps past the end multiply zeros)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a0[i]), __builtin_bit_cast(bf16x8_t, bv[i]), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a1[i]), __builtin_bit_cast(bf16x8_t, bv[i]), acc1, 0, 0, 0);
        }
    }
    float* my = red + (size_t)wave * (2 * 64 * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) { my[(0 * 64 + lane) * 16 + r] = acc0[r]; my[(1 * 64 + lane) * 16 + r] = acc1[r]; }
    __syncthreads();
    for (int o = tid; o < 32 * 32; o += NW * 64) {
        const int nn = o & 31, ch = o >> 5;
        if (nn >= B) continue;
        const int g = blk * 32 + ch;
        const float* const gb = bias + (size_t)nn * bias_bstride;          // per-stream bias under global conditioning (wavenet.py:766-777)
        const float za = acc_at(red, 2 * 64 * 16, NW, 0, ch, nn) + gb[g];
        const float zb = acc_at(red, 2 * 64 * 16, NW, 1, ch, nn) + gb[GH + g];
        const float e = __expf(2.0f * za);
        const float u = (1.0f - 2.0f / (e + 1.0f)) * (1.0f / (1.0f + __expf(-zb)));
        ucur[(size_t)nn * GH + g] = f2bf(u);
    }
}

// ---- stage B: x_next = (W_out u + b + x) * rho -> next layer's ring;  skip_acc += c_l W_skip u   (modules.py:512-521, wavenet.py:833-836)
template <int NW>
__global__ __launch_bounds__(NW * 64) void wn_synth_out(const bf16_t* __restrict__ Wo, const bf16_t* __restrict__ Ws, int ksteps,
                                                    int R, int S, const bf16_t* __restrict__ ucur, int GH,
                                                    const float* __restrict__ out_bias, float rho,
                                                    const bf16_t* __restrict__ ring_cur, int mask_cur,
                                                    bf16_t* __restrict__ ring_next, int mask_next,
                                                    float* __restrict__ skip_acc, int first_layer, int B,
                                                    const int32_t* __restrict__ t_dev) {
    __shared__ float red[NW * 64 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = *t_dev;
    const int n = lane & 31, h = lane >> 5;
    const int mt = blockIdx.x;
    const int nR = R >> 5;
    const bool is_skip = mt >= nR;
    if (!is_skip && ring_next == nullptr) return;        // top layer: residual output unused
    const bf16_t* A = (is_skip ? Ws + ((size_t)(mt - nR) * ksteps * 64 + lane) * 8 : Wo + ((size_t)mt * ksteps * 64 + lane) * 8);
    f32x16_t acc = zero16();
    for (int ks0 = wave; ks0 < ksteps; ks0 += NW * WN_SYN_MAXK) {
        uint4 bv[WN_SYN_MAXK], av[WN_SYN_MAXK];
#pragma unroll
        for (int i = 0; i < WN_SYN_MAXK; ++i) {
            const int ks = ks0 + i * NW;
            bv[i] = make_uint4(0, 0, 0, 0); av[i] = make_uint4(0, 0, 0, 0);
            if (ks < ksteps) {
                if (n < B) bv[i] = *reinterpret_cast<const uint4*>(ucur + (size_t)n * GH + ks * 16 + h * 8);
                av[i] = *reinterpret_cast<const uint4*>(A + (size_t)ks * 512);
            }
        }
#pragma unroll
        for (int i = 0; i < WN_SYN_MAXK; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av[i]), __builtin_bit_cast(bf16x8_t, bv[i]), acc, 0, 0, 0);
    }
    float* my = red + (size_t)wave * (64 * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) my[lane * 16 + r] = acc[r];
    __syncthreads();
    for (int o = tid; o < 32 * 32; o += NW * 64) {
        const int nn = o & 31, ch = o >> 5;
        if (nn >= B) continue;
        const float v = acc_at(red, 64 * 16, NW, 0, ch, nn);
        if (is_skip) {
            const int s = (mt - nR) * 32 + ch;
            float* p = skip_acc + (size_t)nn * S + s;
            *p = first_layer ? v : (*p + v);
        } else {
            const int r = mt * 32 + ch;
            const float x = bf2f(ring_cur[((size_t)(t & mask_cur) * 32 + nn) * R + r]);
            ring_next[((size_t)(t & mask_next) * 32 + nn) * R + r] = f2bf((v + out_bias[r] + x) * rho);
        }
    }
}

// ---- head 1: h2 = relu(W1 relu(skips + b_skip) + b1)     (wavenet.py:840-844)
__global__ __launch_bounds__(256) void wn_synth_head1(const bf16_t* __restrict__ Apk, int ksteps, int S, const float* __restrict__ skip_acc,
                                                      const float* __restrict__ skip_bias, const float* __restrict__ b1,
                                                      bf16_t* __restrict__ h2, int B) {
    __shared__ float red[4 * 64 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5, mt = blockIdx.x;
    const bf16_t* A = Apk + ((size_t)mt * ksteps * 64 + lane) * 8;
    f32x16_t acc = zero16();
    for (int ks = wave; ks < ksteps; ks += 4) {
        uint32_t w[4] = {0, 0, 0, 0};
        if (n < B) {
            const float* p = skip_acc + (size_t)n * S + ks * 16 + h * 8;
            const float* bb = skip_bias + ks * 16 + h * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) w[i] = pack_bf2(fmaxf(p[2 * i] + bb[2 * i], 0.0f), fmaxf(p[2 * i + 1] + bb[2 * i + 1], 0.0f));
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(A + (size_t)ks * 512)),
                                                      __builtin_bit_cast(bf16x8_t, make_uint4(w[0], w[1], w[2], w[3])), acc, 0, 0, 0);
    }
    float* my = red + (size_t)wave * (64 * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) my[lane * 16 + r] = acc[r];
    __syncthreads();
    for (int o = tid; o < 32 * 32; o += 256) {
        const int nn = o & 31, ch = o >> 5;
        if (nn >= B) continue;
        const int s = mt * 32 + ch;
        h2[(size_t)nn * S + s] = f2bf(fmaxf(acc_at(red, 64 * 16, 4, 0, ch, nn) + b1[s], 0.0f));
    }
}

// ---- head 2: y = W2 h2 + b2  -> yraw[n][o]
__global__ __launch_bounds__(256) void wn_synth_head2(const bf16_t* __restrict__ Apk, int ksteps, int S, const bf16_t* __restrict__ h2,
                                                      const float* __restrict__ b2, float* __restrict__ yraw, int O, int OP, int B) {
    __shared__ float red[4 * 64 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5, mt = blockIdx.x;
    const bf16_t* A = Apk + ((size_t)mt * ksteps * 64 + lane) * 8;
    f32x16_t acc = zero16();
    for (int ks = wave; ks < ksteps; ks += 4) {
        uint4 bv = make_uint4(0, 0, 0, 0);
        if (n < B) bv = *reinterpret_cast<const uint4*>(h2 + (size_t)n * S + ks * 16 + h * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(A + (size_t)ks * 512)),
                                                      __builtin_bit_cast(bf16x8_t, bv), acc, 0, 0, 0);
    }
    float* my = red + (size_t)wave * (64 * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) my[lane * 16 + r] = acc[r];
    __syncthreads();
    for (int o = tid; o < 32 * 32; o += 256) {
        const int nn = o & 31, ch = o >> 5;
        const int oc = mt * 32 + ch;
        if (nn >= B || oc >= O) continue;
        yraw[(size_t)nn * OP + oc] = acc_at(red, 64 * 16, 4, 0, ch, nn) + b2[oc];
    }
}

// ---- sampler + bookkeeping + input convolution of the NEXT step      (wavenet.py:847-878, 826)
// mode 0 MoL, 1 Gaussian, 2 categorical.  noise [T][B][nps].
__global__ __launch_bounds__(256) void wn_synth_sample(const float* __restrict__ yraw, int O, int OP, int mode, int nps, float lsmin,
                                                       const float* __restrict__ noise, const void* __restrict__ test_inputs,
                                                       void* __restrict__ out_samples, float* __restrict__ out_raw,
                                                       const float* __restrict__ Wf, const float* __restrict__ bf_, int R,
                                                       bf16_t* __restrict__ ring0, int mask0, int B, int T, int32_t* __restrict__ t_dev) {
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
    // input convolution for step t+1 into ring 0
    for (int o = tid; o < B * R; o += 256) {
        const int n = o / R, r = o - n * R;
        const float v = (mode == 2) ? Wf[(size_t)nxt_i[n] * R + r] + bf_[r] : Wf[r] * nxt_f[n] + bf_[r];
        ring0[((size_t)((t + 1) & mask0) * 32 + n) * R + r] = f2bf(v);
    }
    __syncthreads();
    if (tid == 0) *t_dev = t + 1;
}

// initial input (silence, wavenet.py:433-445) -> ring 0 slot 0; t = 0
__global__ void wn_synth_init(const float* __restrict__ Wf, const float* __restrict__ bf_, int R, int mode, int start_id,
                              bf16_t* __restrict__ ring0, int B, int32_t* t_dev) {
    for (int o = threadIdx.x; o < B * R; o += blockDim.x) {
        const int n = o / R, r = o - n * R;
        const float v = (mode == 2) ? Wf[(size_t)start_id * R + r] + bf_[r] : bf_[r];      // x = 0 for raw / mulaw
        ring0[((size_t)n) * R + r] = f2bf(v);
    }
    if (threadIdx.x == 0) *t_dev = 0;
}

void wn_synth_free(wn_ctx* c) {
    Synth* s = c->synth;
    if (!s) return;
    for (auto p : s->ring) if (p) hipFree(p);
    if (s->ucur) hipFree(s->ucur); if (s->skip_acc) hipFree(s->skip_acc); if (s->h2) hipFree(s->h2);
    if (s->yraw) hipFree(s->yraw); if (s->t_dev) hipFree(s->t_dev);
    if (s->gexec) hipGraphExecDestroy(s->gexec);
    if (s->ev0) hipEventDestroy(s->ev0); if (s->ev1) hipEventDestroy(s->ev1); if (s->priv) hipStreamDestroy(s->priv);
    delete s; c->synth = nullptr;
}

static int enqueue_step(wn_ctx* c, Synth* s, const float* noise, const void* test_inputs, void* out_samples, float* out_raw, hipStream_t st) {
    const int L = c->L, R = c->R, GH = c->GH, S = c->S, C = c->C, B = s->B, T = s->T;
    for (int l = 0; l < L; ++l) {
        // waves per block by the K of the launch: every wave should get its k-steps in ONE pass of <= WN_SYN_MAXK loads
        const int ksg = c->packs[l].w1.K >> 4, kso = GH >> 4;
        const float* gbias_l = c->gin > 0 ? c->gbias + (size_t)l * B * c->G : c->b1sum + (size_t)l * c->G;
#define WN_LAUNCH_GATE(NW_) hipLaunchKernelGGL(wn_synth_gate<NW_>, dim3(GH / 32), dim3(NW_ * 64), 0, st, c->packs[l].w1.dev, ksg, s->ring[l], s->mask[l], \
                           c->dil[l], R, c->cbt, C, T, B, gbias_l, c->gin > 0 ? c->G : 0, GH, s->ucur, s->t_dev, c->packs[l].w1.kil)
        if (ksg > 4 * WN_SYN_MAXK) WN_LAUNCH_GATE(8); else WN_LAUNCH_GATE(4);      // (16 waves = 1024 threads cap the kernel at 128 VGPRs: the up-front loads spill)
#undef WN_LAUNCH_GATE
        const bool top = (l == L - 1);
#define WN_LAUNCH_OUT(NW_) hipLaunchKernelGGL(wn_synth_out<NW_>, dim3(R / 32 + S / 32), dim3(NW_ * 64), 0, st, c->packs[l].wo.dev, c->packs[l].ws.dev, kso, R, S, s->ucur, GH, \
                           c->params_dev + c->lay[l].out_b, c->res_scale, s->ring[l], s->mask[l], top ? nullptr : s->ring[l + 1], top ? 0 : s->mask[l + 1], \
                           s->skip_acc, l == 0 ? 1 : 0, B, s->t_dev)
        if (kso > 4 * WN_SYN_MAXK) WN_LAUNCH_OUT(8); else WN_LAUNCH_OUT(4);
#undef WN_LAUNCH_OUT
    }
    hipLaunchKernelGGL(wn_synth_head1, dim3(S / 32), dim3(256), 0, st, c->wh1.dev, S >> 4, S, s->skip_acc, c->skip_bias_total, c->params_dev + c->fin1_b, s->h2, B);
    hipLaunchKernelGGL(wn_synth_head2, dim3(c->OP / 32), dim3(256), 0, st, c->wh2.dev, S >> 4, S, s->h2, c->params_dev + c->fin2_b, s->yraw, c->O, c->OP, B);
    const int mode = c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE ? 2 : (c->O == 2 ? 1 : 0);
    const float lsmin = mode == 1 ? c->cfg.log_scale_min_gauss : c->cfg.log_scale_min;
    hipLaunchKernelGGL(wn_synth_sample, dim3(1), dim3(256), 0, st, s->yraw, c->O, c->OP, mode, wn_noise_per_step(c), lsmin, noise, test_inputs, out_samples, out_raw,
                       c->params_dev + c->first.dil_k, c->params_dev + c->first.dil_b, R, s->ring[0], s->mask[0], B, T, s->t_dev);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}

// state of the launch-per-layer path: ring queues for up to 32 streams, per-step scratch, the capture stream (sizes do not depend
// on the utterance length).  wn_create calls this on inference-only contexts; training contexts get here on first use.
int wn_synth_reserve(wn_ctx* c) {
    if (c->synth) return WN_OK;
    const int L = c->L, R = c->R;
    Synth* s = new Synth(); c->synth = s;
        s->ring.assign(L, nullptr); s->mask.assign(L, 0);
        for (int l = 0; l < L; ++l) {
            int slots = 4; while (slots < 4 * c->dil[l]) slots <<= 1;
            s->mask[l] = slots - 1;
            WN_HIP(c, hipMalloc((void**)&s->ring[l], (size_t)slots * 32 * R * 2));
        }
        WN_HIP(c, hipMalloc((void**)&s->ucur, 32 * c->GH * 2));
        WN_HIP(c, hipMalloc((void**)&s->skip_acc, 32 * c->S * 4));
        WN_HIP(c, hipMalloc((void**)&s->h2, 32 * c->S * 2));
        WN_HIP(c, hipMalloc((void**)&s->yraw, 32 * c->OP * 4));
        WN_HIP(c, hipMalloc((void**)&s->t_dev, 4));
        WN_HIP(c, hipStreamCreateWithFlags(&s->priv, hipStreamNonBlocking));
        WN_HIP(c, hipEventCreateWithFlags(&s->ev0, hipEventDisableTiming));
        WN_HIP(c, hipEventCreateWithFlags(&s->ev1, hipEventDisableTiming));
    return WN_OK;
}

int wn_synth_impl(wn_ctx* c, const float* cin, int B, int Tc, const float* noise, uint64_t, const void* test_inputs,
                  void* out_samples, float* out_raw, int steps_per_graph, hipStream_t caller_st) {
    const int T = Tc * c->hop;
    if ((int64_t)B * T > c->NT) WN_FAIL(c, WN_E_SHAPE, "synthesis B*T = %d*%d exceeds the workspace (max_batch*max_time = %lld)", B, T, (long long)c->NT);
    if (c->gin > 0 && (!c->have_g || c->gB != B))
        WN_FAIL(c, WN_E_STATE, "global conditioning is enabled: call wn_set_global_condition with this batch (B=%d) first [wavenet.py:766-777]", B);
    // the reference's own arithmetic (fp32 weights, queues, accumulation): its own launch-per-layer path, never the bf16 pipeline
    if (c->cfg.compute_dtype == WN_COMPUTE_F32) return wn_synth_f32_impl(c, cin, B, Tc, noise, test_inputs, out_samples, out_raw, steps_per_graph, caller_st);
    {   // steps_per_graph <= 0 selects the persistent dataflow pipeline (wn_synth_pipe.hip) when the model fits it;
        // WN_SYNTH_MODE=graph|pipe overrides
        const char* m = getenv("WN_SYNTH_MODE");
        const bool want_pipe = m ? (strcmp(m, "pipe") == 0) : (steps_per_graph <= 0);
        // (an inference-only context never grows its pre-sized pipeline: a batch beyond pipe_cap takes the launch-per-layer path, as
        // wn_synth_pipe_eligible tells the caller -- ADVICE round 5: this dispatch used to ignore pipe_cap and fail in wn_pipe_reserve)
        const bool over_cap = c->inference && c->pipe_cap > 0 && B > c->pipe_cap;
        if (want_pipe && !over_cap && wn_pipe_eligible(c, B)) return wn_pipe_synthesize(c, cin, B, Tc, noise, test_inputs, out_samples, out_raw, caller_st);
        if (steps_per_graph <= 0) steps_per_graph = 32;
    }
    int rc0 = wn_synth_reserve(c);
    if (rc0) return rc0;
    const int L = c->L, R = c->R;
    Synth* s = c->synth;
    c->synth_path = 1;
    // everything below runs on the ctx-owned stream, ordered after the caller's stream and before its next op
    hipStream_t st = s->priv;
    WN_HIP(c, hipEventRecord(s->ev0, caller_st));
    WN_HIP(c, hipStreamWaitEvent(st, s->ev0, 0));
    s->B = B; s->T = T;
    c->fB = B; c->fT = T; c->fTc = Tc;
    // upsample the conditioning once for the whole utterance (wavenet.py:781-803); cbt[b*T+t][C]
    int rc = wn_upsample_fwd(c, nullptr, cin, B, Tc, st);
    if (rc) return rc;
    if ((rc = wn_gbias_fwd(c, B, st))) return rc;            // global conditioning of this batch (wavenet.py:766-777)
    for (int l = 0; l < L; ++l) WN_HIP(c, hipMemsetAsync(s->ring[l], 0, (size_t)(s->mask[l] + 1) * 32 * R * 2, st));
    const int mode = c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE ? 2 : (c->O == 2 ? 1 : 0);
    hipLaunchKernelGGL(wn_synth_init, dim3(1), dim3(256), 0, st, c->params_dev + c->first.dil_k, c->params_dev + c->first.dil_b, R, mode, 127,
                       s->ring[0], B, s->t_dev);
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
                rc = enqueue_step(c, s, noise, test_inputs, out_samples, out_raw, st);
                if (rc) { hipStreamEndCapture(st, &graph); return rc; }
            }
            WN_HIP(c, hipStreamEndCapture(st, &graph));
            WN_HIP(c, hipGraphInstantiate(&s->gexec, graph, nullptr, nullptr, 0));
            hipGraphDestroy(graph);
            s->g_steps = steps_per_graph; s->g_B = B; s->g_T = T; memcpy(s->g_key, key, sizeof key);
        }
        for (; done + steps_per_graph <= T; done += steps_per_graph) WN_HIP(c, hipGraphLaunch(s->gexec, st));
    }
    for (; done < T; ++done) { rc = enqueue_step(c, s, noise, test_inputs, out_samples, out_raw, st); if (rc) return rc; }
    WN_HIP(c, hipEventRecord(s->ev1, st));
    WN_HIP(c, hipStreamWaitEvent(caller_st, s->ev1, 0));
    return WN_OK;
}
