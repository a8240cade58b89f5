// fp32-accurate FORWARD of the WaveNet stack (wn_config.compute_dtype = WN_COMPUTE_F32_FWD; hparams mi355_compute_dtype = 'fp32').
//
// The reference computes in fp32 end to end (modules.py:306-320, 471-521; wavenet.py:650-721).  The production tile engine multiplies
// bf16 operands (BASELINE's prescribed training dtype), which leaves y_hat ~1e-2 rel-L2 from the fp32 arithmetic after 24 layers.  This
// translation unit is the other option: the same forward -- input conv, dropout -> dilated taps -> conditioning -> gate -> out / skip 1x1,
// skip sum, head -- with fp32 activations, fp32 weights read straight from the flat parameter buffer (TensorFlow [k][in][out] layouts:
// no packing) and fp32 FMA accumulation in a fixed k order.  It serves WaveNet.step / evaluation / the training-mode loss value; the
// BACKWARD stays on the bf16 engine's saved activations, so wn_train_bwd refuses a forward that ran here.  Not tuned: a 64 x 64 x 16
// LDS-tiled SGEMM on the vector ALU (the f32 MFMA rate on gfx950 equals the f32 vector rate: nothing to gain from the matrix pipe).
#include "wn_common.h"

struct F32State {
    float *X = nullptr, *U = nullptr, *Z = nullptr, *SK = nullptr, *H1 = nullptr, *C32 = nullptr;
    size_t bytes = 0;
};

struct SgemmArgs {
    const float* In; int32_t ld_in, col0;        // A rows: In[row + shift][col0 + k], zero outside the utterance
    int32_t shift;
    const float* W; int32_t ldw;                 // W[k][m] row-major (the TF kernel slice)
    int32_t K, M;
    float* Out; int32_t ld_out;
    int32_t accumulate;                          // Out = (accumulate ? Out : 0) + alpha * acc + bias + add, then * scale, then relu
    float alpha, scale;
    const float* bias; int32_t bias_bstride;     // bias[m] (+ utterance * bias_bstride: global conditioning)
    const float* add; int32_t ld_add;
    int32_t relu;
    int32_t out_bot;                             // 1: Out is [B][M][T] (y_hat layout) instead of [rows][ld_out]
    int32_t B, T;
    uint32_t key_lo, key_hi, thresh16; float keep_scale; int32_t drop_ld;      // thresh16 > 0: dropout mask on the A operand (modules.py:484)
};

#define SG_T 64
#define SG_K 16
__global__ __launch_bounds__(256) void wn_f32_sgemm_kernel(const SgemmArgs a) {
    __shared__ float As[SG_K][SG_T + 4];
    __shared__ float Bs[SG_K][SG_T + 4];
    const int tid = threadIdx.x;
    const int tiles_per_utt = (a.T + SG_T - 1) / SG_T;
    const int b = blockIdx.x / tiles_per_utt, t0 = (blockIdx.x - b * tiles_per_utt) * SG_T;
    const int m0 = blockIdx.y * SG_T;
    const int64_t rowbase = (int64_t)b * a.T;
    const int tr = tid >> 4, tc = tid & 15;        // this thread's 4 x 4 micro tile: rows tr*4.., columns tc*4..
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
    // staging assignment: A: row = tid >> 2, 4 consecutive k = (tid & 3) * 4;  B: k = tid >> 4, 4 consecutive m = (tid & 15) * 4
    const int ar = tid >> 2, ak = (tid & 3) * 4;
    const int bk = tid >> 4, bm = (tid & 15) * 4;
    const int ts = t0 + ar + a.shift;
    const bool a_ok = (t0 + ar < a.T) && ts >= 0 && ts < a.T;
    const int64_t arow = rowbase + ts;
    for (int k0 = 0; k0 < a.K; k0 += SG_K) {
        float av[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (a_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + ak + e;
                if (k < a.K) {
                    float v = a.In[arow * a.ld_in + a.col0 + k];
                    if (a.thresh16) {
                        const uint32_t el = (uint32_t)(arow * a.drop_ld + a.col0 + k);
                        const uint32_t w = wn_drop_word(a.key_lo, a.key_hi, el >> 1);
                        const uint32_t bits = (el & 1u) ? (w >> 16) : (w & 0xffffu);
                        v = bits >= a.thresh16 ? v * a.keep_scale : 0.0f;
                    }
                    av[e] = v;
                }
            }
        }
        float bv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (k0 + bk < a.K) {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (m0 + bm + e < a.M) bv[e] = a.W[(int64_t)(k0 + bk) * a.ldw + m0 + bm + e];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) { As[ak + e][ar] = av[e]; Bs[bk][bm + e] = bv[e]; }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < SG_K; ++kk) {
            float x[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = As[kk][tr * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = Bs[kk][tc * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(x[i], w[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + tr * 4 + i;
        if (t >= a.T) continue;
        const int64_t row = rowbase + t;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + tc * 4 + j;
            if (m >= a.M) continue;
            float* o = a.out_bot ? a.Out + ((int64_t)b * a.M + m) * a.T + t : a.Out + row * a.ld_out + m;
            float v = a.alpha * acc[i][j];
            if (a.accumulate) v += *o;
            if (a.bias) v += a.bias[(int64_t)b * a.bias_bstride + m];
            if (a.add) v += a.add[row * a.ld_add + m];
            v *= a.scale;
            if (a.relu) v = fmaxf(v, 0.0f);
            *o = v;
        }
    }
}

// h0[row][r] = W[cin][r] x + b[r]  (wavenet.py:705): scalar input or a row gather by class id
__global__ void wn_f32_first_conv(const void* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ X0,
                                  int64_t rows, int R, int is_ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * R) return;
    const int64_t row = i / R; const int r = (int)(i - row * R);
    X0[i] = is_ids ? W[(int64_t)((const int32_t*)x)[row] * R + r] + bias[r] : __builtin_fmaf(W[r], ((const float*)x)[row], bias[r]);
}
// u = tanh(z_a) * sigmoid(z_b)   (modules.py:510)
__global__ void wn_f32_gate(const float* __restrict__ Z, float* __restrict__ U, int64_t rows, int GH) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * GH) return;
    const int64_t row = i / GH; const int g = (int)(i - row * GH);
    const float za = Z[row * 2 * GH + g], zb = Z[row * 2 * GH + GH + g];
    U[i] = tanhf(za) * (1.0f / (1.0f + expf(-zb)));
}
// [B][C][T] -> [B*T][C]
__global__ void wn_f32_transpose_c(const float* __restrict__ cup, float* __restrict__ c32, int B, int C, int T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int t = (int)(i % T); const int64_t bc = i / T; const int cc = (int)(bc % C), b = (int)(bc / C);
    c32[((int64_t)b * T + t) * C + cc] = cup[i];
}
__global__ void wn_f32_bias_relu(float* __restrict__ v, const float* __restrict__ bias, int64_t rows, int S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * S) return;
    v[i] = fmaxf(v[i] + bias[i % S], 0.0f);
}

void wn_f32_free(wn_ctx* c) {
    F32State* s = (F32State*)c->f32;
    if (!s) return;
    for (float* p : {s->X, s->U, s->Z, s->SK, s->H1, s->C32}) if (p) hipFree(p);
    delete s; c->f32 = nullptr;
}
static int f32_reserve(wn_ctx* c) {
    if (c->f32) return WN_OK;
    F32State* s = new F32State(); c->f32 = s;
    const int64_t NT = c->NT;
    WN_HIP(c, hipMalloc((void**)&s->X, (size_t)c->L * NT * c->R * 4));
    WN_HIP(c, hipMalloc((void**)&s->U, (size_t)c->L * NT * c->GH * 4));
    WN_HIP(c, hipMalloc((void**)&s->Z, (size_t)NT * c->G * 4));
    WN_HIP(c, hipMalloc((void**)&s->SK, (size_t)NT * c->S * 4));
    WN_HIP(c, hipMalloc((void**)&s->H1, (size_t)NT * c->S * 4));
    WN_HIP(c, hipMalloc((void**)&s->C32, (size_t)NT * c->C * 4));
    s->bytes = (size_t)NT * 4 * ((size_t)c->L * (c->R + c->GH) + c->G + 2 * c->S + c->C);
    return WN_OK;
}
const float* wn_f32_debug(const wn_ctx* c, const char* name, int layer) {
    const F32State* s = (const F32State*)c->f32;
    if (!s) return nullptr;
    if (!strcmp(name, "X")) return s->X + (size_t)layer * c->NT * c->R;
    if (!strcmp(name, "U")) return s->U + (size_t)layer * c->NT * c->GH;
    return nullptr;
}

static int sgemm(wn_ctx* c, SgemmArgs& a, hipStream_t st) {
    a.B = c->fB; a.T = c->fT;
    dim3 grid(cdiv(a.T, SG_T) * a.B, cdiv(a.M, SG_T));
    hipLaunchKernelGGL(wn_f32_sgemm_kernel, grid, dim3(256), 0, st, a);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
static SgemmArgs mk(const float* In, int ld_in, int shift, const float* W, int ldw, int K, int M, float* Out, int ld_out) {
    SgemmArgs a; memset(&a, 0, sizeof a);
    a.In = In; a.ld_in = ld_in; a.shift = shift; a.W = W; a.ldw = ldw; a.K = K; a.M = M; a.Out = Out; a.ld_out = ld_out; a.alpha = 1.0f; a.scale = 1.0f;
    return a;
}

// wavenet.py:650-721 in fp32 on stream st: upsampled conditioning must already be in CUP[cup_final_idx] (wn_upsample_fwd, fp32 as ever)
int wn_f32_forward(wn_ctx* c, hipStream_t st) {
    int rc = f32_reserve(c);
    if (rc) return rc;
    F32State* s = (F32State*)c->f32;
    const int L = c->L, R = c->R, G = c->G, GH = c->GH, S = c->S, C = c->C, O = c->O, B = c->fB, T = c->fT;
    const int64_t NT = c->NT, rows = (int64_t)B * T;
    const float* P = c->params_dev;
    const int is_ids = c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE;
    hipLaunchKernelGGL(wn_f32_transpose_c, dim3(cdiv(rows * C, 256)), dim3(256), 0, st, c->CUP[c->cup_final_idx], s->C32, B, C, T);
    hipLaunchKernelGGL(wn_f32_first_conv, dim3(cdiv(rows * R, 256)), dim3(256), 0, st, c->fx, P + c->first.dil_k, P + c->first.dil_b, s->X, rows, R, is_ids);
    WN_LAUNCH_CHECK(c);
    const bool drop = c->cfg.dropout > 0.0f;
    for (int l = 0; l < L; ++l) {
        const int d = c->dil[l];
        const float* Xl = s->X + (size_t)l * NT * R;
        float* Ul = s->U + (size_t)l * NT * GH;
        for (int tap = 0; tap < 3; ++tap) {      // z = b + sum_taps drop(x)(t - (2 - tap) d) W_tap    (modules.py:484-494; kernel index 2 is the current sample)
            SgemmArgs a = mk(Xl, R, -(2 - tap) * d, P + c->lay[l].dil_k + (int64_t)tap * R * G, G, R, G, s->Z, G);
            a.accumulate = tap > 0;
            if (tap == 0) {
                if (c->gin > 0) { a.bias = c->gbias + (size_t)l * B * G; a.bias_bstride = G; }      // b_dil + b_cin + W_g^T g + b_g per utterance
                else a.bias = c->b1sum + (size_t)l * G;
            }
            if (drop) { wn_layer_key(c->fseed, l, &a.key_lo, &a.key_hi); a.thresh16 = (uint32_t)lrintf(c->cfg.dropout * 65536.0f); a.keep_scale = 1.0f / (1.0f - c->cfg.dropout); a.drop_ld = R; }
            if ((rc = sgemm(c, a, st))) return rc;
        }
        {   // + W_cin c   (modules.py:497-501)
            SgemmArgs a = mk(s->C32, C, 0, P + c->lay[l].cin_k, G, C, G, s->Z, G); a.accumulate = 1;
            if ((rc = sgemm(c, a, st))) return rc;
        }
        hipLaunchKernelGGL(wn_f32_gate, dim3(cdiv(rows * GH, 256)), dim3(256), 0, st, s->Z, Ul, rows, GH);
        {   // skip sum (wavenet.py:706-715 unrolled: the legacy factors are folded into skip_scale)
            SgemmArgs a = mk(Ul, GH, 0, P + c->lay[l].skip_k, S, GH, S, s->SK, S); a.accumulate = l > 0; a.alpha = c->skip_scale[l];
            if ((rc = sgemm(c, a, st))) return rc;
        }
        if (l + 1 < L) {   // x_{l+1} = (W_out u + b + x_l) * rho   (modules.py:515-520)
            SgemmArgs a = mk(Ul, GH, 0, P + c->lay[l].out_k, R, GH, R, s->X + (size_t)(l + 1) * NT * R, R);
            a.bias = P + c->lay[l].out_b; a.add = Xl; a.ld_add = R; a.scale = c->res_scale;
            if ((rc = sgemm(c, a, st))) return rc;
        }
    }
    hipLaunchKernelGGL(wn_f32_bias_relu, dim3(cdiv(rows * S, 256)), dim3(256), 0, st, s->SK, c->skip_bias_total, rows, S);      // ReLU(skips)  (wavenet.py:716-719)
    {   // final_convolution_1 + ReLU
        SgemmArgs a = mk(s->SK, S, 0, P + c->fin1_k, S, S, S, s->H1, S); a.bias = P + c->fin1_b; a.relu = 1;
        if ((rc = sgemm(c, a, st))) return rc;
    }
    {   // final_convolution_2 -> y_hat [B][O][T]
        SgemmArgs a = mk(s->H1, S, 0, P + c->fin2_k, O, S, O, c->YHAT, 0); a.bias = P + c->fin2_b; a.out_bot = 1;
        if ((rc = sgemm(c, a, st))) return rc;
    }
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
