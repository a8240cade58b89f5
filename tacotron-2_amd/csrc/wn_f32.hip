// fp32 training mode: forward AND backward of the WaveNet stack in the reference's arithmetic (wn_config.compute_dtype = WN_COMPUTE_F32; hparams mi355_compute_dtype = 'fp32').
//
// The reference computes in fp32 end to end (modules.py:306-320, 471-521; wavenet.py:650-721).  The production tile engine multiplies
// bf16 operands (BASELINE's prescribed training dtype), which leaves y_hat ~1e-2 rel-L2 from the fp32 arithmetic after 24 layers.  This
// translation unit is the other option: the same forward -- input conv, dropout -> dilated taps -> conditioning -> gate -> out / skip 1x1,
// skip sum, head -- with fp32 activations, fp32 weights read straight from the flat parameter buffer (TensorFlow [k][in][out] layouts:
// no packing) and fp32 FMA accumulation in a fixed k order, AND its backward (replaces tf.gradients, wavenet.py:557, in the same
// arithmetic): every data gradient as the same SGEMM with transposed weight strides, every weight gradient as a time contraction with
// ordered partial sums (no atomics), the gate derivative from the saved fp32 pre-activations.  It serves WaveNet.step / evaluation and a
// complete fp32 TRAINING step (wn_train_fwd + wn_train_bwd + the shared optimiser).  128 x 128 x 16 LDS-tiled SGEMMs on the fp32 matrix
// instruction (v_mfma_f32_32x32x2_f32: the vector FMA's peak with 1/64 of its instruction count); 128 ms per C2 step (56 TFLOP/s algorithmic, profiles/r4w_bench_c2_fp32.json).
#include "wn_common.h"

struct F32State {
    float *X = nullptr, *U = nullptr, *Z = nullptr, *SK = nullptr, *H1 = nullptr, *C32 = nullptr;      // forward: X [L][NT][R], U [L][NT][GH], Z [L][NT][G] (pre-activations)
    float *DY = nullptr, *DH1 = nullptr, *DSK = nullptr, *GU = nullptr, *DZ = nullptr, *GX[2] = {nullptr, nullptr}, *DC = nullptr, *DCT = nullptr, *PART = nullptr;      // backward (lazy)
    size_t part_floats = 0;
    size_t bytes = 0;
};

struct SgemmArgs {
    const float* In; int32_t ld_in, col0;        // A rows: In[row + shift][col0 + k], zero outside the utterance
    int32_t shift;
    const float* W; int32_t ldw;                 // W[k][m] row-major (the TF kernel slice)
    int32_t K, M;
    float* Out; int32_t ld_out;
    int32_t wk, wm;                              // element strides of W along k / m (wk = ldw, wm = 1 unless transposed: the data gradients)
    int32_t accumulate;                          // v = alpha * acc [dropout mask on the OUTPUT element]; Out = ((accumulate ? Out : 0) + v + bias + add) * scale, relu, * (mask > 0)
    const float* mask; int32_t ld_mask;          // ReLU mask operand (the forward activation)
    int32_t drop_on_out;                         // the dropout keys below mask the OUTPUT element (row, m) instead of the A operand (gradient wrt a dropped input)
    float alpha, scale;
    const float* bias; int32_t bias_bstride;     // bias[m] (+ utterance * bias_bstride: global conditioning)
    const float* add; int32_t ld_add;
    int32_t relu;
    int32_t out_bot;                             // 1: Out is [B][M][T] (y_hat layout) instead of [rows][ld_out]
    int32_t B, T;
    uint32_t key_lo, key_hi, thresh16; float keep_scale; int32_t drop_ld;      // thresh16 > 0: dropout mask on the A operand (modules.py:484)
};

#define SG_T 128          // workgroup tile: 128 time rows x 128 output channels, 4 waves of 64 x 64 (2 x 2 MFMA tiles of 32 x 32)
#define SG_K 16
// v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation, in a fixed order.  A operand: lane l holds A[l % 32][l / 32]; B operand
// B[l / 32][l % 32]; accumulator register r of lane l is element (8 (r / 4) + 4 (l / 32) + r % 4, l % 32).  Same peak as the vector FMA
// (157 TFLOP/s) with 1/64 of the instruction count: the loop below is 4 ds_read_b32 per 4 MFMAs.
__global__ __launch_bounds__(256) void wn_f32_sgemm_kernel(const SgemmArgs a) {
    __shared__ float As[SG_K][SG_T + 4];       // [k][row]
    __shared__ float Bs[SG_K][SG_T + 4];       // [k][m]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_per_utt = (a.T + SG_T - 1) / SG_T;
    const int b = blockIdx.x / tiles_per_utt, t0 = (blockIdx.x - b * tiles_per_utt) * SG_T;
    const int m0 = blockIdx.y * SG_T;
    const int64_t rowbase = (int64_t)b * a.T;
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // staging assignment: A: rows ar and ar + 64, 4 consecutive k = (tid & 3) * 4;  B: k = tid >> 4, 8 consecutive m = (tid & 15) * 8
    const int ar = tid >> 2, ak = (tid & 3) * 4;
    const int bk = tid >> 4, bm = (tid & 15) * 8;
    // 16-B loads where the operand allows it (row pitches / offsets multiples of 4 floats; transposed weights are read along their pitch: scalar)
    const bool vecA = (a.ld_in % 4 == 0) && (a.col0 % 4 == 0) && (a.drop_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.In) & 15) == 0);
    const bool vecB = (a.wm == 1) && (a.wk % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.W) & 15) == 0);
    const bool vecBT = (a.wk == 1) && (a.wm % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.W) & 15) == 0);
    float av[2][4], bv[8];
    const int tm = tid >> 1, tk = (tid & 1) * 8;      // transposed weights (data gradients: wk == 1): 8 consecutive k of output channel m0 + tm
    auto load_chunk = [&](const int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int tl = t0 + ar + 64 * h, ts = tl + a.shift;
            const bool a_ok = (tl < a.T) && ts >= 0 && ts < a.T;
            const int64_t arow = rowbase + ts;
            const int kb = k0 + ak;
            if (a_ok && vecA && kb + 4 <= a.K) {      // one 16-B load, one dropout word per element pair (wn_drop_word: 2 x 16 bits)
                const float4 q = *reinterpret_cast<const float4*>(a.In + arow * a.ld_in + a.col0 + kb);
                av[h][0] = q.x; av[h][1] = q.y; av[h][2] = q.z; av[h][3] = q.w;
                if (a.thresh16 && !a.drop_on_out) {
                    const uint32_t el = (uint32_t)(arow * a.drop_ld + a.col0 + kb);          // multiple of 4
                    const uint32_t w0 = wn_drop_word(a.key_lo, a.key_hi, el >> 1), w1 = wn_drop_word(a.key_lo, a.key_hi, (el >> 1) + 1);
                    av[h][0] = (w0 & 0xffffu) >= a.thresh16 ? av[h][0] * a.keep_scale : 0.0f;
                    av[h][1] = (w0 >> 16) >= a.thresh16 ? av[h][1] * a.keep_scale : 0.0f;
                    av[h][2] = (w1 & 0xffffu) >= a.thresh16 ? av[h][2] * a.keep_scale : 0.0f;
                    av[h][3] = (w1 >> 16) >= a.thresh16 ? av[h][3] * a.keep_scale : 0.0f;
                }
            } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kb + e;
                float v = 0.0f;
                if (a_ok && k < a.K) {
                    v = a.In[arow * a.ld_in + a.col0 + k];
                    if (a.thresh16 && !a.drop_on_out) {
                        const uint32_t el = (uint32_t)(arow * a.drop_ld + a.col0 + k);
                        const uint32_t w = wn_drop_word(a.key_lo, a.key_hi, el >> 1);
                        const uint32_t bits = (el & 1u) ? (w >> 16) : (w & 0xffffu);
                        v = bits >= a.thresh16 ? v * a.keep_scale : 0.0f;
                    }
                }
                av[h][e] = v;
            }
            }
        }
        if (vecBT) {
            if (m0 + tm < a.M && k0 + tk + 8 <= a.K) {
                const float4* wp = reinterpret_cast<const float4*>(a.W + (int64_t)(m0 + tm) * a.wm + k0 + tk);
                const float4 q0 = wp[0], q1 = wp[1];
                bv[0] = q0.x; bv[1] = q0.y; bv[2] = q0.z; bv[3] = q0.w; bv[4] = q1.x; bv[5] = q1.y; bv[6] = q1.z; bv[7] = q1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) bv[e] = (m0 + tm < a.M && k0 + tk + e < a.K) ? a.W[(int64_t)(m0 + tm) * a.wm + k0 + tk + e] : 0.0f;
            }
        } else
        if (vecB && k0 + bk < a.K && m0 + bm + 8 <= a.M) {
            const float4* wp = reinterpret_cast<const float4*>(a.W + (int64_t)(k0 + bk) * a.wk + m0 + bm);
            const float4 q0 = wp[0], q1 = wp[1];
            bv[0] = q0.x; bv[1] = q0.y; bv[2] = q0.z; bv[3] = q0.w; bv[4] = q1.x; bv[5] = q1.y; bv[6] = q1.z; bv[7] = q1.w;
        } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (k0 + bk < a.K && m0 + bm + e < a.M) ? a.W[(int64_t)(k0 + bk) * a.wk + (int64_t)(m0 + bm + e) * a.wm] : 0.0f;
        }
    };
    // the global loads of chunk k + 1 are in flight while chunk k is multiplied (registers av / bv)
    load_chunk(0);
    for (int k0 = 0; k0 < a.K; k0 += SG_K) {
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) As[ak + e][ar + 64 * h] = av[h][e];
        if (vecBT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) Bs[tk + e][tm] = bv[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) Bs[bk][bm + e] = bv[e];
        }
        __syncthreads();
        if (k0 + SG_K < a.K) load_chunk(k0 + SG_K);
#pragma unroll
        for (int kk = 0; kk < SG_K; kk += 2) {
            const int kr = kk + (lane >> 5);
            const float a0 = As[kr][wr * 64 + (lane & 31)], a1 = As[kr][wr * 64 + 32 + (lane & 31)];
            const float b0 = Bs[kr][wc * 64 + (lane & 31)], b1 = Bs[kr][wc * 64 + 32 + (lane & 31)];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int t = t0 + wr * 64 + i * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
        if (t >= a.T) continue;
        const int64_t row = rowbase + t;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wc * 64 + j * 32 + (lane & 31);
            if (m >= a.M) continue;
            float* o = a.out_bot ? a.Out + ((int64_t)b * a.M + m) * a.T + t : a.Out + row * a.ld_out + m;
            float v = a.alpha * acc[i][j][r];
            if (a.thresh16 && a.drop_on_out) {
                const uint32_t el = (uint32_t)(row * a.drop_ld + m);
                const uint32_t w = wn_drop_word(a.key_lo, a.key_hi, el >> 1);
                const uint32_t bits = (el & 1u) ? (w >> 16) : (w & 0xffffu);
                v = bits >= a.thresh16 ? v * a.keep_scale : 0.0f;
            }
            if (a.accumulate) v += *o;
            if (a.bias) v += a.bias[(int64_t)b * a.bias_bstride + m];
            if (a.add) v += a.add[row * a.ld_add + m];
            v *= a.scale;
            if (a.relu) v = fmaxf(v, 0.0f);
            if (a.mask && !(a.mask[row * a.ld_mask + m] > 0.0f)) v = 0.0f;
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
    for (float* p : {s->X, s->U, s->Z, s->SK, s->H1, s->C32, s->DY, s->DH1, s->DSK, s->GU, s->DZ, s->GX[0], s->GX[1], s->DC, s->DCT, s->PART}) if (p) hipFree(p);
    delete s; c->f32 = nullptr;
}
static int f32_reserve(wn_ctx* c) {
    if (!c->f32) c->f32 = new F32State();
    F32State* s = (F32State*)c->f32;
    const int64_t NT = c->NT;
    // (each buffer on its own: after a failed allocation the next call retries the missing ones only)
    auto need = [&](float** p, size_t floats) -> int { if (!*p) WN_HIP(c, hipMalloc((void**)p, floats * 4)); return WN_OK; };
    int rc;
    if ((rc = need(&s->X, (size_t)c->L * NT * c->R)) || (rc = need(&s->U, (size_t)c->L * NT * c->GH)) ||
        (rc = need(&s->Z, (size_t)c->L * NT * c->G)) ||      // every layer's pre-activations: the gate derivative of the backward
        (rc = need(&s->SK, (size_t)NT * c->S)) || (rc = need(&s->H1, (size_t)NT * c->S)) || (rc = need(&s->C32, (size_t)NT * c->C))) return rc;
    s->bytes = (size_t)NT * 4 * ((size_t)c->L * (c->R + c->GH + c->G) + 2 * c->S + c->C);
    return WN_OK;
}
const float* wn_f32_debug(const wn_ctx* c, const char* name, int layer) {
    const F32State* s = (const F32State*)c->f32;
    if (!s || !s->X || !s->U) return nullptr;
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
    a.In = In; a.ld_in = ld_in; a.shift = shift; a.W = W; a.ldw = ldw; a.wk = ldw; a.wm = 1; a.K = K; a.M = M; a.Out = Out; a.ld_out = ld_out; a.alpha = 1.0f; a.scale = 1.0f;
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
        float* Zl = s->Z + (size_t)l * NT * G;
        for (int tap = 0; tap < 3; ++tap) {      // z = b + sum_taps drop(x)(t - (2 - tap) d) W_tap    (modules.py:484-494; kernel index 2 is the current sample)
            SgemmArgs a = mk(Xl, R, -(2 - tap) * d, P + c->lay[l].dil_k + (int64_t)tap * R * G, G, R, G, Zl, G);
            a.accumulate = tap > 0;
            if (tap == 0) {
                if (c->gin > 0) { a.bias = c->gbias + (size_t)l * B * G; a.bias_bstride = G; }      // b_dil + b_cin + W_g^T g + b_g per utterance
                else a.bias = c->b1sum + (size_t)l * G;
            }
            if (drop) { wn_layer_key(c->fseed, l, &a.key_lo, &a.key_hi); a.thresh16 = (uint32_t)lrintf(c->cfg.dropout * 65536.0f); a.keep_scale = 1.0f / (1.0f - c->cfg.dropout); a.drop_ld = R; }
            if ((rc = sgemm(c, a, st))) return rc;
        }
        {   // + W_cin c   (modules.py:497-501)
            SgemmArgs a = mk(s->C32, C, 0, P + c->lay[l].cin_k, G, C, G, Zl, G); a.accumulate = 1;
            if ((rc = sgemm(c, a, st))) return rc;
        }
        hipLaunchKernelGGL(wn_f32_gate, dim3(cdiv(rows * GH, 256)), dim3(256), 0, st, Zl, Ul, rows, GH);
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

// =============================================================================================== fp32 backward
// Weight gradient: dW[k][m] = alpha * sum_rows A[row + shift][k] * Bm[row][m]  (contraction over TIME), per row slab into `part`
// [slab][K][M]; wn_f32_wgrad_reduce sums the slabs in order into the TF-layout gradient tensor.  a_ones: A = 1 (K = 1: column sums = bias
// gradients).  Dropout keys: the A operand is the dropout-applied layer input (modules.py:484).
struct WgradF32Args {
    const float* A; int32_t lda, shift, a_ones;          // a_ones: A = 1 (K = 1);  ones_row: one extra A column of ones at k = K - 1 (the bias gradient falls out as the last row)
    int32_t ones_row;
    const float* Bm; int32_t ldb;
    int32_t K, M;
    float* part;
    int32_t B, T, slab, slabs_per_utt;
    uint32_t key_lo, key_hi, thresh16; float keep_scale; int32_t drop_ld;
};
__global__ __launch_bounds__(256) void wn_f32_wgrad_kernel(const WgradF32Args a) {
    __shared__ float As[SG_K][SG_T + 4];       // [row in chunk][k]
    __shared__ float Bs[SG_K][SG_T + 4];       // [row in chunk][m]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;       // this wave's 64 x 64 piece of the 128 (k) x 128 (m) tile
    const int k0 = blockIdx.x * SG_T, m0 = blockIdx.y * SG_T;
    const int b = blockIdx.z / a.slabs_per_utt, sl = blockIdx.z - b * a.slabs_per_utt;
    const int t_lo = sl * a.slab, t_hi = min(a.T, t_lo + a.slab);
    const int64_t rowbase = (int64_t)b * a.T;
    const int lr = tid >> 4, lc = (tid & 15) * 8;  // staging: row lr of the chunk, 8 consecutive columns
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    float av[8], bv[8];
    const bool vecA = !a.a_ones && (a.lda % 4 == 0) && (a.drop_ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
    const bool vecB = (a.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.Bm) & 15) == 0);
    auto load_chunk = [&](const int tc0) {
        const int t = tc0 + lr;
#pragma unroll
        for (int e = 0; e < 8; ++e) { av[e] = 0.0f; bv[e] = 0.0f; }
        if (t < t_hi) {
            const int ts = t + a.shift;
            if (a.a_ones) { if (lc == 0) av[0] = 1.0f; }
            else if (ts >= 0 && ts < a.T) {
                if (vecA && k0 + lc + 8 <= a.K - a.ones_row) {
                    const float4* ap = reinterpret_cast<const float4*>(a.A + (rowbase + ts) * a.lda + k0 + lc);
                    const float4 q0 = ap[0], q1 = ap[1];
                    av[0] = q0.x; av[1] = q0.y; av[2] = q0.z; av[3] = q0.w; av[4] = q1.x; av[5] = q1.y; av[6] = q1.z; av[7] = q1.w;
                    if (a.thresh16) {      // one dropout word per element pair
                        const uint32_t el = (uint32_t)((rowbase + ts) * a.drop_ld + k0 + lc);          // multiple of 8
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const uint32_t w = wn_drop_word(a.key_lo, a.key_hi, (el >> 1) + q);
                            av[2 * q] = (w & 0xffffu) >= a.thresh16 ? av[2 * q] * a.keep_scale : 0.0f;
                            av[2 * q + 1] = (w >> 16) >= a.thresh16 ? av[2 * q + 1] * a.keep_scale : 0.0f;
                        }
                    }
                } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = k0 + lc + e;
                    if (k < a.K - a.ones_row) {
                        float v = a.A[(rowbase + ts) * a.lda + k];
                        if (a.thresh16) {
                            const uint32_t el = (uint32_t)((rowbase + ts) * a.drop_ld + k);
                            const uint32_t w = wn_drop_word(a.key_lo, a.key_hi, el >> 1);
                            const uint32_t bits = (el & 1u) ? (w >> 16) : (w & 0xffffu);
                            v = bits >= a.thresh16 ? v * a.keep_scale : 0.0f;
                        }
                        av[e] = v;
                    }
                }
                }
            }
            if (a.ones_row && a.K - 1 >= k0 + lc && a.K - 1 < k0 + lc + 8) {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (k0 + lc + e == a.K - 1) av[e] = 1.0f;
            }
            if (vecB && m0 + lc + 8 <= a.M) {
                const float4* bp = reinterpret_cast<const float4*>(a.Bm + (rowbase + t) * a.ldb + m0 + lc);
                const float4 q0 = bp[0], q1 = bp[1];
                bv[0] = q0.x; bv[1] = q0.y; bv[2] = q0.z; bv[3] = q0.w; bv[4] = q1.x; bv[5] = q1.y; bv[6] = q1.z; bv[7] = q1.w;
            } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (m0 + lc + e < a.M) bv[e] = a.Bm[(rowbase + t) * a.ldb + m0 + lc + e];
            }
        }
    };
    load_chunk(t_lo);
    for (int tc0 = t_lo; tc0 < t_hi; tc0 += SG_K) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) { As[lr][lc + e] = av[e]; Bs[lr][lc + e] = bv[e]; }
        __syncthreads();
        if (tc0 + SG_K < t_hi) load_chunk(tc0 + SG_K);      // in flight while this chunk is multiplied
#pragma unroll
        for (int rr = 0; rr < SG_K; rr += 2) {      // D[k][m] += sum_r A[r][k] B[r][m]: the MFMA's A operand is A^T (lane l: k = l % 32, row = l / 32)
            const int r = rr + (lane >> 5);
            const float a0 = As[r][wr * 64 + (lane & 31)], a1 = As[r][wr * 64 + 32 + (lane & 31)];
            const float b0 = Bs[r][wc * 64 + (lane & 31)], b1 = Bs[r][wc * 64 + 32 + (lane & 31)];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    float* P = a.part + (int64_t)blockIdx.z * a.K * a.M;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = k0 + wr * 64 + i * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
        if (k >= a.K) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int m = m0 + wc * 64 + j * 32 + (lane & 31); if (m < a.M) P[(int64_t)k * a.M + m] = acc[i][j][r]; }
    }
}
__global__ void wn_f32_wgrad_reduce(const float* __restrict__ part, int nslab, int K, int M, float* __restrict__ out, int ldo, float alpha,
                                    int ones_row, float* __restrict__ bias_out, float* __restrict__ bias_out2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)K * M) return;
    float s = 0.0f;
    for (int z = 0; z < nslab; ++z) s += part[(int64_t)z * K * M + i];
    const int k = (int)(i / M), m = (int)(i - (int64_t)k * M);
    if (ones_row && k == K - 1) {
        if (bias_out) bias_out[m] = alpha * s;
        if (bias_out2) bias_out2[m] = alpha * s;
    } else out[(int64_t)k * ldo + m] = alpha * s;
}
// d z from d u and the saved pre-activations (modules.py:510 differentiated): da = g s (1 - t^2), db = g t s (1 - s)
__global__ void wn_f32_gate_bwd(const float* __restrict__ GU, const float* __restrict__ Z, float* __restrict__ DZ, int64_t rows, int GH) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * GH) return;
    const int64_t row = i / GH; const int g = (int)(i - row * GH);
    const float t = tanhf(Z[row * 2 * GH + g]), s = 1.0f / (1.0f + expf(-Z[row * 2 * GH + GH + g])), gu = GU[i];
    DZ[row * 2 * GH + g] = gu * s * (1.0f - t * t);
    DZ[row * 2 * GH + GH + g] = gu * t * s * (1.0f - s);
}
// [B*T][C] -> [B][C][T]
__global__ void wn_f32_transpose_back(const float* __restrict__ c32, float* __restrict__ cbt, int B, int C, int T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * C * T) return;
    const int t = (int)(i % T); const int64_t bc = i / T; const int cc = (int)(bc % C), b = (int)(bc / C);
    cbt[i] = c32[((int64_t)b * T + t) * C + cc];
}
// one-hot input convolution: d W[id][r] += g0[row][r]  (mu-law-quantize models; float atomics: a row scatter)
__global__ void wn_f32_first_conv_bwd_ids(const int32_t* __restrict__ ids, const float* __restrict__ g0, float* __restrict__ dW, int64_t rows, int R) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * R) return;
    const int64_t row = i / R; const int r = (int)(i - row * R);
    unsafeAtomicAdd(&dW[(int64_t)ids[row] * R + r], g0[i]);
}

// colsum[b][g] = sum_t dz[b, t, g]  (global conditioning: the per-utterance bias gradient, modules.py:499-508); 64 columns x 4 time lanes, ordered
__global__ __launch_bounds__(256) void wn_f32_colsum_utt(const float* __restrict__ DZ, float* __restrict__ colsum, int T, int G) {
    const int g = blockIdx.x * 64 + (threadIdx.x & 63), b = blockIdx.y, q = threadIdx.x >> 6;
    __shared__ float red[4][64];
    float a = 0.0f;
    if (g < G) for (int t = q; t < T; t += 4) a += DZ[((int64_t)b * T + t) * G + g];
    red[q][threadIdx.x & 63] = a;
    __syncthreads();
    if (q == 0 && g < G) colsum[(int64_t)b * G + g] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

float* wn_f32_dy(wn_ctx* c) {
    F32State* s = (F32State*)c->f32;
    if (!s) return nullptr;
    if (!s->DY) {
        const int ldDY = (c->O + 15) / 16 * 16;
        if (hipMalloc((void**)&s->DY, (size_t)c->NT * ldDY * 4) != hipSuccess) return nullptr;
    }
    return s->DY;
}
static int f32_reserve_bwd(wn_ctx* c) {
    F32State* s = (F32State*)c->f32;
    const int64_t NT = c->NT;
    // each buffer on its own: a failed allocation leaves the others to be reused (or freed by wn_f32_free), never leaked by a retry
    auto need = [&](float** p, size_t floats) -> int { if (!*p) WN_HIP(c, hipMalloc((void**)p, floats * 4)); return WN_OK; };
    int rc;
    if ((rc = need(&s->DH1, (size_t)NT * c->S)) || (rc = need(&s->DSK, (size_t)NT * c->S)) || (rc = need(&s->GU, (size_t)NT * c->GH)) ||
        (rc = need(&s->DZ, (size_t)NT * c->G)) || (rc = need(&s->GX[0], (size_t)NT * c->R)) || (rc = need(&s->GX[1], (size_t)NT * c->R)) ||
        (rc = need(&s->DC, (size_t)NT * c->C)) || (rc = need(&s->DCT, (size_t)NT * c->C))) return rc;
    if (!s->PART) {
        const int maxk = std::max(std::max(c->R, c->GH), std::max(c->S, c->C)), maxm = std::max(std::max(c->G, c->S), std::max(c->R, c->O));
        const int slabs = c->maxB * cdiv(c->maxT, 2048);
        const size_t floats = (size_t)slabs * (maxk + 1) * maxm;      // + the bias row
        if ((rc = need(&s->PART, floats))) return rc;
        s->part_floats = floats;
    }
    return WN_OK;
}
// dW (TF layout, row pitch ldo) = alpha * A^T Bm over all rows; bias_out (optional): alpha * column sums of Bm from one extra A column of
// ones in the same launch (bias_out2: the twin bias of the gate pre-activation).  A == nullptr: only the column sums.
static int wgrad32(wn_ctx* c, const float* A, int lda, int shift, int K, const float* Bm, int ldb, int M, float* out, int ldo, float alpha,
                   int drop_layer, float* bias_out, float* bias_out2, hipStream_t st) {
    F32State* s = (F32State*)c->f32;
    WgradF32Args a; memset(&a, 0, sizeof a);
    a.A = A; a.lda = lda; a.shift = shift; a.a_ones = A == nullptr; a.Bm = Bm; a.ldb = ldb; a.M = M; a.part = s->PART;
    a.ones_row = (A != nullptr && (bias_out || bias_out2)) ? 1 : 0;
    a.K = (A ? K : 1) + a.ones_row;
    a.B = c->fB; a.T = c->fT; a.slab = 2048; a.slabs_per_utt = cdiv(a.T, a.slab);
    if (drop_layer >= 0 && c->cfg.dropout > 0.0f) {
        wn_layer_key(c->fseed, drop_layer, &a.key_lo, &a.key_hi); a.thresh16 = (uint32_t)lrintf(c->cfg.dropout * 65536.0f);
        a.keep_scale = 1.0f / (1.0f - c->cfg.dropout); a.drop_ld = lda;
    }
    const int nslab = a.B * a.slabs_per_utt;
    if ((size_t)nslab * a.K * M > s->part_floats) WN_FAIL(c, WN_E_STATE, "fp32 weight-gradient partial buffer too small");
    hipLaunchKernelGGL(wn_f32_wgrad_kernel, dim3(cdiv(a.K, SG_T), cdiv(M, SG_T), nslab), dim3(256), 0, st, a);
    if (a.a_ones)      // column sums only: the single row goes to the bias target(s)
        hipLaunchKernelGGL(wn_f32_wgrad_reduce, dim3(cdiv((int64_t)M, 256)), dim3(256), 0, st, s->PART, nslab, 1, M, (float*)nullptr, 0, alpha, 1, bias_out, bias_out2);
    else
        hipLaunchKernelGGL(wn_f32_wgrad_reduce, dim3(cdiv((int64_t)a.K * M, 256)), dim3(256), 0, st, s->PART, nslab, a.K, M, out, ldo, alpha, a.ones_row, bias_out, bias_out2);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}

// Backward of the last fp32 forward into the flat gradient buffer (effective-parameter layout): heads, then every layer top to bottom
// with its weight gradients computed on the spot, d c_up accumulated over the layers, input conv, upsample net (the fp32 kernels of
// wn_frontend.hip).  Replaces optimizer.compute_gradients (wavenet.py:557) in the reference's own arithmetic.
int wn_f32_backward(wn_ctx* c, float* grads, hipStream_t st) {
    F32State* s = (F32State*)c->f32;
    if (!s || !s->DY) WN_FAIL(c, WN_E_STATE, "fp32 backward without an fp32 forward that computed the loss");
    int rc = f32_reserve_bwd(c);
    if (rc) return rc;
    const int L = c->L, R = c->R, G = c->G, GH = c->GH, S = c->S, C = c->C, O = c->O, B = c->fB, T = c->fT;
    const int ldDY = (O + 15) / 16 * 16;
    const int64_t NT = c->NT, rows = (int64_t)B * T;
    const float* P = c->params_dev;
    // ---- head (wavenet.py:716-721 backwards)
    if ((rc = wgrad32(c, s->H1, S, 0, S, s->DY, ldDY, O, grads + c->fin2_k, O, 1.0f, -1, grads + c->fin2_b, nullptr, st))) return rc;
    {   // d pre1 = (dY W2^T) * (h1 > 0)
        SgemmArgs a = mk(s->DY, ldDY, 0, P + c->fin2_k, O, O, S, s->DH1, S); a.wk = 1; a.wm = O; a.mask = s->H1; a.ld_mask = S;
        if ((rc = sgemm(c, a, st))) return rc;
    }
    if ((rc = wgrad32(c, s->SK, S, 0, S, s->DH1, S, S, grads + c->fin1_k, S, 1.0f, -1, grads + c->fin1_b, nullptr, st))) return rc;
    {   // d skips = (d pre1 W1^T) * (relu(skips) > 0)
        SgemmArgs a = mk(s->DH1, S, 0, P + c->fin1_k, S, S, S, s->DSK, S); a.wk = 1; a.wm = S; a.mask = s->SK; a.ld_mask = S;
        if ((rc = sgemm(c, a, st))) return rc;
    }
    WN_HIP(c, hipMemsetAsync(s->DC, 0, (size_t)rows * C * 4, st));
    for (int l = L - 1; l >= 0; --l) {
        const int d = c->dil[l];
        const float* Xl = s->X + (size_t)l * NT * R;
        const float* Ul = s->U + (size_t)l * NT * GH;
        const float* Zl = s->Z + (size_t)l * NT * G;
        const bool top = (l == L - 1);
        float* gx_up = s->GX[(l + 1) & 1]; float* gx_dn = s->GX[l & 1];
        {   // d u = c_l d skips W_skip^T (+ rho-scaled d h_{l+1} W_out^T)   (modules.py:512-520 backwards)
            SgemmArgs a = mk(s->DSK, S, 0, P + c->lay[l].skip_k, S, S, GH, s->GU, GH); a.wk = 1; a.wm = S; a.alpha = c->skip_scale[l];
            if ((rc = sgemm(c, a, st))) return rc;
            if (!top) {
                SgemmArgs o = mk(gx_up, R, 0, P + c->lay[l].out_k, R, R, GH, s->GU, GH); o.wk = 1; o.wm = R; o.accumulate = 1;
                if ((rc = sgemm(c, o, st))) return rc;
            }
        }
        hipLaunchKernelGGL(wn_f32_gate_bwd, dim3(cdiv(rows * GH, 256)), dim3(256), 0, st, s->GU, Zl, s->DZ, rows, GH);
        // ---- weight gradients of this layer (bias gradients ride along as a ones column of the A operand)
        if ((rc = wgrad32(c, Ul, GH, 0, GH, s->DSK, S, S, grads + c->lay[l].skip_k, S, c->skip_scale[l], -1, c->lbias ? grads + c->lay[l].skip_b : nullptr, nullptr, st))) return rc;
        if (!top && (rc = wgrad32(c, Ul, GH, 0, GH, gx_up, R, R, grads + c->lay[l].out_k, R, 1.0f, -1, c->lbias ? grads + c->lay[l].out_b : nullptr, nullptr, st))) return rc;
        for (int tap = 0; tap < 3; ++tap)
            if ((rc = wgrad32(c, Xl, R, -(2 - tap) * d, R, s->DZ, G, G, grads + c->lay[l].dil_k + (int64_t)tap * R * G, G, 1.0f, l, nullptr, nullptr, st))) return rc;
        if ((rc = wgrad32(c, s->C32, C, 0, C, s->DZ, G, G, grads + c->lay[l].cin_k, G, 1.0f, -1, c->lbias ? grads + c->lay[l].dil_b : nullptr, c->lbias ? grads + c->lay[l].cin_b : nullptr, st))) return rc;
        if (c->gin > 0) hipLaunchKernelGGL(wn_f32_colsum_utt, dim3(cdiv(G, 64), B), dim3(256), 0, st, s->DZ, c->colsum + (size_t)l * B * G, T, G);
        {   // d c_up += d z W_cin^T   (modules.py:497-501 backwards)
            SgemmArgs a = mk(s->DZ, G, 0, P + c->lay[l].cin_k, G, G, C, s->DC, C); a.wk = 1; a.wm = G; a.accumulate = 1;
            if ((rc = sgemm(c, a, st))) return rc;
        }
        // d h_l = rho (mask_l / keep * sum_taps d z(t + (2 - tap) d) W_tap^T + d h_{l+1})   (modules.py:484, 517-520 backwards)
        const float scale = (l > 0) ? c->res_scale : 1.0f;
        for (int tap = 0; tap < 3; ++tap) {
            SgemmArgs a = mk(s->DZ, G, (2 - tap) * d, P + c->lay[l].dil_k + (int64_t)tap * R * G, G, G, R, gx_dn, R); a.wk = 1; a.wm = G;
            a.accumulate = tap > 0;
            if (tap == 0) { a.scale = scale; if (!top) { a.add = gx_up; a.ld_add = R; } }
            else a.alpha = scale;
            if (c->cfg.dropout > 0.0f) {
                wn_layer_key(c->fseed, l, &a.key_lo, &a.key_hi); a.thresh16 = (uint32_t)lrintf(c->cfg.dropout * 65536.0f);
                a.keep_scale = 1.0f / (1.0f - c->cfg.dropout); a.drop_ld = R; a.drop_on_out = 1;
            }
            if ((rc = sgemm(c, a, st))) return rc;
        }
    }
    // ---- input convolution (wavenet.py:705): gx = GX[0] holds dL/dh_0
    const float* g0 = s->GX[0];
    if (c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE)
        hipLaunchKernelGGL(wn_f32_first_conv_bwd_ids, dim3(cdiv(rows * R, 256)), dim3(256), 0, st, (const int32_t*)c->fx, g0, grads + c->first.dil_k, rows, R);
    else if ((rc = wgrad32(c, (const float*)c->fx, 1, 0, 1, g0, R, R, grads + c->first.dil_k, R, 1.0f, -1, grads + c->first.dil_b, nullptr, st))) return rc;
    if (c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE && (rc = wgrad32(c, nullptr, 0, 0, 1, g0, R, R, nullptr, 0, 1.0f, -1, grads + c->first.dil_b, nullptr, st))) return rc;
    if ((rc = wn_gin_bwd(c, grads, st, true))) return rc;      // d W_g, d b_g, d embedding table from the per-utterance sums above
    // ---- upsample net (modules.py:524-770 backwards: the fp32 kernels the bf16 engine uses too)
    if (c->cfg.upsample_type != WN_UP_NEAREST) {
        hipLaunchKernelGGL(wn_f32_transpose_back, dim3(cdiv(rows * C, 256)), dim3(256), 0, st, s->DC, s->DCT, B, C, T);
        if ((rc = wn_upsample_bwd(c, s->DCT, grads, st))) return rc;
    }
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
