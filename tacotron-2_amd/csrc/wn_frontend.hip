// Everything in front of / around the residual stack that is not an MFMA contraction: the input convolution and its gradient (column
// sums, wn_colsum2), the five upsample nets forward / backward (modules.py:524-770), global conditioning (wavenet.py:14-51, 669-678).
// HBM-bound: coalesced along the contiguous axis, 64-wide waves.
#include "wn_common.h"
#include <math.h>
#include <algorithm>

// =================================================================================== input convolution
// wavenet.py:705 / modules.py:336: h0[t][r] = W[cin][r] x[cin][t] + b[r]; Cin = 1 (scalar) or a one-hot row gather.
__global__ void wn_first_conv_fwd(const void* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                  bf16_t* __restrict__ X0, bf16_t* __restrict__ XD0, int64_t rows, int R, int is_ids,
                                  uint32_t key_lo, uint32_t key_hi, uint32_t thresh16, float keep_scale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int r8 = R >> 3;
    if (idx >= rows * r8) return;
    const int64_t row = idx / r8; const int c0 = (int)(idx - row * r8) * 8;
    float v[8];
    if (is_ids) {
        const int id = ((const int32_t*)x)[row];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = W[(int64_t)id * R + c0 + i] + bias[c0 + i];
    } else {
        const float xv = ((const float*)x)[row];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = W[c0 + i] * xv + bias[c0 + i];
    }
    bf16_t hb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) hb[i] = f2bf(v[i]);
    *reinterpret_cast<uint4*>(X0 + row * R + c0) = make_uint4(hb[0] | ((uint32_t)hb[1] << 16), hb[2] | ((uint32_t)hb[3] << 16),
                                                               hb[4] | ((uint32_t)hb[5] << 16), hb[6] | ((uint32_t)hb[7] << 16));
    if (XD0) {      // layer-0 conv input with its dropout mask applied once (modules.py:484)
        const uint32_t e0 = (uint32_t)(row * R + c0);
        uint32_t o[4], wq[4];        // e0 % 8 == 0 (c0 % 8 == 0, R % 8 == 0)
        wn_drop_quad(key_lo, key_hi, e0 >> 2, wq[0], wq[1]); wn_drop_quad(key_lo, key_hi, (e0 >> 2) + 1, wq[2], wq[3]);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t w = wq[p];
            const float lo = ((w & 0xffffu) >= thresh16) ? bf2f(hb[2 * p]) * keep_scale : 0.0f;
            const float hi = ((w >> 16) >= thresh16) ? bf2f(hb[2 * p + 1]) * keep_scale : 0.0f;
            o[p] = pack_bf2(lo, hi);
        }
        *reinterpret_cast<uint4*>(XD0 + row * R + c0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// dW[cin][r] = sum_t x[cin][t] g0[t][r];  db[r] = sum_t g0[t][r]
// One-hot input (mu-law-quantize): a row scatter by class id -- float atomics into the [Q][R] kernel gradient (C1-sized models only).
__global__ void wn_first_conv_bwd_ids(const int32_t* __restrict__ ids, const bf16_t* __restrict__ g0, float* __restrict__ dW,
                                      int64_t rows, int R, int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    for (int r = threadIdx.x; r < R; r += blockDim.x)
        for (int64_t row = r0; row < r1; ++row) unsafeAtomicAdd(&dW[(int64_t)ids[row] * R + r], bf2f(g0[row * R + r]));
}
// Column sums of a bf16 [rows][ld] matrix, optionally also weighted by a per-row scalar: sum_t M[t][c] and sum_t x[t] M[t][c].
// Two stages in a fixed order, no atomics (bit-reproducible): part[blk][0][c], part[blk][1][c], then wn_colsum2_reduce.
// (Round 2's input-conv gradient walked 128 rows per block one 2-byte load at a time and finished with float atomics: 77 us alone,
// 0.5 ms beside the weight-gradient kernels.  This one reads 16 B per lane: scalar-input d W / d b and the head-bias column sums.)
__global__ __launch_bounds__(256) void wn_colsum2_kernel(const bf16_t* __restrict__ M, int ld, int ncols, const float* __restrict__ xw,
                                                         int64_t rows, int rows_per_block, float* __restrict__ part) {
    __shared__ float red[2][2048];                   // [b | w][row lane][ncols]   (row lanes * ncols <= 2048)
    const int c8n = ncols >> 3, rgn = 256 / c8n;
    const int tid = threadIdx.x, c8 = tid % c8n, rg = tid / c8n;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float sb[8], sw[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sb[e] = 0.0f; sw[e] = 0.0f; }
    if (rg < rgn) {
        for (int64_t row = r0 + rg; row < r1; row += rgn) {
            const uint4 v = *reinterpret_cast<const uint4*>(M + row * ld + c8 * 8);
            const float f[8] = {bf2f((bf16_t)(v.x & 0xffff)), bf2f((bf16_t)(v.x >> 16)), bf2f((bf16_t)(v.y & 0xffff)), bf2f((bf16_t)(v.y >> 16)),
                                bf2f((bf16_t)(v.z & 0xffff)), bf2f((bf16_t)(v.z >> 16)), bf2f((bf16_t)(v.w & 0xffff)), bf2f((bf16_t)(v.w >> 16))};
            const float x = xw ? xw[row] : 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sb[e] += f[e]; sw[e] = __builtin_fmaf(x, f[e], sw[e]); }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[0][rg * ncols + c8 * 8 + e] = sb[e]; red[1][rg * ncols + c8 * 8 + e] = sw[e]; }
    }
    __syncthreads();
    for (int i = tid; i < 2 * ncols; i += 256) {
        const int w = i / ncols, cix = i - w * ncols;
        if (w == 1 && !xw) continue;
        float s = 0.0f;
        for (int g = 0; g < rgn; ++g) s += red[w][g * ncols + cix];
        part[((int64_t)blockIdx.x * 2 + w) * ncols + cix] = s;
    }
}
// out_b[c] = sum_blk part[blk][0][c] (c < nvalid), out_w[c] = sum_blk part[blk][1][c]; block = 16 columns x 2 sums x 32 block lanes
// (1024 threads, 4 loads in flight each), combined in a fixed order
__global__ __launch_bounds__(1024) void wn_colsum2_reduce(const float* __restrict__ part, int nblk, int ncols, int nvalid,
                                                          float* __restrict__ out_b, float* __restrict__ out_w) {
    __shared__ float red[32][33];
    const int cw = threadIdx.x & 31, bl = threadIdx.x >> 5;          // cw: (column, which sum); bl: block lane
    const int w = cw >> 4, cix = blockIdx.x * 16 + (cw & 15);
    const bool on = cix < ncols && (w == 0 ? out_b != nullptr : out_w != nullptr);
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (on) {
        const float* p = part + (int64_t)w * ncols + cix;
        int b = bl;
        for (; b + 96 < nblk; b += 128) {
            s0 += p[(int64_t)b * 2 * ncols]; s1 += p[(int64_t)(b + 32) * 2 * ncols]; s2 += p[(int64_t)(b + 64) * 2 * ncols]; s3 += p[(int64_t)(b + 96) * 2 * ncols];
        }
        for (; b < nblk; b += 32) s0 += p[(int64_t)b * 2 * ncols];
    }
    red[bl][cw] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x < 32) {
        float s = 0.0f;
        for (int g = 0; g < 32; ++g) s += red[g][threadIdx.x];
        if (on && cix < nvalid) (w == 0 ? out_b : out_w)[cix] = s;
    }
}
// column sums (+ x-weighted column sums) of M [rows][ld] into out_b / out_w (either may be null); `slot` picks one of the two
// ctx-owned partial regions (launches on different streams may overlap)
int wn_colsum2(wn_ctx* c, const bf16_t* M, int ld, int ncols, int nvalid, const float* xw, int64_t rows, float* out_b, float* out_w, int slot, hipStream_t st) {
    WnTraceScope trace_scope(c, st, WN_TR_COLSUM);
    if (ncols % 8 || ncols > 1024 || 256 / (ncols / 8) * ncols > 2048) WN_FAIL(c, WN_E_SHAPE, "wn_colsum2: %d columns", ncols);
    const int rpb = (int)std::max<int64_t>(64, (rows + WN_CS_MAXBLK - 1) / WN_CS_MAXBLK);
    const int nblk = cdiv(rows, rpb);
    float* part = c->cs_part + (size_t)slot * WN_CS_MAXBLK * 2 * 1024;
    hipLaunchKernelGGL(wn_colsum2_kernel, dim3(nblk), dim3(256), 0, st, M, ld, ncols, out_w ? xw : nullptr, rows, rpb, part);
    hipLaunchKernelGGL(wn_colsum2_reduce, dim3(cdiv(ncols, 16)), dim3(1024), 0, st, part, nblk, ncols, nvalid, out_b, out_w);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}

int wn_first_conv(wn_ctx* c, hipStream_t st) {
    WnTraceScope trace_scope(c, st, WN_TR_INPUT_CONV);
    const int64_t rows = (int64_t)c->fB * c->fT;
    const int is_ids = c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE;
    uint32_t klo = 0, khi = 0; wn_layer_key(c->fseed, 0, &klo, &khi);
    const bool drop = c->cfg.dropout > 0.0f;
    hipLaunchKernelGGL(wn_first_conv_fwd, dim3(cdiv(rows * (c->R / 8), 256)), dim3(256), 0, st, c->fx,
                       c->params_dev + c->first.dil_k, c->params_dev + c->first.dil_b, c->X, drop ? c->XD : nullptr, rows, c->R, is_ids,
                       klo, khi, (uint32_t)lrintf(c->cfg.dropout * 65536.0f), 1.0f / (1.0f - c->cfg.dropout));
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
int wn_first_conv_grad(wn_ctx* c, const bf16_t* g0, float* grads, hipStream_t st) {
    WnTraceScope trace_scope(c, st, WN_TR_INPUT_CONV_BWD);
    const int64_t rows = (int64_t)c->fB * c->fT;
    const int is_ids = c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE;
    if (is_ids) {
        const int rpb = 128;
        hipLaunchKernelGGL(wn_first_conv_bwd_ids, dim3(cdiv(rows, rpb)), dim3(256), 0, st, (const int32_t*)c->fx, g0, grads + c->first.dil_k, rows, c->R, rpb);
        WN_LAUNCH_CHECK(c);
    }
    // d b = column sums of d h_0; scalar input: d W = the x-weighted column sums, from the same pass
    return wn_colsum2(c, g0, c->R, c->R, c->R, is_ids ? nullptr : (const float*)c->fx, rows, grads + c->first.dil_b,
                      is_ids ? nullptr : grads + c->first.dil_k, 0, st);
}

// =================================================================================== upsample net
// modules.py:524-770, wavenet.py:680-702.  Layouts [B][C(freq)][T] fp32; the last layer also emits the bf16
// time-major copy cbt[b*T+t][C] that the gate GEMM stages.
__device__ __forceinline__ float act_fwd(float v, int act, float alpha) {
    if (act == WN_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == WN_ACT_LEAKY_RELU) return v > 0.0f ? v : alpha * v;
    return v;
}
__device__ __forceinline__ float act_grad(float out, int act, float alpha) {   // derivative expressed through the OUTPUT
    if (act == WN_ACT_RELU) return out > 0.0f ? 1.0f : 0.0f;
    if (act == WN_ACT_LEAKY_RELU) return out > 0.0f ? 1.0f : alpha;
    return 1.0f;
}

// type 0: nearest (s = hop); 1: 2D transposed conv k=(fk,s) stride (1,s); 2: SubPixel conv k=(fk,3) + shuffle
__global__ void wn_up_fwd(const float* __restrict__ in, float* __restrict__ out, bf16_t* __restrict__ cbt,
                          const float* __restrict__ K, const float* __restrict__ bias, int B, int C, int Tin, int s,
                          int fk, int type, int act, float alpha) {
    const int Tout = Tin * s;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * C * Tout) return;
    const int to = (int)(idx % Tout); const int64_t bf = idx / Tout;
    const int f = (int)(bf % C), b = (int)(bf / C);
    const int t = to / s, j = to - t * s;
    const float* inb = in + (int64_t)b * C * Tin;
    float v;
    if (type == 0) v = inb[(int64_t)f * Tin + t];
    else if (type == 1) {
        const int pf = (fk - 1) / 2;
        v = bias[0];
        for (int kf = 0; kf < fk; ++kf) { const int fs = f - kf + pf; if (fs >= 0 && fs < C) v += inb[(int64_t)fs * Tin + t] * K[kf * s + j]; }
        v = act_fwd(v, act, alpha);
    } else {
        const int pf = (fk - 1) / 2;
        v = bias[j];
        for (int kf = 0; kf < fk; ++kf) {
            const int fs = f + kf - pf; if (fs < 0 || fs >= C) continue;
            for (int kt = 0; kt < 3; ++kt) { const int tsrc = t + kt - 1; if (tsrc >= 0 && tsrc < Tin) v += inb[(int64_t)fs * Tin + tsrc] * K[(kf * 3 + kt) * s + j]; }
        }
        v = act_fwd(v, act, alpha);
    }
    out[idx] = v;
    if (cbt) cbt[((int64_t)b * Tout + to) * C + f] = f2bf(v);
}

// type 3 'Resize' (modules.py:657-695): nearest-neighbour resize x s along time, then Conv2D 1->1, kernel (fk, s), SAME
//   (TF pads (k-1)/2 before and the rest after on each axis): out[f][to] = b + sum_{kf,kt} up[f+kf-pf][to+kt-pl] K[kf][kt],
//   up[f'][tu] = in[f'][tu / s];
// type 4 '1D' (modules.py:697-733): Conv2DTranspose C->C, kernel (1, s), stride (1, s), TF layout [1][s][out][in]:
//   out[co][t*s+j] = b[co] + sum_ci in[ci][t] K[j][co][ci].
// Both are off in the reference's two hparams files: one straightforward thread per output element.
__global__ void wn_up_fwd_generic(const float* __restrict__ in, float* __restrict__ out, bf16_t* __restrict__ cbt,
                                  const float* __restrict__ K, const float* __restrict__ bias, int B, int C, int Tin, int s,
                                  int fk, int type, int act, float alpha) {
    const int Tout = Tin * s;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * C * Tout) return;
    const int to = (int)(idx % Tout); const int64_t bf = idx / Tout;
    const int f = (int)(bf % C), b = (int)(bf / C);
    const float* inb = in + (int64_t)b * C * Tin;
    float v;
    if (type == 3) {
        const int pf = (fk - 1) / 2, pl = (s - 1) / 2;
        v = bias[0];
        for (int kf = 0; kf < fk; ++kf) {
            const int fs = f + kf - pf; if (fs < 0 || fs >= C) continue;
            for (int kt = 0; kt < s; ++kt) { const int tu = to + kt - pl; if (tu >= 0 && tu < Tout) v += inb[(int64_t)fs * Tin + tu / s] * K[kf * s + kt]; }
        }
    } else {
        const int t = to / s, j = to - t * s;
        v = bias[f];
        const float* Kj = K + ((int64_t)j * C + f) * C;
        for (int ci = 0; ci < C; ++ci) v += inb[(int64_t)ci * Tin + t] * Kj[ci];
    }
    v = act_fwd(v, act, alpha);
    out[idx] = v;
    if (cbt) cbt[((int64_t)b * Tout + to) * C + f] = f2bf(v);
}

// parameter gradients of types 3 / 4: one WAVE per (kernel or bias element, batch slice); lanes stride over time, shuffle
// reduction, one atomic per wave.  blockIdx.y = slice: (b, f) row for 'Resize', b for '1D'.
__global__ void wn_up_bwd_params_generic(const float* __restrict__ in, const float* __restrict__ out, const float* __restrict__ dout,
                                         float* __restrict__ dK, float* __restrict__ dbias, int B, int C, int Tin, int s, int fk,
                                         int type, int act, float alpha) {
    const int Tout = Tin * s;
    const int nk = (type == 3) ? fk * s : s * C * C, nb = (type == 3) ? 1 : C;
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (e >= nk + nb) return;
    float a = 0.0f;
    if (type == 3) {
        const int b = blockIdx.y / C, f = blockIdx.y % C;
        const int pf = (fk - 1) / 2, pl = (s - 1) / 2;
        const int64_t ro = ((int64_t)b * C + f) * Tout;
        if (e < nk) {
            const int kf = e / s, kt = e % s;
            const int fs = f + kf - pf;
            if (fs >= 0 && fs < C) {
                const float* inr = in + ((int64_t)b * C + fs) * Tin;
                for (int to = lane; to < Tout; to += 64) {
                    const int tu = to + kt - pl; if (tu < 0 || tu >= Tout) continue;
                    a += dout[ro + to] * act_grad(out[ro + to], act, alpha) * inr[tu / s];
                }
            }
        } else {
            for (int to = lane; to < Tout; to += 64) a += dout[ro + to] * act_grad(out[ro + to], act, alpha);
        }
    } else {
        const int b = blockIdx.y;
        if (e < nk) {
            const int ci = e % C, co = (e / C) % C, j = e / (C * C);
            const int64_t ro = ((int64_t)b * C + co) * Tout; const float* inr = in + ((int64_t)b * C + ci) * Tin;
            for (int t = lane; t < Tin; t += 64) { const int64_t o = ro + (int64_t)t * s + j; a += dout[o] * act_grad(out[o], act, alpha) * inr[t]; }
        } else {
            const int co = e - nk;
            const int64_t ro = ((int64_t)b * C + co) * Tout;
            for (int to = lane; to < Tout; to += 64) a += dout[ro + to] * act_grad(out[ro + to], act, alpha);
        }
    }
    for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o);
    if (lane == 0 && a != 0.0f) unsafeAtomicAdd(e < nk ? &dK[e] : &dbias[e - nk], a);
}

__global__ void wn_up_bwd_input_generic(const float* __restrict__ out, const float* __restrict__ dout, float* __restrict__ din,
                                        const float* __restrict__ K, int B, int C, int Tin, int s, int fk, int type, int act, float alpha) {
    const int Tout = Tin * s;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * C * Tin) return;
    const int t = (int)(idx % Tin); const int64_t bf = idx / Tin;
    const int f = (int)(bf % C), b = (int)(bf / C);
    const float* ob = out + (int64_t)b * C * Tout; const float* db = dout + (int64_t)b * C * Tout;
    float a = 0.0f;
    if (type == 3) {
        const int pf = (fk - 1) / 2, pl = (s - 1) / 2;
        for (int kf = 0; kf < fk; ++kf) {
            const int fo = f - kf + pf; if (fo < 0 || fo >= C) continue;       // output row whose tap kf reads input row f
            for (int tu = t * s; tu < t * s + s; ++tu)
                for (int kt = 0; kt < s; ++kt) {
                    const int to = tu - kt + pl; if (to < 0 || to >= Tout) continue;
                    const int64_t o = (int64_t)fo * Tout + to;
                    a += K[kf * s + kt] * db[o] * act_grad(ob[o], act, alpha);
                }
        }
    } else {
        for (int j = 0; j < s; ++j)
            for (int co = 0; co < C; ++co) {
                const int64_t o = (int64_t)co * Tout + (int64_t)t * s + j;
                a += K[((int64_t)j * C + co) * C + f] * db[o] * act_grad(ob[o], act, alpha);
            }
    }
    din[idx] = a;
}

// dpre = dout * act'(out);  dK[kf][j], dbias.  One workgroup per (b, f) row; thread x owns phase j = x % s of the stride-s
// output grid (to = j + s*q), so its kernel taps are fixed and accumulate in registers; one LDS atomic per thread and
// tap at the end, one global atomic per workgroup and tap.  (The v0 kernel did an LDS atomic per ELEMENT and tap on ~30
// addresses: 290 us on the last upsample layer.)
#define WN_UP_MAXTAP 27
__global__ __launch_bounds__(256) void wn_up_bwd_params(const float* __restrict__ in, const float* __restrict__ out, const float* __restrict__ dout,
                                 float* __restrict__ dK, float* __restrict__ dbias, int B, int C, int Tin, int s, int fk,
                                 int type, int act, float alpha) {
    extern __shared__ float sh[];          // [nk + nb]
    const int nk = (type == 1) ? fk * s : fk * 3 * s;
    const int nb = (type == 1) ? 1 : s;
    for (int i = threadIdx.x; i < nk + nb; i += blockDim.x) sh[i] = 0.0f;
    __syncthreads();
    const int Tout = Tin * s;
    const int b = blockIdx.x / C, f = blockIdx.x % C;
    const int groups = blockDim.x / s;                 // (threads beyond groups*s idle)
    const int j = threadIdx.x % s, q0 = threadIdx.x / s;
    const int pf = (fk - 1) / 2;
    const int ntap = (type == 1) ? fk : fk * 3;
    float dk[WN_UP_MAXTAP], db = 0.0f;
#pragma unroll
    for (int i = 0; i < WN_UP_MAXTAP; ++i) dk[i] = 0.0f;
    if (q0 < groups) {
        const float* inb = in + (int64_t)b * C * Tin;
        const int64_t rowo = ((int64_t)b * C + f) * Tout;
        for (int t = q0; t < Tin; t += groups) {
            const int64_t o = rowo + (int64_t)t * s + j;
            const float dp = dout[o] * act_grad(out[o], act, alpha);
            db += dp;
            if (type == 1) {
#pragma unroll
                for (int kf = 0; kf < 9; ++kf) {
                    if (kf < fk) { const int fs = f - kf + pf; if (fs >= 0 && fs < C) dk[kf] += inb[(int64_t)fs * Tin + t] * dp; }
                }
            } else {
#pragma unroll
                for (int kf = 0; kf < 9; ++kf) {
                    if (kf < fk) {
                        const int fs = f + kf - pf;
                        if (fs >= 0 && fs < C) {
#pragma unroll
                            for (int kt = 0; kt < 3; ++kt) { const int tsrc = t + kt - 1; if (tsrc >= 0 && tsrc < Tin) dk[kf * 3 + kt] += inb[(int64_t)fs * Tin + tsrc] * dp; }
                        }
                    }
                }
            }
        }
        if (type == 1) atomicAdd(&sh[nk], db); else atomicAdd(&sh[nk + j], db);
#pragma unroll
        for (int i = 0; i < WN_UP_MAXTAP; ++i) if (i < ntap) atomicAdd(&sh[i * s + j], dk[i]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nk; i += blockDim.x) if (sh[i] != 0.0f) unsafeAtomicAdd(&dK[i], sh[i]);
    for (int i = threadIdx.x; i < nb; i += blockDim.x) if (sh[nk + i] != 0.0f) unsafeAtomicAdd(&dbias[i], sh[nk + i]);
}

// din[b][f'][t]: a group of GL lanes (GL = power of two <= 64, >= min(s, 64)) per input element; the lanes stride over the s
// output phases j (contiguous in memory: coalesced), then shuffle-reduce.  (One thread per element walked fk*3*s strided
// addresses: 1.4 ms for the s = 25 SubPixel layer of hparams.py.)
__global__ __launch_bounds__(256) void wn_up_bwd_input(const float* __restrict__ out, const float* __restrict__ dout, float* __restrict__ din,
                                const float* __restrict__ K, int B, int C, int Tin, int s, int fk, int type, int act, float alpha, int GL) {
    const int Tout = Tin * s;
    const int lj = threadIdx.x & (GL - 1);
    const int64_t idx = (int64_t)blockIdx.x * (blockDim.x / GL) + threadIdx.x / GL;
    const bool live = idx < (int64_t)B * C * Tin;
    float a = 0.0f;
    if (live) {
        const int t = (int)(idx % Tin); const int64_t bf = idx / Tin;
        const int f = (int)(bf % C), b = (int)(bf / C);
        const int pf = (fk - 1) / 2;
        const float* ob = out + (int64_t)b * C * Tout; const float* db = dout + (int64_t)b * C * Tout;
        if (type == 0) {
            for (int j = lj; j < s; j += GL) a += db[(int64_t)f * Tout + t * s + j];
        } else if (type == 1) {
            for (int kf = 0; kf < fk; ++kf) {
                const int fo = f + kf - pf; if (fo < 0 || fo >= C) continue;
                for (int j = lj; j < s; j += GL) { const int64_t o = (int64_t)fo * Tout + t * s + j; a += K[kf * s + j] * db[o] * act_grad(ob[o], act, alpha); }
            }
        } else {
            for (int kf = 0; kf < fk; ++kf) {
                const int fo = f - kf + pf; if (fo < 0 || fo >= C) continue;
                for (int kt = 0; kt < 3; ++kt) {
                    const int tt = t - kt + 1; if (tt < 0 || tt >= Tin) continue;
                    for (int j = lj; j < s; j += GL) { const int64_t o = (int64_t)fo * Tout + tt * s + j; a += K[(kf * 3 + kt) * s + j] * db[o] * act_grad(ob[o], act, alpha); }
                }
            }
        }
    }
    for (int o = GL >> 1; o > 0; o >>= 1) a += __shfl_down(a, o, GL);
    if (live && lj == 0) din[idx] = a;
}

// ---- round-2 replacements for types 1 / 2: no float atomics anywhere (bit-reproducible gradients), enough workgroups to fill the part.
// Stage 1: workgroup (row = (b, f), time slice y) accumulates its taps in registers exactly like wn_up_bwd_params, folds the
// threads of one phase j in a FIXED order through LDS and writes ne = nk + nb partial sums to part[block][e].
// Stage 2 (wn_up_bwd_params_reduce): one workgroup per element sums the blocks in a fixed order and STORES dK / dbias.
__global__ __launch_bounds__(256) void wn_up_bwd_params2(const float* __restrict__ in, const float* __restrict__ out, const float* __restrict__ dout,
                                 float* __restrict__ part, int B, int C, int Tin, int s, int fk, int type, int act, float alpha, int tchunk) {
    __shared__ float sh[(WN_UP_MAXTAP + 1) * 256];
    const int nk = (type == 1) ? fk * s : fk * 3 * s;
    const int nb = (type == 1) ? 1 : s;
    const int Tout = Tin * s;
    const int b = blockIdx.x / C, f = blockIdx.x % C;
    const int groups = blockDim.x / s;
    const int j = threadIdx.x % s, q0 = threadIdx.x / s;
    const int pf = (fk - 1) / 2;
    const int ntap = (type == 1) ? fk : fk * 3;
    const int tlo = blockIdx.y * tchunk, thi = min(Tin, tlo + tchunk);
    float dk[WN_UP_MAXTAP], db = 0.0f;
#pragma unroll
    for (int i = 0; i < WN_UP_MAXTAP; ++i) dk[i] = 0.0f;
    if (q0 < groups) {
        const float* inb = in + (int64_t)b * C * Tin;
        const int64_t rowo = ((int64_t)b * C + f) * Tout;
        for (int t = tlo + q0; t < thi; t += groups) {
            const int64_t o = rowo + (int64_t)t * s + j;
            const float dp = dout[o] * act_grad(out[o], act, alpha);
            db += dp;
            if (type == 1) {
#pragma unroll
                for (int kf = 0; kf < 9; ++kf) {
                    if (kf < fk) { const int fs = f - kf + pf; if (fs >= 0 && fs < C) dk[kf] += inb[(int64_t)fs * Tin + t] * dp; }
                }
            } else {
#pragma unroll
                for (int kf = 0; kf < 9; ++kf) {
                    if (kf < fk) {
                        const int fs = f + kf - pf;
                        if (fs >= 0 && fs < C) {
#pragma unroll
                            for (int kt = 0; kt < 3; ++kt) { const int tsrc = t + kt - 1; if (tsrc >= 0 && tsrc < Tin) dk[kf * 3 + kt] += inb[(int64_t)fs * Tin + tsrc] * dp; }
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < WN_UP_MAXTAP; ++i) if (i < ntap) sh[i * 256 + threadIdx.x] = dk[i];
    sh[ntap * 256 + threadIdx.x] = db;
    __syncthreads();
    const int ne = nk + nb;
    float* po = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * ne;
    for (int e = threadIdx.x; e < ne; e += blockDim.x) {
        float a = 0.0f;
        if (e < nk) { const int tap = e / s, jj = e - tap * s; for (int q = 0; q < groups; ++q) a += sh[tap * 256 + q * s + jj]; }
        else if (type == 1) { for (int x = 0; x < groups * s; ++x) a += sh[ntap * 256 + x]; }
        else { const int jj = e - nk; for (int q = 0; q < groups; ++q) a += sh[ntap * 256 + q * s + jj]; }
        po[e] = a;
    }
}

__global__ __launch_bounds__(256) void wn_up_bwd_params_reduce(const float* __restrict__ part, int nblk, int ne, int nk, float* __restrict__ dK, float* __restrict__ dbias) {
    __shared__ float sh[256];
    const int e = blockIdx.x;
    float a = 0.0f;
    for (int i = threadIdx.x; i < nblk; i += 256) a += part[(int64_t)i * ne + e];
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { if (e < nk) dK[e] = sh[0]; else dbias[e - nk] = sh[0]; }
}

// din[b][f'][t0 .. t0+TB): the fk output rows this input row feeds (x act') are staged once, coalesced, into LDS ([fk][W],
// W = (TB + 2 halo) s, halo = 1 frame for the 3-tap SubPixel kernel) together with the kernel; one thread per input frame.
__global__ __launch_bounds__(256) void wn_up_bwd_input2(const float* __restrict__ out, const float* __restrict__ dout, float* __restrict__ din,
                                const float* __restrict__ K, int B, int C, int Tin, int s, int fk, int type, int act, float alpha, int TB) {
    extern __shared__ float dsh[];
    const int halo = (type == 2) ? 1 : 0;
    const int W = (TB + 2 * halo) * s;
    const int nk = (type == 1) ? fk * s : fk * 3 * s;
    float* dp = dsh; float* Ks = dsh + fk * W;
    const int Tout = Tin * s;
    const int b = blockIdx.y / C, f = blockIdx.y % C;
    const int t0 = blockIdx.x * TB;
    const int pf = (fk - 1) / 2;
    for (int i = threadIdx.x; i < nk; i += blockDim.x) Ks[i] = K[i];
    for (int kf = 0; kf < fk; ++kf) {
        const int fo = (type == 1) ? f + kf - pf : f - kf + pf;
        const bool rowok = fo >= 0 && fo < C;
        const int64_t ro = ((int64_t)b * C + fo) * Tout;
        const int to0 = (t0 - halo) * s;
        for (int x = threadIdx.x; x < W; x += blockDim.x) {
            const int to = to0 + x;
            float v = 0.0f;
            if (rowok && to >= 0 && to < Tout) v = dout[ro + to] * act_grad(out[ro + to], act, alpha);
            dp[kf * W + x] = v;
        }
    }
    __syncthreads();
    for (int tl = threadIdx.x; tl < TB && t0 + tl < Tin; tl += blockDim.x) {
        float a = 0.0f;
        if (type == 1) {
            for (int kf = 0; kf < fk; ++kf) { const float* d = dp + kf * W + tl * s; const float* k = Ks + kf * s; for (int j = 0; j < s; ++j) a += k[j] * d[j]; }
        } else {
            for (int kf = 0; kf < fk; ++kf)
                for (int kt = 0; kt < 3; ++kt) { const float* d = dp + kf * W + (tl - kt + 2) * s; const float* k = Ks + (kf * 3 + kt) * s; for (int j = 0; j < s; ++j) a += k[j] * d[j]; }
        }
        din[((int64_t)b * C + f) * Tin + t0 + tl] = a;
    }
}

static int up_type_code(const wn_ctx* c) {
    switch (c->cfg.upsample_type) { case WN_UP_NEAREST: return 0; case WN_UP_2D: return 1; case WN_UP_SUBPIXEL: return 2; case WN_UP_RESIZE: return 3; default: return 4; }
}

// c_in [B,C,Tc] fp32 -> CUP[i] (fp32 per level), cbt (bf16 time-major)
int wn_upsample_fwd(wn_ctx* c, const float*, const float* cin, int B, int Tc, hipStream_t st) {
    WnTraceScope trace_scope(c, st, WN_TR_UPSAMPLE_FWD);
    const int type = up_type_code(c);
    const int C = c->C;
    if (type == 0) {
        const int64_t n = (int64_t)B * C * Tc * c->hop;
        hipLaunchKernelGGL(wn_up_fwd, dim3(cdiv(n, 256)), dim3(256), 0, st, cin, c->CUP[0], c->cbt, nullptr, nullptr, B, C, Tc, c->hop, 1, 0, 0, 0.0f);
        WN_LAUNCH_CHECK(c);
        return WN_OK;
    }
    const float* in = cin; int Tin = Tc;
    for (int i = 0; i < c->cfg.n_upsample; ++i) {
        const int s = c->cfg.upsample_scales[i];
        const bool last = (i == c->cfg.n_upsample - 1);
        const int64_t n = (int64_t)B * C * Tin * s;
        if (type >= 3)
            hipLaunchKernelGGL(wn_up_fwd_generic, dim3(cdiv(n, 256)), dim3(256), 0, st, in, c->CUP[i], last ? c->cbt : nullptr,
                               c->params_dev + c->up_k[i], c->params_dev + c->up_b[i], B, C, Tin, s, c->cfg.freq_axis_kernel_size,
                               type, c->cfg.upsample_activation, c->cfg.leaky_alpha);
        else
        hipLaunchKernelGGL(wn_up_fwd, dim3(cdiv(n, 256)), dim3(256), 0, st, in, c->CUP[i], last ? c->cbt : nullptr,
                           c->params_dev + c->up_k[i], c->params_dev + c->up_b[i], B, C, Tin, s, c->cfg.freq_axis_kernel_size,
                           type, c->cfg.upsample_activation, c->cfg.leaky_alpha);
        WN_LAUNCH_CHECK(c);
        in = c->CUP[i]; Tin *= s;
    }
    return WN_OK;
}

// dc_final [B,C,T] fp32 (d loss / d upsampled conditioning) -> grads of the upsample kernels/biases
int wn_upsample_bwd(wn_ctx* c, const float* dc_final, float* grads, hipStream_t st) {
    WnTraceScope trace_scope(c, st, WN_TR_UPSAMPLE_BWD);
    const int type = up_type_code(c);
    if (type == 0) return WN_OK;               // no parameters
    const int C = c->C, B = c->fB;
    const float* dout = dc_final;
    int Tout = c->fT;
    for (int i = c->cfg.n_upsample - 1; i >= 0; --i) {
        const int s = c->cfg.upsample_scales[i];
        const int Tin = Tout / s;
        const float* in = (i == 0) ? c->fc : c->CUP[i - 1];
        const int fk = c->cfg.freq_axis_kernel_size;
        if (type >= 3) {
            const int nkb = (type == 3) ? fk * s + 1 : s * C * C + C;
            hipLaunchKernelGGL(wn_up_bwd_params_generic, dim3(cdiv(nkb, 4), type == 3 ? B * C : B), dim3(256), 0, st, in, c->CUP[i], dout,
                               grads + c->up_k[i], grads + c->up_b[i], B, C, Tin, s, fk, type, c->cfg.upsample_activation, c->cfg.leaky_alpha);
            WN_LAUNCH_CHECK(c);
            if (i > 0) {
                float* din = c->DCUP[i & 1];
                const int64_t ni = (int64_t)B * C * Tin;
                hipLaunchKernelGGL(wn_up_bwd_input_generic, dim3(cdiv(ni, 256)), dim3(256), 0, st, c->CUP[i], dout, din, c->params_dev + c->up_k[i],
                                   B, C, Tin, s, fk, type, c->cfg.upsample_activation, c->cfg.leaky_alpha);
                WN_LAUNCH_CHECK(c);
                dout = din;
            }
            Tout = Tin;
            continue;
        }
        const int nk = (type == 1) ? fk * s : fk * 3 * s, nb = (type == 1) ? 1 : s;
        if ((size_t)(nk + nb) * 4 > 60000) WN_FAIL(c, WN_E_UNSUPPORTED, "upsample scale %d too large for the LDS partials", s);
        const int64_t n = (int64_t)B * C * Tout;
        if (s > 256) WN_FAIL(c, WN_E_UNSUPPORTED, "upsample scale %d > 256", s);
        (void)n;
        static const bool v1 = getenv("WN_UP_BWD_V1") != nullptr;          // A/B switch: the round-1 kernels (float atomics)
        if (v1) {
        hipLaunchKernelGGL(wn_up_bwd_params, dim3(B * C), dim3(256), (nk + nb) * 4, st, in, c->CUP[i], dout,
                           grads + c->up_k[i], grads + c->up_b[i], B, C, Tin, s, fk, type, c->cfg.upsample_activation, c->cfg.leaky_alpha);
        WN_LAUNCH_CHECK(c);
        } else {
            const int rows = B * C, groups = 256 / s;
            int Y = std::max(1, std::min(cdiv(2048, rows), cdiv(Tin, groups)));
            const int tchunk = cdiv(Tin, Y); Y = cdiv(Tin, tchunk);
            const int ne = nk + nb, nblk = rows * Y;
            if ((int64_t)nblk * ne > c->uppart_floats) WN_FAIL(c, WN_E_STATE, "upsample partial buffer too small (%d x %d)", nblk, ne);
            if ((type == 1 ? fk : fk * 3) > WN_UP_MAXTAP) WN_FAIL(c, WN_E_UNSUPPORTED, "freq_axis_kernel_size %d too large", fk);
            hipLaunchKernelGGL(wn_up_bwd_params2, dim3(rows, Y), dim3(256), 0, st, in, c->CUP[i], dout, c->UPPART, B, C, Tin, s, fk, type,
                               c->cfg.upsample_activation, c->cfg.leaky_alpha, tchunk);
            WN_LAUNCH_CHECK(c);
            hipLaunchKernelGGL(wn_up_bwd_params_reduce, dim3(ne), dim3(256), 0, st, c->UPPART, nblk, ne, nk, grads + c->up_k[i], grads + c->up_b[i]);
            WN_LAUNCH_CHECK(c);
        }
        if (i > 0) {
            float* din = c->DCUP[i & 1];
            const int64_t ni = (int64_t)B * C * Tin;
            if (v1) {
            int GL = 1; while (GL < s && GL < 64) GL <<= 1;
            hipLaunchKernelGGL(wn_up_bwd_input, dim3(cdiv(ni, 256 / GL)), dim3(256), 0, st, c->CUP[i], dout, din, c->params_dev + c->up_k[i],
                               B, C, Tin, s, fk, type, c->cfg.upsample_activation, c->cfg.leaky_alpha, GL);
            } else {
                const int halo = (type == 2) ? 1 : 0;
                int TB = std::min(256, 8192 / (fk * s) - 2 * halo);
                if (TB < 1) WN_FAIL(c, WN_E_UNSUPPORTED, "upsample scale %d x freq kernel %d too large for the LDS stage", s, fk);
                TB = std::min(TB, Tin);
                const size_t lds = ((size_t)fk * (TB + 2 * halo) * s + nk) * 4;
                hipLaunchKernelGGL(wn_up_bwd_input2, dim3(cdiv(Tin, TB), B * C), dim3(256), lds, st, c->CUP[i], dout, din, c->params_dev + c->up_k[i],
                                   B, C, Tin, s, fk, type, c->cfg.upsample_activation, c->cfg.leaky_alpha, TB);
            }
            WN_LAUNCH_CHECK(c);
            dout = din;
        }
        Tout = Tin;
    }
    return WN_OK;
}

// =================================================================================== global conditioning
// wavenet.py:669-678 (embedding lookup + broadcast over time), modules.py:499-508 (z += W_g^T g + b_g).  g is constant over
// time, so its contribution is a per-utterance bias of the gate pre-activation: gbias[l][b][:] = b1sum[l] + W_g[l]^T g_b + b_g[l].
__global__ void wn_gvec_kernel(const float* __restrict__ params, int64_t emb_off, const int32_t* __restrict__ ids, float* __restrict__ gvec,
                               int B, int gin, int n_speakers) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * gin) return;
    const int b = i / gin, k = i - b * gin;
    int id = ids[b]; id = id < 0 ? 0 : (id >= n_speakers ? n_speakers - 1 : id);
    gvec[i] = params[emb_off + (int64_t)id * gin + k];
}
struct GinOff { int64_t k[32], b[32]; };
__global__ void wn_gbias_kernel(const float* __restrict__ params, const float* __restrict__ b1sum, const float* __restrict__ gvec,
                                float* __restrict__ gbias, int B, int G, int gin, GinOff o) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y, l = blockIdx.z;
    if (g >= G) return;
    float a = b1sum[(size_t)l * G + g] + params[o.b[l] + g];
    const float* W = params + o.k[l];
    for (int k = 0; k < gin; ++k) a += gvec[b * gin + k] * W[(int64_t)k * G + g];
    gbias[((size_t)l * B + b) * G + g] = a;
}
int wn_gbias_fwd(wn_ctx* c, int B, hipStream_t st) {
    if (c->gin <= 0) return WN_OK;
    if (c->cfg.use_speaker_embedding)
        hipLaunchKernelGGL(wn_gvec_kernel, dim3(cdiv(B * c->gin, 256)), dim3(256), 0, st, c->params_dev, c->emb_off, c->gids, c->gvec, B, c->gin, c->cfg.n_speakers);
    GinOff o; for (int l = 0; l < c->L; ++l) { o.k[l] = c->lay[l].gin_k; o.b[l] = c->lay[l].gin_b; }
    hipLaunchKernelGGL(wn_gbias_kernel, dim3(cdiv(c->G, 256), B, c->L), dim3(256), 0, st, c->params_dev, c->b1sum, c->gvec, c->gbias, B, c->G, c->gin, o);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
// backward: colsum[l][b][g] = sum_t dz_l[b,t,g];  d b_g[l] = sum_b colsum;  d W_g[l][k][g] = sum_b g_b[k] colsum[l][b][g];
// d g_b[k] = sum_l sum_g W_g[l][k][g] colsum[l][b][g]  (scattered into the embedding row of the utterance's speaker).
__global__ __launch_bounds__(256) void wn_colsum_kernel(const bf16_t* __restrict__ DZ, float* __restrict__ colsum, int64_t NT, int B, int T, int G) {
    // block = (8-channel group, utterance, layer); threads stride over time, LDS tree at the end
    const int c8 = blockIdx.x, b = blockIdx.y, l = blockIdx.z;
    const bf16_t* base = DZ + ((size_t)l * NT + (size_t)b * T) * G + c8 * 8;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = threadIdx.x; t < T; t += 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(base + (size_t)t * G);
        a[0] += bf2f((bf16_t)(v.x & 0xffff)); a[1] += bf2f((bf16_t)(v.x >> 16)); a[2] += bf2f((bf16_t)(v.y & 0xffff)); a[3] += bf2f((bf16_t)(v.y >> 16));
        a[4] += bf2f((bf16_t)(v.z & 0xffff)); a[5] += bf2f((bf16_t)(v.z >> 16)); a[6] += bf2f((bf16_t)(v.w & 0xffff)); a[7] += bf2f((bf16_t)(v.w >> 16));
    }
    __shared__ float red[4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { float s = a[e]; for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o); if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][e] = s; }
    __syncthreads();
    if (threadIdx.x < 8) colsum[((size_t)l * B + b) * G + c8 * 8 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void wn_gin_wgrad_kernel(const float* __restrict__ colsum, const float* __restrict__ gvec, float* __restrict__ grads, int B, int G, int gin, GinOff o, int has_bias) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x, l = blockIdx.z;
    const int k = blockIdx.y;          // k == gin: the bias row
    if (g >= G) return;
    float a = 0.0f;
    for (int b = 0; b < B; ++b) a += (k < gin ? gvec[b * gin + k] : 1.0f) * colsum[((size_t)l * B + b) * G + g];
    if (k < gin) grads[o.k[l] + (int64_t)k * G + g] = a;
    else if (has_bias) grads[o.b[l] + g] = a;
}
__global__ void wn_gin_dg_kernel(const float* __restrict__ params, const float* __restrict__ colsum, const int32_t* __restrict__ ids,
                                 float* __restrict__ grads, int64_t emb_off, int B, int G, int gin, int L, int n_speakers, GinOff o) {
    // one wave per (utterance, k): sum over layers and gate channels, then one atomic into the speaker's embedding row
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= B * gin) return;
    const int b = w / gin, k = w - b * gin;
    float a = 0.0f;
    for (int l = 0; l < L; ++l) {
        const float* W = params + o.k[l] + (int64_t)k * G;
        const float* cs = colsum + ((size_t)l * B + b) * G;
        for (int g = lane; g < G; g += 64) a += W[g] * cs[g];
    }
    for (int s = 32; s > 0; s >>= 1) a += __shfl_down(a, s);
    if (lane == 0) { int id = ids[b]; id = id < 0 ? 0 : (id >= n_speakers ? n_speakers - 1 : id); unsafeAtomicAdd(&grads[emb_off + (int64_t)id * gin + k], a); }
}
// have_colsum: c->colsum already holds the per-utterance column sums (the fp32 backward of wn_f32.hip writes them layer by layer)
int wn_gin_bwd(wn_ctx* c, float* grads, hipStream_t st, bool have_colsum) {
    if (c->gin <= 0) return WN_OK;
    const int B = c->fB, G = c->G, L = c->L;
    if (!have_colsum) hipLaunchKernelGGL(wn_colsum_kernel, dim3(G / 8, B, L), dim3(256), 0, st, c->DZ, c->colsum, c->NT, B, c->fT, G);
    GinOff o; for (int l = 0; l < L; ++l) { o.k[l] = c->lay[l].gin_k; o.b[l] = c->lay[l].gin_b; }
    hipLaunchKernelGGL(wn_gin_wgrad_kernel, dim3(cdiv(G, 256), c->gin + 1, L), dim3(256), 0, st, c->colsum, c->gvec, grads, B, G, c->gin, o, c->lbias ? 1 : 0);
    if (c->cfg.use_speaker_embedding)
        hipLaunchKernelGGL(wn_gin_dg_kernel, dim3(cdiv(B * c->gin, 4)), dim3(256), 0, st, c->params_dev, c->colsum, c->gids, grads, c->emb_off, B, G, c->gin, L, c->cfg.n_speakers, o);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
