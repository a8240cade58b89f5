// MFMA tile engine for the dense (training-time) contractions of the WaveNet stack on gfx950.
//
// Every dense op of the residual stack is computed in the TRANSPOSED form
//        Out^T[m, t] = sum_k  Wpk[m, k] * Act[t, k]
//   * activations live in HBM time-major with channels contiguous, Act[b*T + t][c] bf16, so a dilated
//     tap is a pure ROW shift (rows are 128..1024 B, always 16-B aligned) and every HBM access is a
//     full-row burst;
//   * Wpk (A operand) is pre-packed once per optimiser step in MFMA *fragment order*
//     [mtile][kstep][lane][8]  (v_mfma_f32_32x32x16_bf16: lane&31 = row m, lane>>5 = k-half), so a
//     wave's A fragment is one fully coalesced 1-KiB global load served from L2;
//   * Act tiles (B operand) are staged global -> VGPR -> LDS as [t][<=64 ch] rows with a 144-B row
//     stride (conflict-free ds_read_b128 for the 16-lane groups of gfx950) and read as fragments
//     lane&31 = time row, lane>>5 = k-half;
//   * the accumulator comes out with lane&31 = time, registers = 4-channel groups, which is exactly
//     the shape the fused epilogues want (gate pairs sit in the same lane; 8-B channel-contiguous
//     stores into the [t][c] layout).
// 64-wide wavefronts, 4 waves per workgroup, 2 workgroups per CU (36 KiB LDS each).
#pragma once
#include "wn_common.h"

struct SrcSeg {
    const bf16_t* base;   // [rows][ld] bf16, row = b*T + t
    int32_t ld;           // row stride (elements)
    int32_t col0;         // first channel of the segment
    int32_t nk;           // channels (multiple of 16)
    int32_t shift;        // time shift: source row t + shift, zero outside [0,T)
    int32_t dropout;      // 1: apply the layer's dropout mask while staging
};

struct EpiArgs {
    void* out0; void* out1;
    const void* in0; const void* in1;
    const float* bias;
    int32_t ld_out0, ld_out1, ld_in0;
    float scale;
    int32_t M_valid;
    int32_t relu;
    int32_t GH;
};

struct GemmArgs {
    const bf16_t* Apk;
    int32_t ksteps_total;       // K/16 of the packed matrix (row pitch in k-steps)
    int32_t mblocks;            // grid decode
    int32_t nseg; SrcSeg seg[4];
    int32_t nrep; int64_t rep_stride;     // segment list repeated nrep times, bases advanced by rep_stride elements
    int32_t B, T;
    int32_t tiles_per_utt, ntiles;
    uint32_t key_lo, key_hi, thresh16; float keep_scale; int32_t drop_ld;   // dropout mask spec (row pitch of the dropped tensor)
    EpiArgs e;
};

enum { EPI_GATE = 0, EPI_STORE_BF16 = 1, EPI_STORE_F32_BOT = 2, EPI_DGATE = 3, EPI_MASK_STORE = 4, EPI_DX = 5 };

#define TILE_LDS_STRIDE 72   // halfs per staged row: 64 channels + 8 pad (144 B)

__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1); exact to ~1 ulp of __expf, saturates cleanly
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}
__device__ __forceinline__ float fast_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ bool drop_keep(uint32_t key_lo, uint32_t key_hi, uint32_t thresh16, uint32_t e) {
    uint32_t w = wn_drop_word(key_lo, key_hi, e >> 1);
    uint32_t bits = (e & 1u) ? (w >> 16) : (w & 0xffffu);
    return bits >= thresh16;
}

// apply dropout to 8 consecutive bf16 elements starting at flat element index e0 (e0 % 8 == 0)
__device__ __forceinline__ uint4 drop8(uint4 v, uint32_t key_lo, uint32_t key_hi, uint32_t thresh16, float ks, uint32_t e0) {
    uint32_t in[4] = {v.x, v.y, v.z, v.w};
    uint32_t out[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        uint32_t w = wn_drop_word(key_lo, key_hi, (e0 >> 1) + p);
        float lo = bf2f((bf16_t)(in[p] & 0xffffu)), hi = bf2f((bf16_t)(in[p] >> 16));
        lo = ((w & 0xffffu) >= thresh16) ? lo * ks : 0.0f;
        hi = ((w >> 16) >= thresh16) ? hi * ks : 0.0f;
        out[p] = pack_bf2(lo, hi);
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}

template <int MT, int NT, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64) void wn_gemm_tile_kernel(const GemmArgs a) {
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int NROWS = WN * NT * 32;
    constexpr int PIECES = NROWS * 8 / NTHREADS;
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][NROWS * TILE_LDS_STRIDE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware decode: the `mblocks` workgroups that share one activation tile get ids that are
    // congruent mod 8 (same XCD => the tile is fetched from HBM once and hit in that XCD's L2).
    const int id = blockIdx.x;
    const int xcd = id & 7, q = id >> 3;
    const int mblk = q % a.mblocks;
    const int tile = (q / a.mblocks) * 8 + xcd;
    if (tile >= a.ntiles) return;
    const int b = tile / a.tiles_per_utt;
    const int t0 = (tile - b * a.tiles_per_utt) * NROWS;
    const int T = a.T;
    const int64_t rowbase = (int64_t)b * T;

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int mtile0 = (mblk * WM + wm) * MT;

    // ---- chunk iterator state
    int rep = 0, sg = 0, cc = 0;          // current chunk to STAGE
    int kstep_base = 0;                    // k-step index of the chunk being COMPUTED
    const int total_chunks_per_rep = [&] { int n = 0; for (int s = 0; s < a.nseg; ++s) n += (a.seg[s].nk + 63) >> 6; return n; }();
    const int nchunks = total_chunks_per_rep * a.nrep;

    uint4 st[PIECES];
    int st_kc = 0;

    auto stage_load = [&]() {
        const SrcSeg& s = a.seg[sg];
        const int kc = min(64, s.nk - cc * 64);
        st_kc = kc;
        const bf16_t* base = s.base + (int64_t)rep * a.rep_stride;
        const int col = s.col0 + cc * 64;
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const int piece = tid + p * NTHREADS;
            const int row = piece >> 3, c16 = piece & 7;
            const int t = t0 + row, ts = t + s.shift;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c16 * 8 < kc && t < T && ts >= 0 && ts < T) {
                const int64_t r = rowbase + ts;
                v = *reinterpret_cast<const uint4*>(base + r * s.ld + col + c16 * 8);
                if (s.dropout)
                    v = drop8(v, a.key_lo, a.key_hi, a.thresh16, a.keep_scale,
                              (uint32_t)(r * a.drop_ld + col + c16 * 8));
            }
            st[p] = v;
        }
        // advance iterator
        ++cc;
        if (cc * 64 >= s.nk) { cc = 0; ++sg; if (sg == a.nseg) { sg = 0; ++rep; } }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) {
            const int piece = tid + p * NTHREADS;
            const int row = piece >> 3, c16 = piece & 7;
            *reinterpret_cast<uint4*>(&lds[buf][row * TILE_LDS_STRIDE + c16 * 8]) = st[p];
        }
    };

    stage_load();
    int cur_kc = st_kc;
    stage_store(0);
    __syncthreads();

    const bf16_t* Arow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) Arow[i] = a.Apk + ((int64_t)(mtile0 + i) * a.ksteps_total * 64 + lane) * 8;

    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const bool more = (ch + 1 < nchunks);
        if (more) stage_load();
        const int next_kc = st_kc;
        // ---- compute current chunk from lds[buf]
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks * 16 < cur_kc) {
                bf16x8_t af[MT], bfr[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    af[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Arow[i] + (int64_t)(kstep_base + ks) * 512));
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    bfr[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(
                        &lds[buf][((wn * NT + j) * 32 + (lane & 31)) * TILE_LDS_STRIDE + ks * 16 + (lane >> 5) * 8]));
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        kstep_base += cur_kc >> 4;
        if (more) stage_store(buf ^ 1);
        cur_kc = next_kc;
        __syncthreads();
    }

    // ---- epilogue.  acc[i][j][r]: time t = tbase + j*32 + (lane&31); channel m = (mtile0+i)*32 + 8*(r>>2) + 4*(lane>>5) + (r&3)
    const EpiArgs& e = a.e;
    const int h = lane >> 5;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = t0 + (wn * NT + j) * 32 + (lane & 31);
        if (t >= T) continue;
        const int64_t row = rowbase + t;
        if constexpr (EPI == EPI_GATE) {
            static_assert(EPI != EPI_GATE || MT == 2, "gate epilogue pairs m-tiles");
            const int gblk = (mtile0 >> 1) * 32;
            bf16_t* TS = (bf16_t*)e.out0; bf16_t* U = (bf16_t*)e.out1;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int g = gblk + qd * 8 + h * 4;
                float ta[4], sgm[4], u[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float za = acc[0][j][qd * 4 + r] + e.bias[g + r];
                    float zb = acc[MT - 1][j][qd * 4 + r] + e.bias[e.GH + g + r];
                    ta[r] = fast_tanh(za); sgm[r] = fast_sigmoid(zb); u[r] = ta[r] * sgm[r];
                }
                *reinterpret_cast<uint2*>(TS + row * e.ld_out0 + g) = make_uint2(pack_bf2(ta[0], ta[1]), pack_bf2(ta[2], ta[3]));
                *reinterpret_cast<uint2*>(TS + row * e.ld_out0 + e.GH + g) = make_uint2(pack_bf2(sgm[0], sgm[1]), pack_bf2(sgm[2], sgm[3]));
                *reinterpret_cast<uint2*>(U + row * e.ld_out1 + g) = make_uint2(pack_bf2(u[0], u[1]), pack_bf2(u[2], u[3]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int m = (mtile0 + i) * 32 + qd * 8 + h * 4;
                    if (m >= e.M_valid) continue;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[i][j][qd * 4 + r];
                    if constexpr (EPI == EPI_STORE_BF16) {
                        if (e.bias) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] += e.bias[m + r];
                        }
                        if (e.in0) {
                            uint2 x = *reinterpret_cast<const uint2*>((const bf16_t*)e.in0 + row * e.ld_in0 + m);
                            v[0] += bf2f((bf16_t)(x.x & 0xffff)); v[1] += bf2f((bf16_t)(x.x >> 16));
                            v[2] += bf2f((bf16_t)(x.y & 0xffff)); v[3] += bf2f((bf16_t)(x.y >> 16));
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[r] *= e.scale; if (e.relu) v[r] = fmaxf(v[r], 0.0f); }
                        const uint2 pk = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                        *reinterpret_cast<uint2*>((bf16_t*)e.out0 + row * e.ld_out0 + m) = pk;
                        if (e.out1) {
                            // dropout of the NEXT layer's conv input applied once, here (tf.layers.dropout,
                            // modules.py:484): x~ = bf16(bf16(x) * 1/(1-p)) or 0; the residual path keeps out0.
                            const uint32_t e0 = (uint32_t)(row * a.drop_ld + m);
                            const uint32_t w0 = wn_drop_word(a.key_lo, a.key_hi, e0 >> 1), w1 = wn_drop_word(a.key_lo, a.key_hi, (e0 >> 1) + 1);
                            const float x0 = bf2f((bf16_t)(pk.x & 0xffff)), x1 = bf2f((bf16_t)(pk.x >> 16));
                            const float x2 = bf2f((bf16_t)(pk.y & 0xffff)), x3 = bf2f((bf16_t)(pk.y >> 16));
                            const float d0 = ((w0 & 0xffffu) >= a.thresh16) ? x0 * a.keep_scale : 0.0f;
                            const float d1 = ((w0 >> 16) >= a.thresh16) ? x1 * a.keep_scale : 0.0f;
                            const float d2 = ((w1 & 0xffffu) >= a.thresh16) ? x2 * a.keep_scale : 0.0f;
                            const float d3 = ((w1 >> 16) >= a.thresh16) ? x3 * a.keep_scale : 0.0f;
                            *reinterpret_cast<uint2*>((bf16_t*)e.out1 + row * e.ld_out1 + m) = make_uint2(pack_bf2(d0, d1), pack_bf2(d2, d3));
                        }
                    } else if constexpr (EPI == EPI_STORE_F32_BOT) {
                        float* out = (float*)e.out0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (m + r < e.M_valid) {
                                float y = v[r] * e.scale + (e.bias ? e.bias[m + r] : 0.0f);
                                out[((int64_t)b * e.M_valid + (m + r)) * T + t] = y;
                            }
                        }
                    } else if constexpr (EPI == EPI_DGATE) {
                        const bf16_t* TS = (const bf16_t*)e.in0;
                        uint2 xa = *reinterpret_cast<const uint2*>(TS + row * e.ld_in0 + m);
                        uint2 xb = *reinterpret_cast<const uint2*>(TS + row * e.ld_in0 + e.GH + m);
                        float ta[4] = {bf2f((bf16_t)(xa.x & 0xffff)), bf2f((bf16_t)(xa.x >> 16)), bf2f((bf16_t)(xa.y & 0xffff)), bf2f((bf16_t)(xa.y >> 16))};
                        float sg[4] = {bf2f((bf16_t)(xb.x & 0xffff)), bf2f((bf16_t)(xb.x >> 16)), bf2f((bf16_t)(xb.y & 0xffff)), bf2f((bf16_t)(xb.y >> 16))};
                        float da[4], db[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            da[r] = v[r] * sg[r] * (1.0f - ta[r] * ta[r]);
                            db[r] = v[r] * ta[r] * sg[r] * (1.0f - sg[r]);
                        }
                        bf16_t* DZ = (bf16_t*)e.out0;
                        *reinterpret_cast<uint2*>(DZ + row * e.ld_out0 + m) = make_uint2(pack_bf2(da[0], da[1]), pack_bf2(da[2], da[3]));
                        *reinterpret_cast<uint2*>(DZ + row * e.ld_out0 + e.GH + m) = make_uint2(pack_bf2(db[0], db[1]), pack_bf2(db[2], db[3]));
                    } else if constexpr (EPI == EPI_MASK_STORE) {
                        uint2 x = *reinterpret_cast<const uint2*>((const bf16_t*)e.in0 + row * e.ld_in0 + m);
                        float ref[4] = {bf2f((bf16_t)(x.x & 0xffff)), bf2f((bf16_t)(x.x >> 16)), bf2f((bf16_t)(x.y & 0xffff)), bf2f((bf16_t)(x.y >> 16))};
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (ref[r] > 0.0f) ? v[r] * e.scale : 0.0f;
                        *reinterpret_cast<uint2*>((bf16_t*)e.out0 + row * e.ld_out0 + m) =
                            make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    } else if constexpr (EPI == EPI_DX) {
                        if (a.thresh16 != 0) {
                            const uint32_t e0 = (uint32_t)(row * a.drop_ld + m);
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                v[r] = drop_keep(a.key_lo, a.key_hi, a.thresh16, e0 + r) ? v[r] * a.keep_scale : 0.0f;
                        }
                        if (e.in0) {
                            uint2 x = *reinterpret_cast<const uint2*>((const bf16_t*)e.in0 + row * e.ld_in0 + m);
                            v[0] += bf2f((bf16_t)(x.x & 0xffff)); v[1] += bf2f((bf16_t)(x.x >> 16));
                            v[2] += bf2f((bf16_t)(x.y & 0xffff)); v[3] += bf2f((bf16_t)(x.y >> 16));
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] *= e.scale;
                        *reinterpret_cast<uint2*>((bf16_t*)e.out0 + row * e.ld_out0 + m) =
                            make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    }
                }
            }
        }
    }
}

// Host-side launcher: picks the workgroup shape from M.
template <int EPI>
static inline int wn_launch_gemm(wn_ctx* ctx, GemmArgs& a, int M, hipStream_t st) {
    // all shapes stage 128 time rows per workgroup (36 KiB LDS, 2 workgroups per CU)
    const int nrows = 128;
    const int mrows = (M % 128 == 0) ? 128 : (M % 64 == 0) ? 64 : 32;
    if (EPI == EPI_GATE && M % 64 != 0) WN_FAIL(ctx, WN_E_SHAPE, "gate GEMM needs gate_channels %% 64 == 0 (got M=%d)", M);
    a.mblocks = M / mrows;
    a.tiles_per_utt = cdiv(a.T, nrows);
    a.ntiles = a.tiles_per_utt * a.B;
    const int tile_groups = cdiv(a.ntiles, 8);
    const int grid = tile_groups * a.mblocks * 8;
    if (mrows == 128) hipLaunchKernelGGL((wn_gemm_tile_kernel<2, 2, 2, 2, EPI>), dim3(grid), dim3(256), 0, st, a);
    else if (mrows == 64) hipLaunchKernelGGL((wn_gemm_tile_kernel<2, 1, 1, 4, EPI>), dim3(grid), dim3(256), 0, st, a);
    else {
        if constexpr (EPI == EPI_GATE) { WN_FAIL(ctx, WN_E_SHAPE, "gate GEMM M=%d", M); }
        else hipLaunchKernelGGL((wn_gemm_tile_kernel<1, 1, 1, 4, EPI>), dim3(grid), dim3(256), 0, st, a);
    }
    WN_LAUNCH_CHECK(ctx);
    return WN_OK;
}
