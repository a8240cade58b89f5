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
#include <type_traits>

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
    int32_t bias_bstride;       // floats between the bias vectors of consecutive utterances (0: one vector; > 0: global conditioning)
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
    int32_t B, T;               // utterances of THIS launch (a part of the batch) and samples per utterance
    int32_t b0;                 // first utterance of the part: rows, bias vectors and dropout indices are absolute (b0 + local b)
    int32_t tiles_per_utt, ntiles;
    uint32_t key_lo, key_hi, thresh16; float keep_scale; int32_t drop_ld;   // dropout mask spec (row pitch of the dropped tensor)
    const bf16_t* zero;         // >= 16 B of zeros in device memory (source of out-of-range rows for the LDS-DMA kernel)
    unsigned long long* kprof;  // wn_profile: {min over workgroups of the start, max of the end} of THIS launch in 100 MHz wall-clock ticks (null: off)
    unsigned long long* kclk;   // wn_profile: {shader cycles (s_memtime), 100 MHz ticks} workgroup 0 of THIS launch spent in the kernel: the shader clock
                                // the launch actually ran at = 100 MHz x cycles / ticks (null: off) -- the chip runs this step at its power limit
    int32_t stagger;            // shader cycles the second-resident workgroups of the first round wait before starting (0: off)
    int32_t xcd_span;           // LDS-DMA kernels: > 0 = XCD x owns the contiguous tiles [x * xcd_span, (x + 1) * xcd_span); 0 = tiles interleaved over XCDs
    int32_t kil;                // block size of the K-interleaved pack (32: this header's TAPS kernels; 64: wn_gemm8p_kernel, wn_tile8p.h)
    int32_t taps;               // 3: seg[0..2] are the dilated taps of ONE tensor (same base / ld / nk), staged interleaved in BK-channel blocks
                                //    (tap0, tap1, tap2 of block 0, then of block 1, ...: the K order of a `kil` pack); 0: segments one after the other
    EpiArgs e;
};

enum { EPI_GATE = 0, EPI_STORE_BF16 = 1, EPI_STORE_F32_BOT = 2, EPI_DGATE = 3, EPI_MASK_STORE = 4, EPI_DX = 5 };

#define TILE_LDS_STRIDE 72   // halfs per staged row: 64 channels + 8 pad (144 B)

__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1) on v_exp_f32 + v_rcp_f32 (1 ulp each; the result is rounded to bf16 anyway); saturates
    // cleanly: exp -> inf gives rcp 0 -> +1, exp -> 0 gives -1.  An IEEE divide here costs ~10 VALU instructions per element.
    float e = __builtin_amdgcn_exp2f(x * 2.885390082f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.442695041f)); }
// The forward saves sigmoid(b) and u = tanh(a) * sigmoid(b) (both bf16); the backward needs tanh(a) only inside (1 - tanh^2), the other
// factor being tanh * sigmoid = u exactly.  tanh = u / sigmoid: well conditioned wherever the gradient is not already ~0 (sigmoid is never
// near 0 unless the unit is shut, where d a = g * sigmoid * (...) vanishes with it); clamped to [-1, 1]; 0 when sigmoid underflowed.
// Saves one of the three [rows][G/2] tensors the gate would write (and the backward read): 512 B of 1.5 KB per row and layer at R = 256.
__device__ __forceinline__ float gate_tanh_from(float u, float s) {
    const float t = u * __builtin_amdgcn_rcpf(s);
    return s > 0.0f ? fminf(fmaxf(t, -1.0f), 1.0f) : 0.0f;
}
// d a = g * sigmoid * (1 - tanh^2),  d b = g * tanh * sigmoid * (1 - sigmoid) = g * u * (1 - sigmoid)   (modules.py:510 differentiated).
// The one contraction-prone expression is written as an explicit fma so that every instantiation rounds identically.
__device__ __forceinline__ void gate_backward(float g, float u, float s, float& da, float& db) {
    const float ta = gate_tanh_from(u, s);
    da = g * s * __builtin_fmaf(-ta, ta, 1.0f);
    db = g * u * (1.0f - s);
}

__device__ __forceinline__ bool drop_keep(uint32_t key_lo, uint32_t key_hi, uint32_t thresh16, uint32_t e) {
    uint32_t w = wn_drop_word(key_lo, key_hi, e >> 1);
    uint32_t bits = (e & 1u) ? (w >> 16) : (w & 0xffffu);
    return bits >= thresh16;
}

// apply dropout to 8 consecutive bf16 elements starting at flat element index e0 (e0 % 8 == 0)
__device__ __forceinline__ uint4 drop8(uint4 v, uint32_t key_lo, uint32_t key_hi, uint32_t thresh16, float ks, uint32_t e0) {
    uint32_t in[4] = {v.x, v.y, v.z, v.w};
    uint32_t out[4], wq[4];
    wn_drop_quad(key_lo, key_hi, e0 >> 2, wq[0], wq[1]); wn_drop_quad(key_lo, key_hi, (e0 >> 2) + 1, wq[2], wq[3]);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const uint32_t w = wq[p];
        float lo = bf2f((bf16_t)(in[p] & 0xffffu)), hi = bf2f((bf16_t)(in[p] >> 16));
        lo = ((w & 0xffffu) >= thresh16) ? lo * ks : 0.0f;
        hi = ((w >> 16) >= thresh16) ? hi * ks : 0.0f;
        out[p] = pack_bf2(lo, hi);
    }
    return make_uint4(out[0], out[1], out[2], out[3]);
}

// Fused epilogues, shared by both main loops.  acc[i][j][r]: time t = t0w + j*32 + (lane&31);
// channel m = (mtile0+i)*32 + 8*(r>>2) + 4*(lane>>5) + (r&3).
template <int MT, int NT, int EPI>
__device__ __forceinline__ void wn_tile_epilogue(const GemmArgs& a, f32x16_t (&acc)[MT][NT], const int mtile0, const int t0w,
                                                 const int b, const int T, const int64_t rowbase, const int lane) {
    const EpiArgs& e = a.e;
    const int h = lane >> 5;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = t0w + j * 32 + (lane & 31);
        if (t >= T) continue;
        const int64_t row = rowbase + t;
        if constexpr (EPI == EPI_GATE) {
            static_assert(EPI != EPI_GATE || MT == 2, "gate epilogue pairs m-tiles");
            const int gblk = (mtile0 >> 1) * 32;
            const float* const gb = e.bias + (int64_t)b * e.bias_bstride;
            bf16_t* TS = (bf16_t*)e.out0; bf16_t* U = (bf16_t*)e.out1;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int g = gblk + qd * 8 + h * 4;
                float ta[4], sgm[4], u[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float za = acc[0][j][qd * 4 + r] + gb[g + r];
                    float zb = acc[MT - 1][j][qd * 4 + r] + gb[e.GH + g + r];
                    ta[r] = fast_tanh(za); sgm[r] = fast_sigmoid(zb); u[r] = ta[r] * sgm[r];
                }
                // saved for backward: sigmoid and the gate output u = tanh * sigmoid; tanh itself is recovered as u / sigmoid (EPI_DGATE)
                *reinterpret_cast<uint2*>(TS + row * e.ld_out0 + g) = make_uint2(pack_bf2(sgm[0], sgm[1]), pack_bf2(sgm[2], sgm[3]));
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
                        uint2 xa = *reinterpret_cast<const uint2*>((const bf16_t*)e.in1 + row * e.ld_in0 + m);      // u = tanh * sigmoid
                        uint2 xb = *reinterpret_cast<const uint2*>((const bf16_t*)e.in0 + row * e.ld_in0 + m);      // sigmoid
                        float uu[4] = {bf2f((bf16_t)(xa.x & 0xffff)), bf2f((bf16_t)(xa.x >> 16)), bf2f((bf16_t)(xa.y & 0xffff)), bf2f((bf16_t)(xa.y >> 16))};
                        float sg[4] = {bf2f((bf16_t)(xb.x & 0xffff)), bf2f((bf16_t)(xb.x >> 16)), bf2f((bf16_t)(xb.y & 0xffff)), bf2f((bf16_t)(xb.y >> 16))};
                        float da[4], db[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) gate_backward(v[r], uu[r], sg[r], da[r], db[r]);
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
    const int bl = tile / a.tiles_per_utt;
    const int b = bl + a.b0;
    const int t0 = (tile - bl * a.tiles_per_utt) * NROWS;
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

    wn_tile_epilogue<MT, NT, EPI>(a, acc, mtile0, t0 + wn * NT * 32, b, T, rowbase, lane);
}

// ================================================================================================
// Main loop v2: BOTH operands through LDS, filled by LDS-DMA (global_load_lds_dwordx4), NBUF-deep ring
// with counted vmcnt + raw s_barrier, 8 waves per workgroup (1 workgroup per CU).
//   * workgroup tile  MTILE = WM*MT*32 output channels  x  TTILE = WN*NT*32 time rows, K-chunk BK channels;
//     at 256 x 128 x 64 the L2->CU traffic is 48 B/clk/CU at full MFMA rate (a 128x128 tile needs 64 B/clk,
//     which is the L1/TA limit -- the reason v1, which also re-read every A fragment per wave, stalls);
//   * A: the packed weights are already in MFMA fragment order, one fragment (32 rows x 16 k) = 1 KiB = one
//     wave-wide LDS-DMA; LDS image [mtile][kstep][lane][8], read back lane-linear (conflict free);
//   * B: activation rows [t][BK channels] (BK*2 bytes per row), lane-linear LDS image with the 16-B slot
//     index XOR-swizzled by the row (applied on the per-lane SOURCE address, and again on the ds_read_b128),
//     so the 16-lane groups of a fragment read hit 16 different 16-B slots of the 256-B bank line;
//     out-of-range taps (causal zero padding, utterance boundaries, rows >= T, channels >= nk) read a
//     device zero page instead, so the number of DMAs per chunk is constant (counted vmcnt);
//   * ring buffers are indexed at compile time (loop unrolled by NBUF): with a runtime index hipcc cannot
//     prove the DMA destination and the ds_reads disjoint and drains vmcnt(0) before every read.
// One wave-wide LDS-DMA: lane i's 16 B from `gsrc` land at LDS byte address lds_addr (wave-uniform) + 16*i.  Issued as inline
// asm on purpose: hipcc models the builtin as a FLAT access, and while one is in flight (always, here) its waitcnt pass turns
// every wait on an LDS read into lgkmcnt(0) -- i.e. it serialises "read fragments" and "multiply" -- and inserts vmcnt(0)
// before LDS reads it cannot prove disjoint from the DMA target.  The ordering the DMA needs is enforced by hand in the main
// loops (counted vmcnt + s_barrier before any read of the buffer); the "memory" clobber keeps LDS accesses on their side.
__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory", "m0");
}
// SGPR-base form: address = sbase (wave-uniform, 64 bit) + voff (per lane, 32 bit).  No per-lane 64-bit address arithmetic: the
// weight panel's DMA source advances by a scalar add per chunk.
__device__ __forceinline__ void lds_dma16_s(uint64_t sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(size_t)(const __attribute__((address_space(3))) char*)p;
}

constexpr int lds_gemm_bytes(int MT, int NT, int WM, int WN, int BK, int NBUF) {
    const int ring = NBUF * (WM * MT * 32 + WN * NT * 32) * BK * 2, epi = WN * 32 * (WM * MT * 32 * 4 + 16);
    return ring > epi ? ring : epi;
}
// minimum waves per SIMD for __launch_bounds__: two 8-wave workgroups per CU when LDS allows it
constexpr int lds_gemm_min_waves(int MT, int NT, int WM, int WN, int BK, int NBUF) {
    return ((160 * 1024 / lds_gemm_bytes(MT, NT, WM, WN, BK, NBUF) >= 2 && WM * WN <= 8) ? 2 : 1) * WM * WN / 4;
}
template <int MT, int NT, int WM, int WN, int BK, int NBUF>
struct LdsGemmCfg {
    static constexpr int NW = WM * WN;
    static constexpr int MTILE = WM * MT * 32, TTILE = WN * NT * 32;
    static constexpr int KS = BK / 16;
    static constexpr int A_BYTES = MTILE * BK * 2, B_BYTES = TTILE * BK * 2, BUF_BYTES = A_BYTES + B_BYTES;
    // ring layout: [B slot 0 .. B slot NBUF-1 | A slot 0 .. A slot NBUF-1].  The activation (B) slots come first so that every B
    // fragment address of every slot is one VGPR base + a 16-bit immediate (< 64 KiB): with [A | B] per slot the last slot's reads
    // lay beyond the immediate range and cost one address register per fragment.
    static constexpr int b_base(int buf) { return buf * B_BYTES; }
    static constexpr int a_base(int buf) { return NBUF * B_BYTES + buf * A_BYTES; }
    static constexpr int A_INSTR = A_BYTES / 1024, B_INSTR = B_BYTES / 1024;
    static constexpr int A_PW = A_INSTR / NW, B_PW = B_INSTR / NW, LPC = A_PW + B_PW;   // DMAs per wave per chunk
    static constexpr int RB = BK * 2, SPR = RB / 16, RPL = 256 / RB;                      // row bytes, 16-B slots per row, rows per bank line
    static_assert(A_INSTR % NW == 0 && B_INSTR % NW == 0, "DMA pieces must divide over the waves");
    static_assert(BK == 32 || BK == 64, "BK");
    static constexpr int EPI_PITCH = MTILE * 4 + 16;            // fp32 staging row of the epilogue (+16: conflict-free ds_write_b128)
    static constexpr int EPI_ROWS = WN * 32;
    static constexpr int LDS_BYTES = lds_gemm_bytes(MT, NT, WM, WN, BK, NBUF);
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

// The kernel BODY is a device function of (arguments, LDS arena, workgroup id): wn_gemm_lds_kernel is the one-launch wrapper, and a
// grid that holds the workgroups of several launches (tools/: the fused-pair grid of round 3, the persistent chain prototype) can call it.
template <int MT, int NT, int WM, int WN, int BK, int NBUF, int EPI, int PIPE_ = 1, int TAPS = 0>
__device__ __forceinline__ void wn_gemm_lds_body(const GemmArgs& a, char* const lds, const int wg_id) {
    using Cfg = LdsGemmCfg<MT, NT, WM, WN, BK, NBUF>;
    // PIPE_ names the main-loop schedule; the library has schedule 1 only (every fragment of a chunk is requested from LDS right after the chunk's
    // barrier, the DMAs of chunk + 2 are issued while those reads fly, then the chunk's MFMAs go back to back).  The schedules that were measured against
    // it in rounds 2 - 4 and lost -- fragment reads one k-step ahead, half the waves issuing the DMAs, wave halves half a chunk apart, weight fragments
    // straight into VGPRs -- are gone from the tree (round 6; their tables: profiles/r4m_gemm_harness_variants.txt); the template
    // parameter stays so that kernel names in profiles/ keep their meaning.
    static_assert(PIPE_ == 1, "schedule 1 is the only one in the library");
    static_assert(TAPS == 0 || (TAPS == 3 && NBUF == 3), "interleaved taps: the tap of a chunk is its ring slot (ring depth 3 == 3 taps)");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // Workgroup ids are dealt round-robin to the 8 XCDs.  The `mblocks` workgroups that share one activation tile get ids
    // congruent mod 8 (one L2).  With xcd_span > 0 each XCD also walks a CONTIGUOUS run of time tiles, so the rows a
    // dilated tap reaches back to (<= 2d rows = <= 32 tiles at d = 2048) were fetched by the same XCD a moment ago and are
    // still in its 4 MB L2 (interleaved order: only taps of d % 1024 == 0 stay on their XCD; the others re-fetch).
    const int id = wg_id;
    const int xcd = id & 7, q = id >> 3;
    const int mblk = q % a.mblocks;
    const int tile = a.xcd_span > 0 ? xcd * a.xcd_span + q / a.mblocks : (q / a.mblocks) * 8 + xcd;
    if (tile >= a.ntiles) return;
    if (a.kprof && tid == 0) atomicMin(a.kprof, (unsigned long long)wall_clock64());
#ifdef WN_PHASE_STAMPS      // harness diagnostics (tools/stream_harness.hip -DWN_PHASE_STAMPS): kclk is then a per-workgroup stamp table, kclk[id * 8 + i] = 100 MHz wall clock at phase
                            // boundary i (0 start, 1 set-up done, 2 first chunk landed, 3 main loop done, 4 last store issued, 5 stores drained): profiles/r9i_stream_harness_phases.txt
    unsigned long long* const ph_ = a.kclk ? a.kclk + (size_t)id * 8 : nullptr;
#define WN_STAMP(i) do { if (ph_ && tid == 0) ph_[i] = (unsigned long long)wall_clock64(); } while (0)
    WN_STAMP(0);
#else
#define WN_STAMP(i) do { } while (0)
    if (a.kclk && id == 0 && tid == 0) { a.kclk[0] = __builtin_amdgcn_s_memtime(); a.kclk[1] = (unsigned long long)wall_clock64(); }
#endif
    if (a.stagger > 0 && id < 512) {
        // All tiles cost the same, so co-resident (and neighbouring) workgroups would reach their MFMA-idle, store-heavy
        // epilogues at the same moment.  First-round workgroups therefore start with a placement-dependent delay; later
        // rounds inherit the offsets.  Placement is a heuristic (dispatch order is not architecturally defined): speed only.
        constexpr bool two_res = lds_gemm_min_waves(MT, NT, WM, WN, BK, NBUF) * 4 / (WM * WN) >= 2;
        const int delay = two_res ? (q >= 32 ? a.stagger : 0) : (id < 256 ? ((q & 7) * a.stagger) >> 3 : 0);
        if (delay > 0) {
            const uint64_t t_start = __builtin_amdgcn_s_memtime();
            while (__builtin_amdgcn_s_memtime() - t_start < (uint64_t)delay) __builtin_amdgcn_s_sleep(16);
        }
    }
    const int bl = tile / a.tiles_per_utt;
    const int b = bl + a.b0;
    const int t0 = (tile - bl * a.tiles_per_utt) * Cfg::TTILE;
    const int T = a.T;
    const int64_t rowbase = (int64_t)b * T;

    // The accumulators START at the bias of their output channel (gate: b_dil + b_cin + global-conditioning row of this utterance;
    // 1x1 convs: their bias), so the fused epilogues neither load nor add it: 4 x 16-B loads here, under the first DMAs, instead of
    // 16 bias values per 8-channel item in the epilogue.  acc[i][j][r] is output row (wm*MT + i)*32 + (r/4)*8 + (lane>>5)*4 + r%4.
    f32x16_t acc[MT][NT];
    if constexpr (EPI == EPI_GATE || EPI == EPI_STORE_BF16) {
        const float* bp = a.e.bias;
        if constexpr (EPI == EPI_GATE) bp += (int64_t)b * a.e.bias_bstride;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int ml = (wm * MT + i) * 32 + qd * 8 + (lane >> 5) * 4;       // row inside the workgroup's M tile
                float4 bv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (bp) {
                    if constexpr (EPI == EPI_GATE) {
                        // packed gate rows: 64-row groups of [32 tanh rows | 32 sigmoid rows] of the same 32 gate channels
                        const int gl = (ml >> 6) * 32 + (ml & 31);
                        bv = *reinterpret_cast<const float4*>(bp + ((ml & 32) ? a.e.GH : 0) + mblk * (Cfg::MTILE / 2) + gl);
                    } else {
                        bv = *reinterpret_cast<const float4*>(bp + mblk * Cfg::MTILE + ml);
                    }
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) { acc[i][j][qd * 4] = bv.x; acc[i][j][qd * 4 + 1] = bv.y; acc[i][j][qd * 4 + 2] = bv.z; acc[i][j][qd * 4 + 3] = bv.w; }
            }
    } else {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    }

    const int mtile_wg = mblk * (WM * MT);           // first 32-row m-tile of this workgroup
    const int mtile0 = mtile_wg + wm * MT;            // first m-tile of this wave

    const int chunks_per_rep = [&] { int n = 0; for (int s = 0; s < a.nseg; ++s) n += (a.seg[s].nk + BK - 1) / BK; return n; }();
    const int nchunks = chunks_per_rep * a.nrep;

    // ---- staging iterator (chunk being DMA'd) and compute iterator (chunk being multiplied).  Everything the loop needs
    // from the segment descriptors is kept in registers and refreshed only when the iterator enters a new segment:
    // a dynamically indexed a.seg[i] would be an s_load + s_waitcnt lgkmcnt(0) (which also drains the ds_reads) per chunk.
    int s_rep = 0, s_sg = 0, s_left = a.seg[0].nk, s_kstep = 0;
    const bf16_t* s_ptr = a.seg[0].base + a.seg[0].col0;                // + rep offset + channels already staged (wave-uniform)
    // TAPS: chunk c < t_total is tap c % 3 of the k-block c / 3 of seg[0..2] (one tensor, three row shifts).  Reuse distance of a
    // row shared by two taps = one chunk of the co-resident tiles (a few 100 KiB) instead of a whole segment pass (> L2).
    int t_left = TAPS ? TAPS * (a.seg[0].nk / BK) : 0;
    // per-lane constants of the B image: piece g = wave + p*NW holds rows g*(1024/RB)...; LDS byte lane*16 -> (row, slot)
    constexpr int NWD = Cfg::NW;                                      // every wave issues its share of the chunk's DMAs
    constexpr int APW = Cfg::A_INSTR / NWD, BPW = Cfg::B_INSTR / NWD;
    int b_c8[BPW], b_t[BPW], b_off[BPW];
    auto enter_segment = [&]() {          // per-lane element offsets inside the current segment (-1: reads the zero page)
        const int ld = a.seg[s_sg].ld, shift = a.seg[s_sg].shift;       // one s_load per SEGMENT, not per chunk
#pragma unroll
        for (int p = 0; p < BPW; ++p) {
            const int ts = b_t[p] + shift;
            b_off[p] = (b_t[p] < T && ts >= 0 && ts < T) ? (int)((rowbase + ts) * ld + b_c8[p]) : -1;
        }
    };
#pragma unroll
    for (int p = 0; p < BPW; ++p) {
        const int row = (wave + p * NWD) * (1024 / Cfg::RB) + (lane * 16) / Cfg::RB;
        b_c8[p] = ((lane % Cfg::SPR) ^ ((row / Cfg::RPL) % Cfg::SPR)) * 8;
        b_t[p] = t0 + row;
    }
    int b_offt[TAPS ? TAPS : 1][BPW];          // TAPS: per-tap element offsets of this lane's rows (-1: zero page)
    if constexpr (TAPS > 0) {
#pragma unroll
        for (int k = 0; k < TAPS; ++k) {
            const int ld = a.seg[k].ld, shift = a.seg[k].shift;
#pragma unroll
            for (int p = 0; p < BPW; ++p) {
                const int ts = b_t[p] + shift;
                b_offt[k][p] = (b_t[p] < T && ts >= 0 && ts < T) ? (int)((rowbase + ts) * ld + b_c8[p]) : -1;
            }
        }
        s_sg = TAPS;                               // the sequential iterator takes over at seg[TAPS] (if any) once the taps are staged
        if (TAPS < a.nseg) { s_left = a.seg[TAPS].nk; s_ptr = a.seg[TAPS].base + a.seg[TAPS].col0; enter_segment(); }
        else {
            s_rep = a.nrep; s_left = BK;
#pragma unroll
            for (int p = 0; p < BPW; ++p) b_off[p] = -1;
        }
    } else enter_segment();

    auto stage = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
        char* const abuf = lds + Cfg::a_base(BUF);
        char* const bbuf = lds + Cfg::b_base(BUF);
        if constexpr (TAPS > 0) {
            if (t_left > 0) {
                // chunk c lives in ring slot c % NBUF and NBUF == TAPS: this buffer always holds tap BUF
#pragma unroll
                for (int p = 0; p < APW; ++p) {
                    const int f = wave + p * NWD;
                    const bf16_t* base = a.Apk + ((int64_t)(mtile_wg + f / Cfg::KS) * a.ksteps_total + s_kstep + f % Cfg::KS) * 512;
                    lds_dma16(base + lane * 8, __builtin_amdgcn_readfirstlane(lds_addr_of(abuf + f * 1024)));
                }
                const bf16_t* const tp = a.seg[0].base + a.seg[0].col0 + (s_kstep / (TAPS * Cfg::KS)) * BK;     // k-block of this chunk
#pragma unroll
                for (int p = 0; p < BPW; ++p) {
                    const bf16_t* src = b_offt[BUF][p] >= 0 ? tp + b_offt[BUF][p] : a.zero;
                    lds_dma16(src, __builtin_amdgcn_readfirstlane(lds_addr_of(bbuf + (wave + p * NWD) * 1024)));
                }
                s_kstep += Cfg::KS; --t_left;
                return;
            }
        }
        if constexpr (TAPS == 0) {
            // full chunk inside the current segment (all but one chunk of every launch of the engine): no partial-k masks, the weight
            // panel through the SGPR-base DMA form
            if (s_left >= BK && s_rep < a.nrep) {
#pragma unroll
                for (int p = 0; p < APW; ++p) {
                    const int f = wave + p * NWD;
                    const uint64_t v = (uint64_t)(a.Apk + ((int64_t)(mtile_wg + f / Cfg::KS) * a.ksteps_total + s_kstep + f % Cfg::KS) * 512);
                    const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
                    lds_dma16_s(sb, lane * 16, __builtin_amdgcn_readfirstlane(lds_addr_of(abuf + f * 1024)));
                }
#pragma unroll
                for (int p = 0; p < BPW; ++p) {
                    const bf16_t* src = b_off[p] >= 0 ? s_ptr + b_off[p] : a.zero;
                    lds_dma16(src, __builtin_amdgcn_readfirstlane(lds_addr_of(bbuf + (wave + p * NWD) * 1024)));
                }
                s_kstep += Cfg::KS; s_left -= BK; s_ptr += BK;
                if (s_left <= 0) {
                    ++s_sg; if (s_sg == a.nseg) { s_sg = 0; ++s_rep; }
                    s_left = a.seg[s_sg].nk; s_ptr = a.seg[s_sg].base + a.seg[s_sg].col0 + (int64_t)s_rep * a.rep_stride;
                    enter_segment();
                }
                return;
            }
        }
        const int kc = (s_rep < a.nrep) ? min(BK, s_left) : 0;      // past the last chunk: an all-zero chunk
        // A: fragment f = mt*KS + ks  <-  Apk[(mtile_wg + mt)][s_kstep + ks]; wave-uniform base + lane*16
#pragma unroll
        for (int p = 0; p < APW; ++p) {
            const int f = wave + p * NWD;
            const bf16_t* base = a.Apk + ((int64_t)(mtile_wg + f / Cfg::KS) * a.ksteps_total + s_kstep + f % Cfg::KS) * 512;
            const bf16_t* src = ((f % Cfg::KS) * 16 < kc) ? base + lane * 8 : a.zero;
            lds_dma16(src, __builtin_amdgcn_readfirstlane(lds_addr_of(abuf + f * 1024)));
        }
#pragma unroll
        for (int p = 0; p < BPW; ++p) {
            const bf16_t* src = (b_off[p] >= 0 && b_c8[p] < kc) ? s_ptr + b_off[p] : a.zero;
            lds_dma16(src, __builtin_amdgcn_readfirstlane(lds_addr_of(bbuf + (wave + p * NWD) * 1024)));
        }
        s_kstep += kc >> 4; s_left -= BK; s_ptr += BK;
        if (s_left <= 0) {
            ++s_sg; if (s_sg == a.nseg) { s_sg = 0; ++s_rep; }
            s_left = a.seg[s_sg].nk; s_ptr = a.seg[s_sg].base + a.seg[s_sg].col0 + (int64_t)s_rep * a.rep_stride;
            enter_segment();
        }
    };

    // fragment read offsets (loop invariant): A lane-linear, B swizzled
    int b_rd[NT][Cfg::KS];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int ks = 0; ks < Cfg::KS; ++ks) {
            const int row = (wn * NT + j) * 32 + (lane & 31);
            b_rd[j][ks] = row * Cfg::RB + (((ks * 2 + (lane >> 5)) ^ ((row / Cfg::RPL) % Cfg::SPR)) * 16);
        }
    const int a_rd = (wm * MT * Cfg::KS * 64 + lane) * 16;

    // one ring step: chunk `ch` lives in buffer BUF; chunk ch+NBUF-1 is DMA'd into the buffer freed by chunk ch-1.  Every fragment of
    // the chunk is requested from LDS straight after the barrier, the DMA issue + iterator bookkeeping of the next chunk runs while
    // those reads are in flight, then the chunk's MFMAs go back to back.  In a partial chunk (segment width % BK != 0) the tail
    // k-steps of BOTH operands were DMA'd from the zero page, so they add exact zeros -- no branch around the accumulators.
    auto ring_step = [&](auto bufc, int ch) {
        constexpr int BUF = decltype(bufc)::value;
        // all DMAs except those of the (NBUF-2) youngest chunks have landed
        const int younger = min(NBUF - 2, nchunks - 1 - ch);
        if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::LPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ch == 0) WN_STAMP(2);
        const char* const buf = lds + Cfg::a_base(BUF); const char* const bufb = lds + Cfg::b_base(BUF);
        bf16x8_t af[Cfg::KS][MT], bfr[Cfg::KS][NT];
#pragma unroll
        for (int ks = 0; ks < Cfg::KS; ++ks) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[ks][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(buf + a_rd + (i * Cfg::KS + ks) * 1024));
#pragma unroll
            for (int j = 0; j < NT; ++j)
                bfr[ks][j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(bufb + b_rd[j][ks]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ch + NBUF - 1 < nchunks) stage(std::integral_constant<int, (BUF + NBUF - 1) % NBUF>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < Cfg::KS; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
    };
    static_assert(NBUF == 2 || NBUF == 3, "ring depth");

    WN_STAMP(1);
    // prologue: fill NBUF-1 buffers
    stage(std::integral_constant<int, 0>{});
    if constexpr (NBUF == 3) { if (nchunks > 1) stage(std::integral_constant<int, 1>{}); }
    int ch0 = 0;
    if constexpr (TAPS == 3 && NBUF == 3) {
        // ---- regular part of the K-interleaved taps (all k-blocks but the last): the generic staging iterator above costs ~80 mostly
        // scalar, branchy instructions per chunk IN FRONT of the wave's MFMAs (in-order issue).  Here the three ring steps of one
        // k-block are unrolled with everything they need in registers: the weight panel source is an SGPR base advanced by one
        // scalar add per chunk (SGPR-base DMA form), the activation source one 64-bit add per lane and tap; no segment
        // bookkeeping, no division, one loop branch per three chunks.  Same chunk order, same sums.
        const int nkb = a.seg[0].nk / BK;
        if (nkb >= 2 && nchunks > 3 * (nkb - 1) + 1) {
            uint64_t sa[APW];
#pragma unroll
            for (int p = 0; p < APW; ++p) {
                const int f = wave + p * NWD;
                const uint64_t v = (uint64_t)(a.Apk + ((int64_t)(mtile_wg + f / Cfg::KS) * a.ksteps_total + s_kstep + f % Cfg::KS) * 512);
                sa[p] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
            }
            const uint32_t a_voff = lane * 16;
            const bf16_t* tpb = a.seg[0].base + a.seg[0].col0;          // k-block of the chunk being staged (chunk 2 = tap 2 of k-block 0 is next)
            auto fast_stage = [&](auto bufc) {
                constexpr int BUF = decltype(bufc)::value;              // ring slot == tap
                char* const abuf = lds + Cfg::a_base(BUF);
                char* const bbuf = lds + Cfg::b_base(BUF);
                if constexpr (BUF == 0) tpb += BK;
#pragma unroll
                for (int p = 0; p < APW; ++p) {
                    lds_dma16_s(sa[p], a_voff, __builtin_amdgcn_readfirstlane(lds_addr_of(abuf + (wave + p * NWD) * 1024)));
                    sa[p] += Cfg::KS * 1024;
                }
#pragma unroll
                for (int p = 0; p < BPW; ++p) {
                    const bf16_t* src = b_offt[BUF][p] >= 0 ? tpb + b_offt[BUF][p] : a.zero;
                    lds_dma16(src, __builtin_amdgcn_readfirstlane(lds_addr_of(bbuf + (wave + p * NWD) * 1024)));
                }
            };
            auto fast_step = [&](auto bufc) {
                constexpr int BUF = decltype(bufc)::value;
                // chunk ch landed (only chunk ch+1 may still be in flight); lgkmcnt(0): this wave's fragment reads of chunk ch-1 have
                // RETURNED before the barrier releases its ring slot to the next DMA -- the compiler is free to sink the (register-only)
                // MFMAs of chunk ch-1 and the waits in front of them below the barrier, and does
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(Cfg::LPC) : "memory");
                __builtin_amdgcn_s_barrier();
                const char* const buf = lds + Cfg::a_base(BUF); const char* const bufb = lds + Cfg::b_base(BUF);
                bf16x8_t af[Cfg::KS][MT], bfr[Cfg::KS][NT];
#pragma unroll
                for (int ks = 0; ks < Cfg::KS; ++ks) {
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        af[ks][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(buf + a_rd + (i * Cfg::KS + ks) * 1024));
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        bfr[ks][j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(bufb + b_rd[j][ks]));
                }
                __builtin_amdgcn_sched_barrier(0);
                fast_stage(std::integral_constant<int, (BUF + NBUF - 1) % NBUF>{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < Cfg::KS; ++ks)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
            };
            for (int kb = 0; kb < nkb - 1; ++kb) {
                fast_step(std::integral_constant<int, 0>{});
                fast_step(std::integral_constant<int, 1>{});
                fast_step(std::integral_constant<int, 2>{});
            }
            ch0 = 3 * (nkb - 1);
            s_kstep += Cfg::KS * ch0; t_left -= ch0;
        }
    }
    for (int ch = ch0; ch < nchunks; ch += NBUF) {
        ring_step(std::integral_constant<int, 0>{}, ch);
        if (ch + 1 < nchunks) ring_step(std::integral_constant<int, 1>{}, ch + 1);
        if constexpr (NBUF == 3) { if (ch + 2 < nchunks) ring_step(std::integral_constant<int, 2>{}, ch + 2); }
    }

    WN_STAMP(3);
    if constexpr (EPI == EPI_STORE_F32_BOT) {
        // [B][M][T] fp32 output: lanes are consecutive time steps, already coalesced
        wn_tile_epilogue<MT, NT, EPI>(a, acc, mtile0, t0 + wn * NT * 32, b, T, rowbase, lane);
    } else {
        // ---- epilogue v2: accumulators -> LDS (fp32, [time][channel]) -> one (row, 8 channels) item per thread, so that
        // every global access of the fused epilogue is a 16-B piece of a fully covered row segment.
        constexpr int PITCH = Cfg::EPI_PITCH;                // bytes; +16 keeps the ds_write_b128 of 8-lane groups conflict free
        constexpr int PROWS = Cfg::EPI_ROWS;                 // rows per pass (one 32-row tile of every wave column)
        const EpiArgs& e = a.e;
        const int h = lane >> 5;
        // The barriers of the epilogue order LDS traffic only (no LDS-DMA is in flight any more: the last ring step drained vmcnt), so
        // they must not drain the vector-memory counter: the global loads of a pass are issued BEFORE its two barriers and the stores
        // of the previous pass stay in flight across them.
        auto epi_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        // Item geometry.  A pass holds PROWS rows x MTILE channels; thread -> (row, 8 channels) items it = tid + k * NTH.  NTH is a multiple
        // of the items per row, so a thread keeps ITS channel group and walks rows rl0 + k * RSTEP: every address is
        // (wave-uniform base) + (32-bit lane offset), no 64-bit multiplies per item.
        constexpr int NTH = Cfg::NW * 64;
        auto unpack8 = [](const uint4 x, float* f) {
            f[0] = bf2f((bf16_t)(x.x & 0xffff)); f[1] = bf2f((bf16_t)(x.x >> 16)); f[2] = bf2f((bf16_t)(x.y & 0xffff)); f[3] = bf2f((bf16_t)(x.y >> 16));
            f[4] = bf2f((bf16_t)(x.z & 0xffff)); f[5] = bf2f((bf16_t)(x.z >> 16)); f[6] = bf2f((bf16_t)(x.w & 0xffff)); f[7] = bf2f((bf16_t)(x.w >> 16));
        };
        auto pack8 = [](const float* f) { return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7])); };
        const int64_t tile_row0 = rowbase + t0;            // first row of this workgroup's time tile (wave-uniform)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if constexpr (EPI == EPI_GATE) {
                epi_barrier();
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int rl = wn * 32 + (lane & 31), ml = (wm * MT + i) * 32 + qd * 8 + h * 4;
                        *reinterpret_cast<float4*>(lds + rl * PITCH + ml * 4) =
                            make_float4(acc[i][j][qd * 4], acc[i][j][qd * 4 + 1], acc[i][j][qd * 4 + 2], acc[i][j][qd * 4 + 3]);
                    }
                epi_barrier();
                constexpr int GT = Cfg::MTILE / 2, C8 = GT / 8, ITEMS = PROWS * C8;
                static_assert(ITEMS % NTH == 0 && NTH % C8 == 0, "gate epilogue items");
                constexpr int NIT = ITEMS / NTH, RSTEP = NTH / C8;
                const int c8 = tid % C8, rl0 = tid / C8;
                const int gl = c8 * 8, ml = (gl >> 5) * 64 + (gl & 31);
                bf16_t* const TSb = (bf16_t*)e.out0 + tile_row0 * e.ld_out0 + mblk * GT + gl;      // (the bias is already in the accumulators)
                bf16_t* const Ub = (bf16_t*)e.out1 + tile_row0 * e.ld_out1 + mblk * GT + gl;
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int rl = rl0 + k * RSTEP;
                    const int tr = ((rl >> 5) * NT + j) * 32 + (rl & 31);             // row inside the time tile
                    const float4 a0 = *reinterpret_cast<const float4*>(lds + rl * PITCH + ml * 4), a1 = *reinterpret_cast<const float4*>(lds + rl * PITCH + ml * 4 + 16);
                    const float4 b0 = *reinterpret_cast<const float4*>(lds + rl * PITCH + (ml + 32) * 4), b1 = *reinterpret_cast<const float4*>(lds + rl * PITCH + (ml + 32) * 4 + 16);
                    const float za[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, zb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                    uint32_t ps[4], pu[4];
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float t0_ = fast_tanh(za[2 * p]), t1_ = fast_tanh(za[2 * p + 1]);
                        const float s0_ = fast_sigmoid(zb[2 * p]), s1_ = fast_sigmoid(zb[2 * p + 1]);
                        ps[p] = pack_bf2(s0_, s1_); pu[p] = pack_bf2(t0_ * s0_, t1_ * s1_);
                    }
                    if (t0 + tr < T) {
                        // saved for backward: sigmoid + u (tanh is recovered as u / sigmoid, gate_tanh_from)
                        *reinterpret_cast<uint4*>(TSb + (uint32_t)(tr * e.ld_out0)) = make_uint4(ps[0], ps[1], ps[2], ps[3]);
                        *reinterpret_cast<uint4*>(Ub + (uint32_t)(tr * e.ld_out1)) = make_uint4(pu[0], pu[1], pu[2], pu[3]);
                    }
                }
            } else {
                constexpr int C8 = Cfg::MTILE / 8, ITEMS = PROWS * C8;
                static_assert(ITEMS % NTH == 0 && NTH % C8 == 0, "epilogue items");
                constexpr int NIT = ITEMS / NTH, RSTEP = NTH / C8;
                const int c8 = tid % C8, rl0 = tid / C8;
                const int mo = mblk * Cfg::MTILE + c8 * 8;
                // ---- every global load of the pass first (round 2 walked the items in a loop of load -> wait -> load -> wait -> compute ->
                // store: 8 dependent HBM round trips per workgroup, which is what bounded the HBM-bound launches at ~4 TB/s with 16 waves
                // per CU each holding 32 B in flight).  Rows past the end of the utterance read the last valid row (never stored).  (Round 6: issuing them a PASS
                // AHEAD -- pass j + 1's while pass j is computed, the one-pass tiles' at kernel start -- is bit-identical and changes nothing: the 4.6 us of a pass
                // are ~480 VALU instructions per thread behind two barriers, not the round trip; step 9.81 - 9.84 vs 9.79 ms, profiles/r9j_*; not kept.)
                uint4 l0[NIT], l1[NIT];
                int trs[NIT];
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int rl = rl0 + k * RSTEP;
                    trs[k] = ((rl >> 5) * NT + j) * 32 + (rl & 31);
                    const int trc = min(trs[k], T - 1 - t0);
                    l0[k] = make_uint4(0, 0, 0, 0); l1[k] = make_uint4(0, 0, 0, 0);
                    if constexpr (EPI == EPI_DGATE) {
                        l0[k] = *reinterpret_cast<const uint4*>((const bf16_t*)e.in1 + tile_row0 * e.ld_in0 + mo + (uint32_t)(trc * e.ld_in0));      // u = tanh * sigmoid
                        l1[k] = *reinterpret_cast<const uint4*>((const bf16_t*)e.in0 + tile_row0 * e.ld_in0 + mo + (uint32_t)(trc * e.ld_in0));      // sigmoid
                    } else if constexpr (EPI == EPI_MASK_STORE) {
                        l0[k] = *reinterpret_cast<const uint4*>((const bf16_t*)e.in0 + tile_row0 * e.ld_in0 + mo + (uint32_t)(trc * e.ld_in0));
                    } else {      // EPI_STORE_BF16 / EPI_DX: the residual operand, when there is one
                        if (e.in0) l0[k] = *reinterpret_cast<const uint4*>((const bf16_t*)e.in0 + tile_row0 * e.ld_in0 + mo + (uint32_t)(trc * e.ld_in0));
                    }
                }
                epi_barrier();
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int rl = wn * 32 + (lane & 31), ml = (wm * MT + i) * 32 + qd * 8 + h * 4;
                        *reinterpret_cast<float4*>(lds + rl * PITCH + ml * 4) =
                            make_float4(acc[i][j][qd * 4], acc[i][j][qd * 4 + 1], acc[i][j][qd * 4 + 2], acc[i][j][qd * 4 + 3]);
                    }
                epi_barrier();
                bf16_t* const o0 = (bf16_t*)e.out0 + tile_row0 * e.ld_out0 + mo;
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int rl = rl0 + k * RSTEP, tr = trs[k];
                    const bool valid = t0 + tr < T;
                    const float4 a0 = *reinterpret_cast<const float4*>(lds + rl * PITCH + c8 * 32), a1 = *reinterpret_cast<const float4*>(lds + rl * PITCH + c8 * 32 + 16);
                    float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    if constexpr (EPI == EPI_STORE_BF16) {        // (bias: already in the accumulators)
                        if (e.in0) {
                            float x[8]; unpack8(l0[k], x);
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] += x[r];
                        }
#pragma unroll
                        for (int r = 0; r < 8; ++r) { v[r] *= e.scale; if (e.relu) v[r] = fmaxf(v[r], 0.0f); }
                        const uint4 pk = pack8(v);
                        if (valid) *reinterpret_cast<uint4*>(o0 + (uint32_t)(tr * e.ld_out0)) = pk;
                        if (e.out1) {     // dropout of the next layer's conv input (tf.layers.dropout, modules.py:484), from the ROUNDED value
                            float x[8], dd[8]; unpack8(pk, x);
                            const uint32_t e0 = (uint32_t)((tile_row0 + tr) * a.drop_ld + mo);      // % 8 == 0 (mo % 8 == 0, drop_ld = R % 8 == 0): two whole quads
                            uint32_t wq[4];
                            wn_drop_quad(a.key_lo, a.key_hi, e0 >> 2, wq[0], wq[1]); wn_drop_quad(a.key_lo, a.key_hi, (e0 >> 2) + 1, wq[2], wq[3]);
#pragma unroll
                            for (int p = 0; p < 4; ++p) {
                                const uint32_t w = wq[p];
                                dd[2 * p] = ((w & 0xffffu) >= a.thresh16) ? x[2 * p] * a.keep_scale : 0.0f;
                                dd[2 * p + 1] = ((w >> 16) >= a.thresh16) ? x[2 * p + 1] * a.keep_scale : 0.0f;
                            }
                            if (valid) *reinterpret_cast<uint4*>((bf16_t*)e.out1 + tile_row0 * e.ld_out1 + mo + (uint32_t)(tr * e.ld_out1)) = pack8(dd);
                        }
                    } else if constexpr (EPI == EPI_DGATE) {
                        float uu[8], sg[8], da[8], db[8];
                        unpack8(l0[k], uu); unpack8(l1[k], sg);
#pragma unroll
                        for (int r = 0; r < 8; ++r) gate_backward(v[r], uu[r], sg[r], da[r], db[r]);
                        if (valid) {
                            *reinterpret_cast<uint4*>(o0 + (uint32_t)(tr * e.ld_out0)) = pack8(da);
                            *reinterpret_cast<uint4*>(o0 + e.GH + (uint32_t)(tr * e.ld_out0)) = pack8(db);
                        }
                    } else if constexpr (EPI == EPI_MASK_STORE) {
                        float ref[8]; unpack8(l0[k], ref);
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = (ref[r] > 0.0f) ? v[r] * e.scale : 0.0f;
                        if (valid) *reinterpret_cast<uint4*>(o0 + (uint32_t)(tr * e.ld_out0)) = pack8(v);
                    } else if constexpr (EPI == EPI_DX) {
                        if (a.thresh16 != 0) {
                            const uint32_t e0 = (uint32_t)((tile_row0 + tr) * a.drop_ld + mo);
                            uint32_t wq[4];
                            wn_drop_quad(a.key_lo, a.key_hi, e0 >> 2, wq[0], wq[1]); wn_drop_quad(a.key_lo, a.key_hi, (e0 >> 2) + 1, wq[2], wq[3]);
#pragma unroll
                            for (int p = 0; p < 4; ++p) {
                                const uint32_t w = wq[p];
                                v[2 * p] = ((w & 0xffffu) >= a.thresh16) ? v[2 * p] * a.keep_scale : 0.0f;
                                v[2 * p + 1] = ((w >> 16) >= a.thresh16) ? v[2 * p + 1] * a.keep_scale : 0.0f;
                            }
                        }
                        if (e.in0) {
                            float x[8]; unpack8(l0[k], x);
#pragma unroll
                            for (int r = 0; r < 8; ++r) v[r] += x[r];
                        }
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] *= e.scale;
                        if (valid) *reinterpret_cast<uint4*>(o0 + (uint32_t)(tr * e.ld_out0)) = pack8(v);
                    }
                }
            }
        }
    }
    if (a.kprof && tid == 0) atomicMax(a.kprof + 1, (unsigned long long)wall_clock64());
#ifdef WN_PHASE_STAMPS
    if (ph_) { WN_STAMP(4); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); WN_STAMP(5); }
#else
    if (a.kclk && id == 0 && tid == 0) { a.kclk[0] = __builtin_amdgcn_s_memtime() - a.kclk[0]; a.kclk[1] = (unsigned long long)wall_clock64() - a.kclk[1]; }
#endif
#undef WN_STAMP
}

template <int MT, int NT, int WM, int WN, int BK, int NBUF, int EPI, int PIPE = 1, int TAPS = 0>
__global__ __launch_bounds__(WM * WN * 64, (lds_gemm_min_waves(MT, NT, WM, WN, BK, NBUF)))
void wn_gemm_lds_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(1024))) char lds[LdsGemmCfg<MT, NT, WM, WN, BK, NBUF>::LDS_BYTES];
    wn_gemm_lds_body<MT, NT, WM, WN, BK, NBUF, EPI, PIPE, TAPS>(a, lds, blockIdx.x);
}

// grids of at least this many workgroups (more than two full rounds of 2 x 256 slots) de-phase their second-resident workgroups
#define WN_STAGGER_MIN_GRID 1024

// Host-side launcher: picks the main loop and workgroup shape from M.
template <int EPI>
static inline int wn_launch_gemm(wn_ctx* ctx, GemmArgs& a, int M, hipStream_t st) {
    if (a.taps != 0 && a.kil != 32) WN_FAIL(ctx, WN_E_STATE, "wn_launch_gemm: K-interleaved pack with %d-channel blocks (this header stages 32)", a.kil);
    if (EPI == EPI_GATE && M % 64 != 0) WN_FAIL(ctx, WN_E_SHAPE, "gate GEMM needs gate_channels %% 64 == 0 (got M=%d)", M);
    if (ctx->trace_state == 1 && ctx->trace_n < WN_TRACE_MAX) {      // WN_DEVTRACE: this launch's own stamp slot (takes the slot of wn_profile for this step)
        a.kprof = ctx->trace_dev + 2 * ctx->trace_n;
        ctx->trace_tag[ctx->trace_n].epi = EPI; ctx->trace_tag[ctx->trace_n].st = (void*)st; ctx->trace_tag[ctx->trace_n].rows = a.B * a.T; ++ctx->trace_n;
    }
    if constexpr (EPI == EPI_STORE_BF16) {
        // Launches with fewer 256 x 128 tiles than workgroup slots (2 x 256) -- the out conv / head convs of a half batch: 344 --
        // take 256 x 64 tiles instead (K-chunks of 64, 2-deep ring, still two workgroups per CU): twice the workgroups, half the
        // work each.  Half-batch out conv 31.6 -> 27.7 us in the harness (profiles/r2h_gemm_harness_b4.txt); at full-batch size the
        // 128-row tile is the faster one (46.6 vs 48.2 us), hence the rule.
        bool k64 = a.nrep == 1 && a.taps == 0;
        for (int sgi = 0; sgi < a.nseg; ++sgi) k64 = k64 && a.seg[sgi].nk % 64 == 0;
        if (k64 && M % 256 == 0 && a.e.M_valid == M && a.zero && (int64_t)cdiv(a.T, 128) * a.B * (M / 256) < 512) {
            a.mblocks = M / 256;
            a.tiles_per_utt = cdiv(a.T, 64);
            a.ntiles = a.tiles_per_utt * a.B;
            a.xcd_span = cdiv(a.ntiles, 8);
            const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
            a.stagger = 0;
            hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 1, 4, 2, 64, 2, EPI, 1>), dim3(grid), dim3(512), 0, st, a);
            WN_LAUNCH_CHECK(ctx);
            return WN_OK;
        }
    }
    if constexpr (EPI != EPI_STORE_F32_BOT) {
        if (M % 256 == 0 && a.e.M_valid == M && a.zero) {
            // v2: 256 channels x 128 time rows per 8-wave workgroup, K-chunks of 32, 3-deep LDS-DMA ring, 2 workgroups per CU
            a.mblocks = M / 256;
            a.tiles_per_utt = cdiv(a.T, 128);
            a.ntiles = a.tiles_per_utt * a.B;
            a.xcd_span = cdiv(a.ntiles, 8);
            const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
            a.stagger = grid >= WN_STAGGER_MIN_GRID ? 8000 : 0;      // more than two full rounds: desynchronise the co-resident workgroups
            if constexpr (EPI == EPI_GATE || EPI == EPI_DX) {
                if (a.taps == 3) {       // K-interleaved taps (packs built with kil = 32)
                    hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI, 1, 3>), dim3(grid), dim3(512), 0, st, a);
                    WN_LAUNCH_CHECK(ctx);
                    return WN_OK;
                }
            }
            hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI, 1>), dim3(grid), dim3(512), 0, st, a);
            WN_LAUNCH_CHECK(ctx);
            return WN_OK;
        }
    }
    if constexpr (EPI != EPI_STORE_F32_BOT) {
        if (M % 128 == 0 && a.e.M_valid == M && a.zero) {
            // 128-channel models (hparams.py defaults: R = S = 128): the same LDS-DMA main loop with a 128 x 128 tile (8 waves, 32 x 64 each)
            a.mblocks = M / 128;
            a.tiles_per_utt = cdiv(a.T, 128);
            a.ntiles = a.tiles_per_utt * a.B;
            a.xcd_span = cdiv(a.ntiles, 8);
            const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
            a.stagger = grid >= WN_STAGGER_MIN_GRID ? 8000 : 0;
            if constexpr (EPI == EPI_GATE || EPI == EPI_DX) {
                if (a.taps == 3) {
                    hipLaunchKernelGGL((wn_gemm_lds_kernel<1, 2, 4, 2, 32, 3, EPI, 1, 3>), dim3(grid), dim3(512), 0, st, a);
                    WN_LAUNCH_CHECK(ctx);
                    return WN_OK;
                }
            }
            hipLaunchKernelGGL((wn_gemm_lds_kernel<1, 2, 4, 2, 32, 3, EPI, 1>), dim3(grid), dim3(512), 0, st, a);
            WN_LAUNCH_CHECK(ctx);
            return WN_OK;
        }
    }
    if constexpr (EPI == EPI_STORE_F32_BOT) {
        if (M == 96 && a.zero) {           // d c_up with <= 96 conditioning channels: 96 channels x 192 time rows, 6 waves (32 x 96 each)
            a.mblocks = 1;
            a.tiles_per_utt = cdiv(a.T, 192);
            a.ntiles = a.tiles_per_utt * a.B;
            a.xcd_span = cdiv(a.ntiles, 8);
            const int grid = cdiv(a.ntiles, 8) * 8;
            // K-chunks of 64 with a 2-deep ring (half the ring steps) for narrow gate widths: -1 .. -2.4 % on the step of hparams.py's
            // defaults (G = 256), neutral at G = 512 (profiles/r4u_ab_dc_bk64.txt)
            bool k64 = a.seg[0].nk <= 256;
            for (int sgi = 0; sgi < a.nseg; ++sgi) k64 = k64 && a.seg[sgi].nk % 64 == 0;
            if (k64) hipLaunchKernelGGL((wn_gemm_lds_kernel<1, 3, 3, 2, 64, 2, EPI, 1>), dim3(grid), dim3(384), 0, st, a);
            else
            hipLaunchKernelGGL((wn_gemm_lds_kernel<1, 3, 3, 2, 32, 3, EPI, 1>), dim3(grid), dim3(384), 0, st, a);
            WN_LAUNCH_CHECK(ctx);
            return WN_OK;
        }
        if (M % 128 == 0 && a.zero) {      // d c_up: M = cin padded to 128, K = L*G: 128 channels x 256 time rows per workgroup
            a.mblocks = M / 128;
            a.tiles_per_utt = cdiv(a.T, 256);
            a.ntiles = a.tiles_per_utt * a.B;
            a.xcd_span = cdiv(a.ntiles, 8);
            const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
            hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 2, 4, 32, 3, EPI, 1>), dim3(grid), dim3(512), 0, st, a);
            WN_LAUNCH_CHECK(ctx);
            return WN_OK;
        }
    }
    if (a.taps != 0) WN_FAIL(ctx, WN_E_STATE, "K-interleaved pack (taps = %d) reached a kernel with sequential K order (M = %d)", a.taps, M);
    // v1: all shapes stage 128 time rows per workgroup (36 KiB LDS, 2 workgroups per CU)
    const int nrows = 128;
    const int mrows = (M % 128 == 0) ? 128 : (M % 64 == 0) ? 64 : 32;
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
