// REJECTED ALTERNATIVE, kept with its harness (tools/stream_harness.hip; includes the PRODUCT headers, nothing here is a fork): a "whole-B" variant of
// the tile engine for the HBM-bound 1x1 convolutions of the residual stack (out conv + residual, final_convolution_1, the head's backward mask GEMM).
//
// Same contraction, operands, fragment layouts and fused epilogue arithmetic as wn_gemm_lds_kernel (csrc/wn_tile.h) -- outputs are bit-identical --; what
// differs is WHEN a workgroup asks HBM for its bytes.  The ring kernel requests an activation tile chunk by chunk, one or two chunks ahead (8 - 16 KB in
// flight per workgroup), then the epilogue's residual rows in a second round trip.  Here a workgroup issues EVERYTHING it will ever read at its first
// instructions -- the epilogue's residual operand (into registers), the whole [TTILE rows x K] activation tile (LDS-DMA, 32 KB) and the first two weight
// slots: 64 KB per workgroup, 128 KB per CU in flight --, waits ONCE, and only the weight panel (L2-resident) keeps streaming through a 3-slot ring.
// Hypothesis (round 6): these launches are latency-bound chains and more bytes in flight lifts them from ~4.5 TB/s towards the 6.3 TB/s a copy reaches.
// MEASURED (profiles/r9b_stream_harness.txt, r9b_ab_wb.txt): out conv at C2 widths, full batch 41.9 - 44.1 us vs the ring kernel's 38.5 - 42.3 (4.1 - 4.3 vs
// 4.3 - 4.7 TB/s); half batch 22.8 - 24.0 vs 24.0 - 26.1; hparams.py widths 20.2 - 20.5 vs 18.2 - 18.4 (full), 12.5 - 12.8 vs 13.5 - 14.1 (half); live C2 step
// 9.805 / 9.832 vs 9.814 / 9.830 ms.  Bytes in flight at workgroup START are not what bounds these kernels: with everything requested up front the same
// ~4.2 - 4.7 TB/s comes out -- while a plain streaming kernel with this traffic mix reaches 6.7 TB/s on the same (Infinity-Cache-resident) tensors
// (profiles/r9f_hbm_streams_probe.txt).  What is between: a workgroup is a load phase, a matrix phase and a store phase, and two per CU do not keep requests
// in flight CONTINUOUSLY; a persistent kernel with the tile triple-buffered would.  Not adopted.
#pragma once
#include "wn_tile.h"
#include <utility>

template <int WM, int WN, int NKC>
struct WbCfg {
    static constexpr int NW = WM * WN, NTH = NW * 64;
    static constexpr int MT = 2;                                 // 32-row m-tiles per wave (a wave owns 64 output channels x 32 time rows)
    static constexpr int MTILE = WM * MT * 32, TTILE = WN * 32, K = NKC * 64;
    static constexpr int B_CHUNK = TTILE * 128;                  // one 64-channel chunk of the activation tile: rows of 128 B, 16-B slots XOR-swizzled by (row >> 1) & 7
    static constexpr int B_BYTES = NKC * B_CHUNK;
    static constexpr int A_SLOT = MTILE * 64;                    // 32 channels (2 k-steps) of the weight panel in fragment order
    static constexpr int NA = 3;
    static constexpr int A_BASE = B_BYTES;
    static constexpr int RING = B_BYTES + NA * A_SLOT;
    static constexpr int EPI_PITCH = MTILE * 4 + 16, EPI_BYTES = TTILE * EPI_PITCH;
    static constexpr int LDS_BYTES = RING > EPI_BYTES ? RING : EPI_BYTES;
    static constexpr int B_PW = B_BYTES / 1024 / NW;             // 1-KiB DMAs per wave for the activation tile (8 rows x 128 B each)
    static constexpr int A_PW = A_SLOT / 1024 / NW;              // ... per weight slot
    static constexpr int NSTEP = NKC * 2;                        // ring steps of 32 channels
    static constexpr int C8 = MTILE / 8, NIT = TTILE * C8 / NTH, RSTEP = NTH / C8;      // epilogue items: thread -> (row, 8 channels), NIT rows RSTEP apart
    static_assert(B_BYTES % (1024 * NW) == 0 && A_SLOT % (1024 * NW) == 0, "DMA pieces must divide over the waves");
    static_assert(TTILE * C8 % NTH == 0 && NTH % C8 == 0, "epilogue items");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};

template <int WM, int WN, int NKC, int EPI>
__global__ __launch_bounds__(WM * WN * 64, WM * WN / 2) void wn_gemm_wb_kernel(const GemmArgs a) {
    using Cfg = WbCfg<WM, WN, NKC>;
    static_assert(EPI == EPI_STORE_BF16 || EPI == EPI_MASK_STORE, "whole-B kernel: 1x1 convolutions with a row-wise epilogue");
    __shared__ __attribute__((aligned(1024))) char lds[Cfg::LDS_BYTES];
    constexpr int MT = Cfg::MT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int id = blockIdx.x;
    const int xcd = id & 7, q = id >> 3;
    const int mblk = q % a.mblocks;
    const int tile = a.xcd_span > 0 ? xcd * a.xcd_span + q / a.mblocks : (q / a.mblocks) * 8 + xcd;
    if (tile >= a.ntiles) return;
    if (a.kprof && tid == 0) atomicMin(a.kprof, (unsigned long long)wall_clock64());
    // harness diagnostics: a.kclk != null -> thread 0 of workgroup id stamps the 100 MHz wall clock at the phase boundaries of its life into kclk[id * 8 ..]:
    // [0] start, [1] every request issued, [2] first wait satisfied (everything landed), [3] main loop done, [4] epilogue stores issued, [5] stores drained
    unsigned long long* const ph = a.kclk ? a.kclk + (size_t)id * 8 : nullptr;
    auto stamp = [&](int i) __attribute__((always_inline)) { if (ph && tid == 0) ph[i] = (unsigned long long)wall_clock64(); };
    stamp(0);
    const int bl = tile / a.tiles_per_utt;
    const int b = bl + a.b0;
    const int t0 = (tile - bl * a.tiles_per_utt) * Cfg::TTILE;
    const int T = a.T;
    const int64_t rowbase = (int64_t)b * T;
    const int64_t tile_row0 = rowbase + t0;
    const EpiArgs& e = a.e;
    const int mtile_wg = mblk * (WM * MT);

    // ---- 0. accumulators start at the bias of their output channel (as wn_gemm_lds_body).  These loads come FIRST in the vector-memory queue; hipcc waits for
    // them with vmcnt(0) in front of the first MFMA (it does not see the LDS-DMAs in the queue behind them), which is why step 0 below issues its weight
    // slot AFTER its MFMAs: at that wait only requests that landed long ago are outstanding.  (Scalar loads would keep them out of the queue, but hipcc turns
    // the per-half select into vector loads + vmcnt(0) in front of the DMAs: a whole HBM latency before the activation tile is even requested.)
    f32x16_t acc[MT];
    if constexpr (EPI == EPI_STORE_BF16) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int ml = (wm * MT + i) * 32 + qd * 8 + (lane >> 5) * 4;
                float4 bv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (e.bias) bv = *reinterpret_cast<const float4*>(e.bias + mblk * Cfg::MTILE + ml);
                acc[i][qd * 4] = bv.x; acc[i][qd * 4 + 1] = bv.y; acc[i][qd * 4 + 2] = bv.z; acc[i][qd * 4 + 3] = bv.w;
            }
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    }

    // ---- 1. the epilogue's row operand (residual x_l / ReLU mask source), straight into registers: the oldest requests of the workgroup
    const int c8 = tid % Cfg::C8, rl0 = tid / Cfg::C8;
    const int mo = mblk * Cfg::MTILE + c8 * 8;
    uint4 l0[Cfg::NIT];
#pragma unroll
    for (int k = 0; k < Cfg::NIT; ++k) {
        const int trc = min(rl0 + k * Cfg::RSTEP, T - 1 - t0);            // rows past the end of the utterance read the last valid row (never stored)
        l0[k] = make_uint4(0, 0, 0, 0);
        if (EPI == EPI_MASK_STORE || e.in0) l0[k] = *reinterpret_cast<const uint4*>((const bf16_t*)e.in0 + tile_row0 * e.ld_in0 + mo + (uint32_t)(trc * e.ld_in0));
    }
    // ---- 2. the whole activation tile: chunk c (64 channels) = pieces of 8 rows x 128 B; lane -> (row, 16-B slot), slot XOR-swizzled on the SOURCE
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    {
        const SrcSeg& s = a.seg[0];
        const bf16_t* const sbase = s.base + s.col0;
#pragma unroll
        for (int p = 0; p < Cfg::B_PW; ++p) {
            const int piece = wave + p * Cfg::NW;                        // piece -> (chunk, 8-row group)
            const int chunk = piece / (Cfg::TTILE / 8), rg = piece % (Cfg::TTILE / 8);
            const int row = rg * 8 + (lane >> 3);
            const int c16 = (lane & 7) ^ ((row >> 1) & 7);
            const int t = t0 + row;
            const bf16_t* src = (t < T) ? sbase + (rowbase + t) * s.ld + chunk * 64 + c16 * 8 : a.zero;
            lds_dma16(src, __builtin_amdgcn_readfirstlane(lds_base + chunk * Cfg::B_CHUNK + rg * 1024));
        }
    }
    // ---- 3. weight slots: fragment f = mt * 2 + ks of a slot = Apk[mtile_wg + mt][2 * step + ks], 1 KiB each (SGPR-base DMA)
    uint64_t sa[Cfg::A_PW];
#pragma unroll
    for (int p = 0; p < Cfg::A_PW; ++p) {
        const int f = wave + p * Cfg::NW;
        const uint64_t v = (uint64_t)(a.Apk + ((int64_t)(mtile_wg + f / 2) * a.ksteps_total + f % 2) * 512);
        sa[p] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    }
    const uint32_t a_voff = lane * 16;
    auto stage_a = [&](auto slotc) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slotc)::value;
#pragma unroll
        for (int p = 0; p < Cfg::A_PW; ++p) {
            lds_dma16_s(sa[p], a_voff, __builtin_amdgcn_readfirstlane(lds_base + Cfg::A_BASE + SLOT * Cfg::A_SLOT + (wave + p * Cfg::NW) * 1024));
            sa[p] += 2 * 1024;
        }
    };
    stage_a(std::integral_constant<int, 0>{});
    if constexpr (Cfg::NSTEP > 1) stage_a(std::integral_constant<int, 1>{});

    // fragment read offsets: B row = this wave's 32 time rows, swizzled slot; A lane-linear
    const int rrow = wn * 32 + (lane & 31);
    int b_rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_rd[ks] = rrow * 128 + (((ks * 2 + (lane >> 5)) ^ ((rrow >> 1) & 7)) * 16);
    const int a_rd = Cfg::A_BASE + (wm * MT * 2 * 64 + lane) * 16;

    auto step = [&](auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value, SLOT = S % Cfg::NA;
        // everything older than the one younger weight slot has landed: the residual rows, the WHOLE activation tile, this step's weights;
        // lgkmcnt(0): this wave's fragment reads of step S - 1 have returned before the barrier hands their slot to the next DMA
        if constexpr (S + 1 < Cfg::NSTEP) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(Cfg::A_PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (S == 0) stamp(2);
        bf16x8_t af[2][MT], bfr[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
                af[ks][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(lds + a_rd + SLOT * Cfg::A_SLOT + (i * 2 + ks) * 1024));
            bfr[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(lds + (S / 2) * Cfg::B_CHUNK + b_rd[(S % 2) * 2 + ks]));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (S > 0 && S + 2 < Cfg::NSTEP) stage_a(std::integral_constant<int, (S + 2) % Cfg::NA>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks], acc[i], 0, 0, 0);
        if constexpr (S == 0 && S + 2 < Cfg::NSTEP) {      // (step 0: behind the MFMAs, see the note at the accumulators)
            __builtin_amdgcn_sched_barrier(0);
            stage_a(std::integral_constant<int, (S + 2) % Cfg::NA>{});
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    stamp(1);
    [&]<int... S>(std::integer_sequence<int, S...>) { (step(std::integral_constant<int, S>{}), ...); }(std::make_integer_sequence<int, Cfg::NSTEP>{});
    stamp(3);

    // ---- epilogue (wn_gemm_lds_body's, NT = 1, operands already in registers): accumulators -> LDS fp32 [time][channel] -> (row, 8 channels) items
    constexpr int PITCH = Cfg::EPI_PITCH;
    auto epi_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto unpack8 = [](const uint4 x, float* f) {
        f[0] = bf2f((bf16_t)(x.x & 0xffff)); f[1] = bf2f((bf16_t)(x.x >> 16)); f[2] = bf2f((bf16_t)(x.y & 0xffff)); f[3] = bf2f((bf16_t)(x.y >> 16));
        f[4] = bf2f((bf16_t)(x.z & 0xffff)); f[5] = bf2f((bf16_t)(x.z >> 16)); f[6] = bf2f((bf16_t)(x.w & 0xffff)); f[7] = bf2f((bf16_t)(x.w >> 16));
    };
    auto pack8 = [](const float* f) { return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7])); };
    epi_barrier();                                                   // every wave's last fragment reads have returned: the ring becomes the staging area
    {
        const int h = lane >> 5;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int ml = (wm * MT + i) * 32 + qd * 8 + h * 4;
                *reinterpret_cast<float4*>(lds + rrow * PITCH + ml * 4) = make_float4(acc[i][qd * 4], acc[i][qd * 4 + 1], acc[i][qd * 4 + 2], acc[i][qd * 4 + 3]);
            }
    }
    epi_barrier();
    bf16_t* const o0 = (bf16_t*)e.out0 + tile_row0 * e.ld_out0 + mo;
#pragma unroll
    for (int k = 0; k < Cfg::NIT; ++k) {
        const int tr = rl0 + k * Cfg::RSTEP;
        const bool valid = t0 + tr < T;
        const float4 a0 = *reinterpret_cast<const float4*>(lds + tr * PITCH + c8 * 32), a1 = *reinterpret_cast<const float4*>(lds + tr * PITCH + c8 * 32 + 16);
        float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        if constexpr (EPI == EPI_STORE_BF16) {
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
                const uint32_t e0 = (uint32_t)((tile_row0 + tr) * a.drop_ld + mo);
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
        } else {      // EPI_MASK_STORE
            float ref[8]; unpack8(l0[k], ref);
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = (ref[r] > 0.0f) ? v[r] * e.scale : 0.0f;
            if (valid) *reinterpret_cast<uint4*>(o0 + (uint32_t)(tr * e.ld_out0)) = pack8(v);
        }
    }
    if (ph) { stamp(4); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(5); }
    if (a.kprof && tid == 0) atomicMax(a.kprof + 1, (unsigned long long)wall_clock64());
}

// Does this launch fit?  One unshifted source tensor of exactly K = NKC * 64 channels, M a multiple of the tile, no dropout on the source.
template <int WM, int WN, int NKC>
static inline bool wn_gemm_wb_fits(const GemmArgs& a, int M) {
    using Cfg = WbCfg<WM, WN, NKC>;
    return M % Cfg::MTILE == 0 && a.e.M_valid == M && a.zero && a.nseg == 1 && a.nrep == 1 && a.taps == 0 && a.seg[0].shift == 0 && !a.seg[0].dropout
           && a.seg[0].nk == Cfg::K && a.ksteps_total == Cfg::K / 16;
}
template <int WM, int WN, int NKC, int EPI>
static inline int wn_launch_gemm_wb(wn_ctx* ctx, GemmArgs& a, int M, hipStream_t st) {
    using Cfg = WbCfg<WM, WN, NKC>;
    if (ctx && ctx->trace_state == 1 && ctx->trace_n < WN_TRACE_MAX) {      // WN_DEVTRACE: this launch's own stamp slot (as wn_launch_gemm)
        a.kprof = ctx->trace_dev + 2 * ctx->trace_n;
        ctx->trace_tag[ctx->trace_n].epi = EPI; ctx->trace_tag[ctx->trace_n].st = (void*)st; ctx->trace_tag[ctx->trace_n].rows = a.B * a.T; ++ctx->trace_n;
    }
    a.mblocks = M / Cfg::MTILE;
    a.tiles_per_utt = cdiv(a.T, Cfg::TTILE);
    a.ntiles = a.tiles_per_utt * a.B;
    a.xcd_span = cdiv(a.ntiles, 8);
    a.stagger = 0;
    const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
    hipLaunchKernelGGL((wn_gemm_wb_kernel<WM, WN, NKC, EPI>), dim3(grid), dim3(Cfg::NTH), 0, st, a);
    if (ctx) WN_LAUNCH_CHECK(ctx);
    return WN_OK;
}
