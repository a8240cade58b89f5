// MEASURED AND REJECTED (round 4; profiles/r5l_outws_harness.txt): bitwise-identical outputs, but 74 - 77 us for the whole batch against
// 38 - 43 us of the tile-engine launch it was meant to replace (2.4 vs 4.2 - 4.7 TB/s), 42 - 45 vs 23 - 26 us for a half batch, and the
// two-stream layer loop 664 - 679 vs 496 - 518 us.  One tile of look-ahead (32 + 32 KB per CU in flight) leaves every iteration waiting
// for a load issued ~1.5 us earlier; deeper rings need the whole CU's LDS.  The same run shows WHY the two-stream schedule gains so
// little: a half-batch gate launch (45 - 50 us alone) beside the other half's out conv (23 - 26 us alone) takes 76 - 85 us -- the sum,
// not the maximum.  Harness-only: tools/outws_harness.hip.
//
// Weight-stationary streaming kernel for the out conv of the forward chain (modules.py:515-520):
//     x_{l+1}[t] = (W_out^T u_l[t] + b_out + x_l[t]) * rho      (+ the dropout copy the next layer's dilated conv reads, modules.py:484)
// -- 131 kFLOP and 2 KB of HBM traffic per row: HBM-bound by a factor of 5.  The tile engine (wn_tile.h) runs it as a GEMM launch whose
// every 64-row workgroup DMAs the whole 128-KB weight matrix into LDS again, walks 4 K-chunks, and only then touches its residual input
// and its two outputs: a chain of dependent HBM round trips per workgroup with nothing prefetched (measured: 2.2 TB/s live, 3.2 TB/s with
// the GPU to itself, matrix pipe 10 % busy).  Here the weights never move after the first microsecond:
//   * wave w of a 512-thread workgroup owns output channels [32 w, 32 w + 32) and keeps ITS rows of W_out (32 x K bf16, fragment-ordered
//     pack) in VGPRs for the whole launch: K / 16 fragments = 64 VGPRs at K = 256;
//   * the workgroup streams 32-row time tiles: the gate outputs u (the B operand, K x 32) and the residual x (256 x 32) of tile i + 1 are
//     LDS-DMA'd (global_load_lds_dwordx4, 16-B slots XOR-swizzled by the row) while tile i is multiplied -- 2 x (16 + 16) KB of ring, two
//     workgroups per CU (64 KB each: one also fits beside a 72-KB gate workgroup of the other half batch);
//   * 16 MFMAs (v_mfma_f32_32x32x16_bf16) per wave and tile, accumulators started at the bias; the residual is read from LDS in the
//     accumulator's own layout; x_{l+1} and its dropout copy are staged back into the (dead) LDS tiles in row layout and leave as fully
//     covered 512-B rows, 16 B per lane;
//   * per tile: 64 KB of HBM traffic (16 in + 16 in + 32 out), 3 workgroup barriers, no atomics, no flags; tiles are dealt statically
//     (XCD x walks the contiguous span x of the launch's tiles like the tile engine, so a tile's rows are written by the XCD whose
//     gate workgroups read them next).
// Same k order as the tile engine (one accumulator per output, k-steps in order) => bitwise the same x_{l+1} / dropout copy
// (tools/outws_harness.hip checks exactly that).  Shapes: M = R = 256, K = G / 2 = 256 (paper width); other widths keep the tile engine.
#pragma once
#include "wn_tile.h"

#define OUTWS_TT 32
template <int KS>      // k-steps of 16: K = 16 KS channels of the B operand; geometry below needs K == M == 256
__global__ __launch_bounds__(512, 4) void wn_out_ws_kernel(const GemmArgs a) {
    constexpr int K = KS * 16, M = 256, TT = OUTWS_TT;
    static_assert(K == 256, "staging reuses the u tile for the dropout copy: row geometry of u and x must agree");
    constexpr int ROWB = 512;                       // bytes per staged row (256 bf16), 32 slots of 16 B
    constexpr int TILEB = TT * ROWB;                // 16 KB
    __shared__ __attribute__((aligned(1024))) char lds[4 * TILEB];      // UB[0] UB[1] XB[0] XB[1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int T = a.T;
    if (a.kprof && tid == 0) atomicMin(a.kprof, (unsigned long long)wall_clock64());

    // ---- this wave's rows of W_out: K / 16 fragments of the fragment-ordered pack [mtile][kstep][lane][8], resident for the launch
    bf16x8_t wfrag[KS];
    {
        const bf16_t* Ap = a.Apk + ((int64_t)wave * a.ksteps_total * 64 + lane) * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wfrag[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Ap + (int64_t)ks * 512));
    }
    const EpiArgs& e = a.e;
    const bf16_t* const Ub = a.seg[0].base + a.seg[0].col0;
    const int ldu = a.seg[0].ld;
    const bf16_t* const Xb = (const bf16_t*)e.in0;
    const bool has_xd = e.out1 != nullptr;
    // accumulator start = bias of the output channel (acc[r] <-> channel 32 wave + 8 (r >> 2) + 4 (lane >> 5) + (r & 3), time row lane & 31)
    const int h = lane >> 5, tr = lane & 31;
    f32x16_t acc0;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e.bias) bv = *reinterpret_cast<const float4*>(e.bias + wave * 32 + qd * 8 + h * 4);
        acc0[qd * 4] = bv.x; acc0[qd * 4 + 1] = bv.y; acc0[qd * 4 + 2] = bv.z; acc0[qd * 4 + 3] = bv.w;
    }

    // tiles of this workgroup: span `xcd` of the launch, every nslots-th tile from `slot`
    const int ntl = a.ntiles, span = a.xcd_span;
    const int first = slot, last_excl = min(span, ntl - xcd * span);
    auto tile_rows = [&](int i, int64_t& row0, int& t0) {      // i-th tile of the span -> first row (absolute) and time index
        const int tile = xcd * span + i;
        const int bl = tile / a.tiles_per_utt;
        t0 = (tile - bl * a.tiles_per_utt) * TT;
        row0 = (int64_t)(bl + a.b0) * T + t0;
    };
    // DMA of one tile: 16 + 16 wave-wide pieces of 1 KB (2 rows each); wave w issues pieces w and w + 8 of both operands.
    // lane -> (row, physical slot); the SOURCE slot is the swizzled one, so LDS holds row r's logical slot s at slot s ^ (r & 15)
    auto dma_tile = [&](int i, int buf) {
        int64_t row0; int t0; tile_rows(i, row0, t0);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int piece = wave + 8 * p;
            const int r = piece * 2 + (lane >> 5), ps = lane & 31, ls = ps ^ (r & 15);
            const bool ok = t0 + r < T;
            const bf16_t* su = ok ? Ub + (row0 + r) * ldu + ls * 8 : a.zero;
            const bf16_t* sx = ok ? Xb + (row0 + r) * e.ld_in0 + ls * 8 : a.zero;
            lds_dma16(su, __builtin_amdgcn_readfirstlane(lds_addr_of(lds + buf * TILEB + piece * 1024)));
            lds_dma16(sx, __builtin_amdgcn_readfirstlane(lds_addr_of(lds + (2 + buf) * TILEB + piece * 1024)));
        }
    };
    if (first >= last_excl) { if (a.kprof && tid == 0) atomicMax(a.kprof + 1, (unsigned long long)wall_clock64()); return; }
    dma_tile(first, 0);
    int it = 0; bool prev_partial = false;
    for (int i = first; i < last_excl; i += nslots, ++it) {
        const int cur = it & 1, nxt = cur ^ 1;
        char* const ub = lds + cur * TILEB;
        char* const xb = lds + (2 + cur) * TILEB;
        // tile i landed; the stores of tile i - 1 (issued AFTER these DMAs: vmcnt retires in order) may stay in flight.  lgkmcnt(0): this
        // wave's staging reads of tile i - 1 have returned before the barrier hands those buffers to the DMA of tile i + 1
        // (a tile that ends past the utterance skips some stores wave by wave: the count no longer holds, drain instead -- once per utterance)
        if (it == 0 || prev_partial) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (has_xd) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (i + nslots < last_excl) dma_tile(i + nslots, nxt);
        else {      // keep the count of vector-memory operations per iteration constant (counted waits above)
#pragma unroll
            for (int p = 0; p < 4; ++p) lds_dma16(a.zero, __builtin_amdgcn_readfirstlane(lds_addr_of(lds + (p < 2 ? nxt : 2 + nxt) * TILEB + (wave + 8 * (p & 1)) * 1024)));
        }
        int64_t row0; int t0; tile_rows(i, row0, t0);
        // ---- residual in the accumulator's layout, then the contraction
        const int swz = (tr & 15);
        uint2 xr[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) xr[qd] = *reinterpret_cast<const uint2*>(xb + tr * ROWB + (((wave * 4 + qd) ^ swz) << 4) + h * 8);
        f32x16_t acc = acc0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(ub + tr * ROWB + (((ks * 2 + h) ^ swz) << 4)));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[ks], bf, acc, 0, 0, 0);
        }
        // ---- epilogue in registers: + residual, * rho, round; dropout of the rounded value (tf.layers.dropout of the next layer's input)
        uint2 xo[4], xd[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            float v[4] = {acc[qd * 4], acc[qd * 4 + 1], acc[qd * 4 + 2], acc[qd * 4 + 3]};
            v[0] += bf2f((bf16_t)(xr[qd].x & 0xffff)); v[1] += bf2f((bf16_t)(xr[qd].x >> 16));
            v[2] += bf2f((bf16_t)(xr[qd].y & 0xffff)); v[3] += bf2f((bf16_t)(xr[qd].y >> 16));
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] *= e.scale; if (e.relu) v[r] = fmaxf(v[r], 0.0f); }
            xo[qd] = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            if (has_xd) {
                const int m = wave * 32 + qd * 8 + h * 4;
                const uint32_t e0 = (uint32_t)((row0 + tr) * a.drop_ld + m);
                uint32_t w0, w1; wn_drop_quad(a.key_lo, a.key_hi, e0 >> 2, w0, w1);
                const float x0 = bf2f((bf16_t)(xo[qd].x & 0xffff)), x1 = bf2f((bf16_t)(xo[qd].x >> 16));
                const float x2 = bf2f((bf16_t)(xo[qd].y & 0xffff)), x3 = bf2f((bf16_t)(xo[qd].y >> 16));
                const float d0 = ((w0 & 0xffffu) >= a.thresh16) ? x0 * a.keep_scale : 0.0f, d1 = ((w0 >> 16) >= a.thresh16) ? x1 * a.keep_scale : 0.0f;
                const float d2 = ((w1 & 0xffffu) >= a.thresh16) ? x2 * a.keep_scale : 0.0f, d3 = ((w1 >> 16) >= a.thresh16) ? x3 * a.keep_scale : 0.0f;
                xd[qd] = make_uint2(pack_bf2(d0, d1), pack_bf2(d2, d3));
            }
        }
        // x' into the residual tile (this wave's own channel slice: no other wave reads it), the dropout copy into the u tile once
        // every wave has finished its fragment reads
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<uint2*>(xb + tr * ROWB + (((wave * 4 + qd) ^ swz) << 4) + h * 8) = xo[qd];
        if (has_xd) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<uint2*>(ub + tr * ROWB + (((wave * 4 + qd) ^ swz) << 4) + h * 8) = xd[qd];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- whole rows out: 1024 pieces of 16 B per tensor, two per thread; a wave covers two full 512-B rows per instruction
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int id = tid + 512 * p, r = id >> 5, ps = id & 31, ls = ps ^ (r & 15);
            const uint4 vx = *reinterpret_cast<const uint4*>(xb + r * ROWB + (ps << 4));
            uint4 vd = make_uint4(0, 0, 0, 0);
            if (has_xd) vd = *reinterpret_cast<const uint4*>(ub + r * ROWB + (ps << 4));
            if (t0 + r < T) {
                *reinterpret_cast<uint4*>((bf16_t*)e.out0 + (row0 + r) * e.ld_out0 + ls * 8) = vx;
                if (has_xd) *reinterpret_cast<uint4*>((bf16_t*)e.out1 + (row0 + r) * e.ld_out1 + ls * 8) = vd;
            }
        }
        prev_partial = t0 + TT > T;
    }
    if (a.kprof && tid == 0) atomicMax(a.kprof + 1, (unsigned long long)wall_clock64());
}

// does this EPI_STORE_BF16 launch take the streaming kernel?  (out conv of the paper-width stack; everything else: tile engine)
static inline bool wn_out_ws_fits(const GemmArgs& a, int M) {
    return M == 256 && a.e.M_valid == 256 && a.nseg == 1 && a.nrep == 1 && a.taps == 0 && a.seg[0].nk == 256 && a.seg[0].shift == 0 && !a.seg[0].dropout &&
           a.seg[0].ld == 256 && a.e.in0 && a.e.ld_in0 == 256 && a.e.ld_out0 == 256 && (!a.e.out1 || a.e.ld_out1 == 256) && a.zero && a.ksteps_total == 16 &&
           (!a.e.out1 || a.drop_ld == 256);
}
static inline int wn_launch_out_ws(wn_ctx* ctx, GemmArgs& a, hipStream_t st) {
    a.mblocks = 1;
    a.tiles_per_utt = cdiv(a.T, OUTWS_TT);
    a.ntiles = a.tiles_per_utt * a.B;
    a.xcd_span = cdiv(a.ntiles, 8);
    const int per_xcd = a.xcd_span < 64 ? a.xcd_span : 64;      // 2 workgroups per CU x 32 CUs per XCD
    hipLaunchKernelGGL((wn_out_ws_kernel<16>), dim3(per_xcd * 8), dim3(512), 0, st, a);
    if (ctx) WN_LAUNCH_CHECK(ctx);
    return WN_OK;
}
