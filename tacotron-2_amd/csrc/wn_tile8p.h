// 8-phase "ping-pong" main loop for the MFMA-bound contractions of the residual stack (gate, d x) on gfx950.
//
// Same contraction, operands and fused epilogues as wn_gemm_lds_kernel (wn_tile.h):  Out^T[m, t] = sum_k Wpk[m, k] * Act[t, k],
// both operands through LDS by LDS-DMA, v_mfma_f32_32x32x16_bf16.  What differs is the SCHEDULE -- the 256 x 256 x 64 structure of
// /opt/skills/guides/cdna_hip_programming.md section 5 ("The 256^2 8-phase template"), rebuilt for this contraction:
//   * ONE 8-wave workgroup per CU computes 256 output channels x 256 time rows; K-tiles of 64 channels; LDS holds two K-tiles, each as
//     four 16-KiB half-tiles {B-h0, B-h1, A-h0, A-h1} (128 rows x 64 k); 128 KiB ring, 133 KB with the epilogue's fp32 staging.
//   * waves are 2 (M) x 4 (N); a wave owns four 64 x 32 QUADRANTS strided by 128 in both dimensions (quadrant (mq, nq) = output rows
//     mq*128 + wm*64 .. +64, time rows nq*128 + wn*32 .. +32), so half-tile X-h0 holds what every wave needs for its quadrants with
//     index 0 and X-h1 for index 1.  A K-tile is four PHASES, one quadrant each:
//         phase 1: read B-h0 (4 x ds_read_b128) + A-h0 (8)   MFMA q00      stage A-h1 of K-tile kt+1
//         phase 2: read B-h1 (4)                             MFMA q01      stage B-h0 of K-tile kt+2
//         phase 3: read A-h1 (8)                             MFMA q11      stage A-h0 of K-tile kt+2
//         phase 4: --                                        MFMA q10      stage B-h1 of K-tile kt+2, s_waitcnt vmcnt(6)
//     phase = { reads + 2 LDS-DMAs per wave ; s_barrier ; lgkmcnt(0) ; s_setprio 1 ; 8 MFMAs (256 cycles) ; s_setprio 0 ; s_barrier }.
//   * the two wave ROWS (wm = 0 / 1: one wave of each on every SIMD) run one barrier apart (wm = 1 takes one extra s_barrier up front,
//     wm = 0 one at the end): while one row is inside its MFMA block the other issues its fragment reads and DMAs -- every SIMD always
//     has exactly one wave in a matrix block and one feeding LDS, instead of four free-running waves that all queue on the DMA path
//     in front of their own MFMAs (wn_gemm_lds_kernel: matrix pipe busy 39 %, waves at s_waitcnt / barriers 49 % of their cycles).
//   * DMA never drains in the loop: three half-tiles stay in flight across the barriers (vmcnt(6) once per K-tile).  Ordering rules
//     (guide, section 5): a half-tile is READ no earlier than the phase after the vmcnt that retires it (RAW: the wait sits in front
//     of the phase's first barrier, both wave rows have passed it one barrier later); a region is RE-STAGED two phases after its last
//     read, or one phase after when an lgkmcnt in front of the reading phase's first barrier retired those reads (phase 1: the four
//     B-h0 reads are issued first and `lgkmcnt(8)` retires them, so B-h0 can be re-staged in phase 2).
// K order: the packs interleave the three dilated taps in 64-channel blocks (PackedW::kil = 64): K-tile kt < 3*nk/64 is tap kt % 3
// of block kt / 3 -- one row shift per K-tile --, then the sequential segments (conditioning) in 64-channel K-tiles; a partial last
// K-tile multiplies only its valid k-steps; an odd K-tile count is padded with an all-zero tile.
#pragma once
#include "wn_tile.h"

namespace p8 {
constexpr int BM = 256, BT = 256, BK = 64, NTH = 512;
constexpr int HALF = 128 * BK * 2;                                             // one half-tile: 16 KiB
constexpr int b_off(int buf, int h) { return (buf * 2 + h) * HALF; }           // activation halves first: every fragment address of either operand is one VGPR + a 16-bit immediate
constexpr int A_REGION = 4 * HALF;
constexpr int a_off(int buf, int h) { return A_REGION + (buf * 2 + h) * HALF; }
constexpr int RING = 8 * HALF;                                                 // 128 KiB
constexpr int EPI_PITCH = BM * 4 + 16, EPI_ROWS = 128;                         // fp32 staging of the epilogue: one pass = the 128 time rows of quadrant column nq
constexpr int LDS_BYTES = EPI_PITCH * EPI_ROWS > RING ? EPI_PITCH * EPI_ROWS : RING;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
}

__device__ __forceinline__ uint64_t p8_sgpr64(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// ABL (harness only): 1 = main loop alone (the accumulators are kept live, nothing is stored); 2 = whole kernel without the prologue's wait for K-tile 0 (results wrong:
// the upper bound of what a persistent grid that prefetches the next tile under the epilogue could save)
// SCHED: 0 = LDS-DMAs in the load section of a phase (the guide's template); 1 = between the MFMAs of its matrix block
template <int EPI, int ABL, int SCHED>
__device__ __forceinline__ void wn_gemm8p_body(const GemmArgs& a, char* const lds) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // workgroup -> (M block, time tile): as wn_gemm_lds_body (M blocks of one activation tile on one XCD, contiguous tile run per XCD)
    const int id = blockIdx.x;
    const int xcd = id & 7, q = id >> 3;
    const int mblk = q % a.mblocks;
    const int tile = a.xcd_span > 0 ? xcd * a.xcd_span + q / a.mblocks : (q / a.mblocks) * 8 + xcd;
    if (tile >= a.ntiles) return;
    if (a.kprof && tid == 0) atomicMin(a.kprof, (unsigned long long)wall_clock64());
    if (a.kclk && id == 0 && tid == 0) { a.kclk[0] = __builtin_amdgcn_s_memtime(); a.kclk[1] = (unsigned long long)wall_clock64(); }
    const int bl = tile / a.tiles_per_utt;
    const int b = bl + a.b0;
    const int t0 = (tile - bl * a.tiles_per_utt) * p8::BT;
    const int T = a.T;
    const int64_t rowbase = (int64_t)b * T;
    const int h5 = lane >> 5;

    // accumulators start at the bias (gate, 1x1 convs): acc[mq][nq][ii][r] is output row mq*128 + wm*64 + ii*32 + (r/4)*8 + h5*4 + r%4
    f32x16_t acc[2][2][2];
    if constexpr (EPI == EPI_GATE || EPI == EPI_STORE_BF16) {
        const float* bp = a.e.bias;
        if constexpr (EPI == EPI_GATE) bp += (int64_t)b * a.e.bias_bstride;
#pragma unroll
        for (int mq = 0; mq < 2; ++mq)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int ml = mq * 128 + wm * 64 + ii * 32 + qd * 8 + h5 * 4;
                    float4 bv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (bp) {
                        if constexpr (EPI == EPI_GATE) {
                            const int gl = (ml >> 6) * 32 + (ml & 31);      // packed gate rows: 64-row groups [32 tanh | their 32 sigmoid partners]
                            bv = *reinterpret_cast<const float4*>(bp + ((ml & 32) ? a.e.GH : 0) + mblk * (p8::BM / 2) + gl);
                        } else bv = *reinterpret_cast<const float4*>(bp + mblk * p8::BM + ml);
                    }
#pragma unroll
                    for (int nq = 0; nq < 2; ++nq) { acc[mq][nq][ii][qd * 4] = bv.x; acc[mq][nq][ii][qd * 4 + 1] = bv.y; acc[mq][nq][ii][qd * 4 + 2] = bv.z; acc[mq][nq][ii][qd * 4 + 3] = bv.w; }
                }
    } else {
#pragma unroll
        for (int mq = 0; mq < 2; ++mq)
#pragma unroll
            for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mq][nq][ii][r] = 0.0f;
    }

    // ---- K-tile list
    const int ntap_kt = a.taps ? 3 * (a.seg[0].nk >> 6) : 0;
    int nkt_real = ntap_kt;
    if (a.nseg > 3) nkt_real += (a.seg[3].nk + 63) >> 6;      // (constant indices only: a dynamic one sends the argument struct to scratch)
    const int nkt = (nkt_real + 1) & ~1;
    const int last_ks = min(4, max(0, a.ksteps_total - (nkt - 1) * 4));          // valid k-steps of the final K-tile

    // ---- B staging geometry.  A DMA piece is 8 rows x 128 B; wave w stages pieces w and w + 8 of a half-tile: row (in the 256-row tile)
    // = H*128 + p*64 + r0, r0 = w*8 + lane/8; the lane's 16-B slot is XOR-swizzled on the SOURCE (and again on the read): c8 does not
    // depend on (H, p) because their row offsets are multiples of 16.
    const int r0 = wave * 8 + (lane >> 3);
    const int c8 = ((lane & 7) ^ ((r0 >> 1) & 7)) * 8;                           // first channel (of the K-tile's 64) this lane fetches
    const int rowg = (int)rowbase + t0 + r0;                                      // global row of (H = 0, p = 0)
    const int ldT = a.seg[0].ld;
    const uint32_t offT = (uint32_t)((rowg * ldT + c8) * 2);                      // byte offset inside the tap tensor (wn_gemm8p_fits: < 2^31)
    const int sh0 = a.seg[0].shift, sh1 = a.seg[1].shift, sh2 = a.seg[2].shift;
    uint32_t vmask = 0;                                                           // bit k*4 + H*2 + p: row of tap k in range; bit 12 + H*2 + p: row < T
#pragma unroll
    for (int hp = 0; hp < 4; ++hp) {
        const int t = t0 + r0 + (hp >> 1) * 128 + (hp & 1) * 64;
        const bool in = t < T;
        if (in && t + sh0 >= 0 && t + sh0 < T) vmask |= 1u << hp;
        if (in && t + sh1 >= 0 && t + sh1 < T) vmask |= 1u << (4 + hp);
        if (in && t + sh2 >= 0 && t + sh2 < T) vmask |= 1u << (8 + hp);
        if (in) vmask |= 1u << (12 + hp);
    }
    // bit g: every lane of this WAVE has all four rows of group g (tap 0 / 1 / 2 / unshifted) in range -- all tiles but those at the ends of an
    // utterance: their DMAs take the SGPR-base form (no per-lane address select)
    uint32_t fastbits = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) if (__all(((vmask >> (4 * g)) & 15u) == 15u)) fastbits |= 1u << g;
    fastbits = __builtin_amdgcn_readfirstlane(fastbits);
    const char* const zero = (const char*)a.zero;

    // ---- iterator over the B sources of the K-tiles, in staging order.  Tap K-tiles: the three taps' sources rotate, and a source that has been
    // used is re-queued one 64-channel block further (branch-free: no division, no select between variables -- that becomes a dynamic
    // index into the closure, i.e. scratch).  Then the sequential segment in 64-channel K-tiles, then (odd count) an all-zero K-tile.
    int it_kt = -1, it_c = 0; bool it_more = a.nseg > 3;
    uint64_t rs0 = (uint64_t)(a.seg[0].base + a.seg[0].col0) + (int64_t)sh0 * ldT * 2, rs1 = rs0 + (int64_t)(sh1 - sh0) * ldT * 2, rs2 = rs0 + (int64_t)(sh2 - sh0) * ldT * 2;
    int q0 = 0, q1 = 4, q2 = 8;
    uint64_t it_src = (uint64_t)zero; int it_rs = ldT * 128, it_mbit = 0, it_kc = 64; uint32_t it_off = offT, it_okb = 0; bool it_fast = false;
    auto b_next = [&]() __attribute__((always_inline)) {
        ++it_kt;
        if (it_kt < ntap_kt) {
            it_src = rs0; rs0 = rs1; rs1 = rs2; rs2 = it_src + 128;
            it_mbit = q0; q0 = q1; q1 = q2; q2 = it_mbit;
        } else if (it_more) {
            const int sld = a.seg[3].ld, snk = a.seg[3].nk;                       // the one sequential segment behind the taps (conditioning)
            it_off = (uint32_t)((rowg * sld + c8) * 2);
            it_src = (uint64_t)(a.seg[3].base + a.seg[3].col0 + it_c); it_rs = sld * 128; it_mbit = 12; it_kc = snk - it_c;
            it_c += 64;
            it_more = it_c < snk;
        } else { it_src = (uint64_t)zero; it_rs = 0; it_mbit = 12; it_kc = 0; }
        it_okb = c8 < it_kc ? (vmask >> it_mbit) & 15u : 0u;
        it_fast = it_kc >= 64 && ((fastbits >> (it_mbit >> 2)) & 1u);
    };

    const uint32_t lds_base = __builtin_amdgcn_readfirstlane(lds_addr_of(lds));
    auto stageB1 = [&](auto bufc, auto hc, const int p) __attribute__((always_inline)) {          // piece p (0 / 1) of half-tile B-h<H>
        constexpr int BUF = decltype(bufc)::value, H = decltype(hc)::value;
        const uint64_t sb = it_src + (uint64_t)((H * 2 + p) * it_rs);
        if (it_fast) lds_dma16_s(p8_sgpr64(sb), it_off, lds_base + p8::b_off(BUF, H) + (wave + 8 * p) * 1024);
        else {
            const char* src = ((it_okb >> (H * 2 + p)) & 1u) ? (const char*)sb + it_off : zero;
            lds_dma16(src, lds_base + p8::b_off(BUF, H) + (wave + 8 * p) * 1024);
        }
    };
    auto stageB = [&](auto bufc, auto hc) __attribute__((always_inline)) { stageB1(bufc, hc, 0); stageB1(bufc, hc, 1); };
    // A: fragment f = wave + 8p of a half-tile = (m-tile f / 4 of the half, k-step f % 4); the pack is fragment ordered, one fragment = 1 KiB
    const uint64_t mrow = (uint64_t)a.ksteps_total * 1024;                        // bytes between consecutive 32-row m-tiles of the pack
    const uint64_t abase_w = (uint64_t)a.Apk + ((uint64_t)(mblk * 8 + (wave >> 2)) * a.ksteps_total + (wave & 3)) * 1024;
    const uint32_t a_voff = lane * 16;
    auto stageA1 = [&](auto bufc, auto hc, const int kt, const int p) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, H = decltype(hc)::value;
        const bool ok = kt * 4 + (wave & 3) < a.ksteps_total;                     // k-steps past the end of the pack come from the zero page
        const uint64_t sb = ok ? abase_w + (uint64_t)kt * 4096 + (uint64_t)(H * 4 + 2 * p) * mrow : (uint64_t)zero;
        lds_dma16_s(p8_sgpr64(sb), a_voff, lds_base + p8::a_off(BUF, H) + (wave + 8 * p) * 1024);
    };
    auto stageA = [&](auto bufc, auto hc, const int kt) __attribute__((always_inline)) { stageA1(bufc, hc, kt, 0); stageA1(bufc, hc, kt, 1); };

    // fragment read offsets (loop invariant)
    const int rrow = wn * 32 + (lane & 31);
    int b_rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_rd[ks] = rrow * 128 + (((ks * 2 + h5) ^ ((rrow >> 1) & 7)) * 16);
    const int a_rd = p8::A_REGION + wm * 8192 + lane * 16;

    // the MFMA block of a phase; f0 / f1 run between its MFMA pairs (SCHED 1: the phase's two LDS-DMAs and the iterator's scalar bookkeeping
    // ride in the issue slots the matrix pipe leaves free, instead of lengthening the load section the OTHER wave row's MFMA block waits for)
    auto mma = [&](f32x16_t (&c2)[2], const bf16x8_t (&A)[2][4], const bf16x8_t (&Bv)[4], const int nks, auto f0, auto f1) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < nks) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) c2[ii] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ii][ks], Bv[ks], c2[ii], 0, 0, 0);
            }
            if (ks == 0) { __builtin_amdgcn_sched_barrier(0); f0(); __builtin_amdgcn_sched_barrier(0); }
            if (ks == 2) { __builtin_amdgcn_sched_barrier(0); f1(); __builtin_amdgcn_sched_barrier(0); }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto nofill = []() __attribute__((always_inline)) {};

    // one K-tile = four phases.  TAIL 0: steady state; 1: second-to-last K-tile (nothing left to stage for kt + 2); 2: last K-tile
    auto ktile = [&](auto bufc, auto tailc, const int kt) __attribute__((always_inline)) {
        constexpr int BUF = decltype(bufc)::value, TAIL = decltype(tailc)::value;
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using IB = std::integral_constant<int, BUF>; using IO = std::integral_constant<int, BUF ^ 1>;
        const int nks = TAIL == 2 ? last_ks : 4;
        bf16x8_t A0[2][4], A1[2][4], B0[4], B1[4];
        // ---- phase 1
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) B0[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(lds + b_rd[ks] + p8::b_off(BUF, 0)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) A0[ii][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(lds + a_rd + (p8::a_off(BUF, 0) - p8::A_REGION) + (ii * 4 + ks) * 1024));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TAIL < 2 && SCHED == 0) stageA(IO{}, I1{}, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SCHED == 0) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");          // the four B-h0 reads (issued first) have returned: B-h0 may be re-staged in phase 2
        __builtin_amdgcn_s_barrier();
        if constexpr (TAIL < 2 && SCHED == 1) mma(acc[0][0], A0, B0, nks, [&]() __attribute__((always_inline)) { stageA1(IO{}, I1{}, kt + 1, 0); }, [&]() __attribute__((always_inline)) { stageA1(IO{}, I1{}, kt + 1, 1); });
        else mma(acc[0][0], A0, B0, nks, nofill, nofill);
        __builtin_amdgcn_s_barrier();
        // ---- phase 2
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) B1[ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(lds + b_rd[ks] + p8::b_off(BUF, 1)));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TAIL == 0 && SCHED == 0) { b_next(); stageB(IB{}, I0{}); }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if constexpr (TAIL == 0 && SCHED == 1) mma(acc[0][1], A0, B1, nks, [&]() __attribute__((always_inline)) { b_next(); stageB1(IB{}, I0{}, 0); }, [&]() __attribute__((always_inline)) { stageB1(IB{}, I0{}, 1); });
        else mma(acc[0][1], A0, B1, nks, nofill, nofill);
        __builtin_amdgcn_s_barrier();
        // ---- phase 3
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) A1[ii][ks] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(lds + a_rd + (p8::a_off(BUF, 1) - p8::A_REGION) + (ii * 4 + ks) * 1024));
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (TAIL == 0 && SCHED == 0) stageA(IB{}, I0{}, kt + 2);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if constexpr (TAIL == 0 && SCHED == 1) mma(acc[1][1], A1, B1, nks, [&]() __attribute__((always_inline)) { stageA1(IB{}, I0{}, kt + 2, 0); }, [&]() __attribute__((always_inline)) { stageA1(IB{}, I0{}, kt + 2, 1); });
        else mma(acc[1][1], A1, B1, nks, nofill, nofill);
        __builtin_amdgcn_s_barrier();
        // ---- phase 4: K-tile kt + 1 has landed once only the three youngest half-tiles (those of kt + 2) are in flight
        // (SCHED 1: this phase's own DMAs are issued in its MFMA block, after the wait: only the pieces of phases 2 and 3 are younger)
        if constexpr (TAIL == 0 && SCHED == 0) { stageB(IB{}, I1{}); asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
        else if constexpr (TAIL == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (TAIL == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if constexpr (TAIL == 0 && SCHED == 1) mma(acc[1][0], A1, B0, nks, [&]() __attribute__((always_inline)) { stageB1(IB{}, I1{}, 0); }, [&]() __attribute__((always_inline)) { stageB1(IB{}, I1{}, 1); });
        else mma(acc[1][0], A1, B0, nks, nofill, nofill);
        __builtin_amdgcn_s_barrier();
    };

    {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
        // prologue: K-tile 0 whole, K-tile 1 without A-h1 (phase 1 of K-tile 0 stages it); 14 DMAs per wave, the first 8 must have landed
        b_next(); stageB(I0{}, I0{}); stageA(I0{}, I0{}, 0); stageB(I0{}, I1{}); stageA(I0{}, I1{}, 0);
        b_next(); stageB(I1{}, I0{}); stageA(I1{}, I0{}, 1); stageB(I1{}, I1{});
        if constexpr (ABL != 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // (ABL 2, harness only: start without waiting for K-tile 0 -- wrong results, the time of a kernel whose prologue is hidden)
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();                   // wave row 1 runs one barrier behind row 0
        for (int kt = 0; kt < nkt - 2; kt += 2) { ktile(I0{}, I0{}, kt); ktile(I1{}, I0{}, kt + 1); }
        ktile(I0{}, I1{}, nkt - 2);
        ktile(I1{}, I2{}, nkt - 1);
        if (wm == 0) __builtin_amdgcn_s_barrier();
    }

    if constexpr (ABL == 1) {
#pragma unroll
        for (int mq = 0; mq < 2; ++mq)
#pragma unroll
            for (int nq = 0; nq < 2; ++nq)
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) asm volatile("" ::"v"(acc[mq][nq][ii]));
        if (a.kprof && tid == 0) atomicMax(a.kprof + 1, (unsigned long long)wall_clock64());
        return;
    }

    // ---- epilogue: accumulators -> LDS (fp32 [time row][channel]) -> one (row, 8 channels) item per thread; the arithmetic of every EPI is
    // that of wn_gemm_lds_body's epilogue (wn_tile.h), item for item
    constexpr int PITCH = p8::EPI_PITCH, PROWS = p8::EPI_ROWS, NTH = p8::NTH;
    const EpiArgs& e = a.e;
    auto epi_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto unpack8 = [](const uint4 x, float* f) {
        f[0] = bf2f((bf16_t)(x.x & 0xffff)); f[1] = bf2f((bf16_t)(x.x >> 16)); f[2] = bf2f((bf16_t)(x.y & 0xffff)); f[3] = bf2f((bf16_t)(x.y >> 16));
        f[4] = bf2f((bf16_t)(x.z & 0xffff)); f[5] = bf2f((bf16_t)(x.z >> 16)); f[6] = bf2f((bf16_t)(x.w & 0xffff)); f[7] = bf2f((bf16_t)(x.w >> 16));
    };
    auto pack8 = [](const float* f) { return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7])); };
    const int64_t tile_row0 = rowbase + t0;
    auto write_acc = [&](const int nq) __attribute__((always_inline)) {
#pragma unroll
        for (int mq = 0; mq < 2; ++mq)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int ml = mq * 128 + wm * 64 + ii * 32 + qd * 8 + h5 * 4;
                    *reinterpret_cast<float4*>(lds + rrow * PITCH + ml * 4) =
                        make_float4(acc[mq][nq][ii][qd * 4], acc[mq][nq][ii][qd * 4 + 1], acc[mq][nq][ii][qd * 4 + 2], acc[mq][nq][ii][qd * 4 + 3]);
                }
    };
#pragma unroll
    for (int nq = 0; nq < 2; ++nq) {
        if constexpr (EPI == EPI_GATE) {
            epi_barrier();
            write_acc(nq);
            epi_barrier();
            constexpr int GT = p8::BM / 2, C8 = GT / 8, ITEMS = PROWS * C8;
            constexpr int NIT = ITEMS / NTH, RSTEP = NTH / C8;
            const int c8i = tid % C8, rl0 = tid / C8;
            const int gl = c8i * 8, ml = (gl >> 5) * 64 + (gl & 31);
            bf16_t* const TSb = (bf16_t*)e.out0 + tile_row0 * e.ld_out0 + mblk * GT + gl;
            bf16_t* const Ub = (bf16_t*)e.out1 + tile_row0 * e.ld_out1 + mblk * GT + gl;
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int rl = rl0 + k * RSTEP;
                const int tr = nq * 128 + rl;
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
                    *reinterpret_cast<uint4*>(TSb + (uint32_t)(tr * e.ld_out0)) = make_uint4(ps[0], ps[1], ps[2], ps[3]);
                    *reinterpret_cast<uint4*>(Ub + (uint32_t)(tr * e.ld_out1)) = make_uint4(pu[0], pu[1], pu[2], pu[3]);
                }
            }
        } else {
            constexpr int C8 = p8::BM / 8, ITEMS = PROWS * C8;
            constexpr int NIT = ITEMS / NTH, RSTEP = NTH / C8;
            const int c8i = tid % C8, rl0 = tid / C8;
            const int mo = mblk * p8::BM + c8i * 8;
            uint4 l0[NIT], l1[NIT];
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int tr = nq * 128 + rl0 + k * RSTEP;
                const int trc = min(tr, T - 1 - t0);
                l0[k] = make_uint4(0, 0, 0, 0); l1[k] = make_uint4(0, 0, 0, 0);
                if constexpr (EPI == EPI_DGATE) {
                    l0[k] = *reinterpret_cast<const uint4*>((const bf16_t*)e.in1 + tile_row0 * e.ld_in0 + mo + (uint32_t)(trc * e.ld_in0));
                    l1[k] = *reinterpret_cast<const uint4*>((const bf16_t*)e.in0 + tile_row0 * e.ld_in0 + mo + (uint32_t)(trc * e.ld_in0));
                } else if constexpr (EPI == EPI_MASK_STORE) {
                    l0[k] = *reinterpret_cast<const uint4*>((const bf16_t*)e.in0 + tile_row0 * e.ld_in0 + mo + (uint32_t)(trc * e.ld_in0));
                } else {
                    if (e.in0) l0[k] = *reinterpret_cast<const uint4*>((const bf16_t*)e.in0 + tile_row0 * e.ld_in0 + mo + (uint32_t)(trc * e.ld_in0));
                }
            }
            epi_barrier();
            write_acc(nq);
            epi_barrier();
            bf16_t* const o0 = (bf16_t*)e.out0 + tile_row0 * e.ld_out0 + mo;
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int rl = rl0 + k * RSTEP, tr = nq * 128 + rl;
                const bool valid = t0 + tr < T;
                const float4 a0 = *reinterpret_cast<const float4*>(lds + rl * PITCH + c8i * 32), a1 = *reinterpret_cast<const float4*>(lds + rl * PITCH + c8i * 32 + 16);
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
                    if (e.out1) {
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
    if (a.kprof && tid == 0) atomicMax(a.kprof + 1, (unsigned long long)wall_clock64());
    if (a.kclk && id == 0 && tid == 0) { a.kclk[0] = __builtin_amdgcn_s_memtime() - a.kclk[0]; a.kclk[1] = (unsigned long long)wall_clock64() - a.kclk[1]; }
}

template <int EPI, int ABL = 0, int SCHED = 0>
__global__ __launch_bounds__(512, 2) void wn_gemm8p_kernel(const GemmArgs a) {
    __shared__ __attribute__((aligned(1024))) char lds[p8::LDS_BYTES];
    wn_gemm8p_body<EPI, ABL, SCHED>(a, lds);
}

// Does this launch fit the 8-phase kernel?  (K-interleaved taps in 64-channel blocks + at most sequential tail segments with shift 0;
// 256-row M blocks; 32-bit byte offsets inside the staged tensors.)
static inline bool wn_gemm8p_fits(const GemmArgs& a, int M, int64_t rows_total) {
    if (M % 256 != 0 || a.e.M_valid != M || !a.zero || a.taps != 3 || a.nrep != 1 || a.nseg < 3 || a.nseg > 4) return false;
    if (a.seg[0].nk % 64 != 0 || a.seg[0].nk != a.seg[1].nk || a.seg[0].nk != a.seg[2].nk) return false;
    for (int s = 0; s < a.nseg; ++s) {
        if (a.seg[s].dropout) return false;
        if (s >= 3 && (a.seg[s].shift != 0 || a.seg[s].nk % 16 != 0)) return false;
        if (s >= 3 && s + 1 < a.nseg && a.seg[s].nk % 64 != 0) return false;
        if (rows_total * a.seg[s].ld * 2 >= (int64_t)1 << 31) return false;
    }
    return true;
}
template <int EPI>
static inline int wn_launch_gemm8p(wn_ctx* ctx, GemmArgs& a, int M, hipStream_t st) {
    if (a.kil != 64 || !wn_gemm8p_fits(a, M, (int64_t)(a.b0 + a.B) * a.T))
        WN_FAIL(ctx, WN_E_STATE, "wn_launch_gemm8p: launch does not fit the 8-phase kernel (M = %d, kil = %d, taps = %d, nseg = %d)", M, a.kil, a.taps, a.nseg);
    if (ctx->trace_state == 1 && ctx->trace_n < WN_TRACE_MAX) {      // WN_DEVTRACE: this launch's own stamp slot (as wn_launch_gemm)
        a.kprof = ctx->trace_dev + 2 * ctx->trace_n;
        ctx->trace_tag[ctx->trace_n].epi = EPI; ctx->trace_tag[ctx->trace_n].st = (void*)st; ctx->trace_tag[ctx->trace_n].rows = a.B * a.T; ++ctx->trace_n;
    }
    a.mblocks = M / 256;
    a.tiles_per_utt = cdiv(a.T, 256);
    a.ntiles = a.tiles_per_utt * a.B;
    a.xcd_span = cdiv(a.ntiles, 8);
    a.stagger = 0;
    const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
    hipLaunchKernelGGL((wn_gemm8p_kernel<EPI>), dim3(grid), dim3(512), 0, st, a);
    WN_LAUNCH_CHECK(ctx);
    return WN_OK;
}
