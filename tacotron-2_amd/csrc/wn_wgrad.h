// Weight-gradient kernels of the WaveNet stack (contraction over TIME) for gfx950.
#pragma once
#include "wn_tile.h"

// ================================================================================================
// Weight-gradient kernel:  dW[m][n] (+)= scale * sum_t A[t][m] * Bm[t][n]     (contraction over TIME)
//   A[t][m]  : concatenation of source segments (dilated taps of the layer input with the dropout mask
//              re-generated, conditioning, gate output ...) plus an optional all-ones column whose row of
//              dW is the bias gradient;
//   Bm[t][n] : dz / d_skip / d_out ... [rows][ldb] bf16.
// Both operands have the contraction index as their ROW index, so the MFMA fragments (8 consecutive k per
// lane) are column gathers from the [t][c] LDS tiles (ds_read_u16, bank-conflict free with the 272-B
// row pitch).  Output tile 128x128 per workgroup, time split into slabs, fp32 atomics into the flat
// gradient buffer (lanes 0..31 hit 32 consecutive floats).
struct WgArgs {
    int32_t nseg; SrcSeg seg[4];
    int32_t ones_row;
    int32_t Mrows;               // sum of nk (excluding the ones row)
    const bf16_t* Bm; int32_t ldb, colb0, N;
    float* out; int32_t ldw;
    float* bias_out; float* bias_out2;
    float scale;
    int32_t B, T, slab, slabs_per_utt;
    uint32_t key_lo, key_hi, thresh16; float keep_scale; int32_t drop_ld;
};

#define WG_STRIDE 136   // halfs per LDS row: 128 columns + 8 pad (272 B)
#define WG_KT 32        // time steps per chunk

__global__ __launch_bounds__(256) void wn_wgrad_kernel(const WgArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t As[2][WG_KT * WG_STRIDE];
    __shared__ __attribute__((aligned(16))) bf16_t Bs[2][WG_KT * WG_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int mblk = blockIdx.x, nblk = blockIdx.y;
    const int b = blockIdx.z / a.slabs_per_utt, sl = blockIdx.z % a.slabs_per_utt;
    const int T = a.T;
    const int ts0 = sl * a.slab, ts1 = min(T, ts0 + a.slab);
    const int64_t rowbase = (int64_t)b * T;

    // ---- per-thread staging assignment: column group c16 (8 columns) is fixed, rows r0 and r0+16
    const int c16 = tid & 15, r0 = tid >> 4;
    // A column group -> (segment, channel)
    const int mcol = mblk * 128 + c16 * 8;
    int a_kind = 2;                      // 0 data, 1 ones column, 2 zero
    const bf16_t* a_base = nullptr; int a_ld = 0, a_shift = 0, a_drop = 0, a_col = 0;
    {
        int m0 = 0;
        for (int s = 0; s < a.nseg; ++s) {
            if (mcol >= m0 && mcol < m0 + a.seg[s].nk) {
                a_kind = 0; a_base = a.seg[s].base; a_ld = a.seg[s].ld; a_shift = a.seg[s].shift; a_drop = a.seg[s].dropout;
                a_col = a.seg[s].col0 + (mcol - m0);
            }
            m0 += a.seg[s].nk;
        }
        if (a.ones_row && mcol == a.Mrows) a_kind = 1;
    }
    const int ncol = nblk * 128 + c16 * 8;
    const bool b_ok = ncol < a.N;

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    uint4 sa[2], sb[2];
    auto stage_load = [&](int tc) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int t = tc + r0 + 16 * p;
            uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
            if (t < ts1) {
                if (a_kind == 0) {
                    const int ts = t + a_shift;
                    if (ts >= 0 && ts < T) {
                        const int64_t r = rowbase + ts;
                        va = *reinterpret_cast<const uint4*>(a_base + r * a_ld + a_col);
                        if (a_drop) va = drop8(va, a.key_lo, a.key_hi, a.thresh16, a.keep_scale, (uint32_t)(r * a.drop_ld + a_col));
                    }
                } else if (a_kind == 1) va.x = 0x3f80u;      // bf16 1.0 in column 0 of the group
                if (b_ok) vb = *reinterpret_cast<const uint4*>(a.Bm + (rowbase + t) * a.ldb + a.colb0 + ncol);
            }
            sa[p] = va; sb[p] = vb;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = r0 + 16 * p;
            *reinterpret_cast<uint4*>(&As[buf][row * WG_STRIDE + c16 * 8]) = sa[p];
            *reinterpret_cast<uint4*>(&Bs[buf][row * WG_STRIDE + c16 * 8]) = sb[p];
        }
    };

    const int nchunks = (ts1 - ts0 + WG_KT - 1) / WG_KT;
    if (nchunks <= 0) return;
    stage_load(ts0);
    stage_store(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < nchunks;
        if (more) stage_load(ts0 + (ch + 1) * WG_KT);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int trow = ks * 16 + (lane >> 5) * 8;
            bf16x8_t af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int col = wm * 64 + i * 32 + (lane & 31);
                unsigned short v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = As[buf][(trow + j) * WG_STRIDE + col];
                uint4 pk = make_uint4(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16), v[4] | ((uint32_t)v[5] << 16), v[6] | ((uint32_t)v[7] << 16));
                af[i] = __builtin_bit_cast(bf16x8_t, pk);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int col = wn * 64 + i * 32 + (lane & 31);
                unsigned short v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = Bs[buf][(trow + j) * WG_STRIDE + col];
                uint4 pk = make_uint4(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16), v[4] | ((uint32_t)v[5] << 16), v[6] | ((uint32_t)v[7] << 16));
                bfr[i] = __builtin_bit_cast(bf16x8_t, pk);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (more) stage_store(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: acc[i][j][r] -> m = mblk*128 + wm*64 + i*32 + 8*(r>>2) + 4*(lane>>5) + (r&3); n = nblk*128 + wn*64 + j*32 + (lane&31)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nblk * 128 + wn * 64 + j * 32 + (lane & 31);
        if (n >= a.N) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mblk * 128 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                const float v = acc[i][j][r] * a.scale;
                if (m < a.Mrows) unsafeAtomicAdd(&a.out[(int64_t)m * a.ldw + n], v);
                else if (a.ones_row && m == a.Mrows) {
                    if (a.bias_out) unsafeAtomicAdd(&a.bias_out[n], v);
                    if (a.bias_out2) unsafeAtomicAdd(&a.bias_out2[n], v);
                }
            }
        }
    }
}

// ================================================================================================
// v2: grouped (all layers in one launch), LDS-DMA ring + hardware transposing LDS reads, two-stage reduction.
//   * one launch computes the same-shaped weight gradient of up to WN_MAX_GROUPS layers ("groups"): the
//     operands of every layer are still in HBM after the backward sweep (288 GB: nothing is recycled), so
//     the time contraction of one (layer, utterance[, slab]) "unit" is long (hundreds of 32-row chunks) and
//     the split-K partials are few;
//   * output tile 128 A-columns x 256 B-columns per 8-wave workgroup (waves 2 x 4, 64 x 64 each), time
//     chunks of 32 rows, NBUF-deep ring (24 KiB per stage => 2 workgroups per CU);
//   * both operands are [time][channel] in HBM, i.e. the contraction index is the ROW: the tiles are DMA'd
//     row-major (global_load_lds_dwordx4, 16-B slots XOR-swizzled by (row & 3) << 2 on the source side) and
//     the MFMA fragments (8 consecutive time steps per lane) are read with ds_read_b64_tr_b16, two per
//     fragment: a 16-lane group reads a [4 time][16 channel] block and lane i receives channel i's column.
//     The swizzle puts the 4 rows of a block in the 4 different 64-B quarters of the 256-B bank line;
//   * bias gradients (column sums of the B operand) are accumulated on the VALU from the staged B tile by
//     the workgroups of the first A tile -- no "ones" row, so W_skip / W_out need exactly 2 A tiles;
//   * the workgroups of one unit get ids congruent mod 8 (same XCD): its operand rows are fetched from HBM
//     once and shared through that XCD's L2;
//   * no atomics (512 workgroups adding into the same 128-KiB tile serialise in L2: measured 3-40x slower):
//     every workgroup stores its fp32 tile to a partial buffer and wn_wgrad_reduce_kernel sums the units.
// LDS-DMA as inline asm (see lds_dma16 in wn_tile.h: keeps hipcc's waitcnt pass from turning every LDS-read wait into
// lgkmcnt(0) / vmcnt(0) while DMAs are in flight; ordering is enforced by the counted vmcnt + barrier of the ring).
__device__ __forceinline__ void wg_lds_dma16(const void* gsrc, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_addr) : "memory", "m0");
}
#define WN_MAX_GROUPS 32
struct WgGroup { int64_t out_off, bias_off, bias2_off; int32_t shift[4]; float scale; int32_t has_bias2;
                 int64_t out_off_hi, bias_off_hi; float scale_hi; int32_t pad_; };   // targets of the columns >= split_n (fused launches)
struct WgBatchArgs {
    int32_t ngroups, nseg;
    const bf16_t* seg_base[4]; int64_t seg_gstride[4]; int32_t seg_ld[4], seg_nk[4];
    const bf16_t* Bm; int64_t b_gstride; int32_t ldb, N;
    // fused launch (same A operand, two B operands side by side): columns [split_n, N) come from Bm_hi and go to *_hi targets
    const bf16_t* Bm_hi; int64_t b_gstride_hi; int32_t ldb_hi, split_n, ldw_hi, pad0_;
    float* grads; int32_t ldw;
    // multi-A launches (NA > 1): ONE workgroup multiplies `na` A tiles (the three dilated taps of the layer input, or the two
    // channel halves of u_l) with the SAME staged B tile: B is read na x less often and the DMA issue per MFMA drops accordingly.
    // A tile a of column block h (< hblocks): segment a_seg[a], channels [a_col0[a] + a_colstep h, +128), output rows a_mrow[a] + a_colstep h.
    int32_t na, hblocks, a_colstep, a_seg[3], a_col0[3], a_mrow[3];
    // head launches (small contractions that run beside the backward chain): spu_cap > 0 bounds the time slabs per utterance (every
    // slab costs a partial tile); transpose_out: the result is written as out[n][m] (row pitch ldw, only rows m < m_valid) -- the
    // operand roles are swapped when the natural B operand is narrower than the 256-column tile
    int32_t spu_cap, transpose_out, m_valid, pad1_;
    float* partial;                     // [unit][mtiles*128 + 8][N] fp32; row mtiles*128 = bias partial
    int32_t B, T, slab, spu, Mrows, mtiles, ntiles, nunits;
    const bf16_t* zero;
    unsigned long long* kprof;          // WN_DEVTRACE: {first workgroup's start, last workgroup's end} of this launch (null: off)
    WgGroup g[WN_MAX_GROUPS];
};
#define WG2_KT 32
#define WG2_AB (WG2_KT * 256)     // A stage bytes: 32 rows x 128 bf16
#define WG2_BB (WG2_KT * 512)     // B stage bytes: 32 rows x 256 bf16
template <int NBUF, int NA = 1>
__global__ __launch_bounds__(512, (NA == 1 ? 4 : 2)) void wn_wgrad_lds_kernel(const WgBatchArgs a) {
    constexpr int BUFB = NA * WG2_AB + WG2_BB;
    constexpr int LPC = NA + 2;                  // DMAs per wave per chunk: NA (A tiles) + 2 (B)
    static_assert(NBUF * BUFB <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char lds[NBUF * BUFB];
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 2, wn = wave & 3;
    // XCD-aware decode
    const int mt_u = (NA == 1) ? a.mtiles : a.hblocks;            // A-tile positions per unit
    const int tpu = mt_u * a.ntiles;
    const int id = blockIdx.x, xcd = id & 7, q = id >> 3;
    const int tile = q % tpu, u = (q / tpu) * 8 + xcd;
    if (u >= a.nunits) return;
    if (a.kprof && tid == 0) atomicMin(a.kprof, (unsigned long long)wall_clock64());
    const int mblk = tile % mt_u, nblk = tile / mt_u;
    const int upg = a.B * a.spu;
    const int grp = u / upg, b = (u - grp * upg) / a.spu, sl = (u - grp * upg) % a.spu;
    const int T = a.T;
    const int ts0 = sl * a.slab, ts1 = min(T, ts0 + a.slab);
    const int nchunks = max(0, (ts1 - ts0 + WG2_KT - 1) / WG2_KT);
    const int64_t rowbase = (int64_t)b * T;

    // A tile(s) -> (segment, first channel, valid channels, first output row); segments are multiples of 8 channels
    const bf16_t* a_base[NA]; int a_ld[NA], a_shift[NA], a_valid[NA], a_mrow[NA];
    if constexpr (NA == 1) {
        a_base[0] = a.zero; a_ld[0] = 0; a_shift[0] = 0; a_valid[0] = 0; a_mrow[0] = mblk * 128;
        int m0 = 0; const int mcol = mblk * 128;
        for (int s = 0; s < a.nseg; ++s) {
            if (mcol >= m0 && mcol < m0 + a.seg_nk[s]) {
                a_base[0] = a.seg_base[s] + (int64_t)grp * a.seg_gstride[s] + (mcol - m0); a_ld[0] = a.seg_ld[s]; a_shift[0] = a.g[grp].shift[s];
                a_valid[0] = min(128, a.seg_nk[s] - (mcol - m0));
            }
            m0 += a.seg_nk[s];
        }
    } else {
#pragma unroll
        for (int x = 0; x < NA; ++x) {
            const int sg = a.a_seg[x], col = a.a_col0[x] + mblk * a.a_colstep;
            a_base[x] = a.seg_base[sg] + (int64_t)grp * a.seg_gstride[sg] + col; a_ld[x] = a.seg_ld[sg]; a_shift[x] = a.g[grp].shift[sg];
            a_valid[x] = max(0, min(128, a.seg_nk[sg] - col)); a_mrow[x] = a.a_mrow[x] + mblk * a.a_colstep;
        }
    }
    const int n0 = nblk * 256;
    // B columns [0, split_n) come from Bm, [split_n, N) from Bm_hi (fused launches); the split may fall INSIDE a 256-column
    // tile (narrow models: S = R = 128), so the operand is chosen per 8-column slot, i.e. per lane of the DMA
    const bf16_t* const b_lo = a.Bm + (int64_t)grp * a.b_gstride;
    const bf16_t* const b_hi = a.split_n > 0 ? a.Bm_hi + (int64_t)grp * a.b_gstride_hi : nullptr;

    f32x16_t acc[NA][2][2];
#pragma unroll
    for (int x = 0; x < NA; ++x)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][i][j][r] = 0.0f;
    // bias gradients = column sums of the B tile.  The three-tap workgroups (192 accumulator registers) leave them to the
    // conditioning-kernel launch of the same layers, which reads the same d z (wgrad_cin_args).
    constexpr bool BIAS = (NA != 3);
    const bool do_bias = BIAS && (mblk == 0);
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.0f;

    auto stage = [&](auto bufc, int ch) {
        constexpr int BUF = decltype(bufc)::value;
        char* const abuf = lds + BUF * BUFB;
        char* const bbuf = abuf + NA * WG2_AB;
        const int tc = ts0 + ch * WG2_KT;
#pragma unroll
        for (int x = 0; x < NA; ++x) {   // A tile x: piece `wave` = rows wave*4 .. +3, 16 slots of 16 B each
            const int row = wave * 4 + (lane >> 4);
            const int c = (lane & 15) ^ ((row & 3) << 2);
            const int t = tc + row, ts = t + a_shift[x];
            const bool ok = (c * 8 < a_valid[x]) && (t < ts1) && (ts >= 0) && (ts < T);
            const bf16_t* src = ok ? a_base[x] + (rowbase + ts) * a_ld[x] + c * 8 : a.zero;
            wg_lds_dma16(src, __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) char*)(abuf + x * WG2_AB + wave * 1024)));
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {   // B: piece g = rows g*2, g*2+1, 32 slots each
            const int g = wave + p * 8;
            const int row = g * 2 + (lane >> 5);
            const int c = (lane & 31) ^ ((row & 3) << 2);
            const int t = tc + row;
            const int col = n0 + c * 8;
            const bool ok = (t < ts1) && (col < a.N);
            const bool hi = a.split_n > 0 && col >= a.split_n;
            const bf16_t* src = !ok ? a.zero : hi ? b_hi + (rowbase + t) * a.ldb_hi + (col - a.split_n) : b_lo + (rowbase + t) * a.ldb + col;
            wg_lds_dma16(src, __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) char*)(bbuf + g * 1024)));
        }
    };
    // ---- interior chunks of full tiles: every source address is (wave-uniform base) + (per-lane constant), no bounds to check.
    // The generic stage() above recomputes five 64-bit row addresses with quarter-rate integer multiplies and four range tests per
    // chunk (~130 instructions in front of the wave's MFMAs); this one is five scalar base computations and five SGPR-base DMAs.
    bool fast_wg = (NA > 1) && (n0 + 256 <= a.N) && !(a.split_n > 0 && n0 < a.split_n && n0 + 256 > a.split_n);      // (single-A launches: 128 VGPRs, no room)
    int sh_min = 0, sh_max = 0;
#pragma unroll
    for (int x = 0; x < NA; ++x) { fast_wg = fast_wg && a_valid[x] == 128 && a_ld[x] == a_ld[0]; sh_min = min(sh_min, a_shift[x]); sh_max = max(sh_max, a_shift[x]); }
    const bool b_is_hi = a.split_n > 0 && n0 >= a.split_n;
    const bf16_t* const fb_base = b_is_hi ? b_hi + (n0 - a.split_n) : b_lo + n0;
    const int fb_ld = b_is_hi ? a.ldb_hi : a.ldb;
    const uint32_t voffA = (uint32_t)(((lane >> 4) * a_ld[0] + (((lane & 15) ^ (((lane >> 4) & 3) << 2)) * 8)) * 2);
    const uint32_t voffB = (uint32_t)(((lane >> 5) * fb_ld + (((lane & 31) ^ (((wave * 2 + (lane >> 5)) & 3) << 2)) * 8)) * 2);
    auto sgpr64 = [](uint64_t v) { return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v); };
    auto fast_stage = [&](auto bufc, int ch) {
        constexpr int BUF = decltype(bufc)::value;
        char* const abuf = lds + BUF * BUFB;
        char* const bbuf = abuf + NA * WG2_AB;
        const int64_t r0 = rowbase + ts0 + ch * WG2_KT;
        const uint32_t va = voffA, vb = voffB;
#pragma unroll
        for (int x = 0; x < NA; ++x) {
            const uint64_t sb = sgpr64((uint64_t)(a_base[x] + (r0 + a_shift[x] + wave * 4) * a_ld[0]));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(va), "s"(sb),
                         "s"(__builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) char*)(abuf + x * WG2_AB + wave * 1024))) : "memory", "m0");
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int g = wave + p * 8;
            const uint64_t sb = sgpr64((uint64_t)(fb_base + (r0 + g * 2) * fb_ld));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vb), "s"(sb),
                         "s"(__builtin_amdgcn_readfirstlane((uint32_t)(size_t)(const __attribute__((address_space(3))) char*)(bbuf + g * 1024))) : "memory", "m0");
        }
    };
    auto stage_any = [&](auto bufc, int ch) {
        if constexpr (NA > 1) {
            const int tc = ts0 + ch * WG2_KT;
            if (fast_wg && tc + sh_min >= 0 && tc + WG2_KT + sh_max <= T && tc + WG2_KT <= ts1) { fast_stage(bufc, ch); return; }
        }
        stage(bufc, ch);
    };
    // per-lane constants of the transposing reads: lane -> (row within a 4-row block, 8-B piece within the 16 channels)
    const int tr_row = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int tr_colb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;       // byte offset inside a 32-channel fragment
    auto compute = [&](auto bufc) {
        constexpr int BUF = decltype(bufc)::value;
        const char* const abuf = lds + BUF * BUFB;
        const char* const bbuf = abuf + NA * WG2_AB;
#pragma unroll
        for (int ks = 0; ks < WG2_KT / 16; ++ks) {
            bf16x8_t af[NA][2], bfr[2];
#pragma unroll
            for (int x = 0; x < NA; ++x)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                s16x4 h[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int row = ks * 16 + tr_row + 4 * jj;
                    const int cb = ((wk * 2 + i) * 64 + tr_colb) ^ ((row & 3) << 6);
                    h[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(abuf + x * WG2_AB + row * 256 + cb));
                }
                struct { s16x4 lo, hi; } pk = {h[0], h[1]};
                af[x][i] = __builtin_bit_cast(bf16x8_t, pk);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                s16x4 h[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int row = ks * 16 + tr_row + 4 * jj;
                    const int cb = ((wn * 2 + j) * 64 + tr_colb) ^ ((row & 3) << 6);
                    h[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(bbuf + row * 512 + cb));
                }
                struct { s16x4 lo, hi; } pk = {h[0], h[1]};
                bfr[j] = __builtin_bit_cast(bf16x8_t, pk);
            }
#pragma unroll
            for (int x = 0; x < NA; ++x)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[x][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[x][i], bfr[j], acc[x][i][j], 0, 0, 0);
        }
        if constexpr (BIAS) if (do_bias) {   // column sums of the B tile: thread -> 8 columns x 2 rows
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = (tid >> 5) * 2 + rr;
                const int c = (tid & 31) ^ ((row & 3) << 2);
                const uint4 x = *reinterpret_cast<const uint4*>(bbuf + row * 512 + c * 16);
                bsum[0] += bf2f((bf16_t)(x.x & 0xffff)); bsum[1] += bf2f((bf16_t)(x.x >> 16));
                bsum[2] += bf2f((bf16_t)(x.y & 0xffff)); bsum[3] += bf2f((bf16_t)(x.y >> 16));
                bsum[4] += bf2f((bf16_t)(x.z & 0xffff)); bsum[5] += bf2f((bf16_t)(x.z >> 16));
                bsum[6] += bf2f((bf16_t)(x.w & 0xffff)); bsum[7] += bf2f((bf16_t)(x.w >> 16));
            }
        }
    };
    auto ring_step = [&](auto bufc, int ch) {
        constexpr int BUF = decltype(bufc)::value;
        const int younger = min(NBUF - 2, nchunks - 1 - ch);
        if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPC) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (ch + NBUF - 1 < nchunks) stage_any(std::integral_constant<int, (BUF + NBUF - 1) % NBUF>{}, ch + NBUF - 1);
        compute(bufc);
    };
    static_assert(NBUF == 2 || NBUF == 3, "ring depth");
    if (nchunks > 0) stage_any(std::integral_constant<int, 0>{}, 0);
    if constexpr (NBUF == 3) { if (nchunks > 1) stage_any(std::integral_constant<int, 1>{}, 1); }
    for (int ch = 0; ch < nchunks; ch += NBUF) {
        ring_step(std::integral_constant<int, 0>{}, ch);
        if (ch + 1 < nchunks) ring_step(std::integral_constant<int, 1>{}, ch + 1);
        if constexpr (NBUF == 3) { if (ch + 2 < nchunks) ring_step(std::integral_constant<int, 2>{}, ch + 2); }
    }

#ifdef WN_EPI_ABLATE
    if (a.ldw != -7777) {
#pragma unroll
        for (int x = 0; x < NA; ++x)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[x][i][j][r]));
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(bsum[e]));
        return;
    }
#endif
    // ---- epilogue: the fp32 tile of this unit goes to its partial slot (every element written: no zero-fill needed).
    // acc[i][j][r] -> m = mblk*128 + (wk*2+i)*32 + 8*(r>>2) + 4*(lane>>5) + (r&3); n = n0 + (wn*2+j)*32 + (lane&31)
    const int rows_p = a.mtiles * 128 + 8;
    float* const P = a.partial + (int64_t)u * rows_p * a.N;
#pragma unroll
    for (int x = 0; x < NA; ++x)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + (wn * 2 + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = a_mrow[x] + (wk * 2 + i) * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                if (n < a.N) P[(int64_t)m * a.N + n] = acc[x][i][j][r];          // (N % 256 != 0: the last column tile is partial)
            }
        }
    }
    if constexpr (BIAS) if (do_bias) {      // 16 row-groups hold partial column sums of the same 8 columns: combine through LDS
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);          // [16][256]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 5) * 256 + (tid & 31) * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < 256) {
            float sum = 0.0f;
#pragma unroll
            for (int g = 0; g < 16; ++g) sum += red[g * 256 + tid];
            if (n0 + tid < a.N) P[(int64_t)(a.mtiles * 128) * a.N + n0 + tid] = sum;
        }
    }
    if (a.kprof && tid == 0) atomicMax(a.kprof + 1, (unsigned long long)wall_clock64());
}

// out[m][n] += scale_g * sum_{units of group g} partial[unit][m][n];  bias rows likewise.  One float4 per thread.
__global__ __launch_bounds__(256) void wn_wgrad_reduce_kernel(const WgBatchArgs a) {
    const int rows_p = a.mtiles * 128 + 8;
    const int n4 = a.N >> 2;
    const int64_t per_group = (int64_t)(a.Mrows + 1) * n4;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_group * a.ngroups) return;
    const int grp = (int)(idx / per_group);
    const int64_t rem = idx - (int64_t)grp * per_group;
    const int m = (int)(rem / n4), c4 = (int)(rem % n4);
    const bool is_bias = (m == a.Mrows);
    const int prow = is_bias ? a.mtiles * 128 : m;
    const int upg = a.B * a.spu;
    const float* p = a.partial + ((int64_t)grp * upg * rows_p + prow) * a.N + c4 * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ustride = (int64_t)rows_p * a.N;
    int k = 0;
    for (; k + 4 <= upg; k += 4) {      // four partial tiles in flight per thread (the sum stays in unit order)
        const float4 v0 = *reinterpret_cast<const float4*>(p + (int64_t)k * ustride), v1 = *reinterpret_cast<const float4*>(p + (int64_t)(k + 1) * ustride);
        const float4 v2 = *reinterpret_cast<const float4*>(p + (int64_t)(k + 2) * ustride), v3 = *reinterpret_cast<const float4*>(p + (int64_t)(k + 3) * ustride);
        s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
        s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
        s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
        s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
    }
    for (; k < upg; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(p + (int64_t)k * ustride);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const WgGroup& g = a.g[grp];
    const bool hi = a.split_n > 0 && c4 * 4 >= a.split_n;
    const float sc = hi ? g.scale_hi : g.scale;
    const int col = hi ? c4 * 4 - a.split_n : c4 * 4;
    s.x *= sc; s.y *= sc; s.z *= sc; s.w *= sc;
    if (a.transpose_out) {
        if (!is_bias && m < a.m_valid) {
            float* dst = a.grads + g.out_off + (int64_t)col * a.ldw + m;
            dst[0] += s.x; dst[a.ldw] += s.y; dst[2 * (int64_t)a.ldw] += s.z; dst[3 * (int64_t)a.ldw] += s.w;
        }
        return;
    }
    auto add4 = [&](float* dst) { float4 o = *reinterpret_cast<float4*>(dst); o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w; *reinterpret_cast<float4*>(dst) = o; };
    if (!is_bias) add4(a.grads + (hi ? g.out_off_hi : g.out_off) + (int64_t)m * (hi ? a.ldw_hi : a.ldw) + col);
    else if (hi) {
        if (g.bias_off_hi >= 0) add4(a.grads + g.bias_off_hi + col);
    } else {
        if (g.bias_off >= 0) add4(a.grads + g.bias_off + col);
        if (g.has_bias2) add4(a.grads + g.bias2_off + col);
    }
}

// plan the unit split: enough workgroups for >= ~2 rounds of 2 x 256, slabs a multiple of the 32-row chunk
static inline void wn_wgrad_plan(WgBatchArgs& a) {
    a.Mrows = 0; for (int s = 0; s < a.nseg; ++s) a.Mrows += a.seg_nk[s];
    a.mtiles = cdiv(a.Mrows, 128); a.ntiles = cdiv(a.N, 256);
    if (a.na <= 1) { a.na = 1; a.hblocks = a.mtiles; }
    const int tpu = a.hblocks * a.ntiles;
    // time slabs per utterance: multi-A workgroups are alone on their CU (one round = 256 workgroups) and every extra slab costs a
    // full fp32 output tile in the partial buffer, so launches are split only until one round is full (256 multi-A workgroups, or
    // 2 x 256 single-A ones)
    int spu = cdiv(a.na > 1 ? 256 : 512, (int64_t)tpu * a.B * a.ngroups);
    const int max_spu = a.T / 256 > 0 ? a.T / 256 : 1;
    if (spu > max_spu) spu = max_spu;
    if (spu < 1) spu = 1;
    if (a.na > 1 && a.spu_cap == 0) {
        // multi-A workgroups own their CU: a launch of W workgroups takes ceil(W / 256) rounds, and a last round that is mostly empty is
        // paid in full (d [W_skip | W_out] at C2: 2 x 24 x 8 = 384 workgroups = 1.5 rounds, 25 % of the launch idle).  One or two more slabs
        // per utterance (each costs a partial tile) when that brings the idle share of the rounds under 10 %.
        auto idle = [&](int sp) { const int64_t w = (int64_t)tpu * a.B * a.ngroups * sp; const int64_t r = (w + 255) / 256; return (double)(r * 256 - w) / (double)(r * 256); };
        int best = spu;
        for (int sp = spu; sp <= spu + 2 && sp <= max_spu; ++sp)
            if (idle(sp) < idle(best) - 0.05) best = sp;
        if (idle(spu) > 0.10) spu = best;
    }
    if (a.spu_cap > 0 && spu > a.spu_cap) spu = a.spu_cap;
    a.slab = cdiv(cdiv(a.T, spu), WG2_KT) * WG2_KT;
    a.spu = cdiv(a.T, a.slab);
    a.nunits = a.ngroups * a.B * a.spu;
}
static inline size_t wn_wgrad_partial_bytes(const WgBatchArgs& a) { return (size_t)a.nunits * (a.mtiles * 128 + 8) * a.N * 4; }
static inline bool wn_wgrad_v2_ok(const WgBatchArgs& a) {
    // N: whole 8-column DMA slots and float4 reduce items (the last 256-column tile may be partial: its missing columns stage the zero
    // page and are not written); with a split B operand both halves likewise
    if (a.N % 8 != 0 || (a.split_n > 0 && a.split_n % 8 != 0) || (a.ldw % 4 != 0 && !a.transpose_out) || (a.split_n > 0 && a.ldw_hi % 4 != 0) || a.ngroups > WN_MAX_GROUPS) return false;
    for (int s = 0; s < a.nseg; ++s) if (a.seg_nk[s] % 8 != 0 || (a.seg_nk[s] % 128 != 0 && s != a.nseg - 1)) return false;
    return true;
}
static int launch_wgrad_batch(wn_ctx* c, WgBatchArgs& a, hipStream_t st) {
    wn_wgrad_plan(a);
    if (wn_wgrad_partial_bytes(a) > c->wg_partial_bytes) WN_FAIL(c, WN_E_STATE, "wgrad partial buffer too small (%zu > %zu)", wn_wgrad_partial_bytes(a), c->wg_partial_bytes);
    a.partial = c->wg_partial; a.zero = c->zero_page; a.kprof = nullptr;
    if (c->trace_state == 1 && c->trace_n < WN_TRACE_MAX) {      // WN_DEVTRACE: tag 100 + na
        a.kprof = c->trace_dev + 2 * c->trace_n;
        c->trace_tag[c->trace_n].epi = 100 + a.na; c->trace_tag[c->trace_n].st = (void*)st; c->trace_tag[c->trace_n].rows = a.ngroups; ++c->trace_n;
    }
    const int grid = cdiv(a.nunits, 8) * a.hblocks * a.ntiles * 8;
    if (a.na == 3) hipLaunchKernelGGL((wn_wgrad_lds_kernel<3, 3>), dim3(grid), dim3(512), 0, st, a);
    else if (a.na == 2) hipLaunchKernelGGL((wn_wgrad_lds_kernel<3, 2>), dim3(grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((wn_wgrad_lds_kernel<3, 1>), dim3(grid), dim3(512), 0, st, a);
    WN_LAUNCH_CHECK(c);
    const int64_t items = (int64_t)(a.Mrows + 1) * (a.N / 4) * a.ngroups;
    hipLaunchKernelGGL(wn_wgrad_reduce_kernel, dim3(cdiv(items, 256)), dim3(256), 0, st, a);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}

static int launch_wgrad(wn_ctx* c, WgArgs& a, hipStream_t st) {
    a.Mrows = 0;
    for (int s = 0; s < a.nseg; ++s) a.Mrows += a.seg[s].nk;
    const int mtot = a.Mrows + (a.ones_row ? 1 : 0);
    a.slab = 4096;
    a.slabs_per_utt = cdiv(a.T, a.slab);
    dim3 grid(cdiv(mtot, 128), cdiv(a.N, 128), a.B * a.slabs_per_utt);
    hipLaunchKernelGGL(wn_wgrad_kernel, grid, dim3(256), 0, st, a);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
