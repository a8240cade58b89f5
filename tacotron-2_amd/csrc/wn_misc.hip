// Element-wise / reduction kernels of the WaveNet hot path (everything that is not an MFMA contraction):
// weight packing, input convolution, upsample net, losses (+ their gradients), optimiser, mu-law codec,
// samplers.  All HBM-bound: coalesced along the contiguous axis, 64-wide waves, grid-stride where large.
#include "wn_common.h"
#include "wn_mulaw_tables.h"
#include <math.h>
#include <algorithm>

// =================================================================================== weight packing
// Fragment order of v_mfma_f32_32x32x16_bf16's A operand:  out[((mtile*KS + ks)*64 + lane)*8 + j]
//   <-> W[m = mtile*32 + (lane&31)][k = ks*16 + 8*(lane>>5) + j]
struct PackJob { bf16_t* out; const PackSeg* segs; int32_t M, K, M_valid, gate_il, GH, nseg; int32_t block0, pad; };
// One launch packs every matrix: block -> job by binary search in the jobs' first-block table.
__global__ void wn_pack_kernel(const float* __restrict__ params, const PackJob* __restrict__ jobs, int njobs) {
    // one thread = one lane's 8-element fragment piece (8 consecutive k of one row): 8 loads that are coalesced across the
    // 32 lanes of a row group wherever the source is row-contiguous, one 16-B store.
    int lo = 0, hi = njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const PackJob jb = jobs[lo];
    const int M = jb.M, K = jb.K;
    const int64_t idx8 = (int64_t)(blockIdx.x - jb.block0) * blockDim.x + threadIdx.x;
    if (idx8 * 8 >= (int64_t)M * K) return;
    const int KS = K >> 4;
    const int lane = (int)(idx8 & 63);
    const int64_t rest = idx8 >> 6;
    const int ks = (int)(rest % KS), mtile = (int)(rest / KS);
    const int m = mtile * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
    int mm = m;
    if (jb.gate_il) {       // rows come in 64-row groups [32 tanh rows | their 32 sigmoid partners] (modules.py:494,510)
        const int blk = m >> 6, w = m & 63;
        mm = (w < 32) ? blk * 32 + w : jb.GH + blk * 32 + (w - 32);
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    if (mm < jb.M_valid) {
        // segments are sorted by k0 and they start at multiples of 8 (channel counts are multiples of 16): the 8 elements of this
        // thread lie in ONE segment, found by binary search (a K-interleaved pack has 3 * R/32 + 1 of them)
        int lo2 = 0, hi2 = jb.nseg - 1;
        while (lo2 < hi2) { const int mid = (lo2 + hi2 + 1) >> 1; if (jb.segs[mid].k0 <= k0) lo2 = mid; else hi2 = mid - 1; }
        const PackSeg sg = jb.segs[lo2];
        if (k0 >= sg.k0 && k0 < sg.k0 + sg.nk) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j < sg.k0 + sg.nk)          // (the last segment may end inside the group: out_channels = 30 -> K padded to 32)
                    v[j] = sg.scale * params[sg.base + (int64_t)(k0 + j - sg.k0) * sg.stride_k + (int64_t)mm * sg.stride_m];
        }
    }
    *reinterpret_cast<uint4*>(jb.out + idx8 * 8) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}

struct VecSum { int n; int64_t off[32]; float w[32]; };
__global__ void wn_vecsum_kernel(const float* __restrict__ params, float* __restrict__ out, int len, VecSum vs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    float a = 0.0f;
    for (int j = 0; j < vs.n; ++j) a += vs.w[j] * params[vs.off[j] + i];
    out[i] = a;
}
// out[y][i] = params[a.off[y] + i] + params[b.off[y] + i] for every layer y in one launch (dilated-conv bias + conditioning bias)
struct PairSum { int64_t a[32], b[32]; };
__global__ void wn_pairsum_kernel(const float* __restrict__ params, float* __restrict__ out, int len, PairSum ps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (i < len) out[(size_t)y * len + i] = params[ps.a[y] + i] + params[ps.b[y] + i];
}

static void init_pack(wn_ctx* c, PackedW& w, int M_src, int K_src, int gate_il, int m_align = 32) {
    w.M = (M_src + m_align - 1) / m_align * m_align; w.K = (K_src + 15) / 16 * 16; w.M_valid = M_src; w.gate_interleave = gate_il; w.GH = c->GH;
}

static int finish_pack(wn_ctx* c, PackedW& w) {
    WN_HIP(c, hipMalloc((void**)&w.dev, (size_t)w.M * w.K * 2));
    WN_HIP(c, hipMalloc((void**)&w.dev_segs, w.segs.size() * sizeof(PackSeg)));
    WN_HIP(c, hipMemcpy(w.dev_segs, w.segs.data(), w.segs.size() * sizeof(PackSeg), hipMemcpyHostToDevice));
    return WN_OK;
}

int wn_build_packs(wn_ctx* c) {
    const int L = c->L, R = c->R, G = c->G, GH = c->GH, S = c->S, O = c->O, C = c->C;
    c->packs.resize(L);
    int rc;
    for (int l = 0; l < L; ++l) {
        const WnLayerOffsets& o = c->lay[l];
        WnLayerPacks& p = c->packs[l];
        // W1: rows = gate channels (interleaved), K = [tap0 R | tap1 R | tap2 R | cin C]; W[g][j*R+r] = dil[j][r][g]
        // Matrices that take the LDS-DMA tile engine (M % 128 == 0) interleave the taps along K in 32-channel blocks: the engine then
        // stages (tap0, tap1, tap2) of one k-block back to back, so the rows two taps have in common are re-read while still in L2.
        init_pack(c, p.w1, G, 3 * R + C, 1);
        p.w1.kil = (G % 128 == 0 && R % 32 == 0) ? 32 : 0;
        // the 8-phase kernel (wn_tile8p.h: 256-row M blocks, 64-channel K-tiles, 32-bit byte offsets inside the staged tensors)
        if ((c->gemm8p & 1) && G % 256 == 0 && R % 64 == 0 && C % 16 == 0 && c->NT * std::max(R, C) * 2 < ((int64_t)1 << 31)) p.w1.kil = 64;
        if (p.w1.kil) { const int kl = p.w1.kil; for (int kb = 0; kb < R / kl; ++kb) for (int j = 0; j < 3; ++j) p.w1.segs.push_back({o.dil_k + ((int64_t)j * R + kb * kl) * G, (kb * 3 + j) * kl, kl, G, 1, 1.0f}); }
        else for (int j = 0; j < 3; ++j) p.w1.segs.push_back({o.dil_k + (int64_t)j * R * G, j * R, R, G, 1, 1.0f});
        p.w1.segs.push_back({o.cin_k, 3 * R, C, G, 1, 1.0f});
        if ((rc = finish_pack(c, p.w1))) return rc;
        // Wo: rows = residual channels, K = GH;  W[r][g'] = out_k[g'][r]
        init_pack(c, p.wo, R, GH, 0);
        p.wo.segs.push_back({o.out_k, 0, GH, R, 1, 1.0f});
        if ((rc = finish_pack(c, p.wo))) return rc;
        // Ws (synthesis): rows = skip channels, scaled by the legacy factor
        init_pack(c, p.ws, S, GH, 0);
        p.ws.segs.push_back({o.skip_k, 0, GH, S, 1, c->skip_scale[l]});
        if ((rc = finish_pack(c, p.ws))) return rc;
        // W2T (dgate): rows = g', K = [R | S];  W[g'][r] = out_k[g'][r],  W[g'][R+s] = c_l * skip_k[g'][s]
        init_pack(c, p.w2T, GH, R + S, 0);
        p.w2T.segs.push_back({o.out_k, 0, R, 1, R, 1.0f});
        p.w2T.segs.push_back({o.skip_k, R, S, 1, S, c->skip_scale[l]});
        if ((rc = finish_pack(c, p.w2T))) return rc;
        // W1T (dx): rows = r, K = [tap0 G | tap1 G | tap2 G];  W[r][j*G+g] = dil[j][r][g]
        init_pack(c, p.w1T, R, 3 * G, 0);
        p.w1T.kil = (R % 128 == 0 && G % 32 == 0) ? 32 : 0;
        if ((c->gemm8p & 2) && R % 256 == 0 && G % 64 == 0 && c->NT * G * 2 < ((int64_t)1 << 31)) p.w1T.kil = 64;
        if (p.w1T.kil) { const int kl = p.w1T.kil; for (int kb = 0; kb < G / kl; ++kb) for (int j = 0; j < 3; ++j) p.w1T.segs.push_back({o.dil_k + (int64_t)j * R * G + kb * kl, (kb * 3 + j) * kl, kl, 1, G, 1.0f}); }
        else for (int j = 0; j < 3; ++j) p.w1T.segs.push_back({o.dil_k + (int64_t)j * R * G, j * G, G, 1, G, 1.0f});
        if ((rc = finish_pack(c, p.w1T))) return rc;
    }
    // skip sum as ONE contraction over all layers' gate outputs: rows = s, K = L*GH (wavenet.py:706-715 unrolled)
    init_pack(c, c->wskip, S, L * GH, 0);
    for (int l = 0; l < L; ++l) c->wskip.segs.push_back({c->lay[l].skip_k, l * GH, GH, S, 1, c->skip_scale[l]});
    if ((rc = finish_pack(c, c->wskip))) return rc;
    init_pack(c, c->wh1, S, S, 0); c->wh1.segs.push_back({c->fin1_k, 0, S, S, 1, 1.0f});
    if ((rc = finish_pack(c, c->wh1))) return rc;
    init_pack(c, c->wh2, O, S, 0); c->wh2.segs.push_back({c->fin2_k, 0, S, O, 1, 1.0f});
    if ((rc = finish_pack(c, c->wh2))) return rc;
    init_pack(c, c->wh2T, S, O, 0); c->wh2T.segs.push_back({c->fin2_k, 0, O, 1, O, 1.0f});
    if ((rc = finish_pack(c, c->wh2T))) return rc;
    init_pack(c, c->wh1T, S, S, 0); c->wh1T.segs.push_back({c->fin1_k, 0, S, 1, S, 1.0f});
    if ((rc = finish_pack(c, c->wh1T))) return rc;
    // d_c: rows = cin channel, K = L*G;  W[cc][l*G+g] = cin_k_l[cc][g]
    init_pack(c, c->wcT, C, L * G, 0, C <= 96 ? 96 : 128);      // M padded to the 96- or 128-row tile of the LDS-DMA main loop (80 mels -> 96)
    for (int l = 0; l < L; ++l) c->wcT.segs.push_back({c->lay[l].cin_k, l * G, G, 1, G, 1.0f});
    if ((rc = finish_pack(c, c->wcT))) return rc;

    WN_HIP(c, hipMalloc((void**)&c->b1sum, (size_t)L * G * 4));
    WN_HIP(c, hipMalloc((void**)&c->skip_bias_total, (size_t)S * 4));
    WN_HIP(c, hipMalloc((void**)&c->params_dev, (size_t)(c->n_params + c->zpad) * 4));
    WN_HIP(c, hipMemset(c->params_dev, 0, (size_t)(c->n_params + c->zpad) * 4));       // incl. the zero tail that absent biases read
    if (c->gin > 0) {
        WN_HIP(c, hipMalloc((void**)&c->gvec, (size_t)c->maxB * c->gin * 4));
        WN_HIP(c, hipMalloc((void**)&c->gids, (size_t)c->maxB * 4));
        WN_HIP(c, hipMalloc((void**)&c->gbias, (size_t)L * c->maxB * G * 4));
        WN_HIP(c, hipMalloc((void**)&c->colsum, (size_t)L * c->maxB * G * 4));
    }
    if (c->wnorm) {
        WN_HIP(c, hipMalloc((void**)&c->raw_dev, (size_t)c->n_raw * 4));
        WN_HIP(c, hipMalloc((void**)&c->deff, (size_t)c->n_params * 4));
        WN_HIP(c, hipMalloc((void**)&c->wmap_dev, c->wmap.size() * sizeof(wn_ctx::WnMap)));
        WN_HIP(c, hipMemcpy(c->wmap_dev, c->wmap.data(), c->wmap.size() * sizeof(wn_ctx::WnMap), hipMemcpyHostToDevice));
    }
    const int nt = (int)c->raw_tensors.size();          // per-VARIABLE clipping (wavenet.py:586-598): v and g are separate variables
    std::vector<int32_t> offs(nt + 1);
    for (int i = 0; i < nt; ++i) offs[i] = (int32_t)c->raw_tensors[i].offset;
    offs[nt] = (int32_t)c->n_raw;
    WN_HIP(c, hipMalloc((void**)&c->tensor_offsets_dev, (nt + 1) * 4));
    WN_HIP(c, hipMemcpy(c->tensor_offsets_dev, offs.data(), (nt + 1) * 4, hipMemcpyHostToDevice));
    WN_HIP(c, hipMalloc((void**)&c->norm2_dev, nt * 4));
    // span table of the atomic-free clip norms (wn_norm2_span_kernel): spans of <= WN_NORM_SPAN floats that never cross a tensor;
    // tensor i owns the spans [first[i], first[i + 1])
    {
        std::vector<int32_t> sp; std::vector<int32_t> first(nt + 1);
        for (int i = 0; i < nt; ++i) {
            first[i] = (int32_t)(sp.size() / 2);
            for (int64_t o = offs[i]; o < offs[i + 1]; o += 4096) { sp.push_back((int32_t)o); sp.push_back((int32_t)std::min<int64_t>(offs[i + 1], o + 4096)); }
        }
        first[nt] = (int32_t)(sp.size() / 2);
        c->norm_nspans = first[nt];
        WN_HIP(c, hipMalloc((void**)&c->norm_spans_dev, sp.size() * 4 + 8));
        WN_HIP(c, hipMemcpy(c->norm_spans_dev, sp.data(), sp.size() * 4, hipMemcpyHostToDevice));
        WN_HIP(c, hipMalloc((void**)&c->norm_first_dev, (nt + 1) * 4));
        WN_HIP(c, hipMemcpy(c->norm_first_dev, first.data(), (nt + 1) * 4, hipMemcpyHostToDevice));
        WN_HIP(c, hipMalloc((void**)&c->norm_part_dev, (size_t)c->norm_nspans * 4 + 8));
    }
    return WN_OK;
}

static void add_pack_job(wn_ctx* c, std::vector<PackJob>& jobs, int& nblocks, const PackedW& w) {
    PackJob j; j.out = w.dev; j.segs = w.dev_segs; j.M = w.M; j.K = w.K; j.M_valid = w.gate_interleave ? c->G : w.M_valid;
    j.gate_il = w.gate_interleave; j.GH = w.GH; j.nseg = (int)w.segs.size(); j.block0 = nblocks; j.pad = 0;
    nblocks += cdiv((int64_t)w.M * w.K / 8, 256);
    jobs.push_back(j);
}

int wn_launch_pack(wn_ctx* c, const float* params, hipStream_t st) {
    if (c->wnorm) { int rcw = wn_weightnorm_apply(c, params, st); if (rcw) return rcw; }
    else WN_HIP(c, hipMemcpyAsync(c->params_dev, params, (size_t)c->n_params * 4, hipMemcpyDeviceToDevice, st));
    if (!c->pack_jobs_dev) {          // job table: built once (pack buffers never move)
        std::vector<PackJob> jobs; int nblocks = 0;
        for (int l = 0; l < c->L; ++l) {
            WnLayerPacks& p = c->packs[l];
            add_pack_job(c, jobs, nblocks, p.w1); add_pack_job(c, jobs, nblocks, p.wo); add_pack_job(c, jobs, nblocks, p.ws);
            add_pack_job(c, jobs, nblocks, p.w2T); add_pack_job(c, jobs, nblocks, p.w1T);
        }
        add_pack_job(c, jobs, nblocks, c->wskip); add_pack_job(c, jobs, nblocks, c->wh1); add_pack_job(c, jobs, nblocks, c->wh2);
        add_pack_job(c, jobs, nblocks, c->wh2T); add_pack_job(c, jobs, nblocks, c->wh1T); add_pack_job(c, jobs, nblocks, c->wcT);
        WN_HIP(c, hipMalloc((void**)&c->pack_jobs_dev, jobs.size() * sizeof(PackJob)));
        WN_HIP(c, hipMemcpy(c->pack_jobs_dev, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
        c->pack_njobs = (int)jobs.size(); c->pack_nblocks = nblocks;
    }
    hipLaunchKernelGGL(wn_pack_kernel, dim3(c->pack_nblocks), dim3(256), 0, st, c->params_dev, (const PackJob*)c->pack_jobs_dev, c->pack_njobs);
    WN_LAUNCH_CHECK(c);
    if (c->L > 32) WN_FAIL(c, WN_E_UNSUPPORTED, "layers > 32");
    {
        PairSum ps;
        for (int l = 0; l < c->L; ++l) { ps.a[l] = c->lay[l].dil_b; ps.b[l] = c->lay[l].cin_b; }
        hipLaunchKernelGGL(wn_pairsum_kernel, dim3(cdiv(c->G, 256), c->L), dim3(256), 0, st, c->params_dev, c->b1sum, c->G, ps);
    }
    VecSum vs; vs.n = c->L;
    for (int l = 0; l < c->L; ++l) { vs.off[l] = c->lay[l].skip_b; vs.w[l] = c->skip_scale[l]; }
    hipLaunchKernelGGL(wn_vecsum_kernel, dim3(cdiv(c->S, 256)), dim3(256), 0, st, c->params_dev, c->skip_bias_total, c->S, vs);
    WN_LAUNCH_CHECK(c);
    c->packed = true;
    return WN_OK;
}

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

// =================================================================================== weight normalisation
// modules.py:44-177 (WeightNorm): kernel = tf.nn.l2_normalize(v, axes all but the last) * g  (:98-103).  One thread per
// (tensor, output channel) walks the K = numel / cout elements of its column (coalesced across the channels of a wave).
__global__ void wn_weightnorm_apply_kernel(const float* __restrict__ raw, float* __restrict__ eff, const wn_ctx::WnMap* __restrict__ map, int nt) {
    const wn_ctx::WnMap m = map[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (m.g_off < 0) {                                   // not a normalised kernel: plain copy
        for (int64_t i = c; i < m.numel; i += (int64_t)gridDim.x * blockDim.x) eff[m.eff_off + i] = raw[m.raw_off + i];
        return;
    }
    for (int ch = c; ch < m.cout; ch += gridDim.x * blockDim.x) {
        const int64_t K = m.numel / m.cout;
        float ss = 0.0f;
        for (int64_t k = 0; k < K; ++k) { const float v = raw[m.raw_off + k * m.cout + ch]; ss += v * v; }
        const float sc = raw[m.g_off + ch] * rsqrtf(fmaxf(ss, 1e-12f));      // tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), eps))
        for (int64_t k = 0; k < K; ++k) eff[m.eff_off + k * m.cout + ch] = raw[m.raw_off + k * m.cout + ch] * sc;
    }
}
// d g = sum_k dW v / ||v||;   d v = g / ||v|| * (dW - v * (sum_k dW v) / ||v||^2)
__global__ void wn_weightnorm_grad_kernel(const float* __restrict__ raw, const float* __restrict__ deff, float* __restrict__ draw,
                                          const wn_ctx::WnMap* __restrict__ map, int nt) {
    const wn_ctx::WnMap m = map[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (m.g_off < 0) {
        for (int64_t i = c; i < m.numel; i += (int64_t)gridDim.x * blockDim.x) draw[m.raw_off + i] = deff[m.eff_off + i];
        return;
    }
    for (int ch = c; ch < m.cout; ch += gridDim.x * blockDim.x) {
        const int64_t K = m.numel / m.cout;
        float ss = 0.0f, dot = 0.0f;
        for (int64_t k = 0; k < K; ++k) { const float v = raw[m.raw_off + k * m.cout + ch]; ss += v * v; dot += deff[m.eff_off + k * m.cout + ch] * v; }
        const float inv = rsqrtf(fmaxf(ss, 1e-12f)), g = raw[m.g_off + ch];
        draw[m.g_off + ch] = dot * inv;
        const float a = g * inv, b = dot * inv * inv;
        for (int64_t k = 0; k < K; ++k) draw[m.raw_off + k * m.cout + ch] = a * (deff[m.eff_off + k * m.cout + ch] - raw[m.raw_off + k * m.cout + ch] * b);
    }
}
int wn_weightnorm_apply(wn_ctx* c, const float* raw_params, hipStream_t st) {
    // keep the raw parameters for the backward (caller pointers are borrowed for the call only)
    WN_HIP(c, hipMemcpyAsync(c->raw_dev, raw_params, (size_t)c->n_raw * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(wn_weightnorm_apply_kernel, dim3(4, (unsigned)c->wmap.size()), dim3(256), 0, st, c->raw_dev, c->params_dev,
                       (const wn_ctx::WnMap*)c->wmap_dev, (int)c->wmap.size());
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
int wn_weightnorm_grad(wn_ctx* c, float* raw_grads, hipStream_t st) {
    WN_HIP(c, hipMemsetAsync(raw_grads, 0, (size_t)c->n_raw * 4, st));        // alignment gaps
    hipLaunchKernelGGL(wn_weightnorm_grad_kernel, dim3(4, (unsigned)c->wmap.size()), dim3(256), 0, st, c->raw_dev, c->deff, raw_grads,
                       (const wn_ctx::WnMap*)c->wmap_dev, (int)c->wmap.size());
    WN_LAUNCH_CHECK(c);
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

// =================================================================================== losses
__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.0f) + log1pf(__expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// scal[0] = loss accumulator, [1] = denominator, [2] = 1/denominator, [3] = non-zero count (CE)
__global__ void wn_loss_prep(const int32_t* __restrict__ lengths, int B, int T, float* scal, int shift) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float d = 0.0f;
        for (int b = 0; b < B; ++b) { int l = min(lengths[b], T); d += (float)max(l - shift, 0); }   // sum(mask[:,1:]) wavenet.py:632-638
        scal[0] = 0.0f; scal[1] = d; scal[2] = d > 0.0f ? 1.0f / d : 0.0f; scal[3] = 0.0f;
    }
}

#define WN_MAX_MIX 16
// Discretised mixture of logistics, mixture.py:18-74 + modules.py:800-817, with its gradient.
// one thread per (b, t): prediction at t scored against y[t+1] (wavenet.py:494-495).
__global__ void wn_mol_loss(const float* __restrict__ yhat, const float* __restrict__ y, const int32_t* __restrict__ lengths,
                            bf16_t* __restrict__ dY, int ldDY, float* __restrict__ scal, int B, int T, int M,
                            float num_classes, float log_scale_min, int shift, float* __restrict__ dY32) {
    // dY32 (optional): the same gradient rows in fp32 (the fp32 backward of wn_f32.hip)
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float my = 0.0f;
    if (idx < (int64_t)B * T) {
        const int b = (int)(idx / T), t = (int)(idx - (int64_t)b * T);
        const bool valid = (t + shift < T) && (t + shift < lengths[b]);
        bf16_t* drow = dY + idx * ldDY;
        if (dY32) for (int o = 0; o < ldDY; ++o) dY32[idx * ldDY + o] = 0.0f;
        if (!valid) {
            for (int o = 0; o < ldDY; ++o) drow[o] = 0;
        } else {
            const float inv_den = scal[2];
            const float yv = y[(int64_t)b * T + t + shift];
            const float* yh = yhat + ((int64_t)b * 3 * M) * T + t;
            float logit[WN_MAX_MIX], lp[WN_MAX_MIX], dmu[WN_MAX_MIX], dls[WN_MAX_MIX];
            const float D = 1.0f / (num_classes - 1.0f);
            const float logbin = logf((num_classes - 1.0f) * 0.5f);
            float mx = -INFINITY;
            for (int i = 0; i < M; ++i) { logit[i] = yh[(int64_t)i * T]; mx = fmaxf(mx, logit[i]); }
            float se = 0.0f;
            for (int i = 0; i < M; ++i) se += __expf(logit[i] - mx);
            const float lse = logf(se);
            float mlp = -INFINITY;
            for (int i = 0; i < M; ++i) {
                const float mu = yh[(int64_t)(M + i) * T];
                const float lsr = yh[(int64_t)(2 * M + i) * T];
                const float ls = fmaxf(lsr, log_scale_min);
                const float cy = yv - mu, inv = __expf(-ls);
                const float p = inv * (cy + D), m = inv * (cy - D), mid = inv * cy;
                float l, gm, gs;          // log-prob and its derivatives wrt mu and ls
                if (yv < -0.999f) { l = p - softplusf(p); const float s = sigmoidf_(-p); gm = -inv * s; gs = -p * s; }
                else if (yv > 0.999f) { l = -softplusf(m); const float s = sigmoidf_(m); gm = inv * s; gs = m * s; }
                else {
                    const float sp = sigmoidf_(p), sm = sigmoidf_(m);
                    const float cd = sp - sm;
                    if (cd > 1e-5f) {
                        l = logf(fmaxf(cd, 1e-12f));
                        const float dp = sp * (1.0f - sp), dm = sm * (1.0f - sm);
                        gm = -inv * (dp - dm) / cd; gs = (-p * dp + m * dm) / cd;
                    } else {
                        const float q = 1.0f - 2.0f * sigmoidf_(mid);
                        l = mid - ls - 2.0f * softplusf(mid) - logbin;
                        gm = -inv * q; gs = -mid * q - 1.0f;
                    }
                }
                if (lsr < log_scale_min) gs = 0.0f;       // tf.maximum passes the gradient only where x >= min
                lp[i] = l + (logit[i] - mx - lse);
                dmu[i] = gm; dls[i] = gs;
                mlp = fmaxf(mlp, lp[i]);
            }
            float sw = 0.0f;
            for (int i = 0; i < M; ++i) sw += __expf(lp[i] - mlp);
            const float loss = -(mlp + logf(sw));
            my = loss;
            for (int i = 0; i < M; ++i) {
                const float w = __expf(lp[i] - mlp) / sw;              // responsibility
                const float pi = __expf(logit[i] - mx) / se;           // prior
                drow[i] = f2bf((pi - w) * inv_den);
                drow[M + i] = f2bf(-w * dmu[i] * inv_den);
                drow[2 * M + i] = f2bf(-w * dls[i] * inv_den);
                if (dY32) { float* d32 = dY32 + idx * ldDY; d32[i] = (pi - w) * inv_den; d32[M + i] = -w * dmu[i] * inv_den; d32[2 * M + i] = -w * dls[i] * inv_den; }
            }
            for (int o = 3 * M; o < ldDY; ++o) drow[o] = 0;
        }
    }
    // block reduce
    for (int o = 32; o > 0; o >>= 1) my += __shfl_down(my, o);
    __shared__ float part[8];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = my;
    __syncthreads();
    if (threadIdx.x == 0) { float s = 0.0f; for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += part[i]; unsafeAtomicAdd(&scal[0], s); }
}

__device__ __forceinline__ float ndtrf_(float x) {      // TF special_math._ndtr, piecewise erf/erfc
    const float hs2 = 0.70710678118654752440f;
    const float w = x * hs2, z = fabsf(w);
    const float y = (z < hs2) ? 1.0f + erff(w) : ((w > 0.0f) ? 2.0f - erfcf(z) : erfcf(z));
    return 0.5f * y;
}

// Gaussian MLE, gaussian.py:5-37 + modules.py:819-836, with its gradient.
__global__ void wn_gauss_loss(const float* __restrict__ yhat, const float* __restrict__ y, const int32_t* __restrict__ lengths,
                              bf16_t* __restrict__ dY, int ldDY, float* __restrict__ scal, int B, int T,
                              float num_classes, float log_scale_min, int use_cdf, int shift, float* __restrict__ dY32) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float my = 0.0f;
    if (idx < (int64_t)B * T) {
        const int b = (int)(idx / T), t = (int)(idx - (int64_t)b * T);
        const bool valid = (t + shift < T) && (t + shift < lengths[b]);
        bf16_t* drow = dY + idx * ldDY;
        float g0 = 0.0f, g1 = 0.0f;
        if (valid) {
            const float inv_den = scal[2];
            const float yv = y[(int64_t)b * T + t + shift];
            const float mu = yhat[((int64_t)b * 2) * T + t], lsr = yhat[((int64_t)b * 2 + 1) * T + t];
            const float ls = fmaxf(lsr, log_scale_min);
            float gm, gs, loss;
            if (use_cdf) {
                const float D = 1.0f / (num_classes - 1.0f);
                const float sc = __expf(ls);
                const float zp = (yv + D - mu) / sc, zm = (yv - D - mu) / sc;
                const float diff = ndtrf_(zp) - ndtrf_(zm);
                loss = -logf(fmaxf(diff, 1e-12f));
                if (diff >= 1e-12f) {
                    const float c0 = 0.3989422804014327f;
                    const float pp = c0 * __expf(-0.5f * zp * zp), pm = c0 * __expf(-0.5f * zm * zm);
                    gm = (pp - pm) / (sc * diff);            // d(-lp)/dmu
                    gs = (zp * pp - zm * pm) / diff;         // d(-lp)/dls
                } else { gm = 0.0f; gs = 0.0f; }
            } else {
                const float e2 = __expf(-2.0f * ls), dlt = yv - mu;
                loss = 0.5f * (1.8378770664093453f + 2.0f * ls + dlt * dlt * e2);
                gm = -dlt * e2; gs = 1.0f - dlt * dlt * e2;
            }
            if (lsr < log_scale_min) gs = 0.0f;
            my = loss; g0 = gm * inv_den; g1 = gs * inv_den;
        }
        drow[0] = f2bf(g0); drow[1] = f2bf(g1);
        for (int o = 2; o < ldDY; ++o) drow[o] = 0;
        if (dY32) { float* d32 = dY32 + idx * ldDY; d32[0] = g0; d32[1] = g1; for (int o = 2; o < ldDY; ++o) d32[o] = 0.0f; }
    }
    for (int o = 32; o > 0; o >>= 1) my += __shfl_down(my, o);
    __shared__ float part[8];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = my;
    __syncthreads();
    if (threadIdx.x == 0) { float s = 0.0f; for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += part[i]; unsafeAtomicAdd(&scal[0], s); }
}

// Masked softmax cross-entropy, modules.py:781-798 (denominator = count_nonzero(masked loss)).
// pass 0: per-element loss into `tmp`, sum and non-zero count; pass 1: gradients (needs the count).
__global__ void wn_ce_loss(const float* __restrict__ yhat, const int32_t* __restrict__ y, const int32_t* __restrict__ lengths,
                           bf16_t* __restrict__ dY, int ldDY, float* __restrict__ scal, float* __restrict__ tmp,
                           int B, int T, int Q, int pass, int shift, float* __restrict__ dY32) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float my = 0.0f, cnt = 0.0f;
    if (idx < (int64_t)B * T) {
        const int b = (int)(idx / T), t = (int)(idx - (int64_t)b * T);
        const bool valid = (t + shift < T) && (t + shift < lengths[b]);
        const float* yh = yhat + ((int64_t)b * Q) * T + t;
        if (pass == 0) {
            float l = 0.0f;
            if (valid) {
                float mx = -INFINITY;
                for (int q = 0; q < Q; ++q) mx = fmaxf(mx, yh[(int64_t)q * T]);
                float se = 0.0f;
                for (int q = 0; q < Q; ++q) se += __expf(yh[(int64_t)q * T] - mx);
                const int tgt = y[(int64_t)b * T + t + shift];
                l = mx + logf(se) - yh[(int64_t)tgt * T];
            }
            tmp[idx] = l; my = l; cnt = (l != 0.0f) ? 1.0f : 0.0f;
        } else {
            bf16_t* drow = dY + idx * ldDY;
            if (dY32) for (int q = 0; q < ldDY; ++q) dY32[idx * ldDY + q] = 0.0f;
            if (!valid) { for (int q = 0; q < ldDY; ++q) drow[q] = 0; }
            else {
                const float inv = 1.0f / scal[3];
                float mx = -INFINITY;
                for (int q = 0; q < Q; ++q) mx = fmaxf(mx, yh[(int64_t)q * T]);
                float se = 0.0f;
                for (int q = 0; q < Q; ++q) se += __expf(yh[(int64_t)q * T] - mx);
                const int tgt = y[(int64_t)b * T + t + shift];
                for (int q = 0; q < Q; ++q) {
                    const float gq = (__expf(yh[(int64_t)q * T] - mx) / se - (q == tgt ? 1.0f : 0.0f)) * inv;
                    drow[q] = f2bf(gq);
                    if (dY32) dY32[idx * ldDY + q] = gq;
                }
                for (int q = Q; q < ldDY; ++q) drow[q] = 0;
            }
        }
    }
    if (pass == 0) {
        for (int o = 32; o > 0; o >>= 1) { my += __shfl_down(my, o); cnt += __shfl_down(cnt, o); }
        __shared__ float part[16];
        if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = my; part[8 + (threadIdx.x >> 6)] = cnt; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.0f, n = 0.0f;
            for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { s += part[i]; n += part[8 + i]; }
            unsafeAtomicAdd(&scal[0], s); unsafeAtomicAdd(&scal[3], n);
        }
    }
}

__global__ void wn_loss_finalize(float* scal, float* loss_out, int use_count) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float den = use_count ? scal[3] : scal[1];
        *loss_out = scal[0] / den;
    }
}

// shift = 1: training (prediction at t scored against sample t+1, wavenet.py:488-495);
// shift = 0: evaluation of the incremental loop's raw outputs (wavenet.py:497-506).
int wn_loss_run(wn_ctx* c, const float* yhat, const void* y, const int32_t* lengths, int B, int T, int shift, float* loss_out, hipStream_t st) {
    const int64_t n = (int64_t)B * T;
    if (n > c->NT) WN_FAIL(c, WN_E_SHAPE, "loss: B*T exceeds the workspace");
    const int ldDY = (c->O + 15) / 16 * 16;
    float* const dy32 = c->dy32_next; c->dy32_next = nullptr;      // set by the fp32 forward for ITS loss call only
    hipLaunchKernelGGL(wn_loss_prep, dim3(1), dim3(64), 0, st, lengths, B, T, c->scal, shift);
    if (c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE) {
        float* tmp = c->DC;      // scratch, free at this point of the step
        hipLaunchKernelGGL(wn_ce_loss, dim3(cdiv(n, 256)), dim3(256), 0, st, yhat, (const int32_t*)y, lengths, c->DY, ldDY, c->scal, tmp, B, T, c->O, 0, shift, dy32);
        hipLaunchKernelGGL(wn_ce_loss, dim3(cdiv(n, 256)), dim3(256), 0, st, yhat, (const int32_t*)y, lengths, c->DY, ldDY, c->scal, tmp, B, T, c->O, 1, shift, dy32);
        hipLaunchKernelGGL(wn_loss_finalize, dim3(1), dim3(64), 0, st, c->scal, loss_out, 1);
    } else if (c->O == 2) {
        hipLaunchKernelGGL(wn_gauss_loss, dim3(cdiv(n, 256)), dim3(256), 0, st, yhat, (const float*)y, lengths, c->DY, ldDY, c->scal, B, T,
                           (float)c->cfg.quantize_channels, c->cfg.log_scale_min_gauss, c->cfg.cdf_loss, shift, dy32);
        hipLaunchKernelGGL(wn_loss_finalize, dim3(1), dim3(64), 0, st, c->scal, loss_out, 0);
    } else {
        if (c->O / 3 > WN_MAX_MIX) WN_FAIL(c, WN_E_UNSUPPORTED, "more than %d mixture components", WN_MAX_MIX);
        hipLaunchKernelGGL(wn_mol_loss, dim3(cdiv(n, 256)), dim3(256), 0, st, yhat, (const float*)y, lengths, c->DY, ldDY, c->scal, B, T, c->O / 3,
                           (float)c->cfg.quantize_channels, c->cfg.log_scale_min, shift, dy32);
        hipLaunchKernelGGL(wn_loss_finalize, dim3(1), dim3(64), 0, st, c->scal, loss_out, 0);
    }
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
int wn_loss_fwd_bwd(wn_ctx* c, float* loss_out, hipStream_t st) {
    WnTraceScope trace_scope(c, st, WN_TR_LOSS);
    return wn_loss_run(c, c->YHAT, c->fy, c->flen, c->fB, c->fT, 1, loss_out, st);
}
extern "C" int wn_loss(wn_ctx* c, const float* y_hat, const void* y, const int32_t* lengths, int32_t B, int32_t T, int32_t shift, float* loss_out, void* stream) {
    if (!c || !y_hat || !y || !lengths || !loss_out) return WN_E_ARG;
    if (shift != 0 && shift != 1) WN_FAIL(c, WN_E_ARG, "shift must be 0 or 1");
    if (c->inference) WN_FAIL(c, WN_E_STATE, "wn_loss on an inference-only context (the loss gradient buffer is training workspace)");
    c->have_loss = false;      // DY is overwritten
    return wn_loss_run(c, y_hat, y, lengths, B, T, shift, loss_out, (hipStream_t)stream);
}

// =================================================================================== optimiser
// wavenet.py:586-613: per-tensor tf.clip_by_norm -> tf.clip_by_value -> tf.train.AdamOptimizer (epsilon-hat) -> EMA
__device__ __forceinline__ int find_tensor(const int32_t* __restrict__ offs, int nt, int64_t i) {
    int lo = 0, hi = nt - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (offs[mid] <= i) lo = mid; else hi = mid - 1; }
    return lo;
}
#define WN_NORM_SPAN 4096      // floats per wave
// Per-variable squared norms WITHOUT atomics: the replicas of a data-parallel job apply clip_by_norm to the SAME all-reduced gradient
// and must come out bit-identical, which a float-atomic accumulation order does not give (round 2 flushed one atomic per wave and
// tensor: whenever a norm exceeded the clip threshold the scale differed by ulps between ranks and nothing re-synchronised them).
// Stage 1: one wave per span of the host-built table (never crosses a tensor), fixed lane-strided order + butterfly -> part[span];
// stage 2: one wave per tensor sums its spans in a fixed order.
__global__ __launch_bounds__(256) void wn_norm2_span_kernel(const float* __restrict__ g, const int32_t* __restrict__ spans, int nspans, float* __restrict__ part) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= nspans) return;
    const int i0 = spans[2 * w], i1 = spans[2 * w + 1];
    float s = 0.0f;          // (tensor offsets are multiples of 8 floats: every span starts 16-B aligned)
    for (int j = i0 + lane * 4; j < i1; j += 256) {
        if (j + 3 < i1) { const float4 v = *reinterpret_cast<const float4*>(g + j); s += v.x * v.x; s += v.y * v.y; s += v.z * v.z; s += v.w * v.w; }
        else for (int k = j; k < i1; ++k) s += g[k] * g[k];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) part[w] = s;
}
__global__ __launch_bounds__(256) void wn_norm2_tensor_kernel(const float* __restrict__ part, const int32_t* __restrict__ first, int nt, float* __restrict__ norm2) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= nt) return;
    float s = 0.0f;
    for (int i = first[t] + lane; i < first[t + 1]; i += 64) s += part[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) norm2[t] = s;
}
__global__ void wn_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                               float* __restrict__ ema, const int32_t* __restrict__ offs, int nt, int64_t n,
                               const float* __restrict__ norm2, int clip, float max_norm, float max_value,
                               float lr_t, float b1, float b2, float eps, float ema_decay) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    auto upd = [&](float gi, float& pi, float& mi, float& vi, float& ei, float cs) {
        if (clip) gi = fminf(fmaxf(gi * max_norm / cs, -max_value), max_value);      // tf.clip_by_norm: t * clip / max(norm, clip)
        mi = b1 * mi + (1.0f - b1) * gi;
        vi = b2 * vi + (1.0f - b2) * gi * gi;
        pi = pi - lr_t * mi / (sqrtf(vi) + eps);
        ei = ei - (1.0f - ema_decay) * (ei - pi);
    };
    auto clip_scale = [&](int64_t j) { return fmaxf(sqrtf(norm2[find_tensor(offs, nt, j)]), max_norm); };      // the denominator
    if (i + 3 < n) {
        float4 G = *reinterpret_cast<const float4*>(g + i), P = *reinterpret_cast<float4*>(p + i), M = *reinterpret_cast<float4*>(m + i);
        float4 V = *reinterpret_cast<float4*>(v + i), E = *reinterpret_cast<float4*>(ema + i);
        float c0 = 1.0f, c1 = 1.0f, c2 = 1.0f, c3 = 1.0f;
        if (clip) {
            const int t0 = find_tensor(offs, nt, i);
            const int64_t tend = (t0 + 1 < nt) ? (int64_t)offs[t0 + 1] : n;
            c0 = fmaxf(sqrtf(norm2[t0]), max_norm);
            if (i + 3 < tend) { c1 = c2 = c3 = c0; }
            else { c1 = clip_scale(i + 1); c2 = clip_scale(i + 2); c3 = clip_scale(i + 3); }
        }
        upd(G.x, P.x, M.x, V.x, E.x, c0); upd(G.y, P.y, M.y, V.y, E.y, c1); upd(G.z, P.z, M.z, V.z, E.z, c2); upd(G.w, P.w, M.w, V.w, E.w, c3);
        *reinterpret_cast<float4*>(p + i) = P; *reinterpret_cast<float4*>(m + i) = M; *reinterpret_cast<float4*>(v + i) = V; *reinterpret_cast<float4*>(ema + i) = E;
    } else {
        for (int64_t j = i; j < n; ++j) {
            float pi = p[j], mi = m[j], vi = v[j], ei = ema[j];
            upd(g[j], pi, mi, vi, ei, clip ? clip_scale(j) : 1.0f);
            p[j] = pi; m[j] = mi; v[j] = vi; ema[j] = ei;
        }
    }
}

int wn_optim_impl(wn_ctx* c, float* p, const float* g, float* m, float* v, float* ema, float lr, int64_t step, hipStream_t st) {
    const wn_config& h = c->cfg;
    const int nt = (int)c->raw_tensors.size();
    const int64_t n = c->n_raw;
    struct TraceDone { wn_ctx* c; ~TraceDone() { if (c->trace_state == 3) c->trace_state = 2; } } trace_done{c};      // (destroyed AFTER trace_scope: the end stamp is enqueued first)
    WnTraceScope trace_scope(c, st, WN_TR_OPTIMISER);
    if (h.clip_gradients) {
        hipLaunchKernelGGL(wn_norm2_span_kernel, dim3(cdiv(c->norm_nspans, 4)), dim3(256), 0, st, g, c->norm_spans_dev, c->norm_nspans, c->norm_part_dev);
        hipLaunchKernelGGL(wn_norm2_tensor_kernel, dim3(cdiv(nt, 4)), dim3(256), 0, st, c->norm_part_dev, c->norm_first_dev, nt, c->norm2_dev);
    }
    const double t = (double)(step + 1);
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)h.adam_beta2, t)) / (1.0 - pow((double)h.adam_beta1, t)));
    hipLaunchKernelGGL(wn_adam_kernel, dim3(cdiv(cdiv(n, 4), 256)), dim3(256), 0, st, p, g, m, v, ema, c->tensor_offsets_dev, nt, n, c->norm2_dev,
                       h.clip_gradients, h.gradient_max_norm, h.gradient_max_value, lr_t, h.adam_beta1, h.adam_beta2, h.adam_epsilon, h.ema_decay);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}

// =================================================================================== mu-law codec
// util.py:30-129.  The quantiser is evaluated through the exact float32 decision thresholds of the
// reference's numpy float32 path (wn_mulaw_tables.h, generated by oracle/gen_mulaw_tables.py from the
// reference's own util.py): bit-exact indices for every float32 input, independent of device log1p ULPs.
__global__ void wn_mulaw_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const float s = (v > 0.0f) ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
    y[i] = (float)((double)s * log1p(255.0 * fabs((double)v)) / 5.545177444479562);
}
__global__ void wn_inv_mulaw_kernel(const float* __restrict__ y, float* __restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = y[i];
    const float s = (v > 0.0f) ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
    x[i] = (float)((double)s * (1.0 / 255.0) * (pow(256.0, fabs((double)v)) - 1.0));
}
__device__ __forceinline__ int mulaw_q(float v) {
    // number of thresholds <= v  (thresholds ascending; NaN -> 0)
    int lo = 0, hi = 255;               // answer in [0,255]
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (v >= WN_MULAW_THRESH[mid - 1]) lo = mid; else hi = mid - 1; }
    return lo;
}
__global__ void wn_mulaw_quantize_kernel(const float* __restrict__ x, int32_t* __restrict__ q, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) q[i] = mulaw_q(x[i]);
}
__global__ void wn_inv_mulaw_quantize_kernel(const int32_t* __restrict__ q, float* __restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int v = min(max(q[i], 0), 255); x[i] = WN_MULAW_DECODE[v]; }
}
__global__ void wn_argmax_kernel(const float* __restrict__ logits, int32_t* __restrict__ out, int B, int Q, int T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T) return;
    const int b = (int)(i / T), t = (int)(i - (int64_t)b * T);
    const float* p = logits + ((int64_t)b * Q) * T + t;
    float best = p[0]; int bi = 0;
    for (int q = 1; q < Q; ++q) { const float v = p[(int64_t)q * T]; if (v > best) { best = v; bi = q; } }   // first max wins (tf.argmax)
    out[i] = bi;
}

#define EW_LAUNCH(kern, n, st, ...) do { if ((n) > 0) hipLaunchKernelGGL(kern, dim3(cdiv((n), 256)), dim3(256), 0, (hipStream_t)(st), __VA_ARGS__); \
    hipError_t _e = hipGetLastError(); if (_e != hipSuccess) { g_create_err = hipGetErrorString(_e); return WN_E_HIP; } return WN_OK; } while (0)

extern "C" int wn_mulaw(const float* x, float* y, int64_t n, void* st) { if (!x || !y || n < 0) return WN_E_ARG; EW_LAUNCH(wn_mulaw_kernel, n, st, x, y, n); }
extern "C" int wn_inv_mulaw(const float* y, float* x, int64_t n, void* st) { if (!x || !y || n < 0) return WN_E_ARG; EW_LAUNCH(wn_inv_mulaw_kernel, n, st, y, x, n); }
extern "C" int wn_mulaw_quantize(const float* x, int32_t* q, int64_t n, void* st) { if (!x || !q || n < 0) return WN_E_ARG; EW_LAUNCH(wn_mulaw_quantize_kernel, n, st, x, q, n); }
extern "C" int wn_inv_mulaw_quantize(const int32_t* q, float* x, int64_t n, void* st) { if (!x || !q || n < 0) return WN_E_ARG; EW_LAUNCH(wn_inv_mulaw_quantize_kernel, n, st, q, x, n); }
extern "C" int wn_argmax_channels(const float* l, int32_t* o, int32_t B, int32_t Q, int32_t T, void* st) {
    if (!l || !o || B <= 0 || Q <= 0 || T <= 0) return WN_E_ARG; EW_LAUNCH(wn_argmax_kernel, (int64_t)B * T, st, l, o, B, Q, T); }

// =================================================================================== samplers
// mixture.py:76-107, gaussian.py:39-52, wavenet.py:861-867; noise [T][B][nps] supplied by the caller.
__device__ __forceinline__ float sample_mol(const float* p, int64_t stride, int M, const float* nz, float log_scale_min) {
    float best = -INFINITY; int bi = 0;
    for (int i = 0; i < M; ++i) { const float v = p[(int64_t)i * stride] - logf(-logf(nz[i])); if (v > best) { best = v; bi = i; } }
    const float mu = p[(int64_t)(M + bi) * stride];
    const float ls = fmaxf(p[(int64_t)(2 * M + bi) * stride], log_scale_min);
    const float u = nz[M];
    const float x = mu + expf(ls) * (logf(u) - logf(1.0f - u));
    return fminf(fmaxf(x, -1.0f), 1.0f);
}
__device__ __forceinline__ float sample_gauss(const float* p, int64_t stride, const float* nz, float lsmin) {
    const float x = p[0] + expf(fmaxf(p[stride], lsmin)) * nz[0];
    return fminf(fmaxf(x, -1.0f), 1.0f);
}
__device__ __forceinline__ int sample_cat(const float* p, int64_t stride, int Q, const float* nz) {
    float best = -INFINITY; int bi = 0;
    for (int q = 0; q < Q; ++q) { const float v = p[(int64_t)q * stride] - logf(-logf(nz[q])); if (v > best) { best = v; bi = q; } }
    return bi;
}
__global__ void wn_sample_kernel(const float* __restrict__ yhat, const float* __restrict__ noise, void* __restrict__ out,
                                 int B, int T, int O, int mode, int nps, float lsmin) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T) return;
    const int b = (int)(i / T), t = (int)(i - (int64_t)b * T);
    const float* p = yhat + ((int64_t)b * O) * T + t;
    const float* nz = noise + ((int64_t)t * B + b) * nps;
    if (mode == 0) ((float*)out)[i] = sample_mol(p, T, O / 3, nz, lsmin);
    else if (mode == 1) ((float*)out)[i] = sample_gauss(p, T, nz, lsmin);
    else ((int32_t*)out)[i] = sample_cat(p, T, O, nz);
}
int wn_sample_impl(wn_ctx* c, const float* y_hat, int B, int T, const float* noise, void* out, hipStream_t st) {
    const int mode = c->cfg.input_type == WN_INPUT_MULAW_QUANTIZE ? 2 : (c->O == 2 ? 1 : 0);
    const float lsmin = mode == 1 ? c->cfg.log_scale_min_gauss : c->cfg.log_scale_min;
    hipLaunchKernelGGL(wn_sample_kernel, dim3(cdiv((int64_t)B * T, 256)), dim3(256), 0, st, y_hat, noise, out, B, T, c->O, mode, wn_noise_per_step(c), lsmin);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}

// =================================================================================== device noise stream
// Sampling noise of wn_synthesize(noise = NULL): replaces tf.random_uniform (mixture.py:91,104), Normal.sample (gaussian.py:50) and
// tf.multinomial's generator (wavenet.py:865).  Philox4x32-10 (Salmon et al., SC'11), key = the 64-bit seed, counter = (group index,
// 0, 0): group g yields the four 32-bit words of elements 4g .. 4g+3 of the flat [T][B][nps] buffer, so a draw depends only on
// (seed, element index) -- any launch geometry reproduces it (tests/hip_util.py mirrors it in numpy, bit for bit).
__host__ __device__ inline void wn_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
// 24 random bits -> (0, 1) open at both ends, then clamped to the reference's range [1e-5, 1 - 1e-5] (mixture.py:91,104)
__host__ __device__ inline float wn_u01(uint32_t w) { return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__global__ void wn_noise_kernel(float* __restrict__ out, int64_t n, uint32_t k0, uint32_t k1, int gaussian) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g * 4 >= n) return;
    uint32_t w[4];
    wn_philox4x32_10((uint32_t)g, (uint32_t)(g >> 32), k0, k1, w);
    float v[4];
    if (gaussian) {                                   // Box-Muller on the word pairs (w0, w1) and (w2, w3)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float r = sqrtf(-2.0f * logf(wn_u01(w[2 * h]))), ph = 6.28318530717958647692f * wn_u01(w[2 * h + 1]);
            v[2 * h] = r * cosf(ph); v[2 * h + 1] = r * sinf(ph);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(wn_u01(w[j]), 1e-5f), 1.0f - 1e-5f);     // exact ops only (no rounding: any compiler, and the numpy mirror, give the same bits)
    }
    if (g * 4 + 3 < n) *reinterpret_cast<float4*>(out + g * 4) = make_float4(v[0], v[1], v[2], v[3]);
    else for (int j = 0; j < 4 && g * 4 + j < n; ++j) out[g * 4 + j] = v[j];
}
int wn_fill_noise_impl(wn_ctx* c, float* noise, int B, int T, uint64_t seed, hipStream_t st) {
    const int64_t n = (int64_t)B * T * wn_noise_per_step(c);
    if ((reinterpret_cast<uintptr_t>(noise) & 15) != 0) WN_FAIL(c, WN_E_ARG, "noise buffer must be 16-byte aligned");
    const int gaussian = (c->cfg.input_type != WN_INPUT_MULAW_QUANTIZE && c->O == 2) ? 1 : 0;
    hipLaunchKernelGGL(wn_noise_kernel, dim3(cdiv((n + 3) / 4, 256)), dim3(256), 0, st, noise, n, (uint32_t)seed, (uint32_t)(seed >> 32), gaussian);
    WN_LAUNCH_CHECK(c);
    return WN_OK;
}
