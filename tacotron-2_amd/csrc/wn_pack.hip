// Parameters -> what the kernels read: the fragment-ordered bf16 packs of every MFMA contraction (one launch for all 126 matrices), the
// summed bias vectors, and weight normalisation (raw v, g -> effective kernels and back for the gradients; modules.py:44-177).
#include "wn_common.h"
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
