// Training step of the WaveNet stack: forward (wavenet.py:650-721), loss (476-495) and the hand-written
// backward (replaces tf.gradients / optimizer.compute_gradients, wavenet.py:557).
#include "wn_tile.h"

int wn_first_conv(wn_ctx* c, hipStream_t st);
int wn_first_conv_grad(wn_ctx* c, const bf16_t* g0, float* grads, hipStream_t st);
int wn_upsample_bwd(wn_ctx* c, const float* dc_final, float* grads, hipStream_t st);
int wn_loss_fwd_bwd(wn_ctx* c, float* loss_out, hipStream_t st);

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

// ================================================================================================
static void set_dropout(wn_ctx* c, int layer, uint32_t& klo, uint32_t& khi, uint32_t& th, float& ks, int& ld) {
    wn_layer_key(c->fseed, layer, &klo, &khi);
    th = (uint32_t)lrintf(c->cfg.dropout * 65536.0f);
    ks = 1.0f / (1.0f - c->cfg.dropout);
    ld = c->R;
}

static void base_args(wn_ctx* c, GemmArgs& a, const PackedW& w) {
    memset(&a, 0, sizeof a);
    a.Apk = w.dev; a.ksteps_total = w.K >> 4;
    a.nrep = 1; a.rep_stride = 0;
    a.B = c->fB; a.T = c->fT;
    a.e.scale = 1.0f; a.e.GH = c->GH; a.e.M_valid = w.M_valid;
}

static SrcSeg seg(const bf16_t* base, int ld, int col0, int nk, int shift, int drop) {
    SrcSeg s; s.base = base; s.ld = ld; s.col0 = col0; s.nk = nk; s.shift = shift; s.dropout = drop; return s;
}

static void prof_mark(wn_ctx* c, hipStream_t st) {
    if (c->pev_used == c->pev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; c->pev.push_back(e); }
    (void)hipEventRecord(c->pev[c->pev_used++], st);
}
extern "C" int wn_profile(wn_ctx* c, int32_t enable) { if (!c) return WN_E_ARG; c->prof = enable != 0; c->pev_used = 0; return WN_OK; }
// total milliseconds and number of launches of the dominant kernel (gate GEMM) since wn_profile(ctx, 1); synchronises.
extern "C" int wn_profile_result(wn_ctx* c, double* total_ms, int64_t* launches) {
    if (!c || !total_ms || !launches) return WN_E_ARG;
    double tot = 0.0; int64_t n = 0;
    for (size_t i = 0; i + 1 < c->pev_used; i += 2) {
        float ms = 0.0f;
        WN_HIP(c, hipEventSynchronize(c->pev[i + 1]));
        WN_HIP(c, hipEventElapsedTime(&ms, c->pev[i], c->pev[i + 1]));
        tot += ms; ++n;
    }
    *total_ms = tot; *launches = n;
    return WN_OK;
}

int wn_fwd_impl(wn_ctx* c, hipStream_t st, float* loss_out, float* y_hat_out) {
    const int L = c->L, R = c->R, G = c->G, GH = c->GH, S = c->S, C = c->C;
    const int64_t NT = c->NT;
    int rc;
    if ((rc = wn_upsample_fwd(c, nullptr, c->fc, c->fB, c->fTc, st))) return rc;     // wavenet.py:680-702
    if ((rc = wn_first_conv(c, st))) return rc;                                      // wavenet.py:705
    const int drop = c->cfg.dropout > 0.0f ? 1 : 0;
    for (int l = 0; l < L; ++l) {                                                    // wavenet.py:706-715 / modules.py:471-521
        const int d = c->dil[l];
        const bf16_t* Xl = c->X + (size_t)l * NT * R;
        const bf16_t* XDl = c->XD + (size_t)l * NT * R;      // dropout already applied by the producer
        GemmArgs a; base_args(c, a, c->packs[l].w1);
        a.nseg = 4;
        a.seg[0] = seg(XDl, R, 0, R, -2 * d, 0);
        a.seg[1] = seg(XDl, R, 0, R, -d, 0);
        a.seg[2] = seg(XDl, R, 0, R, 0, 0);
        a.seg[3] = seg(c->cbt, C, 0, C, 0, 0);
        a.e.bias = c->b1sum + (size_t)l * G;
        a.e.out0 = c->TS + (size_t)l * NT * G; a.e.ld_out0 = G;
        a.e.out1 = c->U + (size_t)l * NT * GH; a.e.ld_out1 = GH;
        if (c->prof) prof_mark(c, st);
        if ((rc = wn_launch_gemm<EPI_GATE>(c, a, c->packs[l].w1.M, st))) return rc;
        if (c->prof) prof_mark(c, st);
        if (l + 1 < L) {      // the residual output of the last layer is never consumed (wavenet.py:716)
            GemmArgs o; base_args(c, o, c->packs[l].wo);
            o.nseg = 1; o.seg[0] = seg(c->U + (size_t)l * NT * GH, GH, 0, GH, 0, 0);
            o.e.bias = c->params_dev + c->lay[l].out_b;
            o.e.in0 = Xl; o.e.ld_in0 = R;
            o.e.scale = c->res_scale;
            o.e.out0 = c->X + (size_t)(l + 1) * NT * R; o.e.ld_out0 = R;
            if (drop) {
                o.e.out1 = c->XD + (size_t)(l + 1) * NT * R; o.e.ld_out1 = R;
                set_dropout(c, l + 1, o.key_lo, o.key_hi, o.thresh16, o.keep_scale, o.drop_ld);
            }
            if ((rc = wn_launch_gemm<EPI_STORE_BF16>(c, o, c->packs[l].wo.M, st))) return rc;
        }
    }
    {   // skip sum over all layers as one contraction, + ReLU (wavenet.py:716-719 first activation)
        GemmArgs a; base_args(c, a, c->wskip);
        a.nseg = 1; a.seg[0] = seg(c->U, GH, 0, GH, 0, 0); a.nrep = L; a.rep_stride = NT * GH;
        a.e.bias = c->skip_bias_total; a.e.relu = 1; a.e.out0 = c->R1; a.e.ld_out0 = S;
        if ((rc = wn_launch_gemm<EPI_STORE_BF16>(c, a, c->wskip.M, st))) return rc;
    }
    {   // final_convolution_1 + ReLU
        GemmArgs a; base_args(c, a, c->wh1);
        a.nseg = 1; a.seg[0] = seg(c->R1, S, 0, S, 0, 0);
        a.e.bias = c->params_dev + c->fin1_b; a.e.relu = 1; a.e.out0 = c->H2; a.e.ld_out0 = S;
        if ((rc = wn_launch_gemm<EPI_STORE_BF16>(c, a, c->wh1.M, st))) return rc;
    }
    {   // final_convolution_2 -> y_hat [B,O,T] fp32
        GemmArgs a; base_args(c, a, c->wh2);
        a.nseg = 1; a.seg[0] = seg(c->H2, S, 0, S, 0, 0);
        a.e.bias = c->params_dev + c->fin2_b; a.e.out0 = c->YHAT; a.e.M_valid = c->O;
        if ((rc = wn_launch_gemm<EPI_STORE_F32_BOT>(c, a, c->wh2.M, st))) return rc;
    }
    if (y_hat_out) WN_HIP(c, hipMemcpyAsync(y_hat_out, c->YHAT, (size_t)c->fB * c->O * c->fT * 4, hipMemcpyDeviceToDevice, st));
    if (loss_out) { if ((rc = wn_loss_fwd_bwd(c, loss_out, st))) return rc; c->have_loss = true; }
    else c->have_loss = false;
    return WN_OK;
}

int wn_bwd_impl(wn_ctx* c, float* grads, hipStream_t st) {
    if (!c->have_loss) WN_FAIL(c, WN_E_STATE, "wn_train_bwd needs a forward that computed the loss (loss_out != NULL)");
    const int L = c->L, R = c->R, G = c->G, GH = c->GH, S = c->S, C = c->C, O = c->O;
    const int64_t NT = c->NT;
    const int ldDY = (O + 15) / 16 * 16;
    int rc;
    WN_HIP(c, hipMemsetAsync(grads, 0, (size_t)c->n_params * 4, st));
    const int64_t rows = (int64_t)c->fB * c->fT;
    // ---- head (wavenet.py:136-149)
    {   // d final_convolution_2 = H2^T dY
        WgArgs w; memset(&w, 0, sizeof w);
        w.nseg = 1; w.seg[0] = seg(c->H2, S, 0, S, 0, 0); w.ones_row = 1;
        w.Bm = c->DY; w.ldb = ldDY; w.colb0 = 0; w.N = O;
        w.out = grads + c->fin2_k; w.ldw = O; w.bias_out = grads + c->fin2_b; w.scale = 1.0f; w.B = c->fB; w.T = c->fT;
        if ((rc = launch_wgrad(c, w, st))) return rc;
    }
    {   // d pre1 = (W2 dY) * (H2 > 0)
        GemmArgs a; base_args(c, a, c->wh2T);
        a.nseg = 1; a.seg[0] = seg(c->DY, ldDY, 0, c->wh2T.K, 0, 0);
        a.e.in0 = c->H2; a.e.ld_in0 = S; a.e.out0 = c->DPRE1; a.e.ld_out0 = S;
        if ((rc = wn_launch_gemm<EPI_MASK_STORE>(c, a, c->wh2T.M, st))) return rc;
    }
    {   // d final_convolution_1 = R1^T dpre1
        WgArgs w; memset(&w, 0, sizeof w);
        w.nseg = 1; w.seg[0] = seg(c->R1, S, 0, S, 0, 0); w.ones_row = 1;
        w.Bm = c->DPRE1; w.ldb = S; w.N = S;
        w.out = grads + c->fin1_k; w.ldw = S; w.bias_out = grads + c->fin1_b; w.scale = 1.0f; w.B = c->fB; w.T = c->fT;
        if ((rc = launch_wgrad(c, w, st))) return rc;
    }
    {   // d skip = (W1 dpre1) * (skips > 0)
        GemmArgs a; base_args(c, a, c->wh1T);
        a.nseg = 1; a.seg[0] = seg(c->DPRE1, S, 0, S, 0, 0);
        a.e.in0 = c->R1; a.e.ld_in0 = S; a.e.out0 = c->DSKIP; a.e.ld_out0 = S;
        if ((rc = wn_launch_gemm<EPI_MASK_STORE>(c, a, c->wh1T.M, st))) return rc;
    }
    // ---- residual stack, top to bottom.  GX buffers hold rho * dL/dh_{l+1} (rho = sqrt(.5) if residual_legacy)
    bf16_t* gx_up = c->GX1;           // gradient wrt the output of the current layer (zero for the top layer)
    bf16_t* gx_dn = c->GX0;
    WN_HIP(c, hipMemsetAsync(gx_up, 0, (size_t)rows * R * 2, st));
    const int drop = c->cfg.dropout > 0.0f ? 1 : 0;
    for (int l = L - 1; l >= 0; --l) {
        const int d = c->dil[l];
        const bf16_t* Xl = c->X + (size_t)l * NT * R;
        const bf16_t* Ul = c->U + (size_t)l * NT * GH;
        bf16_t* DZl = c->DZ + (size_t)l * NT * G;
        const bool top = (l == L - 1);
        {   // d z: through the 1x1 convs and the gate (modules.py:510-515)
            GemmArgs a; base_args(c, a, c->packs[l].w2T);
            a.nseg = 2;
            a.seg[0] = seg(gx_up, R, 0, R, 0, 0);
            a.seg[1] = seg(c->DSKIP, S, 0, S, 0, 0);
            a.e.in0 = c->TS + (size_t)l * NT * G; a.e.ld_in0 = G; a.e.out0 = DZl; a.e.ld_out0 = G;
            if ((rc = wn_launch_gemm<EPI_DGATE>(c, a, c->packs[l].w2T.M, st))) return rc;
        }
        {   // d [W_dil; W_cin], d bias
            WgArgs w; memset(&w, 0, sizeof w);
            w.nseg = 4;
            const bf16_t* XDl = c->XD + (size_t)l * NT * R;
            w.seg[0] = seg(XDl, R, 0, R, -2 * d, 0); w.seg[1] = seg(XDl, R, 0, R, -d, 0); w.seg[2] = seg(XDl, R, 0, R, 0, 0);
            w.seg[3] = seg(c->cbt, C, 0, C, 0, 0);
            w.ones_row = 1;
            w.Bm = DZl; w.ldb = G; w.N = G;
            w.out = grads + c->lay[l].dil_k; w.ldw = G; w.bias_out = grads + c->lay[l].dil_b; w.bias_out2 = grads + c->lay[l].cin_b;
            w.scale = 1.0f; w.B = c->fB; w.T = c->fT;
            if ((rc = launch_wgrad(c, w, st))) return rc;
        }
        {   // d W_skip (scaled by the legacy factor c_l), d skip bias
            WgArgs w; memset(&w, 0, sizeof w);
            w.nseg = 1; w.seg[0] = seg(Ul, GH, 0, GH, 0, 0); w.ones_row = 1;
            w.Bm = c->DSKIP; w.ldb = S; w.N = S;
            w.out = grads + c->lay[l].skip_k; w.ldw = S; w.bias_out = grads + c->lay[l].skip_b;
            w.scale = c->skip_scale[l]; w.B = c->fB; w.T = c->fT;
            if ((rc = launch_wgrad(c, w, st))) return rc;
        }
        if (!top) {   // d W_out, d out bias (the top layer's residual branch is dead: zero gradient)
            WgArgs w; memset(&w, 0, sizeof w);
            w.nseg = 1; w.seg[0] = seg(Ul, GH, 0, GH, 0, 0); w.ones_row = 1;
            w.Bm = gx_up; w.ldb = R; w.N = R;
            w.out = grads + c->lay[l].out_k; w.ldw = R; w.bias_out = grads + c->lay[l].out_b;
            w.scale = 1.0f; w.B = c->fB; w.T = c->fT;
            if ((rc = launch_wgrad(c, w, st))) return rc;
        }
        {   // d h_l = dropout-mask * conv^T(dz) + residual path   (modules.py:484, 517-520)
            GemmArgs a; base_args(c, a, c->packs[l].w1T);
            a.nseg = 3;
            a.seg[0] = seg(DZl, G, 0, G, 2 * d, 0);
            a.seg[1] = seg(DZl, G, 0, G, d, 0);
            a.seg[2] = seg(DZl, G, 0, G, 0, 0);
            set_dropout(c, l, a.key_lo, a.key_hi, a.thresh16, a.keep_scale, a.drop_ld);
            if (!drop) a.thresh16 = 0;
            a.e.in0 = top ? nullptr : gx_up; a.e.ld_in0 = R;
            a.e.scale = (l > 0) ? c->res_scale : 1.0f;
            a.e.out0 = gx_dn; a.e.ld_out0 = R;
            if ((rc = wn_launch_gemm<EPI_DX>(c, a, c->packs[l].w1T.M, st))) return rc;
        }
        bf16_t* t = gx_up; gx_up = gx_dn; gx_dn = t;
    }
    // gx_up now holds dL/dh_0
    if ((rc = wn_first_conv_grad(c, gx_up, grads, st))) return rc;
    if (c->cfg.upsample_type != WN_UP_NEAREST) {
        // d c_up[b][cc][t] = sum_l W_cin_l dz_l  as one contraction over all layers
        GemmArgs a; base_args(c, a, c->wcT);
        a.nseg = 1; a.seg[0] = seg(c->DZ, G, 0, G, 0, 0); a.nrep = L; a.rep_stride = NT * G;
        a.e.out0 = c->DC; a.e.M_valid = C;
        if ((rc = wn_launch_gemm<EPI_STORE_F32_BOT>(c, a, c->wcT.M, st))) return rc;
        if ((rc = wn_upsample_bwd(c, c->DC, grads, st))) return rc;
    }
    return WN_OK;
}
