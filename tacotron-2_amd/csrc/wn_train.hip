// Training step of the WaveNet stack: forward (wavenet.py:650-721), loss (476-495) and the hand-written
// backward (replaces tf.gradients / optimizer.compute_gradients, wavenet.py:557).
#include "wn_tile8p.h"
#include "wn_wgrad.h"

int wn_first_conv(wn_ctx* c, hipStream_t st);
int wn_first_conv_grad(wn_ctx* c, const bf16_t* g0, float* grads, hipStream_t st);
int wn_upsample_bwd(wn_ctx* c, const float* dc_final, float* grads, hipStream_t st);
int wn_loss_fwd_bwd(wn_ctx* c, float* loss_out, hipStream_t st);

// Ablation switches of the DIAGNOSTIC build only (csrc/build.py --ablate defines WN_ABLATE_BUILD; the product library has none of this):
// WN_ABLATE bit 0 skips the d c_up GEMM, bit 1 the d W_cin launch, bit 2 every out-conv launch, bit 3 every d z launch.  Results are
// WRONG by construction; the step time says what that launch class costs the LIVE two-stream step, i.e. the most any fusion / speed-up of
// it could return ("measure it rather than price it").
#ifdef WN_ABLATE_BUILD
static int wn_ablate() { static const int v = [] { const char* e = getenv("WN_ABLATE"); return e ? atoi(e) : 0; }(); return v; }
#else
static constexpr int wn_ablate() { return 0; }
#endif

// ================================================================================================
static void set_dropout(wn_ctx* c, int layer, uint32_t& klo, uint32_t& khi, uint32_t& th, float& ks, int& ld) {
    wn_layer_key(c->fseed, layer, &klo, &khi);
    th = (uint32_t)lrintf(c->cfg.dropout * 65536.0f);
    ks = 1.0f / (1.0f - c->cfg.dropout);
    ld = c->R;
}

static void base_args(wn_ctx* c, GemmArgs& a, const PackedW& w, int b0 = 0, int nb = -1) {
    memset(&a, 0, sizeof a);
    a.Apk = w.dev; a.ksteps_total = w.K >> 4;
    a.nrep = 1; a.rep_stride = 0;
    a.B = nb < 0 ? c->fB : nb; a.b0 = b0; a.T = c->fT; a.zero = c->zero_page;
    a.e.scale = 1.0f; a.e.GH = c->GH; a.e.M_valid = w.M_valid;
}

static SrcSeg seg(const bf16_t* base, int ld, int col0, int nk, int shift, int drop) {
    SrcSeg s; s.base = base; s.ld = ld; s.col0 = col0; s.nk = nk; s.shift = shift; s.dropout = drop; return s;
}

static void prof_mark(wn_ctx* c, hipStream_t st) {
    if (c->pev_used == c->pev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; c->pev.push_back(e); }
    (void)hipEventRecord(c->pev[c->pev_used++], st);
}
__global__ void wn_kprof_init_kernel(unsigned long long* k, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { k[2 * i] = ~0ull; k[2 * i + 1] = 0ull; }
}
// Device timeline from in-kernel stamps (wn_common.h): called at the start of every wn_train_fwd and at the end of every wn_train_bwd.
// Armed by wn_trace_arm (results through wn_trace_read) or by WN_DEVTRACE=<file> (the WN_DEVTRACE_STEP-th forward, written two steps later).
static bool devtrace_begin(wn_ctx* c, hipStream_t st) {
    if (!c->trace_dev && hipMalloc((void**)&c->trace_dev, (size_t)WN_TRACE_MAX * 16) != hipSuccess) return false;
    hipLaunchKernelGGL(wn_kprof_init_kernel, dim3(cdiv(WN_TRACE_MAX, 256)), dim3(256), 0, st, c->trace_dev, WN_TRACE_MAX);
    c->trace_n = 0; c->trace_state = 1;
    return true;
}
__global__ void wn_stamp_kernel(unsigned long long* p) { *p = (unsigned long long)wall_clock64(); }
int wn_trace_scope_begin(wn_ctx* c, hipStream_t st, int kind) {
    if ((c->trace_state != 1 && c->trace_state != 3) || c->trace_n >= WN_TRACE_MAX) return -1;
    const int slot = c->trace_n++;
    c->trace_tag[slot].epi = kind; c->trace_tag[slot].st = (void*)st; c->trace_tag[slot].rows = 0;
    hipLaunchKernelGGL(wn_stamp_kernel, dim3(1), dim3(1), 0, st, c->trace_dev + 2 * slot);
    return slot;
}
void wn_trace_scope_end(wn_ctx* c, hipStream_t st, int slot) { hipLaunchKernelGGL(wn_stamp_kernel, dim3(1), dim3(1), 0, st, c->trace_dev + 2 * slot + 1); }
void wn_devtrace_poll(wn_ctx* c, hipStream_t st, bool step_start) {
    static const char* path = getenv("WN_DEVTRACE");
    static const int at = [] { const char* e = getenv("WN_DEVTRACE_STEP"); return e ? atoi(e) : 8; }();
    if (!step_start) { if (c->trace_state == 1) c->trace_state = 3; return; }      // 3: the step's optimiser may still add its group (wn_optim_impl), then 2
    if (c->trace_state == 3) c->trace_state = 2;                                     // (no optimiser step followed: complete as it is)
    ++c->trace_calls;
    if (c->trace_arm_at && c->trace_calls == c->trace_arm_at) { c->trace_arm_at = 0; devtrace_begin(c, st); return; }
    if (!path) return;
    if (c->trace_calls == at) devtrace_begin(c, st);
    else if (c->trace_calls == at + 2 && c->trace_state == 2) {
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * WN_TRACE_MAX);
        if (hipMemcpy(h.data(), c->trace_dev, 16 * (size_t)c->trace_n, hipMemcpyDeviceToHost) == hipSuccess) {
            FILE* f = fopen(path, "w");
            if (f) {
                fprintf(f, "# idx epi stream rows start_ticks end_ticks (100 MHz)\n");
                for (int i = 0; i < c->trace_n; ++i)
                    fprintf(f, "%d %d %p %d %llu %llu\n", i, c->trace_tag[i].epi, c->trace_tag[i].st, c->trace_tag[i].rows, h[2 * i], h[2 * i + 1]);
                fclose(f);
            }
        }
        c->trace_state = 0;
    }
}
extern "C" int wn_trace_arm(wn_ctx* c, int32_t steps_from_now) {
    if (!c || steps_from_now < 1) return WN_E_ARG;
    if (c->inference) WN_FAIL(c, WN_E_STATE, "wn_trace_arm: no training workspace on an inference-only context");
    c->trace_arm_at = c->trace_calls + steps_from_now; c->trace_state = 0;
    return WN_OK;
}
extern "C" int wn_trace_read(wn_ctx* c, int32_t cap, int32_t* kind, uint64_t* stream, uint64_t* start_ticks, uint64_t* end_ticks) {
    if (!c || cap < 0 || !kind || !stream || !start_ticks || !end_ticks) return WN_E_ARG;
    if (c->trace_state == 3) c->trace_state = 2;
    if (c->trace_state != 2) return 0;
    WN_HIP(c, hipDeviceSynchronize());
    std::vector<unsigned long long> h(2 * (size_t)WN_TRACE_MAX);
    WN_HIP(c, hipMemcpy(h.data(), c->trace_dev, 16 * (size_t)c->trace_n, hipMemcpyDeviceToHost));
    const int n = std::min<int>(cap, c->trace_n);
    for (int i = 0; i < n; ++i) {
        kind[i] = c->trace_tag[i].epi; stream[i] = (uint64_t)(uintptr_t)c->trace_tag[i].st;
        start_ticks[i] = h[2 * i]; end_ticks[i] = h[2 * i + 1];
    }
    return n;
}
extern "C" int wn_profile(wn_ctx* c, int32_t enable) {
    if (!c) return WN_E_ARG;
    c->prof = enable != 0; c->pev_used = 0;
    if (c->prof) {
        if (!c->kprof_dev) WN_HIP(c, hipMalloc((void**)&c->kprof_dev, (size_t)WN_KPROF_MAX * 16));
        hipLaunchKernelGGL(wn_kprof_init_kernel, dim3(cdiv(WN_KPROF_MAX, 256)), dim3(256), 0, 0, c->kprof_dev, WN_KPROF_MAX);
        if (!c->kclk_dev) WN_HIP(c, hipMalloc((void**)&c->kclk_dev, (size_t)WN_KPROF_MAX * 16));
        WN_HIP(c, hipMemset(c->kclk_dev, 0, (size_t)WN_KPROF_MAX * 16));
        WN_HIP(c, hipDeviceSynchronize());
    }
    return WN_OK;
}
// the same launches by their IN-KERNEL stamps (first workgroup's start .. last workgroup's end, 100 MHz wall clock): pure kernel time,
// without the queue / CU-slot wait behind the other stream's kernels that the event bracket of wn_profile_result includes
extern "C" int wn_profile_kernel_result(wn_ctx* c, double* total_ms, int64_t* launches) {
    if (!c || !total_ms || !launches) return WN_E_ARG;
    *total_ms = 0.0; *launches = 0;
    const size_t n = std::min<size_t>(c->pev_used / 2, WN_KPROF_MAX);
    if (!c->kprof_dev || n == 0) return WN_OK;
    for (size_t i = 0; i + 1 < c->pev_used; i += 2) WN_HIP(c, hipEventSynchronize(c->pev[i + 1]));
    std::vector<unsigned long long> h(2 * n);
    WN_HIP(c, hipMemcpy(h.data(), c->kprof_dev, 16 * n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i)
        if (h[2 * i] != ~0ull && h[2 * i + 1] >= h[2 * i]) { *total_ms += (double)(h[2 * i + 1] - h[2 * i]) * 1e-5; ++*launches; }
    return WN_OK;
}
// mean shader clock INSIDE the timed gate launches: workgroup 0 of every launch reads s_memtime (shader cycles) and the 100 MHz wall clock at its
// start and end (GemmArgs::kclk); MHz = 100 x sum(cycles) / sum(ticks).  The step runs at the chip's power limit, so the matrix peak that applies is
// 2500 TFLOP/s x this / 2400 (bench.py: roofline.peak_at_clock).
extern "C" int wn_profile_kernel_clock(wn_ctx* c, double* mhz, int64_t* launches) {
    if (!c || !mhz || !launches) return WN_E_ARG;
    *mhz = 0.0; *launches = 0;
    const size_t n = std::min<size_t>(c->pev_used / 2, WN_KPROF_MAX);
    if (!c->kclk_dev || n == 0) return WN_OK;
    for (size_t i = 0; i + 1 < c->pev_used; i += 2) WN_HIP(c, hipEventSynchronize(c->pev[i + 1]));
    std::vector<unsigned long long> h(2 * n);
    WN_HIP(c, hipMemcpy(h.data(), c->kclk_dev, 16 * n, hipMemcpyDeviceToHost));
    double cyc = 0.0, ticks = 0.0;
    for (size_t i = 0; i < n; ++i)
        if (h[2 * i + 1] > 0 && h[2 * i + 1] < (1ull << 40) && h[2 * i] < (1ull << 48)) { cyc += (double)h[2 * i]; ticks += (double)h[2 * i + 1]; ++*launches; }
    if (ticks > 0.0) *mhz = 100.0 * cyc / ticks;
    return WN_OK;
}
// total milliseconds and number of launches of the dominant kernel (gate GEMM) since wn_profile(ctx, 1); synchronises.
// rows (b*T) one timed gate-GEMM launch processed in the last forward (half the batch when the two-stream split is on)
extern "C" int64_t wn_profile_rows_per_launch(const wn_ctx* c) { return c ? c->prof_rows : 0; }
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

// ---- grouped weight-gradient descriptors (group = layer l0 + g) -----------------------------------
static void wgrad_common(wn_ctx* c, WgBatchArgs& w, int ng, int B, int T) {
    memset(&w, 0, sizeof w); w.ngroups = ng; w.B = B; w.T = T;
}
// multi-A weight-gradient workgroups (wn_wgrad.h): WN_WGRAD_MULTI=0 restores one A tile per workgroup (A/B switch)
static bool wgrad_multi() {
    static const int v = [] { const char* e = getenv("WN_WGRAD_MULTI"); return e ? atoi(e) : 1; }();
    return v != 0;
}
// d W_dil of the layers [l0, l0 + ng): the three taps of one 128-channel block of the layer input share ONE workgroup and ONE staged
// d z tile (A = xd(t-2d) | xd(t-d) | xd(t), B = d z); the conditioning kernel's gradient is its own (single-A) launch, wgrad_cin_args.
static void wgrad_taps_args(wn_ctx* c, WgBatchArgs& w, int l0, int ng, int B, int T) {
    const int64_t NT = c->NT; const int R = c->R, G = c->G;
    wgrad_common(c, w, ng, B, T);
    w.nseg = 3;
    for (int s = 0; s < 3; ++s) { w.seg_base[s] = c->XD + (size_t)l0 * NT * R; w.seg_gstride[s] = NT * R; w.seg_ld[s] = R; w.seg_nk[s] = R; }
    w.Bm = c->DZ + (size_t)l0 * NT * G; w.b_gstride = NT * G; w.ldb = G; w.N = G; w.ldw = G;
    w.na = 3; w.hblocks = R / 128; w.a_colstep = 128;
    for (int x = 0; x < 3; ++x) { w.a_seg[x] = x; w.a_col0[x] = 0; w.a_mrow[x] = x * R; }
    for (int g = 0; g < ng; ++g) {
        const int l = l0 + g, d = c->dil[l];
        WgGroup& q = w.g[g];
        q.out_off = c->lay[l].dil_k; q.bias_off = -1; q.bias2_off = 0; q.has_bias2 = 0;        // (bias gradients: wgrad_cin_args)
        q.shift[0] = -2 * d; q.shift[1] = -d; q.shift[2] = 0; q.shift[3] = 0; q.scale = 1.0f;
    }
}
static void wgrad_cin_args(wn_ctx* c, WgBatchArgs& w, int l0, int ng, int B, int T) {
    const int64_t NT = c->NT; const int G = c->G, C = c->C;
    wgrad_common(c, w, ng, B, T);
    w.nseg = 1; w.seg_base[0] = c->cbt; w.seg_gstride[0] = 0; w.seg_ld[0] = C; w.seg_nk[0] = C;
    w.Bm = c->DZ + (size_t)l0 * NT * G; w.b_gstride = NT * G; w.ldb = G; w.N = G; w.ldw = G;
    for (int g = 0; g < ng; ++g) {
        WgGroup& q = w.g[g];
        const int l = l0 + g;      // d (dil bias) = d (cin bias) = column sums of d z: both written from this launch
        q.out_off = c->lay[l].cin_k; q.bias_off = c->lbias ? c->lay[l].dil_b : -1; q.bias2_off = c->lbias ? c->lay[l].cin_b : 0; q.has_bias2 = c->lbias ? 1 : 0; q.scale = 1.0f;
    }
}
static bool wgrad_taps_ok(wn_ctx* c) { return wgrad_multi() && c->R % 128 == 0 && c->G % 256 == 0; }
static void wgrad_w1_args(wn_ctx* c, WgBatchArgs& w, int l0, int ng, int B, int T) {
    const int64_t NT = c->NT; const int R = c->R, G = c->G, C = c->C;
    wgrad_common(c, w, ng, B, T);
    w.nseg = 4;
    for (int s = 0; s < 3; ++s) { w.seg_base[s] = c->XD + (size_t)l0 * NT * R; w.seg_gstride[s] = NT * R; w.seg_ld[s] = R; w.seg_nk[s] = R; }
    w.seg_base[3] = c->cbt; w.seg_gstride[3] = 0; w.seg_ld[3] = C; w.seg_nk[3] = C;
    w.Bm = c->DZ + (size_t)l0 * NT * G; w.b_gstride = NT * G; w.ldb = G; w.N = G; w.ldw = G;
    for (int g = 0; g < ng; ++g) {
        const int l = l0 + g, d = c->dil[l];
        WgGroup& q = w.g[g];
        q.out_off = c->lay[l].dil_k; q.bias_off = c->lbias ? c->lay[l].dil_b : -1; q.bias2_off = c->lbias ? c->lay[l].cin_b : 0; q.has_bias2 = c->lbias ? 1 : 0;
        q.shift[0] = -2 * d; q.shift[1] = -d; q.shift[2] = 0; q.shift[3] = 0; q.scale = 1.0f;
    }
}
static void wgrad_skip_args(wn_ctx* c, WgBatchArgs& w, int l0, int ng, int B, int T) {
    const int64_t NT = c->NT; const int GH = c->GH, S = c->S;
    wgrad_common(c, w, ng, B, T);
    w.nseg = 1; w.seg_base[0] = c->U + (size_t)l0 * NT * GH; w.seg_gstride[0] = NT * GH; w.seg_ld[0] = GH; w.seg_nk[0] = GH;
    w.Bm = c->DSKIP; w.b_gstride = 0; w.ldb = S; w.N = S; w.ldw = S;
    for (int g = 0; g < ng; ++g) {
        const int l = l0 + g; WgGroup& q = w.g[g];
        q.out_off = c->lay[l].skip_k; q.bias_off = c->lbias ? c->lay[l].skip_b : -1; q.bias2_off = 0; q.has_bias2 = 0; q.scale = c->skip_scale[l];
    }
}
static void wgrad_out_args(wn_ctx* c, WgBatchArgs& w, int l0, int ng, int B, int T) {
    const int64_t NT = c->NT; const int GH = c->GH, R = c->R;
    wgrad_common(c, w, ng, B, T);
    w.nseg = 1; w.seg_base[0] = c->U + (size_t)l0 * NT * GH; w.seg_gstride[0] = NT * GH; w.seg_ld[0] = GH; w.seg_nk[0] = GH;
    w.Bm = c->GXall + (size_t)(l0 + 1) * NT * R; w.b_gstride = NT * R; w.ldb = R; w.N = R; w.ldw = R;
    for (int g = 0; g < ng; ++g) {
        const int l = l0 + g; WgGroup& q = w.g[g];
        q.out_off = c->lay[l].out_k; q.bias_off = c->lbias ? c->lay[l].out_b : -1; q.bias2_off = 0; q.has_bias2 = 0; q.scale = 1.0f;
    }
}
// d W_skip and d W_out share the A operand u_l: one launch, B = [d skip | rho dL/dh_{l+1}] side by side (N = S + R).  The top
// layer's residual branch is dead: GXall[L] is zero-filled, so its W_out gradient comes out as exact zeros.
static void wgrad_skipout_args(wn_ctx* c, WgBatchArgs& w, int l0, int ng, int B, int T) {
    const int64_t NT = c->NT; const int GH = c->GH, S = c->S, R = c->R;
    wgrad_common(c, w, ng, B, T);
    w.nseg = 1; w.seg_base[0] = c->U + (size_t)l0 * NT * GH; w.seg_gstride[0] = NT * GH; w.seg_ld[0] = GH; w.seg_nk[0] = GH;
    w.Bm = c->DSKIP; w.b_gstride = 0; w.ldb = S; w.N = S + R; w.ldw = S;
    w.Bm_hi = c->GXall + (size_t)(l0 + 1) * NT * R; w.b_gstride_hi = NT * R; w.ldb_hi = R; w.split_n = S; w.ldw_hi = R;
    if (wgrad_multi() && GH % 256 == 0) {      // both 128-channel halves of a 256-channel block of u_l share the staged [d skip | d h] tile
        w.na = 2; w.hblocks = GH / 256; w.a_colstep = 256;
        for (int x = 0; x < 2; ++x) { w.a_seg[x] = 0; w.a_col0[x] = 128 * x; w.a_mrow[x] = 128 * x; }
    }
    for (int g = 0; g < ng; ++g) {
        const int l = l0 + g; WgGroup& q = w.g[g];
        q.out_off = c->lay[l].skip_k; q.bias_off = c->lbias ? c->lay[l].skip_b : -1; q.bias2_off = 0; q.has_bias2 = 0; q.scale = c->skip_scale[l];
        q.out_off_hi = c->lay[l].out_k; q.bias_off_hi = c->lbias ? c->lay[l].out_b : -1; q.scale_hi = 1.0f;
    }
}
// Head weight gradients through the same LDS-DMA kernel (round 2 ran them on the round-1 atomics kernel wn_wgrad_kernel: 2 x 140 us
// alone, MFMA busy 3.6 %).  d final_convolution_1 [S][S] = R1^T d pre1 (+ bias = column sums of d pre1): both 128-channel halves of a
// 256-channel block of R1 share the staged d pre1 tile.  d final_convolution_2 [S][O] = H2^T dY has only O (30 / 2 / 256) B columns,
// so the roles are swapped -- A = dY (one 128-wide tile, ldDY valid columns), B = H2 -- and the reduce writes the transpose; its bias
// (column sums of dY) comes from wn_colsum2.  Few, long time slabs: these launches run beside the chain, not alone.
static void wgrad_head1_args(wn_ctx* c, WgBatchArgs& w, int B, int T) {
    const int S = c->S;
    wgrad_common(c, w, 1, B, T);
    w.nseg = 1; w.seg_base[0] = c->R1; w.seg_gstride[0] = 0; w.seg_ld[0] = S; w.seg_nk[0] = S;
    w.Bm = c->DPRE1; w.b_gstride = 0; w.ldb = S; w.N = S; w.ldw = S;
    if (wgrad_multi() && S % 256 == 0) {
        w.na = 2; w.hblocks = S / 256; w.a_colstep = 256;
        for (int x = 0; x < 2; ++x) { w.a_seg[x] = 0; w.a_col0[x] = 128 * x; w.a_mrow[x] = 128 * x; }
    }
    w.spu_cap = 4;
    WgGroup& q = w.g[0];
    q.out_off = c->fin1_k; q.bias_off = c->fin1_b; q.bias2_off = 0; q.has_bias2 = 0; q.scale = 1.0f;
}
static void wgrad_head2_args(wn_ctx* c, WgBatchArgs& w, int B, int T) {
    const int S = c->S, O = c->O, ldDY = (O + 15) / 16 * 16;
    wgrad_common(c, w, 1, B, T);
    w.nseg = 1; w.seg_base[0] = c->DY; w.seg_gstride[0] = 0; w.seg_ld[0] = ldDY; w.seg_nk[0] = ldDY;
    w.Bm = c->H2; w.b_gstride = 0; w.ldb = S; w.N = S; w.ldw = O;
    w.transpose_out = 1; w.m_valid = O; w.spu_cap = 4;
    WgGroup& q = w.g[0];
    q.out_off = c->fin2_k; q.bias_off = -1; q.bias2_off = 0; q.has_bias2 = 0; q.scale = 1.0f;      // (the bias row of this launch sums H2: unused)
}
static bool wgrad_heads_ok(wn_ctx* c) {
    static const int v = [] { const char* e = getenv("WN_WGRAD_HEADS"); return e ? atoi(e) : 1; }();      // A/B switch (0: round-1 kernel)
    if (!v || c->S % 8 != 0 || c->O > 128) return false;
    WgBatchArgs w; wgrad_head1_args(c, w, 1, c->maxT); if (!wn_wgrad_v2_ok(w)) return false;
    wgrad_head2_args(c, w, 1, c->maxT); return wn_wgrad_v2_ok(w);
}
// bytes of split-K partials the grouped launches can need for any batch <= max_batch at max_time
size_t wn_wgrad_partial_need(wn_ctx* c) {
    size_t need = 0;
    for (int ng = 1; ng <= min(WN_MAX_GROUPS, c->L); ++ng)          // any bucket size (wn_plan_buckets) up to all layers at once
    for (int B = 1; B <= c->maxB; ++B) {
        WgBatchArgs w;
        wgrad_w1_args(c, w, 0, ng, B, c->maxT); if (wn_wgrad_v2_ok(w)) { wn_wgrad_plan(w); need = std::max(need, wn_wgrad_partial_bytes(w)); }
        if (wgrad_taps_ok(c)) {
            wgrad_taps_args(c, w, 0, ng, B, c->maxT); if (wn_wgrad_v2_ok(w)) { wn_wgrad_plan(w); need = std::max(need, wn_wgrad_partial_bytes(w)); }
            wgrad_cin_args(c, w, 0, ng, B, c->maxT); if (wn_wgrad_v2_ok(w)) { wn_wgrad_plan(w); need = std::max(need, wn_wgrad_partial_bytes(w)); }
        }
        wgrad_skip_args(c, w, 0, ng, B, c->maxT); if (wn_wgrad_v2_ok(w)) { wn_wgrad_plan(w); need = std::max(need, wn_wgrad_partial_bytes(w)); }
        wgrad_out_args(c, w, 0, ng, B, c->maxT); if (wn_wgrad_v2_ok(w)) { wn_wgrad_plan(w); need = std::max(need, wn_wgrad_partial_bytes(w)); }
        wgrad_skipout_args(c, w, 0, ng, B, c->maxT); if (wn_wgrad_v2_ok(w)) { wn_wgrad_plan(w); need = std::max(need, wn_wgrad_partial_bytes(w)); }
    }
    if (wgrad_heads_ok(c))      // (the head launches share the stream of the stack launches: one buffer serves both)
        for (int B = 1; B <= c->maxB; ++B) {
            WgBatchArgs w;
            wgrad_head1_args(c, w, B, c->maxT); wn_wgrad_plan(w); need = std::max(need, wn_wgrad_partial_bytes(w));
            wgrad_head2_args(c, w, B, c->maxT); wn_wgrad_plan(w); need = std::max(need, wn_wgrad_partial_bytes(w));
        }
    return need + (1 << 20);
}

// ---- batch parts on two streams ---------------------------------------------------------------------------------------
static int parts_setup(wn_ctx* c, int np) {
    if (!c->st2) {
        WN_HIP(c, hipStreamCreateWithFlags(&c->st2, hipStreamNonBlocking));
        WN_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        WN_HIP(c, hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        c->stp[1] = c->st2;
        for (int k = 0; k < WN_MAX_PARTS; ++k) WN_HIP(c, hipEventCreateWithFlags(&c->ev_pjoin[k], hipEventDisableTiming));
    }
    // further part streams only on demand: the runtime multiplexes streams onto a few hardware queues (4 by default), and a stream
    // that merely EXISTS can end up sharing a queue with -- i.e. serialising against -- one of the streams that carry the step
    // (measured: two idle extra streams cost 1.2 ms per step)
    for (int k = 2; k < np; ++k)
        if (!c->stp[k]) WN_HIP(c, hipStreamCreateWithFlags(&c->stp[k], hipStreamNonBlocking));
    return WN_OK;
}
static int n_parts(wn_ctx* c) {
    static const int env = [] { const char* e = getenv("WN_BATCH_PARTS"); return e ? atoi(e) : 2; }();   // 1 disables the overlap (A/B switch)
    const int want = c->parts_req > 0 ? c->parts_req : env;
    return std::max(1, std::min(std::min(want, WN_MAX_PARTS), c->fB));
}
extern "C" int wn_set_batch_parts(wn_ctx* c, int32_t parts) { if (!c || parts < 0 || parts > WN_MAX_PARTS) return WN_E_ARG; c->parts_req = parts; return WN_OK; }

// one residual layer of the utterances [b0, b0 + nb), in two launches (wavenet.py:706-715 / modules.py:471-521):
//   fwd_gate: dropout -> dilated taps -> conditioning 1x1 -> tanh * sigmoid   (MFMA-bound),
//   fwd_out : out 1x1 + residual (+ the dropout-applied copy the next gate stages)   (HBM-bound)
static void mk_gate(wn_ctx* c, int l, int b0, int nb, GemmArgs& a) {
    const int R = c->R, G = c->G, GH = c->GH, C = c->C;
    const int64_t NT = c->NT;
    const int d = c->dil[l];
    const bf16_t* XDl = c->XD + (size_t)l * NT * R;      // dropout already applied by the producer
    base_args(c, a, c->packs[l].w1, b0, nb);
    a.nseg = 4;
    a.seg[0] = seg(XDl, R, 0, R, -2 * d, 0);
    a.seg[1] = seg(XDl, R, 0, R, -d, 0);
    a.seg[2] = seg(XDl, R, 0, R, 0, 0);
    a.seg[3] = seg(c->cbt, C, 0, C, 0, 0);
    a.taps = c->packs[l].w1.kil ? 3 : 0; a.kil = c->packs[l].w1.kil;
    if (c->gin > 0) { a.e.bias = c->gbias + (size_t)l * c->fB * G; a.e.bias_bstride = G; }      // + W_g^T g + b_g per utterance
    else a.e.bias = c->b1sum + (size_t)l * G;
    a.e.out0 = c->TS + (size_t)l * NT * GH; a.e.ld_out0 = GH;      // sigmoid half only (tanh = u / sigmoid in the backward)
    a.e.out1 = c->U + (size_t)l * NT * GH; a.e.ld_out1 = GH;
}
static void mk_out(wn_ctx* c, int l, int b0, int nb, GemmArgs& o) {
    const int R = c->R, GH = c->GH;
    const int64_t NT = c->NT;
    base_args(c, o, c->packs[l].wo, b0, nb);
    o.nseg = 1; o.seg[0] = seg(c->U + (size_t)l * NT * GH, GH, 0, GH, 0, 0);
    o.e.bias = c->params_dev + c->lay[l].out_b;
    o.e.in0 = c->X + (size_t)l * NT * R; o.e.ld_in0 = R;
    o.e.scale = c->res_scale;
    o.e.out0 = c->X + (size_t)(l + 1) * NT * R; o.e.ld_out0 = R;
    if (c->cfg.dropout > 0.0f) {
        o.e.out1 = c->XD + (size_t)(l + 1) * NT * R; o.e.ld_out1 = R;
        set_dropout(c, l + 1, o.key_lo, o.key_hi, o.thresh16, o.keep_scale, o.drop_ld);
    }
}
static void prof_gate(wn_ctx* c, GemmArgs& a, hipStream_t st) {
    a.kprof = (c->kprof_dev && c->pev_used / 2 < WN_KPROF_MAX) ? c->kprof_dev + 2 * (c->pev_used / 2) : nullptr;
    a.kclk = (c->kclk_dev && c->pev_used / 2 < WN_KPROF_MAX) ? c->kclk_dev + 2 * (c->pev_used / 2) : nullptr;
    prof_mark(c, st);
}
static int fwd_gate(wn_ctx* c, int l, int b0, int nb, hipStream_t st, bool prof) {
    GemmArgs a; mk_gate(c, l, b0, nb, a);
    if (prof) prof_gate(c, a, st);
    int rc = a.kil == 64 ? wn_launch_gemm8p<EPI_GATE>(c, a, c->packs[l].w1.M, st) : wn_launch_gemm<EPI_GATE>(c, a, c->packs[l].w1.M, st);
    if (prof) prof_mark(c, st);
    return rc;
}
static int fwd_out(wn_ctx* c, int l, int b0, int nb, hipStream_t st) {
    if (l + 1 >= c->L) return WN_OK;      // the residual output of the last layer is never consumed (wavenet.py:716)
    if (wn_ablate() & 4) return WN_OK;
    GemmArgs o; mk_out(c, l, b0, nb, o);
    return wn_launch_gemm<EPI_STORE_BF16>(c, o, c->packs[l].wo.M, st);
}
// skip sum + head of the utterances [b0, b0 + nb) (wavenet.py:716-721)
static int fwd_tail(wn_ctx* c, int b0, int nb, hipStream_t st) {
    const int L = c->L, GH = c->GH, S = c->S;
    const int64_t NT = c->NT;
    int rc;
    {   // skip sum over all layers as one contraction, + ReLU (wavenet.py:716-719 first activation)
        GemmArgs a; base_args(c, a, c->wskip, b0, nb);
        a.nseg = 1; a.seg[0] = seg(c->U, GH, 0, GH, 0, 0); a.nrep = L; a.rep_stride = NT * GH;
        a.e.bias = c->skip_bias_total; a.e.relu = 1; a.e.out0 = c->R1; a.e.ld_out0 = S;
        if ((rc = wn_launch_gemm<EPI_STORE_BF16>(c, a, c->wskip.M, st))) return rc;
    }
    {   // final_convolution_1 + ReLU
        GemmArgs a; base_args(c, a, c->wh1, b0, nb);
        a.nseg = 1; a.seg[0] = seg(c->R1, S, 0, S, 0, 0);
        a.e.bias = c->params_dev + c->fin1_b; a.e.relu = 1; a.e.out0 = c->H2; a.e.ld_out0 = S;
        if ((rc = wn_launch_gemm<EPI_STORE_BF16>(c, a, c->wh1.M, st))) return rc;
    }
    {   // final_convolution_2 -> y_hat [B,O,T] fp32
        GemmArgs a; base_args(c, a, c->wh2, b0, nb);
        a.nseg = 1; a.seg[0] = seg(c->H2, S, 0, S, 0, 0);
        a.e.bias = c->params_dev + c->fin2_b; a.e.out0 = c->YHAT; a.e.M_valid = c->O;
        if ((rc = wn_launch_gemm<EPI_STORE_F32_BOT>(c, a, c->wh2.M, st))) return rc;
    }
    return WN_OK;
}
// layers + skip sum + head of the utterances [b0, b0 + nb) on stream st (wavenet.py:706-721)
static int fwd_part(wn_ctx* c, int b0, int nb, hipStream_t st, bool prof) {
    int rc;
    for (int l = 0; l < c->L; ++l) {
        if ((rc = fwd_gate(c, l, b0, nb, st, prof))) return rc;
        if ((rc = fwd_out(c, l, b0, nb, st))) return rc;
    }
    return fwd_tail(c, b0, nb, st);
}

// Two ways of FORCING the complementary pairing of the two half-batches (MFMA-bound gate / d x of one beside HBM-bound out conv / d z of
// the other) were built and measured in round 3 and are gone from the library: a lockstep chain through cross-stream events (every event
// costs 15-24 us between the signalling kernel's end and the waiting kernel's start: 11.6 vs 10.06 ms/step, profiles/r4d_ab_lockstep.txt)
// and both launches in ONE grid (wn_fused_pair_kernel, in the repository history up to round 5: every grid drains before the next starts, 11.2 vs
// 9.9 ms/step, profiles/r4e_ab_fused.txt).  Two free-running streams hide each kernel's tail under the other stream's next launch.

// run f(b0, nb, stream, is_first_part) for every batch part: part 0 on the caller's stream, part 1 on the ctx-owned one, both
// ordered after everything already enqueued on `st`, and `st` ordered after both when this returns
template <class F> static int for_each_part(wn_ctx* c, hipStream_t st, F f) {
    const int np = n_parts(c);
    c->parts = np;
    if (np == 1) return f(0, c->fB, st, true, 0);
    int rc = parts_setup(c, np);
    if (rc) return rc;
    // part k takes the utterances [k * fB / np, (k + 1) * fB / np): parts 1.. on the ctx-owned streams, part 0 on the caller's
    WN_HIP(c, hipEventRecord(c->ev_fork, st));
    for (int k = np - 1; k >= 0; --k) {
        const int b0 = (int)((int64_t)k * c->fB / np), b1 = (int)((int64_t)(k + 1) * c->fB / np);
        if (k > 0) {
            WN_HIP(c, hipStreamWaitEvent(c->stp[k], c->ev_fork, 0));
            if ((rc = f(b0, b1 - b0, c->stp[k], false, k))) return rc;
            WN_HIP(c, hipEventRecord(c->ev_pjoin[k], c->stp[k]));
        } else if ((rc = f(b0, b1 - b0, st, true, 0))) return rc;
    }
    for (int k = 1; k < np; ++k) WN_HIP(c, hipStreamWaitEvent(st, c->ev_pjoin[k], 0));
    return WN_OK;
}

int wn_fwd_impl(wn_ctx* c, hipStream_t st, float* loss_out, float* y_hat_out) {
    int rc;
    wn_devtrace_poll(c, st, true);
    if ((rc = wn_upsample_fwd(c, nullptr, c->fc, c->fB, c->fTc, st))) return rc;     // wavenet.py:680-702
    if ((rc = wn_first_conv(c, st))) return rc;                                      // wavenet.py:705
    if ((rc = wn_gbias_fwd(c, c->fB, st))) return rc;                                // wavenet.py:669-678
    c->fwd_was_f32 = (c->cfg.compute_dtype == WN_COMPUTE_F32);
    if (c->fwd_was_f32) {      // the reference's fp32 arithmetic for y_hat / the loss value (wn_f32.hip); no saved activations for a backward
        if ((rc = wn_f32_forward(c, st))) return rc;
        if (y_hat_out) WN_HIP(c, hipMemcpyAsync(y_hat_out, c->YHAT, (size_t)c->fB * c->O * c->fT * 4, hipMemcpyDeviceToDevice, st));
        if (loss_out) {      // the loss kernel also writes d y_hat in fp32 for wn_f32_backward
            c->dy32_next = wn_f32_dy(c);
            if (!c->dy32_next) WN_FAIL(c, WN_E_HIP, "hipMalloc(fp32 d y_hat) failed");
            if ((rc = wn_loss_fwd_bwd(c, loss_out, st))) return rc;
        }
        c->have_loss = loss_out != nullptr;
        return WN_OK;
    }
    rc = for_each_part(c, st, [&](int b0, int nb, hipStream_t s, bool first, int) {
        if (first) c->prof_rows = nb * c->fT;      // rows of one timed gate-GEMM launch (wn_profile_result)
        return fwd_part(c, b0, nb, s, c->prof && first);
    });
    if (rc) return rc;
    if (y_hat_out) WN_HIP(c, hipMemcpyAsync(y_hat_out, c->YHAT, (size_t)c->fB * c->O * c->fT * 4, hipMemcpyDeviceToDevice, st));
    if (loss_out) { if ((rc = wn_loss_fwd_bwd(c, loss_out, st))) return rc; c->have_loss = true; }
    else c->have_loss = false;
    return WN_OK;
}

// backward serial chain of the utterances [b0, b0 + nb): head dgrads, then d z / d h of every layer, top to bottom.
// GXall[l] = rho * dL/dh_l is kept for every layer (rho = sqrt(.5) if residual_legacy), so that all weight gradients can be
// contracted afterwards over the whole batch.
static int bwd_head(wn_ctx* c, int b0, int nb, hipStream_t st, int part) {
    const int S = c->S, O = c->O;
    const int ldDY = (O + 15) / 16 * 16;
    int rc;
    {   // d pre1 = (W2 dY) * (H2 > 0)
        GemmArgs a; base_args(c, a, c->wh2T, b0, nb);
        a.nseg = 1; a.seg[0] = seg(c->DY, ldDY, 0, c->wh2T.K, 0, 0);
        a.e.in0 = c->H2; a.e.ld_in0 = S; a.e.out0 = c->DPRE1; a.e.ld_out0 = S;
        if ((rc = wn_launch_gemm<EPI_MASK_STORE>(c, a, c->wh2T.M, st))) return rc;
        WN_HIP(c, hipEventRecord(c->ev_head[part], st));
    }
    {   // d skip = (W1 dpre1) * (skips > 0)
        GemmArgs a; base_args(c, a, c->wh1T, b0, nb);
        a.nseg = 1; a.seg[0] = seg(c->DPRE1, S, 0, S, 0, 0);
        a.e.in0 = c->R1; a.e.ld_in0 = S; a.e.out0 = c->DSKIP; a.e.ld_out0 = S;
        if ((rc = wn_launch_gemm<EPI_MASK_STORE>(c, a, c->wh1T.M, st))) return rc;
    }
    return WN_OK;
}
// d z of layer l: through the 1x1 convs and the gate (modules.py:510-515)   (HBM-bound)
static void mk_dgate(wn_ctx* c, int l, int b0, int nb, GemmArgs& a) {
    const int R = c->R, G = c->G, S = c->S;
    const int64_t NT = c->NT;
    base_args(c, a, c->packs[l].w2T, b0, nb);
    a.nseg = 2;
    a.seg[0] = seg(c->GXall + (size_t)(l + 1) * NT * R, R, 0, R, 0, 0);
    a.seg[1] = seg(c->DSKIP, S, 0, S, 0, 0);
    a.e.in0 = c->TS + (size_t)l * NT * (G / 2); a.e.in1 = c->U + (size_t)l * NT * (G / 2); a.e.ld_in0 = G / 2; a.e.out0 = c->DZ + (size_t)l * NT * G; a.e.ld_out0 = G;
}
// d h_l = dropout-mask * conv^T(dz) + residual path   (modules.py:484, 517-520)   (MFMA-bound)
static void mk_dx(wn_ctx* c, int l, int b0, int nb, GemmArgs& a) {
    const int R = c->R, G = c->G;
    const int64_t NT = c->NT;
    const int d = c->dil[l];
    bf16_t* DZl = c->DZ + (size_t)l * NT * G;
    const bool top = (l == c->L - 1);
    base_args(c, a, c->packs[l].w1T, b0, nb);
    a.nseg = 3;
    a.seg[0] = seg(DZl, G, 0, G, 2 * d, 0);
    a.seg[1] = seg(DZl, G, 0, G, d, 0);
    a.seg[2] = seg(DZl, G, 0, G, 0, 0);
    a.taps = c->packs[l].w1T.kil ? 3 : 0; a.kil = c->packs[l].w1T.kil;
    set_dropout(c, l, a.key_lo, a.key_hi, a.thresh16, a.keep_scale, a.drop_ld);
    if (!(c->cfg.dropout > 0.0f)) a.thresh16 = 0;
    a.e.in0 = top ? nullptr : c->GXall + (size_t)(l + 1) * NT * R; a.e.ld_in0 = R;
    a.e.scale = (l > 0) ? c->res_scale : 1.0f;
    a.e.out0 = c->GXall + (size_t)l * NT * R; a.e.ld_out0 = R;
}
static int bwd_dgate(wn_ctx* c, int l, int b0, int nb, hipStream_t st) {
    if (wn_ablate() & 8) return WN_OK;
    GemmArgs a; mk_dgate(c, l, b0, nb, a);
    return wn_launch_gemm<EPI_DGATE>(c, a, c->packs[l].w2T.M, st);
}
// d z / d h of the layers [l, L) exist for this batch part: the weight gradients of a bucket whose lowest layer is l may start
static int chain_events(wn_ctx* c, int l, int part, hipStream_t st) {
    for (int k = 0; k < c->nbuckets_early; ++k)
        if (c->bucket_lo[k] == l) WN_HIP(c, hipEventRecord(c->ev_chain[part][k], st));
    return WN_OK;
}
static int bwd_dx(wn_ctx* c, int l, int b0, int nb, hipStream_t st, int part) {
    GemmArgs a; mk_dx(c, l, b0, nb, a);
    int rc = a.kil == 64 ? wn_launch_gemm8p<EPI_DX>(c, a, c->packs[l].w1T.M, st) : wn_launch_gemm<EPI_DX>(c, a, c->packs[l].w1T.M, st);
    if (rc) return rc;
    return chain_events(c, l, part, st);
}
static int bwd_part(wn_ctx* c, int b0, int nb, hipStream_t st, int part) {
    int rc;
    if ((rc = bwd_head(c, b0, nb, st, part))) return rc;
    for (int l = c->L - 1; l >= 0; --l) {
        if ((rc = bwd_dgate(c, l, b0, nb, st))) return rc;
        if ((rc = bwd_dx(c, l, b0, nb, st, part))) return rc;
    }
    return WN_OK;
}
static int wn_bwd_eff(wn_ctx* c, float* grads, hipStream_t st);
int wn_bwd_impl(wn_ctx* c, float* grads, hipStream_t st) {
    if (!c->wnorm) return wn_bwd_eff(c, grads, st);
    // weight normalisation: gradients w.r.t. the effective kernels go to a ctx-owned buffer, then d v / d g (modules.py:98-103)
    int rc = wn_bwd_eff(c, c->deff, st);
    if (rc) return rc;
    if ((rc = wn_weightnorm_grad(c, grads, st))) return rc;
    WN_HIP(c, hipEventRecord(c->ev_bucket[WN_MAX_BUCKETS], st));          // (single bucket: final after the v / g mapping)
    return WN_OK;
}
// Gradient buckets (data-parallel overlap, replaces the tower loop of wavenet.py:553-581): the flat gradient buffer is completed in
// `nbuckets` contiguous pieces, top layers first.  The weight gradients of bucket k (MFMA-bound grouped launches) run on a third,
// low-priority stream as soon as the serial d z / d h chain has passed the bucket's lowest layer, i.e. UNDER the rest of the chain
// (whose launches leave CUs idle in their tail rounds), and an event per bucket lets the caller start that bucket's all-reduce
// while the next one is still being computed (wn_bwd_wait_bucket).
static int buckets_setup(wn_ctx* c) {
    if (c->st3) return WN_OK;
    int lo_pri = 0, hi_pri = 0;
    WN_HIP(c, hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));            // (least, greatest)
    WN_HIP(c, hipStreamCreateWithPriority(&c->st3, hipStreamNonBlocking, lo_pri));
    for (int p = 0; p < WN_MAX_PARTS; ++p)
        for (int k = 0; k < WN_MAX_BUCKETS; ++k) WN_HIP(c, hipEventCreateWithFlags(&c->ev_chain[p][k], hipEventDisableTiming));
    for (int p = 0; p < WN_MAX_PARTS; ++p) WN_HIP(c, hipEventCreateWithFlags(&c->ev_head[p], hipEventDisableTiming));
    for (int k = 0; k < WN_MAX_BUCKETS + 2; ++k) WN_HIP(c, hipEventCreateWithFlags(&c->ev_bucket[k], hipEventDisableTiming));
    WN_HIP(c, hipEventCreateWithFlags(&c->ev_w0, hipEventDisableTiming));
    return WN_OK;
}
// bucket table (fixed at wn_create): early buckets = layer groups from the top, then [input conv + lowest layers], then the tail
// (embedding table, upsample net).  Offsets are in the caller's (raw) layout.
void wn_plan_buckets(wn_ctx* c) {
    const int L = c->L;
    c->nbuckets_early = 0; c->nbuckets = 0;
    const int64_t tail0 = (c->emb_off >= 0) ? c->emb_off : (c->up_k.empty() ? c->n_params : c->up_k[0]);
    static const int env_want = [] { const char* e = getenv("WN_BWD_BUCKETS"); return e ? atoi(e) : 0; }();      // A/B override
    const int want = env_want > 0 ? env_want : (c->cfg.grad_buckets > WN_MAX_BUCKETS ? WN_MAX_BUCKETS : c->cfg.grad_buckets);
    const bool early = !c->wnorm && c->gin == 0 && want > 1 && L >= 2 * want;
    if (!early) {      // weight normalisation maps the effective gradients to (v, g) in a final pass; global conditioning adds per-layer
        c->bucket_off[0] = 0; c->bucket_cnt[0] = c->n_raw; c->nbuckets = 1;          // tensors late: ONE bucket, final when the call ends
        return;
    }
    // `want` pieces of (almost) equal depth from the top; L >= 2 * want, so per >= 2 and the last (lowest) piece keeps >= per layers.
    // Pieces deeper than one grouped launch can address (WN_MAX_GROUPS layers) are chunked in stack_wgrads.
    const int per = L / want;
    int hi = L;
    for (int k = 0; k < want - 1; ++k) {
        const int lo = hi - per;
        if (lo <= 0) break;
        c->bucket_lo[k] = lo; c->bucket_hi[k] = hi;
        c->bucket_off[k] = c->lay[lo].dil_k;
        c->bucket_cnt[k] = (k == 0 ? tail0 : c->lay[hi].dil_k) - c->lay[lo].dil_k;      // bucket 0 also carries the head (final_convolution_*)
        hi = lo; ++c->nbuckets_early;
    }
    c->nbuckets = c->nbuckets_early;
    c->bucket_lo[c->nbuckets] = 0; c->bucket_hi[c->nbuckets] = hi;
    c->bucket_off[c->nbuckets] = 0; c->bucket_cnt[c->nbuckets] = c->lay[hi].dil_k; ++c->nbuckets;       // input conv + layers [0, hi)
    if (tail0 < c->n_params) { c->bucket_off[c->nbuckets] = tail0; c->bucket_cnt[c->nbuckets] = c->n_params - tail0; ++c->nbuckets; }
}
extern "C" int wn_bwd_num_buckets(const wn_ctx* c) { return c ? c->nbuckets : WN_E_ARG; }
extern "C" int wn_bwd_bucket_range(const wn_ctx* c, int32_t i, int64_t* offset, int64_t* count) {
    if (!c || i < 0 || i >= c->nbuckets || !offset || !count) return WN_E_ARG;
    *offset = c->bucket_off[i]; *count = c->bucket_cnt[i];
    return WN_OK;
}
extern "C" int wn_bwd_wait_bucket(wn_ctx* c, int32_t i, void* stream) {
    if (!c || i < 0 || i >= c->nbuckets) return WN_E_ARG;
    if (!c->have_bwd) WN_FAIL(c, WN_E_STATE, "wn_bwd_wait_bucket: no wn_train_bwd has been enqueued");
    // buckets whose weight gradients actually ran early (under the chain) in the LAST backward have their own event; every other
    // piece -- incl. the early pieces of a model that took the per-layer kernels (narrow channel counts) -- is final with the call
    WN_HIP(c, hipStreamWaitEvent((hipStream_t)stream, c->ev_bucket[i < c->nearly_live ? i : WN_MAX_BUCKETS], 0));
    return WN_OK;
}

// stack weight gradients of the layers [l0, l0 + ng) on stream st: d [W_dil; W_cin] (+ biases), d W_skip, d W_out (+ biases)
static int stack_wgrads(wn_ctx* c, float* grads, int l0, int ng, bool fused, hipStream_t st) {
    const int L = c->L;
    int rc;
    if (wgrad_taps_ok(c)) {
        {   // d W_dil, d (dil + cin) biases: the three taps of a channel block in one workgroup, sharing the d z tile
            WgBatchArgs w; wgrad_taps_args(c, w, l0, ng, c->fB, c->fT); w.grads = grads;
            if ((rc = launch_wgrad_batch(c, w, st))) return rc;
        }
        if (!(wn_ablate() & 2)) {   // d W_cin:  A = c(t),  B = d z
            WgBatchArgs w; wgrad_cin_args(c, w, l0, ng, c->fB, c->fT); w.grads = grads;
            if ((rc = launch_wgrad_batch(c, w, st))) return rc;
        }
    } else {   // d [W_dil; W_cin], d biases:  A = [xd(t-2d) | xd(t-d) | xd(t) | c(t)],  B = d z
        WgBatchArgs w; wgrad_w1_args(c, w, l0, ng, c->fB, c->fT); w.grads = grads;
        if ((rc = launch_wgrad_batch(c, w, st))) return rc;
    }
    if (fused) {
        // d W_skip (scaled by the legacy factor c_l) and d W_out with their biases in ONE launch: A = u_l, B = [d skip | rho dL/dh_{l+1}]
        WgBatchArgs w; wgrad_skipout_args(c, w, l0, ng, c->fB, c->fT); w.grads = grads;
        if ((rc = launch_wgrad_batch(c, w, st))) return rc;
    } else {
        {   // d W_skip (scaled by the legacy factor c_l), d skip bias:  A = u_l,  B = d skip (shared by all layers)
            WgBatchArgs w; wgrad_skip_args(c, w, l0, ng, c->fB, c->fT); w.grads = grads;
            if ((rc = launch_wgrad_batch(c, w, st))) return rc;
        }
        const int ngo = min(ng, L - 1 - l0);      // the top layer's residual branch is dead: zero gradient
        if (ngo > 0) {   // d W_out, d out bias:  A = u_l,  B = rho * dL/dh_{l+1}
            WgBatchArgs w; wgrad_out_args(c, w, l0, ngo, c->fB, c->fT); w.grads = grads;
            if ((rc = launch_wgrad_batch(c, w, st))) return rc;
        }
    }
    return WN_OK;
}

static int wn_bwd_eff(wn_ctx* c, float* grads, hipStream_t st) {
    if (!c->have_loss) WN_FAIL(c, WN_E_STATE, "wn_train_bwd needs a forward that computed the loss (loss_out != NULL)");
    const int L = c->L, R = c->R, G = c->G, GH = c->GH, S = c->S, C = c->C, O = c->O;
    const int64_t NT = c->NT;
    const int ldDY = (O + 15) / 16 * 16;
    int rc;
    if ((rc = buckets_setup(c))) return rc;
    WN_HIP(c, hipMemsetAsync(grads, 0, (size_t)c->n_params * 4, st));
    if (c->fwd_was_f32) {      // fp32 training mode: the whole backward in the reference's arithmetic, on the caller's stream (wn_f32.hip)
        if ((rc = wn_f32_backward(c, grads, st))) return rc;
        c->nearly_live = 0;
        WN_HIP(c, hipEventRecord(c->ev_bucket[WN_MAX_BUCKETS], st));          // every bucket of the table is final here
        c->have_bwd = true;
        return WN_OK;
    }
    const int64_t rows = (int64_t)c->fB * c->fT;
    WN_HIP(c, hipMemsetAsync(c->GXall + (size_t)L * NT * R, 0, (size_t)rows * R * 2, st));   // top layer: residual branch is dead
    bool grouped, fused;
    { WgBatchArgs w; wgrad_w1_args(c, w, 0, 1, c->fB, c->fT); grouped = wn_wgrad_v2_ok(w);
      wgrad_skipout_args(c, w, 0, 1, c->fB, c->fT); fused = wn_wgrad_v2_ok(w) && c->S % 8 == 0;     // N = S + R (256 for the 128-channel default hparams)
      if (!fused) { wgrad_skip_args(c, w, 0, 1, c->fB, c->fT); grouped = grouped && wn_wgrad_v2_ok(w);
                    wgrad_out_args(c, w, 0, 1, c->fB, c->fT); grouped = grouped && wn_wgrad_v2_ok(w); } }
    const int nearly = grouped ? c->nbuckets_early : 0;       // (per-layer v1 kernels: everything after the chain, as one piece)
    c->nearly_live = nearly;                                  // wn_bwd_wait_bucket: which pieces have their own event in THIS backward
    // ---- weight-gradient stream: head weight gradients first (their operands exist since the forward / loss), then one bucket of
    // stack weight gradients whenever both chain streams have passed its lowest layer
    WN_HIP(c, hipEventRecord(c->ev_w0, st));
    static const bool serial = getenv("WN_SERIAL") != nullptr;      // profiling aid: every kernel on the caller's stream (with WN_BATCH_PARTS=1: exclusive kernel times)
    hipStream_t wst = serial ? st : c->st3;
    WN_HIP(c, hipStreamWaitEvent(wst, c->ev_w0, 0));
    const bool heads_v2 = wgrad_heads_ok(c);
    if (heads_v2) {   // d final_convolution_2 = H2^T dY: both operands exist since the forward / loss -- starts now, under the chain (low priority)
        WgBatchArgs w; wgrad_head2_args(c, w, c->fB, c->fT); w.grads = grads;
        if ((rc = launch_wgrad_batch(c, w, wst))) return rc;
        if ((rc = wn_colsum2(c, c->DY, ldDY, ldDY, O, nullptr, rows, grads + c->fin2_b, nullptr, 1, wst))) return rc;
    } else {
        WgArgs w; memset(&w, 0, sizeof w);
        w.nseg = 1; w.seg[0] = seg(c->H2, S, 0, S, 0, 0); w.ones_row = 1;
        w.Bm = c->DY; w.ldb = ldDY; w.colb0 = 0; w.N = O;
        w.out = grads + c->fin2_k; w.ldw = O; w.bias_out = grads + c->fin2_b; w.scale = 1.0f; w.B = c->fB; w.T = c->fT;
        if ((rc = launch_wgrad(c, w, wst))) return rc;
    }
    // ---- the serial chain, per batch part (two streams)
    if ((rc = for_each_part(c, st, [&](int b0, int nb, hipStream_t s, bool, int part) { return bwd_part(c, b0, nb, s, part); }))) return rc;
    // ---- head weight gradients over the whole batch (wavenet.py:136-149): d final_convolution_1 = R1^T dpre1 needs d pre1 of every
    // part, the first thing each chain stream computes (enqueued after the chain in host order, gated only by those events)
    for (int pk = 0; pk < c->parts; ++pk) WN_HIP(c, hipStreamWaitEvent(wst, c->ev_head[pk], 0));
    if (heads_v2) {
        WgBatchArgs w; wgrad_head1_args(c, w, c->fB, c->fT); w.grads = grads;
        if ((rc = launch_wgrad_batch(c, w, wst))) return rc;
    } else {
        WgArgs w; memset(&w, 0, sizeof w);
        w.nseg = 1; w.seg[0] = seg(c->R1, S, 0, S, 0, 0); w.ones_row = 1;
        w.Bm = c->DPRE1; w.ldb = S; w.N = S;
        w.out = grads + c->fin1_k; w.ldw = S; w.bias_out = grads + c->fin1_b; w.scale = 1.0f; w.B = c->fB; w.T = c->fT;
        if ((rc = launch_wgrad(c, w, wst))) return rc;
    }
    {   // the stack weight gradients below need the chain (all of it, or up to the first bucket's lowest layer)
        if (nearly > 0) {
            for (int pk = 0; pk < c->parts; ++pk) WN_HIP(c, hipStreamWaitEvent(wst, c->ev_chain[pk][0], 0));
        } else {
            WN_HIP(c, hipEventRecord(c->ev_w0, st));
            WN_HIP(c, hipStreamWaitEvent(wst, c->ev_w0, 0));
        }
    }
    // ---- weight gradients of the stack, each kind for all layers of a bucket in one grouped launch (wn_wgrad.h)
    for (int k = 0; k < nearly; ++k) {
        if (k > 0) {
            for (int pk = 0; pk < c->parts; ++pk) WN_HIP(c, hipStreamWaitEvent(wst, c->ev_chain[pk][k], 0));
        }
        for (int l0 = c->bucket_lo[k]; l0 < c->bucket_hi[k]; l0 += WN_MAX_GROUPS)
            if ((rc = stack_wgrads(c, grads, l0, min(WN_MAX_GROUPS, c->bucket_hi[k] - l0), fused, wst))) return rc;
        WN_HIP(c, hipEventRecord(c->ev_bucket[k], wst));
    }
    // ---- after the chain: the lowest layers' weight gradients (weight-gradient stream), and on the second stream everything small
    // or latency-bound (global-conditioning / input-conv / upsample-net gradients, the d c_up GEMM); disjoint regions of `grads`
    WN_HIP(c, hipEventRecord(c->ev_w0, st));
    WN_HIP(c, hipStreamWaitEvent(wst, c->ev_w0, 0));
    hipStream_t side = st;
    if (c->parts >= 2) {
        WN_HIP(c, hipEventRecord(c->ev_fork, st));
        WN_HIP(c, hipStreamWaitEvent(c->st2, c->ev_fork, 0));
        side = c->st2;
    }
    if (grouped) {
        const int hi = nearly > 0 ? c->bucket_lo[nearly - 1] : L;
        for (int l0 = 0; l0 < hi; l0 += WN_MAX_GROUPS)
            if ((rc = stack_wgrads(c, grads, l0, min(WN_MAX_GROUPS, hi - l0), fused, wst))) return rc;
    }
    for (int l = L - 1; !grouped && l >= 0; --l) {      // narrow channel counts (N % 256 != 0): per-layer v1 kernels
        const int d = c->dil[l];
        const bf16_t* Ul = c->U + (size_t)l * NT * GH;
        const bf16_t* XDl = c->XD + (size_t)l * NT * R;
        const bf16_t* gxu = c->GXall + (size_t)(l + 1) * NT * R;
        {
            WgArgs w; memset(&w, 0, sizeof w);
            w.nseg = 4;
            w.seg[0] = seg(XDl, R, 0, R, -2 * d, 0); w.seg[1] = seg(XDl, R, 0, R, -d, 0); w.seg[2] = seg(XDl, R, 0, R, 0, 0);
            w.seg[3] = seg(c->cbt, C, 0, C, 0, 0);
            w.ones_row = 1;
            w.Bm = c->DZ + (size_t)l * NT * G; w.ldb = G; w.N = G;
            w.out = grads + c->lay[l].dil_k; w.ldw = G; w.bias_out = c->lbias ? grads + c->lay[l].dil_b : nullptr; w.bias_out2 = c->lbias ? grads + c->lay[l].cin_b : nullptr;
            w.scale = 1.0f; w.B = c->fB; w.T = c->fT;
            if ((rc = launch_wgrad(c, w, wst))) return rc;
        }
        {
            WgArgs w; memset(&w, 0, sizeof w);
            w.nseg = 1; w.seg[0] = seg(Ul, GH, 0, GH, 0, 0); w.ones_row = 1;
            w.Bm = c->DSKIP; w.ldb = S; w.N = S;
            w.out = grads + c->lay[l].skip_k; w.ldw = S; w.bias_out = c->lbias ? grads + c->lay[l].skip_b : nullptr;
            w.scale = c->skip_scale[l]; w.B = c->fB; w.T = c->fT;
            if ((rc = launch_wgrad(c, w, wst))) return rc;
        }
        if (l != L - 1) {
            WgArgs w; memset(&w, 0, sizeof w);
            w.nseg = 1; w.seg[0] = seg(Ul, GH, 0, GH, 0, 0); w.ones_row = 1;
            w.Bm = gxu; w.ldb = R; w.N = R;
            w.out = grads + c->lay[l].out_k; w.ldw = R; w.bias_out = c->lbias ? grads + c->lay[l].out_b : nullptr;
            w.scale = 1.0f; w.B = c->fB; w.T = c->fT;
            if ((rc = launch_wgrad(c, w, wst))) return rc;
        }
    }
    if ((rc = wn_gin_bwd(c, grads, side))) return rc;     // d W_g, d b_g, d embedding table (modules.py:499-508)
    const bf16_t* gx_up = c->GXall;
    // gx_up now holds dL/dh_0
    if ((rc = wn_first_conv_grad(c, gx_up, grads, side))) return rc;
    if (c->cfg.upsample_type != WN_UP_NEAREST) {
        // d c_up[b][cc][t] = sum_l W_cin_l dz_l  as one contraction over all layers
        GemmArgs a; base_args(c, a, c->wcT);
        a.nseg = 1; a.seg[0] = seg(c->DZ, G, 0, G, 0, 0); a.nrep = L; a.rep_stride = NT * G;
        a.e.out0 = c->DC; a.e.M_valid = C;
        if (!(wn_ablate() & 1)) if ((rc = wn_launch_gemm<EPI_STORE_F32_BOT>(c, a, c->wcT.M, side))) return rc;
        if ((rc = wn_upsample_bwd(c, c->DC, grads, side))) return rc;
    }
    if (side != st) {
        WN_HIP(c, hipEventRecord(c->ev_join, side));
        WN_HIP(c, hipStreamWaitEvent(st, c->ev_join, 0));
    }
    WN_HIP(c, hipEventRecord(c->ev_w0, wst));
    WN_HIP(c, hipStreamWaitEvent(st, c->ev_w0, 0));
    WN_HIP(c, hipEventRecord(c->ev_bucket[WN_MAX_BUCKETS], st));          // everything (incl. the late buckets) is final here
    c->have_bwd = true;
    wn_devtrace_poll(c, st, false);
    return WN_OK;
}
