// Stand-alone check + timing of the weight-gradient kernels (run on the GPU box): v1 (per layer, ds_read_u16 gathers,
// atomics) vs v2 (grouped over layers, LDS-DMA ring + ds_read_b64_tr_b16, partial tiles + reduce).
// Same math, different summation order => compare with a tolerance.
#include "wn_wgrad.h"
#include <vector>
#include <random>
#include <functional>
std::string g_create_err;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
static std::mt19937 rng(99);
static bf16_t* dev_bf16_random(size_t n, float scale) {
    std::vector<bf16_t> h(n); std::uniform_real_distribution<float> d(-scale, scale);
    for (auto& v : h) v = f2bf(d(rng));
    bf16_t* p; CK(hipMalloc(&p, n * 2)); CK(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice)); return p;
}
static float time_ms(const std::function<void()>& f, int iters = 5) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}
static double rel_err(const float* a, const float* b, size_t n) {
    double num = 0, den = 0; for (size_t i = 0; i < n; ++i) { double d = (double)a[i] - b[i]; num += d * d; den += (double)b[i] * b[i]; }
    return sqrt(num / (den + 1e-30));
}
int main(int argc, char** argv) {
    int fails = 0;
    wn_ctx ctx;
    CK(hipMalloc(&ctx.zero_page, 256)); CK(hipMemset(ctx.zero_page, 0, 256));
    ctx.wg_partial_bytes = (size_t)1 << 30; CK(hipMalloc(&ctx.wg_partial, ctx.wg_partial_bytes));
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int B = cfg == 0 ? 8 : 3, T = cfg == 0 ? 11000 : 1111, L = cfg == 0 ? 6 : 3;
        const int R = 256, G = 512, GH = 256, S = 256, C = 80;
        const int64_t NT_ = (int64_t)B * T;
        bf16_t* XD = dev_bf16_random((size_t)L * NT_ * R, 1.0f); bf16_t* cbt = dev_bf16_random(NT_ * C, 1.0f);
        bf16_t* DZ = dev_bf16_random((size_t)L * NT_ * G, 1.0f); bf16_t* U = dev_bf16_random((size_t)L * NT_ * GH, 1.0f); bf16_t* DS = dev_bf16_random(NT_ * S, 1.0f);
        auto mkseg = [](const bf16_t* b, int ld, int col0, int nk, int shift) { SrcSeg s; s.base = b; s.ld = ld; s.col0 = col0; s.nk = nk; s.shift = shift; s.dropout = 0; return s; };
        for (int which = 0; which < 2; ++which) {
            const int K = which == 0 ? 3 * R + C : GH, N = which == 0 ? G : S;
            const char* name = which == 0 ? "W1  " : "skip";
            const size_t per = (size_t)K * N + 2 * N, no = per * L;      // per layer: [K][N] kernel, bias, bias2
            float *o1, *o2; CK(hipMalloc(&o1, no * 4)); CK(hipMalloc(&o2, no * 4));
            auto run_v1 = [&]() {
                for (int l = 0; l < L; ++l) {
                    WgArgs w; memset(&w, 0, sizeof w); w.ones_row = 1; w.B = B; w.T = T;
                    const int d = 1 << l;
                    if (which == 0) { w.nseg = 4; const bf16_t* x = XD + (size_t)l * NT_ * R;
                        w.seg[0] = mkseg(x, R, 0, R, -2 * d); w.seg[1] = mkseg(x, R, 0, R, -d); w.seg[2] = mkseg(x, R, 0, R, 0); w.seg[3] = mkseg(cbt, C, 0, C, 0);
                        w.Bm = DZ + (size_t)l * NT_ * G; w.ldb = G; w.N = G; w.ldw = G; w.scale = 1.0f; w.bias_out2 = o1 + l * per + (size_t)K * N + N; }
                    else { w.nseg = 1; w.seg[0] = mkseg(U + (size_t)l * NT_ * GH, GH, 0, GH, 0); w.Bm = DS; w.ldb = S; w.N = S; w.ldw = S; w.scale = 0.5f + 0.1f * l; }
                    w.out = o1 + l * per; w.bias_out = o1 + l * per + (size_t)K * N;
                    if (launch_wgrad(&ctx, w, 0)) { printf("v1 launch failed: %s\n", ctx.err.c_str()); exit(3); }
                }
            };
            auto run_v2 = [&]() {
                WgBatchArgs w; memset(&w, 0, sizeof w); w.ngroups = L; w.B = B; w.T = T; w.grads = o2;
                if (which == 0) { w.nseg = 4;
                    for (int s = 0; s < 3; ++s) { w.seg_base[s] = XD; w.seg_gstride[s] = NT_ * R; w.seg_ld[s] = R; w.seg_nk[s] = R; }
                    w.seg_base[3] = cbt; w.seg_gstride[3] = 0; w.seg_ld[3] = C; w.seg_nk[3] = C;
                    w.Bm = DZ; w.b_gstride = NT_ * G; w.ldb = G; w.N = G; w.ldw = G; }
                else { w.nseg = 1; w.seg_base[0] = U; w.seg_gstride[0] = NT_ * GH; w.seg_ld[0] = GH; w.seg_nk[0] = GH; w.Bm = DS; w.b_gstride = 0; w.ldb = S; w.N = S; w.ldw = S; }
                for (int l = 0; l < L; ++l) { WgGroup& q = w.g[l]; const int d = 1 << l;
                    q.out_off = l * per; q.bias_off = l * per + (size_t)K * N; q.bias2_off = q.bias_off + N; q.has_bias2 = which == 0;
                    if (which == 0) { q.shift[0] = -2 * d; q.shift[1] = -d; } q.scale = which == 0 ? 1.0f : 0.5f + 0.1f * l; }
                if (!wn_wgrad_v2_ok(w)) { printf("v2 not applicable\n"); exit(3); }
                if (launch_wgrad_batch(&ctx, w, 0)) { printf("v2 launch failed: %s\n", ctx.err.c_str()); exit(3); }
                static int once = 0; if (once++ < 4) printf("    plan: units %d (spu %d, slab %d) tiles %dx%d partial %.1f MB\n", w.nunits, w.spu, w.slab, w.mtiles, w.ntiles, wn_wgrad_partial_bytes(w) / 1e6);
            };
            CK(hipMemset(o1, 0, no * 4)); CK(hipMemset(o2, 0, no * 4));
            run_v1(); run_v2(); CK(hipDeviceSynchronize());
            std::vector<float> h1(no), h2(no);
            CK(hipMemcpy(h1.data(), o1, no * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, no * 4, hipMemcpyDeviceToHost));
            double ew = 0, eb = 0;
            for (int l = 0; l < L; ++l) { ew = std::max(ew, rel_err(h2.data() + l * per, h1.data() + l * per, (size_t)K * N));
                                          eb = std::max(eb, rel_err(h2.data() + l * per + (size_t)K * N, h1.data() + l * per + (size_t)K * N, which == 0 ? 2 * N : N)); }
            const bool ok = ew < 1e-4 && eb < 1e-4; fails += !ok;
            const double fl = 2.0 * K * N * (double)NT_ * L;
            float t1 = time_ms(run_v1), t2 = time_ms(run_v2);
            printf("B=%d T=%5d L=%d %s  v1 %8.1f us/layer %7.1f TF | v2 grouped %8.1f us/layer %7.1f TF | relerr W %.2e bias %.2e %s\n",
                   B, T, L, name, t1 * 1e3 / L, fl / t1 / 1e9, t2 * 1e3 / L, fl / t2 / 1e9, ew, eb, ok ? "ok" : "FAIL");
            CK(hipFree(o1)); CK(hipFree(o2));
        }
        CK(hipFree(XD)); CK(hipFree(cbt)); CK(hipFree(DZ)); CK(hipFree(U)); CK(hipFree(DS));
    }
    printf("wgrad harness %s\n", fails ? "FAILED" : "passed");
    return fails != 0;
}
