// Weight-stationary streaming out conv (tools/wn_outws_variant.h; measured and rejected, see its header) against the tile-engine launch it replaces: bitwise comparison of x_{l+1} and
// its dropout copy on the benchmark geometry (8 x 11 000 rows, the last tile of every utterance partial), then timings -- alone on the
// GPU (whole batch, half batch) and beside the MFMA-bound gate GEMM of the other half batch on a second stream (how the step runs them).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I tacotron-2_amd/csrc -I tools tools/outws_harness.hip -o tools/outws_harness
#include "wn_outws_variant.h"
#include <vector>
#include <random>
std::string g_create_err;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
static std::mt19937 rng(11);
static bf16_t* dev_bf16_random(size_t n, float scale) {
    std::vector<bf16_t> h(n); std::uniform_real_distribution<float> d(-scale, scale);
    for (auto& v : h) v = f2bf(d(rng));
    bf16_t* p; CK(hipMalloc(&p, n * 2)); CK(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice)); return p;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int B = 8, T = 11000, R = 256, G = 512, GH = 256, C = 80;
    const int64_t NT = (int64_t)B * T;
    bf16_t* zero; CK(hipMalloc(&zero, 256)); CK(hipMemset(zero, 0, 256));
    bf16_t* X = dev_bf16_random((size_t)NT * R, 1.0f);
    bf16_t* U = dev_bf16_random((size_t)NT * GH, 1.0f);
    bf16_t* cbt = dev_bf16_random((size_t)NT * C, 1.0f);
    bf16_t* Wo = dev_bf16_random((size_t)R * GH, 0.05f);
    bf16_t* W1 = dev_bf16_random((size_t)G * (3 * R + C), 0.03f);
    std::vector<float> hb(1024); std::uniform_real_distribution<float> d(-0.5f, 0.5f); for (auto& v : hb) v = d(rng);
    float* bias; CK(hipMalloc(&bias, 4096)); CK(hipMemcpy(bias, hb.data(), 4096, hipMemcpyHostToDevice));
    auto alloc = [&](size_t n) { bf16_t* p; CK(hipMalloc(&p, n * 2)); CK(hipMemset(p, 0, n * 2)); return p; };
    bf16_t* O0[2] = {alloc((size_t)NT * R), alloc((size_t)NT * R)}; bf16_t* O1[2] = {alloc((size_t)NT * R), alloc((size_t)NT * R)};
    bf16_t* TS = alloc((size_t)NT * GH); bf16_t* UG = alloc((size_t)NT * GH);
    auto mkseg = [](const bf16_t* b, int ld, int nk, int shift) { SrcSeg s; s.base = b; s.ld = ld; s.col0 = 0; s.nk = nk; s.shift = shift; s.dropout = 0; return s; };
    auto mk_out = [&](int set, int b0, int nb, bool drop) {
        GemmArgs a; memset(&a, 0, sizeof a); a.Apk = Wo; a.ksteps_total = GH / 16; a.nrep = 1; a.B = nb; a.T = T; a.b0 = b0; a.zero = zero; a.e.scale = 0.70710678f; a.e.GH = GH; a.e.M_valid = R;
        a.nseg = 1; a.seg[0] = mkseg(U, GH, GH, 0);
        a.e.bias = bias; a.e.in0 = X; a.e.ld_in0 = R; a.e.out0 = O0[set]; a.e.ld_out0 = R;
        if (drop) { a.e.out1 = O1[set]; a.e.ld_out1 = R; wn_layer_key(1234, 3, &a.key_lo, &a.key_hi); a.thresh16 = (uint32_t)lrintf(0.05f * 65536.0f); a.keep_scale = 1.0f / 0.95f; a.drop_ld = R; }
        return a;
    };
    auto prep = [&](GemmArgs& a, int M, int TT) { a.mblocks = M / 256; a.tiles_per_utt = cdiv(a.T, TT); a.ntiles = a.tiles_per_utt * a.B; a.xcd_span = cdiv(a.ntiles, 8); return cdiv(a.ntiles, 8) * a.mblocks * 8; };
    auto launch_tile = [&](GemmArgs a, hipStream_t st) {       // the library's rule: 256 x 64 tiles when the 128-row tiles would not fill the slots
        if ((int64_t)cdiv(T, 128) * a.B < 512) { const int grid = prep(a, R, 64); hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 1, 4, 2, 64, 2, EPI_STORE_BF16, 1, 0>), dim3(grid), dim3(512), 0, st, a); }
        else { const int grid = prep(a, R, 128); a.stagger = grid >= WN_STAGGER_MIN_GRID ? 8000 : 0; hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI_STORE_BF16, 1, 0>), dim3(grid), dim3(512), 0, st, a); } };
    auto launch_ws = [&](GemmArgs a, hipStream_t st) { if (!wn_out_ws_fits(a, R)) { printf("!! shape does not fit the streaming kernel\n"); exit(3); } wn_launch_out_ws(nullptr, a, st); };
    auto launch_gate = [&](int b0, int nb, hipStream_t st) {
        GemmArgs a; memset(&a, 0, sizeof a); a.Apk = W1; a.ksteps_total = (3 * R + C) / 16; a.nrep = 1; a.B = nb; a.T = T; a.b0 = b0; a.zero = zero; a.e.scale = 1.0f; a.e.GH = GH; a.e.M_valid = G;
        a.nseg = 4; a.seg[0] = mkseg(X, R, R, -2 * 64); a.seg[1] = mkseg(X, R, R, -64); a.seg[2] = mkseg(X, R, R, 0); a.seg[3] = mkseg(cbt, C, C, 0); a.taps = 3;
        a.e.bias = bias; a.e.out0 = TS; a.e.ld_out0 = GH; a.e.out1 = UG; a.e.ld_out1 = GH;
        const int grid = prep(a, G, 128); a.stagger = grid >= WN_STAGGER_MIN_GRID ? 8000 : 0;
        hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI_GATE, 1, 3>), dim3(grid), dim3(512), 0, st, a); };
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ef));
    // ---- bitwise comparison: whole batch and the second half batch (b0 = 4), with and without the dropout copy
    std::vector<bf16_t> ha((size_t)NT * R), hb2((size_t)NT * R);
    auto same = [&](bf16_t* p, bf16_t* q, const char* what) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ha.data(), p, ha.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb2.data(), q, hb2.size() * 2, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0; for (size_t i = 0; i < ha.size(); ++i) if (ha[i] != hb2[i]) { if (!bad) first = i; ++bad; }
        printf("  %-34s %zu of %zu elements differ%s\n", what, bad, ha.size(), bad ? "" : "  (bitwise equal)");
        if (bad) printf("      first at row %zu channel %zu: %g vs %g\n", first / R, first % R, bf2f(ha[first]), bf2f(hb2[first]));
        return bad == 0; };
    bool ok = true;
    for (int drop = 1; drop >= 0; --drop) for (int half = 0; half < 2; ++half) {
        for (int s = 0; s < 2; ++s) { CK(hipMemset(O0[s], 0, (size_t)NT * R * 2)); CK(hipMemset(O1[s], 0, (size_t)NT * R * 2)); }
        const int b0 = half ? 4 : 0, nb = half ? 4 : 8;
        launch_tile(mk_out(0, b0, nb, drop), s0); launch_ws(mk_out(1, b0, nb, drop), s0);
        printf("%s, dropout copy %s:\n", half ? "utterances 4..7" : "whole batch", drop ? "on" : "off");
        ok = same(O0[0], O0[1], "x_{l+1}") && ok; if (drop) ok = same(O1[0], O1[1], "dropout copy") && ok;
    }
    printf(ok ? "RESULTS IDENTICAL\n" : "RESULTS DIFFER\n");
    // ---- timings
    const int REPS = 20; float ms;
    auto timeit = [&](auto f, const char* name, double bytes) {
        f(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s0)); for (int i = 0; i < REPS; ++i) f(); CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-66s %7.1f us", name, ms * 1e3 / REPS); if (bytes > 0) printf("   %.2f TB/s", bytes / (ms * 1e-3 / REPS) / 1e12); printf("\n"); };
    for (int rep = 0; rep < 2; ++rep) {
        timeit([&] { launch_tile(mk_out(0, 0, 8, true), s0); }, "out conv, whole batch, tile engine (256 x 128 tiles)", 2048.0 * NT);
        timeit([&] { launch_ws(mk_out(1, 0, 8, true), s0); }, "out conv, whole batch, weight-stationary streaming", 2048.0 * NT);
        timeit([&] { launch_tile(mk_out(0, 0, 4, true), s0); }, "out conv, half batch, tile engine (256 x 64 tiles)", 1024.0 * NT);
        timeit([&] { launch_ws(mk_out(1, 0, 4, true), s0); }, "out conv, half batch, weight-stationary streaming", 1024.0 * NT);
        timeit([&] { launch_gate(0, 4, s0); }, "gate GEMM, half batch, alone", 0);
        auto pair = [&](bool ws) { CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0));
            launch_gate(0, 4, s0); if (ws) launch_ws(mk_out(1, 4, 4, true), s1); else launch_tile(mk_out(0, 4, 4, true), s1);
            CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s0, ef, 0)); };
        timeit([&] { pair(false); }, "gate (half A) || out conv (half B), tile engine", 0);
        timeit([&] { pair(true); }, "gate (half A) || out conv (half B), weight-stationary streaming", 0);
        auto layer = [&](bool ws) { CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0));
            for (int k = 0; k < 4; ++k) for (int p = 0; p < 2; ++p) { hipStream_t st = p ? s1 : s0; launch_gate(p * 4, 4, st); if (ws) launch_ws(mk_out(1, p * 4, 4, true), st); else launch_tile(mk_out(0, p * 4, 4, true), st); }
            CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s0, ef, 0)); };
        timeit([&] { layer(false); }, "4 layers x (gate -> out conv) on two half-batch streams, tile engine", 0);
        timeit([&] { layer(true); }, "4 layers x (gate -> out conv) on two half-batch streams, streaming out conv", 0);
    }
    return ok ? 0 : 1;
}
