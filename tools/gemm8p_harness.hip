// Stand-alone check + timing of the 8-phase main loop (csrc/wn_tile8p.h) against the production LDS-DMA kernel (csrc/wn_tile.h: THE
// product header, not a copy) on the two MFMA-bound contractions of a C2 layer:
//   gate: M = 512, K = 3*256 + 80 (dilated taps + conditioning), EPI_GATE;   d x: M = 256, K = 3*512, EPI_DX (dropout mask + residual).
// Same logical weights for both kernels (packed K-interleaved in 32- resp. 64-channel blocks), same activations => outputs agree to
// bf16 rounding of sums taken in a different order (<= 2 ulp); the 8-phase kernel is also run repeatedly and must reproduce its own
// bits (race screen: a read that overtakes its DMA shows up as run-to-run differences).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I tacotron-2_amd/csrc tools/gemm8p_harness.hip -o tools/gemm8p_harness
//   tools/gemm8p_harness [B=8] [T=11000] [d=64] [rounds=3]
#include "wn_tile8p.h"
#include "power_sampler.h"
#include <vector>
#include <random>
#include <functional>
#include <algorithm>
#include <cmath>

std::string g_create_err;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static std::mt19937 rng(1234);
static std::vector<bf16_t> host_bf16_random(size_t n, float scale) {
    std::vector<bf16_t> h(n);
    std::uniform_real_distribution<float> d(-scale, scale);
    for (auto& v : h) v = f2bf(d(rng));
    return h;
}
static bf16_t* to_dev(const std::vector<bf16_t>& h) { bf16_t* p; CK(hipMalloc(&p, h.size() * 2)); CK(hipMemcpy(p, h.data(), h.size() * 2, hipMemcpyHostToDevice)); return p; }
static bf16_t* dev_bf16_random(size_t n, float scale) { return to_dev(host_bf16_random(n, scale)); }
static float* dev_f32_random(size_t n, float scale) {
    std::vector<float> h(n);
    std::uniform_real_distribution<float> d(-scale, scale);
    for (auto& v : h) v = d(rng);
    float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p;
}
template <class Tp> static Tp* dev_fill(size_t n) { Tp* p; CK(hipMalloc(&p, n * sizeof(Tp))); CK(hipMemset(p, 0xff, n * sizeof(Tp))); return p; }

// fragment-ordered pack of a logical matrix W[M][K] (values already bf16) whose taps part (3 x nk channels) is K-interleaved in blocks of kil
static std::vector<bf16_t> pack_frag(const std::vector<bf16_t>& W, int M, int K, int nk_tap, int kil) {
    std::vector<int> kmap(K);                       // pack k -> logical k
    int kp = 0;
    for (int kb = 0; kb < nk_tap / kil; ++kb) for (int j = 0; j < 3; ++j) for (int i = 0; i < kil; ++i) kmap[kp++] = j * nk_tap + kb * kil + i;
    for (int k = 3 * nk_tap; k < K; ++k) kmap[kp++] = k;
    const int KS = K / 16;
    std::vector<bf16_t> out((size_t)M * K);
    for (int mt = 0; mt < M / 32; ++mt) for (int ks = 0; ks < KS; ++ks) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
        const int m = mt * 32 + (lane & 31), k = ks * 16 + (lane >> 5) * 8 + j;
        out[(((size_t)mt * KS + ks) * 64 + lane) * 8 + j] = W[(size_t)m * K + kmap[k]];
    }
    return out;
}

struct Cmp { double maxabs; size_t ndiff, nbad, n; };
static Cmp compare_bf16(const bf16_t* da, const bf16_t* db, size_t n) {
    std::vector<bf16_t> a(n), b(n);
    CK(hipMemcpy(a.data(), da, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), db, n * 2, hipMemcpyDeviceToHost));
    Cmp c{0, 0, 0, n};
    for (size_t i = 0; i < n; ++i) {
        if (a[i] == b[i]) continue;
        ++c.ndiff;
        const float x = bf2f(a[i]), y = bf2f(b[i]);
        const double d = std::fabs((double)x - y);
        if (!(d <= 0.0161 * std::max({std::fabs(x), std::fabs(y), 1e-3f}))) ++c.nbad;      // > 2 bf16 ulp (or NaN)
        if (d > c.maxabs || d != d) c.maxabs = d;
    }
    return c;
}
static bool same_bits(const bf16_t* da, const bf16_t* db, size_t n) {
    std::vector<bf16_t> a(n), b(n);
    CK(hipMemcpy(a.data(), da, n * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), db, n * 2, hipMemcpyDeviceToHost));
    return memcmp(a.data(), b.data(), n * 2) == 0;
}
static float time_ms(const std::function<void()>& f, int iters = 10) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); return ms / iters;
}

// the production launch of the gate / dx GEMMs (wn_launch_gemm's TAPS branch)
template <int EPI> static void launch_prod(GemmArgs a, int M, hipStream_t st) {
    a.mblocks = M / 256; a.tiles_per_utt = cdiv(a.T, 128); a.ntiles = a.tiles_per_utt * a.B;
    a.xcd_span = cdiv(a.ntiles, 8); a.taps = 3;
    const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
    a.stagger = grid >= WN_STAGGER_MIN_GRID ? 8000 : 0;
    hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI, 1, 3>), dim3(grid), dim3(512), 0, st, a);
}
template <int EPI, int ABL = 0, int SCHED = 1> static void launch_8p(GemmArgs a, int M, hipStream_t st) {
    a.mblocks = M / 256; a.tiles_per_utt = cdiv(a.T, 256); a.ntiles = a.tiles_per_utt * a.B;
    a.xcd_span = cdiv(a.ntiles, 8); a.taps = 3; a.stagger = 0;
    const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
    hipLaunchKernelGGL((wn_gemm8p_kernel<EPI, ABL, SCHED>), dim3(grid), dim3(512), 0, st, a);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 11000, d = argc > 3 ? atoi(argv[3]) : 64, rounds = argc > 4 ? atoi(argv[4]) : 3;
    const int R = 256, G = 512, GH = 256, C = 80;
    const int64_t NT_ = (int64_t)B * T;
    int fails = 0;
    printf("gemm8p harness: B %d T %d d %d (rows %lld)\n", B, T, d, (long long)NT_);
    bf16_t* zero; CK(hipMalloc(&zero, 4096)); CK(hipMemset(zero, 0, 4096));
    bf16_t* XD = dev_bf16_random(NT_ * R, 1.0f);
    bf16_t* X = dev_bf16_random(NT_ * R, 1.0f);
    bf16_t* cbt = dev_bf16_random(NT_ * C, 1.0f);
    bf16_t* DZ = dev_bf16_random(NT_ * G, 1.0f);
    float* bias = dev_f32_random(1024, 0.5f);
    auto base = [&](GemmArgs& a, const bf16_t* Apk, int K) {
        memset(&a, 0, sizeof a); a.Apk = Apk; a.ksteps_total = K / 16; a.nrep = 1; a.B = B; a.T = T; a.zero = zero; a.taps = 3;
        a.e.scale = 1.0f; a.e.GH = GH;
    };
    auto mkseg = [](const bf16_t* b, int ld, int col0, int nk, int shift) { SrcSeg s; s.base = b; s.ld = ld; s.col0 = col0; s.nk = nk; s.shift = shift; s.dropout = 0; return s; };
    {
        int nb = -1; hipFuncAttributes fa;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)wn_gemm8p_kernel<EPI_GATE, 0>, 512, 0));
        CK(hipFuncGetAttributes(&fa, (const void*)wn_gemm8p_kernel<EPI_GATE, 0>));
        printf("occupancy 8p gate: %d blocks/CU, %d regs, %zu B static LDS, scratch %zu B\n", nb, fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
        CK(hipFuncGetAttributes(&fa, (const void*)wn_gemm8p_kernel<EPI_DX, 0>));
        printf("occupancy 8p dx  : %d regs, %zu B static LDS, scratch %zu B\n", fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes);
    }
    {   // ---------------- gate
        const int M = G, K = 3 * R + C;
        const std::vector<bf16_t> W = host_bf16_random((size_t)M * K, 0.05f);
        bf16_t* A32 = to_dev(pack_frag(W, M, K, R, 32)); bf16_t* A64 = to_dev(pack_frag(W, M, K, R, 64));
        bf16_t* TS1 = dev_fill<bf16_t>(NT_ * GH); bf16_t* U1 = dev_fill<bf16_t>(NT_ * GH);
        bf16_t* TS2 = dev_fill<bf16_t>(NT_ * GH); bf16_t* U2 = dev_fill<bf16_t>(NT_ * GH);
        bf16_t* TS3 = dev_fill<bf16_t>(NT_ * GH); bf16_t* U3 = dev_fill<bf16_t>(NT_ * GH);
        GemmArgs a; base(a, A32, K); a.nseg = 4;
        a.seg[0] = mkseg(XD, R, 0, R, -2 * d); a.seg[1] = mkseg(XD, R, 0, R, -d); a.seg[2] = mkseg(XD, R, 0, R, 0); a.seg[3] = mkseg(cbt, C, 0, C, 0);
        a.e.bias = bias; a.e.ld_out0 = GH; a.e.ld_out1 = GH; a.e.M_valid = M;
        GemmArgs a1 = a; a1.e.out0 = TS1; a1.e.out1 = U1;
        GemmArgs a2 = a; a2.Apk = A64; a2.e.out0 = TS2; a2.e.out1 = U2;
        GemmArgs a3 = a2; a3.e.out0 = TS3; a3.e.out1 = U3;
        if (!wn_gemm8p_fits(a2, M, NT_)) { printf("gate: wn_gemm8p_fits says no\n"); return 1; }
        const double fl = 2.0 * M * K * (double)NT_;
        launch_prod<EPI_GATE>(a1, M, 0); launch_8p<EPI_GATE>(a2, M, 0); CK(hipDeviceSynchronize());
        Cmp cs = compare_bf16(TS1, TS2, NT_ * GH), cu = compare_bf16(U1, U2, NT_ * GH);
        printf("gate   8p vs production: sigmoid %zu of %zu differ (max |d| %.4g, %zu beyond 2 ulp) | u %zu differ (max |d| %.4g, %zu beyond 2 ulp)  %s\n",
               cs.ndiff, cs.n, cs.maxabs, cs.nbad, cu.ndiff, cu.maxabs, cu.nbad, (cs.nbad + cu.nbad) ? "FAIL" : "ok");
        fails += (cs.nbad + cu.nbad) != 0;
        {   // both schedules add the same products in the same order: bitwise equal
            CK(hipMemset(TS3, 0xff, NT_ * GH * 2)); CK(hipMemset(U3, 0xff, NT_ * GH * 2));
            launch_8p<EPI_GATE, 0, 0>(a3, M, 0); CK(hipDeviceSynchronize());
            const bool ok = same_bits(TS2, TS3, NT_ * GH) && same_bits(U2, U3, NT_ * GH);
            printf("gate   8p schedule 0 vs schedule 1: %s\n", ok ? "bitwise-ok" : "FAIL"); fails += !ok;
        }
        for (int rep = 0; rep < 4; ++rep) {        // race screen: the kernel must reproduce its own bits, also beside another copy of itself
            CK(hipMemset(TS3, 0xff, NT_ * GH * 2)); CK(hipMemset(U3, 0xff, NT_ * GH * 2));
            launch_8p<EPI_GATE>(a3, M, 0); if (rep & 1) launch_8p<EPI_GATE>(a3, M, 0);
            CK(hipDeviceSynchronize());
            const bool ok = same_bits(TS2, TS3, NT_ * GH) && same_bits(U2, U3, NT_ * GH);
            if (!ok) { printf("gate   8p race screen run %d: bits differ  FAIL\n", rep); ++fails; }
        }
        for (int rnd = 0; rnd < rounds; ++rnd) {
            const float tp = time_ms([&] { launch_prod<EPI_GATE>(a1, M, 0); });
            const float t8 = time_ms([&] { launch_8p<EPI_GATE, 0, 1>(a2, M, 0); });
            const float tm = time_ms([&] { launch_8p<EPI_GATE, 1, 1>(a2, M, 0); });
            const float t80 = time_ms([&] { launch_8p<EPI_GATE, 0, 0>(a2, M, 0); });
            const float tm0 = time_ms([&] { launch_8p<EPI_GATE, 1, 0>(a2, M, 0); });
            printf("gate   production %7.1f us %6.1f TF | 8-phase DMA-in-MFMA-block %7.1f us %6.1f TF, main loop only %7.1f us %6.1f TF | DMA-in-load-section %7.1f us %6.1f TF, main loop only %7.1f us %6.1f TF\n",
                   tp * 1e3, fl / tp / 1e9, t8 * 1e3, fl / t8 / 1e9, tm * 1e3, fl / tm / 1e9, t80 * 1e3, fl / t80 / 1e9, tm0 * 1e3, fl / tm0 / 1e9);
        }
        for (int rnd = 0; rnd < rounds; ++rnd) {     // VERDICT round 5 item 2, bounded before it is built: a persistent grid can hide the prologue (next tile's first DMAs under the epilogue) --
            const float t8 = time_ms([&] { launch_8p<EPI_GATE, 0, 1>(a2, M, 0); }), t2 = time_ms([&] { launch_8p<EPI_GATE, 2, 1>(a2, M, 0); });      // ABL 2 = the same kernel not waiting for it at all
            printf("gate   8-phase whole kernel %7.1f us | with the prologue's wait removed (upper bound of a persistent grid's prefetch) %7.1f us = %.1f %%\n", t8 * 1e3, t2 * 1e3, 100.0 * (t8 - t2) / t8);
        }
        // energy per launch (power as a bound): same work, same operands
        report_power("gate, ring kernel (production)", measure_power([&] { launch_prod<EPI_GATE>(a1, M, 0); }), fl);
        report_power("gate, 8-phase kernel", measure_power([&] { launch_8p<EPI_GATE, 0, 1>(a2, M, 0); }), fl);
        report_power("gate, 8-phase main loop only", measure_power([&] { launch_8p<EPI_GATE, 1, 1>(a2, M, 0); }), fl);
        if (B >= 2) {       // what the step launches: half batches on two streams
            hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            GemmArgs h1 = a1, h2 = a1, g1 = a2, g2 = a2;
            h1.B = g1.B = B / 2; h2.B = g2.B = B - B / 2; h2.b0 = g2.b0 = B / 2;
            for (int rnd = 0; rnd < rounds; ++rnd) {
                const float tp = time_ms([&] { launch_prod<EPI_GATE>(h1, M, s1); launch_prod<EPI_GATE>(h2, M, s2); CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); }, 5);
                const float t8 = time_ms([&] { launch_8p<EPI_GATE>(g1, M, s1); launch_8p<EPI_GATE>(g2, M, s2); CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); }, 5);
                printf("gate   two half batches on two streams (host-synchronised): production %7.1f us | 8-phase %7.1f us\n", tp * 1e3, t8 * 1e3);
            }
        }
    }
    {   // ---------------- dx
        const int M = R, K = 3 * G;
        const std::vector<bf16_t> W = host_bf16_random((size_t)M * K, 0.05f);
        bf16_t* A32 = to_dev(pack_frag(W, M, K, G, 32)); bf16_t* A64 = to_dev(pack_frag(W, M, K, G, 64));
        bf16_t* O1 = dev_fill<bf16_t>(NT_ * R); bf16_t* O2 = dev_fill<bf16_t>(NT_ * R); bf16_t* O3 = dev_fill<bf16_t>(NT_ * R);
        GemmArgs a; base(a, A32, K); a.nseg = 3;
        a.seg[0] = mkseg(DZ, G, 0, G, 2 * d); a.seg[1] = mkseg(DZ, G, 0, G, d); a.seg[2] = mkseg(DZ, G, 0, G, 0);
        a.key_lo = 0x1234567u; a.key_hi = 0x89abcdefu; a.thresh16 = 3277; a.keep_scale = 1.0f / 0.95f; a.drop_ld = R;
        a.e.in0 = X; a.e.ld_in0 = R; a.e.scale = WN_SQRT_HALF; a.e.ld_out0 = R; a.e.M_valid = M;
        GemmArgs a1 = a; a1.e.out0 = O1; GemmArgs a2 = a; a2.Apk = A64; a2.e.out0 = O2; GemmArgs a3 = a2; a3.e.out0 = O3;
        if (!wn_gemm8p_fits(a2, M, NT_)) { printf("dx: wn_gemm8p_fits says no\n"); return 1; }
        const double fl = 2.0 * M * K * (double)NT_;
        launch_prod<EPI_DX>(a1, M, 0); launch_8p<EPI_DX>(a2, M, 0); CK(hipDeviceSynchronize());
        Cmp c = compare_bf16(O1, O2, NT_ * R);
        printf("dx     8p vs production: %zu of %zu differ (max |d| %.4g, %zu beyond 2 ulp)  %s\n", c.ndiff, c.n, c.maxabs, c.nbad, c.nbad ? "FAIL" : "ok");
        fails += c.nbad != 0;
        {
            CK(hipMemset(O3, 0xff, NT_ * R * 2));
            launch_8p<EPI_DX, 0, 0>(a3, M, 0); CK(hipDeviceSynchronize());
            const bool ok = same_bits(O2, O3, NT_ * R);
            printf("dx     8p schedule 0 vs schedule 1: %s\n", ok ? "bitwise-ok" : "FAIL"); fails += !ok;
        }
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemset(O3, 0xff, NT_ * R * 2));
            launch_8p<EPI_DX>(a3, M, 0); if (rep & 1) launch_8p<EPI_DX>(a3, M, 0);
            CK(hipDeviceSynchronize());
            if (!same_bits(O2, O3, NT_ * R)) { printf("dx     8p race screen run %d: bits differ  FAIL\n", rep); ++fails; }
        }
        for (int rnd = 0; rnd < rounds; ++rnd) {
            const float tp = time_ms([&] { launch_prod<EPI_DX>(a1, M, 0); });
            const float t8 = time_ms([&] { launch_8p<EPI_DX, 0, 1>(a2, M, 0); });
            const float tm = time_ms([&] { launch_8p<EPI_DX, 1, 1>(a2, M, 0); });
            const float t80 = time_ms([&] { launch_8p<EPI_DX, 0, 0>(a2, M, 0); });
            const float tm0 = time_ms([&] { launch_8p<EPI_DX, 1, 0>(a2, M, 0); });
            printf("dx     production %7.1f us %6.1f TF | 8-phase DMA-in-MFMA-block %7.1f us %6.1f TF, main loop only %7.1f us %6.1f TF | DMA-in-load-section %7.1f us %6.1f TF, main loop only %7.1f us %6.1f TF\n",
                   tp * 1e3, fl / tp / 1e9, t8 * 1e3, fl / t8 / 1e9, tm * 1e3, fl / tm / 1e9, t80 * 1e3, fl / t80 / 1e9, tm0 * 1e3, fl / tm0 / 1e9);
        }
        for (int rnd = 0; rnd < rounds; ++rnd) {
            const float t8 = time_ms([&] { launch_8p<EPI_DX, 0, 1>(a2, M, 0); }), t2 = time_ms([&] { launch_8p<EPI_DX, 2, 1>(a2, M, 0); });
            printf("dx     8-phase whole kernel %7.1f us | with the prologue's wait removed %7.1f us = %.1f %%\n", t8 * 1e3, t2 * 1e3, 100.0 * (t8 - t2) / t8);
        }
        report_power("d x, ring kernel (production)", measure_power([&] { launch_prod<EPI_DX>(a1, M, 0); }), fl);
        report_power("d x, 8-phase kernel", measure_power([&] { launch_8p<EPI_DX, 0, 1>(a2, M, 0); }), fl);
    }
    printf("(energy of the d x launches: see above per kernel)\n");
    printf("gemm8p harness %s (%d failing checks)\n", fails ? "FAILED" : "passed", fails);
    return fails != 0;
}
