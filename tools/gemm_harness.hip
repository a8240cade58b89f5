// NOTE (round 5): this harness compiles tools/wn_tile_variants.h -- a FORK of the library's tile engine that keeps the schedules measured and
// rejected in rounds 2 - 4; its "production" arm is that fork's copy of the main loop.  The production kernels THEMSELVES (csrc/wn_tile.h,
// included unchanged) are timed by tools/gemm8p_harness.hip, next to the 8-phase kernel and the vendor GEMM (tools/vendor_gemm_yardstick.py).
// Stand-alone check + timing of the tile-engine main loops (run on the GPU box):
//   v1 = wn_gemm_tile_kernel (A fragments straight from L2), v2 = wn_gemm_lds_kernel (LDS-DMA ring).
// Same GemmArgs, same accumulation order => outputs must be BITWISE identical.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I tacotron-2_amd/csrc -I tools tools/gemm_harness.hip -o tools/gemm_harness
#include "wn_tile_variants.h"
#include <vector>
#include <random>
#include <functional>
#include <map>
#include <algorithm>

std::string g_create_err;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static std::mt19937 rng(1234);
static bf16_t* dev_bf16_random(size_t n, float scale) {
    std::vector<bf16_t> h(n);
    std::uniform_real_distribution<float> d(-scale, scale);
    for (auto& v : h) v = f2bf(d(rng));
    bf16_t* p; CK(hipMalloc(&p, n * 2)); CK(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice)); return p;
}
static float* dev_f32_random(size_t n, float scale) {
    std::vector<float> h(n);
    std::uniform_real_distribution<float> d(-scale, scale);
    for (auto& v : h) v = d(rng);
    float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p;
}
template <class Tp> static Tp* dev_zero(size_t n) { Tp* p; CK(hipMalloc(&p, n * sizeof(Tp))); CK(hipMemset(p, 0xff, n * sizeof(Tp))); return p; }

struct Out { void* p; size_t bytes; };
static bool same(const Out& a, const Out& b, const char* what) {
    std::vector<unsigned char> ha(a.bytes), hb(b.bytes);
    CK(hipMemcpy(ha.data(), a.p, a.bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b.p, b.bytes, hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < a.bytes; ++i) if (ha[i] != hb[i]) { if (!bad) first = i; ++bad; }
    if (bad) printf("    MISMATCH %s: %zu of %zu bytes differ (first at %zu)\n", what, bad, a.bytes, first);
    return bad == 0;
}

template <int MT, int NT, int WM, int WN, int EPI>
static void launch_v1(GemmArgs a, int M, hipStream_t st) {
    const int nrows = WN * NT * 32, mrows = WM * MT * 32;
    a.mblocks = M / mrows; a.tiles_per_utt = cdiv(a.T, nrows); a.ntiles = a.tiles_per_utt * a.B;
    const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
    hipLaunchKernelGGL((wn_gemm_tile_kernel<MT, NT, WM, WN, EPI>), dim3(grid), dim3(WM * WN * 64), 0, st, a);
}
template <int MT, int NT, int WM, int WN, int BK, int NBUF, int EPI, int PIPE = 0>
static void launch_v2(GemmArgs a, int M, hipStream_t st) {
    const int nrows = WN * NT * 32, mrows = WM * MT * 32;
    a.mblocks = M / mrows; a.tiles_per_utt = cdiv(a.T, nrows); a.ntiles = a.tiles_per_utt * a.B;
    const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
    hipLaunchKernelGGL((wn_gemm_lds_kernel<MT, NT, WM, WN, BK, NBUF, EPI, PIPE>), dim3(grid), dim3(WM * WN * 64), 0, st, a);
}
// the production launch of the gate / dx GEMMs: K-interleaved taps (ring slot == tap), contiguous tile run per XCD, staggered start
template <int EPI, int PIPE>
static void launch_prod(GemmArgs a, int M, hipStream_t st) {
    a.mblocks = M / 256; a.tiles_per_utt = cdiv(a.T, 128); a.ntiles = a.tiles_per_utt * a.B;
    a.xcd_span = cdiv(a.ntiles, 8); a.taps = 3;
    const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
    a.stagger = grid >= 1024 ? 8000 : 0;
    hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI, PIPE, 3>), dim3(grid), dim3(512), 0, st, a);
}
// A-direct variant of the production launch (PIPE_ 8: weight fragments straight from the pack into VGPRs, 8 x 1 waves of 32 x 128)
template <int EPI>
static void launch_prod_ad(GemmArgs a, int M, hipStream_t st) {
    a.mblocks = M / 256; a.tiles_per_utt = cdiv(a.T, 128); a.ntiles = a.tiles_per_utt * a.B;
    a.xcd_span = cdiv(a.ntiles, 8); a.taps = 3;
    const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
    a.stagger = grid >= 1024 ? 8000 : 0;
    hipLaunchKernelGGL((wn_gemm_lds_kernel<1, 4, 8, 1, 32, 3, EPI, 8, 3>), dim3(grid), dim3(512), 0, st, a);
}
// 16-wave variant: ONE 1024-thread workgroup per CU computes a 256 x 256 tile as 4 x 4 waves of 64 x 64 (same per-wave code and VGPR budget as
// the production 256 x 128 kernel, same 16 waves per CU), so the A chunk in LDS is shared by twice the time rows: 2 DMAs per wave and chunk instead of 3
template <int EPI>
static void launch_prod16(GemmArgs a, int M, hipStream_t st) {
    a.mblocks = M / 256; a.tiles_per_utt = cdiv(a.T, 256); a.ntiles = a.tiles_per_utt * a.B;
    a.xcd_span = cdiv(a.ntiles, 8); a.taps = 3;
    const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
    a.stagger = 0;
    hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 4, 32, 3, EPI, 1, 3>), dim3(grid), dim3(1024), 0, st, a);
}
static float time_ms(const std::function<void()>& f, int iters = 10) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 11000;
    const int R = 256, G = 512, GH = 256, S = 256, C = 80, L = 4;
    const int64_t NT_ = (int64_t)B * T;
    int fails = 0;
    bf16_t* zero; CK(hipMalloc(&zero, 256)); CK(hipMemset(zero, 0, 256));
    bf16_t* XD = dev_bf16_random(NT_ * R, 1.0f);
    bf16_t* X = dev_bf16_random(NT_ * R, 1.0f);
    bf16_t* cbt = dev_bf16_random(NT_ * C, 1.0f);
    bf16_t* U = dev_bf16_random((size_t)L * NT_ * GH, 1.0f);
    bf16_t* DZ = dev_bf16_random(NT_ * G, 1.0f);
    bf16_t* TSin = dev_bf16_random(NT_ * G, 1.0f);
    float* bias = dev_f32_random(1024, 0.5f);
    auto base = [&](GemmArgs& a, const bf16_t* Apk, int K) {
        memset(&a, 0, sizeof a); a.Apk = Apk; a.ksteps_total = K / 16; a.nrep = 1; a.B = B; a.T = T; a.zero = zero;
        a.e.scale = 1.0f; a.e.GH = GH;
    };
    auto mkseg = [](const bf16_t* b, int ld, int col0, int nk, int shift) { SrcSeg s; s.base = b; s.ld = ld; s.col0 = col0; s.nk = nk; s.shift = shift; s.dropout = 0; return s; };

    {
        int nb = -1; hipFuncAttributes fa;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI_GATE, 1>, 512, 0));
        CK(hipFuncGetAttributes(&fa, (const void*)wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI_GATE, 1>));
        printf("occupancy gate<2,2,4,2,32,3,PIPE1>: %d blocks/CU, %d regs, %zu B static LDS, maxThreads %d\n", nb, fa.numRegs, fa.sharedSizeBytes, fa.maxThreadsPerBlock);
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI_DX, 1>, 512, 0));
        CK(hipFuncGetAttributes(&fa, (const void*)wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI_DX, 1>));
        printf("occupancy dx<2,2,4,2,32,3,PIPE1>: %d blocks/CU, %d regs, %zu B static LDS\n", nb, fa.numRegs, fa.sharedSizeBytes);
    }
    {   // ---------------- gate GEMM: M = 512, K = 3*256 + 80 = 848
        const int M = G, K = 3 * R + C, d = 64;
        bf16_t* Apk = dev_bf16_random((size_t)M * K, 0.05f);
        bf16_t* TS1 = dev_zero<bf16_t>(NT_ * G); bf16_t* U1 = dev_zero<bf16_t>(NT_ * GH);
        bf16_t* TS2 = dev_zero<bf16_t>(NT_ * G); bf16_t* U2 = dev_zero<bf16_t>(NT_ * GH);
        GemmArgs a; base(a, Apk, K); a.nseg = 4;
        a.seg[0] = mkseg(XD, R, 0, R, -2 * d); a.seg[1] = mkseg(XD, R, 0, R, -d); a.seg[2] = mkseg(XD, R, 0, R, 0); a.seg[3] = mkseg(cbt, C, 0, C, 0);
        a.e.bias = bias; a.e.ld_out0 = GH; a.e.ld_out1 = GH; a.e.M_valid = M;      // saved: sigmoid [rows][GH] + u [rows][GH]
        GemmArgs a1 = a; a1.e.out0 = TS1; a1.e.out1 = U1;
        GemmArgs a2 = a; a2.e.out0 = TS2; a2.e.out1 = U2;
        const double fl = 2.0 * M * K * (double)NT_;
        float t1 = time_ms([&] { launch_v1<2, 2, 2, 2, EPI_GATE>(a1, M, 0); });
        printf("gate   v1 128x128            : %8.1f us  %7.1f TF\n", t1 * 1e3, fl / t1 / 1e9);
        // the LDS kernels start their accumulators at the bias (v1 adds it in the epilogue: last-bit differences), so the bitwise
        // reference of every v2 variant is the plain v2 main loop (PIPE 0)
        launch_v2<2, 2, 4, 2, 32, 3, EPI_GATE, 0>(a1, M, 0); CK(hipDeviceSynchronize());
#define TRY_GATE2(MT_, NT_, WM_, WN_, BK_, NB_) { CK(hipMemset(TS2, 0xff, NT_ * G * 2)); CK(hipMemset(U2, 0xff, NT_ * GH * 2)); \
        float t2 = time_ms([&] { launch_v2<MT_, NT_, WM_, WN_, BK_, NB_, EPI_GATE>(a2, M, 0); }); CK(hipDeviceSynchronize()); \
        bool ok = same({TS1, (size_t)NT_ * G * 2}, {TS2, (size_t)NT_ * G * 2}, "TS") & same({U1, (size_t)NT_ * GH * 2}, {U2, (size_t)NT_ * GH * 2}, "U"); \
        printf("gate   v2 MT%d NT%d WM%d WN%d BK%d NBUF%d   : %8.1f us  %7.1f TF  %s\n", MT_, NT_, WM_, WN_, BK_, NB_, t2 * 1e3, fl / t2 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok; }
#define TRY_GATE3(MT_, NT_, WM_, WN_, BK_, NB_, PP_) { CK(hipMemset(TS2, 0xff, NT_ * G * 2)); CK(hipMemset(U2, 0xff, NT_ * GH * 2)); \
        float t2 = time_ms([&] { launch_v2<MT_, NT_, WM_, WN_, BK_, NB_, EPI_GATE, PP_>(a2, M, 0); }); CK(hipDeviceSynchronize()); \
        bool ok = same({TS1, (size_t)NT_ * G * 2}, {TS2, (size_t)NT_ * G * 2}, "TS") & same({U1, (size_t)NT_ * GH * 2}, {U2, (size_t)NT_ * GH * 2}, "U"); \
        printf("gate   v2 MT%d NT%d WM%d WN%d BK%d NBUF%d PIPE%d : %8.1f us  %7.1f TF  %s\n", MT_, NT_, WM_, WN_, BK_, NB_, PP_, t2 * 1e3, fl / t2 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok; }
        // round 2: PIPE 1 (production) vs PIPE 6 / 7 (wave halves half a chunk apart, 7 = + s_setprio around the MFMA block), interleaved rounds
        for (int rnd = 0; rnd < 3; ++rnd) {
            for (int sg : {0, 8000}) { a2.stagger = sg; printf("  stagger %5d: ", sg); TRY_GATE3(2, 2, 4, 2, 32, 3, 1) printf("  stagger %5d: ", sg); TRY_GATE3(2, 2, 4, 2, 32, 3, 6) printf("  stagger %5d: ", sg); TRY_GATE3(2, 2, 4, 2, 32, 3, 7) }
        }
        a2.stagger = 0;
        // fat waves: 4 waves per workgroup, 128 x 64 / 128 x 128 accumulator tiles per wave (fewer LDS fragment bytes per MFMA)
        for (int rnd = 0; rnd < 2; ++rnd) {
            TRY_GATE3(2, 2, 4, 2, 32, 3, 1)
            TRY_GATE3(4, 2, 2, 2, 32, 3, 1) TRY_GATE3(4, 2, 2, 2, 32, 3, 2) TRY_GATE3(2, 4, 2, 2, 32, 3, 1)
            TRY_GATE3(4, 4, 2, 2, 32, 3, 1) TRY_GATE3(4, 4, 2, 2, 32, 3, 2) TRY_GATE3(4, 4, 2, 2, 64, 2, 1)
            TRY_GATE3(4, 2, 2, 4, 32, 3, 1) TRY_GATE3(2, 4, 4, 2, 32, 3, 1) TRY_GATE3(4, 2, 2, 4, 32, 3, 2)
        }
        {   // A-direct vs production (TAPS 3 + the conditioning segment), same pack: bitwise
            bf16_t* TS3 = dev_zero<bf16_t>(NT_ * G); bf16_t* U3 = dev_zero<bf16_t>(NT_ * GH);
            GemmArgs a3 = a; a3.e.out0 = TS3; a3.e.out1 = U3; GemmArgs ap = a2; ap.stagger = 0;
            launch_prod<EPI_GATE, 1>(ap, M, 0); CK(hipDeviceSynchronize());
            for (int rnd = 0; rnd < 3; ++rnd) {
                CK(hipMemset(TS3, 0xff, NT_ * G * 2)); CK(hipMemset(U3, 0xff, NT_ * GH * 2));
                float tp = time_ms([&] { launch_prod<EPI_GATE, 1>(ap, M, 0); }); float ta = time_ms([&] { launch_prod_ad<EPI_GATE>(a3, M, 0); }); CK(hipDeviceSynchronize());
                bool ok = same({TS2, (size_t)NT_ * G * 2}, {TS3, (size_t)NT_ * G * 2}, "TS") & same({U2, (size_t)NT_ * GH * 2}, {U3, (size_t)NT_ * GH * 2}, "U");
                printf("gate   A-direct 8x1 waves: production %8.1f us %7.1f TF   A-direct %8.1f us %7.1f TF  %s\n", tp * 1e3, fl / tp / 1e9, ta * 1e3, fl / ta / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok;
                CK(hipMemset(TS3, 0xff, NT_ * G * 2)); CK(hipMemset(U3, 0xff, NT_ * GH * 2));
                float t16 = time_ms([&] { launch_prod16<EPI_GATE>(a3, M, 0); }); CK(hipDeviceSynchronize());
                ok = same({TS2, (size_t)NT_ * G * 2}, {TS3, (size_t)NT_ * G * 2}, "TS") & same({U2, (size_t)NT_ * GH * 2}, {U3, (size_t)NT_ * GH * 2}, "U");
                printf("gate   16 waves 256x256     : %8.1f us %7.1f TF  %s\n", t16 * 1e3, fl / t16 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok;
#ifdef WN_EPI_ABLATE
                { float tq = time_ms([&] { GemmArgs q = ap; q.mblocks = M / 256; q.tiles_per_utt = cdiv(q.T, 128); q.ntiles = q.tiles_per_utt * q.B; q.xcd_span = cdiv(q.ntiles, 8); q.taps = 3; q.stagger = -128;
                      hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI_GATE, 1, 3>), dim3(cdiv(q.ntiles, 8) * q.mblocks * 8), dim3(512), 0, 0, q); });
                  printf("gate   probe 128 (2 DMAs per wave and chunk instead of 3, sums wrong): %8.1f us %7.1f TF\n", tq * 1e3, fl / tq / 1e9); }
#endif
            }
        }
        TRY_GATE3(2, 2, 4, 2, 32, 2, 1) TRY_GATE3(2, 2, 4, 2, 64, 3, 1) TRY_GATE3(2, 2, 4, 2, 64, 3, 6) TRY_GATE3(2, 2, 4, 2, 64, 2, 6) TRY_GATE3(2, 4, 4, 2, 32, 3, 6) TRY_GATE3(4, 2, 2, 4, 32, 3, 6)
#ifdef WN_EPI_ABLATE
        {   // main-loop timeline of the first workgroups (wave 0 and wave 5): stamps [before vmcnt wait, after it, after barrier, after DMA issue]
            unsigned long long* tr; const size_t trn = (size_t)1024 * 2 * 64 * 4;
            CK(hipMalloc(&tr, trn * 8)); CK(hipMemset(tr, 0, trn * 8));
            a2.trace = tr; a2.stagger = -16;
            launch_v2<2, 2, 4, 2, 32, 3, EPI_GATE, 2>(a2, M, 0); CK(hipDeviceSynchronize());
            CK(hipMemset(tr, 0, trn * 8));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, 0));
            launch_v2<2, 2, 4, 2, 32, 3, EPI_GATE, 2>(a2, M, 0);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float kms; CK(hipEventElapsedTime(&kms, e0, e1));
            std::vector<unsigned long long> h(trn); CK(hipMemcpy(h.data(), tr, trn * 8, hipMemcpyDeviceToHost));
            {
                unsigned long long tmin = ~0ull, tmax = 0;
                for (int wg = 0; wg < 1024; ++wg) { const unsigned long long* t = h.data() + ((size_t)wg * 2) * 64 * 4; if (t[0]) { tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[26 * 4 + 3]); } }
                printf("trace span of the first 1024 workgroups: %llu ticks; kernel %.1f us by hipEvent => >= %.0f MHz tick rate if they span the kernel\n", tmax - tmin, kms * 1e3, (tmax - tmin) / (kms * 1e3));
            }
            for (int wg : {0, 512}) for (int wv = 0; wv < 1; ++wv) {
                const unsigned long long* t = h.data() + ((size_t)wg * 2 + wv) * 64 * 4;
                printf("trace wg %4d wave %d: start %llu\n   chunk: [mfma0+reads -> vmwait -> barrier -> dma-issue -> (mfma1+reads)]\n", wg, wv ? 5 : 0, t[0]);
                for (int ch = 0; ch < 27; ++ch) {
                    const unsigned long long prev = ch ? t[(ch - 1) * 4 + 3] : t[0];
                    printf("   %2d: +%5llu | vm %5llu | bar %5llu | dma %5llu   (t=%llu)\n", ch, t[ch * 4] - prev, t[ch * 4 + 1] - t[ch * 4], t[ch * 4 + 2] - t[ch * 4 + 1], t[ch * 4 + 3] - t[ch * 4 + 2], t[ch * 4 + 3] - t[0]);
                }
            }
            // per-CU kernel duration in shader ticks vs wall time => the clock the kernel actually ran at
            for (int dbgf : {32, 33}) {
                CK(hipMemset(tr, 0, trn * 8));
                a2.stagger = -dbgf;
                launch_v2<2, 2, 4, 2, 32, 3, EPI_GATE, 3>(a2, M, 0); CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, 0));
                launch_v2<2, 2, 4, 2, 32, 3, EPI_GATE, 3>(a2, M, 0);
                CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&kms, e0, e1));
                CK(hipMemcpy(h.data(), tr, trn * 8, hipMemcpyDeviceToHost));
                std::map<unsigned long long, std::pair<unsigned long long, unsigned long long>> cu;   // (xcc, se, cu) -> (min start, max end)
                double wgsum = 0; int nwg = 0;
                for (int wg = 0; wg < 1376; ++wg) {
                    const unsigned long long* t = h.data() + (size_t)wg * 4;
                    if (!t[0]) continue;
                    const unsigned long long key = (t[3] << 32) | (t[2] & 0xff00);      // xcc | se/sh/cu bits
                    auto it = cu.find(key);
                    if (it == cu.end()) cu[key] = {t[0], t[1]}; else { it->second.first = std::min(it->second.first, t[0]); it->second.second = std::max(it->second.second, t[1]); }
                    wgsum += (double)(t[1] - t[0]); ++nwg;
                }
                double mx = 0, av = 0; for (auto& kv : cu) { const double d = (double)(kv.second.second - kv.second.first); mx = std::max(mx, d); av += d; }
                printf("clock probe (flags %d): %zu CUs seen, per-CU span avg %.0f max %.0f ticks, kernel %.1f us => %.0f MHz (by max span); mean WG duration %.0f ticks over %d WGs\n",
                       dbgf, cu.size(), av / cu.size(), mx, kms * 1e3, mx / (kms * 1e3), wgsum / nwg, nwg);
            }
            a2.trace = nullptr; a2.stagger = 0;
        }
        for (int dbgf : {0, 1, 7, 8, 64, 68}) { a2.stagger = dbgf ? -dbgf : 0; printf("  probe flags %d (1 noDMA 2 noLDSread 4 nobarrier 8 hotDMA 64 noMFMA): ", dbgf); TRY_GATE3(2, 2, 4, 2, 32, 3, 2) }
        {   // SURVEY K3 probe (round 3): the MAIN LOOP of a 512-row tile -- what a fused gate + out-conv workgroup (all 512 gate channels of its
            // time rows, u kept on chip) would run -- against the production 256 x 128 tile, same contraction (M = 512, K = 848, sequential
            // segments, epilogue ablated).  No 512-row shape lets two workgroups share a CU: 512 x 64 needs K-chunks of 64 for its B tile to
            // split over 8 waves (2-deep ring: 144 KB), 512 x 128 with K-chunks of 32 needs 120 KB (+ the u tile of the fused kernel).
            a2.stagger = 0; a2.taps = 0;
            GemmArgs a3 = a2; a3.e.out0 = TS2; a3.e.ld_out0 = G; a3.e.out1 = nullptr; a3.e.bias = bias;
            for (int rnd = 0; rnd < 3; ++rnd) {
                float t0 = time_ms([&] { launch_v2<2, 2, 4, 2, 32, 3, EPI_STORE_BF16, 1>(a3, M, 0); });
                float t1k = time_ms([&] { launch_v2<4, 1, 4, 2, 64, 2, EPI_STORE_BF16, 1>(a3, M, 0); });
                float t3k = time_ms([&] { launch_v2<4, 2, 4, 2, 32, 3, EPI_STORE_BF16, 1>(a3, M, 0); });
                printf("K3 probe main loop only: 256x128 BK32 ring3 x2/CU %7.1f us %6.1f TF | 512x64 BK64 ring2 x1/CU %7.1f us %6.1f TF | 512x128 BK32 ring3 x1/CU %7.1f us %6.1f TF\n",
                       t0 * 1e3, fl / t0 / 1e9, t1k * 1e3, fl / t1k / 1e9, t3k * 1e3, fl / t3k / 1e9);
                // 256 x 256 tiles (one workgroup per CU, 4 DMA pieces per wave per 16 MFMAs instead of 3 per 8): wave tile 128 x 64 (2 x 4 waves) or 64 x 128 (4 x 2)
                float tq1 = time_ms([&] { launch_v2<4, 2, 2, 4, 32, 3, EPI_STORE_BF16, 1>(a3, M, 0); });
                float tq2 = time_ms([&] { launch_v2<2, 4, 4, 2, 32, 3, EPI_STORE_BF16, 1>(a3, M, 0); });
                float tq3 = time_ms([&] { launch_v2<4, 2, 2, 4, 64, 2, EPI_STORE_BF16, 1>(a3, M, 0); });
                printf("256x256 probe main loop only: wave 128x64 BK32 ring3 %7.1f us %6.1f TF | wave 64x128 BK32 ring3 %7.1f us %6.1f TF | wave 128x64 BK64 ring2 %7.1f us %6.1f TF\n",
                       tq1 * 1e3, fl / tq1 / 1e9, tq2 * 1e3, fl / tq2 / 1e9, tq3 * 1e3, fl / tq3 / 1e9);
            }
        }
#endif
        a2.stagger = 0;
    }
    {   // ---------------- out conv: M = 256, K = 256, residual add + dropout copy
        const int M = R, K = GH;
        bf16_t* Apk = dev_bf16_random((size_t)M * K, 0.05f);
        bf16_t* O1 = dev_zero<bf16_t>(NT_ * R); bf16_t* D1 = dev_zero<bf16_t>(NT_ * R);
        bf16_t* O2 = dev_zero<bf16_t>(NT_ * R); bf16_t* D2 = dev_zero<bf16_t>(NT_ * R);
        GemmArgs a; base(a, Apk, K); a.nseg = 1; a.seg[0] = mkseg(U, GH, 0, GH, 0);
        a.e.bias = bias; a.e.in0 = X; a.e.ld_in0 = R; a.e.scale = WN_SQRT_HALF; a.e.ld_out0 = R; a.e.ld_out1 = R; a.e.M_valid = M;
        a.key_lo = 0x1234567u; a.key_hi = 0x89abcdefu; a.thresh16 = 3277; a.keep_scale = 1.0f / 0.95f; a.drop_ld = R;
        GemmArgs a1 = a; a1.e.out0 = O1; a1.e.out1 = D1; GemmArgs a2 = a; a2.e.out0 = O2; a2.e.out1 = D2;
        const double fl = 2.0 * M * K * (double)NT_;
        float t1 = time_ms([&] { launch_v1<2, 2, 2, 2, EPI_STORE_BF16>(a1, M, 0); });
        printf("out    v1 128x128            : %8.1f us  %7.1f TF\n", t1 * 1e3, fl / t1 / 1e9);
        launch_v2<2, 2, 4, 2, 32, 3, EPI_STORE_BF16, 0>(a1, M, 0); CK(hipDeviceSynchronize());      // bitwise reference: see the gate section
#define TRY_OUT(WM_, WN_, BK_, NB_) { CK(hipMemset(O2, 0xff, NT_ * R * 2)); CK(hipMemset(D2, 0xff, NT_ * R * 2)); \
        float t2 = time_ms([&] { launch_v2<2, 2, WM_, WN_, BK_, NB_, EPI_STORE_BF16>(a2, M, 0); }); CK(hipDeviceSynchronize()); \
        bool ok = same({O1, (size_t)NT_ * R * 2}, {O2, (size_t)NT_ * R * 2}, "X") & same({D1, (size_t)NT_ * R * 2}, {D2, (size_t)NT_ * R * 2}, "XD"); \
        printf("out    v2 WM%d WN%d BK%d NBUF%d   : %8.1f us  %7.1f TF  %s\n", WM_, WN_, BK_, NB_, t2 * 1e3, fl / t2 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok; }
        TRY_OUT(4, 2, 32, 3)
#define TRY_OUTP(PP_) { CK(hipMemset(O2, 0xff, NT_ * R * 2)); CK(hipMemset(D2, 0xff, NT_ * R * 2)); \
        float t2 = time_ms([&] { launch_v2<2, 2, 4, 2, 32, 3, EPI_STORE_BF16, PP_>(a2, M, 0); }); CK(hipDeviceSynchronize()); \
        bool ok = same({O1, (size_t)NT_ * R * 2}, {O2, (size_t)NT_ * R * 2}, "X") & same({D1, (size_t)NT_ * R * 2}, {D2, (size_t)NT_ * R * 2}, "XD"); \
        printf("out    v2 PIPE%d   : %8.1f us  %7.1f TF  %s\n", PP_, t2 * 1e3, fl / t2 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok; }
        for (int rnd = 0; rnd < 2; ++rnd) { TRY_OUTP(1) TRY_OUTP(6) TRY_OUTP(7) }
        // 64-row time tiles (NT = 1): twice the workgroups, half the work each (balance of the HBM-bound launches at half-batch size)
#define TRY_OUT64(NB_) { CK(hipMemset(O2, 0xff, NT_ * R * 2)); CK(hipMemset(D2, 0xff, NT_ * R * 2)); \
        float t2 = time_ms([&] { launch_v2<2, 1, 4, 2, 64, NB_, EPI_STORE_BF16, 1>(a2, M, 0); }); CK(hipDeviceSynchronize()); \
        bool ok = same({O1, (size_t)NT_ * R * 2}, {O2, (size_t)NT_ * R * 2}, "X") & same({D1, (size_t)NT_ * R * 2}, {D2, (size_t)NT_ * R * 2}, "XD"); \
        printf("out    v2 256x64 BK64 NBUF%d : %8.1f us  %7.1f TF  %s\n", NB_, t2 * 1e3, fl / t2 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok; }
        for (int rnd = 0; rnd < 2; ++rnd) { TRY_OUTP(1) TRY_OUT64(2) }
    }
    {   // ---------------- skip sum: M = 256, K = L*256 via nrep
        const int M = S, K = L * GH;
        bf16_t* Apk = dev_bf16_random((size_t)M * K, 0.05f);
        bf16_t* O1 = dev_zero<bf16_t>(NT_ * S); bf16_t* O2 = dev_zero<bf16_t>(NT_ * S);
        GemmArgs a; base(a, Apk, K); a.nseg = 1; a.seg[0] = mkseg(U, GH, 0, GH, 0); a.nrep = L; a.rep_stride = NT_ * GH;
        a.e.bias = bias; a.e.relu = 1; a.e.ld_out0 = S; a.e.M_valid = M;
        GemmArgs a1 = a; a1.e.out0 = O1; GemmArgs a2 = a; a2.e.out0 = O2;
        const double fl = 2.0 * M * K * (double)NT_;
        float t1 = time_ms([&] { launch_v1<2, 2, 2, 2, EPI_STORE_BF16>(a1, M, 0); });
        printf("skip   v1 128x128            : %8.1f us  %7.1f TF\n", t1 * 1e3, fl / t1 / 1e9);
        launch_v2<2, 2, 4, 2, 32, 3, EPI_STORE_BF16, 0>(a1, M, 0); CK(hipDeviceSynchronize());      // bitwise reference: see the gate section
        { float t2 = time_ms([&] { launch_v2<2, 2, 4, 2, 64, 3, EPI_STORE_BF16>(a2, M, 0); }); CK(hipDeviceSynchronize());
          bool ok = same({O1, (size_t)NT_ * S * 2}, {O2, (size_t)NT_ * S * 2}, "R1");
          printf("skip   v2 WM4 WN2 BK64 NBUF3   : %8.1f us  %7.1f TF  %s\n", t2 * 1e3, fl / t2 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok; }
        { float t2 = time_ms([&] { launch_v2<2, 2, 4, 2, 32, 3, EPI_STORE_BF16>(a2, M, 0); }); CK(hipDeviceSynchronize());
          bool ok = same({O1, (size_t)NT_ * S * 2}, {O2, (size_t)NT_ * S * 2}, "R1");
          printf("skip   v2 WM4 WN2 BK32 NBUF3   : %8.1f us  %7.1f TF  %s\n", t2 * 1e3, fl / t2 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok; }
    }
    {   // ---------------- dx: M = 256, K = 3*512 (taps at +2d, +d, 0), dropout mask + residual-gradient add
        const int M = R, K = 3 * G, d = 128;
        bf16_t* Apk = dev_bf16_random((size_t)M * K, 0.05f);
        bf16_t* O1 = dev_zero<bf16_t>(NT_ * R); bf16_t* O2 = dev_zero<bf16_t>(NT_ * R);
        GemmArgs a; base(a, Apk, K); a.nseg = 3;
        a.seg[0] = mkseg(DZ, G, 0, G, 2 * d); a.seg[1] = mkseg(DZ, G, 0, G, d); a.seg[2] = mkseg(DZ, G, 0, G, 0);
        a.key_lo = 0x1234567u; a.key_hi = 0x89abcdefu; a.thresh16 = 3277; a.keep_scale = 1.0f / 0.95f; a.drop_ld = R;
        a.e.in0 = X; a.e.ld_in0 = R; a.e.scale = WN_SQRT_HALF; a.e.ld_out0 = R; a.e.M_valid = M;
        GemmArgs a1 = a; a1.e.out0 = O1; GemmArgs a2 = a; a2.e.out0 = O2;
        const double fl = 2.0 * M * K * (double)NT_;
        float t1 = time_ms([&] { launch_v1<2, 2, 2, 2, EPI_DX>(a1, M, 0); });
        printf("dx     v1 128x128            : %8.1f us  %7.1f TF\n", t1 * 1e3, fl / t1 / 1e9);
#define TRY_DXP(PP_) { CK(hipMemset(O2, 0xff, NT_ * R * 2)); float t2 = time_ms([&] { launch_v2<2, 2, 4, 2, 32, 3, EPI_DX, PP_>(a2, M, 0); }); CK(hipDeviceSynchronize()); \
          bool ok = same({O1, (size_t)NT_ * R * 2}, {O2, (size_t)NT_ * R * 2}, "GX"); \
          printf("dx     v2 PIPE%d   : %8.1f us  %7.1f TF  %s\n", PP_, t2 * 1e3, fl / t2 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok; }
        for (int rnd = 0; rnd < 2; ++rnd) { TRY_DXP(1) TRY_DXP(6) TRY_DXP(7) }
        for (int rnd = 0; rnd < 3; ++rnd) {      // production configuration (timing only)
            float t0 = time_ms([&] { launch_prod<EPI_DX, 0>(a2, M, 0); }); float t1p = time_ms([&] { launch_prod<EPI_DX, 1>(a2, M, 0); });
            printf("dx     production TAPS3 xcd_span: PIPE0 %8.1f us %7.1f TF   PIPE1 %8.1f us %7.1f TF\n", t0 * 1e3, fl / t0 / 1e9, t1p * 1e3, fl / t1p / 1e9);
        }
        {   // A-direct vs production, same (K-interleaved reading of the) pack: bitwise
            bf16_t* O3 = dev_zero<bf16_t>(NT_ * R); GemmArgs a3 = a; a3.e.out0 = O3;
            launch_prod<EPI_DX, 1>(a2, M, 0); CK(hipDeviceSynchronize());
            for (int rnd = 0; rnd < 3; ++rnd) {
                CK(hipMemset(O3, 0xff, NT_ * R * 2));
                float tp = time_ms([&] { launch_prod<EPI_DX, 1>(a2, M, 0); }); float ta = time_ms([&] { launch_prod_ad<EPI_DX>(a3, M, 0); }); CK(hipDeviceSynchronize());
                bool ok = same({O2, (size_t)NT_ * R * 2}, {O3, (size_t)NT_ * R * 2}, "GX");
                printf("dx     A-direct 8x1 waves: production %8.1f us %7.1f TF   A-direct %8.1f us %7.1f TF  %s\n", tp * 1e3, fl / tp / 1e9, ta * 1e3, fl / ta / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok;
                CK(hipMemset(O3, 0xff, NT_ * R * 2));
                float t16 = time_ms([&] { launch_prod16<EPI_DX>(a3, M, 0); }); CK(hipDeviceSynchronize());
                ok = same({O2, (size_t)NT_ * R * 2}, {O3, (size_t)NT_ * R * 2}, "GX");
                printf("dx     16 waves 256x256     : %8.1f us %7.1f TF  %s\n", t16 * 1e3, fl / t16 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok;
#ifdef WN_EPI_ABLATE
                { GemmArgs ap2 = a2; float tq = time_ms([&] { GemmArgs q = ap2; q.mblocks = M / 256; q.tiles_per_utt = cdiv(q.T, 128); q.ntiles = q.tiles_per_utt * q.B; q.xcd_span = cdiv(q.ntiles, 8); q.taps = 3; q.stagger = -128;
                      hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI_DX, 1, 3>), dim3(cdiv(q.ntiles, 8) * q.mblocks * 8), dim3(512), 0, 0, q); });
                  printf("dx     probe 128 (2 DMAs per wave and chunk instead of 3, sums wrong): %8.1f us %7.1f TF\n", tq * 1e3, fl / tq / 1e9); }
#endif
            }
        }
        { float t2 = time_ms([&] { launch_v2<2, 2, 4, 2, 32, 3, EPI_DX>(a2, M, 0); }); CK(hipDeviceSynchronize());
          bool ok = same({O1, (size_t)NT_ * R * 2}, {O2, (size_t)NT_ * R * 2}, "GX");
          printf("dx     v2 WM4 WN2 BK32 NBUF3   : %8.1f us  %7.1f TF  %s\n", t2 * 1e3, fl / t2 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok; }
    }
    {   // ---------------- dgate: M = 512, K = 256 + 256 (two segments), gate derivative epilogue
        const int M = GH, K = R + S;
        bf16_t* Apk = dev_bf16_random((size_t)M * K, 0.05f);
        bf16_t* O1 = dev_zero<bf16_t>(NT_ * G); bf16_t* O2 = dev_zero<bf16_t>(NT_ * G);
        GemmArgs a; base(a, Apk, K); a.nseg = 2; a.seg[0] = mkseg(X, R, 0, R, 0); a.seg[1] = mkseg(XD, S, 0, S, 0);
        a.e.in0 = TSin; a.e.in1 = U; a.e.ld_in0 = GH; a.e.ld_out0 = G; a.e.M_valid = M;      // sigmoid + u of the forward
        GemmArgs a1 = a; a1.e.out0 = O1; GemmArgs a2 = a; a2.e.out0 = O2;
        const double fl = 2.0 * M * K * (double)NT_;
        float t1 = time_ms([&] { launch_v1<2, 2, 2, 2, EPI_DGATE>(a1, M, 0); });
        printf("dgate  v1 128x128            : %8.1f us  %7.1f TF\n", t1 * 1e3, fl / t1 / 1e9);
#define TRY_DGP(PP_) { CK(hipMemset(O2, 0xff, NT_ * G * 2)); float t2 = time_ms([&] { launch_v2<2, 2, 4, 2, 32, 3, EPI_DGATE, PP_>(a2, M, 0); }); CK(hipDeviceSynchronize()); \
          bool ok = same({O1, (size_t)NT_ * G * 2}, {O2, (size_t)NT_ * G * 2}, "DZ"); \
          printf("dgate  v2 PIPE%d   : %8.1f us  %7.1f TF  %s\n", PP_, t2 * 1e3, fl / t2 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok; }
        for (int rnd = 0; rnd < 2; ++rnd) { TRY_DGP(1) TRY_DGP(6) TRY_DGP(7) }
#define TRY_DG64(NB_) { CK(hipMemset(O2, 0xff, NT_ * G * 2)); float t2 = time_ms([&] { launch_v2<2, 1, 4, 2, 64, NB_, EPI_DGATE, 1>(a2, M, 0); }); CK(hipDeviceSynchronize()); \
          bool ok = same({O1, (size_t)NT_ * G * 2}, {O2, (size_t)NT_ * G * 2}, "DZ"); \
          printf("dgate  v2 256x64 BK64 NBUF%d : %8.1f us  %7.1f TF  %s\n", NB_, t2 * 1e3, fl / t2 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok; }
        for (int rnd = 0; rnd < 2; ++rnd) { TRY_DGP(1) TRY_DG64(2) }
        { float t2 = time_ms([&] { launch_v2<2, 2, 4, 2, 32, 3, EPI_DGATE>(a2, M, 0); }); CK(hipDeviceSynchronize());
          bool ok = same({O1, (size_t)NT_ * G * 2}, {O2, (size_t)NT_ * G * 2}, "DZ");
          printf("dgate  v2 WM4 WN2 BK32 NBUF3   : %8.1f us  %7.1f TF  %s\n", t2 * 1e3, fl / t2 / 1e9, ok ? "bitwise-ok" : "FAIL"); fails += !ok; }
    }
    printf("harness %s (%d failing variants)\n", fails ? "FAILED" : "passed", fails);
    return fails != 0;
}
