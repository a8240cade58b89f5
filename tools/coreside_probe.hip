// Can an HBM-bound launch run BESIDE an MFMA-bound gate GEMM on the same CUs, or only INSTEAD of it?  (round 6)
//
// The live step's ablations (profiles/r9a_ab_ablate.txt) say the HBM-bound launches of the layer chain cost the step their full exclusive time: the
// out convs 1.09 ms, the d z launches 1.56 ms of a 9.74 ms step -- the second stream buys no overlap.  Resource arithmetic says why: two resident gate
// workgroups of the ring kernel take a CU's whole register file (16 waves x 128 VGPRs) and 144 of its 160 KB of LDS, the out conv needs 80 KB: a CU runs
// one kind OR the other.  The 8-phase gate kernel (csrc/wn_tile8p.h) leaves 88 registers per lane and SIMD, 24 wave slots and 27 KB of LDS free.  This probe
// runs each gate kernel on one stream and a LEAN streaming kernel (256 threads, no LDS, < 64 VGPRs) that moves an out conv's bytes (2 rows read, 2 rows
// written per time row) on another, and compares the pair with its parts:
//   pair ~ max(parts): HBM-bound work can hide under the matrix kernel if it is written to fit beside it (build the lean kernels);
//   pair ~ sum(parts): it cannot, whatever its footprint (do not).
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I tacotron-2_amd/csrc tools/coreside_probe.hip -o tools/coreside_probe
#include "wn_tile8p.h"
#include <vector>
#include <random>
#include <functional>

std::string g_create_err;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
static std::mt19937 rng(99);
static bf16_t* dev_bf16_random(size_t n, float scale) {
    std::vector<bf16_t> h(n); std::uniform_real_distribution<float> d(-scale, scale); for (auto& v : h) v = f2bf(d(rng));
    bf16_t* p; CK(hipMalloc(&p, n * 2)); CK(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice)); return p;
}
static std::vector<bf16_t> pack_frag(const std::vector<bf16_t>& W, int M, int K, int nk_tap, int kil) {
    std::vector<int> kmap(K); int kp = 0;
    for (int kb = 0; kb < nk_tap / kil; ++kb) for (int j = 0; j < 3; ++j) for (int i = 0; i < kil; ++i) kmap[kp++] = j * nk_tap + kb * kil + i;
    for (int k = 3 * nk_tap; k < K; ++k) kmap[kp++] = k;
    const int KS = K / 16; std::vector<bf16_t> out((size_t)M * K);
    for (int mt = 0; mt < M / 32; ++mt) for (int ks = 0; ks < KS; ++ks) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j)
        out[(((size_t)mt * KS + ks) * 64 + lane) * 8 + j] = W[(size_t)(mt * 32 + (lane & 31)) * K + kmap[ks * 16 + (lane >> 5) * 8 + j]];
    return out;
}
static float time_ms(const std::function<void()>& f, int iters = 10) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); return ms / iters;
}

// lean streaming kernel: out[i] = a[i] + b[i] (two row streams in), out2[i] = out[i] * 1.0526 (two row streams out); 16 B per lane, UNROLL requests in flight
template <int UNROLL, int LDS_BYTES>
__global__ __launch_bounds__(256) void lean_stream_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o, uint4* __restrict__ o2, int64_t n16, int64_t per_wg) {
    extern __shared__ char dummy_lds[];
    if (LDS_BYTES > 0 && threadIdx.x == 0) dummy_lds[0] = 0;
    const int64_t base = (int64_t)blockIdx.x * per_wg;
    for (int64_t i0 = base + threadIdx.x; i0 < base + per_wg; i0 += 256 * UNROLL) {
        uint4 va[UNROLL], vb[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { const int64_t i = i0 + u * 256; if (i < n16) { va[u] = a[i]; vb[u] = b[i]; } }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t i = i0 + u * 256;
            if (i < n16) {
                uint4 r, r2; const uint32_t* pa = &va[u].x; const uint32_t* pb = &vb[u].x; uint32_t* pr = &r.x; uint32_t* pr2 = &r2.x;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float x0 = bf2f((bf16_t)(pa[k] & 0xffff)) + bf2f((bf16_t)(pb[k] & 0xffff)), x1 = bf2f((bf16_t)(pa[k] >> 16)) + bf2f((bf16_t)(pb[k] >> 16));
                    pr[k] = pack_bf2(x0, x1); pr2[k] = pack_bf2(x0 * 1.0526f, x1 * 1.0526f);
                }
                o[i] = r; o2[i] = r2;
            }
        }
    }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, T = argc > 2 ? atoi(argv[2]) : 11000, rounds = argc > 3 ? atoi(argv[3]) : 3;
    const int R = 256, G = 512, GH = 256, C = 80, d = 64;
    const int64_t NT_ = (int64_t)B * T;
    printf("co-residence probe: B %d T %d; the matrix kernels run a HALF batch (%lld rows), the streaming kernel moves the other half's out-conv bytes\n", B, T, (long long)(NT_ / 2));
    bf16_t* zero; CK(hipMalloc(&zero, 4096)); CK(hipMemset(zero, 0, 4096));
    bf16_t* XD = dev_bf16_random(NT_ * R, 1.0f); bf16_t* cbt = dev_bf16_random(NT_ * C, 1.0f);
    std::vector<float> hb(1024, 0.1f); float* bias; CK(hipMalloc(&bias, 4096)); CK(hipMemcpy(bias, hb.data(), 4096, hipMemcpyHostToDevice));
    const int M = G, K = 3 * R + C;
    std::vector<bf16_t> W((size_t)M * K); { std::uniform_real_distribution<float> dd(-0.05f, 0.05f); for (auto& v : W) v = f2bf(dd(rng)); }
    bf16_t *A32, *A64; { auto p = pack_frag(W, M, K, R, 32); CK(hipMalloc(&A32, p.size() * 2)); CK(hipMemcpy(A32, p.data(), p.size() * 2, hipMemcpyHostToDevice)); }
    { auto p = pack_frag(W, M, K, R, 64); CK(hipMalloc(&A64, p.size() * 2)); CK(hipMemcpy(A64, p.data(), p.size() * 2, hipMemcpyHostToDevice)); }
    bf16_t *TS, *U; CK(hipMalloc(&TS, NT_ * GH * 2)); CK(hipMalloc(&U, NT_ * GH * 2));
    auto mkseg = [](const bf16_t* b, int ld, int col0, int nk, int shift) { SrcSeg s; s.base = b; s.ld = ld; s.col0 = col0; s.nk = nk; s.shift = shift; s.dropout = 0; return s; };
    GemmArgs g; memset(&g, 0, sizeof g); g.ksteps_total = K / 16; g.nrep = 1; g.B = B / 2; g.b0 = 0; g.T = T; g.zero = zero; g.taps = 3; g.e.scale = 1.0f; g.e.GH = GH; g.nseg = 4;
    g.seg[0] = mkseg(XD, R, 0, R, -2 * d); g.seg[1] = mkseg(XD, R, 0, R, -d); g.seg[2] = mkseg(XD, R, 0, R, 0); g.seg[3] = mkseg(cbt, C, 0, C, 0);
    g.e.bias = bias; g.e.out0 = TS; g.e.ld_out0 = GH; g.e.out1 = U; g.e.ld_out1 = GH; g.e.M_valid = M;
    auto ring = [&](hipStream_t st) { GemmArgs a = g; a.Apk = A32; a.kil = 32; a.mblocks = M / 256; a.tiles_per_utt = cdiv(a.T, 128); a.ntiles = a.tiles_per_utt * a.B; a.xcd_span = cdiv(a.ntiles, 8);
        const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8; a.stagger = grid >= WN_STAGGER_MIN_GRID ? 8000 : 0;
        hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI_GATE, 1, 3>), dim3(grid), dim3(512), 0, st, a); };
    auto p8 = [&](hipStream_t st) { GemmArgs a = g; a.Apk = A64; a.kil = 64; a.mblocks = M / 256; a.tiles_per_utt = cdiv(a.T, 256); a.ntiles = a.tiles_per_utt * a.B; a.xcd_span = cdiv(a.ntiles, 8); a.stagger = 0;
        const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
        hipLaunchKernelGGL((wn_gemm8p_kernel<EPI_GATE>), dim3(grid), dim3(512), 0, st, a); };
    // the streaming side: the other half batch's out conv moves (R + R) x 2 B in and (R + R) x 2 B out per row
    const int64_t rows = NT_ - NT_ / 2, n16 = rows * R * 2 / 16;
    bf16_t* sa = dev_bf16_random(rows * R, 1.0f); bf16_t* sb = dev_bf16_random(rows * R, 1.0f); bf16_t *so, *so2; CK(hipMalloc(&so, rows * R * 2)); CK(hipMalloc(&so2, rows * R * 2));
    auto lean = [&](hipStream_t st, int wgs, int lds) {
        const int64_t per = (n16 + wgs - 1) / wgs;
        if (lds == 0) hipLaunchKernelGGL((lean_stream_kernel<4, 0>), dim3(wgs), dim3(256), 0, st, (const uint4*)sa, (const uint4*)sb, (uint4*)so, (uint4*)so2, n16, per);
        else hipLaunchKernelGGL((lean_stream_kernel<4, 1>), dim3(wgs), dim3(256), lds, st, (const uint4*)sa, (const uint4*)sb, (uint4*)so, (uint4*)so2, n16, per);
    };
    { hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)lean_stream_kernel<4, 0>)); printf("lean streaming kernel: %d regs, %zu B static LDS\n", fa.numRegs, fa.sharedSizeBytes); }
    CK(hipFuncSetAttribute((const void*)lean_stream_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const double bytes = (double)rows * R * 2 * 4;
    auto sync2 = [&] { CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); };
    for (int rnd = 0; rnd < rounds; ++rnd) {
        for (int wgs : {1024, 2048, 4096}) {
            for (int lds : {0, 24 * 1024, 80 * 1024}) {
                const float tl = time_ms([&] { lean(s2, wgs, lds); CK(hipStreamSynchronize(s2)); });
                const float tr = time_ms([&] { ring(s1); CK(hipStreamSynchronize(s1)); });
                const float t8 = time_ms([&] { p8(s1); CK(hipStreamSynchronize(s1)); });
                const float trl = time_ms([&] { ring(s1); lean(s2, wgs, lds); sync2(); });
                const float t8l = time_ms([&] { p8(s1); lean(s2, wgs, lds); sync2(); });
                // four of each back to back per stream (no host sync inside): steady-state mix
                const float crl = time_ms([&] { for (int i = 0; i < 4; ++i) { ring(s1); lean(s2, wgs, lds); } sync2(); }, 5);
                const float c8l = time_ms([&] { for (int i = 0; i < 4; ++i) { p8(s1); lean(s2, wgs, lds); } sync2(); }, 5);
                const float cr = time_ms([&] { for (int i = 0; i < 4; ++i) ring(s1); sync2(); }, 5), c8 = time_ms([&] { for (int i = 0; i < 4; ++i) p8(s1); sync2(); }, 5);
                const float cl = time_ms([&] { for (int i = 0; i < 4; ++i) lean(s2, wgs, lds); sync2(); }, 5);
                printf("lean kernel %4d WGs, %2d KB LDS: alone %5.1f us (%4.2f TB/s) | ring gate alone %5.1f, pair %5.1f (sum %5.1f) | 8-phase gate alone %5.1f, pair %5.1f (sum %5.1f) || x4 per stream: lean %6.1f | ring %6.1f, pair %6.1f | 8-phase %6.1f, pair %6.1f\n",
                       wgs, lds / 1024, tl * 1e3, bytes / tl / 1e9, tr * 1e3, trl * 1e3, (tr + tl) * 1e3, t8 * 1e3, t8l * 1e3, (t8 + tl) * 1e3, cl * 1e3, cr * 1e3, crl * 1e3, c8 * 1e3, c8l * 1e3);
            }
        }
    }
    return 0;
}
