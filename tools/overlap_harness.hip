// Does running the two half-batches of the (serial) layer chain on two streams overlap the MFMA/power-bound gate GEMM of one
// half with the HBM-bound out conv of the other?   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I tacotron-2_amd/csrc -I tools tools/overlap_harness.hip -o tools/overlap_harness
#ifdef USE_PRODUCTION_TILE
#include "wn_tile.h"
#else
#include "wn_tile_variants.h"
#endif
#include <vector>
#include <random>
std::string g_create_err;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
static std::mt19937 rng(7);
static bf16_t* dev_bf16_random(size_t n, float scale) {
    std::vector<bf16_t> h(n); std::uniform_real_distribution<float> d(-scale, scale);
    for (auto& v : h) v = f2bf(d(rng));
    bf16_t* p; CK(hipMalloc(&p, n * 2)); CK(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice)); return p;
}
template <int EPI> static void launch(GemmArgs a, int M, hipStream_t st) {
    a.mblocks = M / 256; a.tiles_per_utt = cdiv(a.T, 128); a.ntiles = a.tiles_per_utt * a.B;
    const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
    a.stagger = grid >= 1024 ? 8000 : 0;
    hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI, 1>), dim3(grid), dim3(512), 0, st, a);
}
int main() {
    const int B = 8, T = 11000, R = 256, G = 512, GH = 256, C = 80, L = 12;
    const int64_t NT = (int64_t)B * T;
    bf16_t* zero; CK(hipMalloc(&zero, 256)); CK(hipMemset(zero, 0, 256));
    bf16_t* X = dev_bf16_random((size_t)(L + 1) * NT * R, 1.0f);
    bf16_t* cbt = dev_bf16_random(NT * C, 1.0f);
    bf16_t* TS; CK(hipMalloc(&TS, (size_t)L * NT * G * 2));
    bf16_t* U; CK(hipMalloc(&U, (size_t)L * NT * GH * 2));
    bf16_t* W1 = dev_bf16_random((size_t)G * (3 * R + C), 0.05f);
    bf16_t* Wo = dev_bf16_random((size_t)R * GH, 0.05f);
    float* bias; CK(hipMalloc(&bias, 4096)); CK(hipMemset(bias, 0, 4096));
    auto mkseg = [](const bf16_t* b, int ld, int nk, int shift) { SrcSeg s; s.base = b; s.ld = ld; s.col0 = 0; s.nk = nk; s.shift = shift; s.dropout = 0; return s; };
    auto gate = [&](int l, int b0, int nb, hipStream_t st) {
        GemmArgs a; memset(&a, 0, sizeof a); a.Apk = W1; a.ksteps_total = (3 * R + C) / 16; a.nrep = 1; a.B = nb; a.T = T; a.zero = zero; a.e.scale = 1.0f; a.e.GH = GH; a.e.M_valid = G;
        const bf16_t* x = X + ((size_t)l * NT + (size_t)b0 * T) * R; const int d = 1 << (l % 12);
        a.nseg = 4; a.seg[0] = mkseg(x, R, R, -2 * d); a.seg[1] = mkseg(x, R, R, -d); a.seg[2] = mkseg(x, R, R, 0); a.seg[3] = mkseg(cbt + (size_t)b0 * T * C, C, C, 0);
        a.e.bias = bias; a.e.out0 = TS + ((size_t)l * NT + (size_t)b0 * T) * G; a.e.ld_out0 = G; a.e.out1 = U + ((size_t)l * NT + (size_t)b0 * T) * GH; a.e.ld_out1 = GH;
        launch<EPI_GATE>(a, G, st);
    };
    auto outc = [&](int l, int b0, int nb, hipStream_t st) {
        GemmArgs a; memset(&a, 0, sizeof a); a.Apk = Wo; a.ksteps_total = GH / 16; a.nrep = 1; a.B = nb; a.T = T; a.zero = zero; a.e.scale = 1.0f; a.e.GH = GH; a.e.M_valid = R;
        a.nseg = 1; a.seg[0] = mkseg(U + ((size_t)l * NT + (size_t)b0 * T) * GH, GH, GH, 0);
        a.e.bias = bias; a.e.in0 = X + ((size_t)l * NT + (size_t)b0 * T) * R; a.e.ld_in0 = R;
        a.e.out0 = X + ((size_t)(l + 1) * NT + (size_t)b0 * T) * R; a.e.ld_out0 = R;
        launch<EPI_STORE_BF16>(a, R, st);
    };
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ef));
    auto run_seq = [&]() { for (int l = 0; l < L; ++l) { gate(l, 0, B, s0); outc(l, 0, B, s0); } };
    auto run_split = [&](int parts) {
        // fork: s1 waits for s0's current position
        CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0));
        const int nb = B / parts;
        for (int l = 0; l < L; ++l) for (int p = 0; p < parts; ++p) { hipStream_t st = (p & 1) ? s1 : s0; gate(l, p * nb, nb, st); outc(l, p * nb, nb, st); }
        CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s0, ef, 0));
    };
    hipEvent_t es; CK(hipEventCreateWithFlags(&es, hipEventDisableTiming));
    // interleaved enqueue, part 1 held back until part 0's first gate GEMM has finished: gate(p0) | out(p1) from then on
    auto run_stagger = [&]() {
        CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0));
        const int nb = B / 2;
        for (int l = 0; l < L; ++l) {
            gate(l, 0, nb, s0);
            if (l == 0) { CK(hipEventRecord(es, s0)); CK(hipStreamWaitEvent(s1, es, 0)); }
            outc(l, 0, nb, s0);
            gate(l, nb, nb, s1); outc(l, nb, nb, s1);
        }
        CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s0, ef, 0));
    };
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        run_seq(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s0)); for (int i = 0; i < 3; ++i) run_seq(); CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("one stream, full batch      : %8.1f us per layer (gate+out)\n", ms * 1e3 / 3 / L);
        run_stagger(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s0)); for (int i = 0; i < 3; ++i) run_stagger(); CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("two streams, 2 parts, staggered by one gate GEMM: %8.1f us per layer\n", ms * 1e3 / 3 / L);
        for (int parts : {2}) {
            run_split(parts); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s0)); for (int i = 0; i < 3; ++i) run_split(parts); CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); printf("two streams, %d batch parts  : %8.1f us per layer (gate+out)\n", parts, ms * 1e3 / 3 / L);
        }
    }
    return 0;
}
