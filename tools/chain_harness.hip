// Persistent layer-chain prototype -- FORWARD ONLY, harness only, with a kill criterion:
// >= 15 % faster than the production schedule (two free-running streams, one half batch each, one launch per GEMM) on the same layers,
// or it is dropped and written up.
//
// RESULT (MI355X, round 4; profiles/r5i_chain_harness_fenced.txt, r5j_chain_harness_unfenced.txt): bit-identical outputs, and
//     (A) production 124 - 143 us per layer | (B) one stream 137 - 160 | (C) persistent chain 337 (agent-scope fences) / 190 - 200 (no fences:
//     `chain_harness 1`, legal only while every dependency stays inside one XCD)  =>  C is +40 .. +155 % SLOWER than A.  KILLED.
// Why: an agent-scope release / acquire per tile is an L2 write-back + invalidate on gfx950 (the weights are re-fetched after every
// invalidate); both bodies in one kernel at the 128-VGPR limit of two workgroups per CU spill 40 VGPRs into the main loops (a scratch
// reload makes hipcc drain vmcnt, i.e. the LDS-DMA ring); and what the chain removes -- launch tails -- the two free-running streams
// already hide under each other's next launch (B vs A is only +5 .. 13 %).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 -I tacotron-2_amd/csrc tools/chain_harness.hip -o tools/chain_harness
//
// Workload: NL gated residual layers of the paper model (R = 256, G = 512, 80 conditioning channels, dropout 0.05), B = 8 x T = 11 000,
// dilations incl. 1024 / 2048, per layer the two launches of wn_train_fwd: gate GEMM (K-interleaved taps, EPI_GATE) and out conv
// (+ residual, + dropout copy, EPI_STORE_BF16).  The kernel BODIES are the library's own (wn_gemm_lds_body in csrc/wn_tile.h).
//
// (A) production: for each half batch on its own stream  gate(l) -> out(l) -> gate(l + 1) ...   (out conv of a half batch: 256 x 64 tiles)
// (B) one stream, whole batch per launch (what WN_BATCH_PARTS=1 runs)
// (C) PERSISTENT CHAIN: ONE launch of 2 x 256 workgroups that stay resident and pull tile tasks from per-XCD ticket counters in launch
//     order  gate(0), out(0), gate(1), ...  over the WHOLE batch.  A task waits for exactly the tiles it reads -- gate(l, tile): the <= 6
//     tiles of out(l - 1) its three dilated taps touch (rows t0 - 2d .. t0 + 127); out(l, tile): both M-blocks of gate(l, tile) -- through
//     per-tile completion counters (release: workgroup barrier + agent-scope fence + atomic add; acquire: atomic load + agent-scope fence,
//     i.e. L2 write-back / invalidate across XCDs as the gfx942+ memory model prescribes).  No launch boundary, no drain between layers:
//     tiles of layer l + 1 start while the last tiles of layer l are still running, and the MFMA-bound gate tiles and HBM-bound out-conv
//     tiles of neighbouring layers share the CUs.  Tickets are taken in topological order per XCD, only EARLIER tickets are ever waited
//     for, so a grid smaller than the task list cannot deadlock.  Tile -> XCD mapping is the library's (contiguous tile spans per XCD).
#include "wn_tile.h"
#include <vector>
#include <random>
#include <algorithm>
#include <unistd.h>
std::string g_create_err;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
static std::mt19937 rng(7);
static bf16_t* dev_bf16_random(size_t n, float scale) {
    std::vector<bf16_t> h(n); std::uniform_real_distribution<float> d(-scale, scale);
    for (auto& v : h) v = f2bf(d(rng));
    bf16_t* p; CK(hipMalloc(&p, n * 2)); CK(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice)); return p;
}

#define MAXL 8
struct ChainArgs {
    GemmArgs* args;              // [2 NL]: gate(0), out(0), gate(1), ...
    int32_t nl;
    int32_t per_xcd_prefix[2 * MAXL + 1];      // tickets of one XCD before launch i
    int32_t* ticket;             // [8] (one cache line apart: stride 32 ints)
    int32_t* done;               // [2 NL][tiles]: gate: number of M-blocks finished (2 = ready); out: 1
    int32_t tiles;               // time tiles of the whole batch (128 rows each)
    int32_t tiles_per_utt;
    int32_t dil[MAXL];
    int32_t* abort_flag;
    unsigned long long* t_first; unsigned long long* t_last;
    int32_t flags;               // debug: 1 no fences, 2 no body
    int32_t n_per_xcd;           // resident workgroups per XCD (grid / 8): the ticket stride of a workgroup
    const int32_t* always;       // a word that satisfies every wait (idle poll lanes)
    int32_t* dummy;              // [64][16] sink of the idle lanes' atomics
};

__device__ __forceinline__ int ld_acq(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Control flow inside the task loop is kept WAVE-UNIFORM on purpose: with `if (tid == 0)` regions (ticket atomics, flag polls, the
// release) hipcc's structuriser treats the loop as divergent and moves the tail of an iteration out of the inner loop for the lanes
// that took the branch -- wave 0 then executes more s_barriers than waves 1..7 and every workgroup deadlocks in its first task
// (measured: gpurun_out r5f-r5h).  So: static round-robin tickets (workgroup i of an XCD takes tickets i, i + n, ...: still only EARLIER
// tickets are ever waited for), polls by all 64 lanes of wave 0 with a ballot as the loop condition, and per-lane atomics whose idle
// lanes hit a dummy line.
__global__ __launch_bounds__(512, 4) void chain_kernel(const ChainArgs ca) {
    using Cfg = LdsGemmCfg<2, 2, 4, 2, 32, 3>;
    __shared__ __attribute__((aligned(1024))) char lds[Cfg::LDS_BYTES];
    __shared__ int s_first;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int xcd; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcd)); xcd &= 7;
    // my index among the workgroups of this XCD (once, before the loop), and how many there are (the host launches 8 * n_per_xcd)
    if (tid == 0) { s_first = atomicAdd(ca.ticket + xcd * 32, 1); atomicMin(ca.t_first, (unsigned long long)wall_clock64()); }
    __syncthreads();
    const int my = __builtin_amdgcn_readfirstlane(s_first);
    const int total = ca.per_xcd_prefix[2 * ca.nl];
    int li = 0;
    for (int k = my; k < total; k += ca.n_per_xcd) {
        while (k >= ca.per_xcd_prefix[li + 1]) ++li;
        const int q = k - ca.per_xcd_prefix[li];
        const GemmArgs& a = ca.args[li];
        const int tile = xcd * a.xcd_span + q / a.mblocks;
        const bool is_gate = (li & 1) == 0;
        const int l = li >> 1;
        if (tile >= ca.tiles) continue;                                   // (wave-uniform: padding of the last span)
        if (wave == 0) {
            // ---- wait for the producers of exactly the rows this tile reads: lanes 0..5 = first / last row of the three taps (gate),
            // lane 0 = both M-blocks of the gate tile (out conv); the other lanes watch a word that is always satisfied
            const int32_t* flag = ca.always; int need = 1;
            const int bl = tile / ca.tiles_per_utt, tt = tile - bl * ca.tiles_per_utt;
            const int r0 = tt * 128 - (2 - (lane >> 1)) * ca.dil[l] + (lane & 1) * 127;
            const bool gate_dep = is_gate && l > 0 && lane < 6 && r0 >= 0;
            const bool out_dep = !is_gate && lane == 0;
            if (gate_dep) flag = ca.done + (size_t)(2 * (l - 1) + 1) * ca.tiles + bl * ca.tiles_per_utt + min(r0 >> 7, ca.tiles_per_utt - 1);
            if (out_dep) { flag = ca.done + (size_t)(li - 1) * ca.tiles + tile; need = 2; }
            int spins = 0;
            for (;;) {
                const bool ok = ld_acq(flag) >= need;
                if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;         // wave-uniform exit
                __builtin_amdgcn_s_sleep(4);
                if (++spins > 200000) { __hip_atomic_store(ca.abort_flag, 1 + li, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __syncthreads();
        if (!(ca.flags & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // agent scope: this workgroup's loads see the producers' stores
        if (!(ca.flags & 2)) {
            if (is_gate) wn_gemm_lds_body<2, 2, 4, 2, 32, 3, EPI_GATE, 1, 3>(a, lds, q * 8 + xcd);
            else wn_gemm_lds_body<2, 2, 4, 2, 32, 3, EPI_STORE_BF16, 1, 0>(a, lds, q * 8 + xcd);
        }
        __syncthreads();                                                  // every wave's stores are issued and acknowledged (vmcnt(0))
        if (wave == 0) {
            if (!(ca.flags & 1)) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // agent scope: L2 write-back towards the other XCDs
            int32_t* dst = lane == 0 ? ca.done + (size_t)li * ca.tiles + tile : ca.dummy + lane * 16;      // one atomic per lane, 63 of them on dummy lines
            __hip_atomic_fetch_add(dst, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) atomicMax(ca.t_last, (unsigned long long)wall_clock64());
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int B = 8, T = 11000, R = 256, G = 512, GH = 256, C = 80;
    const int NL = 6; const int dil[NL] = {1, 8, 128, 512, 1024, 2048};
    const int64_t NT = (int64_t)B * T;
    const float pdrop = 0.05f;
    bf16_t* zero; CK(hipMalloc(&zero, 256)); CK(hipMemset(zero, 0, 256));
    // X[l], XD[l] (dropout copy: the gate's input), U[l], TS[l]; three result sets (A, B, C) of X / XD for the comparison
    bf16_t* X0 = dev_bf16_random((size_t)NT * R, 1.0f);
    bf16_t* cbt = dev_bf16_random((size_t)NT * C, 1.0f);
    auto alloc = [&](size_t n) { bf16_t* p; CK(hipMalloc(&p, n * 2)); CK(hipMemset(p, 0, n * 2)); return p; };
    bf16_t* X = alloc((size_t)(NL + 1) * NT * R); bf16_t* XD = alloc((size_t)(NL + 1) * NT * R);
    bf16_t* U = alloc((size_t)NL * NT * GH); bf16_t* TS = alloc((size_t)NL * NT * GH);
    CK(hipMemcpy(X, X0, (size_t)NT * R * 2, hipMemcpyDeviceToDevice)); CK(hipMemcpy(XD, X0, (size_t)NT * R * 2, hipMemcpyDeviceToDevice));
    bf16_t* W1[NL]; bf16_t* Wo[NL];
    for (int l = 0; l < NL; ++l) { W1[l] = dev_bf16_random((size_t)G * (3 * R + C), 0.03f); Wo[l] = dev_bf16_random((size_t)R * GH, 0.05f); }
    float* bias; CK(hipMalloc(&bias, 8192)); CK(hipMemset(bias, 0, 8192));
    auto mkseg = [](const bf16_t* b, int ld, int nk, int shift) { SrcSeg s; s.base = b; s.ld = ld; s.col0 = 0; s.nk = nk; s.shift = shift; s.dropout = 0; return s; };
    auto mk_gate = [&](int l, int b0, int nb) {
        GemmArgs a; memset(&a, 0, sizeof a); a.Apk = W1[l]; a.ksteps_total = (3 * R + C) / 16; a.nrep = 1; a.B = nb; a.T = T; a.b0 = b0; a.zero = zero; a.e.scale = 1.0f; a.e.GH = GH; a.e.M_valid = G;
        const bf16_t* x = XD + (size_t)l * NT * R; const int d = dil[l];
        a.nseg = 4; a.seg[0] = mkseg(x, R, R, -2 * d); a.seg[1] = mkseg(x, R, R, -d); a.seg[2] = mkseg(x, R, R, 0); a.seg[3] = mkseg(cbt, C, C, 0); a.taps = 3;
        a.e.bias = bias; a.e.out0 = TS + (size_t)l * NT * GH; a.e.ld_out0 = GH; a.e.out1 = U + (size_t)l * NT * GH; a.e.ld_out1 = GH;
        return a;
    };
    auto mk_out = [&](int l, int b0, int nb) {
        GemmArgs a; memset(&a, 0, sizeof a); a.Apk = Wo[l]; a.ksteps_total = GH / 16; a.nrep = 1; a.B = nb; a.T = T; a.b0 = b0; a.zero = zero; a.e.scale = 0.70710678f; a.e.GH = GH; a.e.M_valid = R;
        a.nseg = 1; a.seg[0] = mkseg(U + (size_t)l * NT * GH, GH, GH, 0);
        a.e.bias = bias; a.e.in0 = X + (size_t)l * NT * R; a.e.ld_in0 = R;
        a.e.out0 = X + (size_t)(l + 1) * NT * R; a.e.ld_out0 = R; a.e.out1 = XD + (size_t)(l + 1) * NT * R; a.e.ld_out1 = R;
        wn_layer_key(1234, l + 1, &a.key_lo, &a.key_hi); a.thresh16 = (uint32_t)lrintf(pdrop * 65536.0f); a.keep_scale = 1.0f / (1.0f - pdrop); a.drop_ld = R;
        return a;
    };
    // the library's grid decode for a 256 x TT tile launch
    auto prep = [&](GemmArgs& a, int M, int TT) {
        a.mblocks = M / 256; a.tiles_per_utt = cdiv(a.T, TT); a.ntiles = a.tiles_per_utt * a.B; a.xcd_span = cdiv(a.ntiles, 8);
        const int grid = cdiv(a.ntiles, 8) * a.mblocks * 8;
        a.stagger = grid >= WN_STAGGER_MIN_GRID ? 8000 : 0;
        return grid;
    };
    auto launch_gate = [&](int l, int b0, int nb, hipStream_t st) { GemmArgs a = mk_gate(l, b0, nb); const int grid = prep(a, G, 128);
        hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI_GATE, 1, 3>), dim3(grid), dim3(512), 0, st, a); };
    auto launch_out = [&](int l, int b0, int nb, hipStream_t st) { GemmArgs a = mk_out(l, b0, nb);
        if ((int64_t)cdiv(T, 128) * nb < 512) { const int grid = prep(a, R, 64); hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 1, 4, 2, 64, 2, EPI_STORE_BF16, 1, 0>), dim3(grid), dim3(512), 0, st, a); }
        else { const int grid = prep(a, R, 128); hipLaunchKernelGGL((wn_gemm_lds_kernel<2, 2, 4, 2, 32, 3, EPI_STORE_BF16, 1, 0>), dim3(grid), dim3(512), 0, st, a); } };

    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t e0, e1, ef; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ef));
    auto run_A = [&]() {            // production: two free-running streams, half a batch each
        CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0));
        for (int l = 0; l < NL; ++l) for (int p = 0; p < 2; ++p) { hipStream_t st = p ? s1 : s0; launch_gate(l, p * (B / 2), B / 2, st); launch_out(l, p * (B / 2), B / 2, st); }
        CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s0, ef, 0));
    };
    auto run_B = [&]() { for (int l = 0; l < NL; ++l) { launch_gate(l, 0, B, s0); launch_out(l, 0, B, s0); } };

    // ---- persistent chain state
    ChainArgs ca; memset(&ca, 0, sizeof ca);
    std::vector<GemmArgs> hargs(2 * NL);
    ca.nl = NL; ca.per_xcd_prefix[0] = 0; ca.flags = argc > 1 ? atoi(argv[1]) : 0;
    printf("debug flags %d\n", ca.flags);
    for (int l = 0; l < NL; ++l) {
        hargs[2 * l] = mk_gate(l, 0, B); int g0 = prep(hargs[2 * l], G, 128); hargs[2 * l].stagger = 0;
        hargs[2 * l + 1] = mk_out(l, 0, B); int g1 = prep(hargs[2 * l + 1], R, 128); hargs[2 * l + 1].stagger = 0;
        ca.per_xcd_prefix[2 * l + 1] = ca.per_xcd_prefix[2 * l] + g0 / 8;
        ca.per_xcd_prefix[2 * l + 2] = ca.per_xcd_prefix[2 * l + 1] + g1 / 8;
        ca.dil[l] = dil[l];
    }
    ca.tiles = hargs[0].ntiles; ca.tiles_per_utt = hargs[0].tiles_per_utt;
    CK(hipMalloc(&ca.args, sizeof(GemmArgs) * 2 * NL)); CK(hipMemcpy(ca.args, hargs.data(), sizeof(GemmArgs) * 2 * NL, hipMemcpyHostToDevice));
    const size_t state_ints = 8 * 32 + (size_t)2 * NL * ca.tiles + 32 + 64 * 16;
    int32_t* state; CK(hipMalloc(&state, state_ints * 4 + 64));
    ca.ticket = state; ca.done = state + 8 * 32; ca.abort_flag = state + 8 * 32 + (size_t)2 * NL * ca.tiles; ca.dummy = ca.abort_flag + 32;
    int32_t* always_dev; CK(hipMalloc(&always_dev, 64)); { int32_t big = 1 << 30; CK(hipMemcpy(always_dev, &big, 4, hipMemcpyHostToDevice)); } ca.always = always_dev;
    unsigned long long* stamps; CK(hipMalloc(&stamps, 16)); ca.t_first = stamps; ca.t_last = stamps + 1;
    int nres = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nres, chain_kernel, 512, 0));
    const int grid_c = 256 * std::max(1, nres); ca.n_per_xcd = grid_c / 8;
    printf("persistent chain: %d layers, %d tiles, %d tickets per XCD, %d workgroups resident per CU -> grid %d\n", NL, ca.tiles, ca.per_xcd_prefix[2 * NL], nres, grid_c);
    auto run_C = [&]() {
        CK(hipMemsetAsync(state, 0, state_ints * 4, s0));
        CK(hipMemsetAsync(stamps, 0xff, 8, s0)); CK(hipMemsetAsync(stamps + 1, 0, 8, s0));
        hipLaunchKernelGGL(chain_kernel, dim3(grid_c), dim3(512), 0, s0, ca);
    };
    // ---- results: the top layer's output of every schedule must agree with the single-stream one
    std::vector<bf16_t> ref((size_t)NT * R), got((size_t)NT * R);
    auto top = [&](std::vector<bf16_t>& v) { CK(hipDeviceSynchronize()); CK(hipMemcpy(v.data(), X + (size_t)NL * NT * R, (size_t)NT * R * 2, hipMemcpyDeviceToHost)); };
    auto clear_top = [&]() { for (int l = 1; l <= NL; ++l) { CK(hipMemset(X + (size_t)l * NT * R, 0, (size_t)NT * R * 2)); CK(hipMemset(XD + (size_t)l * NT * R, 0, (size_t)NT * R * 2)); } };
    printf("single stream ...\n"); run_B(); top(ref); printf("  done\n");
    auto compare = [&](const char* name) {
        top(got); size_t bad = 0; double md = 0;
        for (size_t i = 0; i < ref.size(); ++i) { if (ref[i] != got[i]) { ++bad; md = std::max(md, (double)fabsf(bf2f(ref[i]) - bf2f(got[i]))); } }
        printf("  %-28s top-layer output vs single stream: %zu of %zu elements differ (max |diff| %.3g)\n", name, bad, ref.size(), md);
    };
    clear_top(); run_A(); compare("two streams (production)");
    printf("persistent chain ...\n"); clear_top(); run_C();
    {   // watchdog: a hung persistent kernel is reported instead of waited for
        int waited = 0;
        while (hipStreamQuery(s0) == hipErrorNotReady && waited < 100) { usleep(100000); ++waited; }
        if (hipStreamQuery(s0) == hipErrorNotReady) { printf("  HUNG after 10 s\n"); fflush(stdout); _exit(3); }
    }
    compare("persistent chain");
    { int ab = 0; CK(hipMemcpy(&ab, ca.abort_flag, 4, hipMemcpyDeviceToHost)); if (ab) printf("  !! persistent chain: a dependency wait timed out (launch %d)\n", ab - 1); }

    const int REPS = 5;
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        auto timeit = [&](auto f, const char* name) {
            f(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s0)); for (int i = 0; i < REPS; ++i) f(); CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-44s %8.1f us per layer (gate + out conv, whole batch)\n", name, ms * 1e3 / REPS / NL);
            return ms * 1e3 / REPS / NL;
        };
        const float tb = timeit(run_B, "(B) one stream, whole batch per launch");
        const float ta = timeit(run_A, "(A) two streams, half batches [production]");
        const float tc = timeit(run_C, "(C) persistent chain, one launch");
        unsigned long long st[2]; CK(hipMemcpy(st, stamps, 16, hipMemcpyDeviceToHost));
        printf("    (C) in-kernel first start .. last end: %.1f us per layer;  C vs A: %+.1f %%   (kill criterion: <= -15 %%);  B vs A: %+.1f %%\n",
               (double)(st[1] - st[0]) / 100.0 / NL, (tc / ta - 1.0f) * 100.0f, (tb / ta - 1.0f) * 100.0f);
    }
    return 0;
}
