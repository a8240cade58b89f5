// What does this part's memory system sustain for the traffic MIX of the HBM-bound layer launches?  (round 6)
// out conv: two row streams read (u, x), two written (x', dropout copy); d z: four read, two written -- they run at 4.3 - 4.8 TB/s, a float4 copy is quoted at
// 6.3.  Is the gap the kernels' structure or the mix?  Plain streaming kernels, nothing but loads, a trivial VALU op and stores: NIN streams in, NOUT out,
// 16 B per lane, UNROLL x NIN requests in flight per lane, persistent grid-stride.  Bytes counted = (NIN + NOUT) x n.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/hbm_streams_probe.hip -o tools/hbm_streams_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <functional>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
struct Ptrs { const uint4* in[4]; uint4* out[2]; };
template <int NIN, int NOUT, int UNROLL>
__global__ __launch_bounds__(256) void streams_kernel(Ptrs p, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * 256 * UNROLL;
    for (int64_t i0 = (int64_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i0 < n16; i0 += stride) {
        uint4 v[NIN][UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int s = 0; s < NIN; ++s) { const int64_t i = i0 + u * 256; v[s][u] = i < n16 ? p.in[s][i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            uint4 r = v[0][u];
#pragma unroll
            for (int s = 1; s < NIN; ++s) { r.x ^= v[s][u].x; r.y += v[s][u].y; r.z ^= v[s][u].z; r.w += v[s][u].w; }
            const int64_t i = i0 + u * 256;
            if (i < n16) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) { uint4 w = r; w.x += o; p.out[o][i] = w; }
            }
        }
    }
}
static float time_ms(const std::function<void()>& f, int iters = 20) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}
template <int NIN, int NOUT, int UNROLL> static void run(const Ptrs& p, int64_t n16, int wgs) {
    const float ms = time_ms([&] { hipLaunchKernelGGL((streams_kernel<NIN, NOUT, UNROLL>), dim3(wgs), dim3(256), 0, 0, p, n16); });
    printf("%d in / %d out, %d x 16 B per stream in flight per lane, %5d workgroups: %7.1f us  %5.2f TB/s\n", NIN, NOUT, UNROLL, wgs, ms * 1e3, (double)(NIN + NOUT) * n16 * 16 / ms / 1e9);
}
int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 88000; const int row_bytes = 512;
    const int64_t n16 = rows * row_bytes / 16;       // one stream = rows x 512 B (a [rows][256] bf16 tensor): 45 MB at the bench batch
    printf("stream = %lld rows x %d B = %.1f MB (the layer tensors of the C2 step); larger: pass rows\n", (long long)rows, row_bytes, rows * row_bytes / 1e6);
    Ptrs p;
    for (int s = 0; s < 4; ++s) { void* q; CK(hipMalloc(&q, n16 * 16)); CK(hipMemset(q, s + 1, n16 * 16)); p.in[s] = (const uint4*)q; }
    for (int o = 0; o < 2; ++o) { void* q; CK(hipMalloc(&q, n16 * 16)); p.out[o] = (uint4*)q; }
    for (int wgs : {1024, 2048, 4096}) {
        run<1, 1, 4>(p, n16, wgs); run<1, 1, 8>(p, n16, wgs);
        run<2, 1, 4>(p, n16, wgs); run<2, 2, 4>(p, n16, wgs); run<2, 2, 8>(p, n16, wgs);
        run<4, 2, 2>(p, n16, wgs); run<4, 2, 4>(p, n16, wgs);
        run<1, 0, 8>(p, n16, wgs); run<2, 0, 8>(p, n16, wgs);
    }
    return 0;
}
