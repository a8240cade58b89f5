// MFMA ceiling on this box: pure v_mfma_f32_32x32x16_bf16 streams, 4 independent accumulators per wave,
// W waves per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(float)(threadIdx.x & 3); b[r] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}
template <int NACC> void run(int wg_per_cu, const char* name) {
    float* out; hipMalloc(&out, 4);
    const int iters = 20000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, 256>>>(out, 100); hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC><<<grid, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)grid * 4 * iters * NACC * 32768.0;
    printf("%s acc/wave %d, waves/SIMD %d: %.2f ms  %.0f TF\n", name, NACC, wg_per_cu, ms, fl / ms / 1e9);
}
int main() { run<4>(1, "mfma"); run<4>(2, "mfma"); run<4>(4, "mfma"); run<2>(4, "mfma"); run<1>(4, "mfma"); run<8>(1, "mfma"); return 0; }
