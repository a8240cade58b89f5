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
int calib();
int main() { calib(); run<4>(1, "mfma"); run<4>(2, "mfma"); run<4>(4, "mfma"); run<2>(4, "mfma"); run<1>(4, "mfma"); run<8>(1, "mfma"); return 0; }
// ---- clock calibration: what does one s_memtime tick correspond to, idle and under MFMA load?
__global__ void spin_ticks(unsigned long long n, unsigned long long* out) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < n) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k_ticks(float* out, int iters, unsigned long long* ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
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
    if (threadIdx.x == 0) ticks[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}
int calib() {
    unsigned long long* d; hipMalloc(&d, 8 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    spin_ticks<<<1, 64>>>(1000000ull, d); hipDeviceSynchronize();
    hipEventRecord(e0); spin_ticks<<<1, 64>>>(20000000ull, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("idle spin: %llu ticks in %.3f ms => %.1f MHz per tick\n", h, ms, h / (ms * 1e3));
    float* out; hipMalloc(&out, 4);
    const int iters = 20000;
    k_ticks<4><<<1024, 256>>>(out, 100, d); hipDeviceSynchronize();
    hipEventRecord(e0); k_ticks<4><<<1024, 256>>>(out, iters, d); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("mfma load (4 waves/SIMD): block 0 ran %llu ticks; kernel %.3f ms => %.1f MHz per tick; MFMA issue period %.2f ticks (ideal 32/4waves.. 8 per wave-mfma at 4 waves)\n",
           h, ms, h / (ms * 1e3), (double)h / (iters * 4.0));
    return 0;
}
