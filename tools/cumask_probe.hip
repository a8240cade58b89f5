// CU-mask probe (run on the GPU box): which (XCC, SE, CU) does a stream created with hipExtStreamCreateWithCUMask get for a given bit
// pattern, and do two masked streams really run side by side?   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <set>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

__global__ void where(uint32_t* out, int spin) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (uint64_t)spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hwid; out[blockIdx.x * 2 + 1] = xcc; }
}
// streaming read: HBM-bound stand-in
__global__ void stream_read(const float4* __restrict__ p, size_t n, float* sink) {
    float a = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; a += v.x + v.y + v.z + v.w; }
    if (a == 12345.678f) *sink = a;
}
// register-heavy MFMA spinner: compute-bound stand-in that owns its CU (launch_bounds 512,2 + big LDS)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512, 2) void mfma_spin(float* sink, int iters) {
    __shared__ char big[120 * 1024];
    big[threadIdx.x] = 1;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b; for (int j = 0; j < 8; ++j) { a[j] = (__bf16)1.0f; b[j] = (__bf16)(threadIdx.x & 1); }
    for (int it = 0; it < iters; ++it)
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.678f) *sink = s + big[threadIdx.x];
}

static int census(hipStream_t st, uint32_t* d, const char* tag) {
    const int n = 4096;
    CK(hipMemsetAsync(d, 0xff, n * 8, st));
    hipLaunchKernelGGL(where, dim3(n), dim3(64), 0, st, d, 20000);
    CK(hipStreamSynchronize(st));
    std::vector<uint32_t> h(n * 2); CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    std::map<int, std::set<uint32_t>> per_xcc;
    for (int i = 0; i < n; ++i) { const uint32_t hw = h[i * 2]; per_xcc[h[i * 2 + 1] & 0xf].insert((hw >> 8) & 0xff | ((hw >> 13) & 0x7) << 8); }   // cu_id [11:8], sh [12], se [15:13]
    int tot = 0; printf("%-28s CUs per XCC:", tag);
    for (auto& kv : per_xcc) { printf(" x%d:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  total %d\n", tot);
    return 0;
}

int main() {
    uint32_t* d; CK(hipMalloc(&d, 4096 * 8));
    hipStream_t s0; CK(hipStreamCreate(&s0));
    census(s0, d, "unmasked");
    auto mk = [&](auto pred, hipStream_t* st) { uint32_t m[8] = {0}; for (int i = 0; i < 256; ++i) if (pred(i)) m[i >> 5] |= 1u << (i & 31); return hipExtStreamCreateWithCUMask(st, 8, m); };
    hipStream_t a, b, c2, d2;
    CK(mk([](int i) { return i < 64; }, &a));            census(a, d, "bits 0..63");
    CK(mk([](int i) { return i % 4 == 0; }, &b));        census(b, d, "every 4th bit");
    CK(mk([](int i) { return i >= 64; }, &c2));          census(c2, d, "bits 64..255");
    CK(mk([](int i) { return i % 4 != 0; }, &d2));       census(d2, d, "all but every 4th");
    // side by side: MFMA spinner on 3/4 of the CUs, streaming read on 1/4 -- vs both unmasked
    const size_t nb = (size_t)2 << 30; float4* buf; CK(hipMalloc(&buf, nb)); CK(hipMemset(buf, 0, nb)); float* sink; CK(hipMalloc(&sink, 4));
    hipStream_t u1, u2; CK(hipStreamCreate(&u1)); CK(hipStreamCreate(&u2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](hipStream_t sm, hipStream_t sr, const char* tag, int nread_blocks) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0)); CK(hipStreamWaitEvent(sm, e0, 0)); CK(hipStreamWaitEvent(sr, e0, 0));
            if (sm) hipLaunchKernelGGL(mfma_spin, dim3(512), dim3(512), 0, sm, sink, 5000);
            if (sr) for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(stream_read, dim3(nread_blocks), dim3(256), 0, sr, buf, nb / 16, sink);
            hipEvent_t em, er; CK(hipEventCreate(&em)); CK(hipEventCreate(&er));
            if (sm) CK(hipEventRecord(em, sm)); if (sr) CK(hipEventRecord(er, sr));
            CK(hipDeviceSynchronize());
            float tm = 0, tr = 0; if (sm) CK(hipEventElapsedTime(&tm, e0, em)); if (sr) CK(hipEventElapsedTime(&tr, e0, er));
            if (rep) printf("%-44s mfma done at %7.3f ms, 8 GB read done at %7.3f ms (%.2f TB/s)\n", tag, tm, tr, tr > 0 ? 8.0 * 1.0737 / tr : 0.0);
        }
        return 0;
    };
    run(u1, nullptr, "mfma alone (unmasked)", 0);
    run(nullptr, u2, "read alone (unmasked, 2048 blocks)", 2048);
    run(nullptr, b, "read alone on every-4th-bit CUs", 2048);
    run(nullptr, a, "read alone on bits 0..63", 2048);
    run(u1, u2, "both unmasked", 2048);
    run(d2, b, "mfma on 3/4, read on 1/4 (interleaved bits)", 2048);
    run(c2, a, "mfma on bits 64.., read on bits 0..63", 2048);
    printf("done\n");
    return 0;
}
