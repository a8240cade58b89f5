// Hop-latency probe for the persistent synthesis pipeline (run on the GPU box).
// STAGES stages of P workgroups (one per CU, forced by a large LDS allocation) form a ring.  Every workgroup of stage s
// waits for the P x G granules {payload32, tag32} published by stage s-1 for step t, sums them, and publishes its own G
// granules.  Stage 0 of step t+1 waits for the last stage of step t (autoregressive dependency).  Reports us per hop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ uint64_t ld_g(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_g(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ u32x4 ld_g16(const u32x4* p) { u32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st_g16(u32x4* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory"); }

// 16-byte granules {3 x payload32, tag32}
template <int P, int G>
__global__ __launch_bounds__(256) void ring16_kernel(u32x4* mbox, int stages, int steps, int* abort_flag, float* out, int spx) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x = blockIdx.x & 7, k = blockIdx.x >> 3; const int stage = spx * x + k / P, j = k % P;
    if (k >= spx * P || stage >= stages) return;
    const int prev = (stage + stages - 1) % stages;
    const u32x4* in = mbox + (size_t)prev * P * G;
    u32x4* mine = mbox + ((size_t)stage * P + j) * G;
    float acc = 0.0f;
    for (int t = 0; t < steps; ++t) {
        const uint32_t want = (stage == 0) ? (uint32_t)t : (uint32_t)(t + 1);
        float sum = 0.0f;
        if (!(stage == 0 && t == 0)) {
            int spins = 0; bool ok = false;
            while (!ok) {
                ok = true; sum = 0.0f;
                for (int p = wave; p < P; p += 4)
                    for (int g = lane; g < G; g += 64) {
                        const u32x4 v = ld_g16(in + (size_t)p * G + g);
                        if (v.w != want) ok = false;
                        sum += __uint_as_float(v.x) + __uint_as_float(v.y) + __uint_as_float(v.z);
                    }
                ok = __all(ok);
                if (++spins > 4000000) { *abort_flag = 1; ok = true; }
                if ((spins & 1023) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
            }
        }
        lds[tid] = sum;
        __syncthreads();
        float tot = 0.0f;
        for (int i = 0; i < 256; i += 64) tot += lds[i + lane];
        acc += tot;
        __syncthreads();
        const float pv = tot * 1e-3f + 1.0f;
        for (int g = tid; g < G; g += 256) { u32x4 v; v.x = v.y = v.z = __float_as_uint(pv); v.w = (uint32_t)(t + 1); st_g16(mine + g, v); }
    }
    if (tid == 0) out[blockIdx.x] = acc;
}
template <int P, int G> static void run16(int stages, int steps, int spx) {
    u32x4* mbox; int* ab; float* out;
    CK(hipMalloc(&mbox, (size_t)stages * P * G * 16)); CK(hipMemset(mbox, 0, (size_t)stages * P * G * 16));
    CK(hipMalloc(&ab, 4)); CK(hipMemset(ab, 0, 4)); CK(hipMalloc(&out, 4096));
    const int lds_bytes = 100 * 1024;
    CK(hipFuncSetAttribute((const void*)ring16_kernel<P, G>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((ring16_kernel<P, G>), dim3(8 * spx * P), dim3(256), lds_bytes, 0, mbox, stages, steps, ab, out, spx);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int h; CK(hipMemcpy(&h, ab, 4, hipMemcpyDeviceToHost));
    printf("16B P=%d G=%3d (%4d B/producer) stages=%d spx=%d: %8.2f us/step  %6.3f us/hop %s\n", P, G, G * 16, stages, spx, ms * 1e3 / steps, ms * 1e3 / steps / stages, h ? "ABORTED (timeout)" : "");
    CK(hipFree(mbox)); CK(hipFree(ab)); CK(hipFree(out));
}

template <int P, int G>
__global__ __launch_bounds__(256) void ring_kernel(uint64_t* mbox /*[stages][P][G]*/, int stages, int steps, int* abort_flag, float* out, int same_xcd_map) {
    extern __shared__ float lds[];           // forces one workgroup per CU
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int stage, j;
    if (same_xcd_map) { const int spx = same_xcd_map; const int x = blockIdx.x & 7, k = blockIdx.x >> 3; stage = spx * x + k / P; j = k % P; if (k >= spx * P) return; }   // spx stages per XCD
    else { stage = blockIdx.x / P; j = blockIdx.x % P; }
    if (stage >= stages) return;
    const int prev = (stage + stages - 1) % stages;
    const uint64_t* in = mbox + (size_t)prev * P * G;
    uint64_t* mine = mbox + ((size_t)stage * P + j) * G;
    float acc = 0.0f;
    for (int t = 0; t < steps; ++t) {
        const uint32_t want = (stage == 0) ? (uint32_t)t : (uint32_t)(t + 1);     // stage 0 consumes the last stage's step t-1 (tag t)
        float sum = 0.0f;
        if (!(stage == 0 && t == 0)) {
            // wave w polls producers w, w+4, ...; lane polls granules lane, lane+64
            int spins = 0; bool ok = false;
            while (!ok) {
                ok = true; sum = 0.0f;
                for (int p = wave; p < P; p += 4)
                    for (int g = lane; g < G; g += 64) {
                        const uint64_t v = ld_g(in + (size_t)p * G + g);
                        if ((uint32_t)(v >> 32) != want) ok = false;
                        sum += __uint_as_float((uint32_t)v);
                    }
                ok = __all(ok);
                if (++spins > 4000000) { *abort_flag = 1; ok = true; }
                if ((spins & 1023) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
            }
        }
        lds[tid] = sum;
        __syncthreads();
        float tot = 0.0f;
        for (int i = 0; i < 256; i += 64) tot += lds[i + lane];     // cheap stand-in for the matvec work
        acc += tot;
        __syncthreads();
        for (int g = tid; g < G; g += 256) st_g(mine + g, ((uint64_t)(uint32_t)(t + 1) << 32) | __float_as_uint(tot * 1e-3f + 1.0f));
    }
    if (tid == 0) out[blockIdx.x] = acc;
}

template <int P, int G> static void run(int stages, int steps, int same_xcd) {
    uint64_t* mbox; int* ab; float* out;
    CK(hipMalloc(&mbox, (size_t)stages * P * G * 8)); CK(hipMemset(mbox, 0, (size_t)stages * P * G * 8));
    CK(hipMalloc(&ab, 4)); CK(hipMemset(ab, 0, 4)); CK(hipMalloc(&out, 4096));
    const int lds_bytes = 100 * 1024;
    CK(hipFuncSetAttribute((const void*)ring_kernel<P, G>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    const int grid = same_xcd ? 8 * same_xcd * P : stages * P;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((ring_kernel<P, G>), dim3(grid), dim3(256), lds_bytes, 0, mbox, stages, steps, ab, out, same_xcd);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int h; CK(hipMemcpy(&h, ab, 4, hipMemcpyDeviceToHost));
    printf("P=%d G=%3d (%4d B/producer) stages=%d %s: %8.2f us/step  %6.3f us/hop %s\n", P, G, G * 8, stages, same_xcd ? "stages-per-XCD>0" : "round-robin      ",
           ms * 1e3 / steps, ms * 1e3 / steps / stages, h ? "ABORTED (timeout)" : "");
    CK(hipFree(mbox)); CK(hipFree(ab)); CK(hipFree(out));
}
int main() {
    run<8, 128>(24, 2000, 3); run<8, 86>(24, 2000, 3); run<4, 128>(24, 2000, 3); run<4, 86>(24, 2000, 3); run<4, 86>(24, 2000, 6); run<4, 86>(24, 2000, 8);
    run16<8, 43>(24, 2000, 3); run16<4, 43>(24, 2000, 3); run16<4, 43>(24, 2000, 6); run16<4, 64>(24, 2000, 3); run16<8, 64>(24, 2000, 3); run16<2, 43>(24, 2000, 3); run16<1, 43>(24, 2000, 3);
    return 0;
}
