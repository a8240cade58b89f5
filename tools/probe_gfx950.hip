// Hardware probe (run once on the GPU box): prints the lane/element mapping of ds_read_b64_tr_b16 and
// checks the A/B/C fragment layout of v_mfma_f32_32x32x16_bf16 that csrc/wn_tile.h assumes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void tr_probe(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    short4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}
static __host__ __device__ unsigned short f2bf(float f) { union { float f; uint32_t u; } v; v.f = f; return (unsigned short)((v.u + 0x7fff + ((v.u >> 16) & 1)) >> 16); }
__global__ void mfma_probe(const unsigned short* A /*[32][16]*/, const unsigned short* B /*[16][32]*/, float* C /*[32][32]*/) {
    const int lane = threadIdx.x;
    unsigned short a[8], b[8];
    for (int j = 0; j < 8; ++j) { int k = 8 * (lane >> 5) + j; a[j] = A[(lane & 31) * 16 + k]; b[j] = B[k * 32 + (lane & 31)]; }
    bf16x8 av, bv;
    __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) { int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), n = lane & 31; C[m * 32 + n] = acc[r]; }
}
int main() {
    short* d; hipMalloc(&d, 512); tr_probe<<<1, 64>>>(d); short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("ds_read_tr16_b64: lane -> 4 source element indices (lds[i]=i, lane l address = 4*l)\n");
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    unsigned short hA[512], hB[512]; float ref[1024] = {0}, hC[1024];
    for (int m = 0; m < 32; ++m) for (int k = 0; k < 16; ++k) hA[m * 16 + k] = f2bf((float)((m * 7 + k * 3) % 11 - 5));
    for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) hB[k * 32 + n] = f2bf((float)((k * 5 + n * 2) % 13 - 6));
    auto bf = [](unsigned short x) { union { float f; uint32_t u; } v; v.u = (uint32_t)x << 16; return v.f; };
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 16; ++k) s += bf(hA[m * 16 + k]) * bf(hB[k * 32 + n]); ref[m * 32 + n] = s; }
    unsigned short *dA, *dB; float* dC; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    mfma_probe<<<1, 64>>>(dA, dB, dC); hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) if (hC[i] != ref[i]) ++bad;
    printf("mfma_f32_32x32x16_bf16 fragment layout check: %d mismatches of 1024 (asymmetric A,B)\n", bad);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device: %s CUs=%d clock=%d MHz mem=%.1f GB L2=%d KB\n", p.name, p.multiProcessorCount, p.clockRate / 1000, p.totalGlobalMem / 1e9, p.l2CacheSize / 1024);
    return bad != 0;
}
