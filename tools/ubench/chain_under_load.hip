// Micro-benchmark: how much slower does a one-wave dependent ALU/LDS chain run while the rest of the
// chip is busy with fp32 MFMAs - (a) alone on its CU (the load leaves one CU free), (b) sharing its CU
// with a load workgroup?  Times come from the constant 100 MHz counter read inside the kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void load_kernel(float* out, int iters, float seed) {
    extern __shared__ float pad[];                    // only reserves LDS
    f32x4 a0 = {seed, 0, 0, 0}, a1 = {0, seed, 0, 0};
    float av = seed * threadIdx.x, bv = seed + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, a1, 0, 0, 0);
        }
    }
    if (seed == 12345.f) pad[threadIdx.x] = a0[0];
    out[blockIdx.x * 1024 + threadIdx.x] = a0[0] + a1[1];
}

// kind 0: dependent FMAs; kind 1: FMA + LDS write/read + barrier per step (256 threads)
__global__ __launch_bounds__(256) void chain_kernel(float* out, long long* ticks, int steps, int kind, float seed) {
    extern __shared__ float lds[];
    __builtin_amdgcn_s_setprio(3);
    float x = seed + threadIdx.x;
    const long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = fmaf(x, 1.0001f, seed);
        if (kind == 1) {
            lds[threadIdx.x] = x;
            __syncthreads();
            x += lds[(threadIdx.x + 64) & 255];
            __syncthreads();
        }
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

int main() {
    float* out; long long* ticks; long long h = 0;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&ticks, 8);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipFuncSetAttribute((const void*)load_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    const int steps = 20000;
    for (int kind = 0; kind < 2; ++kind) {
        // chain's LDS: 100 KB -> cannot share a CU with a load workgroup that also holds 100 KB
        for (int mode = 0; mode < 3; ++mode) {        // 0 alone, 1 load on 255 CUs (chain has its own), 2 load everywhere, CU shared
            const int load_wgs = mode == 1 ? 255 : 256;
            const size_t load_lds = mode == 1 ? 100 * 1024 : 1024;
            const size_t chain_lds = mode == 2 ? 4096 : 100 * 1024;
            if (mode > 0) load_kernel<<<load_wgs, 1024, load_lds, sa>>>(out, 6000, 0.f);   // ~ 20 ms of MFMAs
            chain_kernel<<<1, 256, chain_lds, sb>>>(out, ticks, steps, kind, 0.f);
            hipDeviceSynchronize();
            hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost);
            printf("kind %d mode %d: %.1f us for %d steps (%.1f ns/step)\n", kind, mode, h / 100.0, steps, h * 10.0 / steps);
            fflush(stdout);
        }
    }
    return 0;
}
