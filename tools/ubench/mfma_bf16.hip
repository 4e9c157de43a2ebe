// Micro-benchmark: rate of the bf16 matrix instructions against the fp32 ones (same 16x16 output tile),
// and whether vector ALU work of other waves on the SIMD overlaps with them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: fp32 16x16x4   1: bf16 16x16x32   2: bf16 in half the waves, FMAs in the others
__global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    const int rank = __builtin_amdgcn_readfirstlane(wave) >> 2;
    f32x4 a0 = {seed, 0, 0, 0}, a1 = {0, seed, 0, 0}, a2 = {0, 0, seed, 0}, a3 = {0, 0, 0, seed};
    float av = seed * threadIdx.x, bv = seed + threadIdx.x;
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(seed + i); bh[i] = (__bf16)(seed - i); }
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3;
    const bool valu = MODE == 2 && (rank & 1);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, av, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, bv, a3, 0, 0, 0);
            }
        } else if (!valu) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, ah, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, bh, a3, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                x0 = fmaf(x0, av, bv);
                x1 = fmaf(x1, av, bv);
                x2 = fmaf(x2, av, bv);
                x3 = fmaf(x3, av, bv);
            }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + x0 + x1 + x2 + x3;
}

template <int MODE>
float run(float* out, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256, 1024>>>(out, 10, 0.f);
    (void)hipEventRecord(e0);
    k<MODE><<<256, 1024>>>(out, iters, 0.f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 400;
    // per wave per iteration: 64 MFMAs.  fp32 16x16x4: 2048 flop each; bf16 16x16x32: 16384 flop each
    const double waves = 256.0 * 16, per_it = 64;
    float t0 = run<0>(out, iters), t1 = run<1>(out, iters), t2 = run<2>(out, iters);
    printf("fp32 16x16x4   : %.3f ms  %.1f TFLOP/s\n", t0, waves * per_it * iters * 2048 / t0 * 1e-9);
    printf("bf16 16x16x32  : %.3f ms  %.1f TFLOP/s\n", t1, waves * per_it * iters * 16384 / t1 * 1e-9);
    printf("bf16 in half the waves, 256 FMAs/iteration in the other half: %.3f ms (bf16 alone at half the waves would be %.3f)\n",
           t2, t1 / 2);
    return 0;
}
