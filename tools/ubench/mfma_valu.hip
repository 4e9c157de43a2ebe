// Micro-benchmark: do fp32 MFMAs and fp32 VALU FMAs issued by different waves of one SIMD overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
    const int wave = threadIdx.x >> 6;
    const int rank = __builtin_amdgcn_readfirstlane(wave) >> 2;   // 0..3 = position among the 4 waves of a SIMD
    bool do_mfma, do_valu;
    if (MODE == 0) { do_mfma = true; do_valu = false; }
    else if (MODE == 1) { do_mfma = false; do_valu = true; }
    else if (MODE == 2) { do_mfma = (rank & 1) == 0; do_valu = !do_mfma; }
    else { do_mfma = true; do_valu = true; }
    f32x4 a0 = {seed, 0, 0, 0}, a1 = {0, seed, 0, 0};
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3;
    float av = seed * threadIdx.x, bv = seed + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av, a1, 0, 0, 0);
            }
        }
        if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
        if (do_valu) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                x0 = fmaf(x0, av, bv);
                x1 = fmaf(x1, av, bv);
                x2 = fmaf(x2, av, bv);
                x3 = fmaf(x3, av, bv);
            }
        }
        if (MODE == 3) __builtin_amdgcn_sched_barrier(0);
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0[0] + a1[1] + x0 + x1 + x2 + x3;
}

template <int MODE>
float run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 1024>>>(out, 10, 0.f);
    hipEventRecord(e0);
    k<MODE><<<256, 1024>>>(out, iters, 0.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 400;
    // per wave per iteration: 64 MFMA (64*32 = 2048 pipe cycles), 256 FMA (256*4 = 1024 issue cycles)
    printf("mode0 all waves MFMA          %.3f ms\n", run<0>(out, iters));
    printf("mode1 all waves VALU          %.3f ms\n", run<1>(out, iters));
    printf("mode2 half MFMA / half VALU   %.3f ms\n", run<2>(out, iters));
    printf("mode3 every wave MFMA then VALU %.3f ms\n", run<3>(out, iters));
    printf("ideal cycles/iter/SIMD: mode0 8192, mode1 4096, mode2 overlap 4096 / additive 6144, mode3 overlap 8192 / additive 12288\n");
    return 0;
}
