// Micro-benchmark: v_mfma_f32_32x32x2_f32 issue rate with 1, 2, 4 waves per SIMD, with and without LDS operand reads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDSREAD>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = seed + i;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int lane = threadIdx.x & 63;
    float a0v = seed * lane, a1v = seed + lane, b0v = seed - lane, b1v = seed * 2 + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float av0[4], av1[4], bv0[4], bv1[4];
            if (LDSREAD) {
                const float4 a0 = *reinterpret_cast<const float4*>(lds + ((lane * 4 + q * 256 + it * 16) & 8188));
                const float4 a1 = *reinterpret_cast<const float4*>(lds + ((lane * 4 + q * 256 + 1024 + it * 16) & 8188));
                const float4 b0 = *reinterpret_cast<const float4*>(lds + ((lane * 4 + q * 256 + 2048 + it * 16) & 8188));
                const float4 b1 = *reinterpret_cast<const float4*>(lds + ((lane * 4 + q * 256 + 3072 + it * 16) & 8188));
                av0[0] = a0.x; av0[1] = a0.y; av0[2] = a0.z; av0[3] = a0.w;
                av1[0] = a1.x; av1[1] = a1.y; av1[2] = a1.z; av1[3] = a1.w;
                bv0[0] = b0.x; bv0[1] = b0.y; bv0[2] = b0.z; bv0[3] = b0.w;
                bv1[0] = b1.x; bv1[1] = b1.y; bv1[2] = b1.z; bv1[3] = b1.w;
            } else {
                for (int j = 0; j < 4; ++j) { av0[j] = a0v; av1[j] = a1v; bv0[j] = b0v; bv1[j] = b1v; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv1[j], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv1[j], acc[1][1], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int LDSREAD>
void run(float* out, int wgs_per_cu, const char* name) {
    const int iters = 200;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<LDSREAD><<<256 * wgs_per_cu, 256>>>(out, 10, 1.f);
    hipEventRecord(e0);
    k<LDSREAD><<<256 * wgs_per_cu, 256>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * wgs_per_cu * 4 * iters * 64 * 4096.0;
    printf("%-44s %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s\n", name, wgs_per_cu, ms, flops / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
    for (int w = 1; w <= 4; w *= 2) run<0>(out, w, "MFMA 32x32x2 f32, register operands");
    for (int w = 1; w <= 4; w *= 2) run<1>(out, w, "MFMA 32x32x2 f32, operands from LDS (b128)");
    return 0;
}
