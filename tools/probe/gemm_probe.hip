// Where do the idle matrix-pipe cycles of the 128x128x32 fp32 product kernel go?  The same loop as
// csrc/gemm.hip (both operands [x][k], interior tiles), with parts switched off one at a time.
// Build: hipcc -O3 --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops gemm_probe.hip -o gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int BK = 32, TILE = 128 * BK;

__device__ __forceinline__ int lds_off(int x, int k) {
    const int sw = (x & 7) ^ (2 * ((x >> 3) & 1));
    return x * BK + 4 * ((k >> 2) ^ sw) + (k & 3);
}

// MODE bits: 1 = global loads + LDS stores, 2 = barrier, 4 = LDS operand reads
template <int MODE, int WAVES_N>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                             int lda, int ldb, int ntiles, int tiles_n, int kper) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;
    float* Bs = lds + 2 * TILE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64, l31 = lane & 31, lhi = lane >> 5;
    const int m0 = (blockIdx.x / tiles_n) * 128, n0 = (blockIdx.x % tiles_n) * 128;
    const int kbeg = blockIdx.z * kper;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
    const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0x7fffffff, 0x00020000);
    int voffA[4], voffB[4];
    for (int p = 0; p < 4; ++p) {
        voffA[p] = (((t >> 3) + 32 * p) * lda + (t & 7) * 4) * 4;
        voffB[p] = (((t >> 3) + 32 * p) * ldb + (t & 7) * 4) * 4;
    }
    const unsigned sA = (unsigned)(((size_t)m0 * lda + kbeg) * 4), sB = (unsigned)(((size_t)n0 * ldb + kbeg) * 4);
    u32x4 ra[4], rb[4], ra2[4], rb2[4];
    auto gload = [&](int tile, u32x4 (&xa)[4], u32x4 (&xb)[4]) {
        for (int p = 0; p < 4; ++p) {
            xa[p] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA[p], sA + tile * BK * 4, 0);
            xb[p] = __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB[p], sB + tile * BK * 4, 2);
        }
    };
    auto lstore = [&](float* Sa, float* Sb, const u32x4 (&xa)[4], const u32x4 (&xb)[4]) {
        for (int p = 0; p < 4; ++p) {
            const int x = (t >> 3) + 32 * p, k = (t & 7) * 4;
            *reinterpret_cast<u32x4*>(Sa + lds_off(x, k)) = xa[p];
            *reinterpret_cast<u32x4*>(Sb + lds_off(x, k)) = xb[p];
        }
    };
    gload(0, ra, rb);
    lstore(As, Bs, ra, rb);
    gload(1, ra, rb);
    __syncthreads();
    auto round = [&](int tt, u32x4 (&xa)[4], u32x4 (&xb)[4], u32x4 (&ya)[4], u32x4 (&yb)[4]) {
        const int cur = tt & 1;
        if ((MODE & 1) && !(MODE & 8)) gload(min(tt + 2, ntiles - 1), ya, yb);
        const float* Ac = As + cur * TILE;
        const float* Bc = Bs + cur * TILE;
        auto fetch = [&](int q, float4& a0, float4& a1, float4& b0, float4& b1) {
            const int k = 8 * q + 4 * lhi;
            a0 = *reinterpret_cast<const float4*>(Ac + lds_off(wm + l31, k));
            a1 = *reinterpret_cast<const float4*>(Ac + lds_off(wm + 32 + l31, k));
            b0 = *reinterpret_cast<const float4*>(Bc + lds_off(wn + l31, k));
            b1 = *reinterpret_cast<const float4*>(Bc + lds_off(wn + 32 + l31, k));
        };
        float4 a0, a1, b0, b1, na0, na1, nb0, nb1;
        fetch(0, a0, a1, b0, b1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if ((MODE & 4) && q + 1 < 4) fetch(q + 1, na0, na1, nb0, nb1);
            if (!(MODE & 8)) {
                if ((MODE & 1) && q == 2) lstore(As + (cur ^ 1) * TILE, Bs + (cur ^ 1) * TILE, xa, xb);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
            const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv0[j], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv1[j], acc[0][1], 0, 0, 0);
                if (MODE & 8) {
                    // one memory instruction behind every second MFMA: it issues while the MFMA runs
                    const int p = j;
                    if ((MODE & 1) && q == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        ya[p] = __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA[p], sA + min(tt + 2, ntiles - 1) * BK * 4, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if ((MODE & 1) && q == 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        *reinterpret_cast<u32x4*>(As + (cur ^ 1) * TILE + lds_off((t >> 3) + 32 * p, (t & 7) * 4)) = xa[p];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv0[j], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv1[j], acc[1][1], 0, 0, 0);
                if (MODE & 8) {
                    const int p = j;
                    if ((MODE & 1) && q == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        yb[p] = __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB[p], sB + min(tt + 2, ntiles - 1) * BK * 4, 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if ((MODE & 1) && q == 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        *reinterpret_cast<u32x4*>(Bs + (cur ^ 1) * TILE + lds_off((t >> 3) + 32 * p, (t & 7) * 4)) = xb[p];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE & 4) { a0 = na0; a1 = na1; b0 = nb0; b1 = nb1; }
        }
        if (MODE & 2) __syncthreads();
    };
    for (int tt = 0; tt < ntiles; tt += 2) {
        round(tt, ra, rb, ra2, rb2);
        round(tt + 1, ra2, rb2, ra, rb);
        asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[1][0]), "+a"(acc[1][1]));
    }
    float* out = C + (size_t)blockIdx.z * gridDim.x * 128 * 128 + (size_t)blockIdx.x * 128 * 128;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e)
                out[(wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lhi) * 128 + wn + 32 * j + l31] = acc[i][j][e];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
int run(const char* name, const float* A, const float* B, float* C, int lda, int ldb, int tiles, int tiles_n, int splits, int ntiles) {
    const size_t lds = 4 * TILE * sizeof(float);
    CK(hipFuncSetAttribute((const void*)probe<MODE, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int kper = ntiles * BK;
    for (int w = 0; w < 400; ++w) hipLaunchKernelGGL((probe<MODE, 2>), dim3(tiles, 1, splits), dim3(256), lds, 0, A, B, C, lda, ldb, ntiles, tiles_n, kper);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 50;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<MODE, 2>), dim3(tiles, 1, splits), dim3(256), lds, 0, A, B, C, lda, ldb, ntiles, tiles_n, kper);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    const double ideal = (double)ntiles * 64 * 64 * 2 / 2.4e3;       // two waves per SIMD, 64 MFMAs of 64 cycles per round
    printf("%-44s %8.1f us   (in-loop ideal at 2.4 GHz, two workgroups per CU: %.1f us)\n", name, us, ideal);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    // the aggregate pair's shape: 20 tiles x 24 slabs, 86 k-tiles each
    const int tiles = 20, tiles_n = 10, splits = argc > 1 ? atoi(argv[1]) : 24, ntiles = 86;
    const int K = splits * ntiles * BK;
    float *A, *B, *C;
    CK(hipMalloc(&A, (size_t)256 * K * 4));
    CK(hipMalloc(&B, (size_t)1280 * K * 4));
    CK(hipMalloc(&C, (size_t)tiles * splits * 128 * 128 * 4));
    std::vector<float> h((size_t)1280 * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(rand() & 1023) / 1024.f;
    CK(hipMemcpy(A, h.data(), (size_t)256 * K * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), (size_t)1280 * K * 4, hipMemcpyHostToDevice));
    if (run<7>("all (loads, barrier, LDS reads)", A, B, C, K, K, tiles, tiles_n, splits, ntiles)) return 1;
    if (run<15>("all, memory instructions between the MFMAs", A, B, C, K, K, tiles, tiles_n, splits, ntiles)) return 1;
    if (run<6>("no global loads / LDS stores", A, B, C, K, K, tiles, tiles_n, splits, ntiles)) return 1;
    if (run<4>("no loads, no barrier", A, B, C, K, K, tiles, tiles_n, splits, ntiles)) return 1;
    if (run<0>("MFMAs only", A, B, C, K, K, tiles, tiles_n, splits, ntiles)) return 1;
    if (run<2>("MFMAs + barrier", A, B, C, K, K, tiles, tiles_n, splits, ntiles)) return 1;
    if (run<5>("loads + LDS reads, no barrier (racy: timing only)", A, B, C, K, K, tiles, tiles_n, splits, ntiles)) return 1;
    return 0;
}
