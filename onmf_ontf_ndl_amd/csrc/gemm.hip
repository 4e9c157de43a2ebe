// fp32 MFMA matrix products for the ONMF step on gfx950.
//
// One kernel covers every contraction of the path:
//   Gm = W^T W, Cm = W^T X      (contraction index is the slow index of both operands)
//   H H^T, H X^T, X X^T         (contraction index is the fast index of both operands)
//   Qt = A^T W^T                (mixed)
// Operand tiles are staged in LDS in "k-major" form [BK][128]; an operand whose memory layout
// is [x][k] is transposed on the way into LDS (row pitch 129 keeps those scattered stores and
// the later row reads conflict-free).  Each of the 4 waves owns a 64x64 block of the 128x128
// output tile as 2x2 v_mfma_f32_32x32x2_f32 accumulators (exact fp32, k-ordered fma chain).
// Long contractions (K = batch) are split over blockIdx.z into slabs that a second kernel sums
// in a fixed order, so results are bitwise reproducible.
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 32;
constexpr int NTHREADS = 256;
constexpr int PITCH_KMAJ = 128;   // memory already k-major: float4 row copies
constexpr int PITCH_XMAJ = 129;   // transposing store: (k*129 + x) spreads over all 32 banks

struct GemmParams {
    const float* A;
    const float* B;
    const float* Cin;
    float* C;
    float* slab;
    int M, N, K;
    int lda, ldb, ldc;
    int k_per_split;
    int splits;
    int tiles_n;
    int vecA, vecB;
    float alpha, beta;
};

// Global -> registers for one BK x 128 operand tile (4 float4 per thread).
template <bool KMAJ>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int xdim, int x0,
                                          int k0, int kend, int vec_ok, float4 (&rg)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KMAJ) {
            const int k = k0 + (t >> 5) + 8 * p;
            const int x = x0 + (t & 31) * 4;
            if (k < kend) {
                const float* src = P + (size_t)k * ld + x;
                if (vec_ok && x + 3 < xdim) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (x + 0 < xdim) v.x = src[0];
                    if (x + 1 < xdim) v.y = src[1];
                    if (x + 2 < xdim) v.z = src[2];
                    if (x + 3 < xdim) v.w = src[3];
                }
            }
        } else {
            const int x = x0 + (t >> 3) + 32 * p;
            const int k = k0 + (t & 7) * 4;
            if (x < xdim) {
                const float* src = P + (size_t)x * ld + k;
                if (vec_ok && k + 3 < kend) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (k + 0 < kend) v.x = src[0];
                    if (k + 1 < kend) v.y = src[1];
                    if (k + 2 < kend) v.z = src[2];
                    if (k + 3 < kend) v.w = src[3];
                }
            }
        }
        rg[p] = v;
    }
}

// Registers -> LDS, always producing the k-major image S[k][x].
template <bool KMAJ>
__device__ __forceinline__ void store_tile(float* __restrict__ S, const float4 (&rg)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        if (KMAJ) {
            const int k = (t >> 5) + 8 * p;
            const int x = (t & 31) * 4;
            *reinterpret_cast<float4*>(S + k * PITCH_KMAJ + x) = rg[p];
        } else {
            const int x = (t >> 3) + 32 * p;
            const int k = (t & 7) * 4;
            S[(k + 0) * PITCH_XMAJ + x] = rg[p].x;
            S[(k + 1) * PITCH_XMAJ + x] = rg[p].y;
            S[(k + 2) * PITCH_XMAJ + x] = rg[p].z;
            S[(k + 3) * PITCH_XMAJ + x] = rg[p].w;
        }
    }
}

template <bool A_KM, bool B_KN>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PA = A_KM ? PITCH_KMAJ : PITCH_XMAJ;
    constexpr int PB = B_KN ? PITCH_KMAJ : PITCH_XMAJ;
    constexpr int A_ELEMS = BK * PA;
    constexpr int B_ELEMS = BK * PB;
    // layout: [A buf0][A buf1][B buf0][B buf1]; every part starts 16-byte aligned
    constexpr int A_STRIDE = (A_ELEMS + 3) & ~3;
    constexpr int B_STRIDE = (B_ELEMS + 3) & ~3;
    float* As = lds;
    float* Bs = lds + 2 * A_STRIDE;

    const int tile = blockIdx.x;
    const int m0 = (tile / p.tiles_n) * BM;
    const int n0 = (tile % p.tiles_n) * BN;
    const int z = blockIdx.z;
    const int kbeg = z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64;
    const int wn = (wave & 1) * 64;
    const int l31 = lane & 31;
    const int lhi = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[4], rb[4];
    const int ntiles = (kend - kbeg + BK - 1) / BK;
    if (ntiles > 0) {
        load_tile<A_KM>(p.A, p.lda, p.M, m0, kbeg, kend, p.vecA, ra);
        load_tile<B_KN>(p.B, p.ldb, p.N, n0, kbeg, kend, p.vecB, rb);
        store_tile<A_KM>(As, ra);
        store_tile<B_KN>(Bs, rb);
    }
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        const bool more = (t + 1 < ntiles);
        if (more) {
            load_tile<A_KM>(p.A, p.lda, p.M, m0, kbeg + (t + 1) * BK, kend, p.vecA, ra);
            load_tile<B_KN>(p.B, p.ldb, p.N, n0, kbeg + (t + 1) * BK, kend, p.vecB, rb);
        }
        const float* Ac = As + cur * A_STRIDE;
        const float* Bc = Bs + cur * B_STRIDE;
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const int k = 2 * s + lhi;
            const float a0 = Ac[k * PA + wm + l31];
            const float a1 = Ac[k * PA + wm + 32 + l31];
            const float b0 = Bc[k * PB + wn + l31];
            const float b1 = Bc[k * PB + wn + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            store_tile<A_KM>(As + (cur ^ 1) * A_STRIDE, ra);
            store_tile<B_KN>(Bs + (cur ^ 1) * B_STRIDE, rb);
        }
        __syncthreads();
    }

    // accumulator element e of lane l: row (e&3) + 8*(e>>2) + 4*(l>>5), column l&31
    const bool direct = (p.splits == 1);
    float* out = direct ? p.C : p.slab + (size_t)z * p.M * p.N;
    const int ldo = direct ? p.ldc : p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + 32 * j + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (row < p.M && col < p.N) {
                    float v = acc[i][j][e];
                    if (direct) {
                        v *= p.alpha;
                        if (p.beta != 0.f) v += p.beta * p.Cin[(size_t)row * p.ldc + col];
                    }
                    out[(size_t)row * ldo + col] = v;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab,
                                                             int splits, int M, int N, float alpha,
                                                             float beta, const float* Cin,
                                                             float* C, int ldc) {
    const size_t mn = (size_t)M * N;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < mn;
         i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += slab[(size_t)z * mn + i];
        const int row = (int)(i / N), col = (int)(i % N);
        float v = alpha * s;
        if (beta != 0.f) v += beta * Cin[(size_t)row * ldc + col];
        C[(size_t)row * ldc + col] = v;
    }
}

template <bool A_KM, bool B_KN>
int run(onmf_ctx* c, const GemmParams& p, dim3 grid, hipStream_t s) {
    constexpr int PA = A_KM ? PITCH_KMAJ : PITCH_XMAJ;
    constexpr int PB = B_KN ? PITCH_KMAJ : PITCH_XMAJ;
    constexpr size_t bytes =
        sizeof(float) * 2 * ((((size_t)BK * PA + 3) & ~(size_t)3) + (((size_t)BK * PB + 3) & ~(size_t)3));
    static bool attr_done = false;
    if (!attr_done) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)gemm_f32_kernel<A_KM, B_KN>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<A_KM, B_KN>), grid, dim3(NTHREADS), bytes, s, p);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

}  // namespace

int launch_gemm(onmf_ctx* c, bool a_km, bool b_kn, int M, int N, int K, const float* A, int lda,
                const float* B, int ldb, float alpha, float beta, const float* Cin, float* C,
                int ldc, hipStream_t s, DevBuf* slab) {
    if (!slab) slab = &c->slab;
    if (M <= 0 || N <= 0 || K < 0) return onmf_fail(c, ONMF_ERR_ARG, "gemm: bad shape %dx%dx%d", M, N, K);
    if (beta != 0.f && Cin == nullptr) return onmf_fail(c, ONMF_ERR_ARG, "gemm: beta without Cin");
    GemmParams p;
    p.A = A; p.B = B; p.Cin = Cin; p.C = C; p.slab = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.beta = beta;
    p.vecA = (((uintptr_t)A & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
    p.vecB = (((uintptr_t)B & 15) == 0 && (ldb & 3) == 0) ? 1 : 0;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    p.tiles_n = tiles_n;
    const int tiles = tiles_m * tiles_n;
    const int ktiles = (K + BK - 1) / BK;
    int splits = 1;
    if (tiles < 256 && ktiles >= 16) {
        splits = (512 + tiles - 1) / tiles;
        const int max_by_k = ktiles / 8;          // at least 8 k-tiles of work per slab
        if (splits > max_by_k) splits = max_by_k;
        if (splits < 1) splits = 1;
    }
    int kt_per = (ktiles + splits - 1) / splits;
    if (kt_per < 1) kt_per = 1;
    splits = (ktiles + kt_per - 1) / kt_per;
    if (splits < 1) splits = 1;
    p.k_per_split = kt_per * BK;
    p.splits = splits;
    if (splits > 1) {
        ONMF_TRY(onmf_reserve(c, *slab, (size_t)splits * M * N * sizeof(float), "split-K slabs"));
        p.slab = (float*)slab->p;
    }
    dim3 grid(tiles, 1, splits);
    int rc;
    if (a_km && b_kn) rc = run<true, true>(c, p, grid, s);
    else if (a_km && !b_kn) rc = run<true, false>(c, p, grid, s);
    else if (!a_km && b_kn) rc = run<false, true>(c, p, grid, s);
    else rc = run<false, false>(c, p, grid, s);
    if (rc != ONMF_OK) return rc;
    if (splits > 1) {
        const size_t mn = (size_t)M * N;
        int blocks = (int)((mn + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.slab, splits, M, N,
                           alpha, beta, Cin, C, ldc);
        HIP_TRY(c, hipGetLastError());
    }
    return ONMF_OK;
}
