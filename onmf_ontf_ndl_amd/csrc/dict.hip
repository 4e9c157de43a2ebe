// Dictionary block-coordinate pass of the ONMF step on gfx950 (src/onmf.py:92-116 of the
// reference).  For j = 0..r-1 in order:
//     u = max(0, W[:,j] - (W A[:,j] - B[j,:]^T) / (A[j,j] + 1));   W[:,j] = u / max(1, ||u||_2)
// with later atoms seeing earlier updates.
//
// Formulation: keep Qt[j][i] = sum_k W[i][k] A[k][j] (r x d) current instead of re-forming
// W A[:,j] per atom.  Qt starts as one MFMA product; when atom j moves by delta (a d-vector), row
// jj > j of Qt moves by A[j][jj] * delta.  A workgroup is one wave owning 64 rows of W; its slice
// of Qt lives in LDS as [r][64].  The only coupling between workgroups is the atom norm: r
// sequential d-wide reductions.  Each workgroup publishes its partial sum of squares as one
// 8-byte {tag, value} word with an agent-scope store, and every workgroup sums all words in the
// same fixed order, so all of them scale by the bitwise-identical norm.
#include "common.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(64) void dict_pass_kernel(int d, int r, int nwg,
                                                       const float* __restrict__ Qt0,
                                                       const float* __restrict__ A,
                                                       const float* __restrict__ B,
                                                       const float* Win, float* Wout,
                                                       u64* granules,
                                                       int* timeout_flag, long long spin_limit) {
    extern __shared__ __attribute__((aligned(16))) float qs[];
    const int lane = threadIdx.x;
    const int wg = blockIdx.x;
    const int i = wg * 64 + lane;
    const bool valid = i < d;
    for (int j = 0; j < r; ++j) qs[j * 64 + lane] = valid ? Qt0[(size_t)j * d + i] : 0.f;
    bool dead = false;
    for (int j = 0; j < r; ++j) {
        const float* __restrict__ arow = A + (size_t)j * r;
        const float inv = 1.0f / (arow[j] + 1.0f);
        const float w = valid ? Win[(size_t)i * r + j] : 0.f;
        const float bj = valid ? B[(size_t)j * d + i] : 0.f;
        const float u = valid ? fmaxf(w - (qs[j * 64 + lane] - bj) * inv, 0.f) : 0.f;
        float total = wave_sum(u * u);
        if (nwg > 1) {
            // publish this workgroup's partial: one aligned 8-byte word, tag = j + 1 (never 0)
            if (lane == 0) {
                const u64 word = ((u64)(unsigned)(j + 1) << 32) | (u64)__float_as_uint(total);
                __hip_atomic_store(&granules[(size_t)j * nwg + wg], word, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            float part = 0.f;
            for (int g0 = 0; g0 < nwg; g0 += 64) {
                const int g = g0 + lane;
                const long long t0 = wall_clock64();
                float val = 0.f;
                for (;;) {
                    bool ok = true;
                    if (g < nwg) {
                        const u64 x = __hip_atomic_load(&granules[(size_t)j * nwg + g],
                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = (unsigned)(x >> 32) == (unsigned)(j + 1);
                        val = __uint_as_float((unsigned)x);
                    }
                    if (__all(ok) || dead) break;
                    if (wall_clock64() - t0 > spin_limit) {
                        dead = true;      // wave-uniform: every lane reads the same clock value
                        if (lane == 0) atomicExch(timeout_flag, 1);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                part += val;
            }
            total = wave_sum(part);
        }
        const float scale = 1.0f / fmaxf(1.0f, sqrtf(total));
        const float wn = u * scale;
        const float delta = wn - w;
        if (valid) Wout[(size_t)i * r + j] = wn;
        for (int jj = j + 1; jj < r; ++jj) qs[jj * 64 + lane] = fmaf(arow[jj], delta, qs[jj * 64 + lane]);
    }
}

__global__ void zero_u64_kernel(u64* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        p[i] = 0ull;
}

}  // namespace

int launch_dict_pass(onmf_ctx* c, int d, int r, const float* Win, const float* A, const float* B,
                     float* Wout, hipStream_t s) {
    const int nwg = (d + 63) / 64;
    if (nwg > 256)
        return onmf_fail(c, ONMF_ERR_ARG, "update_dict: d=%d needs %d co-resident workgroups (max 256)", d, nwg);
    size_t lds = (size_t)r * 64 * sizeof(float);
    if (lds > 160 * 1024)
        return onmf_fail(c, ONMF_ERR_ARG, "update_dict: r=%d needs %zu B of LDS", r, lds);
    // more than half of a CU's LDS: at most one workgroup per CU, so all nwg <= 256 are resident
    if (nwg > 1 && lds < 81 * 1024) lds = 81 * 1024;
    // Qt0[j][i] = sum_k A[k][j] * W[i][k]
    ONMF_TRY(onmf_reserve(c, c->qt, (size_t)r * d * sizeof(float), "dictionary Qt"));
    float* Qt = (float*)c->qt.p;
    ONMF_TRY(launch_gemm(c, /*a_km=*/true, /*b_kn=*/false, r, d, r, A, r, Win, r, 1.f, 0.f, nullptr,
                         Qt, d, s, &c->slab_side));
    const size_t ngran = (size_t)r * nwg;
    ONMF_TRY(onmf_reserve(c, c->sync, ngran * sizeof(u64) + 64, "dictionary sync words"));
    u64* gran = (u64*)c->sync.p;
    int* flag = (int*)((char*)c->sync.p + ngran * sizeof(u64));
    hipLaunchKernelGGL(zero_u64_kernel, dim3((unsigned)((ngran + 8 + 255) / 256)), dim3(256), 0, s, gran,
                       ngran + 8);
    HIP_TRY(c, hipGetLastError());
    static size_t attr_bytes = 0;
    if (lds > attr_bytes) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)dict_pass_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_bytes = lds;
    }
    const long long spin_limit = 200000000LL;   // 2 s of the 100 MHz wall clock
    hipLaunchKernelGGL(dict_pass_kernel, dim3(nwg), dim3(64), lds, s, d, r, nwg, Qt, A, B, Win, Wout,
                       gran, flag, spin_limit);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}
