// Gauss-Seidel code sweeps of the ONMF sparse-code solve on gfx950 (src/onmf.py:252-263 of the
// reference):  for k in order:  g = Gm[k,:] . h - Cm[k] + alpha ;  h[k] = max(0, h[k] - g * step_k)
// with step_k = 1 / (sqrt(i+10) * (Gm[k,k] + 1)).  Columns of H are independent.
#include "common.h"

#include <cmath>

namespace {

// ---------------------------------------------------------------------------------------------
// Generic kernel: one column per lane, the column's h in LDS ([r][64], conflict-free), rows of
// Gm read through wave-uniform (scalar) loads.  Any r with r*256 B <= 160 KB.  It is the
// fallback for shapes the tiled kernel does not cover and the cross-check for it.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sweep_column_kernel(int r, int b, const float* __restrict__ Gm,
                                                          const float* __restrict__ Cm,
                                                          const float* __restrict__ Hin,
                                                          float* __restrict__ Hout, float alpha,
                                                          float damp, int exact_cd,
                                                          double* __restrict__ stats,
                                                          const float* __restrict__ row0_scale) {
    extern __shared__ __attribute__((aligned(16))) float hs[];
    const int lane = threadIdx.x;
    const int c = blockIdx.x * 64 + lane;
    const bool valid = c < b;
    for (int k = 0; k < r; ++k) hs[k * 64 + lane] = valid ? Hin[(size_t)k * b + c] : 0.f;
    float dsum = 0.f, hsum = 0.f;
    // the radius form: the first row's move is shortened by this factor (see radius_row0_kernel)
    const float scale0 = row0_scale ? *row0_scale : 1.0f;
    for (int k = 0; k < r; ++k) {
        const float* __restrict__ gk = Gm + (size_t)k * r;
        float dot = 0.f;
        for (int j = 0; j < r; ++j) dot = fmaf(gk[j], hs[j * 64 + lane], dot);
        const float gkk = gk[k];
        const float step = exact_cd ? (gkk > 0.f ? 1.0f / gkk : 0.f) : 1.0f / (damp * (gkk + 1.0f));
        const float hold = hs[k * 64 + lane];
        const float cm = valid ? Cm[(size_t)k * b + c] : 0.f;
        const float grad = dot - cm + alpha;
        float hnew = valid ? fmaxf(hold - grad * step, 0.f) : 0.f;
        if (row0_scale && k == 0) hnew = hold + scale0 * (hnew - hold);
        hs[k * 64 + lane] = hnew;
        const float dl = hnew - hold;
        dsum = fmaf(dl, dl, dsum);
        hsum = fmaf(hold, hold, hsum);
        if (valid) Hout[(size_t)k * b + c] = hnew;
    }
    // wave reduction (fixed tree), one pair of doubles per workgroup
    double ds = dsum, hh = hsum;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ds += __shfl_xor(ds, off);
        hh += __shfl_xor(hh, off);
    }
    if (lane == 0) {
        stats[2 * blockIdx.x + 0] = ds;
        stats[2 * blockIdx.x + 1] = hh;
    }
}

// The radius argument of the reference's solver (src/onmf.py:260-263): after every row's update the
// iterate is pulled back to within spectral distance `radius` of H0 - and H0 is then rebound to the
// iterate itself (line 263), so from the second row on the distance is zero and nothing is pulled.  What
// is left is the first row of the first sweep: its move delta (the only non-zero row of H1 - H0, whose
// spectral norm is therefore the row's Euclidean norm) is scaled by radius / max(radius, ||delta||_2).
// This kernel forms the squared norm of that move, one partial per workgroup in a fixed order ...
__global__ __launch_bounds__(64) void radius_row0_kernel(int r, int b, const float* __restrict__ Gm,
                                                         const float* __restrict__ Cm,
                                                         const float* __restrict__ Hin, float alpha, float damp,
                                                         double* __restrict__ partial) {
    const int lane = threadIdx.x;
    const int c = blockIdx.x * 64 + lane;
    const bool valid = c < b;
    float dot = 0.f;
    for (int j = 0; j < r; ++j) dot = fmaf(Gm[j], valid ? Hin[(size_t)j * b + c] : 0.f, dot);
    const float step = 1.0f / (damp * (Gm[0] + 1.0f));
    const float hold = valid ? Hin[c] : 0.f;
    const float grad = dot - (valid ? Cm[c] : 0.f) + alpha;
    const float dl = valid ? fmaxf(hold - grad * step, 0.f) - hold : 0.f;
    double ds = (double)dl * (double)dl;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ds += __shfl_xor(ds, off);
    if (lane == 0) partial[blockIdx.x] = ds;
}

// ... and this one the factor itself.
__global__ __launch_bounds__(64) void radius_scale_kernel(int groups, const double* __restrict__ partial, float radius,
                                                          float* __restrict__ scale) {
    double acc = 0.0;
    for (int g = threadIdx.x; g < groups; g += 64) acc += partial[g];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (threadIdx.x == 0) {
        const double dn = sqrt(acc);
        *scale = (float)((double)radius / fmax((double)radius, dn));
    }
}

// The same with every sweep of the solve in one launch and a stopping test per column (see the tile
// kernel's per-column mode): a lane stops updating once its own column has converged.
__global__ __launch_bounds__(64) void sweep_column_each_kernel(int r, int b, const float* __restrict__ Gm,
                                                               const float* __restrict__ Cm, float* H,
                                                               float alpha, int exact_cd, int sub_iter,
                                                               EachStop es) {
    extern __shared__ __attribute__((aligned(16))) float hs[];
    const int lane = threadIdx.x;
    const int c = blockIdx.x * 64 + lane;
    const bool valid = c < b;
    for (int k = 0; k < r; ++k) hs[k * 64 + lane] = valid ? H[(size_t)k * b + c] : 0.f;
    bool active = valid;
    int n = 0;
    for (int i = 0; i < sub_iter; ++i) {
        if (__ballot(active) == 0) break;
        const float rdamp = 1.0f / (float)sqrt((double)i + 10.0);
        float dsum = 0.f, hsum = 0.f;
        for (int k = 0; k < r; ++k) {
            const float* __restrict__ gk = Gm + (size_t)k * r;
            float dot = 0.f;
            for (int j = 0; j < r; ++j) dot = fmaf(gk[j], hs[j * 64 + lane], dot);
            const float gkk = gk[k];
            const float step = exact_cd ? (gkk > 0.f ? 1.0f / gkk : 0.f) : rdamp / (gkk + 1.0f);
            const float hold = hs[k * 64 + lane];
            const float cm = valid ? Cm[(size_t)k * b + c] : 0.f;
            const float grad = dot - cm + alpha;
            const float hnew = active ? fmaxf(fmaf(-grad, step, hold), 0.f) : hold;
            hs[k * 64 + lane] = hnew;
            const float dl = hnew - hold;
            dsum = fmaf(dl, dl, dsum);
            hsum = fmaf(hold, hold, hsum);
        }
        if (active) {
            ++n;
            active = sqrt((double)dsum) / sqrt((double)hsum) > (double)es.tol;
        }
    }
    if (valid) {
        for (int k = 0; k < r; ++k) H[(size_t)k * b + c] = hs[k * 64 + lane];
        if (es.sweeps != nullptr) es.sweeps[c] = n;
    }
}

}  // namespace

int launch_sweep_each(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* H, float alpha,
                      int exact_cd, int sub_iter, EachStop es, hipStream_t s) {
    if (sub_iter < 1) return ONMF_OK;
    if (sweep_tile_covers(r, b) && c->sweep_kernel != 1)
        return launch_sweep_tile_each(c, r, b, Gm, Cm, H, alpha, exact_cd, sub_iter, es, s);
    if (c->sweep_kernel >= 2)
        return onmf_fail(c, ONMF_ERR_ARG, "sweep: the tile kernel covers r <= 256 (r=%d)", r);
    const size_t lds = (size_t)r * 64 * sizeof(float);
    if (lds > 160 * 1024) return onmf_fail(c, ONMF_ERR_ARG, "sweep: r=%d needs %zu B of LDS", r, lds);
    static LdsLimit lim;
    ONMF_TRY(onmf_raise_lds_limit(c, lim, (const void*)sweep_column_each_kernel, lds));
    hipLaunchKernelGGL(sweep_column_each_kernel, dim3((b + 63) / 64), dim3(64), lds, s, r, b, Gm, Cm, H, alpha,
                       exact_cd, sub_iter, es);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

static int launch_sweep_column(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm,
                               const float* Hin, float* Hout, float alpha, float damp, int exact_cd,
                               double* stats, int* n_groups, hipStream_t s, const float* row0_scale = nullptr) {
    const size_t lds = (size_t)r * 64 * sizeof(float);
    if (lds > 160 * 1024) return onmf_fail(c, ONMF_ERR_ARG, "sweep: r=%d needs %zu B of LDS", r, lds);
    static LdsLimit lim;
    ONMF_TRY(onmf_raise_lds_limit(c, lim, (const void*)sweep_column_kernel, lds));
    const int groups = (b + 63) / 64;
    *n_groups = groups;
    hipLaunchKernelGGL(sweep_column_kernel, dim3(groups), dim3(64), lds, s, r, b, Gm, Cm, Hin, Hout,
                       alpha, damp, exact_cd, stats, row0_scale);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

// First sweep of a solve within a radius (src/onmf.py:260-263): the factor of the first row's move, then
// the sweep by the column kernel.  `stats` holds 2 * ceil(b/64) + 1 doubles at least.
int launch_sweep_radius_first(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, const float* Hin,
                              float* Hout, float alpha, float radius, double* stats, int* n_groups,
                              hipStream_t s) {
    if (!(radius > 0.f)) return onmf_fail(c, ONMF_ERR_ARG, "sweep: the radius must be positive");
    const int groups = (b + 63) / 64;
    const float damp = (float)std::sqrt(10.0);
    float* scale = reinterpret_cast<float*>(stats + 2 * (size_t)groups);
    hipLaunchKernelGGL(radius_row0_kernel, dim3(groups), dim3(64), 0, s, r, b, Gm, Cm, Hin, alpha, damp, stats);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(radius_scale_kernel, dim3(1), dim3(64), 0, s, groups, stats, radius, scale);
    HIP_TRY(c, hipGetLastError());
    return launch_sweep_column(c, r, b, Gm, Cm, Hin, Hout, alpha, damp, 0, stats, n_groups, s, scale);
}

bool sweep_tile_covers(int r, int b) {
    return (r + 15) / 16 <= 16 && (double)r * (double)b * 4.0 < 2147483000.0;
}

static int cu_count(onmf_ctx* c) {
    if (c->cu_count == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess) c->cu_count = cus;
    }
    return c->cu_count;
}

// The small-workgroup probing kernel (64 columns per workgroup) when its grid still fits the chip in
// one round - the condition for running all sweeps of a solve in one launch.
static bool use_small_workgroups(onmf_ctx* c, int r, int b, bool probing) {
    return probing && c->sweep_kernel == 0 && sweep_tile_small_covers(r) && (b + 63) / 64 <= cu_count(c);
}

// ... and the 128-column one when that fits and the 64-column one does not.
static bool use_medium_workgroups(onmf_ctx* c, int r, int b, bool probing) {
    return probing && c->sweep_kernel == 0 && !use_small_workgroups(c, r, b, probing) && (b + 127) / 128 <= cu_count(c);
}

bool sweep_has_control_wave_at(onmf_ctx* c, int r, int b) {
    return sweep_tile_covers(r, b) && c->sweep_kernel != 1 && use_small_workgroups(c, r, b, true);
}

int sweep_tile_workgroups(onmf_ctx* c, int r, int b, bool probing) {
    if (use_small_workgroups(c, r, b, probing)) return (b + 63) / 64;
    if (use_medium_workgroups(c, r, b, probing)) return (b + 127) / 128;
    return (b + 255) / 256;
}

int launch_sweep(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* const* hb,
                 float alpha, int exact_cd, int k_first, int k_count, double* stats, int* n_groups,
                 hipStream_t s, const ScreenDev* screen) {
    if (k_count < 1 || k_first < 0) return onmf_fail(c, ONMF_ERR_ARG, "sweep: bad sweep range");
    const bool want_tile = sweep_tile_covers(r, b) && c->sweep_kernel != 1;
    if (!want_tile) {
        if (c->sweep_kernel >= 2)
            return onmf_fail(c, ONMF_ERR_ARG, "sweep: the tile kernel covers r <= 256 (r=%d)", r);
        if (screen || k_count != 1) return onmf_fail(c, ONMF_ERR_ARG, "sweep: probes need the tile kernel");
        const float damp = (float)std::sqrt((double)k_first + 10.0);
        return launch_sweep_column(c, r, b, Gm, Cm, hb[k_first % 3], hb[(k_first + 1) % 3], alpha, damp,
                                   exact_cd, stats, n_groups, s);
    }
    // Column tiles per wave: option values 3 / 4 pin 1 / 2.  Otherwise 2 for the probing kernel
    // (twice the registers per lane: its extra state fits without spills, and half the LDS operand
    // traffic) and 1 for the plain one (4 waves per SIMD hide its shorter chain better).
    if (use_small_workgroups(c, r, b, screen != nullptr))
        return launch_sweep_tile_ct2_small(c, r, b, Gm, Cm, hb, alpha, exact_cd, k_first, k_count, stats, n_groups,
                                           s, screen);
    if (use_medium_workgroups(c, r, b, screen != nullptr))
        return launch_sweep_tile_ct2_medium(c, r, b, Gm, Cm, hb, alpha, exact_cd, k_first, k_count, stats, n_groups,
                                            s, screen);
    if (c->sweep_kernel == 4 || (c->sweep_kernel != 3 && screen))
        return launch_sweep_tile_ct2(c, r, b, Gm, Cm, hb, alpha, exact_cd, k_first, k_count, stats, n_groups, s,
                                     screen);
    return launch_sweep_tile_ct1(c, r, b, Gm, Cm, hb, alpha, exact_cd, k_first, k_count, stats, n_groups, s,
                                 screen);
}
