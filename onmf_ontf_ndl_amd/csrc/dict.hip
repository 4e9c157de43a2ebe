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
//
// That kernel pays one cross-workgroup exchange per atom (microseconds each).  For d <= 1024 a
// second kernel keeps the whole pass inside ONE workgroup (dict_panel_kernel below), where the
// norm costs an LDS round and a barrier; it occupies a single CU and is several times faster.
#include "common.h"

#include <cstdlib>

#include <algorithm>

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
                        ok = (unsigned)(x >> 32) == (unsigned)(j + 1) && spin_limit != SPIN_FORCE_TIMEOUT;
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

// ---------------------------------------------------------------------------------------------
// One-workgroup pass (d <= 1024).  256 threads; thread t owns rows t, t + 256, ... (RPT of them).
// Atoms are taken in panels of 16.  For a panel starting at atom j0 the workgroup first brings its
// entries of Q - B up to date from the deltas of all earlier atoms (left-looking:
// q[row][n] += sum_{k < j0} delta[k][row] * A[k][j0+n], an MFMA product; the deltas are read back
// from `Dt`, stored [atom][row]), then runs the 16 atoms in order: the norm is a wave sum, one LDS
// slot per wave and a barrier; later atoms of the panel take the delta at once, with the entries
// of A wave-uniform (64-byte scalar loads from a copy of A whose rows are padded with zeros to a
// multiple of 16).  W is read and written transposed ([atom][row], contiguous along a wave); the
// launcher transposes on the way in and out.
// ---------------------------------------------------------------------------------------------
// 4x4 transpose between a register index and the lane row (16-lane group) of a wave: afterwards
// register j of lane row g holds what register g of lane row j held (v_permlane32_swap exchanges
// rows 2,3 of its first operand with rows 0,1 of the second; v_permlane16_swap the odd rows of the
// first with the even rows of the second).
__device__ __forceinline__ void xpose4(float& r0, float& r1, float& r2, float& r3) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(r0), __float_as_uint(r2), false, false);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1), __float_as_uint(r3), false, false);
    auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
    auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    r0 = __uint_as_float(c[0]);
    r1 = __uint_as_float(c[1]);
    r2 = __uint_as_float(d[0]);
    r3 = __uint_as_float(d[1]);
}

struct alignas(64) Row16 {
    float v[16];
};

// sum of a row of 16 lanes, left in lane 15 of the row (row_shr 1, 2, 4, 8)
__device__ __forceinline__ float row_total(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, true));
    return x;
}

// sum over the wave, returned wave-uniform: row totals, then row_bcast15 into rows 1 and 3,
// row_bcast31 into rows 2 and 3; lane 63 holds the total
__device__ __forceinline__ float wave_sum_uniform(float x) {
    x = row_total(x);
    // rows outside row_mask keep x (the instruction leaves their destination alone)
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc\n\ts_nop 0"
                 : "+v"(x));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// Everything the panel kernel reads, in ONE launch of many small workgroups (one dependent round
// trip to memory each), because this is the head of a latency chain that runs beside a chip-filling
// product of another stream, where every dependent memory access costs several times its usual time:
//   Qt0[j][i] = sum_k A[k][j] W[i][k] - B[j][i]      (r x d; src/onmf.py:112 before any atom has moved)
//   Wt[k][i]  = W[i][k]                              (r x d)
//   Ap        = A padded to lda columns, except the diagonal, which the pass never reads as an entry
//               of A: Ap[k][k] = -1 / (A[k][k] + 1), the (negated) step of atom k
// A workgroup owns a tile of 32 atoms x 32 rows; per chunk of up to 256 k it stages A[k][J0..J0+31] and
// W[I0..I0+31][k] (transposed) in LDS, 64 KB, and every thread accumulates 2 x 2 outputs.  The tiles
// with J0 == 0 also write their rows of Wt, those with I0 == 0 their columns of Ap.
constexpr int HEAD_T = 32, HEAD_KC = 256;
__global__ __launch_bounds__(256) void dict_head_kernel(int d, int r, int lda, const float* __restrict__ A,
                                                        const float* __restrict__ W, const float* __restrict__ B,
                                                        float* __restrict__ Qt0, float* __restrict__ Wt,
                                                        float* __restrict__ Ap) {
    extern __shared__ __attribute__((aligned(16))) float hs[];
    float* As = hs;                        // [HEAD_KC][32]  As[k][jj] = A[k0 + k][J0 + jj]
    float* Ws = hs + HEAD_KC * HEAD_T;     // [HEAD_KC][32]  Ws[k][ii] = W[I0 + ii][k0 + k]
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    const int I0 = blockIdx.x * HEAD_T, J0 = blockIdx.y * HEAD_T;
    const int tj = 2 * (tid >> 4), ti = 2 * (tid & 15);
    // the tile's entries of B first: their latency passes under the staging
    float bsub[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = J0 + tj + a, i = I0 + ti + c;
            bsub[a][c] = (j < r && i < d) ? B[(size_t)j * d + i] : 0.f;
        }
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    const bool vecA = (r & 3) == 0 && (((uintptr_t)A) & 15) == 0;
    const bool vecW = (r & 3) == 0 && (((uintptr_t)W) & 15) == 0;
    for (int k0 = 0; k0 < r; k0 += HEAD_KC) {
        const int kc = min(HEAD_KC, r - k0);
        if (k0 > 0) __syncthreads();
        // A panel: item = (k, float4 jq): rows of 128 B
        {
            float4 st[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int item = tid + 256 * it, k = item >> 3, jq = item & 7;
                const int j = J0 + 4 * jq;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < kc) {
                    const float* src = A + (size_t)(k0 + k) * r + j;
                    if (vecA && j + 3 < r) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (j + 0 < r) v.x = src[0];
                        if (j + 1 < r) v.y = src[1];
                        if (j + 2 < r) v.z = src[2];
                        if (j + 3 < r) v.w = src[3];
                    }
                }
                st[it] = v;
            }
            // W panel: item = (ii, float4 kq) with ii fastest, so that the transposing LDS stores of a
            // wave fall into 32 different banks
            float4 sw[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int item = tid + 256 * it, ii = item & 31, kq = item >> 5;
                const int i = I0 + ii, k = 4 * kq;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < d && k < kc) {
                    const float* src = W + (size_t)i * r + k0 + k;
                    if (vecW && k + 3 < kc) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (k + 0 < kc) v.x = src[0];
                        if (k + 1 < kc) v.y = src[1];
                        if (k + 2 < kc) v.z = src[2];
                        if (k + 3 < kc) v.w = src[3];
                    }
                }
                sw[it] = v;
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int item = tid + 256 * it, k = item >> 3, jq = item & 7;
                *reinterpret_cast<float4*>(As + k * HEAD_T + 4 * jq) = st[it];
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int item = tid + 256 * it, ii = item & 31, kq = item >> 5;
                Ws[(4 * kq + 0) * HEAD_T + ii] = sw[it].x;
                Ws[(4 * kq + 1) * HEAD_T + ii] = sw[it].y;
                Ws[(4 * kq + 2) * HEAD_T + ii] = sw[it].z;
                Ws[(4 * kq + 3) * HEAD_T + ii] = sw[it].w;
            }
        }
        __syncthreads();
        if (blockIdx.y == 0) {             // Wt[k0 + k][I0 + ii]
            for (int item = tid; item < kc * HEAD_T; item += 256) {
                const int k = item >> 5, ii = item & 31;
                if (I0 + ii < d) Wt[(size_t)(k0 + k) * d + I0 + ii] = Ws[k * HEAD_T + ii];
            }
        }
        if (blockIdx.x == 0) {             // Ap[k0 + k][J0 + jj]
            for (int item = tid; item < kc * HEAD_T; item += 256) {
                const int k = item >> 5, jj = item & 31;
                const int j = J0 + jj;
                if (j < lda) {
                    float v = As[k * HEAD_T + jj];                 // 0 for j >= r
                    if (j == k0 + k) v = -1.0f / (v + 1.0f);
                    Ap[(size_t)(k0 + k) * lda + j] = v;
                }
            }
        }
#pragma unroll 8
        for (int k = 0; k < kc; ++k) {
            const float2 a = *reinterpret_cast<const float2*>(As + k * HEAD_T + tj);
            const float2 w = *reinterpret_cast<const float2*>(Ws + k * HEAD_T + ti);
            acc[0][0] = fmaf(a.x, w.x, acc[0][0]);
            acc[0][1] = fmaf(a.x, w.y, acc[0][1]);
            acc[1][0] = fmaf(a.y, w.x, acc[1][0]);
            acc[1][1] = fmaf(a.y, w.y, acc[1][1]);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int j = J0 + tj + a, i = I0 + ti + c;
            if (j < r && i < d) Qt0[(size_t)j * d + i] = acc[a][c] - bsub[a][c];
        }
}

#ifdef ONMF_TRACE_WG
// Diagnostic build only (tools/wg_trace.py): when and where the pass's workgroups ran.
__device__ unsigned long long g_panel_trace[4 * 4096];
__device__ unsigned g_panel_trace_n;
#endif

template <int RPT>
__global__ __launch_bounds__(256) void dict_panel_kernel(int d, int r, int lda,
                                                         const float* __restrict__ Qt0,   // A^T W^T - B
                                                         const float* __restrict__ Ap,
                                                         const float* __restrict__ Wt,
                                                         float* __restrict__ WtOut, float* Dt, int nwg,
                                                         u64* granules, int* timeout_flag, long long spin_limit,
                                                         int j_begin, int j_end) {
    // atoms j_begin .. j_end-1 (multiples of the panel width): a later launch goes on where an earlier
    // one stopped - all it needs of the earlier atoms are their deltas in Dt (the left-looking update
    // reads them from memory anyway)
    // nwg > 1: workgroup g owns rows g * 256 RPT .. (g + 1) * 256 RPT - 1 and the workgroups exchange
    // their parts of every atom's squared norm (below); everything else is per row.
    constexpr int PW = 16;
#ifdef ONMF_TRACE_WG
    const unsigned long long trace_t0 = wall_clock64();
#endif
    // a latency chain that shares its CU with workgroups of the products running beside it: its
    // instructions go first
    __builtin_amdgcn_s_setprio(3);
    const int wg = blockIdx.x;
    const int row0 = wg * (RPT * 256);
    bool dead = false;                                        // gave up waiting for the other workgroups
    __shared__ __attribute__((aligned(16))) float part[2][16];   // one slot per wave; unused slots stay 0
    __shared__ __attribute__((aligned(16))) float ablk[256];     // the panel's 16 x 16 block of Ap
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, c16 = lane & 15, g4 = lane >> 4;
    const bool lane0 = lane == 0;
    if (tid < 32) part[tid >> 4][tid & 15] = 0.f;
    __syncthreads();
    // All [atom][row] arrays go through buffer descriptors: byte address = per-thread row offset +
    // wave-uniform atom offset, no per-access vector arithmetic; rows past d and atoms past r read
    // 0 and drop their stores (range check), so they need no masks: their u is max(0 - 0, 0) = 0.
    const unsigned nbytes = (unsigned)r * (unsigned)d * 4u;
    const auto rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)Qt0, 0, nbytes, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, nbytes, 0x00020000);
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)WtOut, 0, nbytes, 0x00020000);
    const auto rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)Dt, 0, nbytes, 0x00020000);
    const auto rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (unsigned)r * (unsigned)lda * 4u, 0x00020000);
    const int rowbytes = d * 4;
    int voff[RPT];            // the thread's own rows: p * 256 + tid
    int voffB[RPT][4];        // MFMA B operand: row p * 256 + wave * 64 + 16 mt + c16, atom offset g4
#pragma unroll
    for (int p = 0; p < RPT; ++p) {
        const int i = row0 + p * 256 + tid;
        voff[p] = i < d ? i * 4 : (int)0x80000000;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int rowb = row0 + p * 256 + wave * 64 + 16 * mt + c16;
            voffB[p][mt] = rowb < d ? (rowb + g4 * d) * 4 : (int)0x80000000;
        }
    }
    const int voffA = (g4 * lda + c16) * 4;

    for (int j0 = j_begin; j0 < j_end; j0 += PW) {
        float q[RPT][PW], w[RPT][PW];
        // this panel's entries of Q0 - B (formed by the product's epilogue) and of W
#pragma unroll
        for (int p = 0; p < RPT; ++p) {
#pragma unroll
            for (int n = 0; n < PW; ++n) {
                const int so = (j0 + n) * rowbytes;
                q[p][n] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_q, voff[p], so, 0));
                w[p][n] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_w, voff[p], so, 0));
            }
            __builtin_amdgcn_sched_barrier(0);           // one row set at a time: 48 loads in flight, not 192
        }
        // Left-looking update from the atoms before the panel, on the matrix cores (twice the
        // vector rate for fp32).  Per wave, for each of its 4 RPT tiles of 16 rows:
        //     D[atom n][row] += sum_k Ap[k][j0 + n] * Dt[k][row]        (v_mfma_f32_16x16x4_f32)
        // A operand: lane (g, c) = Ap[k0 + g][j0 + c]; B operand: Dt[k0 + g][tile row c].
        // D leaves atoms 4g..4g+3 of row c in lane (g, c); a 4x4 transpose between the tile index
        // and the lane row then hands lane 16g + c all 16 atoms of ITS row (tile g, row c).
        // Operands are fetched four groups of four atoms ahead (one wave per SIMD: nothing else
        // covers the L2 latency).
        {
            f32x4 D[RPT][4];
#pragma unroll
            for (int p = 0; p < RPT; ++p)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) D[p][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int ngroups = j0 >> 2;                      // a multiple of 4
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, b0[RPT][4], b1[RPT][4], b2[RPT][4], b3[RPT][4];
            auto fetch = [&](int kg, float& a, float (&bv)[RPT][4]) {
                a = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_a, voffA, (4 * kg * lda + j0) * 4, 0));
#pragma unroll
                for (int p = 0; p < RPT; ++p)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        bv[p][mt] = __uint_as_float(
                            __builtin_amdgcn_raw_buffer_load_b32(rs_d, voffB[p][mt], 4 * kg * rowbytes, 0));
            };
            auto apply = [&](float a, const float (&bv)[RPT][4]) {
#pragma unroll
                for (int p = 0; p < RPT; ++p)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        D[p][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv[p][mt], D[p][mt], 0, 0, 0);
            };
            if (ngroups > 0) {                                // ngroups is a multiple of 4
                fetch(0, a0, b0);
                fetch(1, a1, b1);
                fetch(2, a2, b2);
                fetch(3, a3, b3);
            }
            for (int kg = 0; kg < ngroups; kg += 4) {         // past the end: fetches the last group again
                apply(a0, b0);
                fetch(min(kg + 4, ngroups - 1), a0, b0);
                apply(a1, b1);
                fetch(min(kg + 5, ngroups - 1), a1, b1);
                apply(a2, b2);
                fetch(min(kg + 6, ngroups - 1), a2, b2);
                apply(a3, b3);
                fetch(min(kg + 7, ngroups - 1), a3, b3);
            }
#pragma unroll
            for (int p = 0; p < RPT; ++p) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float x0 = D[p][0][t], x1 = D[p][1][t], x2 = D[p][2][t], x3 = D[p][3][t];
                    xpose4(x0, x1, x2, x3);
                    q[p][t] += x0;
                    q[p][4 + t] += x1;
                    q[p][8 + t] += x2;
                    q[p][12 + t] += x3;
                }
            }
        }
        // The panel's atoms in order (padded atoms j >= r run on zeros of Ap: harmless, not stored).
        // This chain is what bounds the pass: per atom the norm takes 7 cross-lane steps per wave,
        // one LDS slot per wave, a barrier that orders LDS traffic only (waiting for the global
        // stores as well, as __syncthreads would, costs a memory round trip per atom), one LDS read
        // per lane and 5 more cross-lane steps; 1 / max(1, norm) is one v_rsq_f32.  Four waves (one
        // per SIMD) run it: every wave repeats the chain, and waves that share a SIMD would take
        // turns at every one of its steps.  The row of A for the next atom is fetched right after
        // the barrier, a whole atom ahead of its use.
        // The panel's 16 x 16 block of Ap goes through LDS: every lane then reads the same address
        // (a broadcast), in order with the other LDS traffic of the chain, where a scalar load
        // would make each wait for LDS also a wait for the scalar cache.
        __syncthreads();                                      // previous panel's readers are done
        for (int e = tid; e < 256; e += (int)blockDim.x)       // rows past r read as 0 (range check)
            ablk[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                rs_a, ((e >> 4) * lda + (e & 15)) * 4, (j0 * lda + j0) * 4, 0));
        __syncthreads();
        float acur[PW];
#pragma unroll
        for (int m = 0; m < PW; m += 4) {
            const float4 v = *reinterpret_cast<const float4*>(ablk + m);
            acur[m] = v.x; acur[m + 1] = v.y; acur[m + 2] = v.z; acur[m + 3] = v.w;
        }
#pragma unroll
        for (int n = 0; n < PW; ++n) {
            const int j = j0 + n;
            const float ninv = acur[n];                       // diagonal slot: -1 / (A[j][j] + 1)
            float u[RPT];
            float ss = 0.f;
#pragma unroll
            for (int p = 0; p < RPT; ++p) {
                u[p] = fmaxf(fmaf(q[p][n], ninv, w[p][n]), 0.f);
                ss = fmaf(u[p], u[p], ss);
            }
            const float ws = wave_sum_uniform(ss);
            if (lane0) part[n & 1][wave] = ws;
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the LDS write has landed
            __builtin_amdgcn_s_barrier();
            // the 16 slots, one per lane of a row, summed by the same tree in every wave
            const float psum = part[n & 1][c16];
            float anext[PW];
#pragma unroll
            for (int m = 0; m < PW; ++m) anext[m] = acur[m];
            if (n + 1 < PW) {
#pragma unroll
                for (int m = 0; m < PW; m += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(ablk + (n + 1) * 16 + m);
                    anext[m] = v.x; anext[m + 1] = v.y; anext[m + 2] = v.z; anext[m + 3] = v.w;
                }
            }
            float total = __builtin_bit_cast(
                float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, row_total(psum)), 15));
            if (nwg > 1) {
                // This workgroup's part goes out as one aligned 8-byte word {tag = j + 1, value}; every
                // wave then reads all nwg words itself (lane g reads workgroup g's) until the tags match
                // and adds them in lane order - the same number in every wave of every workgroup.
                // The wait is bounded; after a timeout the pass runs on without waiting (its result is
                // discarded: the host reports the flag).
                if (wave == 0 && lane0) {
                    const u64 word = ((u64)(unsigned)(j + 1) << 32) | (u64)__float_as_uint(total);
                    __hip_atomic_store(&granules[(size_t)j * nwg + wg], word, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
                float val = 0.f;
                const long long t0 = wall_clock64();
                while (!dead) {
                    bool ok = true;
                    if (lane < nwg) {
                        const u64 x = __hip_atomic_load(&granules[(size_t)j * nwg + lane], __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
                        ok = (unsigned)(x >> 32) == (unsigned)(j + 1) && spin_limit != SPIN_FORCE_TIMEOUT;
                        val = __uint_as_float((unsigned)x);
                    }
                    if (__all(ok)) break;
                    if (wall_clock64() - t0 > spin_limit) {
                        dead = true;                          // wave-uniform (one clock value per wave)
                        if (lane0) atomicExch(timeout_flag, 1);
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                total = wave_sum_uniform(lane < nwg ? val : 0.f);
            }
            const float scale = total > 1.0f ? __builtin_amdgcn_rsqf(total) : 1.0f;
#pragma unroll
            for (int p = 0; p < RPT; ++p) {
                const float wn = u[p] * scale;
                const float dv = wn - w[p][n];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(wn), rs_o, voff[p], j * rowbytes, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dv), rs_d, voff[p], j * rowbytes, 0);
#pragma unroll
                for (int m = 0; m < PW; ++m)             // (constant bounds: unrolls before n is known)
                    if (m > n) q[p][m] = fmaf(acur[m], dv, q[p][m]);
            }
#pragma unroll
            for (int m = 0; m < PW; ++m) acur[m] = anext[m];
        }
    }
#ifdef ONMF_TRACE_WG
    if (threadIdx.x == 0) {
        const unsigned slot = atomicAdd(&g_panel_trace_n, 1u);
        if (slot < 4096u) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_panel_trace[4 * slot + 0] = trace_t0;
            g_panel_trace[4 * slot + 1] = wall_clock64();
            g_panel_trace[4 * slot + 2] = ((unsigned long long)xcc << 32) | hw;
            g_panel_trace[4 * slot + 3] = ((unsigned long long)j_begin << 32) | (unsigned)j_end;
        }
    }
#endif
}


__global__ void zero_u64_kernel(u64* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
         i += (size_t)gridDim.x * blockDim.x)
        p[i] = 0ull;
}

// ---------------------------------------------------------------------------------------------
// The pass with no waiting between workgroups: per atom j one launch that forms
//     u = max(0, W[:,j] - (W A[:,j] - B[j,:]^T) / (A[j,j] + 1))
// from the current W (rows are independent: one thread per row, the row's r entries of W against
// column j of A) and leaves one partial sum of squares per workgroup, and a second launch that adds
// the partials in a fixed order and writes W[:,j] = u / max(1, ||u||).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dict_atom_move_kernel(int d, int r, int j, const float* __restrict__ W,
                                                             const float* __restrict__ A,
                                                             const float* __restrict__ B, float* __restrict__ u,
                                                             float* __restrict__ partial) {
    __shared__ float red[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    if (i < d) {
        const float* __restrict__ wrow = W + (size_t)i * r;
        float dot = 0.f;
        for (int k = 0; k < r; ++k) dot = fmaf(wrow[k], A[(size_t)k * r + j], dot);
        const float inv = 1.0f / (A[(size_t)j * r + j] + 1.0f);
        v = fmaxf(wrow[j] - (dot - B[(size_t)j * d + i]) * inv, 0.f);
        u[i] = v;
    }
    const float ws = wave_sum(v * v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void dict_atom_scale_kernel(int d, int r, int j, float* __restrict__ W,
                                                              const float* __restrict__ u,
                                                              const float* __restrict__ partial, int nparts) {
    float total = 0.f;
    for (int p = 0; p < nparts; ++p) total += partial[p];      // same order in every thread
    const float scale = 1.0f / fmaxf(1.0f, sqrtf(total));
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < d) W[(size_t)i * r + j] = u[i] * scale;
}


// ---------------------------------------------------------------------------------------------
// The pass redone on the device: enqueued behind every pass whose workgroups wait for each other,
// it looks at that pass's timeout flag and returns at once when it is clear (the normal case: one
// near-empty launch).  When a workgroup did give up waiting, this ONE workgroup of 256 threads (few
// registers: it finds a place on a chip full of the products' workgroups) redoes
// the whole pass with nobody to wait for - atom by atom, the arithmetic and summation order of
// launch_dict_pass_serial - so the host never has to wait for a pass to learn how it went.  `events`
// (host memory) counts the redone passes.  src: W_in, or (transposed != 0) the r x d copy the head
// kernel staged, for an in-place pass whose W went with the discarded result.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dict_redo_kernel(int d, int r, const float* __restrict__ src, int transposed,
                                                        const float* __restrict__ A, const float* __restrict__ B,
                                                        float* W, const int* __restrict__ flag, int* events) {
    if (*flag == 0) return;                                   // uniform
    __shared__ float red[4];
    __shared__ float part[64];                                // one partial sum of squares per 256 rows (d <= 16384)
    const int tid = threadIdx.x;
    // (a load and a store, not an atomic: host memory need not support atomics from the device, and two of
    // these workgroups at once on one context would take two timeouts on two streams)
    if (tid == 0) {
        const int seen = __hip_atomic_load(events, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(events, seen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const size_t n = (size_t)d * r;
    if (transposed) {
        for (size_t e = tid; e < n; e += 256) W[e] = src[(e % r) * d + e / r];
    } else if (src != W) {
        for (size_t e = tid; e < n; e += 256) W[e] = src[e];
    }
    __threadfence_block();
    __syncthreads();
    const int nparts = (d + 255) / 256;
    for (int j = 0; j < r; ++j) {
        const float inv = 1.0f / (A[(size_t)j * r + j] + 1.0f);
        // rows are independent (row i reads row i only): the unscaled entry goes straight into W
        for (int p = 0; p < nparts; ++p) {
            const int i = 256 * p + tid;
            float v = 0.f;
            if (i < d) {
                float* wrow = W + (size_t)i * r;
                float dot = 0.f;
                for (int k = 0; k < r; ++k) dot = fmaf(wrow[k], A[(size_t)k * r + j], dot);
                v = fmaxf(wrow[j] - (dot - B[(size_t)j * d + i]) * inv, 0.f);
                wrow[j] = v;
            }
            const float ws = wave_sum(v * v);
            if ((tid & 63) == 0) red[tid >> 6] = ws;
            __syncthreads();
            if (tid == 0) part[p] = (red[0] + red[1]) + (red[2] + red[3]);
            __syncthreads();
        }
        float total = 0.f;
        for (int p = 0; p < nparts; ++p) total += part[p];    // same order in every thread
        const float scale = 1.0f / fmaxf(1.0f, sqrtf(total));
        for (int p = 0; p < nparts; ++p) {
            const int i = 256 * p + tid;
            if (i < d) W[(size_t)i * r + j] *= scale;
        }
        __threadfence_block();
        __syncthreads();
    }
}

}  // namespace

bool dict_pass_splits(const onmf_ctx* c, int d, int r) { return d <= 16 * 1024 && c->dict_kernel != 1 && r >= 32; }

static inline size_t up64(size_t n) { return (n + 63) & ~(size_t)63; }

// Qt | Ap | Dt | Wt | WtOut, every part starting on a 256-byte boundary
size_t dict_workspace_floats(int d, int r) {
    const size_t lda = (size_t)(r + 15) / 16 * 16;
    return 4 * up64((size_t)d * r) + up64((size_t)r * lda);
}

const float* dict_staged_wt(const onmf_ctx* c, int d, int r) {
    const size_t lda = (size_t)(r + 15) / 16 * 16;
    if (!c->ddelta.p || c->ddelta.bytes < (3 * (size_t)d * r + (size_t)r * lda + 16) * sizeof(float)) return nullptr;
    return (const float*)c->ddelta.p + (size_t)r * lda + (size_t)d * r;
}

int launch_dict_pass_serial(onmf_ctx* c, int d, int r, const float* Win, const float* A, const float* B,
                            float* Wout, hipStream_t s) {
    const int nparts = (d + 255) / 256;
    ONMF_TRY(onmf_reserve(c, c->small, ((size_t)d + nparts + 2 * (size_t)r * r) * sizeof(float), "dictionary atom scratch"));
    float* u = (float*)c->small.p + 2 * (size_t)r * r;
    float* partial = u + d;
    if (Wout != Win)
        HIP_TRY(c, hipMemcpyAsync(Wout, Win, (size_t)d * r * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (int j = 0; j < r; ++j) {
        hipLaunchKernelGGL(dict_atom_move_kernel, dim3(nparts), dim3(256), 0, s, d, r, j, Wout, A, B, u, partial);
        hipLaunchKernelGGL(dict_atom_scale_kernel, dim3(nparts), dim3(256), 0, s, d, r, j, Wout, u, partial, nparts);
    }
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

int launch_dict_pass(onmf_ctx* c, int d, int r, const float* Win, const float* A, const float* B,
                     float* Wout, hipStream_t s, int part, float* ws) {
    if (part != 0 && !dict_pass_splits(c, d, r))
        return onmf_fail(c, ONMF_ERR_ARG, "update_dict: this shape is not taken in two parts");
    if (ws && (d > 1024 || part != 0))
        return onmf_fail(c, ONMF_ERR_ARG, "update_dict: a caller's workspace serves the one-workgroup pass only (d <= 1024)");
    if (part != 2 && !ws) c->dict_coop_last = false;
    // bound of the workgroups' waits for each other (the same option bounds the sweeps' waits)
    const long long spin_limit = c->spin_limit;
    if (d <= 16 * 1024 && (c->dict_kernel != 1 || ws)) {
        // one workgroup per 1024 rows; several exchange the atoms' norms (they are few: co-resident)
        const int nwg = (d + 1023) / 1024;
        const int lda = (r + 15) / 16 * 16;
        // part 1: atoms before the middle panel boundary, part 2: the rest, part 0: all
        // (measured at d=1024, r=256, first part = n/16 of the atoms: 1.59 ms per step for n = 8, 9, 10 and
        // 1.70 ms for n = 5, 6, 7 - a step, not a slope: with the longer second part the aggregate
        // product beside it takes 435 us instead of 330 in most steps, one of its workgroups starting late)
        const int jmid = std::max(1, lda / 16 / 2) * 16;
        const int j_begin = part == 2 ? jmid : 0, j_end = part == 1 ? jmid : lda;
        float *Qt, *Ap, *Dt, *Wt, *WtOut;
        if (ws) {
            const size_t dr = up64((size_t)d * r);
            Qt = ws;
            Ap = Qt + dr;                                // 64-byte aligned rows
            Dt = Ap + up64((size_t)r * lda);
            Wt = Dt + dr;
            WtOut = Wt + dr;
        } else {
            ONMF_TRY(onmf_reserve(c, c->qt, (size_t)r * d * sizeof(float), "dictionary Qt"));
            Qt = (float*)c->qt.p;
            ONMF_TRY(onmf_reserve(c, c->ddelta, (3 * (size_t)d * r + (size_t)r * lda + 16) * sizeof(float),
                                  "dictionary deltas"));
            Ap = (float*)c->ddelta.p;                    // 64-byte aligned rows
            Dt = Ap + (size_t)r * lda;
            Wt = Dt + (size_t)d * r;                     // W^T in, r x d
            WtOut = Wt + (size_t)d * r;
        }
        u64* gran = nullptr;
        int* flag = nullptr;
        if (nwg > 1) {
            const size_t ngran = (size_t)lda * nwg;                  // padded atoms exchange too
            ONMF_TRY(onmf_reserve(c, c->sync, ngran * sizeof(u64) + 64, "dictionary sync words"));
            gran = (u64*)c->sync.p;
            c->dict_flag_off = ngran * sizeof(u64);
            flag = (int*)((char*)c->sync.p + c->dict_flag_off);
            if (part != 2) {
                hipLaunchKernelGGL(zero_u64_kernel, dim3((unsigned)((ngran + 8 + 255) / 256)), dim3(256), 0, s, gran,
                                   ngran + 8);
                HIP_TRY(c, hipGetLastError());
            }
        }
        if (part != 2) {
            // Qt0 = A^T W^T - B, Wt = W^T and the padded A, one launch
            static LdsLimit lim;
            constexpr size_t head_lds = 2 * (size_t)HEAD_KC * HEAD_T * sizeof(float);
            ONMF_TRY(onmf_raise_lds_limit(c, lim, (const void*)dict_head_kernel, head_lds));
            hipLaunchKernelGGL(dict_head_kernel, dim3((d + HEAD_T - 1) / HEAD_T, (lda + HEAD_T - 1) / HEAD_T),
                               dim3(256), head_lds, s, d, r, lda, A, Win, B, Qt, Wt, Ap);
            HIP_TRY(c, hipGetLastError());
        }
        // 256 threads, thread t owns rows t, t + 256, ... of its workgroup's 1024
        const int rpt = nwg > 1 ? 4 : (d + 255) / 256;
        const int nt = rpt > 1 ? 256 : (d + 63) / 64 * 64;
        // (At four rows per thread the workgroup takes 392 of a SIMD's 512 registers per lane, so no
        // workgroup of the products that run beside it - 216 registers - fits on its CU: it has the CU to
        // itself, and a product that counts on all 16 places of a shader engine leaves one workgroup
        // waiting for the pass to end.)
#define PANEL_CASE(RPTV)                                                                              \
    hipLaunchKernelGGL((dict_panel_kernel<RPTV>), dim3(nwg), dim3(nt), 0, s, d, r, lda, Qt, Ap, Wt, WtOut, Dt, \
                       nwg, gran, flag, spin_limit, j_begin, j_end);
        if (rpt <= 1) PANEL_CASE(1)
        else if (rpt <= 2) PANEL_CASE(2)
        else PANEL_CASE(4)
#undef PANEL_CASE
        HIP_TRY(c, hipGetLastError());
        if (part == 1) return ONMF_OK;
        ONMF_TRY(launch_transpose(r, d, WtOut, Wout, s));
        if (nwg > 1) {
            // what to do if a workgroup gave up waiting: decided and done on the device (one near-empty
            // launch when all went well), the host does not wait for the pass
            const bool in_place = Win == Wout;
            hipLaunchKernelGGL(dict_redo_kernel, dim3(1), dim3(256), 0, s, d, r, in_place ? (const float*)Wt : Win,
                               in_place ? 1 : 0, A, B, Wout, (const int*)flag, c->host_events_dev);
            HIP_TRY(c, hipGetLastError());
        }
        return ONMF_OK;
    }
    if (c->dict_kernel == 2)
        return onmf_fail(c, ONMF_ERR_ARG, "update_dict: the panel kernel covers d <= 16384 (d=%d)", d);
    c->dict_coop_last = true;
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
    c->dict_flag_off = ngran * sizeof(u64);
    int* flag = (int*)((char*)c->sync.p + c->dict_flag_off);
    hipLaunchKernelGGL(zero_u64_kernel, dim3((unsigned)((ngran + 8 + 255) / 256)), dim3(256), 0, s, gran,
                       ngran + 8);
    HIP_TRY(c, hipGetLastError());
    static LdsLimit lim;
    ONMF_TRY(onmf_raise_lds_limit(c, lim, (const void*)dict_pass_kernel, lds));
    hipLaunchKernelGGL(dict_pass_kernel, dim3(nwg), dim3(64), lds, s, d, r, nwg, Qt, A, B, Win, Wout,
                       gran, flag, spin_limit);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

#ifdef ONMF_TRACE_WG
extern "C" int onmf_debug_panel_trace(unsigned long long* host, int max_records) {
    unsigned n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_panel_trace_n), sizeof(n)) != hipSuccess) return -1;
    if (n > 4096u) n = 4096u;
    if ((int)n > max_records) n = (unsigned)max_records;
    if (n && hipMemcpyFromSymbol(host, HIP_SYMBOL(g_panel_trace), (size_t)n * 32) != hipSuccess) return -1;
    const unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_panel_trace_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return (int)n;
}
#endif
