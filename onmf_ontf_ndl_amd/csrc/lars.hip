// Positive LARS-lasso coder of Online_NTF on gfx950.
//
// The reference hands every column of a minibatch to scikit-learn (src/ontf.py:79-86):
//   SparseCoder(dictionary=W.T, transform_alpha=alpha or 2, transform_algorithm='lasso_lars',
//               positive_code=True).transform(X.T)
// which runs, per column, sklearn.linear_model._least_angle._lars_path_solver with method="lasso",
// positive=True, a precomputed Gram matrix W^T W and Xy = W^T x, return_path=False, alpha_min = alpha / d.
// That routine does NOT end at the optimum of the convex problem: when it runs out of atoms with a positive
// correlation it takes the largest correlation among the INACTIVE atoms for the active set's common
// correlation and interpolates the last segment with it.  The learned dictionary depends on those codes, so
// this file follows the routine event by event instead of solving the convex problem:
//   * `lars_pos_kernel`: one column per wavefront.  Where the library keeps a Cholesky factor of the active
//     Gram matrix and solves two triangular systems per event (two dependent chains per event), this kernel
//     keeps the INVERSE of the active Gram matrix, updated by one symmetric rank-one step when an atom
//     enters or leaves - every pass over it is a loop of independent row operations across the lanes.
//     All of the column's state is double precision: the inverse (packed symmetric in LDS, a slot per
//     active atom, freed slots reused; in a global workspace when it does not fit), the correlations, the
//     coefficients.  Lane l owns atoms l, l + 64, ... and slots l, l + 64, ...
//   * `prod_f64_kernel`: W^T W and W^T X accumulated in double (`v_mfma_f64_16x16x4_f64`) from the float
//     operands.  With float accumulation the rounding of the two products alone moves the codes by
//     2e-5 ... 2e-4 (the active Gram matrices have condition numbers of 1e3 ... 1e4); in double the codes
//     follow the reference's to 1e-7.
// Deviations from the routine, all at the level of exact ties or rounding: ties of the largest correlation
// go to the lowest atom index (the routine: first in its permuted order); correlations are not rounded to
// 15 decimals; atoms that reach zero in the same event all leave (the routine miscounts them); after a
// degenerate pivot (< 1e-7) the atom's correlation is zeroed and nothing else is permuted.
#include "common.h"

#include <cfloat>
#include <climits>

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Out[m][n] = sum_k A[k*lda + m] * B[k*ldb + n]   (float operands, double accumulation and result)
// 64 x 64 output tile per workgroup of four waves; wave w: rows 16w .. 16w+15, four 16-column blocks.
// ---------------------------------------------------------------------------------------------
constexpr int PT = 64, PK = 64, PPITCH = 80;

// NBL: 16-column blocks per wave (output tile 64 x 16 NBL).  Two instead of four doubles the number of
// workgroups: at the shapes of this path (a hundred atoms, a few thousand columns) the grid is otherwise one
// four-wave workgroup per CU, and one wave per SIMD leaves every LDS-read -> convert -> MFMA chain exposed.
template <int NBL>
// A second product with the same left operand rides in the same launch: the column tiles from tile n1_tiles
// on contract with B2 (N2 columns) and go to Out2.
__global__ __launch_bounds__(256) void prod_f64_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                       const float* __restrict__ B, int ldb,
                                                       double* __restrict__ Out, int ldo, int n1_tiles,
                                                       const float* __restrict__ B2, int ldb2, int N2,
                                                       double* __restrict__ Out2, int ldo2) {
    constexpr int PTN = 16 * NBL;                     // columns of the output tile
    constexpr int BPITCH = PTN + 16;
    __shared__ float sa[PK * PPITCH], sb[PK * BPITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * PT;
    int n0 = blockIdx.x * PTN;
    if ((int)blockIdx.x >= n1_tiles) {                // (workgroup-uniform) the second product's tiles
        n0 = ((int)blockIdx.x - n1_tiles) * PTN;
        B = B2; ldb = ldb2; N = N2; Out = Out2; ldo = ldo2;
    }
    f64x4 acc[NBL];
#pragma unroll
    for (int i = 0; i < NBL; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
    // a tile is PK contraction steps x 64 (A) or PTN (B) columns: thread t fetches four consecutive columns of
    // rows (t >> 4) + 16 j of A and of rows (t / (PTN / 4)) + (1024 / PTN) j of B - the next tile's loads are
    // in flight while this one is multiplied
    const int lk = tid >> 4, lx = (tid & 15) * 4;
    constexpr int BROWS = 1024 / PTN;                 // rows of B one pass of the workgroup covers
    const int bk = tid / (PTN / 4), bx = (tid % (PTN / 4)) * 4;
    float ra[PK / 16][4], rb[PK / BROWS][4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < PK / 16; ++j) {
            const int k = k0 + lk + 16 * j;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + lx + i;
                ra[j][i] = (k < K && m < M) ? A[(size_t)k * lda + m] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < PK / BROWS; ++j) {
            const int k = k0 + bk + BROWS * j;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + bx + i;
                rb[j][i] = (k < K && n < N) ? B[(size_t)k * ldb + n] : 0.f;
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += PK) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PK / 16; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) sa[(lk + 16 * j) * PPITCH + lx + i] = ra[j][i];
#pragma unroll
        for (int j = 0; j < PK / BROWS; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) sb[(bk + BROWS * j) * BPITCH + bx + i] = rb[j][i];
        __syncthreads();
        if (k0 + PK < K) fetch(k0 + PK);
#pragma unroll
        for (int kk = 0; kk < PK / 4; ++kk) {
            const int krow = kk * 4 + (lane >> 4);
            const double av = (double)sa[krow * PPITCH + wave * 16 + (lane & 15)];
#pragma unroll
            for (int nb = 0; nb < NBL; ++nb) {
                const double bv = (double)sb[krow * BPITCH + nb * 16 + (lane & 15)];
                acc[nb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[nb], 0, 0, 0);
            }
        }
    }
    // result register i of a lane: row (lane >> 4) + 4 i, column lane & 15 of the 16 x 16 block
#pragma unroll
    for (int nb = 0; nb < NBL; ++nb) {
        const int n = n0 + nb * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wave * 16 + (lane >> 4) + 4 * i;
            if (m < M && n < N) Out[(size_t)m * ldo + n] = acc[nb][i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// wave-wide helpers on doubles.  Rows of 16 lanes are reduced with DPP rotations (no LDS crossbar), the
// four row results are fetched with v_readlane; every lane (and the scalar unit) ends with the same bits.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
constexpr int ROR1 = 0x121, ROR2 = 0x122, ROR4 = 0x124, ROR8 = 0x128;      // row_ror:n

__device__ __forceinline__ double lane_value(double v, int l) {       // l wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<ROR1>(v);
    v += dpp_f64<ROR2>(v);
    v += dpp_f64<ROR4>(v);
    v += dpp_f64<ROR8>(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ double wave_min(double v) {
    v = fmin(v, dpp_f64<ROR1>(v));
    v = fmin(v, dpp_f64<ROR2>(v));
    v = fmin(v, dpp_f64<ROR4>(v));
    v = fmin(v, dpp_f64<ROR8>(v));
    return fmin(fmin(lane_value(v, 0), lane_value(v, 16)), fmin(lane_value(v, 32), lane_value(v, 48)));
}
__device__ __forceinline__ int wave_min(int v) {
    v = min(v, dpp_i32<ROR1>(v));
    v = min(v, dpp_i32<ROR2>(v));
    v = min(v, dpp_i32<ROR4>(v));
    v = min(v, dpp_i32<ROR8>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// largest value, lowest index among equals
__device__ __forceinline__ void take_better(double& v, int& idx, double ov, int oi) {
    if (ov > v || (ov == v && oi < idx)) {
        v = ov;
        idx = oi;
    }
}
__device__ __forceinline__ void wave_argmax(double& v, int& idx) {
    take_better(v, idx, dpp_f64<ROR1>(v), dpp_i32<ROR1>(idx));
    take_better(v, idx, dpp_f64<ROR2>(v), dpp_i32<ROR2>(idx));
    take_better(v, idx, dpp_f64<ROR4>(v), dpp_i32<ROR4>(idx));
    take_better(v, idx, dpp_f64<ROR8>(v), dpp_i32<ROR8>(idx));
    double bv = lane_value(v, 0);
    int bi = __builtin_amdgcn_readlane(idx, 0);
#pragma unroll
    for (int l = 16; l < 64; l += 16) take_better(bv, bi, lane_value(v, l), __builtin_amdgcn_readlane(idx, l));
    v = bv;
    idx = bi;
}
// what one pass over the inverse wrote must be visible to the next pass's lanes (same wave)
__device__ __forceinline__ void pass_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

struct LarsArgs {
    int r, b;
    const double* G;        // r x r
    const double* Cd;       // r x ldc: column c = W^T x_c
    int ldc;
    double alpha;           // transform_alpha (the penalty of the convex problem)
    double n_samples;       // d: the library scales the penalty and the path's alphas by it
    int max_iter;
    float* H;               // r x ldh out
    int ldh;
    int* info;              // nullable: per column {events, how the path ended (1 alpha reached, 2 all atoms
                            // active or event limit, 3 alpha went up, 4 safety limit), active atoms, atoms dropped}
    double* Pws;            // global workspace of the inverse (P_LDS == false), p_stride doubles per wave
    size_t p_stride;
    int wave_doubles;       // LDS doubles per wave
    int n_cap;              // slots of the inverse (active atoms a column can hold at a time)
    // meta[0]: columns that needed more than n_cap slots (their indices go to `overflow`; a second launch
    // with room for every atom codes them), meta[1]: most slots any column of the call used
    int* meta;
    int* overflow;
    const int* cols;        // not null: the launch codes columns cols[0 .. *n_cols - 1] (the overflow list)
    const int* n_cols;
};

// The inverse of the active Gram matrix, indexed by slot.  PACKED: lower triangle, row after row (LDS);
// else a full square with leading dimension ld (global memory: rows are read coalesced).
template <bool PACKED>
struct Inv {
    double* p;
    int ld;
    __device__ __forceinline__ double& at(int i, int j) const {
        if (PACKED) {
            const int hi = i >= j ? i : j, lo = i >= j ? j : i;
            return p[(hi * (hi + 1) >> 1) + lo];
        }
        return p[(size_t)i * ld + j];
    }
    // the lane of column j changes entry (i, j) of row i: in packed form each entry exists once
    __device__ __forceinline__ bool mine(int i, int j) const { return PACKED ? j <= i : true; }
};

// The used slots GN at a time: f(q, pos, n) with pos[0 .. n-1] the slots' positions in word q of the mask
// (pos[n .. GN-1] repeat pos[0]: callers give them weight zero).  A pass over the inverse, or over rows of G,
// is a chain of dependent memory round trips when it takes its rows one by one; in groups, the rows of a
// group travel together.
template <int GN, int SQ, class F>
__device__ __forceinline__ void for_used(const unsigned long long (&used)[SQ], F&& f) {
#pragma unroll
    for (int qi = 0; qi < SQ; ++qi) {
        unsigned long long m = used[qi];
        while (m) {
            int pos[GN];
            int n = 0;
#pragma unroll
            for (int u = 0; u < GN; ++u) {
                if (m) {
                    pos[u] = __builtin_ctzll(m);
                    m &= m - 1;
                    n = u + 1;
                } else {
                    pos[u] = pos[0];
                }
            }
            f(qi, pos, n);
        }
    }
}

// Q: atoms per lane (r <= 64 Q); SQ: slots per lane (n_cap <= 64 SQ)
template <int Q, int SQ, bool P_LDS>
__global__ __launch_bounds__(256) void lars_pos_kernel(const LarsArgs a) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpg = blockDim.x >> 6;
    const int r = a.r, n_cap = a.n_cap;
    double* lsv = lds_raw + (size_t)wave * a.wave_doubles;      // direction by slot (n_cap doubles)
    Inv<P_LDS> P;
    if (P_LDS) {
        P.p = lsv + ((n_cap + 1) & ~1);
        P.ld = 0;
    } else {
        P.p = a.Pws + (size_t)(blockIdx.x * wpg + wave) * a.p_stride;
        P.ld = 64 * SQ;
    }
    const double* __restrict__ G = a.G;
    const double tiny32 = (double)FLT_MIN, eq_tol = (double)FLT_EPSILON;
    const double alpha_min = a.alpha / a.n_samples;
    const int n_cols = a.cols ? *a.n_cols : a.b;
    int peak_slots = 0;
    int over16 = 0, over32 = 0, over48 = 0, over64 = 0;     // columns of this wave that needed more slots than that

    for (int ci = blockIdx.x * wpg + wave; ci < n_cols; ci += gridDim.x * wpg) {
        const int col = a.cols ? a.cols[ci] : ci;
        // atom space (atom j = lane + 64 q)
        double xy[Q], cov[Q], coef[Q], prev[Q];
        int slot[Q];
        // slot space (slot k = lane + 64 q)
        int act[SQ];
        double w[SQ];
        unsigned long long used[SQ];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int j = lane + 64 * q;
            xy[q] = j < r ? a.Cd[(size_t)j * a.ldc + col] : 0.0;
            cov[q] = xy[q];
            coef[q] = prev[q] = 0.0;
            slot[q] = -1;
        }
#pragma unroll
        for (int q = 0; q < SQ; ++q) {
            act[q] = 0;
            w[q] = 0.0;
            used[q] = 0ull;
        }
        int n_active = 0, n_iter = 0, n_drops = 0, end = 0;
        // every trip of the loop below is an event or a skipped degenerate atom (at most r of those between
        // two events in the library; the bound makes sure that every wave leaves the loop whatever the data)
        int trips_left = a.max_iter + 2 * r + 8;
        bool drop = false, overflow = false;
        int col_peak = 0;
        double alpha = 0.0, prev_alpha = 0.0;

        auto slot_used = [&](int q) -> bool { return (used[q] >> lane) & 1ull; };
        // Loads of the passes below are unconditional, from an address that is always inside the wave's own
        // storage (the lane's slot, or the last slot for lanes beyond them), and what an unused lane read is
        // replaced by zero afterwards: a select per value instead of a masked branch per load.
        int kc[SQ], jc[Q];
#pragma unroll
        for (int q = 0; q < SQ; ++q) kc[q] = min(lane + 64 * q, n_cap - 1);
#pragma unroll
        for (int q = 0; q < Q; ++q) jc[q] = min(lane + 64 * q, r - 1);
        // w = inverse . 1 (the column sums of the inverse) is carried along: an atom that enters or leaves
        // changes it by a multiple of the vector that changes the inverse

        for (;;) {
            // largest correlation among the inactive atoms (_least_angle.py:634-647)
            double C = -DBL_MAX;
            int c_idx = INT_MAX;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int j = lane + 64 * q;
                if (j < r && slot[q] < 0 && cov[q] > C) {
                    C = cov[q];
                    c_idx = j;
                }
            }
            wave_argmax(C, c_idx);
            c_idx = uni(c_idx);
            C = uni(C);
            if (n_active >= r) C = 0.0;
            alpha = C / a.n_samples;
            if (alpha <= alpha_min + eq_tol) {                       // :656-667
                if (fabs(alpha - alpha_min) > eq_tol && n_iter > 0) {
                    const double ss = (prev_alpha - alpha_min) / (prev_alpha - alpha);
#pragma unroll
                    for (int q = 0; q < Q; ++q) coef[q] = prev[q] + ss * (coef[q] - prev[q]);
                }
                end = 1;
                break;
            }
            if (n_iter >= a.max_iter || n_active >= r) {             // :669-670
                end = 2;
                break;
            }
            if (--trips_left < 0) {
                end = 4;
                break;
            }
            if (!drop) {                                             // :671-741: atom c_idx enters
                const int v = c_idx;
                double g[SQ], u[SQ];
#pragma unroll
                for (int q = 0; q < SQ; ++q) {
                    g[q] = slot_used(q) ? G[(size_t)v * r + act[q]] : 0.0;
                    u[q] = 0.0;
                }
                const double gvv = G[(size_t)v * r + v];
                // u = inverse . g
                for_used<4>(used, [&](int qi, const int (&pos)[4], int n) {
                    double pv[4][SQ], gi[4];
#pragma unroll
                    for (int u4 = 0; u4 < 4; ++u4) {
                        gi[u4] = u4 < n ? lane_value(g[qi], pos[u4]) : 0.0;
#pragma unroll
                        for (int q = 0; q < SQ; ++q) pv[u4][q] = P.at(64 * qi + pos[u4], kc[q]);
                    }
#pragma unroll
                    for (int u4 = 0; u4 < 4; ++u4)
#pragma unroll
                        for (int q = 0; q < SQ; ++q) pv[u4][q] = slot_used(q) ? pv[u4][q] : 0.0;
#pragma unroll
                    for (int u4 = 0; u4 < 4; ++u4)
#pragma unroll
                        for (int q = 0; q < SQ; ++q) u[q] += pv[u4][q] * gi[u4];
                });
                double gu = 0.0;
#pragma unroll
                for (int q = 0; q < SQ; ++q) gu += g[q] * u[q];
                gu = uni(wave_sum(gu));
                const double s = gvv - gu;                           // Schur complement = pivot^2 (:715-717)
                const double diag = fmax(sqrt(fabs(s)), DBL_EPSILON);
                if (diag < 1e-7) {                                   // :719-738
#pragma unroll
                    for (int q = 0; q < Q; ++q)
                        if (lane + 64 * q == v) cov[q] = 0.0;
                    continue;
                }
                const double sinv = 1.0 / (diag * diag);
                int knew = 64 * SQ;
#pragma unroll
                for (int q = SQ - 1; q >= 0; --q)
                    if (~used[q]) knew = 64 * q + __builtin_ctzll(~used[q]);
                if (knew >= n_cap) {                                 // no room: the second launch codes this column
                    overflow = true;
                    break;
                }
                col_peak = max(col_peak, knew + 1);
                // inverse += u u^T / s over the used slots; the new slot's row and column
                for_used<4>(used, [&](int qi, const int (&pos)[4], int n) {
                    double pv[4][SQ], ui[4];
#pragma unroll
                    for (int u4 = 0; u4 < 4; ++u4) {
                        ui[u4] = u4 < n ? lane_value(u[qi], pos[u4]) : 0.0;
#pragma unroll
                        for (int q = 0; q < SQ; ++q) pv[u4][q] = P.at(64 * qi + pos[u4], kc[q]);
                    }
#pragma unroll
                    for (int u4 = 0; u4 < 4; ++u4)
#pragma unroll
                        for (int q = 0; q < SQ; ++q) {
                            const int i = 64 * qi + pos[u4], k = lane + 64 * q;
                            if (u4 < n && slot_used(q) && P.mine(i, k)) P.at(i, k) = pv[u4][q] + (ui[u4] * u[q]) * sinv;
                        }
                });
#pragma unroll
                for (int q = 0; q < SQ; ++q) {
                    const int k = lane + 64 * q;
                    if (slot_used(q)) {
                        P.at(knew, k) = -u[q] * sinv;
                        if (!P_LDS) P.at(k, knew) = -u[q] * sinv;
                    }
                    if (k == knew) {
                        P.at(knew, knew) = sinv;
                        act[q] = v;
                    }
                }
#pragma unroll
                for (int q = 0; q < Q; ++q)
                    if (lane + 64 * q == v) slot[q] = knew;
#pragma unroll
                for (int q = 0; q < SQ; ++q)
                    if ((knew >> 6) == q) used[q] |= 1ull << (knew & 63);
                ++n_active;
                pass_fence();
                // inverse . 1: old slots gain u (sum(u) - 1) / s, the new slot holds (1 - sum(u)) / s
                double su = 0.0;
#pragma unroll
                for (int q = 0; q < SQ; ++q) su += u[q];
                su = uni(wave_sum(su));
                const double f = (su - 1.0) * sinv;
#pragma unroll
                for (int q = 0; q < SQ; ++q) {
                    w[q] += u[q] * f;
                    if (lane + 64 * q == knew) w[q] = (1.0 - su) * sinv;
                }
            }
            if (n_iter > 0 && prev_alpha < alpha) {                  // :748-761
                end = 3;
                break;
            }
            // equiangular direction (:764-788): least_squares = inverse . 1, scaled by AA
            double sw = 0.0;
#pragma unroll
            for (int q = 0; q < SQ; ++q) sw += slot_used(q) ? w[q] : 0.0;
            sw = uni(wave_sum(sw));
            double AA = 1.0 / sqrt(sw);
            if (!(fabs(AA) <= DBL_MAX)) AA = 1.0 / sqrt(fmax(sw, DBL_EPSILON));
            double ls[SQ];
#pragma unroll
            for (int q = 0; q < SQ; ++q) {
                ls[q] = slot_used(q) ? w[q] * AA : 0.0;
                if (slot_used(q)) lsv[lane + 64 * q] = ls[q];
            }
            pass_fence();
            // correlation of every atom with the direction (:800); lanes beyond r compute a value nobody reads
            double corr[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) corr[q] = 0.0;
            for_used<8>(used, [&](int qi, const int (&pos)[8], int n) {
                double gv[8][Q], li[8];
#pragma unroll
                for (int u8 = 0; u8 < 8; ++u8) {
                    const int av = __builtin_amdgcn_readlane(act[qi], pos[u8]);
                    li[u8] = u8 < n ? lane_value(ls[qi], pos[u8]) : 0.0;
#pragma unroll
                    for (int q = 0; q < Q; ++q) gv[u8][q] = G[(size_t)av * r + jc[q]];
                }
#pragma unroll
                for (int u8 = 0; u8 < 8; ++u8)
#pragma unroll
                    for (int q = 0; q < Q; ++q) corr[q] += gv[u8][q] * li[u8];
            });
            // step to the next atom that ties with the active ones, or to zero correlation (:806-808)
            double t1 = DBL_MAX;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int j = lane + 64 * q;
                if (j < r && slot[q] < 0) {
                    const double t = (C - cov[q]) / (AA - corr[q] + tiny32);
                    if (t > 0.0 && t < t1) t1 = t;
                }
            }
            double gamma = fmin(uni(wave_min(t1)), C / AA);
            // ... or to the first active coefficient that reaches zero (:815-826)
            double z[Q], lsj[Q], zmin = DBL_MAX;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                z[q] = 0.0;
                lsj[q] = 0.0;
                if (slot[q] >= 0) {
                    lsj[q] = lsv[slot[q]];
                    z[q] = -coef[q] / (lsj[q] + tiny32);
                    if (z[q] > 0.0 && z[q] < zmin) zmin = z[q];
                }
            }
            const double z_pos = uni(wave_min(zmin));
            drop = false;
            if (z_pos < gamma) {
                gamma = z_pos;
                drop = true;
            }
            ++n_iter;
            prev_alpha = alpha;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                prev[q] = coef[q];
                coef[q] = slot[q] >= 0 ? prev[q] + gamma * lsj[q] : 0.0;      // :847
                if (slot[q] < 0) cov[q] -= gamma * corr[q];                   // :850
            }
            if (drop) {                                                        // :853-893
                for (;;) {
                    // lowest atom among those that reached zero in this event
                    int cand = INT_MAX;
#pragma unroll
                    for (int q = Q - 1; q >= 0; --q)
                        if (slot[q] >= 0 && z[q] == z_pos) cand = lane + 64 * q;
                    const int v = uni(wave_min(cand));
                    if (v == INT_MAX) break;
                    int k = 0;
#pragma unroll
                    for (int q = 0; q < Q; ++q)
                        if ((v >> 6) == q) k = __builtin_amdgcn_readlane(slot[q], v & 63);
                    double p[SQ];
#pragma unroll
                    for (int q = 0; q < SQ; ++q) {
                        p[q] = P.at(k, kc[q]);
                        p[q] = slot_used(q) ? p[q] : 0.0;
                    }
                    double wk = 0.0;
#pragma unroll
                    for (int q = 0; q < SQ; ++q)
                        if ((k >> 6) == q) wk = lane_value(w[q], k & 63);
                    double pkk = 0.0;
#pragma unroll
                    for (int q = 0; q < SQ; ++q)
                        if ((k >> 6) == q) pkk = lane_value(p[q], k & 63);
                    const double pinv = 1.0 / pkk;
#pragma unroll
                    for (int q = 0; q < SQ; ++q)
                        if ((k >> 6) == q) used[q] &= ~(1ull << (k & 63));
                    pass_fence();
                    // inverse of the remaining atoms: inverse -= p p^T / p_kk
                    for_used<4>(used, [&](int qi, const int (&pos)[4], int n) {
                        double pv[4][SQ], pi[4];
#pragma unroll
                        for (int u4 = 0; u4 < 4; ++u4) {
                            pi[u4] = u4 < n ? lane_value(p[qi], pos[u4]) : 0.0;
#pragma unroll
                            for (int q = 0; q < SQ; ++q) pv[u4][q] = P.at(64 * qi + pos[u4], kc[q]);
                        }
#pragma unroll
                        for (int u4 = 0; u4 < 4; ++u4)
#pragma unroll
                            for (int q = 0; q < SQ; ++q) {
                                const int i = 64 * qi + pos[u4], kk = lane + 64 * q;
                                if (u4 < n && slot_used(q) && P.mine(i, kk))
                                    P.at(i, kk) = pv[u4][q] - (pi[u4] * p[q]) * pinv;
                            }
                    });
                    // inverse . 1 without the atom: w - p w_k / p_kk (zero in its own slot)
#pragma unroll
                    for (int q = 0; q < SQ; ++q) w[q] = slot_used(q) ? w[q] - p[q] * (wk * pinv) : 0.0;
                    // its correlation, recomputed from the data (:889); coefficients of lanes beyond r are zero
                    double dot = 0.0;
#pragma unroll
                    for (int q = 0; q < Q; ++q) dot += G[(size_t)v * r + jc[q]] * coef[q];
                    dot = uni(wave_sum(dot));
#pragma unroll
                    for (int q = 0; q < Q; ++q)
                        if (lane + 64 * q == v) {
                            cov[q] = xy[q] - dot;
                            slot[q] = -1;
                        }
                    --n_active;
                    ++n_drops;
                    pass_fence();
                }
            }
        }
        if (overflow) {
            if (lane == 0) {
                const int at = __hip_atomic_fetch_add(a.meta, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                a.overflow[at] = col;
            }
        } else {
            // (a column that ran out of room is counted where it is coded: in the second launch)
            peak_slots = max(peak_slots, col_peak);
            over16 += col_peak > 16;
            over32 += col_peak > 32;
            over48 += col_peak > 48;
            over64 += col_peak > 64;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int j = lane + 64 * q;
                if (j < r) a.H[(size_t)j * a.ldh + col] = (float)coef[q];
            }
            if (a.info && lane == 0) {
                int* rec = a.info + 4 * (size_t)col;
                rec[0] = n_iter;
                rec[1] = end;
                rec[2] = n_active;
                rec[3] = n_drops;
            }
        }
        pass_fence();
    }
    if (lane == 0 && peak_slots > 0) {
        __hip_atomic_fetch_max(a.meta + 1, peak_slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (over16) __hip_atomic_fetch_add(a.meta + 2, over16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (over32) __hip_atomic_fetch_add(a.meta + 3, over32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (over48) __hip_atomic_fetch_add(a.meta + 4, over48, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (over64) __hip_atomic_fetch_add(a.meta + 5, over64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// behind the launches of a call: the most slots a column used, and how many columns needed more than 16, 32, 48
// and 64, go to host memory the device can write (the next call sizes its first launch by them), and the
// device words are cleared for that call
__global__ void lars_publish_kernel(int* meta, int* host_words, int r, int b) {
    host_words[0] = meta[1];
    host_words[1] = r;
    host_words[2] += meta[0];          // columns coded by a second launch since the context was created
#pragma unroll
    for (int i = 0; i < 4; ++i) host_words[3 + i] = meta[2 + i];
    host_words[7] = b;
#pragma unroll
    for (int i = 0; i < 6; ++i) meta[i] = 0;
}

constexpr int LARS_HOST_WORDS = ONMF_LARS_HOST_WORDS;      // first of the words above in onmf_ctx::host_events

template <int Q, int SQ, bool P_LDS>
int launch_lars_form(onmf_ctx* c, LarsArgs a, int n_cap, bool second, hipStream_t s) {
    static LdsLimit lds_limit;
    const int wpg = 4;
    a.n_cap = n_cap;
    const size_t ls_doubles = (size_t)((n_cap + 1) & ~1);
    auto kern = lars_pos_kernel<Q, SQ, P_LDS>;
    int grid;
    size_t lds;
    if (P_LDS) {
        const size_t packed = ((size_t)n_cap * (n_cap + 1) / 2 + 1) & ~(size_t)1;
        a.wave_doubles = (int)(ls_doubles + packed);
        int waves = wpg;
        lds = (size_t)a.wave_doubles * sizeof(double) * waves;
        while (lds > 160 * 1024 && waves > 1) lds = (size_t)a.wave_doubles * sizeof(double) * --waves;
        if (lds > 160 * 1024) return onmf_fail(c, ONMF_ERR_ARG, "LARS coder: %d slots do not fit in LDS", n_cap);
        if (lds > 48 * 1024) ONMF_TRY(onmf_raise_lds_limit(c, lds_limit, (const void*)kern, lds));
        grid = (a.b + waves - 1) / waves;
        if (second && grid > 1024) grid = 1024;       // few columns overflow; the kernel strides over its list
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * waves), lds, s, a);
    } else {
        // the inverse in global memory: a bounded number of waves, each with its own square
        const size_t stride = (size_t)(64 * SQ) * (64 * SQ);
        const int waves_total = a.b < 2048 ? a.b : 2048;
        grid = (waves_total + wpg - 1) / wpg;
        ONMF_TRY(onmf_reserve(c, c->lars_ws, (size_t)grid * wpg * stride * sizeof(double), "LARS workspace"));
        a.Pws = (double*)c->lars_ws.p;
        a.p_stride = stride;
        a.wave_doubles = (int)ls_doubles;
        lds = ls_doubles * sizeof(double) * wpg;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wpg), lds, s, a);
    }
    if (hipGetLastError() != hipSuccess) return onmf_fail(c, ONMF_ERR_HIP, "LARS coder: launch failed");
    return ONMF_OK;
}

// packed inverse of n slots + the direction vector, per wave
size_t lars_lds_bytes(int n) { return ((((size_t)n + 1) & ~(size_t)1) + ((((size_t)n * (n + 1) / 2) + 1) & ~(size_t)1)) * 8; }

template <int Q>
int launch_lars_q(onmf_ctx* c, const LarsArgs& a0, hipStream_t s) {
    LarsArgs a = a0;
    const int r = a.r;
    // state of the call: two device words, and room for the indices of the columns that overflow
    const bool fresh = c->lars_meta.p == nullptr;
    ONMF_TRY(onmf_reserve(c, c->lars_meta, 16 * sizeof(int), "LARS state"));
    if (fresh) HIP_TRY(c, hipMemsetAsync(c->lars_meta.p, 0, 16 * sizeof(int), s));
    ONMF_TRY(onmf_reserve(c, c->lars_overflow, (size_t)a.b * sizeof(int), "LARS overflow list"));
    a.meta = (int*)c->lars_meta.p;
    a.overflow = (int*)c->lars_overflow.p;
    // How many columns of the previous call (same r) needed more than 16, 32, 48, 64 slots: the first launch
    // gets the smallest of these capacities that all but a few columns (1 in 64) would have fitted in - small
    // inverses let several waves per SIMD be resident - and the columns that need more go to the second launch,
    // which has room for all r atoms.  A first call, with nothing to go by, tries 64.
    volatile int* hw = c->host_events + LARS_HOST_WORDS;
    const bool have = hw[1] == r && c->lars_hint_ok && hw[7] > 0;
    int n_cap = r;
    if (c->lars_slots > 0) {
        n_cap = c->lars_slots < r ? c->lars_slots : r;
        if (n_cap > 64 && n_cap < r) n_cap = 64;
    } else if (have) {
        const int few = hw[7] / 64;
        const int caps[4] = {16, 32, 48, 64};
        for (int i = 0; i < 4; ++i)
            if (hw[3 + i] <= few && caps[i] < r) {
                n_cap = caps[i];
                break;
            }
    } else if (r > 64) {
        n_cap = 64;
    }
    const bool full_fits = lars_lds_bytes(r) <= 160 * 1024;
    if (n_cap < r) {
        ONMF_TRY((launch_lars_form<Q, 1, true>(c, a, n_cap, false, s)));
        LarsArgs a2 = a;
        a2.cols = a.overflow;
        a2.n_cols = a.meta;
        if (full_fits) ONMF_TRY((launch_lars_form<Q, Q, true>(c, a2, r, true, s)));
        else ONMF_TRY((launch_lars_form<Q, Q, false>(c, a2, r, true, s)));
    } else if (full_fits) {
        ONMF_TRY((launch_lars_form<Q, Q, true>(c, a, r, false, s)));
    } else {
        ONMF_TRY((launch_lars_form<Q, Q, false>(c, a, r, false, s)));
    }
    hipLaunchKernelGGL(lars_publish_kernel, dim3(1), dim3(1), 0, s, a.meta, c->host_events_dev + LARS_HOST_WORDS, r, a.b);
    if (hipGetLastError() != hipSuccess) return onmf_fail(c, ONMF_ERR_HIP, "LARS coder: launch failed");
    c->lars_hint_ok = true;
    return ONMF_OK;
}

}  // namespace

// Out1 = A^T B1 (M x N1) and, when B2 is given, Out2 = A^T B2 (M x N2) in the same launch (A [k][m], B [k][n])
int launch_prod_f64_pair(onmf_ctx* c, int M, int K, const float* A, int lda, const float* B1, int ldb1, int N1,
                         double* Out1, int ldo1, const float* B2, int ldb2, int N2, double* Out2, int ldo2,
                         hipStream_t s) {
    const int tiles_m = (M + PT - 1) / PT;
    if (!B2) N2 = 0;
    if ((long long)tiles_m * ((N1 + 63) / 64 + (N2 + 63) / 64) >= 1024) {   // several workgroups per CU as it is
        const int t1 = (N1 + 63) / 64;
        dim3 grid(t1 + (N2 + 63) / 64, tiles_m);
        hipLaunchKernelGGL(prod_f64_kernel<4>, grid, dim3(256), 0, s, M, N1, K, A, lda, B1, ldb1, Out1, ldo1, t1, B2, ldb2,
                           N2, Out2, ldo2);
    } else {
        const int t1 = (N1 + 31) / 32;
        dim3 grid(t1 + (N2 + 31) / 32, tiles_m);
        hipLaunchKernelGGL(prod_f64_kernel<2>, grid, dim3(256), 0, s, M, N1, K, A, lda, B1, ldb1, Out1, ldo1, t1, B2, ldb2,
                           N2, Out2, ldo2);
    }
    if (hipGetLastError() != hipSuccess) return onmf_fail(c, ONMF_ERR_HIP, "double-precision product: launch failed");
    return ONMF_OK;
}

int launch_prod_f64(onmf_ctx* c, int M, int N, int K, const float* A, int lda, const float* B, int ldb, double* Out,
                    int ldo, hipStream_t s) {
    return launch_prod_f64_pair(c, M, K, A, lda, B, ldb, N, Out, ldo, nullptr, 0, 0, nullptr, 0, s);
}

int launch_lars_pos(onmf_ctx* c, int r, int b, const double* G, const double* Cd, int ldc, double alpha,
                    double n_samples, int max_iter, float* H, int ldh, int* info, hipStream_t s) {
    LarsArgs a{};
    a.r = r;
    a.b = b;
    a.G = G;
    a.Cd = Cd;
    a.ldc = ldc;
    a.alpha = alpha;
    a.n_samples = n_samples;
    a.max_iter = max_iter;
    a.H = H;
    a.ldh = ldh;
    a.info = info;
    if (r <= 64) return launch_lars_q<1>(c, a, s);
    if (r <= 128) return launch_lars_q<2>(c, a, s);
    if (r <= 192) return launch_lars_q<3>(c, a, s);
    if (r <= 256) return launch_lars_q<4>(c, a, s);
    return onmf_fail(c, ONMF_ERR_ARG, "LARS coder: r = %d > 256 is not supported", r);
}
