// Blocked Gauss-Seidel code sweep on the gfx950 matrix cores (src/onmf.py:252-263 of the reference).
// Included by sweep_ct1.hip / sweep_ct2.hip, which instantiate it for one or two column tiles per wave.
#pragma once
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Tile kernel: blocked Gauss-Seidel on the matrix cores.
//
// A wave owns CT tiles of 16 columns of H; a workgroup of 16/CT waves owns 256 columns.  Rows are
// cut into NB blocks of 16.  For block K the products
//     R = Gm[K, :] . H          (16 x r times r x 16, v_mfma_f32_16x16x4_f32, exact fp32)
// use the current H (blocks < K already updated this sweep), then the 16 rows of the block are
// updated in order on the VALU with the in-block corrections  Gm[k, j] * (h_j_new - h_j_old), j < k.
//
// Register layout = the MFMA's own: lane l holds column (l & 15); lane group g = l >> 4 holds rows
// 16J + 4g + t (t = 0..3) of every block J, both for H (the B operand, k index = g) and for the
// accumulator R (row 4g + t).  H and Cm stay in registers for the whole sweep.
//
// LDS holds Gm as 16x16 blocks (1 KB each), XOR-swizzled so that the A-operand read of block (K,J)
// - one ds_read_b128 per lane: row (l & 15), columns 4g..4g+3 - is conflict-free.  When all NB*NB
// blocks do not fit (r = 256), only blocks J <= K are stored and block (K,J), J > K, is read
// transposed from block (J,K) (Gm is symmetric).  A second set of NB blocks holds the strictly
// lower triangles of the diagonal blocks for the in-block corrections.
//
// In-block update, per quad Q of 4 consecutive rows (they live in lane group Q): the owner group
// runs its 4 rows in order; the 4 deltas are then broadcast to the other groups (ds_bpermute) and
// pushed into the accumulators of the rows that come later in the block.
// ---------------------------------------------------------------------------------------------
//
// Every A-operand read (one Gm block slice per lane) feeds the MFMAs of all CT tiles of the wave.
// With CT = 1 the 16 waves of a workgroup each read all of Gm from the LDS once per sweep and the
// LDS port, not the matrix pipe, bounds the kernel; CT = 2 halves that traffic (8 waves, twice
// the registers per lane).
constexpr int TILE_COLS_PER_WG = 256;

__device__ __forceinline__ int swz(int row) { return (-(row >> 2)) & 3; }   // column-quad XOR per row

// dword offset of element (i, j) inside one swizzled 16x16 block
__device__ __forceinline__ int blk_off(int i, int j) { return i * 16 + 4 * ((j >> 2) ^ swz(i)) + (j & 3); }

template <int NB, bool HALF>
__device__ __forceinline__ int blk_index(int K, int J) {
    return HALF ? (K * (K + 1) / 2 + J) : (K * NB + J);
}

// 4x4 exchange of a register index against the lane group (16-lane row) of a wave: afterwards register j
// of lane group g holds what register g of lane group j held (v_permlane32_swap exchanges rows 2,3 of its
// first operand with rows 0,1 of the second; v_permlane16_swap the odd rows of the first with the even
// rows of the second).
__device__ __forceinline__ void xpose_reg_group(float& r0, float& r1, float& r2, float& r3) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(r0), __float_as_uint(r2), false, false);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1), __float_as_uint(r3), false, false);
    auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
    auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
    r0 = __uint_as_float(c[0]);
    r1 = __uint_as_float(c[1]);
    r2 = __uint_as_float(d[0]);
    r3 = __uint_as_float(d[1]);
}

// sum over the 16 lanes of a DPP row (lanes 16g..16g+15); the total lands in lane 16g+15
__device__ __forceinline__ float row_sum16(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, true));
    return x;
}

// PROBE = true adds what the stopping test needs without a second pass over H (see screen.hip):
// P = U^T (H_new - H_old) for 16 unit probe vectors U (4 extra MFMAs per block) and, per probe,
// sum_c P[p][c]^2 -> a rigorous lower bound of sigma_max(H_new - H_old)^2; together with the two
// Frobenius sums that bounds the test value from both sides.
//
// One launch runs sweeps k_first .. k_first + k_count - 1.  Columns are independent, so between two
// sweeps a workgroup needs nothing from the others but the verdict of the stopping test: H stays in
// registers, Gm in LDS, and nothing is staged or launched again.  Each sweep's sums go to that
// sweep's own accumulator set (agent-scope atomics, waited for) followed by a count of workgroups
// done.  Before sweep k a workgroup evaluates the test of sweep k - 2 (its count is normally long
// complete: the others are at most one sweep behind), so at a "stop" one sweep has run ahead in
// vain - it wrote the third of three rotating H buffers, and both buffers of the stopping sweep are
// intact.  The waits are bounded; a grid that is not resident as a whole would make them fail, so
// the host launches k_count > 1 only for grids that fit the chip.
//
// EACH = true is the solve of many independent one-column problems in one launch (the reference
// codes an image patch by patch, image_reconstruction.py:384, each call with its own stopping
// test): a column takes part in sweep k only while its own  ||h_new - h_old|| / ||h_old||  of the
// sweep before exceeds es.tol - for one column the spectral norms of src/onmf.py:265 are vector
// norms.  A column that has stopped runs with step 0 (exact no-op); a wave whose 16 columns have all
// stopped skips the arithmetic.  Nothing is exchanged between workgroups, H is loaded once and stored
// once (in place is fine), and es.sweeps receives the number of sweeps each column ran.
//
// WPG = waves per workgroup (16 / CT by default: 256 columns per workgroup).  The per-column mode also
// comes with 4 waves (64 columns) per workgroup: a batch of a few thousand columns then spreads over
// that many more CUs with one wave per SIMD each, instead of four waves sharing every SIMD's matrix
// pipe on a few CUs while the rest of the chip idles.
// The probing kernel with one tile per wave and four tile waves per workgroup (batches of a few thousand columns:
// sweeps of a few microseconds) comes with a FIFTH wave that owns no tile: the control wave.  What a sweep's
// synchronisation costs - the workgroup's atomics out and back, the look at the other workgroups' count, the
// round trip for their sums, the test's arithmetic: 3 us a sweep, as much as the sweep's own work - it does
// while the tile waves are in the NEXT sweep's block loop: during sweep k it posts the workgroup's sums of
// sweep k - 1 (double-buffered in LDS), waits for every workgroup's, evaluates that sweep's test and leaves
// the verdict for the top of sweep k + 1.  Same verdicts at the same sweeps as the other forms (a test is
// acted on two sweeps later), one barrier less per sweep, and nothing of it on the tile waves' path.
template <int NB, bool HALF, bool PROBE, int CT, bool EACH, int WPG>
constexpr bool sweep_has_control_wave() { return PROBE && !EACH && CT == 1 && WPG == 4; }

template <int NB, bool HALF, bool PROBE, int CT, bool EACH = false, int WPG = 16 / CT>
__global__ __launch_bounds__(64 * WPG + (sweep_has_control_wave<NB, HALF, PROBE, CT, EACH, WPG>() ? 64 : 0)) void sweep_tile_kernel(
    int r, int b, const float* __restrict__ Gm, const float* __restrict__ Cm, float* hb0, float* hb1,
    float* hb2, float alpha, int exact_cd, int k_first_arg, int k_count_arg, double* __restrict__ stats,
    ScreenDev sc, EachStop es) {
    static_assert(!EACH || (!PROBE && CT == 1), "per-column stopping: plain kernel, one tile per wave");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NBLK = HALF ? NB * (NB + 1) / 2 : NB * NB;
    constexpr int NT = 64 * WPG;           // threads per workgroup that own tiles
    constexpr bool CTRL = sweep_has_control_wave<NB, HALF, PROBE, CT, EACH, WPG>();
    // Two tiles per wave: the in-block update runs on both at once.  A v_permlane16_swap of the two tiles'
    // registers leaves lane group g with 8 consecutive rows (8 (g >> 1) .. + 7) of ONE column of tile g & 1;
    // the rows are then updated in order inside the lane, first half of the block in the lower half of the
    // wave, then the second half in the upper one - every instruction serves both tiles and nothing crosses
    // lanes inside a half; between the halves the first half's deltas are pushed into the second half's
    // gradients by two MFMAs per tile.  Same operations on every row in the same order as the
    // quad-by-quad form: bit-identical codes.
    constexpr bool PAIR = (CT == 2) && !EACH;
    constexpr int WAVES = WPG;
    float* Gs = lds;                       // NBLK swizzled 16x16 blocks of Gm
    // The current sweep's step sizes, negated, as four tables of 16*NB: table Q holds the entries of the
    // rows whose quad within their block is Q and zeros elsewhere (and zeros for padded rows) - what the
    // lanes read while quad Q moves: their own steps in lane group Q, step 0 in the other groups.
    float* steps = lds + NBLK * 256;
    float* gdiag = steps + 4 * 16 * NB;    // diagonal of Gm
    float* Us = gdiag + 16 * NB;           // PROBE: NB swizzled blocks of U^T
    float* p2wg = Us + NB * 256;           // PROBE: workgroup partial of the 16 probe energies
    double* sumwg = reinterpret_cast<double*>(p2wg + 16);   // PROBE: workgroup partial of the two sums
    int* verdict_s = reinterpret_cast<int*>(p2wg + 20);     // PROBE: verdict handed from wave 0 to the rest
    // Collatz-Wielandt by-product (probing kernels, r <= 128): after a sweep, y = H_new (H_new^T v) for the
    // screen's positive vector v - max_k y_k / v_k bounds sigma_max(H_new)^2, the denominator of the NEXT
    // sweep's test, far more tightly than the bound inherited from sweep to sweep.  About 10 % more vector
    // instructions per sweep for these small ranks, where borderline tests are frequent.
    constexpr bool CWB = PROBE && NB <= SCREEN_CW_ROWS / 16;
    float* ywg = p2wg + 24;                                 // CWB: 16*NB workgroup partial sums of y
    // CTRL: the workgroup's partial sums exist twice (the set of sweep k is filled while the control wave sends
    // off the set of sweep k - 1), and so does the verdict slot (written for the next sweep while this sweep's
    // may still be read)
    constexpr int SET_FLOATS = 24 + 16 * NB;
    auto p2_of = [&](int set) -> float* { return p2wg + (CTRL ? set * SET_FLOATS : 0); };
    auto sum_of = [&](int set) -> double* { return reinterpret_cast<double*>(p2_of(set) + 16); };
    auto y_of = [&](int set) -> float* { return p2_of(set) + 24; };
    auto verdict_of = [&](int slot) -> int* { return verdict_s + (CTRL ? slot : 0); };

    const int tid = threadIdx.x;
    int k_first = k_first_arg, k_count = k_count_arg, v_first = sc.v_first;
    if (PROBE) {
        // an earlier launch's test ended the loop (grid-uniform)
        if (reinterpret_cast<const int*>(sc.base)[ScreenLayout::FLAG] != 0) return;
        if (sc.start_dev != nullptr) {
            // enqueued before its first sweep was known (behind a device-side resolution of an
            // inconclusive verdict): it is in device memory
            k_first = *sc.start_dev;
            k_count = sc.k_end - k_first;
            v_first = k_first;
            if (k_count <= 0) return;
        }
    }

    const int lane = tid & 63;
    const int wave = tid >> 6;
    if constexpr (CTRL) {
        if (wave == WPG) {
            // ---- the control wave ----
            float ubc = sc.base[ScreenLayout::UBH + v_first];
            const int cw = CWB ? 16 * NB : 0;
            auto wait_count = [&](int j) -> bool {         // every workgroup's sums of sweep j in place? (bounded)
                const int* cnt = reinterpret_cast<const int*>(sc.base) + ScreenLayout::CNT + j;
                const long long t0 = wall_clock64();
                for (;;) {
                    // (a limit of SPIN_FORCE_TIMEOUT makes the wait fail even when the count is complete: the
                    // host's way to exercise the recovery path)
                    if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (int)gridDim.x &&
                        sc.spin_limit != SPIN_FORCE_TIMEOUT)
                        return true;
                    if (wall_clock64() - t0 > sc.spin_limit) return false;          // wave-uniform (one clock value)
                    __builtin_amdgcn_s_sleep(4);
                }
            };
            auto test_of = [&](int j) -> int {             // verdict of sweep j (0 = go on)
                if (!wait_count(j)) {
                    if (lane == 0)
                        __hip_atomic_store(reinterpret_cast<int*>(sc.base) + ScreenLayout::FLAG, SCREEN_FLAG_TIMEOUT,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return SCREEN_FLAG_TIMEOUT;
                }
                return screen_eval_wave(sc.base, j, ubc, sc.tol, sc.rank_cap, blockIdx.x == 0, -1.0f, cw);
            };
            // this workgroup's sums of sweep j go out (post_issue), then its count (post_finish)
            auto post_issue = [&](int j, float& back_f, double& back_d) {
                float* cur = sc.base + ScreenLayout::SETS + j * ScreenLayout::SETSZ;
                const int rep = blockIdx.x % SCREEN_REPL;
                back_f = 0.f;
                back_d = 0.0;
                if (lane < 16) {
                    back_f = atomicAdd(&cur[ScreenLayout::SET_P2 + rep * 16 + lane], p2_of(j & 1)[lane]);
                } else if (lane < 18) {
                    back_d = atomicAdd(reinterpret_cast<double*>(cur + ScreenLayout::SET_SUMS) + rep * 2 + (lane - 16),
                                       sum_of(j & 1)[lane - 16]);
                }
                if (CWB) {
                    const int yrep = blockIdx.x % SCREEN_YREPL;
                    for (int q = lane; q < 16 * NB; q += 64)
                        back_d += atomicAdd(reinterpret_cast<double*>(cur + ScreenLayout::SET_Y) + yrep * SCREEN_CW_ROWS + q,
                                            (double)y_of(j & 1)[q]);
                }
            };
            auto post_finish = [&](int j, float back_f, double back_d) {
                // (the atomics return, and the count goes out only after every one of them has: whoever sees the
                // count complete finds all sums in place, without any cache maintenance)
                asm volatile("s_waitcnt vmcnt(0)" ::"v"(back_f), "v"(back_d) : "memory");
                if (lane == 0)
                    __hip_atomic_fetch_add(reinterpret_cast<int*>(sc.base) + ScreenLayout::CNT + j, 1, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            };
            auto post_sums = [&](int j) {
                float back_f;
                double back_d;
                post_issue(j, back_f, back_d);
                post_finish(j, back_f, back_d);
            };
            // the verdict the first sweep of the launch starts with (its sums were all posted before the launch)
            // sc.lag = 2: the test of sweep k - 1 is evaluated during sweep k, behind this workgroup's own sums
            // of that sweep - it waits for the other workgroups' (the tile waves wait for it when their sweep
            // is shorter than that chain).  sc.lag = 3 (the host's choice when it can run the solve again): the
            // test of sweep k - 2, whose sums all went out a sweep ago - nothing to wait for, and the tile
            // waves never wait.  A test is then acted on three sweeps later: two sweeps run ahead in vain at a
            // stop, and the second of them overwrites the stopping sweep's input - which only an inconclusive
            // test needs (the host's full evaluation); the host then runs that solve again.
            const int lag = sc.lag == 3 ? 3 : 2;
            int verdict = 0;
            if (k_first - lag >= v_first) verdict = test_of(k_first - lag);    // (its sums went out in an earlier launch)
            if (lane == 0) *verdict_of(k_first & 1) = verdict;
            __syncthreads();
            if (verdict != 0) return;
            for (int k = k_first; k < k_first + k_count; ++k) {
                __syncthreads();                           // the tile waves enter the block loop of sweep k
                verdict = 0;
                const int j = k + 1 - lag;
                if (lag == 3) {
                    // the tested sweep's sums do not include the ones going out now: its loads travel with
                    // these atomics (one round trip to memory instead of two in a row)
                    float back_f = 0.f;
                    double back_d = 0.0;
                    const bool posting = k - 1 >= k_first;
                    if (posting) post_issue(k - 1, back_f, back_d);
                    if (j >= v_first) verdict = test_of(j);
                    if (posting) post_finish(k - 1, back_f, back_d);
                } else {
                    if (k - 1 >= k_first) post_sums(k - 1);
                    if (j >= v_first) verdict = test_of(j);
                }
                if (lane == 0) *verdict_of((k + 1) & 1) = verdict;
                __syncthreads();                           // the tile waves' sums of sweep k are in LDS
                if (verdict != 0) return;                  // (the tile waves leave at the top of sweep k + 1)
            }
            post_sums(k_first + k_count - 1);
            return;
        }
    }
    const int c_lane = lane & 15;          // column of the tile (doubles as the A-operand row index)
    const int g_lane = lane >> 4;          // lane group: rows 4g .. 4g+3 of every block

    // H and Cm go through buffer descriptors: byte address = (per-lane voffset) + (wave-uniform
    // soffset = row offset * b), no per-access address registers, and the hardware range check
    // returns 0 for rows >= r (their offset is past r*b floats) and drops such stores.  Columns
    // >= b get an out-of-range offset: they load zeros, run the arithmetic on zeros (alpha is
    // masked for them too, so they stay zero) and their stores are dropped.
    // The host guarantees r * b * 4 < 2^31.
    const unsigned nbytes = (unsigned)r * (unsigned)b * 4u;
    const auto rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)Cm, 0, nbytes, 0x00020000);
    int voff[CT];
    float alpha_c[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = ((blockIdx.x * WAVES + wave) * CT + ct) * 16 + c_lane;
        voff[ct] = col < b ? (4 * g_lane * b + col) * 4 : (int)0x80000000;
        alpha_c[ct] = col < b ? alpha : 0.f;
    }
    float h[CT][NB][4];
    {
        float* hin = (k_first % 3 == 0) ? hb0 : (k_first % 3 == 1) ? hb1 : hb2;
        const auto rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)hin, 0, nbytes, 0x00020000);
#pragma unroll
        for (int J = 0; J < NB; ++J)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    h[ct][J][t] = __uint_as_float(
                        __builtin_amdgcn_raw_buffer_load_b32(rs_h, voff[ct], (16 * J + t) * b * 4, 0));
    }
    // (H is requested first: its loads are in flight while Gm is staged)
    // ---- stage Gm into LDS (whole workgroup): one float4 (4 consecutive columns of one row of a
    // block) per item, all of a thread's loads in flight together ----
    {
        constexpr int ITEMS = NBLK * 64;
        constexpr int PER_THREAD = (ITEMS + NT - 1) / (NT);
        const bool vec_ok = ((r & 3) == 0) && ((((uintptr_t)Gm) & 15) == 0);
        float4 stage[PER_THREAD];
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            const int item = tid + it * NT;
            const int n = min(item >> 6, NBLK - 1);
            int K, J;
            if (HALF) {
                K = (int)((sqrtf(8.0f * (float)n + 1.0f) - 1.0f) * 0.5f);
                while (K * (K + 1) / 2 > n) --K;
                while ((K + 1) * (K + 2) / 2 <= n) ++K;
                J = n - K * (K + 1) / 2;
            } else {
                K = n / NB;
                J = n - K * NB;
            }
            const int q = item & 63;
            const int row = 16 * K + (q >> 2), col = 16 * J + 4 * (q & 3);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < r) {
                const float* src = Gm + (size_t)row * r + col;
                if (vec_ok && col + 3 < r) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (col + 0 < r) v.x = src[0];
                    if (col + 1 < r) v.y = src[1];
                    if (col + 2 < r) v.z = src[2];
                    if (col + 3 < r) v.w = src[3];
                }
            }
            stage[it] = v;
        }
#pragma unroll
        for (int it = 0; it < PER_THREAD; ++it) {
            const int item = tid + it * NT;
            if (item < ITEMS) {
                const int q = item & 63, i = q >> 2;
                *reinterpret_cast<float4*>(Gs + (item >> 6) * 256 + i * 16 + 4 * ((q & 3) ^ swz(i))) = stage[it];
            }
        }
    }
    for (int j = tid; j < 16 * NB; j += NT) gdiag[j] = j < r ? Gm[(size_t)j * r + j] : 0.f;
    if (PROBE) {
        for (int e = tid; e < NB * 256; e += NT) Us[e] = sc.base[ScreenLayout::UBLK + e];
    }

    float cmn[CT][4];                      // Cm of the block that comes next (block 0 to begin with)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            cmn[ct][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_c, voff[ct], t * b * 4, 0));
    float ub = PROBE ? sc.base[ScreenLayout::UBH + v_first] : -1.0f;       // bound carried by wave 0
    float vcw[CWB ? NB : 1][4];            // CWB: the lane's rows of v (constant for the launch)
    if constexpr (CWB) {
#pragma unroll
        for (int J = 0; J < NB; ++J)
#pragma unroll
            for (int t = 0; t < 4; ++t) vcw[J][t] = sc.base[ScreenLayout::CWV + 16 * J + 4 * g_lane + t];
    }
    bool active = true;                    // EACH: this lane's column is still iterating
    int col_sweeps = 0;
    // PROBE, wave 0: the test that the next sweep starts with is evaluated at the END of this sweep, behind
    // this workgroup's own atomics and while they travel (their return is waited for anyway), if the count
    // of the tested sweep is complete by then - it nearly always is.  Otherwise at the next sweep's start.
    bool tested_ahead = false;
    int verdict_ahead = 0;

#ifdef ONMF_TRACE_WG
    // diagnostic build (tools/sweep_marks.py): workgroup 0's time marks, 16 per sweep, behind the screen state
    unsigned long long* marks = nullptr;
    if (PROBE && blockIdx.x == 0 && tid == 0)
        marks = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(sc.base) +
                                                      (((size_t)ScreenLayout::TOTAL * sizeof(float) + 63) & ~(size_t)63));
#define SWEEP_MARK(i) do { if (marks && k < SCREEN_MAX_SWEEPS) marks[16 * k + (i)] = wall_clock64(); } while (0)
#else
#define SWEEP_MARK(i) do { } while (0)
#endif
    if constexpr (CTRL) __syncthreads();   // the control wave has left the verdict the first sweep starts with
    for (int k = k_first; k < k_first + k_count; ++k) {
        SWEEP_MARK(0);
        if constexpr (CTRL) {
            // the test of sweep k - 2, evaluated by the control wave during sweep k - 1 (workgroup-uniform, and
            // the same in every workgroup)
            if (*verdict_of(k & 1) != 0) return;
        }
        if (PROBE && !CTRL) {
            // the test of sweep k - 2, once every workgroup's sums of that sweep are in
            if (wave == 0) {
                int verdict = 0;
                const int j = k - 2;
                if (j >= v_first) {
                    const int* cnt = reinterpret_cast<const int*>(sc.base) + ScreenLayout::CNT + j;
                    // (requested together with the first look at the count; 0 = none, as negative)
                    const float seeded = tested_ahead ? 0.f : fmaxf(screen_load_bound(sc.base, j + 1), 0.f);
                    const long long t0 = wall_clock64();
                    while (!tested_ahead) {
                        // (the sums are read only after the count has been seen complete: they were
                        // all in place before its last increment)
                        // (a limit of SPIN_FORCE_TIMEOUT makes the wait fail even when the count is complete:
                        // the host's way to exercise the recovery path)
                        if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (int)gridDim.x &&
                            sc.spin_limit != SPIN_FORCE_TIMEOUT)
                            break;
                        if (wall_clock64() - t0 > sc.spin_limit) {        // wave-uniform (one clock value)
                            verdict = SCREEN_FLAG_TIMEOUT;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    if (tested_ahead) {
                        verdict = verdict_ahead;
                    } else if (verdict == 0) {
                        verdict = screen_eval_wave(sc.base, j, ub, sc.tol, sc.rank_cap, blockIdx.x == 0, seeded, CWB ? 16 * NB : 0);
                    } else if (lane == 0) {
                        __hip_atomic_store(reinterpret_cast<int*>(sc.base) + ScreenLayout::FLAG, verdict,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (lane == 0) *verdict_s = verdict;
            }
        }
        // this sweep's step sizes (src/onmf.py:254: 1 / (sqrt(k + 10) * (Gm[j][j] + 1)))
        {
            const float rdamp = 1.0f / (float)sqrt((double)k + 10.0);
            for (int j = tid; j < 16 * NB; j += NT) {
                float st = 0.f;
                if (j < r) {
                    const float gjj = gdiag[j];
                    st = exact_cd ? (gjj > 0.f ? 1.0f / gjj : 0.f) : rdamp / (gjj + 1.0f);
                }
                if (PAIR) {
                    // table [phase][half of the wave][block][8]: in phase ph the lanes of half hh hold rows
                    // 8 hh .. 8 hh + 7 of the block and move only when hh == ph
                    const int Kb = j >> 4, ph = (j >> 3) & 1, i8 = j & 7;
                    steps[((ph * 2 + ph) * NB + Kb) * 8 + i8] = -st;
                    steps[((ph * 2 + (1 - ph)) * NB + Kb) * 8 + i8] = 0.f;
                } else {
#pragma unroll
                    for (int Q = 0; Q < 4; ++Q) steps[Q * 16 * NB + j] = (((j >> 2) & 3) == Q) ? -st : 0.f;
                }
            }
            if (PROBE) {
                if (tid < 16) p2_of(k & 1)[tid] = 0.f;
                if (tid < 2) sum_of(k & 1)[tid] = 0.0;
                if (CWB)
                    for (int j = tid; j < 16 * NB; j += NT) y_of(k & 1)[j] = 0.f;
            }
        }
        __syncthreads();
        SWEEP_MARK(1);
        if (PROBE && !CTRL) {
            if (*verdict_s != 0) return;   // workgroup-uniform, and the same in every workgroup
        }

        // (opaque per-sweep copies of the lane coordinates: hoisted out of this loop, the hundreds of
        // LDS addresses derived from them would each occupy a register for the whole kernel)
        if (EACH) {
            if (__ballot(active) == 0) {   // wave-uniform: all 16 columns are done
                if (k + 1 < k_first + k_count) __syncthreads();
                continue;
            }
        }
        int lane_k = threadIdx.x;
        asm volatile("" : "+v"(lane_k));
        const int c = lane_k & 15, g = (lane_k >> 4) & 3, sw_c = swz(c);
        float* hout = ((k + 1) % 3 == 0) ? hb0 : ((k + 1) % 3 == 1) ? hb1 : hb2;
        const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)hout, 0, nbytes, 0x00020000);
        float hsum = 0.f, dsum = 0.f;
#pragma unroll
        for (int J = 0; J < NB; ++J)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int t = 0; t < 4; ++t) hsum = fmaf(h[ct][J][t], h[ct][J][t], hsum);
        f32x4 pacc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) pacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (Cm is touched once per sweep and fetched one block ahead - cmn, requested before the loop and
        // at the end of the sweep before: alpha - Cm is what the block's accumulator starts from, so
        // the MFMAs leave the gradient itself)

    #pragma unroll
        for (int K = 0; K < NB; ++K) {
            // keep the compiler from hoisting later blocks' LDS reads above this point (register budget)
            __builtin_amdgcn_sched_barrier(0);
            // (row offsets are recomputed from an opaque copy of b per block: kept as common
            // subexpressions, the 4*NB scalar offsets would occupy most of the scalar register file)
            int bk = b;
            asm volatile("" : "+s"(bk));
            f32x4 acc0[CT], acc1[CT];
    #pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
    #pragma unroll
                for (int t = 0; t < 4; ++t) acc0[ct][t] = alpha_c[ct] - cmn[ct][t];
                acc1[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // The block's memory instructions - the 4 CT loads of the next block's Cm (the last block
            // fetches block 0 again: the next sweep starts without waiting for it) and the 4 CT stores of
            // the block before, whose rows are final - go out one at a time between the MFMA groups: each
            // issues while an MFMA runs.  Sent as bursts they hold the wave's issue slot for longer than
            // the MFMA in flight lasts.  The A operand of step J + 1 is read before the MFMAs of step J.
            const int Kn = (K + 1 < NB) ? K + 1 : 0;
            constexpr int NI = 8 * CT;
            // Transposed blocks of Gm beyond the 64 KB that an LDS instruction's offset field reaches: ONE
            // address addition for the block's four reads, made opaque - or the compiler folds the block's
            // offset into every read and spends an addition on each of them.  (The same for the rows of
            // directly read blocks and for the diagonal block's quads costs more than it saves: the lane
            // part of those addresses is recomputed per use.)
            typedef __attribute__((address_space(3))) const float lds_cfloat;
            auto read_a = [&](int J, float (&a)[4]) {
#ifdef ONMF_DIAG_DIRECT_ONLY
                // Counter diagnostics only (wrong results): blocks above the diagonal are read by rows like the
                // others, which takes the transposed reads - four ds_read_b32 per lane, issued as two
                // ds_read2_b32, the source of nearly all LDS bank-conflict cycles of the half layout - out of
                // the kernel.  The kernel is not measurably faster without them: they hide behind the matrix
                // and vector work.
                if (HALF && J > K) {
                    const float4 v = *reinterpret_cast<const float4*>(
                        Gs + blk_index<NB, HALF>(J, K) * 256 + c * 16 + 4 * (g ^ sw_c));
                    a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
                    return;
                }
#endif
                if (!HALF || J <= K) {
                    const float4 v = *reinterpret_cast<const float4*>(
                        Gs + blk_index<NB, HALF>(K, J) * 256 + c * 16 + 4 * (g ^ sw_c));
                    a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
                } else {
                    // Gm[16K+c][16J+4g+t] = Gm[16J+4g+t][16K+c]: rows 4g+t of block (J,K), column c
                    const int boff = blk_index<NB, HALF>(J, K) * 256;
                    if (boff * 4 + 1024 <= 65536) {
                        const float* blk = Gs + boff + 64 * g + 4 * ((c >> 2) ^ ((-g) & 3)) + (c & 3);
                        a[0] = blk[0]; a[1] = blk[16]; a[2] = blk[32]; a[3] = blk[48];
                    } else {
                        unsigned addr = (unsigned)(size_t)(lds_cfloat*)(Gs + 64 * g + 4 * ((c >> 2) ^ ((-g) & 3)) + (c & 3)) +
                                        (unsigned)boff * 4u;
                        asm volatile("" : "+v"(addr));
                        lds_cfloat* blk = (lds_cfloat*)(size_t)addr;
                        a[0] = blk[0]; a[1] = blk[16]; a[2] = blk[32]; a[3] = blk[48];
                    }
                }
            };
            float a_cur[4], a_nxt[4];
            read_a(0, a_cur);
    #pragma unroll
            for (int J = 0; J < NB; ++J) {
                if (J + 1 < NB) read_a(J + 1, a_nxt);
    #pragma unroll
                for (int i = J * NI / NB; i < (J + 1) * NI / NB; ++i) {
                    if (i < 4 * CT) {
                        cmn[i >> 2][i & 3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            rs_c, voff[i >> 2], (16 * Kn + (i & 3)) * bk * 4, 0));
                    } else if (!EACH && K > 0) {
                        const int ii = i - 4 * CT;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h[ii >> 2][K > 0 ? K - 1 : 0][ii & 3]), rs_o,
                                                              voff[ii >> 2], (16 * (K - 1) + (ii & 3)) * bk * 4, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc0[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[0], h[ct][J][0], acc0[ct], 0, 0, 0);
    #pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[1], h[ct][J][1], acc1[ct], 0, 0, 0);
    #pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc0[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[2], h[ct][J][2], acc0[ct], 0, 0, 0);
    #pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[3], h[ct][J][3], acc1[ct], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int t = 0; t < 4; ++t) a_cur[t] = a_nxt[t];
            }
            __builtin_amdgcn_sched_barrier(0);
            // acc = gradient of rows 4g+t of block K (without the in-block corrections)
            float acc[CT][4], hold4[CT][4];
    #pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
    #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[ct][t] = acc0[ct][t] + acc1[ct][t];
                    hold4[ct][t] = h[ct][K][t];
                }
            }

#if defined(ONMF_DIAG_SWEEP) && ONMF_DIAG_SWEEP == 1
            // timing diagnostics only (wrong results): no in-block update at all
            if constexpr (PAIR) {
    #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    h[0][K][t] = fmaxf(h[0][K][t] - 1e-4f * acc[0][t], 0.f);
                    h[CT - 1][K][t] = fmaxf(h[CT - 1][K][t] - 1e-4f * acc[CT - 1][t], 0.f);
                }
            } else
#endif
            if constexpr (PAIR) {
                float a8[8], h8[8], d8[8];
    #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const auto sa = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][t]), __float_as_uint(acc[CT - 1][t]), false, false);
                    a8[t] = __uint_as_float(sa[0]);
                    a8[4 + t] = __uint_as_float(sa[1]);
                    const auto sh = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[0][K][t]), __float_as_uint(h[CT - 1][K][t]), false, false);
                    h8[t] = __uint_as_float(sh[0]);
                    h8[4 + t] = __uint_as_float(sh[1]);
                }
                const int hh = g >> 1;
                // the diagonal block's address in a register of its own (made opaque): the entries read below,
                // the same in every lane, are then immediate offsets from it
                unsigned gd_addr = (unsigned)(size_t)(lds_cfloat*)(Gs + blk_index<NB, HALF>(K, K) * 256);
                asm volatile("" : "+v"(gd_addr));
                lds_cfloat* Gd = (lds_cfloat*)(size_t)gd_addr;
                typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;
                auto half_update = [&](auto ph_c) {
                    constexpr int ph = decltype(ph_c)::value;
                    const float4 n0 = *reinterpret_cast<const float4*>(steps + ((ph * 2) * NB + K) * 8 + hh * (NB * 8));
                    const float4 n1 = *reinterpret_cast<const float4*>(steps + ((ph * 2) * NB + K) * 8 + hh * (NB * 8) + 4);
                    const float nst[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
                    const float floor_p = (hh == ph) ? 0.f : -__builtin_inff();
    #pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        // row 8 ph + j of the block; its entries of Gm towards the later rows of this half
                        // (Gm is symmetric: column entries are read as the row's), the same in every lane
                        const int rj = 8 * ph + j;
                        float grow[16];
    #pragma unroll
                        for (int q4 = (rj + 1) >> 2; q4 < 2 * ph + 2; ++q4) {
#if defined(ONMF_DIAG_SWEEP) && ONMF_DIAG_SWEEP == 2
                            // timing diagnostics only (wrong results): the entries of Gm from registers
                            const f32x4 v = {nst[0], nst[1], nst[2], nst[3]};
#else
                            const f32x4 v = *reinterpret_cast<lds_cf32x4*>(Gd + rj * 16 + 4 * (q4 ^ swz(rj)));
#endif
                            grow[4 * q4 + 0] = v[0]; grow[4 * q4 + 1] = v[1]; grow[4 * q4 + 2] = v[2]; grow[4 * q4 + 3] = v[3];
                        }
                        const float hold = h8[j];
                        const float hnew = fmaxf(fmaf(a8[j], nst[j], hold), floor_p);
                        const float dl = hnew - hold;
                        h8[j] = hnew;
                        d8[j] = dl;
    #pragma unroll
                        for (int i = j + 1; i < 8; ++i) a8[i] = fmaf(grow[8 * ph + i], dl, a8[i]);
                    }
                };
                half_update(std::integral_constant<int, 0>{});
                // The first half's deltas into the second half's gradients: two MFMAs per tile on the
                // accumulators in their own layout (rows 8 .. 15 of them are untouched so far),
                // acc[16 rows][col] += G[rows][4q .. 4q+3] . delta[4q .. 4q+3][col].  The B operand wants delta
                // row 4q + g in lane group g: the deltas sit in lane group 0 (first tile) and 1 (second tile)
                // as registers 4q .. 4q+3, so a 4 x 4 exchange of register index against lane group leaves
                // the first tile's operand in register 0 and the second tile's in register 1.
    #pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float x0 = d8[4 * q], x1 = d8[4 * q + 1], x2 = d8[4 * q + 2], x3 = d8[4 * q + 3];
                    xpose_reg_group(x0, x1, x2, x3);
                    const float a_push = Gs[blk_index<NB, HALF>(K, K) * 256 + c * 16 + 4 * (q ^ sw_c) + g];
                    f32x4 av0 = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
                    f32x4 av1 = {acc[CT - 1][0], acc[CT - 1][1], acc[CT - 1][2], acc[CT - 1][3]};
                    av0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_push, x0, av0, 0, 0, 0);
                    av1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_push, x1, av1, 0, 0, 0);
    #pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc[0][t] = av0[t];
                        acc[CT - 1][t] = av1[t];
                    }
                }
    #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const auto sa = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][t]), __float_as_uint(acc[CT - 1][t]), false, false);
                    a8[t] = __uint_as_float(sa[0]);
                    a8[4 + t] = __uint_as_float(sa[1]);
                }
                half_update(std::integral_constant<int, 1>{});
    #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const auto sh = __builtin_amdgcn_permlane16_swap(__float_as_uint(h8[t]), __float_as_uint(h8[4 + t]), false, false);
                    h[0][K][t] = __uint_as_float(sh[0]);
                    h[CT - 1][K][t] = __uint_as_float(sh[1]);
                }
            } else {
    #pragma unroll
            for (int Q = 0; Q < 4; ++Q) {
                const bool own = (g == Q);
                // Gq[t][tj] = Gm[16K+4g+t][16K+4Q+tj] from the diagonal block (rows 4g+t, column quad Q)
                float Gq[4][4];
    #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 v = *reinterpret_cast<const float4*>(
                        Gs + blk_index<NB, HALF>(K, K) * 256 + (4 * g + t) * 16 + 4 * (Q ^ ((-g) & 3)));
                    Gq[t][0] = v.x; Gq[t][1] = v.y; Gq[t][2] = v.z; Gq[t][3] = v.w;
                }
                // Only lane group Q moves in this quad.  The other groups run the same instructions with
                // step 0 and floor -inf: h - 0 * grad = h, max(h, -inf) = h, delta = 0 exactly.
                const float4 n4 = *reinterpret_cast<const float4*>(steps + Q * 16 * NB + 16 * K + 4 * g);
                float nst[4] = {n4.x, n4.y, n4.z, n4.w};
                if (EACH) {
    #pragma unroll
                    for (int tj = 0; tj < 4; ++tj) nst[tj] = active ? nst[tj] : 0.f;
                }
                const float floor_q = own ? 0.f : -__builtin_inff();
                float dq[CT][4];
    #pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
    #pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const float hold = h[ct][K][tj];
                        const float hnew = fmaxf(fmaf(acc[ct][tj], nst[tj], hold), floor_q);
                        const float dl = hnew - hold;
                        h[ct][K][tj] = hnew;
                        dq[ct][tj] = dl;
                        // rows of this quad that come later (elsewhere dl == 0)
    #pragma unroll
                        for (int t = tj + 1; t < 4; ++t) acc[ct][t] = fmaf(Gq[t][tj], dl, acc[ct][t]);
                    }
                }
                if (Q < 3) {
                    // Push the four deltas into the gradients of the lane groups that come later in the
                    // block (the groups already done take the same update: their gradients are dead): one
                    // MFMA per tile,  acc[16 rows][col] += G[rows][4Q .. 4Q+3] . delta[4Q .. 4Q+3][col].
                    // Its B operand wants delta row 4Q + g in lane group g: the deltas sit in lane group Q
                    // as registers 0..3 (zero in the other groups), so a 4 x 4 exchange of register index
                    // against lane group (four v_permlane swaps) leaves exactly that in register Q.  The
                    // MFMA adds its four products in row order onto the accumulator, as the sixteen
                    // multiply-adds it replaces did (same bits); it costs 32 matrix-pipe cycles against
                    // their 16 vector instructions and the four LDS shuffles that fed them.
                    const float a_push = Gs[blk_index<NB, HALF>(K, K) * 256 + c * 16 + 4 * (Q ^ sw_c) + g];
    #pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        float x0 = dq[ct][0], x1 = dq[ct][1], x2 = dq[ct][2], x3 = dq[ct][3];
                        xpose_reg_group(x0, x1, x2, x3);
                        const float bq = Q == 0 ? x0 : Q == 1 ? x1 : x2;
                        f32x4 av = {acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]};
                        av = __builtin_amdgcn_mfma_f32_16x16x4f32(a_push, bq, av, 0, 0, 0);
    #pragma unroll
                        for (int t = 0; t < 4; ++t) acc[ct][t] = av[t];
                    }
                }
            }
            }   // (quad-by-quad form)
    #pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
    #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    hold4[ct][t] = h[ct][K][t] - hold4[ct][t];     // from here on: the block's deltas
                    dsum = fmaf(hold4[ct][t], hold4[ct][t], dsum);
                }
            }
            asm volatile("" : "+v"(dsum));     // keep the running sum current (no deferred chain)
            __builtin_amdgcn_s_setprio(0);
            // rows of block K are final: the next block's MFMA loop stores them; the last block's go now
            if (!EACH && K == NB - 1) {
    #pragma unroll
                for (int ct = 0; ct < CT; ++ct)
    #pragma unroll
                    for (int t = 0; t < 4; ++t)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h[ct][K][t]), rs_o, voff[ct],
                                                              (16 * K + t) * bk * 4, 0);
            }
            if (PROBE) {
                const float4 u4 = *reinterpret_cast<const float4*>(Us + K * 256 + c * 16 + 4 * (g ^ sw_c));
    #pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    pacc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(u4.x, hold4[ct][0], pacc[ct], 0, 0, 0);
                    pacc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(u4.y, hold4[ct][1], pacc[ct], 0, 0, 0);
                    pacc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(u4.z, hold4[ct][2], pacc[ct], 0, 0, 0);
                    pacc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(u4.w, hold4[ct][3], pacc[ct], 0, 0, 0);
                }
            }
        }

        if (EACH) {
            // the column's own test: its rows live in lanes c, c+16, c+32, c+48
            float dc = dsum, hc = hsum;
            dc += __shfl_xor(dc, 16);
            hc += __shfl_xor(hc, 16);
            dc += __shfl_xor(dc, 32);
            hc += __shfl_xor(hc, 32);
            if (active) {
                ++col_sweeps;
                // 0/0 (NaN) ends the loop, x/0 (+inf) does not: the comparison of src/onmf.py:252
                active = sqrt((double)dc) / sqrt((double)hc) > (double)es.tol;
            }
            if (k + 1 < k_first + k_count) __syncthreads();    // the next sweep rewrites the step sizes
            continue;
        }
        SWEEP_MARK(2);
        double ds = dsum, hh = hsum;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            ds += __shfl_xor(ds, off);
            hh += __shfl_xor(hh, off);
        }
        if (lane == 0) {
            const int slot = blockIdx.x * WAVES + wave;
            stats[2 * slot + 0] = ds;
            stats[2 * slot + 1] = hh;
            if (PROBE) {
                atomicAdd(&sum_of(k & 1)[0], ds);
                atomicAdd(&sum_of(k & 1)[1], hh);
            }
        }

        if (PROBE) {
            __builtin_amdgcn_sched_barrier(0);
            // energy of each probe's projection over this wave's columns: lane (c,g) reg t = probe 4g+t
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float e2 = 0.f;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) e2 = fmaf(pacc[ct][t], pacc[ct][t], e2);
                const float e = row_sum16(e2);
                if (c == 15) atomicAdd(&p2_of(k & 1)[4 * g + t], e);
            }
            if constexpr (CWB) {
                // t_col = v . H_new[:, col] (the lane's rows, then the four lane groups), y_row += H_new[row][col] t_col
                float tcol[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    float p = 0.f;
#pragma unroll
                    for (int J = 0; J < NB; ++J)
#pragma unroll
                        for (int t = 0; t < 4; ++t) p = fmaf(vcw[J][t], h[ct][J][t], p);
                    p += __shfl_xor(p, 16);
                    p += __shfl_xor(p, 32);
                    tcol[ct] = p;
                }
#pragma unroll
                for (int J = 0; J < NB; ++J)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        float yv = 0.f;
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) yv = fmaf(h[ct][J][t], tcol[ct], yv);
                        yv = row_sum16(yv);
                        if (c == 15) atomicAdd(&y_of(k & 1)[16 * J + 4 * g + t], yv);
                    }
            }
            SWEEP_MARK(3);
            __syncthreads();
            SWEEP_MARK(4);
            if constexpr (CTRL) continue;      // the control wave sends the sums off and evaluates the tests
            // One set of global atomics per workgroup, spread over replicas to keep contention low.
            // They return, and the count below goes out only after every one of them has: whoever
            // sees the count complete finds all sums in place, without any cache maintenance.
            const int rep = blockIdx.x % SCREEN_REPL;
            float* cur = sc.base + ScreenLayout::SETS + k * ScreenLayout::SETSZ;
            // (what the atomics return is consumed further down, behind the test evaluated ahead: the two
            // travel together)
            float back_f = 0.f;
            double back_d = 0.0;
            if (tid < 16) {
                back_f = atomicAdd(&cur[ScreenLayout::SET_P2 + rep * 16 + tid], p2wg[tid]);
            } else if (tid < 18) {
                back_d = atomicAdd(reinterpret_cast<double*>(cur + ScreenLayout::SET_SUMS) + rep * 2 + (tid - 16),
                                   sumwg[tid - 16]);
            }
            if (CWB) {
                const int yrep = blockIdx.x % SCREEN_YREPL;
                for (int j = tid; j < 16 * NB; j += NT)
                    back_d += atomicAdd(reinterpret_cast<double*>(cur + ScreenLayout::SET_Y) + yrep * SCREEN_CW_ROWS + j,
                                        (double)ywg[j]);
            }
            tested_ahead = false;
            if (wave == 0 && k + 1 < k_first + k_count && k - 1 >= v_first) {
                // (whatever this reads of the seeded bound - the value put there before the launch or the
                // minimum another workgroup has since taken with it - the test arrives at the same number)
                const int seen = __builtin_amdgcn_readfirstlane(
                    __hip_atomic_load(reinterpret_cast<const int*>(sc.base) + ScreenLayout::CNT + k - 1, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT));
                const float seeded = fmaxf(screen_load_bound(sc.base, k), 0.f);
                SWEEP_MARK(8);
                if (seen >= (int)gridDim.x && sc.spin_limit != SPIN_FORCE_TIMEOUT) {
                    verdict_ahead = screen_eval_wave(sc.base, k - 1, ub, sc.tol, sc.rank_cap, blockIdx.x == 0, seeded,
                                                     CWB ? 16 * NB : 0);
                    tested_ahead = true;
                }
            }
            SWEEP_MARK(5);
            asm volatile("" ::"v"(back_f), "v"(back_d));
            SWEEP_MARK(6);
            __syncthreads();
            SWEEP_MARK(7);
            if (tid == 0)
                __hip_atomic_fetch_add(reinterpret_cast<int*>(sc.base) + ScreenLayout::CNT + k, 1, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        } else if (k + 1 < k_first + k_count) {
            __syncthreads();               // the next sweep rewrites the step sizes
        }
    }
    if (EACH) {
        const int kend = k_first + k_count;
        float* hout = (kend % 3 == 0) ? hb0 : (kend % 3 == 1) ? hb1 : hb2;
        const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)hout, 0, nbytes, 0x00020000);
#pragma unroll
        for (int J = 0; J < NB; ++J)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h[0][J][t]), rs_o, voff[0], (16 * J + t) * b * 4, 0);
        if (es.sweeps != nullptr && g_lane == 0) {
            const int col = (blockIdx.x * WAVES + wave) * 16 + c_lane;
            if (col < b) es.sweeps[col] = col_sweeps;
        }
    }
}

template <int NB, bool HALF, bool PROBE, int CT, bool EACH = false, int WPG = 16 / CT>
int run_tile(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* const* hb, float alpha,
             int exact_cd, int k_first, int k_count, double* stats, int* n_groups, const ScreenDev& sc,
             hipStream_t s, EachStop es = EachStop{}) {
    constexpr int NBLK = HALF ? NB * (NB + 1) / 2 : NB * NB;
    constexpr bool CTRL = sweep_has_control_wave<NB, HALF, PROBE, CT, EACH, WPG>();
    constexpr size_t lds = sizeof(float) * ((size_t)NBLK * 256 + 5 * NB * 16 + (PROBE ? (NB * 256 + 16 + 4 + 4 + 16 * NB) : 0) +
                                            (CTRL ? (24 + 16 * NB) : 0));
    static_assert(lds <= 160 * 1024, "Gm blocks do not fit the LDS");
    static LdsLimit lim;
    ONMF_TRY(onmf_raise_lds_limit(c, lim, (const void*)sweep_tile_kernel<NB, HALF, PROBE, CT, EACH, WPG>, lds));
    constexpr int NT = 64 * WPG;
    constexpr int COLS = 16 * CT * WPG;    // columns per workgroup
    static_assert(WPG != 16 / CT || COLS == TILE_COLS_PER_WG, "the default geometry is 256 columns per workgroup");
    const int wgs = (b + COLS - 1) / COLS;
    *n_groups = wgs * WPG;
    hipLaunchKernelGGL((sweep_tile_kernel<NB, HALF, PROBE, CT, EACH, WPG>), dim3(wgs), dim3(NT + (CTRL ? 64 : 0)), lds, s, r, b, Gm, Cm,
                       hb[0], hb[1], hb[2], alpha, exact_cd, k_first, k_count, stats, sc, es);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

// all block counts for one CT (one translation unit per CT keeps the build parallel)
template <int CT>
int launch_sweep_tile(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, float* const* hb,
                      float alpha, int exact_cd, int k_first, int k_count, double* stats, int* n_groups,
                      hipStream_t s, const ScreenDev* screen) {
    const int nb = (r + 15) / 16;
    // Several sweeps per launch: either the screen ends the loop on the device, or the mode has no test
    // between the sweeps of a launch (exact coordinate descent, tested once per chunk of sweeps).
    if (k_count > 1 && !screen && !exact_cd)
        return onmf_fail(c, ONMF_ERR_ARG, "sweep: several damped sweeps per launch need the screen");
    ScreenDev none = {};
#define TILE_CASE(NBV, HALFV)                                                                            \
    case NBV:                                                                                            \
        return screen ? run_tile<NBV, HALFV, true, CT>(c, r, b, Gm, Cm, hb, alpha, exact_cd, k_first,   \
                                                       k_count, stats, n_groups, *screen, s)             \
                      : run_tile<NBV, HALFV, false, CT>(c, r, b, Gm, Cm, hb, alpha, exact_cd, k_first,  \
                                                        k_count, stats, n_groups, none, s)
    switch (nb) {
        TILE_CASE(1, false);
        TILE_CASE(2, false);
        TILE_CASE(3, false);
        TILE_CASE(4, false);
        TILE_CASE(5, false);
        TILE_CASE(6, false);
        TILE_CASE(7, false);
        TILE_CASE(8, false);
        TILE_CASE(9, false);
        TILE_CASE(10, false);
        TILE_CASE(11, false);
        TILE_CASE(12, true);
        TILE_CASE(13, true);
        TILE_CASE(14, true);
        TILE_CASE(15, true);
        TILE_CASE(16, true);
    }
#undef TILE_CASE
    return onmf_fail(c, ONMF_ERR_ARG, "sweep: unsupported r=%d", r);
}

}  // namespace
