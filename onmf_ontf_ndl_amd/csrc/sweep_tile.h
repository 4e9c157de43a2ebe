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

// sum over the 16 lanes of a DPP row (lanes 16g..16g+15); the total lands in lane 16g+15
__device__ __forceinline__ float row_sum16(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x111, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x112, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x114, 0xf, 0xf, true));
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x118, 0xf, 0xf, true));
    return x;
}

// PROBE = true adds what the stopping test needs without a second pass over H (see screen.hip):
//   * P = U^T (H_new - H_old) for 16 unit probe vectors U (4 extra MFMAs per block) and, per probe,
//     sum_c P[p][c]^2  -> a rigorous lower bound of sigma_max(H_new - H_old)^2;
//   * y = H_new (H_new^T v) for a positive vector v -> Collatz-Wielandt upper bound of
//     sigma_max(H_new)^2 (H H^T is entrywise nonnegative), used by the NEXT sweep's test;
// (The power step that refreshes the probe subspace runs on a sample of the columns in its own
// small kernel, screen.hip.)
template <int NB, bool HALF, bool PROBE, int CT>
__global__ __launch_bounds__(1024 / CT) void sweep_tile_kernel(
    int r, int b, const float* __restrict__ Gm, const float* __restrict__ Cm,
    const float* __restrict__ Hin, float* __restrict__ Hout, float alpha, float rdamp, int exact_cd,
    double* __restrict__ stats, ScreenDev sc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NBLK = HALF ? NB * (NB + 1) / 2 : NB * NB;
    constexpr int NT = 1024 / CT;          // threads per workgroup
    constexpr int WAVES = NT / 64;
    float* Gs = lds;                       // NBLK swizzled 16x16 blocks of Gm
    float* steps = lds + NBLK * 256;       // 16*NB step sizes (0 for padded rows)
    float* Us = steps + 16 * NB;           // PROBE: NB swizzled blocks of U^T
    float* vs = Us + NB * 256;             // PROBE: v
    float* ywg = vs + 16 * NB;             // PROBE: workgroup partial of y
    float* p2wg = ywg + 16 * NB;           // PROBE: workgroup partial of the 16 probe energies
    double* sumwg = reinterpret_cast<double*>(p2wg + 16);   // PROBE: workgroup partial of the two sums
    float* redwg = p2wg + 20;              // PROBE: 32 floats of reduction scratch
    int* verdict_s = reinterpret_cast<int*>(p2wg + 52);     // PROBE: the previous sweep's verdict

    const int tid = threadIdx.x;
    if (PROBE) {
        // an earlier sweep's test ended the loop (grid-uniform)
        if (reinterpret_cast<const int*>(sc.base)[ScreenLayout::FLAG] != 0) return;
    }

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
    for (int k = tid; k < 16 * NB; k += NT) {
        float st = 0.f;
        if (k < r) {
            const float gkk = Gm[(size_t)k * r + k];
            st = exact_cd ? (gkk > 0.f ? 1.0f / gkk : 0.f) : rdamp / (gkk + 1.0f);
        }
        steps[k] = st;
        if (PROBE) {
            vs[k] = sc.base[ScreenLayout::V + (sc.k & 1) * 256 + k];
            ywg[k] = 0.f;
        }
    }
    if (PROBE) {
        for (int e = tid; e < NB * 256; e += NT) Us[e] = sc.base[ScreenLayout::UBLK + e];
        if (tid < 16) p2wg[tid] = 0.f;
        if (tid < 2) sumwg[tid] = 0.0;
        // The previous sweep's stopping test, from the by-products it left in its accumulator set.
        // Every workgroup evaluates it (same numbers, same arithmetic, same answer) instead of
        // waiting for a kernel of its own between two sweeps; workgroup 0 records it.
        if (tid < 64) {
            int verdict = 0;
            if (!sc.skip_verdict) {
                const float* prev = sc.base + ScreenLayout::SETS + ((sc.k + 2) % 3) * ScreenLayout::SETSZ;
                float e = 0.f;
                if (tid < 16)
                    for (int rp = 0; rp < SCREEN_REPL; ++rp) e += prev[ScreenLayout::SET_P2 + rp * 16 + tid];
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) e = fmaxf(e, __shfl_xor(e, off));
                if (tid == 0) {
                    const double* sums = reinterpret_cast<const double*>(prev + ScreenLayout::SET_SUMS);
                    double dsum = 0.0, hsum = 0.0;
                    for (int rp = 0; rp < SCREEN_REPL; ++rp) {
                        dsum += sums[2 * rp];
                        hsum += sums[2 * rp + 1];
                    }
                    float lower, upper, ubh;
                    verdict = screen_verdict(e, dsum, hsum, sc.base[ScreenLayout::UBH + sc.k - 1], sc.tol,
                                             sc.rank_cap, lower, upper, ubh);
                    if (blockIdx.x == 0) {
                        float* rec = sc.base + ScreenLayout::REC + 4 * (sc.k - 1);
                        rec[0] = lower;
                        rec[1] = upper;
                        rec[2] = (float)(verdict + 1);        // 0 = no verdict yet
                        rec[3] = ubh;
                        if (verdict != 0) reinterpret_cast<int*>(sc.base)[ScreenLayout::FLAG] = verdict;
                    }
                }
            }
            if (tid == 0) *verdict_s = verdict;
        }
    }
    __syncthreads();
    if (PROBE) {
        if (*verdict_s != 0) return;       // workgroup-uniform, and the same in every workgroup
        if (blockIdx.x == 0) {
            // Housekeeping of workgroup 0, for the sweeps to come: the Collatz-Wielandt bound of
            // sigma_max(H before this sweep) from the previous sweep's y = H (H^T v), the vector v
            // of the next sweep, and a clean accumulator set for it.
            float* nxt_v = sc.base + ScreenLayout::V + ((sc.k + 1) & 1) * 256;   // also v of sweep k - 1
            if (sc.k > 0) {
                const float* prev = sc.base + ScreenLayout::SETS + ((sc.k + 2) % 3) * ScreenLayout::SETSZ;
                float yk = 0.f, ratio = 0.f;
                if (tid < r) {
                    for (int rp = 0; rp < SCREEN_REPL; ++rp) yk += prev[ScreenLayout::SET_Y + rp * 256 + tid];
                    ratio = yk / nxt_v[tid];                  // v > 0 by construction
                }
                float m0 = ratio, m1 = yk;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    m0 = fmaxf(m0, __shfl_xor(m0, off));
                    m1 = fmaxf(m1, __shfl_xor(m1, off));
                }
                if ((tid & 63) == 0) {
                    redwg[tid >> 6] = m0;
                    redwg[16 + (tid >> 6)] = m1;
                }
                __syncthreads();
                float cw = 0.f, ymax = 0.f;
                for (int wv = 0; wv < WAVES; ++wv) {
                    cw = fmaxf(cw, redwg[wv]);
                    ymax = fmaxf(ymax, redwg[16 + wv]);
                }
                if (tid == 0) sc.base[ScreenLayout::UBH + sc.k] = (cw > 0.f) ? sqrtf(cw) * (1.0f + 2e-4f) : -1.0f;
                if (tid < r) nxt_v[tid] = (ymax > 0.f) ? (yk / ymax + 1e-3f) : 1.0f;
            }
            float* nxt = sc.base + ScreenLayout::SETS + ((sc.k + 1) % 3) * ScreenLayout::SETSZ;
            for (int e = tid; e < ScreenLayout::SETSZ; e += NT) nxt[e] = 0.f;
        }
    }

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = lane & 15;
    const int g = lane >> 4;
    const int sw_c = swz(c);               // swizzle of row c (c doubles as the A-operand row index)

    // H and Cm go through buffer descriptors: byte address = (per-lane voffset) + (wave-uniform
    // soffset = row offset * b), no per-access address registers, and the hardware range check
    // returns 0 for rows >= r (their offset is past r*b floats) and drops such stores.  Columns
    // >= b get an out-of-range offset: they load zeros, run the arithmetic on zeros (alpha is
    // masked for them too, so they stay zero) and their stores are dropped.
    // The host guarantees r * b * 4 < 2^31.
    const unsigned nbytes = (unsigned)r * (unsigned)b * 4u;
    const auto rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)Hin, 0, nbytes, 0x00020000);
    const auto rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)Cm, 0, nbytes, 0x00020000);
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)Hout, 0, nbytes, 0x00020000);
    int voff[CT];
    float alpha_c[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = ((blockIdx.x * WAVES + wave) * CT + ct) * 16 + c;
        voff[ct] = col < b ? (4 * g * b + col) * 4 : (int)0x80000000;
        alpha_c[ct] = col < b ? alpha : 0.f;
    }
    float h[CT][NB][4];
    float hsum = 0.f, dsum = 0.f;
#pragma unroll
    for (int J = 0; J < NB; ++J) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = __uint_as_float(
                    __builtin_amdgcn_raw_buffer_load_b32(rs_h, voff[ct], (16 * J + t) * b * 4, 0));
                h[ct][J][t] = v;
                hsum = fmaf(v, v, hsum);
            }
        }
    }
    f32x4 pacc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) pacc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Cm is touched once per sweep and fetched one block ahead: alpha - Cm is what the block's
    // accumulator starts from, so the MFMAs leave the gradient itself
    float cmn[CT][4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int t = 0; t < 4; ++t)
            cmn[ct][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_c, voff[ct], t * b * 4, 0));

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
        if (K + 1 < NB) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    cmn[ct][t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rs_c, voff[ct], (16 * (K + 1) + t) * bk * 4, 0));
        }
#pragma unroll
        for (int J = 0; J < NB; ++J) {
            // bound how many A-operand reads are in flight at once (register budget)
            if (NB > 8 && J > 0 && (J % 6) == 0) __builtin_amdgcn_sched_barrier(0);
            float a0, a1, a2, a3;
            if (!HALF || J <= K) {
                const float4 a = *reinterpret_cast<const float4*>(
                    Gs + blk_index<NB, HALF>(K, J) * 256 + c * 16 + 4 * (g ^ sw_c));
                a0 = a.x; a1 = a.y; a2 = a.z; a3 = a.w;
            } else {
                // Gm[16K+c][16J+4g+t] = Gm[16J+4g+t][16K+c]: rows 4g+t of block (J,K), column c
                const float* blk = Gs + blk_index<NB, HALF>(J, K) * 256 + 64 * g +
                                   4 * ((c >> 2) ^ ((-g) & 3)) + (c & 3);
                a0 = blk[0]; a1 = blk[16]; a2 = blk[32]; a3 = blk[48];
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc0[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, h[ct][J][0], acc0[ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, h[ct][J][1], acc1[ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc0[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, h[ct][J][2], acc0[ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc1[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, h[ct][J][3], acc1[ct], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // acc = gradient of rows 4g+t of block K (without the in-block corrections)
        float acc[CT][4], hold4[CT][4];
        const float4 st4 = *reinterpret_cast<const float4*>(steps + 16 * K + 4 * g);
        const float st[4] = {st4.x, st4.y, st4.z, st4.w};
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[ct][t] = acc0[ct][t] + acc1[ct][t];
                hold4[ct][t] = h[ct][K][t];
            }
        }

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
            float nst[4];
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) nst[tj] = own ? -st[tj] : 0.f;
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
                // push the four deltas into the gradients of the lane groups that come later in the
                // block (the groups already done take the same update: their gradients are dead)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const float all = __shfl(dq[ct][tj], 16 * Q + c, 64);
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[ct][t] = fmaf(Gq[t][tj], all, acc[ct][t]);
                    }
                }
            }
        }
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
        // rows of block K are final: store them now, under the remaining blocks' arithmetic
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(h[ct][K][t]), rs_o, voff[ct],
                                                      (16 * K + t) * bk * 4, 0);
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
            atomicAdd(&sumwg[0], ds);
            atomicAdd(&sumwg[1], hh);
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
            if (c == 15) atomicAdd(&p2wg[4 * g + t], e);
        }
        // Collatz-Wielandt: ph[c] = sum_rows v[row] * H_new[row][c], then y[row] += H_new[row][c] * ph[c]
        float ph[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) ph[ct] = 0.f;
#pragma unroll
        for (int J = 0; J < NB; ++J) {
            if ((J & 3) == 0) __builtin_amdgcn_sched_barrier(0);
            const float4 v4 = *reinterpret_cast<const float4*>(vs + 16 * J + 4 * g);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                ph[ct] = fmaf(v4.x, h[ct][J][0], ph[ct]);
                ph[ct] = fmaf(v4.y, h[ct][J][1], ph[ct]);
                ph[ct] = fmaf(v4.z, h[ct][J][2], ph[ct]);
                ph[ct] = fmaf(v4.w, h[ct][J][3], ph[ct]);
            }
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            ph[ct] += __shfl_xor(ph[ct], 16);
            ph[ct] += __shfl_xor(ph[ct], 32);
        }
#pragma unroll
        for (int J = 0; J < NB; ++J) {
            __builtin_amdgcn_sched_barrier(0);     // one block at a time: bounded register use
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float yv = 0.f;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) yv = fmaf(h[ct][J][t], ph[ct], yv);
                const float e = row_sum16(yv);
                if (c == 15) atomicAdd(&ywg[16 * J + 4 * g + t], e);
            }
        }
        __syncthreads();
        // one set of global atomics per workgroup, spread over replicas to keep contention low
        const int rep = blockIdx.x % SCREEN_REPL;
        float* cur = sc.base + ScreenLayout::SETS + (sc.k % 3) * ScreenLayout::SETSZ;
        for (int k = tid; k < 16 * NB; k += NT) atomicAdd(&cur[ScreenLayout::SET_Y + rep * 256 + k], ywg[k]);
        if (tid < 16) atomicAdd(&cur[ScreenLayout::SET_P2 + rep * 16 + tid], p2wg[tid]);
        if (tid < 2) atomicAdd(reinterpret_cast<double*>(cur + ScreenLayout::SET_SUMS) + rep * 2 + tid, sumwg[tid]);
    }
}

template <int NB, bool HALF, bool PROBE, int CT>
int run_tile(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, const float* Hin, float* Hout,
             float alpha, float damp, int exact_cd, double* stats, int* n_groups, const ScreenDev& sc,
             hipStream_t s) {
    constexpr int NBLK = HALF ? NB * (NB + 1) / 2 : NB * NB;
    constexpr size_t lds = sizeof(float) * ((size_t)NBLK * 256 + NB * 16 +
                                            (PROBE ? (NB * 256 + 2 * NB * 16 + 16 + 4 + 32 + 4) : 0));
    static_assert(lds <= 160 * 1024, "Gm blocks do not fit the LDS");
    static bool attr_done = false;
    if (!attr_done) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)sweep_tile_kernel<NB, HALF, PROBE, CT>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    constexpr int NT = 1024 / CT;
    const int wgs = (b + TILE_COLS_PER_WG - 1) / TILE_COLS_PER_WG;
    *n_groups = wgs * (NT / 64);
    hipLaunchKernelGGL((sweep_tile_kernel<NB, HALF, PROBE, CT>), dim3(wgs), dim3(NT), lds, s, r, b,
                       Gm, Cm, Hin, Hout, alpha, 1.0f / damp, exact_cd, stats, sc);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

// all block counts for one CT (one translation unit per CT keeps the build parallel)
template <int CT>
int launch_sweep_tile(onmf_ctx* c, int r, int b, const float* Gm, const float* Cm, const float* Hin,
                      float* Hout, float alpha, float damp, int exact_cd, double* stats, int* n_groups,
                      hipStream_t s, const ScreenDev* screen) {
    const int nb = (r + 15) / 16;
    ScreenDev none = {};
#define TILE_CASE(NBV, HALFV)                                                                            \
    case NBV:                                                                                            \
        return screen ? run_tile<NBV, HALFV, true, CT>(c, r, b, Gm, Cm, Hin, Hout, alpha, damp,         \
                                                       exact_cd, stats, n_groups, *screen, s)            \
                      : run_tile<NBV, HALFV, false, CT>(c, r, b, Gm, Cm, Hin, Hout, alpha, damp,        \
                                                        exact_cd, stats, n_groups, none, s)
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
