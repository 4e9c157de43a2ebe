// fp32 MFMA matrix products for the ONMF step on gfx950.
//
// One kernel covers every contraction of the path:
//   Gm = W^T W, Cm = W^T X      (contraction index is the slow index of both operands)
//   H H^T, H X^T, X X^T         (contraction index is the fast index of both operands)
//   Qt = A^T W^T                (mixed)
// Operand tiles are staged in LDS as [x][k] (x = row of the output tile for A, column for B; the
// BK = 32 contraction indices of a row are contiguous), in 16-byte chunks of four k that are
// XOR-swizzled per row so that all three access patterns are bank-conflict-free: 16-byte stores
// when memory is [x][k], transposing 4-byte stores when memory is [k][x], and the 16-byte operand
// reads.  One such read hands a lane four k values of its row, i.e. its operand for four MFMAs:
// lanes 0-31 take chunk 2q, lanes 32-63 chunk 2q+1, so MFMA j of round q contracts k = 8q + j and
// 8q + 4 + j (any fixed pairing is valid as long as A and B use the same one).
// Each of the 4 waves owns a 64x64 block of the 128x128 output tile as 2x2
// v_mfma_f32_32x32x2_f32 accumulators (exact fp32).
// Long contractions (K = batch) are split over blockIdx.z into slabs that a second kernel sums
// in a fixed order, so results are bitwise reproducible.
#include "common.h"

#include <algorithm>
#include <type_traits>

namespace {

constexpr int BN = 128;
constexpr int BK = 32;
constexpr int NTHREADS = 256;
constexpr int TILE_ELEMS = 128 * BK;      // one operand tile in LDS (16 KB)
// workgroups of a split product, of the chip's 512 slots (two per CU).  510 is faster alone (the aggregate
// pair: 328 -> 315 us) and slower inside the step (1.616 -> 1.641 ms), where the dictionary pass and the
// small kernels of the other streams hold slots of their own: workgroups that find none run second.
#ifndef SPLIT_WG_BUDGET
#define SPLIT_WG_BUDGET 480
#endif

// dword offset of element (x, k) of an LDS operand tile
__device__ __forceinline__ int lds_off(int x, int k) {
    const int sw = (x & 7) ^ (2 * ((x >> 3) & 1));
    return x * BK + 4 * ((k >> 2) ^ sw) + (k & 3);
}

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
    // optional second B operand (two products that share A in one launch): the first n1 output columns
    // contract with B, padded to n1p columns (a multiple of BN; the padding is computed on zeros and
    // dropped), the columns from n1p on with B2; N = n1p + (columns of the second product); n1 == 0: none
    const float* B2;
    int ldb2, n1, n1p, vecB2;
    // finer grain at the end of an unsplit launch: workgroups from tail_first on compute the tiles from
    // tail_first on in tail_splits pieces of tail_k contraction steps each, into tile-shaped slabs
    // (tail_slab[(tile - tail_first) * tail_splits + piece][128][128]) that tail_reduce_kernel sums; -1: none
    int tail_first, tail_splits, tail_k;
    float* tail_slab;
};

// Which element of a BK x ROWS operand tile (ROWS = 128 or 64 rows of the output it feeds) thread t handles
// as its part p (ROWS / 32 float4 per thread).
// KMAJ (memory [k][x]): x = 4 chunk .. +3 with ROWS / 4 chunks per k-row - a wave reads 8 (or 4) k-rows
// of 128 (or 256) contiguous bytes; k = (t & 7) + 8 p for 128 rows, (t & 7) + 8 (2p + (t >> 7)) for 64.
// Otherwise (memory [x][k]): x = (t >> 3) + 32p, k = 4 (t & 7) .. +3.
template <bool KMAJ, int ROWS>
__device__ __forceinline__ int part_k(int t, int p) {
    return KMAJ ? (t & 7) + 8 * (ROWS == 128 ? p : 2 * p + (t >> 7)) : (t & 7) * 4;
}
template <bool KMAJ, int ROWS>
__device__ __forceinline__ int part_x(int t, int p) {
    return KMAJ ? 4 * ((t >> 3) & (ROWS / 4 - 1)) : (t >> 3) + 32 * p;
}

// Global -> registers for one BK x ROWS operand tile, checked element by element.
template <bool KMAJ, int ROWS>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int xdim, int x0,
                                          int k0, int kend, int vec_ok, float4 (&rg)[4]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k = k0 + part_k<KMAJ, ROWS>(t, p);
        const int x = x0 + part_x<KMAJ, ROWS>(t, p);
        if (KMAJ) {
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

// One part of: registers -> LDS tile S[x][k] (swizzled).
template <bool KMAJ, int ROWS>
__device__ __forceinline__ void store_part(float* __restrict__ S, const float4& v, int p) {
    const int t = threadIdx.x;
    const int k = part_k<KMAJ, ROWS>(t, p), x = part_x<KMAJ, ROWS>(t, p);
    if (KMAJ) {
        S[lds_off(x + 0, k)] = v.x;
        S[lds_off(x + 1, k)] = v.y;
        S[lds_off(x + 2, k)] = v.z;
        S[lds_off(x + 3, k)] = v.w;
    } else {
        *reinterpret_cast<float4*>(S + lds_off(x, k)) = v;
    }
}

template <bool KMAJ, int ROWS>
__device__ __forceinline__ void store_tile(float* __restrict__ S, const float4 (&rg)[4]) {
#pragma unroll
    for (int p = 0; p < ROWS / 32; ++p) store_part<KMAJ, ROWS>(S, rg[p], p);
}

// Tile source.  Interior tiles of aligned operands come through a buffer descriptor: the per-thread
// byte offsets are computed once, the tile origin is a scalar offset, so a tile costs no vector
// arithmetic at all (vector ALU cycles are not hidden behind fp32 MFMAs on this chip).  Edge tiles
// and unaligned operands take the checked path.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool KMAJ, int ROWS>
struct TileSource {
    static constexpr int NP = ROWS / 32;      // float4 per thread and tile
    const float* P;
    int ld, xdim, x0, kbeg, kend, vec_ok;
    bool fast;
    __amdgpu_buffer_rsrc_t rs;
    int voff[4];
    unsigned soff0, sstep;

    __device__ __forceinline__ void init(const float* P_, int ld_, int xdim_, int kdim, int x0_, int kbeg_,
                                         int kend_, int vec_ok_) {
        P = P_; ld = ld_; xdim = xdim_; x0 = x0_; kbeg = kbeg_; kend = kend_; vec_ok = vec_ok_;
        const int t = threadIdx.x;
        const size_t rows = KMAJ ? (size_t)kdim : (size_t)xdim;
        const size_t bytes = rows * (size_t)ld * sizeof(float);
        // A tile that reaches beyond the operand's last row / column (x >= xdim) stays on this path when the
        // buffer is small enough for an offset that is out of range by itself: lanes whose x is beyond the
        // edge get that offset and the hardware's range check hands them zeros - no branch, no select.
        // (Operands narrower than a tile - r = 16 ... 100 - would otherwise take the checked path for every
        // tile of every product.)  [k][x] memory: a lane's four x must be all inside or all outside.
        const bool whole = x0 + ROWS <= xdim;
        const bool maskable = bytes < 0x7ffff000ull && (!KMAJ || (xdim & 3) == 0);
        fast = vec_ok && bytes < 0xfffff000ull && (whole || maskable);
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)P, 0, (unsigned)(fast ? bytes : 0), 0x00020000);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int k = part_k<KMAJ, ROWS>(t, p), x = part_x<KMAJ, ROWS>(t, p);
            voff[p] = KMAJ ? (k * ld + x) * 4 : (x * ld + k) * 4;
            if (x0 + x >= xdim) voff[p] = (int)0x80000000u;
        }
        soff0 = KMAJ ? (unsigned)(((size_t)kbeg * ld + x0) * 4) : (unsigned)(((size_t)x0 * ld + kbeg) * 4);
        sstep = KMAJ ? (unsigned)((size_t)BK * ld * 4) : (unsigned)(BK * 4);
    }
    // NT: non-temporal loads - for an operand that is streamed (each element read by one or two
    // workgroups), so that it does not push the small matrices other kernels work on out of the L2
    // SURE: the caller has checked that this source is `fast` and the tile a whole one: no branch at all,
    // so that the compiler can count the loads in flight (behind a branch it falls back to waiting for
    // ALL of them wherever it needs one, which turns a prefetch two tiles ahead into half a tile)
    template <bool NT = false, bool SURE = false>
    __device__ __forceinline__ void load(int tile, float4 (&rg)[4]) const {
        const int k0 = kbeg + tile * BK;
        if (SURE || (fast && k0 + BK <= kend)) {             // uniform
            const unsigned so = soff0 + (unsigned)tile * sstep;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[p], so, NT ? 2 : 0);
                rg[p] = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]),
                                    __uint_as_float(v[3]));
            }
        } else {
            load_tile<KMAJ, ROWS>(P, ld, xdim, x0, k0, kend, vec_ok, rg);
        }
    }
    // one quarter of a whole tile of a `fast` source
    template <bool NT>
    __device__ __forceinline__ void load_part(int tile, int p, float4& v) const {
        const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[p], soff0 + (unsigned)tile * sstep, NT ? 2 : 0);
        v = make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
    }
};

#ifdef ONMF_TRACE_WG
// Diagnostic build only (tools/wg_trace.py): when and where every workgroup ran.
__device__ unsigned long long g_wg_trace[4 * 65536];
__device__ unsigned g_wg_trace_n;
#endif

// BMT: rows of the output tile, 128 or 64.  With 64 (products whose output has at most 64 rows: a 128-row tile
// would spend half of its matrix instructions on rows of zeros) the four waves sit side by side, each on a
// 64 x 32 block; the A tile is half as large and comes as two parts per thread instead of four.
template <bool A_KM, bool B_KN, int BMT>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(GemmParams p) {
    constexpr int AP = BMT / 32;                 // parts of the A tile per thread
    constexpr int NJ = BMT == 128 ? 2 : 1;       // 32-column blocks per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef ONMF_TRACE_WG
    const unsigned long long trace_t0 = wall_clock64();
#endif
    // layout: [A buf0][A buf1][B buf0][B buf1]
    constexpr int A_STRIDE = BMT * BK;
    constexpr int B_STRIDE = TILE_ELEMS;
    float* As = lds;
    float* Bs = lds + 2 * A_STRIDE;

    // Workgroups go to the 8 XCDs in turn (linear id mod 8), and each XCD has its own L2.  Workgroups that
    // read the same operand slice are therefore given ids 8 apart - same XCD, next to each other in time -
    // so that the slice crosses the fabric once instead of once per XCD that happens to want it:
    //  - split contractions: all tiles of one slab (they share the slab's slices of A and B);
    //  - unsplit products with several rows of tiles: the tiles of one column of tiles (they share B's).
    // Which workgroup computes which tile does not change any result.
    // MEASURED AND NOT ADOPTED (diagnostic builds only, -DONMF_GEMM_XCD_MAP): at d=1024 r=256 b=65536 the
    // bytes that cross the fabric per step fall from 4.35 to 3.17 GB (aggregate pair 957 -> 390 MB per
    // launch, W^T X 548 -> 300 MB), and both products get SLOWER (aggregate pair 318 -> 333 us, W^T X
    // 234 -> 240 us, step 1.587 -> 1.605 ms): they are bound by the matrix pipe, and workgroups of one XCD
    // that walk the same slice in step all miss on the same lines at the same time.
#ifdef ONMF_GEMM_XCD_MAP
    int tile = blockIdx.x, z = blockIdx.z;
    {
        const int G = gridDim.x, S = gridDim.z;
        if (S >= 8) {
            const int L = tile + G * z, full = (S >> 3) << 3;
            if (L < full * G) {
                const int slot = L >> 3;
                z = (slot / G) * 8 + (L & 7);
                tile = slot % G;
            }
        } else if (S == 1) {
            const int tm = G / p.tiles_n;                 // rows of tiles
            const int full = (p.tiles_n >> 3) << 3;       // columns of tiles that go in eights
            if (tm > 1 && tile < full * tm) {
                const int grp = tile / (8 * tm), rem = tile - grp * 8 * tm;
                tile = (rem >> 3) * p.tiles_n + grp * 8 + (rem & 7);
            } else if (tm > 1) {
                const int rest = p.tiles_n - full, t2 = tile - full * tm;
                tile = (t2 / rest) * p.tiles_n + full + t2 % rest;
            }
        }
    }
#else
    int tile = blockIdx.x;
    const int z = blockIdx.z;
#endif
    int kbeg = z * p.k_per_split;
    int kend = min(p.K, kbeg + p.k_per_split);
    int tail_piece = -1;                         // index of this workgroup's slab when it computes a piece of a tail tile
    if (p.tail_first >= 0 && tile >= p.tail_first) {
        tail_piece = tile - p.tail_first;
        tile = p.tail_first + tail_piece / p.tail_splits;
        kbeg = (tail_piece % p.tail_splits) * p.tail_k;
        kend = min(p.K, kbeg + p.tail_k);
    }
    const int m0 = (tile / p.tiles_n) * BMT;
    const int n0 = (tile % p.tiles_n) * BN;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wm = BMT == 128 ? (wave >> 1) * 64 : 0;
    const int wn = BMT == 128 ? (wave & 1) * 64 : wave * 32;
    const int l31 = lane & 31;
    const int lhi = lane >> 5;

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Software pipeline over the k-tiles, two tiles deep: tile t+2 travels from memory to
    // registers while tile t+1 (in registers since the previous round) is written to the free LDS
    // buffer in the middle of tile t's MFMAs.  The barrier at the end of a round then waits for
    // LDS stores only, never for memory.  Rounds come in pairs so the two register sets swap
    // roles without copies.
    float4 ra[4], rb[4], ra2[4], rb2[4];
    const int ntiles = (kend - kbeg + BK - 1) / BK;
    TileSource<A_KM, BMT> srcA;
    TileSource<B_KN, 128> srcB;
    srcA.init(p.A, p.lda, p.M, p.K, m0, kbeg, kend, p.vecA);
    if (p.n1 > 0 && n0 >= p.n1p) srcB.init(p.B2, p.ldb2, p.N - p.n1p, p.K, n0 - p.n1p, kbeg, kend, p.vecB2);
    else srcB.init(p.B, p.ldb, p.n1 > 0 ? p.n1 : p.N, p.K, n0, kbeg, kend, p.vecB);
    // interior workgroup of aligned operands over whole k-tiles: the branch-free loads (uniform)
    const bool sure = srcA.fast && srcB.fast && ((kend - kbeg) % BK) == 0 && ntiles > 0;

    auto pipeline = [&](auto sure_c) {
        constexpr bool SURE = decltype(sure_c)::value;
        if (SURE) {
            // tiles 0 and 1 travel together; the loop is entered with no load in flight (a load still
            // pending at the loop's entry makes the compiler wait at the loop's top on EVERY trip, for
            // registers it reuses there - which would cut the prefetch distance to half a tile again)
            srcA.template load<false, true>(0, ra2);
            srcB.template load<true, true>(0, rb2);
            srcA.template load<false, true>(min(1, ntiles - 1), ra);
            srcB.template load<true, true>(min(1, ntiles - 1), rb);
            store_tile<A_KM, BMT>(As, ra2);
            store_tile<B_KN, 128>(Bs, rb2);
            __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
        } else if (ntiles > 0) {
            srcA.template load<false, false>(0, ra);
            srcB.template load<true, false>(0, rb);
            store_tile<A_KM, BMT>(As, ra);
            store_tile<B_KN, 128>(Bs, rb);
            if (ntiles > 1) {
                srcA.template load<false, false>(1, ra);
                srcB.template load<true, false>(1, rb);
            }
        }
        __syncthreads();

        // one round: MFMAs of tile t; (xa, xb) hold tile t+1, (ya, yb) receive tile t+2
        auto round = [&](int t, float4 (&xa)[4], float4 (&xb)[4], float4 (&ya)[4], float4 (&yb)[4]) {
            const int cur = t & 1;
            // past the end: the last tile again (never used) rather than a branch around the loads
            const int tl = min(t + 2, ntiles - 1);
            if (!SURE && t + 2 < ntiles) {
                srcA.template load<false, false>(t + 2, ya);
                srcB.template load<true, false>(t + 2, yb);
            }
            const float* Ac = As + cur * A_STRIDE;
            const float* Bc = Bs + cur * B_STRIDE;
            float* An = As + (cur ^ 1) * A_STRIDE;
            float* Bn = Bs + (cur ^ 1) * B_STRIDE;
            // operands of step q + 1 are read while the MFMAs of step q run (left to itself the compiler
            // reads them after the MFMAs and waits: an LDS round trip with an idle matrix pipe per step)
            auto fetch = [&](int q, float4& a0, float4& a1, float4& b0, float4& b1) {
                const int k = 8 * q + 4 * lhi;
                a0 = *reinterpret_cast<const float4*>(Ac + lds_off(wm + l31, k));
                a1 = *reinterpret_cast<const float4*>(Ac + lds_off(wm + 32 + l31, k));
                b0 = *reinterpret_cast<const float4*>(Bc + lds_off(wn + l31, k));
                if (NJ == 2) b1 = *reinterpret_cast<const float4*>(Bc + lds_off(wn + 32 + l31, k));
            };
            float4 a0, a1, b0, b1, na0, na1, nb0, nb1;
            fetch(0, a0, a1, b0, b1);
#pragma unroll
            for (int q = 0; q < BK / 8; ++q) {
                if (q + 1 < BK / 8) fetch(q + 1, na0, na1, nb0, nb1);
                if (!SURE && q == BK / 16 && t + 1 < ntiles) {
                    store_tile<A_KM, BMT>(An, xa);
                    store_tile<B_KN, 128>(Bn, xb);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
                const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // SURE: the tile's 8 loads (step 0) and 8 stores (step 2) go out one at a time, each
                    // behind a pair of MFMAs - it issues while they run.  Sent as a burst, the memory
                    // instructions hold the wave's issue slot for longer than the one MFMA in flight
                    // lasts (measured on this loop: 343 us with bursts, 318 us interleaved, 307 us with
                    // no memory instructions at all).
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv0[j], acc[0][0], 0, 0, 0);
                    if (NJ == 2) acc[0][NJ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[j], bv1[j], acc[0][NJ - 1], 0, 0, 0);
                    if (SURE && (q == 0 || q == BK / 16) && j < AP) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (q == 0) srcA.template load_part<false>(tl, j, ya[j]);
                        else store_part<A_KM, BMT>(An, xa[j], j);     // (last round: a tile nobody reads)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv0[j], acc[1][0], 0, 0, 0);
                    if (NJ == 2) acc[1][NJ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[j], bv1[j], acc[1][NJ - 1], 0, 0, 0);
                    if (SURE && (q == 0 || q == BK / 16)) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (q == 0) srcB.template load_part<true>(tl, j, yb[j]);
                        else store_part<B_KN, 128>(Bn, xb[j], j);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
            __syncthreads();
        };
        // whole pairs of rounds in the loop, an odd last round behind it: a path from the first round of
        // a trip straight back to the loop's top would carry that round's loads, still in flight, there
        int t = 0;
        for (; t + 1 < ntiles; t += 2) {
            round(t, ra, rb, ra2, rb2);
            round(t + 1, ra2, rb2, ra, rb);
            // the accumulators stay in the accumulation registers across the loop edge (left alone, the
            // compiler carries them through vector registers: 128 copies per trip)
            if (NJ == 2) asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][NJ - 1]), "+a"(acc[1][0]), "+a"(acc[1][NJ - 1]));
            else asm volatile("" : "+a"(acc[0][0]), "+a"(acc[1][0]));
        }
        if (t < ntiles) round(t, ra, rb, ra2, rb2);
    };
#ifdef ONMF_TRACE_WG
    const unsigned long long trace_t1 = wall_clock64();        // set-up done, nothing loaded yet
#endif
    if (sure) pipeline(std::true_type{});
    else pipeline(std::false_type{});
#ifdef ONMF_TRACE_WG
    const unsigned long long trace_t2 = wall_clock64();        // main loop done
#endif

    // accumulator element e of lane l: row (e&3) + 8*(e>>2) + 4*(l>>5), column l&31
    if (tail_piece >= 0) {
        // a piece of a tail tile: the bare sums, tile-shaped
        float* slab = p.tail_slab + (size_t)tail_piece * (BMT * BN);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    slab[(wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * lhi) * BN + wn + 32 * j + l31] = acc[i][j][e];
        return;
    }
    const bool direct = (p.splits == 1);
    float* out = direct ? p.C : p.slab + (size_t)z * p.M * p.N;
    const int ldo = direct ? p.ldc : p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
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
#ifdef ONMF_TRACE_WG
    if (threadIdx.x == 0) {
        const unsigned slot = atomicAdd(&g_wg_trace_n, 1u);
        if (slot < 65536u) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_wg_trace[4 * slot + 0] = trace_t0;
            g_wg_trace[4 * slot + 1] = wall_clock64();
            // (bits 36..49 / 50..63: set-up and main loop done, in units of 4 clock ticks after the start)
            const unsigned long long d1 = ((trace_t1 - trace_t0) >> 2) & 0x3fffull, d2 = ((trace_t2 - trace_t0) >> 2) & 0x3fffull;
            g_wg_trace[4 * slot + 2] = ((unsigned long long)(xcc & 15u) << 32) | hw | (d1 << 36) | (d2 << 50);
            g_wg_trace[4 * slot + 3] = ((unsigned long long)gridDim.x << 32) | ((unsigned long long)gridDim.z << 20) |
                                       (A_KM ? 2u : 0u) | (B_KN ? 1u : 0u);
        }
    }
#endif
}

// Sum of the split-K slabs with the  alpha * sum + beta * Cin  epilogue.  64 outputs per workgroup,
// four threads per output: thread group g adds slabs g, g+4, g+8, ... in order, the four partial sums
// are added in order - a fixed summation order whatever the grid, so results are reproducible, and a
// few outputs with many slabs (the small Gram matrices) still spread over the chip.
// With n1 > 0 the columns below n1 go to C, those from n1p on to a second output (C2, ldc2) with their own
// beta2 * Cin2, and those in between (padding) nowhere.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab,
                                                             int splits, int M, int N, float alpha,
                                                             float beta, const float* Cin,
                                                             float* C, int ldc, int n1, int n1p, float beta2,
                                                             const float* Cin2, float* C2, int ldc2) {
    __shared__ float part[4][64];
    const size_t mn = (size_t)M * N;
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (size_t base = (size_t)blockIdx.x * 64; base < mn; base += (size_t)gridDim.x * 64) {
        const size_t i = base + o;
        float s = 0.f;
        if (i < mn)
            for (int z = g; z < splits; z += 4) s += slab[(size_t)z * mn + i];
        part[g][o] = s;
        __syncthreads();
        if (g == 0 && i < mn) {
            const float t = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
            const int row = (int)(i / N), col = (int)(i % N);
            float v = alpha * t;
            if (n1 > 0 && col >= n1 && col < n1p) {
                // padding
            } else if (n1 > 0 && col >= n1p) {
                const size_t at = (size_t)row * ldc2 + (col - n1p);
                if (beta2 != 0.f) v += beta2 * Cin2[at];
                C2[at] = v;
            } else {
                if (beta != 0.f) v += beta * Cin[(size_t)row * ldc + col];
                C[(size_t)row * ldc + col] = v;
            }
        }
        __syncthreads();
    }
}

// Sum of the pieces of the tail tiles (fixed order) into C: one workgroup per tile, four outputs per thread
// and trip.
__global__ __launch_bounds__(256) void tail_reduce_kernel(const float* __restrict__ slab, int tail_first, int tail_tiles,
                                                           int splits, int tiles_n, int bmt, int M, int N, float alpha,
                                                           float* __restrict__ C, int ldc) {
    const int tile = tail_first + blockIdx.x;
    const int m0 = (tile / tiles_n) * bmt, n0 = (tile % tiles_n) * BN;
    const float* base = slab + (size_t)blockIdx.x * splits * (bmt * BN);
    for (int e = threadIdx.x * 4; e < bmt * BN; e += 256 * 4) {
        float4 sum = *reinterpret_cast<const float4*>(base + e);
        for (int z = 1; z < splits; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)z * (bmt * BN) + e);
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        const int row = m0 + e / BN, col = n0 + e % BN;
        if (row >= M) continue;
        float* dst = C + (size_t)row * ldc + col;
        if (col + 3 < N && ((ldc & 3) == 0) && (((uintptr_t)C & 15) == 0)) {
            *reinterpret_cast<float4*>(dst) = make_float4(alpha * sum.x, alpha * sum.y, alpha * sum.z, alpha * sum.w);
        } else {
            if (col + 0 < N) dst[0] = alpha * sum.x;
            if (col + 1 < N) dst[1] = alpha * sum.y;
            if (col + 2 < N) dst[2] = alpha * sum.z;
            if (col + 3 < N) dst[3] = alpha * sum.w;
        }
    }
}

template <bool A_KM, bool B_KN, int BMT>
int run_bm(onmf_ctx* c, const GemmParams& p, dim3 grid, hipStream_t s) {
    constexpr size_t bytes = sizeof(float) * (2 * BMT * BK + 2 * TILE_ELEMS);
    static LdsLimit lim;
    ONMF_TRY(onmf_raise_lds_limit(c, lim, (const void*)gemm_f32_kernel<A_KM, B_KN, BMT>, bytes));
    hipLaunchKernelGGL((gemm_f32_kernel<A_KM, B_KN, BMT>), grid, dim3(NTHREADS), bytes, s, p);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

// rows of the output tile for an output of M rows
inline int tile_rows(int M) { return M <= 64 ? 64 : 128; }

template <bool A_KM, bool B_KN>
int run(onmf_ctx* c, const GemmParams& p, dim3 grid, hipStream_t s) {
    return tile_rows(p.M) == 64 ? run_bm<A_KM, B_KN, 64>(c, p, grid, s) : run_bm<A_KM, B_KN, 128>(c, p, grid, s);
}

}  // namespace

int launch_gemm(onmf_ctx* c, bool a_km, bool b_kn, int M, int N, int K, const float* A, int lda,
                const float* B, int ldb, float alpha, float beta, const float* Cin, float* C,
                int ldc, hipStream_t s, DevBuf* slab, int tail_spec) {
    if (!slab) slab = &c->slab;
    if (M <= 0 || N <= 0 || K < 0) return onmf_fail(c, ONMF_ERR_ARG, "gemm: bad shape %dx%dx%d", M, N, K);
    if (beta != 0.f && Cin == nullptr) return onmf_fail(c, ONMF_ERR_ARG, "gemm: beta without Cin");
    GemmParams p;
    p.A = A; p.B = B; p.Cin = Cin; p.C = C; p.slab = nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.beta = beta;
    p.vecA = (((uintptr_t)A & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
    p.vecB = (((uintptr_t)B & 15) == 0 && (ldb & 3) == 0) ? 1 : 0;
    p.B2 = nullptr; p.ldb2 = 0; p.n1 = 0; p.n1p = 0; p.vecB2 = 0;
    p.tail_first = -1; p.tail_splits = 0; p.tail_k = 0; p.tail_slab = nullptr;
    const int bm = tile_rows(M);
    const int tiles_m = (M + bm - 1) / bm, tiles_n = (N + BN - 1) / BN;
    p.tiles_n = tiles_n;
    const int tiles = tiles_m * tiles_n;
    const int ktiles = (K + BK - 1) / BK;
    int splits = 1;
    if (tiles < 256 && ktiles >= 4) {
        // few tiles: one workgroup per CU is enough, and half as many slabs for the reduction to read;
        // otherwise a little under two per CU, so that the launch is still ONE round of workgroups when
        // a small kernel on another stream holds a few of the chip's slots (with exactly two per CU a
        // single missing slot leaves one workgroup for a second round: up to twice the time)
        splits = (tiles <= 4 ? 256 + tiles - 1 : SPLIT_WG_BUDGET) / tiles;
        const int max_by_k = ktiles / 2;          // at least 2 k-tiles of work per slab
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
    // An unsplit product of several rounds of workgroups ends with whichever CU drew one workgroup more than
    // the others - a whole tile's time when other kernels hold places.  tail_spec = 100 * percent + pieces:
    // the last `percent` of the tiles are computed in `pieces` pieces along the contraction (workgroups
    // that short fill the last round evenly) and summed by a small kernel behind the launch.
    int tail_tiles = 0;
    if (tail_spec > 0 && splits == 1 && beta == 0.f && tiles >= 512) {
        const int pieces = tail_spec % 100, percent = tail_spec / 100;
        if (pieces >= 2 && percent >= 1 && percent <= 90 && ktiles >= 4 * pieces) {
            tail_tiles = tiles * percent / 100;
            const int kt = (ktiles + pieces - 1) / pieces;
            p.tail_first = tiles - tail_tiles;
            p.tail_splits = (ktiles + kt - 1) / kt;
            p.tail_k = kt * BK;
            ONMF_TRY(onmf_reserve(c, *slab, (size_t)tail_tiles * p.tail_splits * bm * BN * sizeof(float), "tail slabs"));
            p.tail_slab = (float*)slab->p;
            grid.x = p.tail_first + tail_tiles * p.tail_splits;
        }
    }
    int rc;
    if (a_km && b_kn) rc = run<true, true>(c, p, grid, s);
    else if (a_km && !b_kn) rc = run<true, false>(c, p, grid, s);
    else if (!a_km && b_kn) rc = run<false, true>(c, p, grid, s);
    else rc = run<false, false>(c, p, grid, s);
    if (rc != ONMF_OK) return rc;
    if (tail_tiles > 0) {
        hipLaunchKernelGGL(tail_reduce_kernel, dim3(tail_tiles), dim3(256), 0, s, p.tail_slab, p.tail_first, tail_tiles,
                           p.tail_splits, tiles_n, bm, M, N, alpha, C, ldc);
        HIP_TRY(c, hipGetLastError());
    }
    if (splits > 1) {
        const size_t mn = (size_t)M * N;
        int blocks = (int)((mn + 63) / 64);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.slab, splits, M, N,
                           alpha, beta, Cin, C, ldc, 0, 0, 0.f, (const float*)nullptr, (float*)nullptr, 0);
        HIP_TRY(c, hipGetLastError());
    }
    return ONMF_OK;
}

// C1 = beta1*Cin1 + alpha * A B1^T (M x N1) and C2 = beta2*Cin2 + alpha * A B2^T (M x N2) in ONE launch when
// the shapes allow (A [m][k], B1 [n][k], B2 [n][k]; N1 a multiple of the tile width; a contraction
// long enough to be split): one round of workgroups over both outputs and one reduction, instead of
// two launches each with its own tail.  Otherwise two ordinary products.
int launch_gemm_nt_pair(onmf_ctx* c, int M, int K, const float* A, int lda, const float* B1, int ldb1, int N1,
                        const float* B2, int ldb2, int N2, float alpha, float beta1, const float* Cin1, float* C1,
                        int ldc1, float beta2, const float* Cin2, float* C2, int ldc2, hipStream_t s) {
    return launch_gemm_pair(c, false, false, M, K, A, lda, B1, ldb1, N1, B2, ldb2, N2, alpha, beta1, Cin1, C1, ldc1,
                            beta2, Cin2, C2, ldc2, s, nullptr);
}

int launch_gemm_pair(onmf_ctx* c, bool a_km, bool b_kn, int M, int K, const float* A, int lda, const float* B1,
                     int ldb1, int N1, const float* B2, int ldb2, int N2, float alpha, float beta1, const float* Cin1,
                     float* C1, int ldc1, float beta2, const float* Cin2, float* C2, int ldc2, hipStream_t s,
                     DevBuf* slab) {
    if (!slab) slab = &c->slab;
    // the product whose width is a whole number of tiles goes first: nothing to pad then
    if ((N1 % BN) != 0 && (N2 % BN) == 0)
        return launch_gemm_pair(c, a_km, b_kn, M, K, A, lda, B2, ldb2, N2, B1, ldb1, N1, alpha, beta2, Cin2, C2, ldc2,
                                beta1, Cin1, C1, ldc1, s, slab);
    const int N1p = (N1 + BN - 1) / BN * BN;          // the second product's tiles start on a tile boundary
    const int N = N1p + N2;
    const int bm = tile_rows(M);
    const int tiles_m = (M + bm - 1) / bm, tiles_n = (N + BN - 1) / BN;
    const int tiles = tiles_m * tiles_n;
    const int ktiles = (K + BK - 1) / BK;
    // (a handful of tiles: one workgroup per CU, as for a single product of that size)
    int splits = (tiles < 256 && ktiles >= 4) ? std::min((tiles <= 4 ? 256 + tiles - 1 : SPLIT_WG_BUDGET) / tiles, ktiles / 2)
                                              : 1;
    // (padding of more than half a tile per row of tiles of the first product is not worth the launch saved
    // when that product is most of the work - unless the pair is a handful of tiles, where a launch and its
    // reduction cost more than any padding)
    const bool pad_ok = N1p == N1 || (N1p - N1) * 4 <= N || tiles <= 4;
    if (!pad_ok || splits < 2 || (beta1 != 0.f && !Cin1) || (beta2 != 0.f && !Cin2)) {
        ONMF_TRY(launch_gemm(c, a_km, b_kn, M, N1, K, A, lda, B1, ldb1, alpha, beta1, Cin1, C1, ldc1, s, slab));
        return launch_gemm(c, a_km, b_kn, M, N2, K, A, lda, B2, ldb2, alpha, beta2, Cin2, C2, ldc2, s, slab);
    }
    GemmParams p;
    p.A = A; p.B = B1; p.Cin = Cin1; p.C = C1;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb1; p.ldc = ldc1;
    p.alpha = alpha; p.beta = beta1;
    p.vecA = (((uintptr_t)A & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
    p.vecB = (((uintptr_t)B1 & 15) == 0 && (ldb1 & 3) == 0) ? 1 : 0;
    p.B2 = B2; p.ldb2 = ldb2; p.n1 = N1; p.n1p = N1p;
    p.vecB2 = (((uintptr_t)B2 & 15) == 0 && (ldb2 & 3) == 0) ? 1 : 0;
    p.tiles_n = tiles_n;
    p.tail_first = -1; p.tail_splits = 0; p.tail_k = 0; p.tail_slab = nullptr;
    const int kt_per = (ktiles + splits - 1) / splits;
    splits = (ktiles + kt_per - 1) / kt_per;
    p.k_per_split = kt_per * BK;
    p.splits = splits;
    ONMF_TRY(onmf_reserve(c, *slab, (size_t)splits * M * N * sizeof(float), "split-K slabs"));
    p.slab = (float*)slab->p;
    {
        const dim3 grid(tiles, 1, splits);
        int rc;
        if (a_km && b_kn) rc = run<true, true>(c, p, grid, s);
        else if (a_km && !b_kn) rc = run<true, false>(c, p, grid, s);
        else if (!a_km && b_kn) rc = run<false, true>(c, p, grid, s);
        else rc = run<false, false>(c, p, grid, s);
        ONMF_TRY(rc);
    }
    const size_t mn = (size_t)M * N;
    int blocks = (int)((mn + 63) / 64);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.slab, splits, M, N, alpha, beta1, Cin1,
                       C1, ldc1, N1, N1p, beta2, Cin2, C2, ldc2);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

#ifdef ONMF_TRACE_WG
// kind 0: copy the records out and reset the count; returns the number of records
extern "C" int onmf_debug_gemm_trace(unsigned long long* host, int max_records) {
    unsigned n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_wg_trace_n), sizeof(n)) != hipSuccess) return -1;
    if (n > 65536u) n = 65536u;
    if ((int)n > max_records) n = (unsigned)max_records;
    if (n && hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_trace), (size_t)n * 32) != hipSuccess) return -1;
    const unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace_n), &zero, sizeof(zero)) != hipSuccess) return -1;
    return (int)n;
}
#endif
