// Device-side screen of the sweep loop's stopping test (src/onmf.py:252, :265 of the reference):
//     continue while  dist = sigma_max(H_new - H_old) / sigma_max(H_old)  >  stopping_diff.
// Evaluating dist exactly needs two r x r Gram matrices and their top eigenvalues per sweep, as
// much arithmetic as the sweep itself.  The sweep kernel instead produces, as by-products,
//     lb2  = max_p || u_p^T (H_new - H_old) ||^2     for 16 unit probe vectors u_p   (<= sigma_max^2)
//     cw   = max_k y_k / v_k,  y = H (H^T v), v > 0   (Collatz-Wielandt: >= sigma_max(H)^2, because
//                                                      H H^T is entrywise nonnegative)
// and this file's one-workgroup kernel turns them into a three-way verdict per sweep:
//     lower bound of dist > tol  -> keep sweeping (certain)
//     upper bound of dist <= tol -> stop (certain)
//     otherwise                  -> inconclusive: the host evaluates the test in full
// Bounds are rigorous for ANY probes, so the verdicts never differ from the full test; good probes
// only make "inconclusive" rare.  The probes follow the dominant left singular direction of
// H_new - H_old by one step of subspace iteration per sweep (Z accumulated by the sweep kernel),
// re-orthonormalised here with a Cholesky-QR, the top Ritz direction placed in probe 0.
#include "common.h"

namespace {

__device__ __forceinline__ int swz(int row) { return (-(row >> 2)) & 3; }
__device__ __forceinline__ int blk_off(int i, int j) { return i * 16 + 4 * ((j >> 2) ^ swz(i)) + (j & 3); }

__device__ __forceinline__ float hash_unit(unsigned x) {      // deterministic value in [-1, 1)
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

__global__ __launch_bounds__(256) void screen_init_kernel(float* base, int r) {
    const int tid = threadIdx.x;
    const float scale = 1.0f / sqrtf((float)r);              // |u_p|^2 <= r * scale^2 = 1
    for (int e = tid; e < 4096; e += 256) base[ScreenLayout::UBLK + e] = 0.f;
    for (int e = tid; e < 4096; e += 256) base[ScreenLayout::ZACC + e] = 0.f;
    __syncthreads();
    for (int k = tid; k < 256; k += 256) {
        base[ScreenLayout::V + k] = (k < r) ? 1.0f : 0.f;
        base[ScreenLayout::YACC + k] = 0.f;
        if (k < r) {
            const int K = k >> 4, j = k & 15;
            for (int p = 0; p < 16; ++p)
                base[ScreenLayout::UBLK + K * 256 + blk_off(p, j)] = scale * hash_unit(k * 16 + p + 1);
        }
    }
    if (tid < 16) base[ScreenLayout::P2ACC + tid] = 0.f;
    if (tid < 4) base[ScreenLayout::SUMS + tid] = 0.f;
    if (tid == 0) {
        reinterpret_cast<int*>(base)[ScreenLayout::FLAG] = 0;
        base[ScreenLayout::UBPREV] = -1.0f;
    }
    for (int e = tid; e < 4 * SCREEN_MAX_SWEEPS; e += 256) base[ScreenLayout::REC + e] = 0.f;
}

__global__ void screen_clear_flag_kernel(float* base) {
    if (threadIdx.x == 0) reinterpret_cast<int*>(base)[ScreenLayout::FLAG] = 0;
}


// Power step of the probe subspace on a sample of the columns (16 of every `stride` columns):
//     Z += dH_S (U^T dH_S)^T,   dH = H_new - H_old
// One wave per sampled 16-column tile, registers in the MFMA layout of the sweep kernel (lane l:
// column l & 15, rows 16J + 4(l >> 4) + t).  P = U^T dH_S contracts over rows (native layout); the
// second product contracts over the 16 columns, so dH blocks and P pass through a 16x17 LDS tile to
// put the column index on the MFMA's k lanes.  Any U yields a valid bound, so a sample suffices.
template <int NB>
__global__ __launch_bounds__(64) void screen_refresh_kernel(int r, int b, int stride,
                                                            const float* __restrict__ Hnew,
                                                            const float* __restrict__ Hold, float* base) {
    __shared__ float sd[16 * 17];
    __shared__ float sp[16 * 17];
    if (reinterpret_cast<const int*>(base)[ScreenLayout::FLAG] != 0) return;
    const float* ublk = base + ScreenLayout::UBLK;
    float* zacc = base + ScreenLayout::ZACC;
    const int lane = threadIdx.x;
    const int c = lane & 15, g = lane >> 4;
    const int col = blockIdx.x * stride + c;
    const bool cvalid = col < b;
    const int sw_c = swz(c);
    const unsigned nbytes = (unsigned)r * (unsigned)b * 4u;
    const auto rs_n = __builtin_amdgcn_make_buffer_rsrc((void*)Hnew, 0, nbytes, 0x00020000);
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)Hold, 0, nbytes, 0x00020000);
    const int voff = (4 * g * b + min(col, b - 1)) * 4;
    float dh[NB][4];
    f32x4 pacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int J = 0; J < NB; ++J) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float hn = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_n, voff, (16 * J + t) * b * 4, 0));
            const float ho = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_o, voff, (16 * J + t) * b * 4, 0));
            dh[J][t] = cvalid ? (hn - ho) : 0.f;
        }
        const float4 u4 = *reinterpret_cast<const float4*>(ublk + J * 256 + c * 16 + 4 * (g ^ sw_c));
        pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(u4.x, dh[J][0], pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(u4.y, dh[J][1], pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(u4.z, dh[J][2], pacc, 0, 0, 0);
        pacc = __builtin_amdgcn_mfma_f32_16x16x4f32(u4.w, dh[J][3], pacc, 0, 0, 0);
    }
    // pacc[t] = P[probe 4g+t][column c]
#pragma unroll
    for (int t = 0; t < 4; ++t) sp[(4 * g + t) * 17 + c] = pacc[t];
    __builtin_amdgcn_wave_barrier();
    float pb[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) pb[s4] = sp[c * 17 + 4 * s4 + g];       // B[k = col 4s+g][j = probe c]
#pragma unroll
    for (int K = 0; K < NB; ++K) {
#pragma unroll
        for (int t = 0; t < 4; ++t) sd[(4 * g + t) * 17 + c] = dh[K][t];
        __builtin_amdgcn_wave_barrier();
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)                                     // A[i = row c][k = col 4s+g]
            z = __builtin_amdgcn_mfma_f32_16x16x4f32(sd[c * 17 + 4 * s4 + g], pb[s4], z, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
        // z[t] = Z[row 16K+4g+t][probe c]
#pragma unroll
        for (int t = 0; t < 4; ++t) atomicAdd(&zacc[(16 * K + 4 * g + t) * 16 + c], z[t]);
    }
}

template <int NB>
int run_refresh(onmf_ctx* c, int r, int b, const float* Hnew, const float* Hold, float* base,
                       hipStream_t s) {
    const int stride = 256;
    const int tiles = (b + stride - 1) / stride;
    hipLaunchKernelGGL((screen_refresh_kernel<NB>), dim3(tiles), dim3(64), 0, s, r, b, stride, Hnew, Hold, base);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

__device__ __forceinline__ float block_max(float v, float* red) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// One workgroup of 256 threads; thread k owns row k of the r_pad <= 256 rows.
__global__ __launch_bounds__(256) void screen_decide_kernel(float* base, int r, int sweep_index,
                                                            float tol, float rank_cap) {
    __shared__ float Z[256 * 17];
    __shared__ float M[16 * 16];      // Z^T Z
    __shared__ float Gy[16 * 16];     // Gram of Y = [t, Z_0 .. Z_14], then its Cholesky factor
    __shared__ float avec[16], mvec[16], nrm[16], red[4];
    __shared__ int verdict_s, entry_s;
    int* flagp = reinterpret_cast<int*>(base) + ScreenLayout::FLAG;
    const int tid = threadIdx.x;
    if (tid == 0) entry_s = *flagp;
    __syncthreads();
    if (entry_s != 0) return;                               // loop already ended (uniform exit)
    float* zacc = base + ScreenLayout::ZACC;
    float* yacc = base + ScreenLayout::YACC;
    float* vvec = base + ScreenLayout::V;
    const double* sums = reinterpret_cast<const double*>(base + ScreenLayout::SUMS);

    // ---- verdict for this sweep ----
    if (tid == 0) {
        const double dsum = sums[0], hsum = sums[1];
        const double fD = sqrt(dsum), fH = sqrt(hsum);
        float lb2 = 0.f;
        for (int p = 0; p < 16; ++p) lb2 = fmaxf(lb2, base[ScreenLayout::P2ACC + p]);
        const float ubp = base[ScreenLayout::UBPREV];
        double ubH = fH * (1.0 + 1e-5);                       // sigma_max <= Frobenius norm
        if (ubp > 0.f && (double)ubp < ubH) ubH = ubp;
        int verdict;                                          // 0 continue, 1 stop, 2 inconclusive
        double lower = 0.0, upper = 1e300;
        if (hsum == 0.0) {            // dist = 0/0 (NaN: loop ends) or x/0 (+inf: continues)
            verdict = (dsum == 0.0) ? 1 : 0;
            lower = upper = (dsum == 0.0) ? 0.0 : 1e300;
        } else if (dsum == 0.0) {     // dist = 0
            verdict = (0.0 > (double)tol) ? 0 : 1;
            lower = upper = 0.0;
        } else {
            const double lbs = fmax(sqrt((double)lb2) * (1.0 - 2e-4), fD / sqrt((double)rank_cap) * (1.0 - 1e-4));
            lower = lbs / ubH;
            upper = fD * (1.0 + 1e-4) / (fH / sqrt((double)rank_cap));
            verdict = (lower > (double)tol) ? 0 : ((upper <= (double)tol) ? 1 : 2);
        }
        verdict_s = verdict;
        float* rec = base + ScreenLayout::REC + 4 * sweep_index;
        rec[0] = (float)lower;
        rec[1] = (float)fmin(upper, 3.0e38);
        rec[2] = (float)(verdict + 1);                        // 0 = sweep did not run
        rec[3] = (float)(fD / fmax(fH, 1e-300));
        if (verdict != 0) *flagp = verdict;
    }

    // ---- Collatz-Wielandt bound of sigma_max(H_new) for the next sweep's test, next v ----
    const float yk = (tid < r) ? yacc[tid] : 0.f;
    const float vk = (tid < r) ? vvec[tid] : 1.f;
    const float ratio = (tid < r) ? yk / vk : 0.f;            // v > 0 by construction
    const float cw = block_max(ratio, red);
    const float ymax = block_max(yk, red);
    if (tid == 0) base[ScreenLayout::UBPREV] = (cw > 0.f) ? sqrtf(cw) * (1.0f + 2e-4f) : -1.0f;
    if (tid < r) vvec[tid] = (ymax > 0.f) ? (yk / ymax + 1e-3f) : 1.0f;
    if (tid < 256) yacc[tid] = 0.f;

    // ---- refresh the probes: Z -> [top direction, Z_0..Z_14] -> Cholesky-QR -> unit columns ----
    float zrow[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        zrow[p] = (tid < r) ? zacc[tid * 16 + p] : 0.f;
        Z[tid * 17 + p] = zrow[p];
        zacc[tid * 16 + p] = 0.f;
    }
    if (tid < 16) base[ScreenLayout::P2ACC + tid] = 0.f;
    __syncthreads();
    {
        const int p = tid >> 4, q = tid & 15;
        float acc = 0.f;
        for (int k = 0; k < 256; ++k) acc = fmaf(Z[k * 17 + p], Z[k * 17 + q], acc);
        M[p * 16 + q] = acc;
    }
    __syncthreads();
    float trace = 0.f;
    for (int p = 0; p < 16; ++p) trace += M[p * 16 + p];
    const bool have_z = (trace > 0.f) && (trace < 3.0e38f);
    if (tid == 0) {
        volatile double* s2 = reinterpret_cast<volatile double*>(base + ScreenLayout::SUMS);
        s2[0] = 0.0;
        s2[1] = 0.0;
    }
    if (!have_z) return;                                      // keep the old probes (uniform exit)

    // top eigenvector of M by power iteration (one wave, lane p owns component p)
    if (tid < 64) {
        float a = (tid < 16) ? 0.25f : 0.f;
        for (int it = 0; it < 48; ++it) {
            if (tid < 16) avec[tid] = a;
            __builtin_amdgcn_wave_barrier();
            float m = 0.f;
            if (tid < 16)
                for (int q = 0; q < 16; ++q) m = fmaf(M[tid * 16 + q], avec[q], m);
            float n2 = m * m;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) n2 += __shfl_xor(n2, off);
            a = (n2 > 0.f) ? m * rsqrtf(n2) : a;
            __builtin_amdgcn_wave_barrier();
        }
        if (tid < 16) avec[tid] = a;
        __builtin_amdgcn_wave_barrier();
        if (tid < 16) {
            float m = 0.f;
            for (int q = 0; q < 16; ++q) m = fmaf(M[tid * 16 + q], avec[q], m);
            mvec[tid] = m;                                    // (M a)_p
        }
    }
    __syncthreads();
    // Gram of Y = [t = Z a, Z_0 .. Z_14]
    {
        const int p = tid >> 4, q = tid & 15;
        float v;
        if (p == 0 && q == 0) {
            v = 0.f;
            for (int i = 0; i < 16; ++i) v = fmaf(avec[i], mvec[i], v);
        } else if (p == 0) v = mvec[q - 1];
        else if (q == 0) v = mvec[p - 1];
        else v = M[(p - 1) * 16 + (q - 1)];
        if (p == q) v += 1e-6f * trace;                       // keeps the factorisation defined
        Gy[p * 16 + q] = v;
    }
    __syncthreads();
    if (tid < 64) {
        // Cholesky Gy = L L^T, right-looking, lane i holds row i in registers
        const int li = tid & 15;
        float arow[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) arow[q] = Gy[li * 16 + q];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float ajj = __shfl(arow[j], j, 64);
            const float dgn = sqrtf(fmaxf(ajj, 1e-12f * trace));
            const float lij = arow[j] / dgn;                  // L[i][j] for i >= j (lane j: the pivot)
            arow[j] = (li == j) ? dgn : lij;
#pragma unroll
            for (int k = j + 1; k < 16; ++k) {
                const float lkj = __shfl(lij, k, 64);
                arow[k] = fmaf(-lij, lkj, arow[k]);
            }
        }
        if (tid < 16) {
#pragma unroll
            for (int q = 0; q < 16; ++q) Gy[li * 16 + q] = arow[q];
        }
    }
    if (tid < 16) nrm[tid] = 0.f;
    __syncthreads();
    // row k of U = row k of Y times L^-T (forward substitution along the row)
    float yrow[16], urow[16];
    yrow[0] = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) yrow[0] = fmaf(zrow[p], avec[p], yrow[0]);
#pragma unroll
    for (int p = 1; p < 16; ++p) yrow[p] = zrow[p - 1];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float s = yrow[j];
#pragma unroll
        for (int i = 0; i < j; ++i) s -= urow[i] * Gy[j * 16 + i];
        urow[j] = s / Gy[j * 16 + j];
    }
    // exact column norms so that every probe is a unit vector (the bound needs |u_p| <= 1)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        float s = urow[j] * urow[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if ((tid & 63) == 0) atomicAdd(&nrm[j], s);
    }
    __syncthreads();
    if (tid < 16 * ((r + 15) / 16)) {
        const int K = tid >> 4, jj = tid & 15;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float n2 = nrm[p];
            const float u = (n2 > 0.f && tid < r) ? urow[p] * rsqrtf(n2) * (1.0f - 1e-6f) : 0.f;
            base[ScreenLayout::UBLK + K * 256 + blk_off(p, jj)] = u;
        }
    }
}

}  // namespace

ScreenDev screen_pointers(float* base) {
    ScreenDev sc;
    sc.ublk = base + ScreenLayout::UBLK;
    sc.v = base + ScreenLayout::V;
    sc.zacc = base + ScreenLayout::ZACC;
    sc.p2acc = base + ScreenLayout::P2ACC;
    sc.yacc = base + ScreenLayout::YACC;
    sc.sums = reinterpret_cast<double*>(base + ScreenLayout::SUMS);
    sc.flag = reinterpret_cast<const int*>(base) + ScreenLayout::FLAG;
    return sc;
}

int launch_screen_init(onmf_ctx* c, float* base, int r, hipStream_t s) {
    hipLaunchKernelGGL(screen_init_kernel, dim3(1), dim3(256), 0, s, base, r);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

int launch_screen_decide(onmf_ctx* c, float* base, int r, int sweep_index, float tol, float rank_cap,
                         hipStream_t s) {
    hipLaunchKernelGGL(screen_decide_kernel, dim3(1), dim3(256), 0, s, base, r, sweep_index, tol, rank_cap);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}

int launch_screen_refresh(onmf_ctx* c, float* base, int r, int b, const float* Hnew, const float* Hold,
                          hipStream_t s) {
    switch ((r + 15) / 16) {
#define RCASE(N) case N: return run_refresh<N>(c, r, b, Hnew, Hold, base, s)
        RCASE(1); RCASE(2); RCASE(3); RCASE(4); RCASE(5); RCASE(6); RCASE(7); RCASE(8);
        RCASE(9); RCASE(10); RCASE(11); RCASE(12); RCASE(13); RCASE(14); RCASE(15); RCASE(16);
#undef RCASE
    }
    return onmf_fail(c, ONMF_ERR_ARG, "screen refresh: unsupported r=%d", r);
}

int launch_screen_clear_flag(onmf_ctx* c, float* base, hipStream_t s) {
    hipLaunchKernelGGL(screen_clear_flag_kernel, dim3(1), dim3(64), 0, s, base);
    HIP_TRY(c, hipGetLastError());
    return ONMF_OK;
}
